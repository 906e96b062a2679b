"""The RCCL entry points of the C ABI (sr_comm_*, csrc/sr_comm.cpp) on the one GPU a test box has: RCCL refuses two ranks
on one device, so what can execute here is a world-size-1 communicator -- creation from a unique id, a grouped batch of
ncclSend / ncclRecv to itself on the context's stream (the same code path peers take), the metric all-reduce, and the
plan-driven sr_comm_exchange_tile_rows (no transfers at world 1).  World sizes 2-8 are covered on the host side by
tests/test_planner.py::test_exchange_xfers_match_plan (the exact batches) and the gloo rehearsals."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_comm_world1_self_exchange(ctx, rng):
    import _native
    import device_pipeline as dp
    comm = _native.Comm(ctx, _native.comm_unique_id(), 1, 0)
    try:
        a = rng.integers(0, 256, 3_000_001, dtype=np.uint8)
        b = rng.integers(0, 256, 70_001, dtype=np.uint8)
        da, db = ctx.upload(a), ctx.upload(b)
        ra, rb = ctx.alloc(a.nbytes), ctx.alloc(b.nbytes)
        try:
            comm.exchange([(0, da.ptr, a.nbytes), (0, db.ptr, b.nbytes)], [(0, ra.ptr, a.nbytes), (0, rb.ptr, b.nbytes)])
            ctx.sync()
            assert np.array_equal(ctx.download(ra.ptr, a.shape, np.uint8), a)
            assert np.array_equal(ctx.download(rb.ptr, b.shape, np.uint8), b)
            sums = np.array([1.5, -2.25, 3e300, 7.0])
            ds = ctx.upload(sums)
            comm.allreduce_f64(ds.ptr, 4)
            ctx.sync()
            assert np.array_equal(ctx.download(ds.ptr, (4,), np.float64), sums)
            ds.free()
            with pytest.raises((_native.SrNativeError, ValueError)):
                comm.exchange([(1, da.ptr, 16)], [])                 # peer outside the communicator
        finally:
            for d in (da, db, ra, rb):
                d.free()
        geo = dp.workload_geometry("4MP")
        plan = dp.make_exchange_plan(geo, 1)
        n = len(geo.rects)
        tiles = [ctx.alloc(w * h * geo.cn) for (_, _, w, h) in geo.rects]
        try:
            comm.exchange_tile_rows(geo.rects, geo.cn, plan.need, plan.owners, [t.ptr for t in tiles],
                                    [w * geo.cn for (_, _, w, _) in geo.rects], [0] * n)
            ctx.sync()
        finally:
            for t in tiles:
                t.free()
    finally:
        comm.close()


def _lcg_bytes(seed, n):
    """The C example's generator, vectorised: s_k = a^k s_0 + c (a^(k-1) + ... + 1) mod 2^32, value = s_k >> 24."""
    a, c = np.uint32(1664525), np.uint32(1013904223)
    with np.errstate(over="ignore"):
        ak = np.cumprod(np.full(n, a, dtype=np.uint32), dtype=np.uint32)           # a^1 .. a^n (wraps mod 2^32)
        geo = np.cumsum(np.concatenate([[np.uint32(1)], ak[:-1]]), dtype=np.uint32)   # 1 + a + ... + a^(k-1)
        s = ak * np.uint32(seed) + c * geo
    return (s >> np.uint32(24)).astype(np.int64)


def test_c_host_example(ctx, tmp_path):
    """examples/strip_host.c -- a plain C99 host that includes nothing but include/sr_hip.h -- is compiled with gcc, linked
    against libsrhip.so and run on the GPU (one rank: plan, RCCL communicator, exchange, strip blend, assessment,
    all-reduce); its metric sums and the hash of its canvas must equal what the Python host computes for the same
    synthetic images."""
    import json
    import os
    import shutil
    import subprocess
    import _native
    if shutil.which("gcc") is None:
        pytest.skip("no gcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(_native.LIB_PATH)
    exe = str(tmp_path / "strip_host")
    subprocess.run(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                    os.path.join(root, "examples", "strip_host.c"), "-L", libdir, "-lsrhip", "-Wl,-rpath," + libdir, "-o", exe],
                   check=True, capture_output=True, text=True, timeout=300)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
    got = json.loads(r.stdout.strip().splitlines()[-1])
    # the same images, the same path through the Python host
    tile, ov, grid, cn = 512, 100, 2, 3
    side = grid * (tile - ov) + ov
    xs = np.arange(side * cn) // cn
    base = 96 + (xs[None, :] * 5 + np.arange(side)[:, None] * 3) % 64
    img = (base + _lcg_bytes(20260313, side * side * cn).reshape(side, side * cn) % 25).astype(np.uint8)
    ref = (base + _lcg_bytes(42, side * side * cn).reshape(side, side * cn) % 25).astype(np.uint8)
    rects = [((t % grid) * (tile - ov), (t // grid) * (tile - ov), tile, tile) for t in range(grid * grid)]
    d_img, d_ref = ctx.upload(img), ctx.upload(ref)
    tiles = [ctx.alloc(tile * tile * cn) for _ in rects]
    canvas = ctx.alloc(side * side * cn)
    sums = ctx.alloc(32)
    try:
        ctx.memset(canvas.ptr, 0, side * side * cn)
        ctx.tile_extract(d_img.ptr, side, side, cn, side * cn, rects, [t.ptr for t in tiles], [tile * cn] * len(rects))
        plan = _native.BlendPlan(ctx, rects, cn, side, side, 6, "cosine")
        plan.blend([t.ptr for t in tiles], [tile * cn] * len(rects), canvas.ptr, side * cn)
        ctx.assess_u8_async(d_ref.ptr, side * cn, canvas.ptr, side * cn, side, side, cn, sums.ptr)
        ctx.sync()
        want = ctx.download(sums.ptr, (4,), np.float64)
        out = ctx.download(canvas.ptr, (side * side * cn,), np.uint8)
        plan.close()
    finally:
        for b in [d_img, d_ref, canvas, sums] + tiles:
            b.free()
    assert got["rows"] == [0, side] and got["world"] == 1
    assert [got["sse"], got["ssim_uniform"], got["ssim_gauss"], got["ssim_simple"]] == want.tolist()
    h = 1469598103934665603
    for v in out.tolist():
        h = ((h ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert got["strip_fnv1a"] == f"{h:016x}"
    assert want[0] > 0 and 0.0 < want[2] / (side * side) < 1.0


def test_sharded_blend_rejects_mismatched_arguments(ctx, rng):
    """sr_laplacian_blend_sharded cross-checks its handles before anything is posted: a plan made on another context (the
    exchange and the blend would run on different streams), a tile count or channel count that is not the plan's, a
    destroyed plan.  (Stride / owner / buffer checks of received tiles need world > 1: tests/test_planner.py drives them
    through the host-only sr_sharded_tile_bases.)  The well-formed call equals the plain strip blend."""
    import _native
    import device_pipeline as dp
    geo = dp.grid_geometry(tile_w=200, tile_h=160, rows=2, cols=2, ov_x=40)
    n, cn = len(geo.rects), geo.cn
    xp = dp.make_exchange_plan(geo, 1)
    comm = _native.Comm(ctx, _native.comm_unique_id(), 1, 0)
    other = _native.Context(0)
    tiles = [ctx.upload(rng.integers(0, 256, (h, w * cn), dtype=np.uint8)) for (_, _, w, h) in geo.rects]
    canvas, canvas2 = ctx.alloc(geo.canvas_h * geo.canvas_w * cn), ctx.alloc(geo.canvas_h * geo.canvas_w * cn)
    ptrs, strides = [t.ptr for t in tiles], [w * cn for (_, _, w, _) in geo.rects]
    plan = _native.BlendPlan(ctx, geo.rects, cn, geo.canvas_h, geo.canvas_w, geo.levels, geo.weight_type)
    foreign = _native.BlendPlan(other, geo.rects, cn, geo.canvas_h, geo.canvas_w, geo.levels, geo.weight_type)
    try:
        args = (geo.rects, cn, xp.need, xp.owners, ptrs, strides, [0] * n, canvas.ptr, geo.canvas_w * cn)
        comm.blend_sharded(plan, *args)
        plan.blend(ptrs, strides, canvas2.ptr, geo.canvas_w * cn)
        ctx.sync()
        size = (geo.canvas_h * geo.canvas_w * cn,)
        assert np.array_equal(ctx.download(canvas.ptr, size, np.uint8), ctx.download(canvas2.ptr, size, np.uint8))
        with pytest.raises(ValueError, match="another context"):
            comm.blend_sharded(foreign, *args)
        with pytest.raises(_native.SrShapeError):                                   # fewer tiles than the plan indexes
            comm.blend_sharded(plan, geo.rects[:3], cn, [xp.need[0][:3]], xp.owners[:3], ptrs[:3], strides[:3], [0] * 3,
                               canvas.ptr, geo.canvas_w * cn)
        with pytest.raises(_native.SrShapeError):                                   # channel count
            comm.blend_sharded(plan, geo.rects, 1, xp.need, xp.owners, ptrs, strides, [0] * n, canvas.ptr, geo.canvas_w * cn)
        stale = type("StalePlan", (), {"handle": foreign.handle})()         # the raw handle of a plan that is then destroyed
        foreign.close()
        with pytest.raises(ValueError, match="destroyed plan"):
            comm.blend_sharded(stale, *args)
    finally:
        plan.close()
        foreign.close()
        for b in tiles + [canvas, canvas2]:
            b.free()
        other.close()
        comm.close()


def test_one_rccl_and_one_rocm_smi_mapped(ctx):
    """Regression guard for the round-2 exit abort (two copies of librocm_smi64's statics in one process): after the C ABI's
    communicator AND torch.distributed's RCCL group have been initialised, exactly one librccl and at most one
    librocm_smi64 are mapped into this process."""
    import os
    import re
    import torch
    import torch.distributed as dist
    import _native
    comm = _native.Comm(ctx, _native.comm_unique_id(), 1, 0)
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        t = torch.ones(4, device="cuda")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        libs = set()
        with open("/proc/self/maps") as f:
            for line in f:
                m = re.search(r"(/\S*lib(rccl|rocm_smi64)[^/\s]*\.so[^/\s]*)", line)
                if m:
                    libs.add(os.path.realpath(m.group(1)))
        rccl = sorted(p for p in libs if "librccl" in p)
        smi = sorted(p for p in libs if "librocm_smi64" in p)
        assert len(rccl) == 1, rccl
        assert len(smi) <= 1, smi
    finally:
        if created:
            dist.destroy_process_group()
        comm.close()


def test_bench_two_rank_rehearsal_matches_one_rank():
    """The driver's scale command has a tested twin: `SR_DIST_BACKEND=gloo python bench.py --gpus 2 --workload 4MP` -- the
    self-launcher plus two ranks sharing this GPU (3 processes; rows staged through the host, flagged as a rehearsal) --
    prints ONE JSON line with n_gpus 2 / strips2, per-rank exchange records, and exactly the scores of the 1-rank run."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    common = ["--workload", "4MP", "--steps", "2", "--warmup", "1", "--sweep", "none", "--no-pcie", "--no-cpu-baseline"]

    def run(extra_env, args):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args + common, env={**env, **extra_env},
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        return json.loads(lines[0])

    two = run({"SR_DIST_BACKEND": "gloo"}, ["--gpus", "2", "--deadline-s", "500"])
    one = run({}, ["--gpus", "1"])
    assert two["n_gpus"] == 2 and two["config"]["parallelism"] == "strips2" and two["scaling"] == "strong"
    assert "REHEARSAL" in two["data"] and one["data"] == "synthetic"
    assert two["quality"]["psnr"] == one["quality"]["psnr"]
    for k in ("ssim_uniform", "ssim_gauss", "ssim_simple"):
        assert two["quality"][k] == pytest.approx(one["quality"][k], rel=1e-12)
    b = two["config"]["strip_bounds"]
    assert b[0] == 0 and b[-1] == 1640 and len(b) == 3
    assert [r["rank"] for r in two["ranks"]] == [0, 1]
    assert sum(r["exchange_bytes_in"] for r in two["ranks"]) == sum(r["exchange_bytes_out"] for r in two["ranks"]) > 0
    assert all(r["strip_rows"] == [b[i], b[i + 1]] for i, r in enumerate(two["ranks"]))


def _bench_json(extra_env, args, timeout=900):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    common = ["--steps", "2", "--warmup", "1", "--sweep", "none", "--no-pcie", "--no-cpu-baseline"]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args + common, env={**env, **extra_env},
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_rank_batch_rehearsal_150mp():
    """BASELINE config 4 (batch of images at 150 MP, one per GPU, no RCCL on the data path) with two gloo ranks on this GPU
    (launcher + 2 ranks = 3 processes): weak scaling, every rank's own line gathered, no exchange bytes, and rank 0's
    scores are exactly the one-rank run's (its image is the same)."""
    two = _bench_json({"SR_DIST_BACKEND": "gloo"}, ["--gpus", "2", "--mode", "batch", "--workload", "150MP", "--deadline-s", "800"])
    one = _bench_json({}, ["--gpus", "1", "--workload", "150MP"])
    assert two["n_gpus"] == 2 and two["config"]["parallelism"] == "batch2" and two["scaling"] == "weak"
    assert "REHEARSAL" in two["data"]
    assert [r["rank"] for r in two["ranks"]] == [0, 1]
    assert all(r["exchange_bytes_in"] == 0 and r["exchange_bytes_out"] == 0 and r["collectives_on_data_path"] == 0 for r in two["ranks"])
    assert two["ranks"][0]["quality"] == one["quality"] == two["quality"]
    assert two["ranks"][1]["quality"]["psnr"] != one["quality"]["psnr"]          # the second rank's image is another noise draw
    assert "strip_bounds" not in two["config"]


def test_bench_two_rank_strips_rehearsal_200mp_kd():
    """BASELINE config 5's geometry (non-uniform k-d tiling of the 200 MP canvas) in strips over two gloo ranks on this GPU:
    the scores equal the one-rank run (PSNR exactly, SSIM to the summation order), every tile has an owner, the rows
    that cross the strip boundary are a small part of the tiles."""
    two = _bench_json({"SR_DIST_BACKEND": "gloo"}, ["--gpus", "2", "--workload", "200MP-kd", "--deadline-s", "800"])
    one = _bench_json({}, ["--gpus", "1", "--workload", "200MP-kd"])
    assert two["n_gpus"] == 2 and two["config"]["parallelism"] == "strips2" and two["scaling"] == "strong"
    assert two["quality"]["psnr"] == one["quality"]["psnr"]
    for k in ("ssim_uniform", "ssim_gauss", "ssim_simple"):
        assert two["quality"][k] == pytest.approx(one["quality"][k], rel=1e-12)
    b = two["config"]["strip_bounds"]
    assert b[0] == 0 and b[-1] == 11550 and len(b) == 3
    assert sum(r["tiles_owned"] for r in two["ranks"]) == 32
    moved = sum(r["exchange_bytes_in"] for r in two["ranks"])
    assert 0 < moved == sum(r["exchange_bytes_out"] for r in two["ranks"]) < 0.25 * two["config"]["tile_pixels"] * 3

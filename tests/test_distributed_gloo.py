"""CPU rehearsal of the N > 1 path (world_size 2 and 3, gloo): the strip partition, tile ownership,
the exchange plan and the point-to-point row exchange are the product code of device_pipeline.py;
only the per-strip arithmetic is done by the CPU oracle here (no GPU in this container).

The check: after the exchange every rank holds, for each tile, exactly the rows the window planner says
its strip needs (everything else poisoned).  Blending those with the oracle must reproduce the rows of
the monolithic result inside the rank's strip -- which proves the halo plan is sufficient and the
exchange delivers the right bytes to the right place."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "super-resolution-system_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import device_pipeline as dp
    from oracle import oracle_c as oc
    oc.set_threads(2)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        geo = dp.grid_geometry(tile_w=260, tile_h=300, rows=3, cols=2, ov_x=60, levels=6)
        rng = np.random.default_rng(99)                     # same tiles on every rank (the "SR output")
        tiles = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for (_, _, w, h) in geo.rects]
        xp = dp.make_exchange_plan(geo, world)
        assert sorted(set(xp.owners)) == sorted(set(xp.owners) & set(range(world)))
        local = {t: torch.from_numpy(tiles[t].reshape(tiles[t].shape[0], -1).copy())
                 for t, o in enumerate(xp.owners) if o == rank}
        recv = {t: torch.full((b - a, geo.rects[t][2] * 3), 0x55, dtype=torch.uint8) for (_, t, a, b) in xp.recvs(rank)}
        for w_ in dp.exchange_tile_rows(xp, rank, local, recv):
            w_.wait()
        # assemble what this rank now holds: needed rows real, everything else poison
        held = []
        for t, (x, y, w, h) in enumerate(geo.rects):
            a, b = xp.need[rank][t]
            arr = np.full((h, w, 3), 0xAA, np.uint8)
            if a < b:
                src = local[t][a:b] if t in local else recv[t]
                arr[a:b] = src.numpy().reshape(b - a, w, 3)
            held.append(arr)
        pos = [(y, x) for (x, y, _, _) in geo.rects]
        shape = (geo.canvas_h, geo.canvas_w)
        mine = oc.laplacian_fusion(held, pos, shape, geo.levels, geo.weight_type)
        full = oc.laplacian_fusion(tiles, pos, shape, geo.levels, geo.weight_type)
        r0, r1 = xp.rows[rank]
        ok_rows = bool(np.array_equal(mine[r0:r1], full[r0:r1]))
        # metric partial sums add up over the strips
        ref = np.clip(full.astype(np.int16) + 3, 0, 255).astype(np.uint8)
        s0, s1 = xp.bounds[rank], xp.bounds[rank + 1]
        part = torch.tensor([float(((ref[s0:s1].astype(np.int64) - full[s0:s1].astype(np.int64)) ** 2).sum())],
                            dtype=torch.float64)
        dist.all_reduce(part)
        total = float(((ref.astype(np.int64) - full.astype(np.int64)) ** 2).sum())
        recv_bytes = xp.bytes_received(rank, geo)
        # the final image on request: every rank contributes exactly its strip, rank 0 ends up with the monolithic canvas
        canvas = torch.full((geo.canvas_h, geo.canvas_w * 3), 0x77, dtype=torch.uint8)
        canvas[s0:s1] = torch.from_numpy(mine[s0:s1].reshape(s1 - s0, -1))
        whole = dp.gather_strips(canvas, xp.bounds, rank, None, dst=0)
        ok_gather = (whole is None) if rank != 0 else bool(np.array_equal(whole.numpy().reshape(full.shape), full))
        q.put((rank, ok_rows and ok_gather, float(part.item()) == total, recv_bytes, None))
    except Exception as exc:  # noqa: BLE001
        import traceback
        q.put((rank, False, False, 0, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_strip_exchange_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok_rows, ok_sum, recv_bytes, err in sorted(results):
        assert err is None, err
        assert ok_rows, f"rank {rank}: strip rows differ from the monolithic blend"
        assert ok_sum, f"rank {rank}: all-reduced SSE differs"
    assert any(r[3] > 0 for r in results), "nothing was exchanged"


def test_exchange_plan_properties():
    import device_pipeline as dp
    geo = dp.workload_geometry("200MP")
    assert (geo.canvas_w, geo.canvas_h, len(geo.rects)) == (17320, 11550, 25)
    assert dp.workload_geometry("100MP").canvas_w == 12246 and dp.workload_geometry("150MP").canvas_h == 10002
    assert (dp.workload_geometry("4MP").canvas_w, dp.workload_geometry("4MP").canvas_h) == (2459, 1640)
    for world in (1, 2, 4, 8):
        xp = dp.make_exchange_plan(geo, world)
        assert xp.bounds[0] == 0 and xp.bounds[-1] == geo.canvas_h and len(xp.bounds) == world + 1
        # every (receiver, tile) need is served by exactly one send of the owner, same rows
        sends = {(r, peer, t): (a, b) for r in range(world) for (peer, t, a, b) in xp.sends(r)}
        recvs = {(peer, r, t): (a, b) for r in range(world) for (peer, t, a, b) in xp.recvs(r)}
        assert sends == recvs
        for r in range(world):
            for t, (a, b) in enumerate(xp.need[r]):
                x, y, w, h = geo.rects[t]
                lo, hi = max(xp.rows[r][0] - y, 0), min(xp.rows[r][1] - y, h)
                if lo < hi:
                    below, above = dp._native.pyramid_halo(geo.levels)
                    assert a <= lo and hi <= b and (lo - a) <= below and (b - hi) <= above
                else:
                    assert a >= b
        if world == 1:
            assert not xp.sends(0) and not xp.recvs(0) and all(n == (0, geo.rects[t][3]) for t, n in enumerate(xp.need[0]))
        if world > 1:
            def max_pair(plan):
                pair = {}
                for r in range(world):
                    for (peer, t, a, b) in plan.recvs(r):
                        pair[(peer, r)] = pair.get((peer, r), 0) + (b - a) * geo.rects[t][2] * geo.cn
                return max(pair.values())
            others = [dp.make_exchange_plan(geo, world, owner_policy=p) for p in ("roundrobin", "locality")]
            assert max_pair(xp) <= min(max_pair(o) for o in others)      # the balanced policy has the lightest busiest link
        if world == 8:
            assert max(xp.bytes_received(r, geo) for r in range(8)) < 200e6


def test_kd_geometry_covers_canvas_with_overlap():
    """Config 5 tiling generator: deterministic, covers every canvas pixel, neighbours overlap, and the exchange plan
    of a strip partition over it delivers every row a strip reads."""
    import device_pipeline as dp
    g = dp.kd_geometry(2000, 1500, leaves=16, overlap=0.10, seed=3)
    assert g.rects == dp.kd_geometry(2000, 1500, leaves=16, overlap=0.10, seed=3).rects
    cover = np.zeros((g.canvas_h, g.canvas_w), np.int32)
    for (x, y, w, h) in g.rects:
        assert x >= 0 and y >= 0 and x + w <= g.canvas_w and y + h <= g.canvas_h and w >= 16 and h >= 16
        cover[y:y + h, x:x + w] += 1
    assert cover.min() >= 1 and cover.max() >= 2
    assert (cover >= 2).mean() > 0.2                      # >= 10 % margins on every side of every box
    plan = dp.make_exchange_plan(g, 4)
    for r in range(4):
        for t in range(len(g.rects)):
            a, b = plan.need[r][t]
            if a < b and plan.owners[t] != r:
                assert any(tt == t and aa <= a and bb >= b for (_, tt, aa, bb) in plan.recvs(r))


def test_bench_self_launch_reaches_the_ranks():
    """`python bench.py --gpus N` (the driver's scale command) starts its own ranks as child processes before touching
    torch / HIP: on a box without a GPU every child must get as far as the device check (rc 2, "no GPU visible"), the
    launcher itself must not refuse; a WORLD_SIZE that contradicts --gpus is an error, not silently accepted."""
    import subprocess
    import sys
    import _native
    if _native.device_count() > 0:
        pytest.skip("a GPU is present: the ranks would run the benchmark")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["SR_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, r.stderr
    assert "no GPU visible" in r.stderr and "needs torch.distributed.run" not in r.stderr and r.stdout.strip() == ""
    env.update({"WORLD_SIZE": "2", "RANK": "0"})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr


def test_bench_launcher_deadline_names_the_stuck_rank():
    """Ranks that never come back (here: they sleep right after start-up, as ranks stuck in an RCCL call would) must not
    leave the launcher waiting for the driver's kill: after --deadline-s it terminates them, prints every rank's last
    completed stage and returns 124.  A rank that FAILS stops the others at once with its status."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update({"SR_DIST_BACKEND": "gloo", "SR_BENCH_TEST_HANG_RANKS": "0,1"})
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--deadline-s", "4"]
    t0 = time.monotonic()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 124 and time.monotonic() - t0 < 120, (r.returncode, r.stderr[-800:])
    assert "deadline of 4 s passed with 2 of 2 ranks still running" in r.stderr
    for rank in (0, 1):
        assert f"rank {rank} [running] last completed stage: process started" in r.stderr
    assert r.stdout.strip() == ""
    import _native
    if _native.device_count() == 0:
        env["SR_BENCH_TEST_HANG_RANKS"] = "1"           # rank 0 fails its device check (rc 2) while rank 1 sleeps
        t0 = time.monotonic()
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 2 and time.monotonic() - t0 < 120, (r.returncode, r.stderr[-800:])
        assert "rank 0 exited with status 2; stopping the others" in r.stderr

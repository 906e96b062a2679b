"""The strip-sharded path behind the reference's entry point: ``SuperResolutionPipeline.process()`` under WORLD_SIZE > 1
and ``python main.py in out --gpus N`` (reference: main.py:269-441 stage order; its own fan-out is ParallelBlender,
blending_module.py:1665-1705).  CPU: the launcher, the rendezvous and the shard plan (host code) at world 2 and 3 over gloo.
GPU: a two-rank rehearsal (gloo, both ranks on the one card, rows staged through the host) whose output file and QA report
equal the one-rank run byte for byte.  NOT measured on more than one GPU: RCCL refuses two ranks on one device."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN = os.path.join(ROOT, "super-resolution-system_amd", "main.py")


def _source(path, h=300, w=420, seed=3):
    from PIL import Image
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.clip((128 + 64 * np.sin(xx / 37.0) + 48 * np.cos(yy / 23.0))[..., None] + rng.integers(-12, 13, (h, w, 3)), 0, 255)
    Image.fromarray(img.astype(np.uint8)).save(path)


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    e.update(env or {})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    return subprocess.run([sys.executable, MAIN] + args, capture_output=True, text=True, timeout=timeout, env=e)


@pytest.mark.parametrize("world", [2, 3])
def test_plan_only_ranks_agree(tmp_path, world):
    src = str(tmp_path / "in.png")
    _source(src)
    r = _run([src, str(tmp_path / "out.tiff"), "--block-size", "128", "--gpus", str(world), "--plan-only"])
    assert r.returncode == 0, r.stderr[-2000:]
    plan = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert plan["world"] == world and plan["canvas"] == [840, 600]
    b = plan["bounds"]
    assert b[0] == 0 and b[-1] == 600 and all(x <= y for x, y in zip(b, b[1:])) and all(x % 2 == 0 for x in b[:-1])
    assert len(plan["owners"]) == len(plan["rects"]) == 12 and set(plan["owners"]) <= set(range(world))
    # every rank blends its strip plus the SSIM halo rows, clipped to the canvas
    for rnk, (a, e) in enumerate(plan["rows"]):
        assert a <= b[rnk] and e >= b[rnk + 1] and 0 <= a <= e <= 600


def test_shard_plan_matches_one_rank_geometry():
    sys.path.insert(0, os.path.join(ROOT, "super-resolution-system_amd"))
    import main as m
    cfg = m.PipelineConfig(block_size=128)
    one, two = m.shard_plan((420, 300), cfg, 1), m.shard_plan((420, 300), cfg, 2)
    assert one["rects"] == two["rects"] and one["bounds"] == [0, 600] and one["owners"] == [0] * 12
    assert one["bytes_received"] == [0]
    # a two-rank plan moves rows only across the strip boundary: far less than the tiles themselves
    tile_bytes = sum(w * h * 3 for (_, _, w, h) in two["rects"])
    assert 0 < sum(two["bytes_received"]) < tile_bytes


def test_two_ranks_without_gpu_fail_loudly(tmp_path):
    """No CPU fallback behind the entry point either: without a GPU the ranks report the failure and the launcher's status is
    non-zero (on a GPU box this test is the rehearsal below)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: covered by the two-rank rehearsal")
    src = str(tmp_path / "in.png")
    _source(src, 160, 200)
    r = _run([src, str(tmp_path / "out.tiff"), "--block-size", "128", "--gpus", "2"], env={"SR_DIST_BACKEND": "gloo"})
    assert r.returncode != 0
    assert "no GPU visible" in (r.stdout + r.stderr) or "HIP" in (r.stdout + r.stderr)
    assert not os.path.exists(tmp_path / "out.tiff")


@pytest.mark.gpu
@pytest.mark.parametrize("ext", ["tiff", "png"])
def test_two_rank_rehearsal_equals_one_rank(tmp_path, ext):
    """launcher + 2 gloo ranks on the one GPU (3 processes) against the plain one-GPU run: same file bytes, same QA report."""
    src = str(tmp_path / "in.png")
    _source(src)
    out1, out2 = str(tmp_path / f"one.{ext}"), str(tmp_path / f"two.{ext}")
    r1 = _run([src, out1, "--block-size", "128"], timeout=600)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-2000:]
    r2 = _run([src, out2, "--block-size", "128", "--gpus", "2"], env={"SR_DIST_BACKEND": "gloo"}, timeout=900)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-2000:]
    assert open(out1, "rb").read() == open(out2, "rb").read()
    q1 = json.load(open(out1.rsplit(".", 1)[0] + "_qa_report.json"))
    q2 = json.load(open(out2.rsplit(".", 1)[0] + "_qa_report.json"))
    q1.pop("timestamp"), q2.pop("timestamp")
    assert q1 == q2 and np.isfinite(q1["full_reference"]["psnr"]) and 0.0 < q1["full_reference"]["ssim"] <= 1.0


@pytest.mark.gpu
def test_laplacian_fusion_across_ranks(tmp_path):
    """BlendingModule.laplacian_fusion called SPMD under an initialised process group (2 gloo ranks on the one GPU): every
    rank blends its strip, the strips are all-gathered -- the canvas equals the one-process result byte for byte."""
    worker = os.path.join(ROOT, "tests", "_fusion_worker.py")
    one, two = str(tmp_path / "one.npy"), str(tmp_path / "two.npy")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r1 = subprocess.run([sys.executable, worker, one], capture_output=True, text=True, timeout=600, env=env)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-2000:]
    r2 = subprocess.run([sys.executable, worker, two, "--launch", "2"], capture_output=True, text=True, timeout=900, env=env)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-2000:]
    a, b = np.load(one), np.load(two)
    assert a.shape == b.shape == (300, 420, 3) and np.array_equal(a, b)

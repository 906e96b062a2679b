#!/usr/bin/env python3
"""Runs ON THE GPU BOX.  A/B of the assessment on the second stream of the image stream:
  arm A  the fused march (SSE + Gaussian-11 cropped/full + integer uniform-7 in one kernel: 134 VGPRs, 3 blocks per CU),
  arm B  SR_ASSESS_SPLIT=1: the Gaussian(+SSE) variant (no uniform-7 ring: four blocks per CU) and the integer uniform-7
         variant as two launches on the same stream.
Alternated pairs on one box; per arm: image-stream step, the assessment's time inside the timed region and stand-alone.
    python tools/assess_split_ab.py [pairs] > profiles/r04_assess_split.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(split):
    env = dict(os.environ)
    env["SR_ASSESS_SPLIT"] = "1" if split else "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--sweep", "none",
                        "--no-cpu-baseline", "--no-pcie"], env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        raise SystemExit(r.stderr[-3000:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    k = d["kernels"]["assess_all"]
    roof = d["roofline"]
    in_region = roof["avg_launch_ms"] * k["launches_per_step"] if roof["kernel"] == "assess_all" else None
    return {"split": split, "ms_per_step": d["ms_per_step"], "step_ms_median": d["step_ms"]["median"],
            "assess_standalone_ms": k["ms_per_step"], "assess_launches_per_step": k["launches_per_step"],
            "assess_in_region_ms": None if in_region is None else round(in_region, 4),
            "quality": d["quality"]}


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    runs = []
    for _ in range(pairs):
        runs.append(run(False))
        runs.append(run(True))
    def mean(key, split):
        v = [r[key] for r in runs if r["split"] == split and r[key] is not None]
        return round(sum(v) / len(v), 4) if v else None
    out = {"what": "assessment A/B inside the image stream, alternated pairs on one box (tools/assess_split_ab.py)",
           "arms": {"A_fused": {k: mean(k, False) for k in ("ms_per_step", "step_ms_median", "assess_standalone_ms", "assess_in_region_ms")},
                    "B_split": {k: mean(k, True) for k in ("ms_per_step", "step_ms_median", "assess_standalone_ms", "assess_in_region_ms")}},
           "scores_equal": all(r["quality"]["psnr"] == runs[0]["quality"]["psnr"] and
                               abs(r["quality"]["ssim_uniform"] - runs[0]["quality"]["ssim_uniform"]) < 1e-12 for r in runs),
           "runs": runs}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

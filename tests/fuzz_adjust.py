#!/usr/bin/env python3
"""Randomised parity sweep (GPU box) of sr_color_correct_u8 -- the LUT map and both guided-filter branches of
BlendingModule.color_correction (blending_module.py:969-1146) -- against oracle/oracle_np.py, bit for bit: random image sizes
(around and above the 64 x 32 / 64 x 16 block shapes, so interiors, every border and ragged edges occur), 1 / 3 / 4 channels,
integer tables (the fused kernels k_cc_fused8 / k_gfx_coeff17), float tables (k_cc_fused8f where their box sums are exact in
fp64, else the pass-structured kernels), in-place calls.
Test infrastructure (it drives the oracle); tests/test_gpu_fuzz.py runs a short sweep.
usage: python tests/fuzz_adjust.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "super-resolution-system_amd")):
    sys.path.insert(0, p)
import numpy as np            # noqa: E402
import _native                # noqa: E402
from oracle import oracle_np as onp   # noqa: E402  (checker only)

F32 = np.float32


def run(cases: int = 60, seed: int = 1) -> int:
    rng = np.random.default_rng(seed)
    ctx = _native.default_context(0)
    bad = 0
    classes = [0, 0, 0]                                           # tables per sr_color_table_class
    for it in range(cases):
        mode = int(rng.choice([0, 1, 1, 2, 2]))
        cn = int(rng.choice([1, 3, 3, 4])) if mode != 2 else int(rng.choice([1, 3, 3]))
        h, w = int(rng.integers(2, 150)), int(rng.integers(2, 260))
        if rng.integers(0, 3) == 0:
            w = (w + 3) // 4 * 4                                  # rows of whole 16-byte groups: the vectorised loaders
        kind = int(rng.integers(0, 4))
        if kind == 0:
            img = rng.integers(0, 256, (h, w, cn), dtype=np.uint8)
        elif kind == 1:
            img = np.full((h, w, cn), int(rng.integers(0, 256)), np.uint8)
            img[rng.integers(0, h), rng.integers(0, w)] = rng.integers(0, 256)
        else:
            yy, xx = np.mgrid[0:h, 0:w]
            base = 128 + 90 * np.sin(xx / rng.uniform(3, 40)) * np.cos(yy / rng.uniform(3, 40))
            img = np.clip(base[..., None] + rng.integers(-20, 21, (h, w, cn)) + np.arange(cn) * 7, 0, 255).astype(np.uint8)
        tk = int(rng.integers(0, 4))
        if tk == 0:
            lut = np.tile(np.arange(256, dtype=F32), (cn, 1))
        elif tk == 1:
            lut = np.sort(rng.integers(0, 256, (cn, 256)), axis=1).astype(F32)
        elif tk == 2:
            lut = (np.arange(256, dtype=F32)[None, :] - rng.uniform(60, 180, (cn, 1)).astype(F32)) * \
                  rng.uniform(0.3, 1.9, (cn, 1)).astype(F32) + rng.uniform(60, 180, (cn, 1)).astype(F32)
        else:
            # a mean_std line that crosses zero inside the table (dark reference): negative entries, and now and then an
            # entry so close to zero that the sums are no longer exact (class 0: ordered kernels)
            lut = (np.arange(256, dtype=F32)[None, :] - rng.uniform(20, 120, (cn, 1)).astype(F32)) * \
                  rng.uniform(0.5, 3.0, (cn, 1)).astype(F32) + (rng.uniform(-1, 1, (cn, 1)) ** 5).astype(F32)
        classes[_native.color_table_class(lut)] += 1
        in_place = bool(rng.integers(0, 5) == 0)
        d_img = ctx.upload(img)
        d_out = d_img if in_place else ctx.alloc(img.size)
        ctx.color_correct_u8(d_img.ptr, w * cn, h, w, cn, lut, mode, 8, 0.01, d_out.ptr, w * cn)
        got = ctx.download(d_out.ptr, img.shape, np.uint8)
        d_img.free()
        if not in_place:
            d_out.free()
        corrected = np.stack([lut[c][img[..., c]] for c in range(cn)], axis=-1).astype(F32)
        src = img.astype(F32)
        if mode == 0:
            res = corrected
        elif mode == 1:
            res = onp.simple_guided_filter(corrected, src, 8, 0.01)
        elif cn == 1:
            res = onp.guided_filter_ximgproc(corrected[..., 0], src[..., 0], 8, 0.01)[..., None]
        else:
            res = onp.guided_filter_ximgproc(corrected, src, 8, 0.01)
        want = np.clip(res, 0, 255).astype(np.uint8)
        if not np.array_equal(got, want):
            bad += 1
            print(f"MISMATCH case {it}: mode {mode} cn {cn} {h}x{w} image kind {kind} table kind {tk} in_place {in_place}: "
                  f"{int((got != want).sum())} bytes differ")
    print(f"fuzz_adjust: {cases} colour-correction cases, {bad} mismatches; table classes (ordered, integer, exact-float) {classes}")
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    sys.exit(1 if run(n, sd) else 0)

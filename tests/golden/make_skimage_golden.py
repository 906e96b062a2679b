"""Generates tests/golden/metrics_skimage.npz with scikit-image 0.18.3 -- the library functions the
reference calls (quality_assessment_module.py:311,368-384), with the reference's keyword arguments.

Run in THIS container with the interpreter that has scikit-image:
    /opt/conda/bin/python3.9 tests/golden/make_skimage_golden.py
Inputs are stored next to the expected values, so nothing depends on RNG stream stability.
"""
import os
import warnings

import numpy as np

warnings.filterwarnings("ignore")
from skimage import __version__ as skv
from skimage.metrics import peak_signal_noise_ratio as psnr
from skimage.metrics import structural_similarity as ssim

out = {}
rs = np.random.RandomState(20260313)
cases = []
for name, (h, w) in {"a64": (64, 64), "odd": (193, 257), "s128": (128, 160)}.items():
    a = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    b = np.clip(a.astype(np.float32) + rs.randn(h, w, 3) * (3 + 4 * len(cases)), 0, 255).astype(np.uint8)
    # smooth structure so that SSIM is not noise-only
    yy, xx = np.mgrid[0:h, 0:w]
    base = (128 + 60 * np.sin(xx / 11.0) + 40 * np.cos(yy / 7.0))[..., None]
    a = np.clip(0.5 * a + 0.5 * base, 0, 255).astype(np.uint8)
    b = np.clip(0.5 * b + 0.5 * base + 2, 0, 255).astype(np.uint8)
    cases.append(name)
    out[f"{name}_a"], out[f"{name}_b"] = a, b
    out[f"{name}_psnr"] = psnr(a, b, data_range=255.0)
    ga, gb = a[..., 1].copy(), b[..., 1].copy()          # any single channel serves as the gray input
    out[f"{name}_ssim_uniform"] = ssim(ga, gb, data_range=255.0, multichannel=False)
    out[f"{name}_ssim_gauss"] = ssim(ga, gb, data_range=255.0, multichannel=False, gaussian_weights=True,
                                     sigma=1.5, use_sample_covariance=False)
# the reference's own example pair (quality_assessment_module.py:1394-1400), legacy seed 42
np.random.seed(42)
o = np.random.randint(0, 256, (512, 512, 3), dtype=np.uint8)
u = np.clip(o.astype(np.float32) + np.random.randn(512, 512, 3) * 5, 0, 255).astype(np.uint8)
out["ex_psnr"] = psnr(o, u, data_range=255.0)
out["ex_ssim_uniform_ch0"] = ssim(o[..., 0], u[..., 0], data_range=255.0, multichannel=False)
out["ex_ssim_gauss_ch0"] = ssim(o[..., 0], u[..., 0], data_range=255.0, multichannel=False, gaussian_weights=True,
                                sigma=1.5, use_sample_covariance=False)
out["cases"] = np.array(cases)
out["skimage_version"] = np.array(skv)
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "metrics_skimage.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, {k: float(v) for k, v in out.items() if k.endswith(("psnr", "uniform", "gauss", "ch0"))})

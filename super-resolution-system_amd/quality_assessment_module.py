"""quality_assessment_module -- MI355X-native mirror of the reference's full-reference metrics.

Call surface of quality_assessment_module.py:35-133,169-195,226-417,467-609: same class / method
names, arguments, return types and crop / preprocess rules.  PSNR (exact integer sum of squared
differences), RGB->gray, the three SSIM variants and the bicubic resize run as HIP kernels behind the
C ABI; only the final scalar formulas run on the host.  No CPU compute fallback.

SSIM branches: the reference calls skimage with ``multichannel=False``; a skimage that
accepts / ignores the keyword gives branch A (uniform 7x7 for ``multiscale=False``, Gaussian sigma 1.5
for ``multiscale=True``, cropped mean); one that rejects it gives branch B (``_calculate_ssim_simple``
for both).  ``ssim_branch`` selects it ('A' default).  ``gray_shift`` selects OpenCV's 15-bit (>= 4.x,
default) or 14-bit RGB2GRAY constants.

LPIPS (quality_assessment_module.py:135-146,197-224,419-465): the AlexNet / VGG16 forward of the ``lpips`` package
runs as hand-written fp32 MFMA convolutions (csrc/sr_lpips.hip), streamed in tiles so a 200 MP pair fits.  The
reference's constructor downloads pretrained weights by model name; this one never fetches anything: weights come
from ``lpips_weights={'vgg': <.npz path or dict>, 'alex': ...}`` (or SR_LPIPS_WEIGHTS_VGG / SR_LPIPS_WEIGHTS_ALEX),
a flat .npz of ``lpips.LPIPS(net).state_dict()`` arrays read with allow_pickle=False.  Without weights
``lpips_model_vgg`` stays None exactly like the reference when its import fails: ``evaluate_full_reference`` omits
the LPIPS keys and ``calculate_lpips`` raises RuntimeError.  NIQE / BRISQUE / commercial heuristics are outside
the tile -> blend -> assess path; ``evaluate_commercial`` returns an empty, labelled result so
main.process keeps its report structure.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from enum import Enum
from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np

import _native


class AssessmentLevel(Enum):
    EXCELLENT = "excellent"
    GOOD = "good"
    FAIR = "fair"
    POOR = "poor"
    BAD = "bad"


@dataclass
class QualityThresholds:
    """quality_assessment_module.py:44-75."""
    PSNR_EXCELLENT: float = 40.0
    PSNR_GOOD: float = 35.0
    PSNR_FAIR: float = 30.0
    SSIM_EXCELLENT: float = 0.98
    SSIM_GOOD: float = 0.95
    SSIM_FAIR: float = 0.90
    LPIPS_EXCELLENT: float = 0.02
    LPIPS_GOOD: float = 0.05
    LPIPS_FAIR: float = 0.10
    NIQE_EXCELLENT: float = 3.0
    NIQE_GOOD: float = 5.0
    NIQE_FAIR: float = 8.0
    BRISQUE_EXCELLENT: float = 20.0
    BRISQUE_GOOD: float = 35.0
    BRISQUE_FAIR: float = 50.0
    DELTA_E_EXCELLENT: float = 1.0
    DELTA_E_GOOD: float = 3.0
    DELTA_E_FAIR: float = 5.0


@dataclass
class ScaleConfig:
    """quality_assessment_module.py:78-86."""
    scale_factors: List[float] = field(default_factory=lambda: [0.1, 0.2, 0.4])
    scale_names: Dict[float, str] = field(default_factory=lambda: {
        0.1: "structure_color", 0.2: "mid_frequency", 0.4: "high_frequency"})


class _DevImage:
    """A u8 image resident in HBM (dense HWC or HW): uploaded from an array, freshly allocated, or a view of memory
    somebody else owns (``ptr=``: the device-resident pipeline hands its source and canvas over this way)."""

    def __init__(self, ctx, arr: Optional[np.ndarray] = None, shape=None, ptr: Optional[int] = None):
        self.ctx = ctx
        self._ptr = None
        if ptr is not None:
            self.shape = tuple(shape)
            self.buf = None
            self._ptr = int(ptr)
        elif arr is not None:
            arr = np.ascontiguousarray(arr, dtype=np.uint8)
            self.shape = arr.shape
            self.buf = ctx.upload(arr)
        else:
            self.shape = tuple(shape)
            self.buf = ctx.alloc(int(np.prod(self.shape)))

    @property
    def h(self): return self.shape[0]

    @property
    def w(self): return self.shape[1]

    @property
    def cn(self): return self.shape[2] if len(self.shape) == 3 else 1

    @property
    def stride(self): return self.w * self.cn

    @property
    def ptr(self): return self._ptr if self.buf is None else self.buf.ptr

    def free(self):
        if self.buf is not None:
            self.buf.free()


class QualityAssessmentModule:
    def __init__(self, device: str = 'cpu', thresholds: Optional[QualityThresholds] = None,
                 scale_config: Optional[ScaleConfig] = None, gpu_index: int = 0, ssim_branch: str = 'A',
                 gray_shift: int = 15, lpips_weights: Optional[Dict[str, Any]] = None, lpips_tile: int = 4096):
        # the reference's `device` only places the LPIPS networks; the metrics here always run on the GPU
        self.device = device
        self.thresholds = thresholds or QualityThresholds()
        self.scale_config = scale_config or ScaleConfig()
        self.gpu_index = gpu_index
        if ssim_branch not in ('A', 'B'):
            raise ValueError("ssim_branch must be 'A' or 'B'")
        self.ssim_branch = ssim_branch
        self.gray_shift = gray_shift
        self.lpips_tile = int(lpips_tile)
        self.lpips_model_vgg = None      # stay None without caller-supplied weights (nothing is fetched)
        self.lpips_model_alex = None
        self._init_lpips_models(lpips_weights)
        self._niqe_available = False
        self._brisque_available = False

    def _ctx(self) -> "_native.Context":
        return _native.default_context(self.gpu_index)

    def _init_lpips_models(self, lpips_weights: Optional[Dict[str, Any]] = None) -> None:
        """quality_assessment_module.py:135-146 without the download: build the GPU models from caller-supplied
        state-dict arrays ({'vgg': path | dict, 'alex': path | dict}; env SR_LPIPS_WEIGHTS_<NET> as a default)."""
        src = dict(lpips_weights or {})
        for net in ("vgg", "alex"):
            if net not in src and os.environ.get(f"SR_LPIPS_WEIGHTS_{net.upper()}"):
                src[net] = os.environ[f"SR_LPIPS_WEIGHTS_{net.upper()}"]
        for net, w in src.items():
            if net not in ("vgg", "alex"):
                raise ValueError(f"lpips_weights: unknown net {net!r}")
            if isinstance(w, (str, os.PathLike)):
                w = _native.load_lpips_weights(os.fspath(w))
            setattr(self, f"lpips_model_{net}", _native.LpipsModel(self._ctx(), net, w))

    # -- preprocessing (quality_assessment_module.py:169-195, 304-308) --------------------------------
    def _preprocess_image(self, image: Any, to_tensor: bool = False) -> np.ndarray:
        if hasattr(image, "detach") and hasattr(image, "dim"):        # torch tensor, CHW or 1xCHW
            if image.dim() == 4:
                image = image.squeeze(0)
            if image.dim() == 3:
                image = image.permute(1, 2, 0)
            image = image.detach().cpu().numpy()
        elif not isinstance(image, np.ndarray):                        # PIL image (main.py hands these in)
            image = np.asarray(image)
        if image.max() <= 1.0:
            image = (image * 255).astype(np.uint8)
        return image

    @staticmethod
    def _crop_pair(a: np.ndarray, b: np.ndarray):
        if a.shape != b.shape:
            mh, mw = min(a.shape[0], b.shape[0]), min(a.shape[1], b.shape[1])
            a, b = a[:mh, :mw], b[:mh, :mw]
        return a, b

    @staticmethod
    def _require_u8(img: np.ndarray, what: str) -> np.ndarray:
        if img.dtype != np.uint8:
            raise NotImplementedError(f"{what}: only uint8 images (after the reference's preprocess rule) are on "
                                      f"the HIP path; got {img.dtype}")
        return np.ascontiguousarray(img)

    # -- resize -------------------------------------------------------------------------------------------
    def _resize_dev(self, src: _DevImage, new_h: int, new_w: int) -> _DevImage:
        out_shape = (new_h, new_w, src.cn) if len(src.shape) == 3 else (new_h, new_w)
        dst = _DevImage(src.ctx, shape=out_shape)
        src.ctx.resize_cubic_u8(src.ptr, src.stride, src.h, src.w, src.cn, dst.ptr, dst.stride, new_h, new_w)
        return dst

    def downsample_bicubic(self, image: np.ndarray, scale_factor: float) -> np.ndarray:
        if scale_factor >= 1.0 or scale_factor <= 0:
            raise ValueError(f"scale_factor必须在(0, 1)范围内，当前值: {scale_factor}")
        h, w = image.shape[:2]
        return self.upsample_bicubic(image, (int(h * scale_factor), int(w * scale_factor)))

    def upsample_bicubic(self, image: np.ndarray, target_size: Tuple[int, int]) -> np.ndarray:
        img = self._require_u8(np.asarray(image), "bicubic resize")
        ctx = self._ctx()
        src = _DevImage(ctx, img)
        dst = self._resize_dev(src, int(target_size[0]), int(target_size[1]))
        out = ctx.download(dst.ptr, dst.shape, np.uint8)
        src.free(); dst.free()
        return out

    # -- device-level metrics ---------------------------------------------------------------------------
    def _psnr_dev(self, a: _DevImage, b: _DevImage, data_range: float) -> float:
        h, w = min(a.h, b.h), min(a.w, b.w)
        rowlen = w * a.cn
        sse = a.ctx.sse_u8(a.ptr, a.stride, b.ptr, b.stride, h, rowlen)
        return _native.psnr_from_sse(sse, h * rowlen, data_range)

    def _ssim_dev(self, a: _DevImage, b: _DevImage, multiscale: bool, data_range: float) -> float:
        h, w = min(a.h, b.h), min(a.w, b.w)
        mode = "simple" if self.ssim_branch == 'B' else ("gauss" if multiscale else "uniform")
        s, n = a.ctx.ssim_u8(a.ptr, a.stride, b.ptr, b.stride, h, w, a.cn, mode, self.gray_shift, data_range)
        return s / n

    # -- public metrics (quality_assessment_module.py:277-417) -----------------------------------------------
    def _pair(self, img1, img2, what: str, preprocessed: bool = False):
        # preprocessed: the caller has run _preprocess_image already (a second pass would scan both images for their
        # maximum again -- hundreds of milliseconds of host time at 200 MP)
        a = self._require_u8(img1 if preprocessed else self._preprocess_image(img1), what)
        b = self._require_u8(img2 if preprocessed else self._preprocess_image(img2), what)
        if a.ndim != b.ndim or (a.ndim == 3 and a.shape[2] != b.shape[2]):
            raise ValueError(f"{what}: images have different channel layouts {a.shape} vs {b.shape}")
        return a, b

    def calculate_psnr(self, img1: np.ndarray, img2: np.ndarray, data_range: float = 255.0) -> float:
        p1, p2 = self._preprocess_image(img1), self._preprocess_image(img2)
        if p1.dtype != np.uint8 or p2.dtype != np.uint8:
            # float images with max > 1 stay float in the reference: skimage promotes to >= fp32 and averages in fp64
            a32, b32 = self._crop_pair(np.asarray(p1), np.asarray(p2))
            a32 = np.ascontiguousarray(a32, dtype=np.float32)
            b32 = np.ascontiguousarray(b32, dtype=np.float32)
            ctx = self._ctx()
            da, db = ctx.upload(a32), ctx.upload(b32)
            try:
                rowlen = a32.size // a32.shape[0]
                sse = ctx.sse_f32(da.ptr, rowlen * 4, db.ptr, rowlen * 4, a32.shape[0], rowlen)
            finally:
                da.free(); db.free()
            mse = sse / a32.size
            return float("inf") if mse == 0 else float(10 * np.log10((data_range ** 2) / mse))
        a, b = self._pair(p1, p2, "calculate_psnr", preprocessed=True)
        ctx = self._ctx()
        da, db = _DevImage(ctx, a), _DevImage(ctx, b)
        try:
            return float(self._psnr_dev(da, db, data_range))
        finally:
            da.free(); db.free()

    def _ssim_float(self, p1: np.ndarray, p2: np.ndarray, multiscale: bool, data_range: float) -> float:
        """Images that stay non-u8 after _preprocess_image (float arrays with max > 1, wider integers): the reference hands
        them on as they are -- a float32 RGB pair through cv2's float RGB2GRAY, a 2-D pair straight to skimage, which
        computes in float64 (quality_assessment_module.py:351-417).  Runs on sr_ssim_float; float64 RGB raises like
        cv2.cvtColor does."""
        a, b = self._crop_pair(np.asarray(p1), np.asarray(p2))
        if a.ndim != b.ndim or (a.ndim == 3 and a.shape[2] != b.shape[2]):
            raise ValueError(f"calculate_ssim: images have different channel layouts {a.shape} vs {b.shape}")
        if a.ndim == 3:
            if a.shape[2] != 3:
                raise ValueError("calculate_ssim: colour images must have 3 channels (cv2.COLOR_RGB2GRAY)")
            if a.dtype not in (np.uint8, np.float32) or b.dtype not in (np.uint8, np.float32):
                raise ValueError("calculate_ssim: cv2.cvtColor(RGB2GRAY) supports 8-bit, 16-bit and float32 images only "
                                 f"(got {a.dtype} / {b.dtype}); 16-bit RGB is not on the HIP path")
            if a.dtype != b.dtype:
                # cv2.cvtColor grays each image in its OWN dtype (quality_assessment_module.py:359-360): the uint8 partner gets
                # the rounded fixed-point gray, the float32 one the float formula -- two gray planes up to 0.5 level apart.
                # sr_ssim_float grays both with the float formula, so the mixed pair is refused rather than answered differently.
                raise NotImplementedError("calculate_ssim: one RGB image is uint8 and the other float32 after preprocessing; "
                                          "the HIP path takes RGB pairs of one dtype (convert one of them)")
            dt, code = np.float32, _native.SR_F32
        else:
            dt, code = (np.float32, _native.SR_F32) if (a.dtype == np.float32 and b.dtype == np.float32) else (np.float64, _native.SR_F64)
        a = np.ascontiguousarray(a, dtype=dt)
        b = np.ascontiguousarray(b, dtype=dt)
        h, w = a.shape[:2]
        cn = 3 if a.ndim == 3 else 1
        mode = "simple" if self.ssim_branch == 'B' else ("gauss" if multiscale else "uniform")
        ctx = self._ctx()
        da, db = ctx.upload(a), ctx.upload(b)
        try:
            es = a.dtype.itemsize
            s, n = ctx.ssim_float(da.ptr, w * cn * es, db.ptr, w * cn * es, h, w, cn, code, mode, data_range)
            return float(s / n)
        finally:
            da.free(); db.free()

    def calculate_ssim(self, img1: np.ndarray, img2: np.ndarray, multiscale: bool = True,
                       data_range: float = 255.0) -> float:
        p1, p2 = self._preprocess_image(img1), self._preprocess_image(img2)
        if np.asarray(p1).dtype != np.uint8 or np.asarray(p2).dtype != np.uint8:
            return self._ssim_float(p1, p2, multiscale, data_range)
        a, b = self._pair(p1, p2, "calculate_ssim", preprocessed=True)
        if a.ndim == 3 and a.shape[2] != 3:
            raise ValueError("calculate_ssim: colour images must have 3 channels (cv2.COLOR_RGB2GRAY)")
        ctx = self._ctx()
        da, db = _DevImage(ctx, a), _DevImage(ctx, b)
        try:
            return float(self._ssim_dev(da, db, multiscale, data_range))
        finally:
            da.free(); db.free()

    def _calculate_ssim_simple(self, img1: np.ndarray, img2: np.ndarray) -> float:
        """quality_assessment_module.py:391-417 on already-gray u8 images."""
        a = self._require_u8(np.asarray(img1), "_calculate_ssim_simple")
        b = self._require_u8(np.asarray(img2), "_calculate_ssim_simple")
        ctx = self._ctx()
        da, db = _DevImage(ctx, a), _DevImage(ctx, b)
        try:
            s, n = ctx.ssim_u8(da.ptr, da.stride, db.ptr, db.stride, min(da.h, db.h), min(da.w, db.w), da.cn,
                               "simple", self.gray_shift)
            return float(s / n)
        finally:
            da.free(); db.free()

    def _to_lpips_tensor(self, image: np.ndarray) -> np.ndarray:
        """quality_assessment_module.py:197-224 as an ndarray (1, 3, H, W) in [-1, 1]; for inspection -- the GPU path
        applies the same arithmetic inside the stem convolution and never materialises this tensor."""
        img = np.asarray(image).astype(np.float32) / 255.0
        if img.ndim == 2:
            img = np.stack([img, img, img], axis=-1)
        elif img.shape[2] == 1:
            img = np.repeat(img, 3, axis=-1)
        elif img.shape[2] == 4:
            img = img[:, :, :3]
        return np.ascontiguousarray(img.transpose(2, 0, 1))[None] * np.float32(2.0) - np.float32(1.0)

    def _lpips_dev(self, a: _DevImage, b: _DevImage, net: str) -> float:
        model = self.lpips_model_vgg if net == 'vgg' else self.lpips_model_alex
        if model is None:
            raise RuntimeError("LPIPS模型未成功加载")
        h, w = min(a.h, b.h), min(a.w, b.w)                    # common top-left rectangle (:449-453)
        return model.value(a.ptr, a.stride, b.ptr, b.stride, h, w, a.cn, tile=self.lpips_tile)

    def calculate_lpips(self, img1: np.ndarray, img2: np.ndarray, net: str = 'vgg') -> float:
        if self.lpips_model_vgg is None:
            raise RuntimeError("LPIPS模型未成功加载")   # same error the reference raises without its models
        a = self._require_u8(self._preprocess_image(img1), "calculate_lpips")
        b = self._require_u8(self._preprocess_image(img2), "calculate_lpips")
        cn_a = a.shape[2] if a.ndim == 3 else 1
        cn_b = b.shape[2] if b.ndim == 3 else 1
        if cn_a != cn_b or cn_a not in (1, 3, 4):
            raise ValueError(f"calculate_lpips: channel layouts {a.shape} vs {b.shape}")
        ctx = self._ctx()
        da, db = _DevImage(ctx, a), _DevImage(ctx, b)
        try:
            return float(self._lpips_dev(da, db, net))
        finally:
            da.free(); db.free()

    # -- full-reference evaluation (quality_assessment_module.py:467-609) ------------------------------------
    def evaluate_full_reference(self, original: np.ndarray, upscaled: np.ndarray, scale_factor: int = 4) -> Dict[str, float]:
        o, u = self._pair(original, upscaled, "evaluate_full_reference")
        ctx = self._ctx()
        d_o, d_u = _DevImage(ctx, o), _DevImage(ctx, u)      # uploaded once, every metric reads HBM
        return self._evaluate_full_reference_dev(d_o, d_u)

    def evaluate_full_reference_device(self, d_original: int, original_shape, d_upscaled: int, upscaled_shape,
                                       scale_factor: int = 4) -> Dict[str, float]:
        """evaluate_full_reference on two dense u8 images that already live in HBM (device addresses + shapes): what
        the device-resident pipeline calls -- nothing is uploaded, only the scalar sums come back."""
        ctx = self._ctx()
        if len(original_shape) != len(upscaled_shape) or (len(original_shape) == 3 and original_shape[2] != upscaled_shape[2]):
            raise ValueError(f"evaluate_full_reference: images have different channel layouts {original_shape} vs {upscaled_shape}")
        return self._evaluate_full_reference_dev(_DevImage(ctx, shape=original_shape, ptr=d_original),
                                                 _DevImage(ctx, shape=upscaled_shape, ptr=d_upscaled))

    def _evaluate_full_reference_dev(self, d_o: _DevImage, d_u: _DevImage) -> Dict[str, float]:
        ctx = d_o.ctx
        try:
            metrics: Dict[str, Any] = {}
            metrics.update(self._downsample_comparison_dev(d_o, d_u))
            if d_o.shape == d_u.shape:
                # PSNR, SSIM and MS-SSIM of the full-size pair from ONE pass over both images (sr_assess_u8)
                h, w, cn = d_o.h, d_o.w, d_o.cn
                if self.ssim_branch == 'B':
                    flags, m1, m2 = _native.ASSESS_SSE | _native.ASSESS_SIMPLE, "simple", "simple"
                else:
                    flags, m1, m2 = _native.ASSESS_SSE | _native.ASSESS_UNIFORM7 | _native.ASSESS_GAUSS11, "uniform", "gauss"
                r = ctx.assess_u8(d_o.ptr, d_o.stride, d_u.ptr, d_u.stride, h, w, cn, flags=flags,
                                  gray_shift=self.gray_shift)
                metrics['psnr'] = float(_native.psnr_from_sse(int(round(r["sse"])), h * w * cn, 255.0))
                metrics['ssim'] = float(r[f"ssim_{m1}"] / _native.ssim_count(h, w, m1))
                metrics['ms_ssim'] = float(r[f"ssim_{m2}"] / _native.ssim_count(h, w, m2))
            else:
                metrics['psnr'] = float(self._psnr_dev(d_o, d_u, 255.0))
                metrics['ssim'] = float(self._ssim_dev(d_o, d_u, False, 255.0))
                metrics['ms_ssim'] = float(self._ssim_dev(d_o, d_u, True, 255.0))
            metrics['psnr_level'] = self._assess_psnr(metrics['psnr'])
            metrics['ssim_level'] = self._assess_ssim(metrics['ms_ssim'])
            if self.lpips_model_vgg is not None:               # only when weights were given (:508-511)
                metrics['lpips_vgg'] = float(self._lpips_dev(d_o, d_u, 'vgg'))
                if self.lpips_model_alex is not None:
                    metrics['lpips_alex'] = float(self._lpips_dev(d_o, d_u, 'alex'))
                metrics['lpips_level'] = self._assess_lpips(metrics['lpips_vgg'])
            metrics['overall_score'] = self._calculate_overall_score(metrics)
            return metrics
        finally:
            d_o.free(); d_u.free()

    def _downsample_comparison_dev(self, d_o: _DevImage, d_u: _DevImage) -> Dict[str, float]:
        out = {}
        for scale in self.scale_config.scale_factors:
            if scale >= 1.0 or scale <= 0:
                raise ValueError(f"scale_factor必须在(0, 1)范围内，当前值: {scale}")
            name = self.scale_config.scale_names.get(scale, f"scale_{scale}")
            if d_o.shape == d_u.shape and int(d_o.h * scale) >= 1 and int(d_o.w * scale) >= 1:
                # both bicubic resizes, PSNR and SSIM of the resized pair in one kernel: the resized images are
                # sampled on the fly and never written (sr_assess_resized_u8)
                dh, dw, cn = int(d_o.h * scale), int(d_o.w * scale), d_o.cn
                mode = "simple" if self.ssim_branch == 'B' else "uniform"
                bit = _native.ASSESS_SIMPLE if mode == "simple" else _native.ASSESS_UNIFORM7
                r = d_o.ctx.assess_resized_u8(d_o.ptr, d_o.stride, d_u.ptr, d_u.stride, d_o.h, d_o.w, cn, dh, dw,
                                              flags=_native.ASSESS_SSE | bit, gray_shift=self.gray_shift)
                out[f'psnr_{name}'] = float(_native.psnr_from_sse(int(round(r["sse"])), dh * dw * cn, 255.0))
                out[f'ssim_{name}'] = float(r[f"ssim_{mode}"] / _native.ssim_count(dh, dw, mode))
                continue
            sr = self._resize_dev(d_u, int(d_u.h * scale), int(d_u.w * scale))
            hr = self._resize_dev(d_o, int(d_o.h * scale), int(d_o.w * scale))
            out[f'psnr_{name}'] = float(self._psnr_dev(hr, sr, 255.0))
            out[f'ssim_{name}'] = float(self._ssim_dev(hr, sr, False, 255.0))
            sr.free(); hr.free()
        return out

    def _evaluate_downsample_comparison(self, original: np.ndarray, upscaled: np.ndarray, scale_factor: int) -> Dict[str, float]:
        o, u = self._pair(original, upscaled, "_evaluate_downsample_comparison")
        ctx = self._ctx()
        d_o, d_u = _DevImage(ctx, o), _DevImage(ctx, u)
        try:
            return self._downsample_comparison_dev(d_o, d_u)
        finally:
            d_o.free(); d_u.free()

    def _assess_psnr(self, v: float) -> str:
        t = self.thresholds
        if v >= t.PSNR_EXCELLENT:
            return AssessmentLevel.EXCELLENT.value
        if v >= t.PSNR_GOOD:
            return AssessmentLevel.GOOD.value
        if v >= t.PSNR_FAIR:
            return AssessmentLevel.FAIR.value
        return AssessmentLevel.POOR.value

    def _assess_ssim(self, v: float) -> str:
        t = self.thresholds
        if v >= t.SSIM_EXCELLENT:
            return AssessmentLevel.EXCELLENT.value
        if v >= t.SSIM_GOOD:
            return AssessmentLevel.GOOD.value
        if v >= t.SSIM_FAIR:
            return AssessmentLevel.FAIR.value
        return AssessmentLevel.POOR.value

    def _assess_lpips(self, v: float) -> str:
        t = self.thresholds
        if v <= t.LPIPS_EXCELLENT:
            return AssessmentLevel.EXCELLENT.value
        if v <= t.LPIPS_GOOD:
            return AssessmentLevel.GOOD.value
        if v <= t.LPIPS_FAIR:
            return AssessmentLevel.FAIR.value
        return AssessmentLevel.POOR.value

    def _calculate_overall_score(self, metrics: Dict[str, float]) -> float:
        scores = []
        if 'psnr' in metrics:
            scores.append(min(100, max(0, metrics['psnr'])))
        if 'ms_ssim' in metrics:
            scores.append(metrics['ms_ssim'] * 100)
        if 'lpips_vgg' in metrics:
            scores.append(max(0, (1 - metrics['lpips_vgg']) * 100))
        return float(np.mean(scores)) if scores else 0.0

    # -- outside the path -------------------------------------------------------------------------------------
    def evaluate_commercial(self, image: Any, roi_regions: Optional[List[Dict]] = None) -> Dict[str, Any]:
        return {"available": False,
                "note": "commercial / no-reference heuristics are outside the MI355X tile->blend->assess path",
                "roi_count": len(roi_regions or [])}

    def evaluate_no_reference(self, image: Any) -> Dict[str, Any]:
        raise NotImplementedError("NIQE / BRISQUE stand-ins are outside the MI355X tile->blend->assess path")

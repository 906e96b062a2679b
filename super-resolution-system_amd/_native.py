"""ctypes binding of include/sr_hip.h (libsrhip.so) -- the only way the Python host code
reaches the GPU.  There is deliberately no CPU fallback: if the library or a device is missing,
the compute entry points raise (SrNativeError), they never route through oracle/.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SR_LIB_PATH") or os.path.join(_HERE, "libsrhip.so")   # SR_LIB_PATH: an experiment build

SR_OK = 0
SR_ERR_INVALID_ARG, SR_ERR_SHAPE, SR_ERR_OOM, SR_ERR_HIP, SR_ERR_COMM, SR_ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6
PAD_MODES = {"mirror": 0, "replicate": 1, "reflect": 2, "constant": 3}
WEIGHT_TYPES = {"linear": 0, "cosine": 1, "sigmoid": 2, "ones": 3}
SSIM_MODES = {"uniform": 0, "gauss": 1, "simple": 2}
SR_U8, SR_F32, SR_F64 = 0, 1, 2


class SrNativeError(RuntimeError):
    """HIP / library failure (SR_ERR_HIP, SR_ERR_OOM, SR_ERR_COMM, SR_ERR_UNSUPPORTED)."""


class SrShapeError(ValueError):
    """SR_ERR_SHAPE: the reference raises cv2.error / ValueError for these."""


class Xfer(C.Structure):
    _fields_ = [("peer", C.c_int), ("d_ptr", C.c_void_p), ("bytes", C.c_uint64)]


class TileRect(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("w", C.c_int), ("h", C.c_int)]


class MergeTile(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("x", "y", "src_w", "src_h", "out_w", "out_h",
                                        "ov_t", "ov_b", "ov_l", "ov_r")]


class SeamRecord(C.Structure):
    _fields_ = [("tile", C.c_int), ("x", C.c_int), ("y", C.c_int), ("pad", C.c_int), ("score", C.c_double)]


class AssessSums(C.Structure):
    _fields_ = [("sse", C.c_double), ("ssim_uniform", C.c_double), ("ssim_gauss", C.c_double),
                ("ssim_simple", C.c_double)]


ASSESS_SSE, ASSESS_UNIFORM7, ASSESS_GAUSS11, ASSESS_SIMPLE, ASSESS_ALL = 1, 2, 4, 8, 15


class ProfRecord(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("ms", C.c_double), ("launches", C.c_int64)]


_lib = None
_lib_lock = threading.Lock()

# name -> (restype, argtypes); every symbol include/sr_hip.h declares
_vp, _i, _i64, _sz, _dbl = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_double
_pi = C.POINTER(C.c_int)
SIGNATURES = {
    "sr_version": (_i, []),
    "sr_source_digest": (C.c_char_p, []),
    "sr_last_error": (C.c_char_p, []),
    "sr_device_count": (_i, [_pi]),
    "sr_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "sr_ctx_create_on_stream": (_i, [_i, _vp, C.POINTER(_vp)]),
    "sr_ctx_destroy": (_i, [_vp]),
    "sr_ctx_sync": (_i, [_vp]),
    "sr_dev_alloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "sr_dev_free": (_i, [_vp, _vp]),
    "sr_memcpy_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "sr_memcpy_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "sr_memcpy_d2d": (_i, [_vp, _vp, _vp, _sz]),
    "sr_memset_d": (_i, [_vp, _vp, _i, _sz]),
    "sr_prof_enable": (_i, [_vp, _i]),
    "sr_prof_select": (_i, [_vp, C.c_char_p]),
    "sr_prof_reset": (_i, [_vp]),
    "sr_prof_get": (_i, [_vp, C.POINTER(ProfRecord), _i, _pi]),
    "sr_tile_plan": (_i, [_i, _i, _i, _i, _pi, _pi, _i]),
    "sr_tile_overlaps": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _pi]),
    "sr_tile_neighbors": (_i, [_pi, _i, _i, _i, _pi]),
    "sr_target_size": (_i, [_i, _i, _i, _pi, _pi]),
    "sr_weight_lut": (_i, [_i, _i, C.POINTER(C.c_float)]),
    "sr_tile_extract_pad": (_i, [_vp, _vp, _i, _i, _i, _i64, _pi, _i, _i, _i, _vp]),
    "sr_tile_extract": (_i, [_vp, _vp, _i, _i, _i, _i64, _pi, _i, C.POINTER(_vp), C.POINTER(_i64)]),
    "sr_pyr_down": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sr_pyr_up": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i]),
    "sr_pyr_up_sub": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "sr_pyr_up_add": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "sr_blend_plan_create": (_i, [_vp, C.POINTER(TileRect), _i, _i, _i, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "sr_blend_plan_destroy": (_i, [_vp]),
    "sr_strip_tile_rows": (_i, [C.POINTER(TileRect), _i, _i, _i, _i, _i, _pi]),
    "sr_pyramid_halo": (_i, [_i, _pi, _pi]),
    "sr_comm_unique_id": (_i, [_vp]),
    "sr_comm_init": (_i, [_vp, _vp, _i, _i, C.POINTER(_vp)]),
    "sr_comm_wrap": (_i, [_vp, C.POINTER(_vp)]),
    "sr_comm_info": (_i, [_vp, _pi, _pi]),
    "sr_comm_destroy": (_i, [_vp]),
    "sr_comm_exchange": (_i, [_vp, _vp, C.POINTER(Xfer), _i, C.POINTER(Xfer), _i]),
    "sr_comm_exchange_tile_rows": (_i, [_vp, _vp, C.POINTER(TileRect), _i, _i, _pi, _pi, C.POINTER(_vp), C.POINTER(_i64),
                                        C.POINTER(_vp)]),
    "sr_comm_allreduce_f64": (_i, [_vp, _vp, _vp, _i]),
    "sr_laplacian_blend_sharded": (_i, [_vp, _vp, _vp, C.POINTER(TileRect), _i, _i, _pi, _pi, C.POINTER(_vp), C.POINTER(_i64),
                                        C.POINTER(_vp), _vp, _i64]),
    "sr_sharded_tile_bases": (_i, [C.POINTER(TileRect), _i, _i, _i, _i, _pi, _pi, C.POINTER(_vp), C.POINTER(_i64), C.POINTER(_vp),
                                   C.POINTER(_vp)]),
    "sr_exchange_xfers": (_i, [C.POINTER(TileRect), _i, _i, _i, _i, _pi, _pi, C.POINTER(_vp), C.POINTER(_i64), C.POINTER(_vp),
                               C.POINTER(Xfer), _i, _pi, C.POINTER(Xfer), _i, _pi]),
    "sr_strip_bounds": (_i, [C.POINTER(TileRect), _i, _i, _i, _i, _i, _pi]),
    "sr_exchange_plan": (_i, [C.POINTER(TileRect), _i, _i, _i, _i, _i, _i, _i, _i, _pi, _pi, _pi, _pi]),
    "sr_blend_plan_tile_rows": (_i, [_vp, _i, _pi, _pi]),
    "sr_blend_plan_workspace_bytes": (_i, [_vp, C.POINTER(_sz)]),
    "sr_laplacian_blend": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_i64), _vp, _i64, _vp]),
    "sr_weighted_blend": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_i64), _vp, _i64, _vp]),
    "sr_blend_pyramids": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_i64), _pi, _i, _i]),
    "sr_blend_gather": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_i64), _vp, _i64, _vp]),
    "sr_laplacian_fusion_host": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(TileRect), _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "sr_weighted_fusion_host": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(TileRect), _i, _i, _i, _i, _i, _vp, _vp]),
    "sr_feather_merge": (_i, [_vp, C.POINTER(MergeTile), _i, C.POINTER(_vp), C.POINTER(_i64), _i, _vp, _i64, _i, _i]),
    "sr_feather_merge_dt": (_i, [_vp, _i, C.POINTER(MergeTile), _i, C.POINTER(_vp), C.POINTER(_i64), _i, _vp, _i64, _i, _i]),
    "sr_sse_u8": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i64, C.POINTER(C.c_uint64)]),
    "sr_sse_u8_async": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i64, _vp]),
    "sr_psnr_from_sse": (_dbl, [C.c_uint64, C.c_uint64, _dbl]),
    "sr_seam_scan": (_i, [_vp, _vp, _i64, _i, _i, _i, C.POINTER(TileRect), C.POINTER(_vp), C.POINTER(_i64), _i, _i, _i, _i,
                          _dbl, C.POINTER(SeamRecord), _i, _pi]),
    "sr_sse_f32": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i64, C.POINTER(_dbl)]),
    "sr_weighted_blend_custom": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_i64), C.POINTER(_vp), C.POINTER(_i64), _vp, _i64, _vp]),
    "sr_ssim_u8": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _dbl, _i, _i, C.POINTER(_dbl), C.POINTER(C.c_uint64)]),
    "sr_ssim_float": (_i, [_vp, _i, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _dbl, _i, _i, C.POINTER(_dbl), C.POINTER(C.c_uint64)]),
    "sr_ssim_u8_async": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _dbl, _i, _i, _vp, C.POINTER(C.c_uint64)]),
    "sr_assess_u8_async": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _dbl, _i, _i, _i, _vp]),
    "sr_assess_u8": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _dbl, _i, _i, _i, C.POINTER(AssessSums)]),
    "sr_assess_resized_u8_async": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _i, _dbl, _i, _vp]),
    "sr_assess_resized_u8": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _i, _dbl, _i, C.POINTER(AssessSums)]),
    "sr_ssim_count": (_i, [_i, _i, _i, _i, _i, C.POINTER(C.c_uint64)]),
    "sr_rgb2gray_u8": (_i, [_vp, _vp, _i64, _i, _i, _i, _vp, _i64]),
    "sr_resize_cubic_u8": (_i, [_vp, _vp, _i64, _i, _i, _i, _vp, _i64, _i, _i]),
    "sr_resize_cubic_window_u8": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i64]),
    "sr_gray_moments_u8": (_i, [_vp, _vp, _i, _i64, _i64, _i, _i, _i, _i, C.POINTER(C.c_uint64)]),
    "sr_histogram_u8": (_i, [_vp, _vp, _i64, _i, _i, _i, C.POINTER(C.c_uint64)]),
    "sr_color_table_class": (_i, [C.POINTER(C.c_float), _i, _i, C.POINTER(_i)]),
    "sr_color_correct_u8": (_i, [_vp, _vp, _i64, _i, _i, _i, C.POINTER(C.c_float), _i, _i, C.c_float, _vp, _i64]),
    "sr_encode_tiff_lzw": (_i, [_vp, _i, _i, _i, _i64, C.c_char_p, _i]),
    "sr_encode_png": (_i, [_vp, _i, _i, _i, _i64, _i, C.c_char_p, _i]),
    "sr_encode_jpeg": (_i, [_vp, _i, _i, _i, _i64, _i, C.c_char_p, _i]),
    "sr_lpips_create": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_vp), _i, C.POINTER(_vp), _i, _vp, _vp, C.POINTER(_vp)]),
    "sr_lpips_destroy": (_i, [_vp]),
    "sr_lpips_layer_sizes": (_i, [_i, _i, _i, _pi]),
    "sr_lpips_tile_count": (_i, [_i, _i, _i, _pi]),
    "sr_lpips_u8": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _i, C.POINTER(_dbl)]),
}


def _preload_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels ship their own libamdhip64.so (SONAME
    libamdhip64.so.7, like /opt/rocm's); if libsrhip.so pulled in the system copy first and torch is
    imported later, two HIP/HSA runtimes would fight over the device ("No HIP GPUs are available").
    Loading torch's copy by path first makes both resolve to the same object (glibc matches our NEEDED
    entry by SONAME and torch's by inode).  Without torch installed the system runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def _preload_rccl() -> None:
    """One RCCL per process, and the one PyTorch was built against: libtorch_hip.so needs "librccl.so" and would adopt
    whichever copy the process already holds, so before sr_comm.cpp binds RCCL (dlopen; a held copy wins) the wheel's own
    copy is loaded by path.  Without torch installed the ROCm copy is used.  RTLD_LOCAL on purpose: librccl.so drags in
    librocm_smi64.so, whose statics libamd_smi.so (loaded later by torch's device queries) re-defines -- with the symbols
    global both copies destroyed the same objects at exit (double free, seen in the full GPU suite)."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "librccl.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand)
        except OSError:
            pass


def load():
    """Load libsrhip.so, rebuilding it first when it is missing or older than csrc/ / include/ (the .so is git-ignored
    and travels with the tree, so an edited source must never meet yesterday's binary).  Raises SrNativeError."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        _preload_hip_runtime()
        import importlib.util
        spec = importlib.util.spec_from_file_location("_sr_build", os.path.join(_HERE, "_build.py"))
        bld = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bld)
        if not os.environ.get("SR_LIB_PATH"):
            try:
                if bld._stale():                       # missing, or not built from these sources / flags: never run a stale binary silently
                    bld.build_native()
            except Exception as exc:  # noqa: BLE001
                raise SrNativeError(f"libsrhip.so is missing or older than its sources and could not be rebuilt: {exc}") from exc
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as exc:
            raise SrNativeError(f"cannot load {LIB_PATH}: {exc}") from exc
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def last_error() -> str:
    return load().sr_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc == SR_OK:
        return
    msg = last_error()
    if rc == SR_ERR_INVALID_ARG:
        raise ValueError(msg)
    if rc == SR_ERR_SHAPE:
        raise SrShapeError(msg)
    if rc == SR_ERR_OOM:
        raise MemoryError(msg)
    raise SrNativeError(f"[sr {rc}] {msg}")


def device_count() -> int:
    n = C.c_int(0)
    rc = load().sr_device_count(C.byref(n))
    return n.value if rc == SR_OK else 0


# ------------------------------------------------------------------------------------------
# host-only bookkeeping wrappers
# ------------------------------------------------------------------------------------------
def tile_plan(image_w: int, image_h: int, block: int, overlap_px: int) -> List[Tuple[int, int, int, int]]:
    lib = load()
    n = C.c_int(0)
    check(lib.sr_tile_plan(image_w, image_h, block, overlap_px, C.byref(n), None, 0))
    buf = (C.c_int * (4 * n.value))()
    check(lib.sr_tile_plan(image_w, image_h, block, overlap_px, C.byref(n), buf, n.value))
    return [tuple(buf[4 * i: 4 * i + 4]) for i in range(n.value)]


def tile_overlaps(x, y, w, h, image_w, image_h, block, overlap_px) -> Tuple[int, int, int, int]:
    out = (C.c_int * 4)()
    check(load().sr_tile_overlaps(x, y, w, h, image_w, image_h, block, overlap_px, out))
    return tuple(out)


def tile_neighbors(xywh: Sequence[Tuple[int, int, int, int]], block: int, overlap_px: int):
    n = len(xywh)
    flat = (C.c_int * (4 * n))(*[v for t in xywh for v in t])
    out = (C.c_int * (4 * n))()
    check(load().sr_tile_neighbors(flat, n, block, overlap_px, out))
    return [tuple(out[4 * i: 4 * i + 4]) for i in range(n)]


def target_size(width: int, height: int, preset_mp: int) -> Tuple[int, int]:
    w, h = C.c_int(0), C.c_int(0)
    check(load().sr_target_size(width, height, preset_mp, C.byref(w), C.byref(h)))
    return w.value, h.value


def weight_lut(fw: int, weight_type: str) -> np.ndarray:
    out = np.empty(fw + 1, dtype=np.float32)
    check(load().sr_weight_lut(fw, WEIGHT_TYPES[weight_type], out.ctypes.data_as(C.POINTER(C.c_float))))
    return out


def strip_tile_rows(rects_xywh, levels: int, canvas_h: int, row_begin: int, row_end: int):
    """Host-only: tile-local input rows [(r0, r1)] every tile must supply for canvas rows [row_begin,row_end)."""
    n = len(rects_xywh)
    rects = (TileRect * n)(*[TileRect(int(x), int(y), int(w), int(h)) for (x, y, w, h) in rects_xywh])
    out = (C.c_int * (2 * n))()
    check(load().sr_strip_tile_rows(rects, n, int(levels), int(canvas_h), int(row_begin), int(row_end), out))
    return [(out[2 * i], out[2 * i + 1]) for i in range(n)]


_JPEG_EXT = (".jpg", ".jpeg", ".jpe", ".jfif")


def write_image(arr: np.ndarray, path: str, threads: int = 0, png_level: int = 3, jpeg_quality: int = 95) -> str:
    """Stage 5 of the pipeline (main.py:399-404) with the native multi-threaded writers: .tif / .tiff -> TIFF-LZW,
    .png -> PNG (compress_level 3), .jpg / .jpeg / .jpe / .jfif -> JPEG quality 95.  Any other extension goes to Pillow's
    ``save(path, quality=95)`` -- the reference's own else-branch (main.py:403), which picks the format from the extension
    (.bmp, .webp, ...) and raises for an unknown one -- never a JPEG stream under another name.  arr: HxW or HxWxC uint8.
    Returns the format."""
    a = np.ascontiguousarray(arr)
    if a.dtype != np.uint8 or a.ndim not in (2, 3):
        raise ValueError("write_image expects an HxW or HxWxC uint8 array")
    h, w = a.shape[:2]
    cn = a.shape[2] if a.ndim == 3 else 1
    lib, p, low = load(), os.fsencode(path), str(path).lower()
    ptr = a.ctypes.data_as(C.c_void_p)
    if low.endswith(".tif") or low.endswith(".tiff"):
        check(lib.sr_encode_tiff_lzw(ptr, h, w, cn, w * cn, p, threads))
        return "TIFF"
    if low.endswith(".png"):
        check(lib.sr_encode_png(ptr, h, w, cn, w * cn, png_level, p, threads))
        return "PNG"
    if low.endswith(_JPEG_EXT):
        if cn == 4:
            raise OSError("cannot write mode RGBA as JPEG")             # what Pillow raises (main.py:403 would propagate it)
        check(lib.sr_encode_jpeg(ptr, h, w, cn, w * cn, jpeg_quality, p, threads))
        return "JPEG"
    from PIL import Image
    img = Image.fromarray(a if cn != 1 else a.reshape(h, w))
    img.save(path, quality=jpeg_quality)                               # ValueError for an unknown extension, like the reference
    return (img.format or os.path.splitext(low)[1].lstrip(".")).upper()


OWNER_POLICIES = {"balanced": 0, "roundrobin": 1, "locality": 2}


def _rects(rects_xywh):
    n = len(rects_xywh)
    return n, (TileRect * max(n, 1))(*[TileRect(int(x), int(y), int(w), int(h)) for (x, y, w, h) in rects_xywh])


def strip_bounds(rects_xywh, levels: int, canvas_h: int, canvas_w: int, world: int) -> List[int]:
    """Host-only: work-balanced, even strip boundaries (sr_strip_bounds)."""
    n, rects = _rects(rects_xywh)
    out = (C.c_int * (world + 1))()
    check(load().sr_strip_bounds(rects, n, int(levels), int(canvas_h), int(canvas_w), int(world), out))
    return list(out)


def exchange_plan(rects_xywh, cn: int, levels: int, canvas_h: int, canvas_w: int, world: int, metric_halo: int,
                  owner_policy: str = "balanced"):
    """Host-only: (bounds, rows per rank, need[r][t] = (r0, r1), owners) of the strip exchange (sr_exchange_plan)."""
    n, rects = _rects(rects_xywh)
    bounds, rows = (C.c_int * (world + 1))(), (C.c_int * (2 * world))()
    need, owner = (C.c_int * (2 * world * n))(), (C.c_int * n)()
    check(load().sr_exchange_plan(rects, n, int(cn), int(levels), int(canvas_h), int(canvas_w), int(world), int(metric_halo),
                                  OWNER_POLICIES[owner_policy], bounds, rows, need, owner))
    return (list(bounds), [(rows[2 * r], rows[2 * r + 1]) for r in range(world)],
            [[(need[(r * n + t) * 2], need[(r * n + t) * 2 + 1]) for t in range(n)] for r in range(world)], list(owner))


def pyramid_halo(levels: int) -> Tuple[int, int]:
    """Host-only: worst-case (rows below, rows above) a strip reads of a tile beyond its own rows."""
    a, b = C.c_int(0), C.c_int(0)
    check(load().sr_pyramid_halo(int(levels), C.byref(a), C.byref(b)))
    return a.value, b.value


def color_table_class(glut: np.ndarray, terms: int = 64) -> int:
    """Host-only: how the first stage of the guided filter may sum a guide table (cn x 256 float32) -- 1 whole numbers
    (32-bit sliding sums), 2 box sums of `terms` values exact in fp64 (sliding fp64 sums), 0 ordered sums."""
    g = np.ascontiguousarray(glut, dtype=np.float32).reshape(-1, 256)
    cls = C.c_int(-1)
    check(load().sr_color_table_class(g.ctypes.data_as(C.POINTER(C.c_float)), int(g.shape[0]), int(terms), C.byref(cls)))
    return cls.value


def ssim_count(h: int, w: int, mode: str, row_begin: int = 0, row_end: Optional[int] = None) -> int:
    n = C.c_uint64(0)
    check(load().sr_ssim_count(h, w, SSIM_MODES[mode], row_begin, h if row_end is None else row_end, C.byref(n)))
    return n.value


def psnr_from_sse(sse: int, count: int, data_range: float = 255.0) -> float:
    return float(load().sr_psnr_from_sse(C.c_uint64(sse), C.c_uint64(count), data_range))


# ------------------------------------------------------------------------------------------
# device context
# ------------------------------------------------------------------------------------------
class DeviceBuffer:
    """Owned HBM allocation."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(ctx.lib.sr_dev_alloc(ctx.handle, max(self.nbytes, 1), C.byref(p)))
        self.ptr = p.value

    def free(self):
        if self.ptr and self.ctx.handle:
            self.ctx.lib.sr_dev_free(self.ctx.handle, C.c_void_p(self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


class Context:
    """One GPU + one stream (sr_ctx).  Thread-compatible: calls are serialised inside the library."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self.lib = load()
        h = C.c_void_p()
        if stream is None:
            check(self.lib.sr_ctx_create(int(device), C.byref(h)))
        else:
            check(self.lib.sr_ctx_create_on_stream(int(device), C.c_void_p(stream), C.byref(h)))
        self.handle = h
        self.device = int(device)
        self.h2d_bytes = 0               # bytes moved by upload() / download() (transfer accounting for the tests and
        self.d2h_bytes = 0               # the device-resident pipeline's "one upload, one download" claim)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.sr_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def sync(self):
        check(self.lib.sr_ctx_sync(self.handle))

    # memory ---------------------------------------------------------------------------
    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def upload(self, arr: np.ndarray) -> DeviceBuffer:
        arr = np.ascontiguousarray(arr)
        buf = DeviceBuffer(self, arr.nbytes)
        check(self.lib.sr_memcpy_h2d(self.handle, C.c_void_p(buf.ptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        self.h2d_bytes += arr.nbytes
        return buf

    def download(self, ptr: int, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        check(self.lib.sr_memcpy_d2h(self.handle, out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes))
        self.d2h_bytes += out.nbytes
        return out

    def memset(self, ptr: int, value: int, nbytes: int):
        check(self.lib.sr_memset_d(self.handle, C.c_void_p(ptr), value, nbytes))

    # profiling ------------------------------------------------------------------------
    def prof_enable(self, on: bool = True):
        check(self.lib.sr_prof_enable(self.handle, 1 if on else 0))

    def prof_select(self, name: Optional[str] = None):
        """Time only this kernel family (None: all)."""
        check(self.lib.sr_prof_select(self.handle, name.encode() if name else None))

    def prof_reset(self):
        check(self.lib.sr_prof_reset(self.handle))

    def prof_get(self):
        n = C.c_int(0)
        recs = (ProfRecord * 64)()
        check(self.lib.sr_prof_get(self.handle, recs, 64, C.byref(n)))
        return {recs[i].name.decode(): (recs[i].ms, recs[i].launches) for i in range(min(n.value, 64))}

    # device-pointer ops ------------------------------------------------------------------
    def tile_extract_pad(self, d_img: int, img_h, img_w, cn, img_stride, xywh, block, pad_mode: str, d_tiles: int):
        n = len(xywh)
        flat = (C.c_int * (4 * n))(*[int(v) for t in xywh for v in t])
        check(self.lib.sr_tile_extract_pad(self.handle, C.c_void_p(d_img), img_h, img_w, cn, img_stride, flat, n,
                                           block, PAD_MODES[pad_mode], C.c_void_p(d_tiles)))

    def tile_extract(self, d_img: int, img_h, img_w, cn, img_stride, xywh, d_tiles: Sequence[int],
                     strides: Sequence[int]):
        n = len(xywh)
        flat = (C.c_int * (4 * n))(*[int(v) for t in xywh for v in t])
        ptrs = (C.c_void_p * n)(*[C.c_void_p(p) for p in d_tiles])
        st = (C.c_int64 * n)(*[int(s) for s in strides])
        check(self.lib.sr_tile_extract(self.handle, C.c_void_p(d_img), img_h, img_w, cn, img_stride, flat, n, ptrs, st))

    def sse_u8(self, d_a: int, stride_a: int, d_b: int, stride_b: int, h: int, rowlen: int) -> int:
        out = C.c_uint64(0)
        check(self.lib.sr_sse_u8(self.handle, C.c_void_p(d_a), stride_a, C.c_void_p(d_b), stride_b, h, rowlen,
                                 C.byref(out)))
        return out.value

    def seam_scan(self, d_canvas: int, canvas_stride: int, canvas_h: int, canvas_w: int, cn: int, rects_xywh,
                  d_tiles: Sequence[int], strides: Sequence[int], window: int, stride: int, threshold: float,
                  gray_shift: int = 15):
        """-> [(tile, x, y, score)] of the windows below threshold, sorted by (tile, y, x)."""
        n = len(rects_xywh)
        rects = (TileRect * n)(*[TileRect(int(x), int(y), int(w), int(h)) for (x, y, w, h) in rects_xywh])
        ptrs = (C.c_void_p * n)(*[C.c_void_p(p) for p in d_tiles])
        st = (C.c_int64 * n)(*[int(s) for s in strides])
        cap = 1 << 18                  # records the first pass can return; a fuller scan is repeated once with room
        while True:
            recs = (SeamRecord * cap)()
            cnt = C.c_int(0)
            check(self.lib.sr_seam_scan(self.handle, C.c_void_p(d_canvas), canvas_stride, canvas_h, canvas_w, cn, rects,
                                        ptrs, st, n, window, stride, gray_shift, threshold, recs, cap, C.byref(cnt)))
            if cnt.value <= cap:
                break
            cap = cnt.value
        arr = np.frombuffer(recs, dtype=np.dtype([("tile", "<i4"), ("x", "<i4"), ("y", "<i4"), ("pad", "<i4"),
                                                  ("score", "<f8")]), count=cnt.value)
        arr = arr[np.lexsort((arr["x"], arr["y"], arr["tile"]))]
        return list(zip(arr["tile"].tolist(), arr["x"].tolist(), arr["y"].tolist(), arr["score"].tolist()))

    def sse_f32(self, d_a: int, stride_a: int, d_b: int, stride_b: int, h: int, rowlen: int) -> float:
        out = C.c_double(0.0)
        check(self.lib.sr_sse_f32(self.handle, C.c_void_p(d_a), stride_a, C.c_void_p(d_b), stride_b, h, rowlen, C.byref(out)))
        return out.value

    def sse_u8_async(self, d_a, stride_a, d_b, stride_b, h, rowlen, d_out: int):
        check(self.lib.sr_sse_u8_async(self.handle, C.c_void_p(d_a), stride_a, C.c_void_p(d_b), stride_b, h, rowlen,
                                       C.c_void_p(d_out)))

    def ssim_u8(self, d_a, stride_a, d_b, stride_b, h, w, cn, mode: str, gray_shift=15, data_range=255.0,
                row_begin=0, row_end=None) -> Tuple[float, int]:
        s, n = C.c_double(0.0), C.c_uint64(0)
        check(self.lib.sr_ssim_u8(self.handle, C.c_void_p(d_a), stride_a, C.c_void_p(d_b), stride_b, h, w, cn,
                                  SSIM_MODES[mode], gray_shift, data_range, row_begin,
                                  h if row_end is None else row_end, C.byref(s), C.byref(n)))
        return s.value, n.value

    def ssim_float(self, d_a, stride_a, d_b, stride_b, h, w, cn, dtype, mode: str, data_range=255.0, row_begin=0, row_end=None):
        """sr_ssim_float: (sum, count) of the SSIM map of two float32 (dtype SR_F32) or float64 (SR_F64) images."""
        s, n = C.c_double(0.0), C.c_uint64(0)
        check(self.lib.sr_ssim_float(self.handle, int(dtype), C.c_void_p(d_a), stride_a, C.c_void_p(d_b), stride_b, h, w, cn,
                                     SSIM_MODES[mode], float(data_range), row_begin, h if row_end is None else row_end,
                                     C.byref(s), C.byref(n)))
        return s.value, n.value

    def ssim_u8_async(self, d_a, stride_a, d_b, stride_b, h, w, cn, mode: str, d_sum: int, gray_shift=15,
                      data_range=255.0, row_begin=0, row_end=None) -> int:
        n = C.c_uint64(0)
        check(self.lib.sr_ssim_u8_async(self.handle, C.c_void_p(d_a), stride_a, C.c_void_p(d_b), stride_b, h, w, cn,
                                        SSIM_MODES[mode], gray_shift, data_range, row_begin,
                                        h if row_end is None else row_end, C.c_void_p(d_sum), C.byref(n)))
        return n.value

    def assess_u8_async(self, d_a, stride_a, d_b, stride_b, h, w, cn, d_out: int, flags=ASSESS_ALL, gray_shift=15,
                        data_range=255.0, row_begin=0, row_end=None):
        """Fused PSNR-SSE + SSIM partial sums -> 4 doubles at d_out (sse, uniform, gauss, simple); no sync."""
        check(self.lib.sr_assess_u8_async(self.handle, C.c_void_p(d_a), stride_a, C.c_void_p(d_b), stride_b, h, w, cn,
                                          gray_shift, data_range, row_begin, h if row_end is None else row_end,
                                          flags, C.c_void_p(d_out)))

    def assess_u8(self, d_a, stride_a, d_b, stride_b, h, w, cn, flags=ASSESS_ALL, gray_shift=15, data_range=255.0,
                  row_begin=0, row_end=None) -> dict:
        out = AssessSums()
        check(self.lib.sr_assess_u8(self.handle, C.c_void_p(d_a), stride_a, C.c_void_p(d_b), stride_b, h, w, cn,
                                    gray_shift, data_range, row_begin, h if row_end is None else row_end, flags,
                                    C.byref(out)))
        return {"sse": out.sse, "ssim_uniform": out.ssim_uniform, "ssim_gauss": out.ssim_gauss,
                "ssim_simple": out.ssim_simple}

    def assess_resized_u8(self, d_a, stride_a, d_b, stride_b, h, w, cn, dst_h, dst_w, flags=ASSESS_SSE | ASSESS_UNIFORM7,
                          gray_shift=15, data_range=255.0) -> dict:
        """The assess_u8 sums taken on the INTER_CUBIC resize of both images to dst_h x dst_w (sampled on the fly)."""
        out = AssessSums()
        check(self.lib.sr_assess_resized_u8(self.handle, C.c_void_p(d_a), stride_a, C.c_void_p(d_b), stride_b, h, w, cn,
                                            dst_h, dst_w, gray_shift, data_range, flags, C.byref(out)))
        return {"sse": out.sse, "ssim_uniform": out.ssim_uniform, "ssim_gauss": out.ssim_gauss,
                "ssim_simple": out.ssim_simple}

    def gray_std_u8(self, d_tiles, n, tile_bytes, stride, h, w, gray_shift=15, swap_rb=True) -> np.ndarray:
        """np.std of the u8 gray image of n RGB tiles in HBM (from exact integer moments, float64)."""
        sums = np.zeros(2 * max(n, 1), dtype=np.uint64)
        check(self.lib.sr_gray_moments_u8(self.handle, C.c_void_p(d_tiles), int(n), int(tile_bytes), int(stride), int(h),
                                          int(w), int(gray_shift), 1 if swap_rb else 0,
                                          sums.ctypes.data_as(C.POINTER(C.c_uint64))))
        cnt = float(h) * float(w)
        s1, s2 = sums[0:2 * n:2].astype(np.float64), sums[1:2 * n:2].astype(np.float64)
        m = s1 / cnt
        return np.sqrt(np.maximum(s2 / cnt - m * m, 0.0))

    def histogram_u8(self, d_img, stride, h, w, cn) -> np.ndarray:
        """-> (cn, 256) int64 counts, np.histogram(channel, 256, [0, 256]) of u8 data."""
        out = np.zeros((cn, 256), dtype=np.uint64)
        check(self.lib.sr_histogram_u8(self.handle, C.c_void_p(d_img), int(stride), int(h), int(w), int(cn),
                                       out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out.astype(np.int64)

    def color_correct_u8(self, d_img, stride, h, w, cn, glut: np.ndarray, local_filter: int, radius: int, eps: float,
                         d_out, out_stride):
        g = np.ascontiguousarray(glut, dtype=np.float32).reshape(cn, 256)
        check(self.lib.sr_color_correct_u8(self.handle, C.c_void_p(d_img), int(stride), int(h), int(w), int(cn),
                                           g.ctypes.data_as(C.POINTER(C.c_float)), int(local_filter), int(radius),
                                           C.c_float(eps), C.c_void_p(d_out), int(out_stride)))

    def rgb2gray_u8(self, d_rgb, stride, h, w, d_gray, gray_stride, gray_shift=15):
        check(self.lib.sr_rgb2gray_u8(self.handle, C.c_void_p(d_rgb), stride, h, w, gray_shift, C.c_void_p(d_gray),
                                      gray_stride))

    def resize_cubic_u8(self, d_src, src_stride, h, w, cn, d_dst, dst_stride, dh, dw):
        check(self.lib.sr_resize_cubic_u8(self.handle, C.c_void_p(d_src), src_stride, h, w, cn, C.c_void_p(d_dst),
                                          dst_stride, dh, dw))

    def resize_cubic_window_u8(self, d_src, src_stride, h, w, cn, dh, dw, x0, y0, ww, wh, d_dst, dst_stride):
        check(self.lib.sr_resize_cubic_window_u8(self.handle, C.c_void_p(d_src), src_stride, h, w, cn, dh, dw, x0, y0,
                                                 ww, wh, C.c_void_p(d_dst), dst_stride))

    # numpy-in / numpy-out conveniences (stage through HBM) ----------------------------------
    def pyr_down_np(self, img: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(img, dtype=np.float32)
        h, w = a.shape[:2]
        cn = a.shape[2] if a.ndim == 3 else 1
        out_shape = ((h + 1) // 2, (w + 1) // 2) + a.shape[2:]
        src = self.upload(a)
        dst = self.alloc(int(np.prod(out_shape)) * 4)
        check(self.lib.sr_pyr_down(self.handle, C.c_void_p(src.ptr), h, w, cn, C.c_void_p(dst.ptr)))
        res = self.download(dst.ptr, out_shape, np.float32)
        src.free(); dst.free()
        return res

    def pyr_up_np(self, img: np.ndarray, dst_hw, a: Optional[np.ndarray] = None, mode: str = "up") -> np.ndarray:
        """mode 'up': pyrUp(img); 'sub': a - pyrUp(img); 'add': pyrUp(img) + a."""
        s = np.ascontiguousarray(img, dtype=np.float32)
        hs, ws = s.shape[:2]
        cn = s.shape[2] if s.ndim == 3 else 1
        hd, wd = int(dst_hw[0]), int(dst_hw[1])
        out_shape = (hd, wd) + s.shape[2:]
        src = self.upload(s)
        dst = self.alloc(int(np.prod(out_shape)) * 4)
        if mode == "up":
            check(self.lib.sr_pyr_up(self.handle, C.c_void_p(src.ptr), hs, ws, cn, C.c_void_p(dst.ptr), hd, wd))
        else:
            if (hs, ws) != ((hd + 1) // 2, (wd + 1) // 2):
                raise SrShapeError(f"pyrUp: source {hs}x{ws} does not match destination {hd}x{wd}")
            aa = self.upload(np.ascontiguousarray(a, dtype=np.float32))
            fn = self.lib.sr_pyr_up_sub if mode == "sub" else self.lib.sr_pyr_up_add
            check(fn(self.handle, C.c_void_p(aa.ptr), hd, wd, cn, C.c_void_p(src.ptr), C.c_void_p(dst.ptr)))
            aa.free()
        res = self.download(dst.ptr, out_shape, np.float32)
        src.free(); dst.free()
        return res

    def fusion_np(self, tiles: Sequence[np.ndarray], positions_yx, output_shape, levels: int, weight_type: str,
                  laplacian: bool = True, return_float: bool = False):
        """tiles: list of HWC (or HW) arrays, all u8 or all float; positions (y, x)."""
        n = len(tiles)
        is_u8 = all(t.dtype == np.uint8 for t in tiles)
        arrs = [np.ascontiguousarray(t if is_u8 else t.astype(np.float32)) for t in tiles]
        cn = arrs[0].shape[2] if arrs[0].ndim == 3 else 1
        if any(a.ndim != arrs[0].ndim or a.shape[2:] != arrs[0].shape[2:] for a in arrs):
            raise ValueError("fusion: every tile must have the same number of channels")   # the C side copies h*w*cn per tile
        H, W = int(output_shape[0]), int(output_shape[1])
        rects = (TileRect * n)(*[TileRect(int(p[1]), int(p[0]), a.shape[1], a.shape[0])
                                 for a, p in zip(arrs, positions_yx)])
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        out = np.empty((H, W) if cn == 1 else (H, W, cn), dtype=np.uint8)
        outf = np.empty(out.shape, dtype=np.float32) if return_float else None
        fptr = outf.ctypes.data_as(C.c_void_p) if outf is not None else None
        dt = SR_U8 if is_u8 else SR_F32
        self.h2d_bytes += sum(a.nbytes for a in arrs)        # the *_host entry points stage tiles and canvas themselves
        self.d2h_bytes += out.nbytes + (outf.nbytes if outf is not None else 0)
        if laplacian:
            check(self.lib.sr_laplacian_fusion_host(self.handle, dt, ptrs, rects, n, cn, H, W, int(levels),
                                                    WEIGHT_TYPES[weight_type], out.ctypes.data_as(C.c_void_p), fptr))
        else:
            check(self.lib.sr_weighted_fusion_host(self.handle, dt, ptrs, rects, n, cn, H, W,
                                                   WEIGHT_TYPES[weight_type], out.ctypes.data_as(C.c_void_p), fptr))
        return (out, outf) if return_float else out


def _feather_merge_np(self, arrays, descs, output_width: int, output_height: int, blending: bool = True) -> np.ndarray:
    """TilingModule.merge_tiles on the GPU: arrays are HxWx3 tiles, ALL uint8 or ALL float32; descs dicts with the
    sr_merge_tile fields."""
    n = len(arrays)
    is_f32 = n > 0 and arrays[0].dtype == np.float32
    if any(a.dtype != (np.float32 if is_f32 else np.uint8) for a in arrays):
        raise ValueError("feather_merge_np: tiles must be all uint8 or all float32")
    es = 4 if is_f32 else 1
    bufs = [self.upload(a) for a in arrays]
    mt = (MergeTile * n)(*[MergeTile(*[int(d[k]) for k in ("x", "y", "src_w", "src_h", "out_w", "out_h",
                                                             "ov_t", "ov_b", "ov_l", "ov_r")]) for d in descs])
    ptrs = (C.c_void_p * n)(*[C.c_void_p(b.ptr) for b in bufs])
    st = (C.c_int64 * n)(*[a.shape[1] * 3 * es for a in arrays])
    canvas = self.alloc(output_width * output_height * 3)
    try:
        check(self.lib.sr_feather_merge_dt(self.handle, SR_F32 if is_f32 else SR_U8, mt, n, ptrs, st, 1 if blending else 0,
                                           C.c_void_p(canvas.ptr), output_width * 3, output_height, output_width))
        return self.download(canvas.ptr, (output_height, output_width, 3), np.uint8)
    finally:
        self.sync()
        for b in bufs:
            b.free()
        canvas.free()


Context.feather_merge_np = _feather_merge_np


COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """sr_comm_unique_id: the 128-byte rendezvous token rank 0 creates and ships to every rank."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _preload_rccl()
    check(load().sr_comm_unique_id(buf))
    return buf.raw


def _xfer_args(rects_xywh, world, need, owners, owned_ptrs, owned_strides, recv_ptrs):
    n, rects = _rects(rects_xywh)
    flat = (C.c_int * (2 * world * n))(*[v for r in range(world) for t in range(n) for v in need[r][t]])
    own = (C.c_int * n)(*[int(o) for o in owners])
    dp = (_vp * n)(*[int(p) if p else None for p in owned_ptrs])
    st = (_i64 * n)(*[int(v) for v in owned_strides])
    rp = (_vp * n)(*[int(p) if p else None for p in recv_ptrs])
    return n, rects, flat, own, dp, st, rp


def exchange_xfers(rects_xywh, cn: int, world: int, rank: int, need, owners, owned_ptrs, owned_strides, recv_ptrs):
    """Host-only (sr_exchange_xfers): ([(peer, pointer, bytes)] sends, receives) of one rank's grouped exchange."""
    n, rects, flat, own, dp, st, rp = _xfer_args(rects_xywh, world, need, owners, owned_ptrs, owned_strides, recv_ptrs)
    sends, recvs = (Xfer * (n * world))(), (Xfer * n)()
    ns, nr = C.c_int(0), C.c_int(0)
    check(load().sr_exchange_xfers(rects, n, int(cn), int(world), int(rank), flat, own, dp, st, rp, sends, n * world, C.byref(ns),
                                   recvs, n, C.byref(nr)))
    return ([(sends[i].peer, sends[i].d_ptr or 0, sends[i].bytes) for i in range(ns.value)],
            [(recvs[i].peer, recvs[i].d_ptr or 0, recvs[i].bytes) for i in range(nr.value)])


def sharded_tile_bases(rects_xywh, cn: int, world: int, rank: int, need, owners, owned_ptrs, strides, recv_ptrs) -> List[int]:
    """Host-only (sr_sharded_tile_bases): the per-tile base pointers rank `rank` hands to its strip blend (0 for tiles its
    strip does not read); validates owners, rows, dense strides of received tiles and receive buffers."""
    n, rects, flat, own, dp, st, rp = _xfer_args(rects_xywh, world, need, owners, owned_ptrs, strides, recv_ptrs)
    base = (_vp * n)()
    check(load().sr_sharded_tile_bases(rects, n, int(cn), int(world), int(rank), flat, own, dp, st, rp, base))
    return [int(b or 0) for b in base]


class Comm:
    """RCCL communicator of the C ABI (sr_comm_*): one process per GPU, transfers on the context's stream."""

    def __init__(self, ctx: "Context", unique_id: bytes, world: int, rank: int):
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("unique_id must be the 128 bytes of comm_unique_id()")
        self.ctx, self.world, self.rank = ctx, int(world), int(rank)
        h = _vp()
        _preload_rccl()
        check(load().sr_comm_init(ctx.handle, C.create_string_buffer(unique_id, COMM_ID_BYTES), self.world, self.rank, C.byref(h)))
        self.h = h

    def exchange(self, sends, recvs):
        """sends / recvs: [(peer, device pointer, bytes)] -- one grouped batch, asynchronous on the context's stream."""
        s = (Xfer * max(len(sends), 1))(*[Xfer(int(p), int(d), int(b)) for (p, d, b) in sends])
        r = (Xfer * max(len(recvs), 1))(*[Xfer(int(p), int(d), int(b)) for (p, d, b) in recvs])
        check(load().sr_comm_exchange(self.ctx.handle, self.h, s, len(sends), r, len(recvs)))

    def exchange_tile_rows(self, rects_xywh, cn: int, need, owners, owned_ptrs, owned_strides, recv_ptrs):
        """sr_comm_exchange_tile_rows: need[r][t] = (r0, r1) and owners from exchange_plan(); owned_ptrs[t] / recv_ptrs[t]
        are device pointers or 0."""
        n, rects, flat, own, dp, st, rp = _xfer_args(rects_xywh, self.world, need, owners, owned_ptrs, owned_strides, recv_ptrs)
        check(load().sr_comm_exchange_tile_rows(self.ctx.handle, self.h, rects, n, int(cn), flat, own, dp, st, rp))

    def blend_sharded(self, plan: "BlendPlan", rects_xywh, cn: int, need, owners, owned_ptrs, strides, recv_ptrs, d_canvas: int,
                      canvas_stride: int, ctx: Optional["Context"] = None):
        """sr_laplacian_blend_sharded: the row exchange and this rank's strip blend on one stream (plan made on the same
        context, for the same tiles)."""
        n, rects, flat, own, dp, st, rp = _xfer_args(rects_xywh, self.world, need, owners, owned_ptrs, strides, recv_ptrs)
        check(load().sr_laplacian_blend_sharded((ctx or self.ctx).handle, self.h, plan.handle, rects, n, int(cn), flat, own, dp, st, rp,
                                                _vp(int(d_canvas)), int(canvas_stride)))

    def allreduce_f64(self, d_ptr: int, count: int):
        check(load().sr_comm_allreduce_f64(self.ctx.handle, self.h, _vp(int(d_ptr)), int(count)))

    def close(self):
        if getattr(self, "h", None):
            h, self.h = self.h, None
            check(load().sr_comm_destroy(h))


class BlendPlan:
    """sr_blend_plan: device workspace for one tile arrangement (optionally one canvas strip)."""

    def __init__(self, ctx: Context, rects_xywh: Sequence[Tuple[int, int, int, int]], cn: int, canvas_h: int,
                 canvas_w: int, levels: int = 6, weight_type: str = "cosine", row_begin: int = 0,
                 row_end: Optional[int] = None):
        self.ctx = ctx
        self.n = len(rects_xywh)
        self.cn = cn
        self.canvas_h, self.canvas_w = int(canvas_h), int(canvas_w)
        self.row_begin = int(row_begin)
        self.row_end = int(canvas_h if row_end is None else row_end)
        rects = (TileRect * self.n)(*[TileRect(int(x), int(y), int(w), int(h)) for (x, y, w, h) in rects_xywh])
        h = C.c_void_p()
        check(ctx.lib.sr_blend_plan_create(ctx.handle, rects, self.n, cn, self.canvas_h, self.canvas_w, int(levels),
                                           WEIGHT_TYPES[weight_type], self.row_begin, self.row_end, C.byref(h)))
        self.handle = h

    def tile_rows(self, t: int) -> Tuple[int, int]:
        a, b = C.c_int(0), C.c_int(0)
        check(self.ctx.lib.sr_blend_plan_tile_rows(self.handle, t, C.byref(a), C.byref(b)))
        return a.value, b.value

    def workspace_bytes(self) -> int:
        b = C.c_size_t(0)
        check(self.ctx.lib.sr_blend_plan_workspace_bytes(self.handle, C.byref(b)))
        return b.value

    def blend(self, d_tiles: Sequence[int], strides: Sequence[int], d_canvas: int, canvas_stride: int,
              dtype: int = SR_U8, d_canvas_f32: Optional[int] = None, laplacian: bool = True):
        ptrs = (C.c_void_p * self.n)(*[C.c_void_p(p) for p in d_tiles])
        st = (C.c_int64 * self.n)(*[int(s) for s in strides])
        fn = self.ctx.lib.sr_laplacian_blend if laplacian else self.ctx.lib.sr_weighted_blend
        check(fn(self.handle, dtype, ptrs, st, C.c_void_p(d_canvas), int(canvas_stride),
                 C.c_void_p(d_canvas_f32) if d_canvas_f32 else None))

    def blend_custom_weights(self, d_tiles, strides, d_weights, weight_strides, d_canvas: int, canvas_stride: int,
                             dtype: int = SR_U8, d_canvas_f32: Optional[int] = None):
        ptrs, st = self._tile_args(d_tiles, strides)
        wp = (C.c_void_p * self.n)(*[C.c_void_p(p) for p in d_weights])
        ws = (C.c_int64 * self.n)(*[int(s) for s in weight_strides])
        check(self.ctx.lib.sr_weighted_blend_custom(self.handle, dtype, ptrs, st, wp, ws, C.c_void_p(d_canvas),
                                                    int(canvas_stride), C.c_void_p(d_canvas_f32) if d_canvas_f32 else None))

    def _tile_args(self, d_tiles, strides):
        return ((C.c_void_p * self.n)(*[C.c_void_p(p) for p in d_tiles]), (C.c_int64 * self.n)(*[int(s) for s in strides]))

    def pyramids(self, d_tiles: Sequence[int], strides: Sequence[int], tile_idx: Sequence[int], first: bool,
                 dtype: int = SR_U8):
        """Stage A for the listed tiles (first=True also builds the weight pyramids)."""
        ptrs, st = self._tile_args(d_tiles, strides)
        idx = (C.c_int * max(len(tile_idx), 1))(*[int(t) for t in tile_idx])
        check(self.ctx.lib.sr_blend_pyramids(self.handle, dtype, ptrs, st, idx, len(tile_idx), 1 if first else 0))

    def gather(self, d_tiles: Sequence[int], strides: Sequence[int], d_canvas: int, canvas_stride: int,
               dtype: int = SR_U8, d_canvas_f32: Optional[int] = None):
        """Stage B: canvas gather over all tiles."""
        ptrs, st = self._tile_args(d_tiles, strides)
        check(self.ctx.lib.sr_blend_gather(self.handle, dtype, ptrs, st, C.c_void_p(d_canvas), int(canvas_stride),
                                           C.c_void_p(d_canvas_f32) if d_canvas_f32 else None))

    def close(self):
        if getattr(self, "handle", None):
            self.ctx.lib.sr_blend_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


# ------------------------------------------------------------------------------------------
# LPIPS (sr_lpips_*): caller-supplied weights, state-dict names of lpips.LPIPS(net) (lpips >= 0.1.4)
# ------------------------------------------------------------------------------------------
LPIPS_NETS = {"alex": 0, "vgg": 1}
# convolutions of the backbones in forward order: "net.slice<S>.<index in torchvision's features>"
LPIPS_CONV_KEYS = {
    "alex": ["net.slice1.0", "net.slice2.3", "net.slice3.6", "net.slice4.8", "net.slice5.10"],
    "vgg": ["net.slice1.0", "net.slice1.2", "net.slice2.5", "net.slice2.7", "net.slice3.10", "net.slice3.12",
            "net.slice3.14", "net.slice4.17", "net.slice4.19", "net.slice4.21", "net.slice5.24", "net.slice5.26",
            "net.slice5.28"],
}
LPIPS_CONV_SHAPES = {
    "alex": [(64, 3, 11), (192, 64, 5), (384, 192, 3), (256, 384, 3), (256, 256, 3)],
    "vgg": [(64, 3, 3), (64, 64, 3), (128, 64, 3), (128, 128, 3), (256, 128, 3), (256, 256, 3), (256, 256, 3),
            (512, 256, 3), (512, 512, 3), (512, 512, 3), (512, 512, 3), (512, 512, 3), (512, 512, 3)],
}
LPIPS_TAP_CHANNELS = {"alex": (64, 192, 384, 256, 256), "vgg": (64, 128, 256, 512, 512)}


def load_lpips_weights(path: str) -> dict:
    """Reads a flat .npz of ``lpips.LPIPS(net).state_dict()`` arrays (numpy.load with allow_pickle=False: nothing in the
    file is executed).  Returns {name: ndarray}."""
    with np.load(path, allow_pickle=False) as z:
        return {k: np.asarray(z[k]) for k in z.files}


def lpips_pack_weights(net: str, weights: dict):
    """Validates a state-dict of arrays against the backbone's shapes -> (conv weights, conv biases, lin weights, shift,
    scale) as contiguous fp32 arrays in forward order.  Host only."""
    if net not in LPIPS_NETS:
        raise ValueError(f"LPIPS net must be 'alex' or 'vgg', got {net!r}")
    convs_w, convs_b = [], []
    for key, (co, ci, k) in zip(LPIPS_CONV_KEYS[net], LPIPS_CONV_SHAPES[net]):
        if key + ".weight" not in weights or key + ".bias" not in weights:
            raise ValueError(f"LPIPS weights for {net!r} lack {key}.weight / .bias")
        w = np.ascontiguousarray(weights[key + ".weight"], dtype=np.float32)
        b = np.ascontiguousarray(weights[key + ".bias"], dtype=np.float32)
        if w.shape != (co, ci, k, k) or b.shape != (co,):
            raise ValueError(f"LPIPS weight {key}: expected {(co, ci, k, k)} / {(co,)}, got {w.shape} / {b.shape}")
        convs_w.append(w)
        convs_b.append(b)
    lins = []
    for i, c in enumerate(LPIPS_TAP_CHANNELS[net]):
        name = next((n for n in (f"lin{i}.model.1.weight", f"lins.{i}.model.1.weight", f"lin{i}.model.0.weight")
                     if n in weights), None)
        if name is None:
            raise ValueError(f"LPIPS weights for {net!r} lack lin{i}.model.1.weight")
        lw = np.ascontiguousarray(np.asarray(weights[name], dtype=np.float32).reshape(-1))
        if lw.shape != (c,):
            raise ValueError(f"LPIPS weight {name}: expected {c} values, got {lw.shape}")
        lins.append(lw)
    shift = np.ascontiguousarray(weights.get("scaling_layer.shift", np.array([-.030, -.088, -.188])), dtype=np.float32).reshape(-1)
    scale = np.ascontiguousarray(weights.get("scaling_layer.scale", np.array([.458, .448, .450])), dtype=np.float32).reshape(-1)
    if shift.shape != (3,) or scale.shape != (3,):
        raise ValueError("LPIPS scaling_layer.shift / .scale must hold 3 values")
    return convs_w, convs_b, lins, shift, scale


def lpips_synthetic_weights(net: str, seed: int = 20260313) -> dict:
    """Seeded random weights of the real shapes (He-scaled convolutions, non-negative 1x1 `lin` layers) for timing and
    structural tests: the pretrained LPIPS weights cannot be fetched offline.  NOT a substitute for them."""
    rng = np.random.default_rng(seed)
    w = {}
    for key, (co, ci, k) in zip(LPIPS_CONV_KEYS[net], LPIPS_CONV_SHAPES[net]):
        w[key + ".weight"] = (rng.standard_normal((co, ci, k, k)) * np.sqrt(2.0 / (ci * k * k))).astype(np.float32)
        w[key + ".bias"] = rng.uniform(0, 0.1, co).astype(np.float32)
    for i, c in enumerate(LPIPS_TAP_CHANNELS[net]):
        w[f"lin{i}.model.1.weight"] = rng.uniform(0, 2.0 / c, (1, c, 1, 1)).astype(np.float32)
    return w


class LpipsModel:
    """sr_lpips_model: one backbone + lin layers resident on the GPU."""

    def __init__(self, ctx: Context, net: str, weights: dict):
        convs_w, convs_b, lins, shift, scale = lpips_pack_weights(net, weights)
        self.ctx, self.net = ctx, net
        n = len(convs_w)
        pw = (C.c_void_p * n)(*[a.ctypes.data for a in convs_w])
        pb = (C.c_void_p * n)(*[a.ctypes.data for a in convs_b])
        pl = (C.c_void_p * 5)(*[a.ctypes.data for a in lins])
        h = C.c_void_p()
        check(ctx.lib.sr_lpips_create(ctx.handle, LPIPS_NETS[net], pw, pb, n, pl, 5, shift.ctypes.data_as(C.c_void_p),
                                      scale.ctypes.data_as(C.c_void_p), C.byref(h)))
        self.handle = h

    def layer_sizes(self, h: int, w: int):
        out = (C.c_int * 10)()
        check(self.ctx.lib.sr_lpips_layer_sizes(LPIPS_NETS[self.net], int(h), int(w), out))
        return [(out[2 * i], out[2 * i + 1]) for i in range(5)]

    def tile_count(self, h: int, w: int, tile: int) -> int:
        n = C.c_int(0)
        check(self.ctx.lib.sr_lpips_tile_count(int(h), int(w), int(tile), C.byref(n)))
        return n.value

    def layer_sums(self, d_a: int, stride_a: int, d_b: int, stride_b: int, h: int, w: int, cn: int, tile: int = 4096,
                   tile_begin: int = 0, tile_end: int = -1):
        """Per-tap sums of the lin maps over tiles [tile_begin, tile_end) (additive across disjoint tile ranges)."""
        out = (C.c_double * 5)()
        check(self.ctx.lib.sr_lpips_u8(self.handle, C.c_void_p(d_a), int(stride_a), C.c_void_p(d_b), int(stride_b), int(h),
                                       int(w), int(cn), int(tile), int(tile_begin), int(tile_end), out))
        return [out[i] for i in range(5)]

    def value(self, d_a: int, stride_a: int, d_b: int, stride_b: int, h: int, w: int, cn: int, tile: int = 4096,
              per_layer: bool = False):
        sums = self.layer_sums(d_a, stride_a, d_b, stride_b, h, w, cn, tile)
        vals = [s / (lh * lw) for s, (lh, lw) in zip(sums, self.layer_sizes(h, w))]
        return (float(sum(vals)), vals) if per_layer else float(sum(vals))

    def close(self):
        if getattr(self, "handle", None):
            self.ctx.lib.sr_lpips_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


_default_ctx = {}
_default_lock = threading.Lock()


def default_context(device: int = 0) -> Context:
    """Process-wide context per device.  Raises SrNativeError when no MI355X / HIP device exists."""
    with _default_lock:
        ctx = _default_ctx.get(device)
        if ctx is None:
            ctx = Context(device)
            _default_ctx[device] = ctx
        return ctx

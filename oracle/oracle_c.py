"""ctypes front end of oracle/sr_oracle.c (the fast CPU checker / CPU baseline).

TEST INFRASTRUCTURE ONLY -- see the header of sr_oracle.c.  ``build()`` compiles the
C file with gcc (``-O2 -ffp-contract=off -fopenmp``) into oracle/libsr_oracle.so.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "sr_oracle.c")
_LIB = os.path.join(_HERE, "libsr_oracle.so")
_lib = None

WTYPE = {"linear": 0, "cosine": 1, "sigmoid": 2}
PADMODE = {"mirror": 0, "replicate": 1, "reflect": 2, "constant": 3}
SSIM_MODE = {"uniform": 0, "gauss": 1, "simple": 2}


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC):
        cmd = ["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", "-fvisibility=hidden",
               "-o", _LIB, _SRC, "-lm"]
        subprocess.check_call(cmd)
    return _LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.orc_psnr_u8.restype = C.c_double
        _lib.orc_ssim_u8.restype = C.c_double
        _lib.orc_laplacian_fusion.restype = C.c_int
        _lib.orc_weighted_fusion.restype = C.c_int
        _lib.orc_get_threads.restype = C.c_int
        _lib.orc_set_threads(default_threads())
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def pyr_down(src: np.ndarray) -> np.ndarray:
    s = np.ascontiguousarray(src, dtype=np.float32)
    h, w = s.shape[:2]
    cn = s.shape[2] if s.ndim == 3 else 1
    out = np.empty(((h + 1) // 2, (w + 1) // 2) + s.shape[2:], dtype=np.float32)
    lib().orc_pyr_down_f32(_p(s), h, w, cn, _p(out))
    return out


def pyr_up(src: np.ndarray, dst_hw: Tuple[int, int]) -> np.ndarray:
    s = np.ascontiguousarray(src, dtype=np.float32)
    hs, ws = s.shape[:2]
    cn = s.shape[2] if s.ndim == 3 else 1
    out = np.empty(tuple(dst_hw) + s.shape[2:], dtype=np.float32)
    lib().orc_pyr_up_f32(_p(s), hs, ws, cn, _p(out), int(dst_hw[0]), int(dst_hw[1]))
    return out


def weight_lut(fw: int, weight_type: str = "cosine") -> np.ndarray:
    out = np.empty(fw + 1, dtype=np.float32)
    lib().orc_weight_lut(fw, WTYPE[weight_type], _p(out))
    return out


def _fusion(fn, tiles, positions, output_shape, extra, return_float):
    tiles = [np.ascontiguousarray(t) for t in tiles]
    is_f32 = tiles[0].dtype != np.uint8
    if is_f32:
        tiles = [np.ascontiguousarray(t, dtype=np.float32) for t in tiles]
    cn = tiles[0].shape[2] if tiles[0].ndim == 3 else 1
    n = len(tiles)
    if output_shape is None:
        output_shape = (max(p[0] + t.shape[0] for t, p in zip(tiles, positions)),
                        max(p[1] + t.shape[1] for t, p in zip(tiles, positions)))
    H, W = int(output_shape[0]), int(output_shape[1])
    ptrs = (C.c_void_p * n)(*[t.ctypes.data for t in tiles])
    hw = np.array([[t.shape[0], t.shape[1]] for t in tiles], dtype=np.int32)
    pos = np.array([[p[0], p[1]] for p in positions], dtype=np.int32)
    out = np.empty((H, W) if cn == 1 else (H, W, cn), dtype=np.uint8)
    outf = np.empty(out.shape, dtype=np.float32) if return_float else None
    rc = fn(ptrs, _p(hw), _p(pos), n, cn, int(is_f32), H, W, *extra, _p(out),
            _p(outf) if outf is not None else None)
    if rc != 0:
        raise ValueError("oracle: tile with min side < 8 (reference divides by a zero feather width)")
    return (out, outf) if return_float else out


def laplacian_fusion(tiles: Sequence[np.ndarray], positions, output_shape=None, levels: int = 6,
                     weight_type: str = "cosine", return_float: bool = False):
    """positions are (y, x)."""
    return _fusion(lib().orc_laplacian_fusion, tiles, positions, output_shape,
                   (int(levels), WTYPE[weight_type]), return_float)


def weighted_average_fusion(tiles, positions, output_shape=None, weight_type: str = "cosine",
                            return_float: bool = False):
    return _fusion(lib().orc_weighted_fusion, tiles, positions, output_shape,
                   (WTYPE[weight_type],), return_float)


def tile_extract_pad(img: np.ndarray, x, y, w, h, block, mode: str) -> np.ndarray:
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape[:2]
    cn = img.shape[2] if img.ndim == 3 else 1
    out = np.empty((block, block) + img.shape[2:], dtype=np.uint8)
    lib().orc_tile_extract_pad(_p(img), H, W, cn, x, y, w, h, block, PADMODE[mode], _p(out))
    return out


def rgb2gray_u8(img: np.ndarray, shift: int = 15) -> np.ndarray:
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty(img.shape[:2], dtype=np.uint8)
    lib().orc_rgb2gray_u8(_p(img), C.c_size_t(out.size), shift, _p(out))
    return out


def psnr(a: np.ndarray, b: np.ndarray, data_range: float = 255.0) -> float:
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    assert a.shape == b.shape
    rowlen = a.size // a.shape[0]
    return float(lib().orc_psnr_u8(_p(a), C.c_size_t(rowlen), _p(b), C.c_size_t(rowlen), a.shape[0],
                                   C.c_size_t(rowlen), C.c_double(data_range)))


def ssim(g1: np.ndarray, g2: np.ndarray, mode: str = "gauss", data_range: float = 255.0) -> float:
    g1 = np.ascontiguousarray(g1, dtype=np.uint8)
    g2 = np.ascontiguousarray(g2, dtype=np.uint8)
    assert g1.shape == g2.shape and g1.ndim == 2
    return float(lib().orc_ssim_u8(_p(g1), _p(g2), g1.shape[0], g1.shape[1], SSIM_MODE[mode],
                                   C.c_double(data_range)))


def resize_cubic_u8(img: np.ndarray, dst_w: int, dst_h: int) -> np.ndarray:
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape[:2]
    cn = img.shape[2] if img.ndim == 3 else 1
    out = np.empty((dst_h, dst_w) + img.shape[2:], dtype=np.uint8)
    lib().orc_resize_cubic_u8(_p(img), h, w, cn, _p(out), dst_h, dst_w)
    return out


def default_threads() -> int:
    """OMP_NUM_THREADS if set, else the CPUs this process may use, capped at 16 (a GPU box gives a
    one-GPU job a 16-CPU share although it shows every core; oversubscribed OpenMP crawls)."""
    env = os.environ.get("OMP_NUM_THREADS")
    if env and env.isdigit() and int(env) > 0:
        return int(env)
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def set_threads(n: int) -> None:
    lib().orc_set_threads(int(n))


def num_threads() -> int:
    return int(lib().orc_get_threads())

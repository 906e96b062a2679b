"""tiling_module -- MI355X-native mirror of the reference's tiling_module.py call surface.

Keeps the reference's names, fields, integer bookkeeping and error behaviour
(tiling_module.py:40-171,373-425,428-504,572-646,671-852,1074-1175).  Tile positions, overlaps and
the neighbour graph come from the C ABI's host functions (bit-exact restatements); tile extraction
with border padding and the feather merge run as HIP kernels.  No CPU compute fallback.

Not on this path: ContentAnalyzer (Haar / MSER / saliency need OpenCV models) -- tiles
are never moved by it in the reference either (positions are always the uniform grid), so only the
``roi_flags`` annotation is absent.  The L1/L2 tile caches and the JSON checkpoint are host-side
persistence outside the tile -> blend -> assess path and are NOT rebuilt; only the constructor's cache-directory
side effect and the ``restore_from_cache`` probe of main.py:299-304 exist.

Reference quirks kept: the last-row/column overlap override (can exceed the tile size), merge_tiles
resizing padded tiles into the unpadded output size and casting without clip, the cache directory
created by the constructor.  Image decode uses Pillow (cv2 is not a dependency): same RGB result as
cv2.imread + BGR2RGB for 8-bit files.
"""
from __future__ import annotations

import hashlib
import json
import logging
import os
import threading
import time
import uuid
from dataclasses import asdict, dataclass, field
from enum import Enum, auto
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np

import _native

logger = logging.getLogger(__name__)


class PaddingMode(Enum):
    MIRROR = "mirror"        # cv2.BORDER_REFLECT_101
    REPLICATE = "replicate"  # cv2.BORDER_REPLICATE
    REFLECT = "reflect"      # cv2.BORDER_REFLECT
    CONSTANT = "constant"    # zeros


class TileStatus(Enum):
    PENDING = auto()
    PROCESSING = auto()
    COMPLETED = auto()
    FAILED = auto()
    CACHED = auto()


class CacheLevel(Enum):
    L1_MEMORY = "L1"
    L2_DISK = "L2"
    L3_CLOUD = "L3"


@dataclass
class TileMetadata:
    """Per-tile record (tiling_module.py:64-125)."""
    block_id: str = field(default_factory=lambda: str(uuid.uuid4()))
    global_x: int = 0
    global_y: int = 0
    input_w: int = 2048
    input_h: int = 2048
    output_w: int = 4096
    output_h: int = 4096
    overlap_top: int = 0
    overlap_bottom: int = 0
    overlap_left: int = 0
    overlap_right: int = 0
    roi_flags: Dict[str, bool] = field(default_factory=dict)
    status: TileStatus = TileStatus.PENDING
    neighbor_ids: Dict[str, Optional[str]] = field(default_factory=lambda: {
        "top": None, "bottom": None, "left": None, "right": None})
    image_hash: str = ""
    complexity_score: float = 0.0
    priority: int = 0
    created_at: float = field(default_factory=time.time)
    updated_at: float = field(default_factory=time.time)

    def to_dict(self) -> Dict:
        data = asdict(self)
        data['status'] = self.status.name
        return data

    @classmethod
    def from_dict(cls, data: Dict) -> 'TileMetadata':
        data = dict(data)
        data['status'] = TileStatus[data['status']]
        return cls(**data)


@dataclass
class Tile:
    """Tile pixels + metadata (tiling_module.py:128-171)."""
    metadata: TileMetadata
    data: Optional[np.ndarray] = None
    mask: Optional[np.ndarray] = None
    cache_path: Optional[str] = None

    def get_overlap_region(self) -> Tuple[int, int, int, int]:
        m = self.metadata
        return (m.overlap_top, m.overlap_bottom, m.overlap_left, m.overlap_right)

    def get_effective_region(self) -> Tuple[int, int, int, int]:
        m = self.metadata
        x1 = m.global_x + m.overlap_left
        y1 = m.global_y + m.overlap_top
        x2 = x1 + m.input_w - m.overlap_left - m.overlap_right
        y2 = y1 + m.input_h - m.overlap_top - m.overlap_bottom
        return (x1, y1, x2, y2)


class DeviceTileSet:
    """What split_array(device_resident=True) leaves in HBM: the source image and its n padded block x block tiles."""

    def __init__(self, ctx, d_img, d_tiles, n: int, block: int, image_h: int, image_w: int):
        self.ctx, self.d_img, self.d_tiles = ctx, d_img, d_tiles
        self.n, self.block, self.image_h, self.image_w = n, block, image_h, image_w

    @property
    def tile_bytes(self) -> int:
        return self.block * self.block * 3

    def tile_ptr(self, i: int) -> int:
        return self.d_tiles.ptr + i * self.tile_bytes

    def free(self):
        self.ctx.sync()
        for b in (self.d_img, self.d_tiles):
            if b is not None:
                b.free()
        self.d_img = self.d_tiles = None


def _load_rgb(image_path: str) -> np.ndarray:
    from PIL import Image
    try:
        with Image.open(image_path) as im:
            return np.asarray(im.convert("RGB"), dtype=np.uint8)
    except Exception as exc:  # noqa: BLE001 - cv2.imread returns None for anything unreadable
        raise ValueError(f"无法加载图像: {image_path}") from exc


class TilingModule:
    """Overlap tiling of an image and feather re-assembly (tiling_module.py:428-1217)."""

    def __init__(self, block_size: int = 2048, overlap_ratio: float = 0.2, padding_mode: str = 'mirror',
                 output_scale: float = 2.0, l1_cache_size: int = 50, l2_cache_dir: Optional[str] = None,
                 enable_content_aware: bool = True, device: int = 0):
        if not (0.1 <= overlap_ratio <= 0.3):
            raise ValueError(f"重叠率必须在0.1-0.3之间，当前值: {overlap_ratio}")
        self.block_size = block_size
        self.overlap_ratio = overlap_ratio
        self.padding_mode = PaddingMode(padding_mode)
        self.output_scale = output_scale
        self.enable_content_aware = enable_content_aware
        self.output_size = int(block_size * output_scale)
        self.overlap_pixels = int(block_size * overlap_ratio)
        self.content_analyzer = None          # out of scope, see module docstring
        self.l1_cache_size = l1_cache_size    # accepted for signature parity; the tile caches are out of scope
        if l2_cache_dir is None:
            l2_cache_dir = os.path.expanduser("~/.cache/super_resolution/tiling")
        self.l2_cache_dir = Path(l2_cache_dir)
        self.l2_cache_dir.mkdir(parents=True, exist_ok=True)
        self.tile_registry: Dict[str, Tile] = {}
        self.registry_lock = threading.Lock()
        self.processing_state: Dict[str, dict] = {}
        self.device = device
        self.device_tiles: Optional[DeviceTileSet] = None
        logger.info("TilingModule初始化完成: block_size=%s, overlap_ratio=%s, padding_mode=%s",
                    block_size, overlap_ratio, padding_mode)

    def _ctx(self) -> "_native.Context":
        return _native.default_context(self.device)

    # -- bookkeeping (host functions of the C ABI) ---------------------------------------------
    def _compute_image_hash(self, image_path: str) -> str:
        md5 = hashlib.md5()
        with open(image_path, "rb") as f:
            for chunk in iter(lambda: f.read(8192), b""):
                md5.update(chunk)
        return md5.hexdigest()

    def _calculate_tile_positions(self, image_width: int, image_height: int) -> List[Tuple[int, int, int, int]]:
        return _native.tile_plan(image_width, image_height, self.block_size, self.overlap_pixels)

    def _calculate_overlap_for_tile(self, x: int, y: int, w: int, h: int, image_width: int,
                                    image_height: int) -> Tuple[int, int, int, int]:
        return _native.tile_overlaps(x, y, w, h, image_width, image_height, self.block_size, self.overlap_pixels)

    def _apply_padding(self, image: np.ndarray, pad_top: int, pad_bottom: int, pad_left: int,
                       pad_right: int) -> np.ndarray:
        """cv2.copyMakeBorder equivalent on the GPU (bottom/right pads, as split_image uses it)."""
        if pad_top or pad_left:
            raise NotImplementedError("only bottom/right padding is on the tiling path (tiling_module.py:718-724)")
        img = np.ascontiguousarray(image, dtype=np.uint8)
        h, w = img.shape[:2]
        if pad_bottom != pad_right + (w - h) and (h + pad_bottom != w + pad_right):
            raise NotImplementedError("the HIP extract pads to a square block")
        block = h + pad_bottom
        cn = img.shape[2] if img.ndim == 3 else 1
        ctx = self._ctx()
        d_img = ctx.upload(img)
        d_out = ctx.alloc(block * block * cn)
        ctx.tile_extract_pad(d_img.ptr, h, w, cn, w * cn, [(0, 0, w, h)], block, self.padding_mode.value, d_out.ptr)
        out = ctx.download(d_out.ptr, (block, block) + img.shape[2:], np.uint8)
        d_img.free(); d_out.free()
        return out

    def create_tile_metadata(self, tile: Tile, global_x: int, global_y: int) -> TileMetadata:
        m = tile.metadata
        m.global_x, m.global_y, m.updated_at = global_x, global_y, time.time()
        return m

    # -- split (tiling_module.py:671-784) ----------------------------------------------------------
    def split_image(self, image_path: str, save_metadata: bool = True) -> List[Tile]:
        image = _load_rgb(image_path)
        return self.split_array(image, image_hash=self._compute_image_hash(image_path),
                                save_metadata=save_metadata, image_path=image_path)

    def split_array(self, image: np.ndarray, image_hash: str = "", save_metadata: bool = True,
                    image_path: str = "", device_resident: bool = False) -> List[Tile]:
        """split_image on an in-memory RGB u8 array (extension: the reference only takes a path).

        ``device_resident=True`` (the pipeline's mode): the image is uploaded once and stays in HBM together with the
        padded tiles (``self.device_tiles``); the returned tiles carry metadata only (``data is None``), the complexity
        score comes from exact gray moments taken on the GPU -- no pixel comes back to the host."""
        image = np.ascontiguousarray(image, dtype=np.uint8)
        if image.ndim != 3 or image.shape[2] != 3:
            raise ValueError("split_array expects an HxWx3 uint8 RGB image")
        ih, iw = image.shape[:2]
        positions = self._calculate_tile_positions(iw, ih)
        n, block = len(positions), self.block_size
        ctx = self._ctx()
        d_img = ctx.upload(image)
        d_tiles = ctx.alloc(n * block * block * 3)
        ctx.tile_extract_pad(d_img.ptr, ih, iw, 3, iw * 3, positions, block, self.padding_mode.value, d_tiles.ptr)
        data, scores = None, None
        if device_resident:
            self.release_device_tiles()
            self.device_tiles = DeviceTileSet(ctx, d_img, d_tiles, n, block, ih, iw)
            if self.enable_content_aware:
                scores = ctx.gray_std_u8(d_tiles.ptr, n, block * block * 3, block * 3, block, block)
        else:
            data = ctx.download(d_tiles.ptr, (n, block, block, 3), np.uint8)
            d_img.free(); d_tiles.free()
        tiles: List[Tile] = []
        for idx, (x, y, w, h) in enumerate(positions):
            top, bottom, left, right = self._calculate_overlap_for_tile(x, y, w, h, iw, ih)
            meta = TileMetadata(global_x=x, global_y=y, input_w=w, input_h=h,
                                output_w=int(w * self.output_scale), output_h=int(h * self.output_scale),
                                overlap_top=top, overlap_bottom=bottom, overlap_left=left, overlap_right=right,
                                image_hash=image_hash, status=TileStatus.PENDING)
            tile_img = data[idx] if data is not None else None
            if scores is not None:
                meta.complexity_score = float(scores[idx])
            elif self.enable_content_aware:
                # the reference applies COLOR_BGR2GRAY to RGB data (tiling_module.py:748): R/B swapped
                t = tile_img.astype(np.int64)
                gray = (t[..., 0] * 3735 + t[..., 1] * 19235 + t[..., 2] * 9798 + (1 << 14)) >> 15
                meta.complexity_score = float(np.std(gray.astype(np.uint8)))
            tile = Tile(metadata=meta, data=tile_img)
            tiles.append(tile)
            if save_metadata:
                with self.registry_lock:
                    self.tile_registry[meta.block_id] = tile
        self._build_neighbor_relationships(tiles)
        self.processing_state[image_hash] = {
            'image_path': image_path, 'image_width': iw, 'image_height': ih, 'num_tiles': len(tiles),
            'tile_ids': [t.metadata.block_id for t in tiles], 'timestamp': time.time()}
        return tiles

    def release_device_tiles(self):
        ts = getattr(self, "device_tiles", None)
        if ts is not None:
            ts.free()
        self.device_tiles = None

    def _build_neighbor_relationships(self, tiles: List[Tile]):
        xywh = [(t.metadata.global_x, t.metadata.global_y, t.metadata.input_w, t.metadata.input_h) for t in tiles]
        nbr = _native.tile_neighbors(xywh, self.block_size, self.overlap_pixels)
        for tile, (top, bottom, left, right) in zip(tiles, nbr):
            ids = tile.metadata.neighbor_ids
            for key, j in (("top", top), ("bottom", bottom), ("left", left), ("right", right)):
                if j >= 0:
                    ids[key] = tiles[j].metadata.block_id

    def get_neighbor_tiles(self, tile_id: str) -> List[Tile]:
        with self.registry_lock:
            if tile_id not in self.tile_registry:
                return []
            ids = self.tile_registry[tile_id].metadata.neighbor_ids
            return [self.tile_registry[n] for n in (ids.get('top'), ids.get('bottom'), ids.get('left'), ids.get('right'))
                    if n and n in self.tile_registry]

    def load_tile_streaming(self, image_path: str, tile: Tile, use_mmap: bool = True) -> np.ndarray:
        if tile.data is not None:
            return tile.data
        m = tile.metadata
        img = _load_rgb(image_path)
        return img[m.global_y:m.global_y + m.input_h, m.global_x:m.global_x + m.input_w]

    # -- cache / checkpoint: OUT OF SCOPE (host-side persistence, not compute) --------------------
    # The reference's L1 LRU / L2 pickle cache and JSON checkpoint (tiling_module.py:373-425,899-1072) are not rebuilt.
    # What the pipeline touches is one probe (main.py:299-304: hash the file, ask for a checkpoint, ignore the answer):
    def restore_from_cache(self, image_hash: str) -> Optional[Dict]:
        """The probe main.py:299-304 makes: the checkpoint record for this image if a reference run left one in the
        cache directory, else None.  Nothing is restored from it."""
        path = self.l2_cache_dir / f"checkpoint_{image_hash}.json"
        if not path.exists():
            return None
        try:
            with open(path, 'r') as f:
                return json.load(f)
        except (OSError, ValueError) as exc:
            logger.error("恢复检查点失败: %s", exc)
            return None

    # -- feather merge (tiling_module.py:1074-1175) ----------------------------------------------------
    def merge_tiles(self, tiles: List[Tile], output_width: int, output_height: int, blending: bool = True) -> np.ndarray:
        live = [t for t in tiles if t.data is not None]
        if not live:
            return np.zeros((output_height, output_width, 3), dtype=np.uint8)
        s = self.output_scale
        descs, arrays = [], []
        # The reference casts whatever dtype the tiles carry (tiling_module.py:1104-1109: astype(float32), or cv2.resize
        # in the data's own type).  uint8 tiles take the u8 path; anything else is taken as float32 -- exact for the
        # no-resize branch of every dtype float32 represents, cv2's float INTER_LINEAR arithmetic where sizes differ
        # (a float64 or 16-bit tile that needs resizing would go through cv2's double / fixed-point path instead:
        # refused rather than approximated).
        all_u8 = all(np.asarray(t.data).dtype == np.uint8 for t in live)
        for t in live:
            m = t.metadata
            data = np.asarray(t.data)
            if not all_u8:
                resized = data.shape[0] != m.output_h or data.shape[1] != m.output_w
                if resized and data.dtype not in (np.float32, np.uint8):
                    raise NotImplementedError(f"merge_tiles: {data.dtype} tile data that needs resizing is not on the HIP path "
                                              "(cv2.resize would run its own arithmetic for that type)")
                if resized and data.dtype == np.uint8:
                    raise NotImplementedError("merge_tiles: uint8 tiles that need resizing mixed with float tiles")
                data = data.astype(np.float32)
            data = np.ascontiguousarray(data)
            if data.ndim != 3 or data.shape[2] != 3:
                raise ValueError("merge_tiles expects HxWx3 tiles")
            h, w = data.shape[:2]
            descs.append(dict(x=int(m.global_x * s), y=int(m.global_y * s), src_w=w, src_h=h,
                              out_w=m.output_w, out_h=m.output_h,
                              ov_t=int(m.overlap_top * s), ov_b=int(m.overlap_bottom * s),
                              ov_l=int(m.overlap_left * s), ov_r=int(m.overlap_right * s)))
            arrays.append(data)
        return self._ctx().feather_merge_np(arrays, descs, output_width, output_height, blending)

    def _create_blend_weight(self, tile: Tile) -> np.ndarray:
        """Linear-ramp feather weight (tiling_module.py:1137-1175); host helper for inspection --
        merge_tiles evaluates the same ramps inside the HIP kernel."""
        m = tile.metadata
        h, w = m.output_h, m.output_w
        weight = np.ones((h, w), dtype=np.float32)
        s = self.output_scale
        t, b = int(m.overlap_top * s), int(m.overlap_bottom * s)
        l, r = int(m.overlap_left * s), int(m.overlap_right * s)
        if t > 0:
            weight[:t, :] *= np.linspace(0, 1, t).reshape(-1, 1)
        if b > 0:
            weight[-b:, :] *= np.linspace(1, 0, b).reshape(-1, 1)
        if l > 0:
            weight[:, :l] *= np.linspace(0, 1, l).reshape(1, -1)
        if r > 0:
            weight[:, -r:] *= np.linspace(1, 0, r).reshape(1, -1)
        return weight

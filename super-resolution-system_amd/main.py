#!/usr/bin/env python3
"""main -- SuperResolutionPipeline / PipelineConfig entry points over the MI355X-native stages.

Keeps the reference's names, config fields and defaults, result record and five-stage order
(main.py:47-89,92-134,157-192,269-441).  Stages 1, 3 and 4 (tile -> blend -> assess) run on the GPU
through tiling_module / blending_module / quality_assessment_module.  Stage 2 of the reference is a
remote vendor API (out of scope): here it is a pluggable ``sr_backend`` whose default
is the bicubic stub of BASELINE.json's configs, executed on the GPU with the resize kernel.

What is wired differently from the reference, because the reference's own wiring cannot run
(SURVEY.md 0.3): tiles are read from ``tile.data``; ``TileInfo(image, x, y, row, col)`` is built with
positional fields; the fused ndarray is saved through Pillow; QA receives ndarrays.  The fixed x2 tile
scale of main.py:217,322 is kept as the default ``sr_scale`` (an added field; every original field
keeps its name and default).  Importing this module does not create ``super_resolution.log``.

Several GPUs (one process per GPU, RANK / WORLD_SIZE / LOCAL_RANK in the environment -- ``python main.py in out
--gpus N`` starts its own ranks, a launcher such as torchrun works too): ``process()`` runs the sharded form of the same
stages.  The SR stage's tiles are owned one set per rank (the reference's fan-out is ``ParallelBlender``'s thread pool,
blending_module.py:1665-1705; here it is processes over RCCL), the canvas is blended in horizontal strips whose
owners receive the tile rows (+ pyramid halo) they need from the tile owners, the strips are gathered on rank 0, which
assesses and writes exactly what the one-GPU run writes: same file bytes, same scores.
"""
from __future__ import annotations

import asyncio
import json
import logging
import os
import sys
import time
from dataclasses import dataclass
from datetime import datetime
from pathlib import Path
from typing import Any, Callable, Dict, List, Optional

import numpy as np

from blending_module import BlendingModule, TileInfo
from quality_assessment_module import QualityAssessmentModule
from tiling_module import Tile, TileMetadata, TilingModule

logger = logging.getLogger(__name__)


@dataclass
class PipelineConfig:
    """main.py:47-75 (same fields and defaults) + sr_scale."""
    block_size: int = 2048
    overlap_ratio: float = 0.2
    padding_mode: str = 'mirror'
    target_resolution: str = "100MP"
    seedream_strength: float = 0.5
    seedream_steps: int = 50
    blend_method: str = 'laplacian'
    num_pyramid_levels: int = 6
    max_agents: int = 60
    max_concurrent: int = 30
    enable_qa: bool = True
    qa_device: str = 'cpu'
    volc_ak: str = ""
    volc_sk: str = ""
    volc_region: str = "cn-beijing"
    sr_scale: int = 2          # the reference hard-codes 2 (main.py:217,322)
    device_resident: bool = True   # with the built-in SR stub: source uploaded once, every stage on device pointers,
                                   # only the canvas comes back for the writer (a custom sr_backend gets host arrays)


@dataclass
class PipelineResult:
    """main.py:78-89."""
    success: bool
    output_path: Optional[str]
    processing_time: float
    total_blocks: int
    successful_blocks: int
    failed_blocks: int
    quality_score: Optional[float]
    quality_report: Optional[Dict[str, Any]]
    error_message: Optional[str]


def bicubic_stub_backend(pipeline: "SuperResolutionPipeline", tile: Tile, prompt: str) -> Optional[np.ndarray]:
    """SR stand-in: cv2.INTER_CUBIC-style upscale of the padded tile by sr_scale, on the GPU."""
    s = pipeline.config.sr_scale
    data = tile.data
    return pipeline.quality_module.upsample_bicubic(data, (data.shape[0] * s, data.shape[1] * s))


class SuperResolutionPipeline:
    """tiling -> super-resolution -> blending -> quality assessment -> output."""

    def __init__(self, config: PipelineConfig, sr_backend: Optional[Callable] = None):
        self.config = config
        self.logger = logging.getLogger(self.__class__.__name__)
        self.tiling_module = TilingModule(block_size=config.block_size, overlap_ratio=config.overlap_ratio,
                                          padding_mode=config.padding_mode, output_scale=float(config.sr_scale))
        self.blending_module = BlendingModule(method=config.blend_method, num_levels=config.num_pyramid_levels)
        self.quality_module = QualityAssessmentModule(device=config.qa_device)
        self.sr_backend = sr_backend or bicubic_stub_backend
        self.sr_module = None
        self.scheduler = None

    async def __aenter__(self):
        return self

    async def __aexit__(self, exc_type, exc_val, exc_tb):
        return None

    def _calculate_target_size(self, original_size: tuple, target_resolution: str) -> tuple:
        """main.py:157-192 (the reference defines it but process() never applies it; kept for callers)."""
        width, height = original_size
        presets = {"100MP": 100, "150MP": 150, "200MP": 200}
        if target_resolution in presets:
            import _native
            return _native.target_size(int(width), int(height), presets[target_resolution])
        try:
            w, h = map(int, target_resolution.split('x'))
            return (w, h)
        except Exception:  # noqa: BLE001
            self.logger.warning("无法解析目标分辨率: %s，使用默认100MP", target_resolution)
            return (12245, 8163)

    async def _process_single_tile(self, tile: Tile, prompt: str) -> Optional[np.ndarray]:
        try:
            return self.sr_backend(self, tile, prompt)
        except Exception as exc:  # noqa: BLE001 - a failed tile is dropped from the blend (main.py:221-223,310-325)
            self.logger.error("分块 %s 处理失败: %s", tile.metadata.block_id, exc)
            return None

    async def _parallel_upscale(self, tiles: List[Tile], prompt: str) -> List[Optional[np.ndarray]]:
        sem = asyncio.Semaphore(self.config.max_concurrent)

        async def limited(t: Tile):
            async with sem:
                return await self._process_single_tile(t, prompt)

        return list(await asyncio.gather(*[limited(t) for t in tiles]))

    def _write_outputs(self, fused: np.ndarray, output_path: str, report: Optional[Dict[str, Any]]):
        """Stage 5 (main.py:399-410): TIFF-LZW / PNG (compress_level 3) / JPEG-95 by extension + the QA report JSON.
        The reference calls Pillow's single-threaded writers; here the three formats are written by the native
        multi-threaded encoders behind the C ABI (sr_encode_*): same pixels back from any decoder."""
        import _native
        Path(output_path).parent.mkdir(parents=True, exist_ok=True)
        _native.write_image(fused, output_path, png_level=3, jpeg_quality=95)
        if report:
            with open(output_path.rsplit('.', 1)[0] + '_qa_report.json', 'w', encoding='utf-8') as f:
                json.dump(report, f, indent=2, ensure_ascii=False, default=str)

    async def _process_device(self, input_path: str, output_path: str, roi_regions, start: float) -> PipelineResult:
        """The five stages with the data resident in HBM (main.py:293-410 order): the decoded source goes up once
        (the only large H2D), tiles are cut and padded, the bicubic SR stand-in runs per tile, the tiles are fused and
        the canvas assessed against the source -- all on device addresses -- and only the finished canvas comes down
        (the only large D2H) for the writer.  Same arithmetic as the host-array path, so the same results."""
        from tiling_module import _load_rgb
        tm, s = self.tiling_module, self.config.sr_scale
        self.transfers = None
        st = self.stage_times = {}
        t_mark = time.perf_counter()

        def lap(name):
            nonlocal t_mark
            ctx.sync()
            now = time.perf_counter()
            st[name] = now - t_mark
            t_mark = now

        original = _load_rgb(input_path)
        ih, iw = original.shape[:2]
        ctx = self.quality_module._ctx()
        h2d0, d2h0 = ctx.h2d_bytes, ctx.d2h_bytes
        lap("decode")
        # Stage 1: tiling (metadata on the host, pixels stay on the GPU)
        tiles = tm.split_array(original, image_hash=tm._compute_image_hash(input_path), image_path=input_path,
                               device_resident=True)
        ts = tm.device_tiles
        block, out_block = tm.block_size, tm.block_size * s
        sr_bufs, canvas = [], None
        lap("upload+tile")
        try:
            # Stage 2: SR stand-in, tile by tile, HBM -> HBM
            for i in range(len(tiles)):
                buf = ctx.alloc(out_block * out_block * 3)
                sr_bufs.append(buf)
                ctx.resize_cubic_u8(ts.tile_ptr(i), block * 3, block, block, 3, buf.ptr, out_block * 3, out_block, out_block)
            lap("sr_stub")
            # Stage 3: blending (the canvas is cropped to the un-padded image, scaled)
            rects = [(t.metadata.global_x * s, t.metadata.global_y * s, out_block, out_block) for t in tiles]
            H, W = ih * s, iw * s
            canvas = self.blending_module.fuse_device([b.ptr for b in sr_bufs], [out_block * 3] * len(tiles), rects, (H, W), 3,
                                                      laplacian=self.config.blend_method != 'weighted')
            lap("blend")
            # Stage 4: quality assessment, source (still resident from stage 1) vs canvas
            report, score = None, None
            if self.config.enable_qa:
                qa = self.quality_module.evaluate_full_reference_device(ts.d_img.ptr, (ih, iw, 3), canvas.ptr, (H, W, 3),
                                                                        scale_factor=W / iw)
                report = {'full_reference': qa,
                          'commercial': self.quality_module.evaluate_commercial(None, roi_regions or []),
                          'timestamp': datetime.now().isoformat()}
                score = qa.get('overall_score', 0)
            lap("assess")
            # Stage 5: the one download, then the writer
            fused = ctx.download(canvas.ptr, (H, W, 3), np.uint8)
            lap("download")
        finally:
            ctx.sync()
            for b in sr_bufs + ([canvas] if canvas is not None else []):
                b.free()
            tm.release_device_tiles()
        self.transfers = {"h2d_bytes": ctx.h2d_bytes - h2d0, "d2h_bytes": ctx.d2h_bytes - d2h0,
                          "source_bytes": int(original.nbytes), "canvas_bytes": int(fused.nbytes)}
        self._write_outputs(fused, output_path, report)
        st["write"] = time.perf_counter() - t_mark
        return PipelineResult(True, output_path, time.time() - start, len(tiles), len(tiles), 0, score, report, None)

    async def _process_sharded(self, input_path: str, output_path: str, roi_regions, start: float, rank: int, world: int,
                               local_rank: int) -> PipelineResult:
        """process() on ``world`` GPUs, this process being rank ``rank`` (main.py:269-441 stage order; the fan-out the
        reference does with ParallelBlender's threads, blending_module.py:1665-1705, is one process per GPU here).
          stage 1  every rank decodes the source and cuts the (input-space) tiles -- the source is the small image;
          stage 2  the SR stand-in runs for the tiles this rank OWNS (owners chosen by the exchange planner), straight
                   into the buffers the exchange sends from;
          stage 3  the canvas is split into horizontal strips; every strip owner receives the tile rows (+ pyramid halo)
                   it needs -- one grouped batch of point-to-point transfers (RCCL over xGMI; gloo in the rehearsal) --
                   and blends its rows with the kernels of the one-GPU path: bit-identical rows;
          stage 4/5  the strips are gathered on rank 0, which assesses (same calls as the one-GPU path) and writes.
        Every rank returns the same result record (rank 0's, broadcast)."""
        import torch
        import torch.distributed as dist
        import _launch
        import device_pipeline as dp
        from tiling_module import _load_rgb
        backend = os.environ.get("SR_DIST_BACKEND", "nccl")
        ndev = max(torch.cuda.device_count(), 1)
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible -- the HIP path has no CPU fallback")
        dev_index = local_rank if backend == "nccl" else local_rank % ndev    # gloo: the one-GPU rehearsal shares a card
        torch.cuda.set_device(dev_index)
        _launch.stage("GPU visible")
        created = False
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            created = True
        _launch.stage(f"process group ({backend}, world {world}) initialised")
        pipe = None
        tm, s = self.tiling_module, self.config.sr_scale
        try:
            tm.device = self.blending_module.device = dev_index
            self.quality_module.gpu_index = dev_index
            original = _load_rgb(input_path)
            ih, iw = original.shape[:2]
            # stage 1: tiling (every rank; metadata on the host, pixels on this rank's GPU)
            tiles = tm.split_array(original, image_hash=tm._compute_image_hash(input_path), image_path=input_path,
                                   device_resident=True)
            ts = tm.device_tiles
            block, out_block = tm.block_size, tm.block_size * s
            rects = [(t.metadata.global_x * s, t.metadata.global_y * s, out_block, out_block) for t in tiles]
            H, W = ih * s, iw * s
            geo = dp.Geometry(W, H, rects, 3, self.config.num_pyramid_levels, "cosine")
            pipe = dp.DevicePipeline(geo, rank, world, dev_index)
            self.shard_info = {"rank": rank, "world": world, "owned_tiles": list(pipe.owned), "strip": list(pipe.strip),
                               "bytes_received": pipe.xplan.bytes_received(rank, geo)}
            tm._ctx().sync()                                  # the tiles were cut on the module's stream
            _launch.stage("tiles cut")
            # stage 2: the SR stand-in for the tiles this rank owns, into the buffers the exchange sends from
            for t in pipe.owned:
                buf = pipe.local_tiles[t]
                pipe.ctx.resize_cubic_u8(ts.tile_ptr(t), block * 3, block, block, 3, buf.data_ptr(), buf.stride(0), out_block, out_block)
            # stage 3: rows to the strip owners, strip blend
            pending = pipe.stage_exchange(0)
            pipe.stage_blend(pending)
            _launch.stage("strip blended")
            canvas = pipe.gather_canvas(dst=0)
            _launch.stage("canvas gathered")
            result = None
            if rank == 0:
                report, score = None, None
                qctx = self.quality_module._ctx()
                if self.config.enable_qa:
                    qa = self.quality_module.evaluate_full_reference_device(ts.d_img.ptr, (ih, iw, 3), canvas.data_ptr(), (H, W, 3),
                                                                            scale_factor=W / iw)
                    report = {'full_reference': qa,
                              'commercial': self.quality_module.evaluate_commercial(None, roi_regions or []),
                              'timestamp': datetime.now().isoformat()}
                    score = qa.get('overall_score', 0)
                fused = qctx.download(canvas.data_ptr(), (H, W, 3), np.uint8)
                self._write_outputs(fused, output_path, report)
                result = PipelineResult(True, output_path, time.time() - start, len(tiles), len(tiles), 0, score, report, None)
            box = [result]
            dist.broadcast_object_list(box, src=0)
            _launch.stage("result broadcast")
            return box[0]
        finally:
            torch.cuda.synchronize()
            if pipe is not None:
                pipe.close()
            tm.release_device_tiles()
            if created:
                dist.destroy_process_group()

    async def process(self, input_path: str, output_path: str, prompt: str = "",
                      roi_regions: Optional[List[Dict]] = None) -> PipelineResult:
        start = time.time()
        import _launch
        rank, world, local_rank = _launch.dist_env()
        if (world > 1 and self.config.device_resident and self.sr_backend is bicubic_stub_backend
                and self.config.blend_method != 'weighted'):
            try:
                return await self._process_sharded(input_path, output_path, roi_regions, start, rank, world, local_rank)
            except Exception as exc:  # noqa: BLE001 - the reference reports every failure in the result record
                self.logger.error("Pipeline执行失败: %s", exc, exc_info=True)
                return PipelineResult(False, None, time.time() - start, 0, 0, 0, None, None, str(exc))
        if world > 1 and rank != 0:
            # what does not shard (a custom SR backend hands host arrays around; the weighted blend has no strip form): rank 0
            # runs the one-GPU path, the others have nothing to do
            return PipelineResult(True, None, time.time() - start, 0, 0, 0, None, None, None)
        if self.config.device_resident and self.sr_backend is bicubic_stub_backend:
            try:
                return await self._process_device(input_path, output_path, roi_regions, start)
            except Exception as exc:  # noqa: BLE001 - the reference reports every failure in the result record
                self.logger.error("Pipeline执行失败: %s", exc, exc_info=True)
                return PipelineResult(False, None, time.time() - start, 0, 0, 0, None, None, str(exc))
        try:
            # Stage 1: tiling
            tiles = self.tiling_module.split_image(input_path)
            # Stage 2: super-resolution of every tile
            results = await self._parallel_upscale(tiles, prompt)
            ok = [r for r in results if r is not None]
            failed = len(results) - len(ok)
            if not ok:
                raise RuntimeError("所有分块处理失败")
            s = self.config.sr_scale
            step = self.tiling_module.block_size - self.tiling_module.overlap_pixels
            infos = []
            for tile, res in zip(tiles, results):
                if res is not None:
                    m = tile.metadata
                    infos.append(TileInfo(res, m.global_x * s, m.global_y * s, m.global_y // step, m.global_x // step))
            # Stage 3: blending (the canvas is cropped to the un-padded image, scaled)
            from PIL import Image
            with Image.open(input_path) as im:
                iw, ih = im.size
                original = np.asarray(im.convert("RGB"), dtype=np.uint8)
            out_shape = (ih * s, iw * s)
            if self.config.blend_method == 'weighted':
                fused = self.blending_module.weighted_average_fusion(infos, output_shape=out_shape)
            else:
                fused = self.blending_module.laplacian_fusion(infos, None, output_shape=out_shape)
            # Stage 4: quality assessment
            report, score = None, None
            if self.config.enable_qa:
                qa = self.quality_module.evaluate_full_reference(original=original, upscaled=fused,
                                                                 scale_factor=fused.shape[1] / original.shape[1])
                report = {'full_reference': qa,
                          'commercial': self.quality_module.evaluate_commercial(fused, roi_regions or []),
                          'timestamp': datetime.now().isoformat()}
                score = qa.get('overall_score', 0)
            # Stage 5: output
            self._write_outputs(fused, output_path, report)
            return PipelineResult(True, output_path, time.time() - start, len(tiles), len(ok), failed, score, report, None)
        except Exception as exc:  # noqa: BLE001 - the reference reports every failure in the result record
            self.logger.error("Pipeline执行失败: %s", exc, exc_info=True)
            return PipelineResult(False, None, time.time() - start, 0, 0, 0, None, None, str(exc))


def shard_plan(image_size, config: PipelineConfig, world: int) -> Dict[str, Any]:
    """Host only: what process() on ``world`` ranks will do with an image of ``image_size`` (w, h) -- the output-space tile
    rectangles, the strip of every rank, the owner of every tile and the bytes every rank receives.  (The planner is host
    code of the C ABI, so this runs without a GPU.)"""
    import device_pipeline as dp
    tm = TilingModule(block_size=config.block_size, overlap_ratio=config.overlap_ratio, padding_mode=config.padding_mode,
                      output_scale=float(config.sr_scale))
    iw, ih = image_size
    s, out_block = config.sr_scale, config.block_size * config.sr_scale
    rects = [(x * s, y * s, out_block, out_block) for (x, y, _, _) in tm._calculate_tile_positions(iw, ih)]
    geo = dp.Geometry(iw * s, ih * s, rects, 3, config.num_pyramid_levels, "cosine")
    xp = dp.make_exchange_plan(geo, world)
    return {"canvas": [iw * s, ih * s], "rects": rects, "bounds": list(xp.bounds), "owners": list(xp.owners),
            "rows": [list(r) for r in xp.rows], "bytes_received": [xp.bytes_received(r, geo) for r in range(world)]}


async def main() -> int:
    import argparse
    ap = argparse.ArgumentParser(description="tile -> bicubic-stub SR -> Laplacian blend -> QA on MI355X GPUs")
    ap.add_argument("input")
    ap.add_argument("output")
    ap.add_argument("--block-size", type=int, default=2048)
    ap.add_argument("--sr-scale", type=int, default=2)
    ap.add_argument("--gpus", type=int, default=1, help="GPUs of this node to use: N > 1 starts one process per GPU (RCCL)")
    ap.add_argument("--deadline-s", type=float, default=1800.0,
                    help="--gpus N: ranks still running after this many seconds are terminated (status 124)")
    ap.add_argument("--plan-only", action="store_true",
                    help="no GPU work: every rank computes the shard plan of the input, the ranks compare them over the "
                         "process group and rank 0 prints it as one JSON line")
    args = ap.parse_args()
    import _launch
    if args.gpus < 1:
        print("main.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # children only: this process never touches torch or the GPU (and nothing that has is ever exec'ed)
        return _launch.launch_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:], args.deadline_s, who="main.py")
    rank, world, _ = _launch.dist_env()
    if world != args.gpus:
        print(f"main.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}", file=sys.stderr)
        return 2
    logging.basicConfig(level=logging.INFO, stream=sys.stdout)
    cfg = PipelineConfig(block_size=args.block_size, sr_scale=args.sr_scale)
    if args.plan_only:
        from PIL import Image
        with Image.open(args.input) as im:
            plan = shard_plan(im.size, cfg, world)
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            plans = [None] * world
            dist.all_gather_object(plans, plan)
            same = all(p == plans[0] for p in plans)
            dist.destroy_process_group()
            if not same:
                print("main.py: the ranks disagree on the shard plan", file=sys.stderr)
                return 3
        if rank == 0:
            print(json.dumps({"world": world, **plan}), flush=True)
        return 0
    async with SuperResolutionPipeline(cfg) as pipe:
        res = await pipe.process(args.input, args.output, prompt="")
    if rank == 0:
        print(res)
    return 0 if res.success else 1


if __name__ == "__main__":
    sys.exit(asyncio.run(main()))

"""Device-resident tile -> blend -> assess pipeline, one process per GPU.

This is the data-parallel form of the reference's stages 1/3/4 (main.py:293-379): tiles live in HBM
on the GPU that owns them, the canvas is partitioned into horizontal strips (one per rank), every
strip owner receives the tile rows (plus pyramid halo) it needs over RCCL / xGMI and blends its rows
with exactly the kernels of the single-GPU path, so its rows are bit-identical to a 1-GPU run
(SURVEY.md 8(e)).  Quality metrics are partial sums per strip + one 4-element all-reduce.

A stream of images is pipelined (``pipeline_begin / pipeline_step / pipeline_finish``): the assessment of one image
runs on a second HIP stream beside the tile stage and pyramids of the next, and with several ranks the row exchange
of the next image runs under the blend of the current one.  ``step()`` is the one-image-at-a-time form.

torch is used for what it is good at here: device buffers, streams and events and
torch.distributed (backend "nccl" == RCCL on ROCm; "gloo" for the CPU rehearsal in tests/).
All arithmetic goes through the C ABI (``_native``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

import _native

SSIM_HALO = 5   # rows of canvas a strip needs beyond its own for the 11-tap SSIM windows


@dataclass
class Geometry:
    """Output-space tile arrangement (rects are (x, y, w, h) like blending_module.TileInfo)."""
    canvas_w: int
    canvas_h: int
    rects: List[Tuple[int, int, int, int]]
    cn: int = 3
    levels: int = 6
    weight_type: str = "cosine"

    @property
    def tile_pixels(self) -> int:
        return sum(w * h for (_, _, w, h) in self.rects)

    @property
    def canvas_pixels(self) -> int:
        return self.canvas_w * self.canvas_h


def grid_geometry(tile_w: int, tile_h: int, rows: int, cols: int, ov_x: int, ov_y: Optional[int] = None,
                  levels: int = 6, weight_type: str = "cosine") -> Geometry:
    """blending_module.create_tile_grid rule: x = col*(tw-ov), y = row*(th-ov) (blending_module.py:1518-1520)."""
    ov_y = ov_x if ov_y is None else ov_y
    rects = [((i % cols) * (tile_w - ov_x), (i // cols) * (tile_h - ov_y), tile_w, tile_h)
             for i in range(rows * cols)]
    return Geometry(cols * tile_w - (cols - 1) * ov_x, rows * tile_h - (rows - 1) * ov_y, rects, 3, levels,
                    weight_type)


# SURVEY.md 8(d) geometries (BASELINE.json configs)
WORKLOADS = {
    "4MP": dict(tile_w=1366, tile_h=911, rows=2, cols=2, ov_x=273, ov_y=182),
    "100MP": dict(tile_w=4710, tile_h=3349, rows=3, cols=3, ov_x=942),
    "150MP": dict(tile_w=4412, tile_h=3162, rows=4, cols=4, ov_x=882),
    "200MP": dict(tile_w=4124, tile_h=2970, rows=5, cols=5, ov_x=825),
}


def kd_geometry(canvas_w: int, canvas_h: int, leaves: int = 32, overlap: float = 0.10, seed: int = 20260313,
                levels: int = 6, weight_type: str = "cosine") -> Geometry:
    """Non-uniform (content-aware style) tiling, BASELINE config 5 / SURVEY.md 8(d): a seeded k-d split of the canvas
    into ``leaves`` boxes (largest box first, along its longer side, at 35-65 %), each grown by ``overlap`` of its own
    size on every side and clipped to the canvas -- so neighbours overlap by at least 10 % and the rectangles are
    what a caller would pass as ``TileInfo(x, y)`` (blending_module.py:96-112).  Row-major by (y, x)."""
    rng = np.random.default_rng(seed)
    boxes = [(0, 0, canvas_w, canvas_h)]
    while len(boxes) < leaves:
        i = max(range(len(boxes)), key=lambda k: (boxes[k][2] * boxes[k][3], -k))
        x, y, w, h = boxes.pop(i)
        f = float(rng.uniform(0.35, 0.65))
        if w >= h:
            c = min(max(int(w * f), 1), w - 1)
            boxes += [(x, y, c, h), (x + c, y, w - c, h)]
        else:
            c = min(max(int(h * f), 1), h - 1)
            boxes += [(x, y, w, c), (x, y + c, w, h - c)]
    rects = []
    for (x, y, w, h) in boxes:
        mx, my = int(np.ceil(w * overlap)), int(np.ceil(h * overlap))
        x0, y0 = max(x - mx, 0), max(y - my, 0)
        x1, y1 = min(x + w + mx, canvas_w), min(y + h + my, canvas_h)
        rects.append((x0, y0, x1 - x0, y1 - y0))
    rects.sort(key=lambda r: (r[1], r[0]))
    return Geometry(canvas_w, canvas_h, rects, 3, levels, weight_type)


def workload_geometry(name: str) -> Geometry:
    if name.endswith("-kd"):                      # e.g. "200MP-kd": same canvas, non-uniform rectangles
        base = grid_geometry(**WORKLOADS[name[:-3]])
        return kd_geometry(base.canvas_w, base.canvas_h)
    return grid_geometry(**WORKLOADS[name])


# ---------------------------------------------------------------------------------------------
# strip partition + exchange plan (host only; runs identically on every rank)
# ---------------------------------------------------------------------------------------------
# The planner itself is host code of the C ABI (csrc/sr_host.cpp: sr_strip_bounds, sr_exchange_plan), so a non-Python
# host shards with the same code; these are its Python faces.
def pyramid_halo(levels: int = 6) -> int:
    """Rows of level 0 a strip recomputes beyond each of its borders: the analytic worst case of the window planner
    (sr_pyramid_halo; 155 below / 125 above at 6 levels) -- the cost model charges the larger side."""
    return max(_native.pyramid_halo(levels))


def strip_bounds(canvas_h: int, world: int, geo: Optional["Geometry"] = None) -> List[int]:
    """Strip boundaries.  Without a geometry: equal row counts.  With one: equal *work* (sr_strip_bounds) -- a strip
    [a, b) costs the assessment and gather work of its own rows plus the pyramid work of rows a - halo .. b + halo
    (clipped to the canvas), where a row's tile work is the tile pixels covering it.  Even boundaries (the kernels
    pair rows), deterministic on every rank."""
    if world <= 1 or geo is None:
        return [canvas_h * r // world for r in range(world + 1)]
    return _native.strip_bounds(geo.rects, geo.levels, canvas_h, geo.canvas_w, world)


@dataclass
class ExchangePlan:
    world: int
    bounds: List[int]
    owners: List[int]
    rows: List[Tuple[int, int]]            # canvas rows [begin, end) each rank blends (strip + SSIM halo)
    need: List[List[Tuple[int, int]]]      # need[r][t] = tile-local rows rank r reads of tile t

    def sends(self, rank: int) -> List[Tuple[int, int, int, int]]:
        """(peer, tile, r0, r1) this rank sends, in global (peer-major, tile) order."""
        out = []
        for r in range(self.world):
            if r == rank:
                continue
            for t, (a, b) in enumerate(self.need[r]):
                if a < b and self.owners[t] == rank:
                    out.append((r, t, a, b))
        return out

    def recvs(self, rank: int) -> List[Tuple[int, int, int, int]]:
        out = []
        for t, (a, b) in enumerate(self.need[rank]):
            if a < b and self.owners[t] != rank:
                out.append((self.owners[t], t, a, b))
        return sorted(out)

    def bytes_received(self, rank: int, geo: Geometry) -> int:
        return sum((b - a) * geo.rects[t][2] * geo.cn for (_, t, a, b) in self.recvs(rank))


def make_exchange_plan(geo: Geometry, world: int, halo: int = SSIM_HALO, owner_policy: str = "balanced") -> ExchangePlan:
    """sr_exchange_plan: strip bounds, rows per rank (strip + SSIM halo), the tile rows every rank reads and an owner per
    tile.  Owner policies: "balanced" (greedy: minimise the busiest rank-to-rank link -- xGMI is point-to-point, the
    heaviest pair bounds the exchange -- then the bytes added, then the owner's tile count), "roundrobin" (tile t on
    rank t % world: independent SR workers), "locality" (the rank whose strip holds the tile's centre row)."""
    bounds, rows, need, owners = _native.exchange_plan(geo.rects, geo.cn, geo.levels, geo.canvas_h, geo.canvas_w, world, halo,
                                                       owner_policy)
    return ExchangePlan(world, bounds, owners, rows, need)


class _StagedWork:
    """Work handle of the host-staged rehearsal exchange: wait() finishes the receive and copies it to the GPU."""

    def __init__(self, work, host=None, dev=None):
        self.work, self.host, self.dev = work, host, dev

    def wait(self):
        self.work.wait()
        if self.dev is not None:
            self.dev.copy_(self.host)


def exchange_ops(plan: ExchangePlan, rank: int, local_tiles: Dict[int, "object"], recv_bufs: Dict[int, "object"], group=None):
    """The P2POp list of one rank's exchange (sends in (reader, tile) order, receives in (owner, tile) order -- the order
    sr_exchange_xfers lists them in).  The buffers of a set do not change between images, so a stream of images builds
    this once per set and posts the same list every step."""
    import torch.distributed as dist
    ops = []
    for (peer, t, a, b) in plan.sends(rank):
        ops.append(dist.P2POp(dist.isend, local_tiles[t][a:b], peer, group))
    for (peer, t, a, b) in plan.recvs(rank):
        ops.append(dist.P2POp(dist.irecv, recv_bufs[t], peer, group))
    return ops


def exchange_tile_rows(plan: ExchangePlan, rank: int, local_tiles: Dict[int, "object"],
                       recv_bufs: Dict[int, "object"], group=None, ops=None):
    """Send the rows other strips need of the tiles this rank owns, receive the rows this strip
    needs of tiles owned elsewhere.  Tiles are 2-D uint8 tensors [h, w*cn]; recv_bufs[t] has exactly
    (r1 - r0) rows.  One grouped batch of point-to-point ops (ncclSend/ncclRecv under
    ncclGroupStart/End on RCCL; plain isend/irecv on gloo).  Returns the work handles.  ``ops``: a list from
    exchange_ops() for these buffers (built here when not given).

    gloo cannot move device tensors point-to-point: with that backend and GPU tensors (the single-GPU rehearsal
    of the multi-rank path) rows are staged through host memory -- a test vehicle, never the measured path."""
    import torch.distributed as dist
    sends, recvs = plan.sends(rank), plan.recvs(rank)
    if not sends and not recvs:
        return []
    on_gpu = any(t.is_cuda for t in list(local_tiles.values()) + list(recv_bufs.values()))
    if on_gpu and dist.get_backend(group) == "gloo":
        works = []
        for (peer, t, a, b) in sends:
            works.append(_StagedWork(dist.isend(local_tiles[t][a:b].cpu(), peer, group=group, tag=t)))
        for (peer, t, a, b) in recvs:
            host = recv_bufs[t].new_empty(recv_bufs[t].shape, device="cpu")
            works.append(_StagedWork(dist.irecv(host, peer, group=group, tag=t), host, recv_bufs[t]))
        return works
    if ops is None:
        ops = exchange_ops(plan, rank, local_tiles, recv_bufs, group)
    # One grouped batch (ncclGroupStart / End around every ncclSend / ncclRecv): inside a group NCCL matches the
    # transfers of a pair whatever their posting order, so two strips that send to each other cannot deadlock.  There is
    # deliberately no ungrouped fallback: posted one by one, each rank would queue its sends ahead of its receives on the
    # pair's communicator stream and multi-MB transfers would wait on each other forever -- a failure here must surface.
    return dist.batch_isend_irecv(ops)


def gather_strips(canvas, bounds: Sequence[int], rank: int, group=None, dst: int = 0):
    """The final image, only when the caller asks for the array: every rank's canvas tensor [H, W*cn] holds
    its own strip rows bounds[rank] .. bounds[rank + 1]; rank ``dst`` receives the other strips into its canvas (rows are
    contiguous: one transfer per peer, one grouped batch) and returns it, the others send and return None.  With gloo
    and GPU tensors (rehearsal) the rows are staged through host memory."""
    import torch.distributed as dist
    world = len(bounds) - 1
    if world == 1:
        return canvas
    staged = canvas.is_cuda and dist.get_backend(group) == "gloo"
    if rank != dst:
        a, b = bounds[rank], bounds[rank + 1]
        if a < b:
            rows = canvas[a:b]
            if staged:
                dist.send(rows.cpu(), dst, group=group)
            else:
                for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, rows, dst, group)]):
                    w.wait()
        return None
    peers = [(r, bounds[r], bounds[r + 1]) for r in range(world) if r != dst and bounds[r] < bounds[r + 1]]
    if staged:
        for (r, a, b) in peers:
            host = canvas.new_empty((b - a, canvas.shape[1]), device="cpu")
            dist.recv(host, r, group=group)
            canvas[a:b].copy_(host)
    elif peers:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, canvas[a:b], r, group) for (r, a, b) in peers]):
            w.wait()
    return canvas


# ---------------------------------------------------------------------------------------------
# per-rank device pipeline
# ---------------------------------------------------------------------------------------------
class DevicePipeline:
    """tile (overlap extract) -> [exchange] -> Laplacian blend of this rank's strip -> PSNR/SSIM partials.

    Inputs (resident before the timed region): ``image`` = the output-space image the tiles are cut
    from (the SR result; the benchmark's stub is a bicubic upscale), ``reference`` = the image the
    canvas is assessed against; both uint8 [H, W*3] torch tensors on this rank's GPU.
    """

    RESULT_FIELDS = ("sse", "ssim_uniform_sum", "ssim_gauss_sum", "ssim_simple_sum")

    def __init__(self, geo: Geometry, rank: int = 0, world: int = 1, device: Optional[int] = None,
                 group=None, ssim_modes: Sequence[str] = ("uniform", "gauss", "simple"), depth: int = 2):
        import torch
        self.torch = torch
        self.geo, self.rank, self.world, self.group = geo, rank, world, group
        self.device = torch.cuda.current_device() if device is None else device
        self.dev = torch.device("cuda", self.device)
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        self.ctx = _native.Context(self.device, stream=stream)
        self.ssim_modes = tuple(ssim_modes)
        self.xplan = make_exchange_plan(geo, world)
        self.row_begin, self.row_end = self.xplan.rows[rank]
        self.strip = (self.xplan.bounds[rank], self.xplan.bounds[rank + 1])
        cn = geo.cn
        u8 = dict(dtype=torch.uint8, device=self.dev)
        self.owned = [t for t, o in enumerate(self.xplan.owners) if o == rank]
        need = self.xplan.need[rank]
        self._local_needed = [t for t in range(len(geo.rects)) if need[t][0] < need[t][1] and t in self.owned]
        self._remote_needed = [t for t in range(len(geo.rects)) if need[t][0] < need[t][1] and t not in self.owned]
        self._strides = [w * cn for (_, _, w, _) in geo.rects]
        # Buffer sets: the tiles this rank owns (dense), the row windows it receives of the others, the per-tile
        # (virtual) base pointers for the blend and the metric partial sums.  Two sets when the exchange of the next
        # image is to overlap the blend of the current one (pipeline_* below); step() uses set 0 only.
        self.sets = []
        for _ in range(max(1, depth if world > 1 else 1)):
            local = {t: torch.empty((geo.rects[t][3], geo.rects[t][2] * cn), **u8) for t in self.owned}
            recv = {t: torch.empty((b - a, geo.rects[t][2] * cn), **u8) for (_, t, a, b) in self.xplan.recvs(rank)}
            ptrs = []
            for t in range(len(geo.rects)):
                a, b = need[t]
                if a >= b:
                    ptrs.append(0)
                elif t in local:
                    ptrs.append(local[t].data_ptr())
                else:
                    ptrs.append(recv[t].data_ptr() - a * self._strides[t])   # virtual row 0
            self.sets.append(dict(local=local, recv=recv, ptrs=ptrs))
        # Canvas and metric sums are double-buffered too: in a stream of images the (VALU-bound) assessment of image
        # i runs on a second HIP stream beside the (bandwidth-bound) tile stage and pyramids of image i+1.
        self.canvases = [torch.zeros((geo.canvas_h, geo.canvas_w * cn), **u8) for _ in range(2)]
        self.results_bufs = [torch.zeros(4, dtype=torch.float64, device=self.dev) for _ in range(2)]   # sr_assess_sums
        self.main_stream = torch.cuda.current_stream(self.dev)
        import os as _os
        # SR_QA_STREAM_PRIO (experiments): HIP priority of the assessment's stream (lower number = served first; the chain
        # runs on the caller's stream at the default priority 0)
        _prio = _os.environ.get("SR_QA_STREAM_PRIO")
        self.qa_stream = torch.cuda.Stream(self.dev, priority=int(_prio)) if _prio else torch.cuda.Stream(self.dev)
        self.qa_ctx = _native.Context(self.device, stream=self.qa_stream.cuda_stream)
        self._assess_split = _os.environ.get("SR_ASSESS_SPLIT", "0") == "1"
        self._qa_gate = _os.environ.get("SR_QA_GATE", "0") == "1"
        self._gated = None
        self._results_u = None
        self._e_qa = [None, None]         # assessment of the image in canvas slot j has finished
        self._slot = 0                    # canvas / result slot of the image in progress
        self.plan = _native.BlendPlan(self.ctx, geo.rects, cn, geo.canvas_h, geo.canvas_w, geo.levels,
                                      geo.weight_type, self.row_begin, self.row_end)
        self._cur = 0                 # buffer set of the step in progress
        self._pending = None          # exchange work handles of the set in progress (pipeline_*)
        self._reduce_work = [[], []]
        self._deferred = None         # slot whose metric all-reduce is still to be posted
        self._done = 0                # canvas / result slot of the last finished image
        self._first = True
        self.exchange_wait_s = 0.0    # host time spent in wait() of the exchange work handles (N > 1 diagnostics)
        self.exchange_waits = 0

    # set 0 under the names the single-step path and the tests use
    @property
    def local_tiles(self):
        return self.sets[0]["local"]

    @property
    def recv_bufs(self):
        return self.sets[0]["recv"]

    @property
    def results(self):
        return self.results_bufs[self._done]

    @property
    def canvas(self):
        return self.canvases[self._done]

    @property
    def _ptrs(self):
        return self.sets[0]["ptrs"]

    # -- stages ---------------------------------------------------------------------------------
    def stage_tile(self, image, k: int = 0):
        """Overlap-tile extract of the tiles this rank owns (tiling_module.py:713-715 slice)."""
        if not self.owned:
            return
        g, local = self.geo, self.sets[k]["local"]
        self.ctx.tile_extract(image.data_ptr(), g.canvas_h, g.canvas_w, g.cn, image.stride(0),
                              [g.rects[t] for t in self.owned],
                              [local[t].data_ptr() for t in self.owned],
                              [local[t].stride(0) for t in self.owned])

    def stage_exchange(self, k: int = 0):
        """Posts the grouped sends / receives; returns the work handles (empty on one GPU)."""
        if self.world == 1:
            return []
        st = self.sets[k]
        if "ops" not in st:                                # same buffers every image: the op list is built once per set
            import torch.distributed as dist
            staged = dist.get_backend(self.group) == "gloo" and self.dev.type == "cuda"
            st["ops"] = None if staged else exchange_ops(self.xplan, self.rank, st["local"], st["recv"], self.group)
        return exchange_tile_rows(self.xplan, self.rank, st["local"], st["recv"], self.group, ops=st["ops"])

    def stage_blend(self, pending=(), k: int = 0, slot: int = 0):
        """Pyramids of the tiles this rank already holds run while the exchange is in flight; the tiles that
        arrive are processed after the wait, then the canvas gather over all of them."""
        ptrs, canvas = self.sets[k]["ptrs"], self.canvases[slot]
        if not pending:
            self.plan.blend(ptrs, self._strides, canvas.data_ptr(), canvas.stride(0))
            return
        self.plan.pyramids(ptrs, self._strides, self._local_needed, first=True)
        for w in pending:
            w.wait()
        self.plan.pyramids(ptrs, self._strides, self._remote_needed, first=False)
        self.plan.gather(ptrs, self._strides, canvas.data_ptr(), canvas.stride(0))

    def stage_assess(self, reference, slot: int = 0, ctx=None):
        """PSNR (exact integer SSE) and the three SSIM variants over this rank's strip, as partial sums left on
        the device: one fused pass over both images -- sr_assess_u8_async (on ``ctx``'s stream)."""
        g = self.geo
        ctx = self.ctx if ctx is None else ctx
        canvas = self.canvases[slot]
        s0, s1 = self.strip
        flags = _native.ASSESS_SSE
        for mode, bit in (("uniform", _native.ASSESS_UNIFORM7), ("gauss", _native.ASSESS_GAUSS11),
                          ("simple", _native.ASSESS_SIMPLE)):
            if mode in self.ssim_modes:
                flags |= bit
        if self._assess_split and (flags & _native.ASSESS_UNIFORM7) and (flags & ~(_native.ASSESS_UNIFORM7 | _native.ASSESS_SSE)):
            # A/B arm (SR_ASSESS_SPLIT=1, profiles/r04_assess_split.json): the Gaussian(+SSE) variant of the march (no
            # uniform-7 ring: 128 registers, four blocks per CU) and the integer uniform-7 variant as two launches
            if self._results_u is None:
                self._results_u = [self.torch.zeros(4, dtype=self.torch.float64, device=self.dev) for _ in range(2)]
            args = (reference.data_ptr(), reference.stride(0), canvas.data_ptr(), canvas.stride(0), g.canvas_h, g.canvas_w, g.cn)
            ctx.assess_u8_async(*args, self.results_bufs[slot].data_ptr(), flags=flags & ~_native.ASSESS_UNIFORM7, row_begin=s0, row_end=s1)
            ctx.assess_u8_async(*args, self._results_u[slot].data_ptr(), flags=_native.ASSESS_UNIFORM7, row_begin=s0, row_end=s1)
            self.results_bufs[slot][1:2].copy_(self._results_u[slot][1:2])     # torch's current stream is the context's here
            return
        ctx.assess_u8_async(reference.data_ptr(), reference.stride(0), canvas.data_ptr(), canvas.stride(0),
                            g.canvas_h, g.canvas_w, g.cn, self.results_bufs[slot].data_ptr(), flags=flags,
                            row_begin=s0, row_end=s1)

    def stage_reduce(self, slot: int = 0, async_op: bool = False):
        if self.world == 1:
            return None
        import torch.distributed as dist
        res = self.results_bufs[slot]
        if dist.get_backend(self.group) == "gloo":          # rehearsal backend: reduce on the host
            host = res.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            res.copy_(host)
            return None
        return dist.all_reduce(res, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def step(self, image, reference):
        """One image, start to finish (buffer set 0)."""
        self.pipeline_finish()                            # a stream in progress writes the same buffers
        self.stage_tile(image)
        pending = self.stage_exchange()
        self.stage_blend(pending)
        self.stage_assess(reference)
        self.stage_reduce()
        self._done = 0

    # -- a stream of images --------------------------------------------------------------------------------------
    # Two overlaps, both across images, every image still gets every stage:
    #   * the exchange of image i+1 runs under the blend of image i (double-buffered tile / receive sets),
    #   * the assessment of image i runs on a second HIP stream under the tile stage and pyramids of image i+1
    #     (double-buffered canvas and sums): the assessment is fp64-VALU-bound and leaves HBM idle, the pyramids are
    #     bandwidth-bound and leave the VALUs idle.
    def pipeline_begin(self, image):
        """Tile stage + posted exchange of the first image of a stream."""
        self._cur = 0
        self._first = True
        self.stage_tile(image, 0)
        self._pending = self.stage_exchange(0)

    def pipeline_step(self, reference, next_image=None):
        """Finishes the image in flight.  When ``next_image`` is given its tile stage and exchange are started as soon
        as this image's rows have arrived, into the other buffer set.  The assessment is queued on the second stream
        behind this image's gather; the metric all-reduce follows it there.  One exchange batch is in flight at any
        time."""
        torch = self.torch
        k, j = self._cur, self._slot
        ptrs, pending = self.sets[k]["ptrs"], self._pending
        # Only the first image of a stream has its exchange still in flight when its blend starts (worth splitting the
        # pyramids into held / arriving tiles); later images' rows were moved under the previous image's work, so
        # they take the monolithic blend: half the kernel launches.
        staged = bool(pending) and self._first
        if staged:
            self.plan.pyramids(ptrs, self._strides, self._local_needed, first=True)
        if pending:
            import time as _time
            t_w = _time.perf_counter()
            for w in pending:
                w.wait()
            self.exchange_wait_s += _time.perf_counter() - t_w
            self.exchange_waits += 1
        self._first = False
        nk = (k + 1) % len(self.sets)
        overlap = next_image is not None and nk != k
        if overlap:                                       # start image i+1 before the rest of image i
            self.stage_tile(next_image, nk)
            self._pending = self.stage_exchange(nk)
        # The previous image's metric all-reduce is posted only now, BEHIND the exchange just posted: collectives of one
        # communicator run in posting order, and an all-reduce posted right after its assessment would sit in front of
        # this exchange until that assessment (still running beside this image's blend) has finished.
        self._post_deferred_reduce()
        if self._qa_gate and self.world == 1:
            # Schedule experiment (SR_QA_GATE=1, profiles/r04_qa_gate.json): the previous image's assessment is released
            # only when THIS image's pyramids are done, so the fp64-bound assessment starts beside the (memory-bound,
            # marched) gather and runs on beside the next image's tile stage and pyramids.
            self.plan.pyramids(ptrs, self._strides, list(range(len(self.geo.rects))), first=True)
            pyr_done = torch.cuda.Event()
            pyr_done.record(self.main_stream)
            self._release_gated(pyr_done)
            if self._e_qa[j] is not None:
                self.main_stream.wait_event(self._e_qa[j])
            canvas = self.canvases[j]
            self.plan.gather(ptrs, self._strides, canvas.data_ptr(), canvas.stride(0))
            blended = torch.cuda.Event()
            blended.record(self.main_stream)
            self._gated = (reference, j, blended)
            self._deferred = None
            if next_image is not None:
                self.stage_tile(next_image, nk)
                self._pending = self.stage_exchange(nk)
            else:
                self._pending = None
            self._done, self._cur, self._slot = j, nk, 1 - j
            return
        if self._e_qa[j] is not None:                     # canvas slot j: its previous image has been assessed
            self.main_stream.wait_event(self._e_qa[j])
        self._finish_blend(k, staged, j)
        blended = torch.cuda.Event()
        blended.record(self.main_stream)
        with torch.cuda.stream(self.qa_stream):
            self.qa_stream.wait_event(blended)
            for w in self._reduce_work[j]:                # the sums of slot j two images ago have been reduced
                w.wait()
            self.stage_assess(reference, j, self.qa_ctx)
            self._e_qa[j] = torch.cuda.Event()
            self._e_qa[j].record(self.qa_stream)
        self._deferred = j
        if next_image is not None and not overlap:        # one buffer set (single GPU): next tile stage in plain order
            self.stage_tile(next_image, nk)
            self._pending = self.stage_exchange(nk)
        elif next_image is None:
            self._pending = None
        self._done, self._cur, self._slot = j, nk, 1 - j

    def _release_gated(self, after=None):
        """SR_QA_GATE: queue the held assessment on the second stream, behind `after` (an event of the main stream)."""
        if self._gated is None:
            return
        torch = self.torch
        reference, j, blended = self._gated
        self._gated = None
        with torch.cuda.stream(self.qa_stream):
            self.qa_stream.wait_event(blended)
            if after is not None:
                self.qa_stream.wait_event(after)
            self.stage_assess(reference, j, self.qa_ctx)
            self._e_qa[j] = torch.cuda.Event()
            self._e_qa[j].record(self.qa_stream)

    def _post_deferred_reduce(self):
        j = self._deferred
        if j is None:
            return
        self._deferred = None
        with self.torch.cuda.stream(self.qa_stream):      # ordered behind the assessment that produced the sums
            w = self.stage_reduce(j, async_op=True)
        self._reduce_work[j] = [w] if w is not None else []

    def _finish_blend(self, k, staged, slot=0):
        ptrs, canvas = self.sets[k]["ptrs"], self.canvases[slot]
        if staged:
            self.plan.pyramids(ptrs, self._strides, self._remote_needed, first=False)
            self.plan.gather(ptrs, self._strides, canvas.data_ptr(), canvas.stride(0))
        else:
            self.plan.blend(ptrs, self._strides, canvas.data_ptr(), canvas.stride(0))

    def pipeline_finish(self):
        """Joins the second stream: after this the main stream (and a device synchronise) see every image's sums."""
        self._release_gated()
        self._post_deferred_reduce()
        for j in (0, 1):
            for w in self._reduce_work[j]:
                w.wait()
            self._reduce_work[j] = []
            if self._e_qa[j] is not None:
                self.main_stream.wait_event(self._e_qa[j])

    def gather_canvas(self, dst: int = 0):
        """The whole blended image on rank ``dst`` (a device tensor [H, W*cn]; None elsewhere): the strips stay on their
        GPUs unless this is called."""
        self.pipeline_finish()
        self.torch.cuda.current_stream(self.dev).synchronize()
        return gather_strips(self.canvas, self.xplan.bounds, self.rank, self.group, dst)

    # -- single-process rehearsal of a rank (tests): same buffers, same staged kernels, no communicator ------------
    def rehearse_fill(self, full_tiles: Dict[int, "object"]):
        """Put into this rank's buffers exactly what the tile stage + exchange would: its own tiles whole, and of
        the others only the rows the exchange plan delivers."""
        for t, buf in self.local_tiles.items():
            buf.copy_(full_tiles[t])
        for (_, t, a, b) in self.xplan.recvs(self.rank):
            self.recv_bufs[t].copy_(full_tiles[t][a:b])

    def rehearse_step(self, reference, staged: bool = True):
        """Blend (staged, as with an exchange in flight; or monolithic, as for the later images of a stream) + assess,
        without reduce."""
        if staged:
            self.plan.pyramids(self._ptrs, self._strides, self._local_needed, first=True)
            self.plan.pyramids(self._ptrs, self._strides, self._remote_needed, first=False)
            self.plan.gather(self._ptrs, self._strides, self.canvases[0].data_ptr(), self.canvases[0].stride(0))
        else:
            self.plan.blend(self._ptrs, self._strides, self.canvases[0].data_ptr(), self.canvases[0].stride(0))
        self.stage_assess(reference)
        self._done = 0

    # -- results ----------------------------------------------------------------------------------
    def metrics(self) -> Dict[str, float]:
        """Whole-image scores from the (all-reduced) partial sums; synchronises."""
        g = self.geo
        self.torch.cuda.synchronize(self.dev)             # both streams
        vals = self.results.cpu().numpy()
        out = {"psnr": _native.psnr_from_sse(int(round(vals[0])), g.canvas_h * g.canvas_w * g.cn, 255.0)}
        for i, mode in enumerate(("uniform", "gauss", "simple")):
            if mode in self.ssim_modes:
                out[f"ssim_{mode}"] = float(vals[1 + i] / _native.ssim_count(g.canvas_h, g.canvas_w, mode))
        return out

    def close(self):
        self.torch.cuda.synchronize(self.dev)
        self.plan.close()
        self.qa_ctx.close()
        self.ctx.close()

"""The RCCL entry points of the C ABI (sr_comm_*, csrc/sr_comm.cpp) on the one GPU a test box has: RCCL refuses two ranks
on one device, so what can execute here is a world-size-1 communicator -- creation from a unique id, a grouped batch of
ncclSend / ncclRecv to itself on the context's stream (the same code path peers take), the metric all-reduce, and the
plan-driven sr_comm_exchange_tile_rows (no transfers at world 1).  World sizes 2-8 are covered on the host side by
tests/test_planner.py::test_exchange_xfers_match_plan (the exact batches) and the gloo rehearsals."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_comm_world1_self_exchange(ctx, rng):
    import _native
    import device_pipeline as dp
    comm = _native.Comm(ctx, _native.comm_unique_id(), 1, 0)
    try:
        a = rng.integers(0, 256, 3_000_001, dtype=np.uint8)
        b = rng.integers(0, 256, 70_001, dtype=np.uint8)
        da, db = ctx.upload(a), ctx.upload(b)
        ra, rb = ctx.alloc(a.nbytes), ctx.alloc(b.nbytes)
        try:
            comm.exchange([(0, da.ptr, a.nbytes), (0, db.ptr, b.nbytes)], [(0, ra.ptr, a.nbytes), (0, rb.ptr, b.nbytes)])
            ctx.sync()
            assert np.array_equal(ctx.download(ra.ptr, a.shape, np.uint8), a)
            assert np.array_equal(ctx.download(rb.ptr, b.shape, np.uint8), b)
            sums = np.array([1.5, -2.25, 3e300, 7.0])
            ds = ctx.upload(sums)
            comm.allreduce_f64(ds.ptr, 4)
            ctx.sync()
            assert np.array_equal(ctx.download(ds.ptr, (4,), np.float64), sums)
            ds.free()
            with pytest.raises((_native.SrNativeError, ValueError)):
                comm.exchange([(1, da.ptr, 16)], [])                 # peer outside the communicator
        finally:
            for d in (da, db, ra, rb):
                d.free()
        geo = dp.workload_geometry("4MP")
        plan = dp.make_exchange_plan(geo, 1)
        n = len(geo.rects)
        tiles = [ctx.alloc(w * h * geo.cn) for (_, _, w, h) in geo.rects]
        try:
            comm.exchange_tile_rows(geo.rects, geo.cn, plan.need, plan.owners, [t.ptr for t in tiles],
                                    [w * geo.cn for (_, _, w, _) in geo.rects], [0] * n)
            ctx.sync()
        finally:
            for t in tiles:
                t.free()
    finally:
        comm.close()

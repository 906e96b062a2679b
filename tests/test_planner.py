"""The multi-GPU planner of the C ABI (sr_strip_bounds / sr_exchange_plan, host only): checked against an
independent NumPy restatement (tests/_planner_ref.py) and through the properties a strip exchange needs -- the strips tile
the canvas with even bounds, every rank's needed rows are covered by exactly one owner's send, and the balanced owner
policy never loads a rank-to-rank link more than round-robin does."""
import pytest

import _native
import _planner_ref as ref
import device_pipeline as dp

WORKLOADS = ["4MP", "100MP", "150MP", "200MP", "200MP-kd"]


@pytest.mark.parametrize("workload", WORKLOADS)
@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_matches_restatement(workload, world):
    geo = dp.workload_geometry(workload)
    for policy in ("balanced", "roundrobin", "locality"):
        got = _native.exchange_plan(geo.rects, geo.cn, geo.levels, geo.canvas_h, geo.canvas_w, world, dp.SSIM_HALO, policy)
        want = ref.make_exchange_plan(geo, world, dp.SSIM_HALO, policy)
        assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2] and got[3] == want[3], (workload, world, policy)
    assert dp.strip_bounds(geo.canvas_h, world, geo) == ref.strip_bounds(geo.canvas_h, world, geo)


@pytest.mark.parametrize("workload", ["150MP", "200MP-kd"])
@pytest.mark.parametrize("world", [2, 5, 8])
def test_plan_properties(workload, world):
    geo = dp.workload_geometry(workload)
    plan = dp.make_exchange_plan(geo, world)
    assert plan.bounds[0] == 0 and plan.bounds[-1] == geo.canvas_h
    assert all(b % 2 == 0 or b == geo.canvas_h for b in plan.bounds)
    assert all(plan.bounds[r] < plan.bounds[r + 1] for r in range(world))
    # what every rank receives is exactly what the owners send it
    sent = sorted((rank, r, t, a, b) for rank in range(world) for (r, t, a, b) in plan.sends(rank))
    recv = sorted((o, rank, t, a, b) for rank in range(world) for (o, t, a, b) in plan.recvs(rank))
    assert sent == recv

    def busiest(owners):
        link = {}
        for r in range(world):
            for t, (a, b) in enumerate(plan.need[r]):
                if a < b and owners[t] != r:
                    link[(owners[t], r)] = link.get((owners[t], r), 0) + (b - a) * geo.rects[t][2] * geo.cn
        return max(link.values())

    assert busiest(plan.owners) <= busiest([t % world for t in range(len(geo.rects))])


def test_strip_costs_balanced():
    """Equal work, not equal rows: inner strips pay the pyramid halo on both sides, so they are shorter."""
    geo = dp.workload_geometry("200MP")
    b = dp.strip_bounds(geo.canvas_h, 8, geo)
    rows = [b[r + 1] - b[r] for r in range(8)]
    assert min(rows[0], rows[-1]) > max(rows[1:-1])          # rows inside tile overlaps cost twice, so inner strips differ too
    assert sum(rows) == geo.canvas_h


def test_bad_arguments():
    geo = dp.workload_geometry("4MP")
    with pytest.raises((_native.SrNativeError, _native.SrShapeError, ValueError)):
        _native.strip_bounds(geo.rects, geo.levels, geo.canvas_h, geo.canvas_w, 0)
    with pytest.raises((_native.SrNativeError, _native.SrShapeError, ValueError)):
        _native.exchange_plan(geo.rects, geo.cn, geo.levels, geo.canvas_h, geo.canvas_w, 2, -1)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_exchange_xfers_match_plan(world):
    """sr_exchange_xfers (what sr_comm_exchange_tile_rows posts to RCCL) lists exactly ExchangePlan.sends / recvs, with the
    byte offsets of dense tiles; fake device addresses, nothing is dereferenced."""
    geo = dp.workload_geometry("200MP-kd")
    plan = dp.make_exchange_plan(geo, world)
    n = len(geo.rects)
    strides = [w * geo.cn for (_, _, w, _) in geo.rects]
    posted_s, posted_r = [], []
    for rank in range(world):
        owned = [(0x10000000 + t * 0x4000000) if plan.owners[t] == rank else 0 for t in range(n)]
        recv = [(0x7000000000 + t * 0x4000000) if (plan.owners[t] != rank and plan.need[rank][t][0] < plan.need[rank][t][1]) else 0
                for t in range(n)]
        sends, recvs = _native.exchange_xfers(geo.rects, geo.cn, world, rank, plan.need, plan.owners, owned, strides, recv)
        want_s = [(r, owned[t] + a * strides[t], (b - a) * strides[t]) for (r, t, a, b) in plan.sends(rank)]
        want_r = [(o, recv[t], (b - a) * strides[t]) for (o, t, a, b) in plan.recvs(rank)]
        assert sends == want_s and recvs == want_r
        posted_s += [(rank, p, nbytes) for (p, _, nbytes) in sends]
        posted_r += [(p, rank, nbytes) for (p, _, nbytes) in recvs]
    assert sorted(posted_s) == sorted(posted_r)              # every send has its receive, pair by pair in the same order
    for a in range(world):
        for b in range(world):
            assert [x[2] for x in posted_s if x[:2] == (a, b)] == [x[2] for x in posted_r if x[:2] == (a, b)]
    # a padded (non-dense) owned tile that must be sent is refused, a missing receive buffer too
    t_sent = plan.sends(0)[0][1]
    owned = [(0x10000000 + t * 0x4000000) if plan.owners[t] == 0 else 0 for t in range(n)]
    recv = [0x7000000000 if plan.owners[t] != 0 else 0 for t in range(n)]
    bad = list(strides)
    bad[t_sent] += 64
    with pytest.raises((_native.SrNativeError, ValueError)):
        _native.exchange_xfers(geo.rects, geo.cn, world, 0, plan.need, plan.owners, owned, bad, recv)
    if plan.recvs(0):
        recv[plan.recvs(0)[0][1]] = 0
        with pytest.raises((_native.SrNativeError, ValueError)):
            _native.exchange_xfers(geo.rects, geo.cn, world, 0, plan.need, plan.owners, owned, strides, recv)


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_tile_bases(world):
    """sr_sharded_tile_bases (host only; what sr_laplacian_blend_sharded hands to the strip blend): owned tiles keep their
    pointer, received windows are moved up to their virtual row 0 with the DENSE stride, tiles the strip does not read are
    NULL -- and every inconsistency the sharded blend would turn into a wild device read is refused on the host."""
    geo = dp.workload_geometry("200MP")
    plan = dp.make_exchange_plan(geo, world)
    n = len(geo.rects)
    strides = [w * geo.cn for (_, _, w, _) in geo.rects]
    for rank in (0, world - 1):
        owned = [(0x10000000 + t * 0x4000000) if plan.owners[t] == rank else 0 for t in range(n)]
        need = plan.need[rank]
        recv = [(0x7000000000 + t * 0x4000000) if (plan.owners[t] != rank and need[t][0] < need[t][1]) else 0 for t in range(n)]
        base = _native.sharded_tile_bases(geo.rects, geo.cn, world, rank, plan.need, plan.owners, owned, strides, recv)
        for t in range(n):
            a, b = need[t]
            if a >= b:
                assert base[t] == 0
            elif plan.owners[t] == rank:
                assert base[t] == owned[t]
            else:
                assert base[t] == recv[t] - a * strides[t]
        got = [t for t in range(n) if plan.owners[t] != rank and need[t][0] < need[t][1]]
        assert got, "the 200 MP plan has remote tiles for every rank"
        t = got[0]
        for bad_stride in (0, strides[t] + 64, -strides[t]):       # stale / padded / nonsense stride of a RECEIVED tile
            bad = list(strides)
            bad[t] = bad_stride
            with pytest.raises(_native.SrShapeError):
                _native.sharded_tile_bases(geo.rects, geo.cn, world, rank, plan.need, plan.owners, owned, bad, recv)
        no_buf = list(recv)
        no_buf[t] = 0
        with pytest.raises(ValueError):
            _native.sharded_tile_bases(geo.rects, geo.cn, world, rank, plan.need, plan.owners, owned, strides, no_buf)
        owners = list(plan.owners)
        owners[t] = world                                          # owner outside the communicator
        with pytest.raises(ValueError):
            _native.sharded_tile_bases(geo.rects, geo.cn, world, rank, plan.need, owners, owned, strides, recv)
        need_bad = [list(r) for r in plan.need]
        need_bad[rank][t] = (need[t][0], geo.rects[t][3] + 1)      # rows beyond the tile
        with pytest.raises(ValueError):
            _native.sharded_tile_bases(geo.rects, geo.cn, world, rank, need_bad, plan.owners, owned, strides, recv)
        mine = [t for t in range(n) if plan.owners[t] == rank and need[t][0] < need[t][1]]
        if mine:
            short = list(strides)
            short[mine[0]] -= 1                                    # an owned tile whose rows overlap
            with pytest.raises(ValueError):
                _native.sharded_tile_bases(geo.rects, geo.cn, world, rank, plan.need, plan.owners, owned, short, recv)

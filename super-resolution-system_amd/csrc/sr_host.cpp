// sr_host.cpp -- host-only part of the C ABI: error strings, the integer tile bookkeeping of
// tiling_module.py / main.py, the weight LUT and the strip-window planner.  No HIP calls.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <utility>
#include <vector>

#include "sr_internal.h"

static thread_local char g_err[512] = "";

int sr_set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" {

int sr_version(void) { return 200; }

#ifndef SR_SOURCE_DIGEST
#define SR_SOURCE_DIGEST "unknown"
#endif
const char *sr_source_digest(void) { return SR_SOURCE_DIGEST; }

const char *sr_last_error(void) { return g_err; }

// tiling_module.py:572-608.  np.ceil((dim - ov) / step) on a float64 quotient.
int sr_tile_plan(int image_w, int image_h, int block_size, int overlap_px, int *n_tiles, int *h_xywh,
                 int cap)
{
    if (!n_tiles) return sr_set_error(SR_ERR_INVALID_ARG, "sr_tile_plan: n_tiles is null");
    if (image_w <= 0 || image_h <= 0 || block_size <= 0 || overlap_px < 0 || overlap_px >= block_size)
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_tile_plan: bad geometry %dx%d block %d overlap %d",
                            image_w, image_h, block_size, overlap_px);
    const int step = block_size - overlap_px;
    const int nx = std::max(1, (int)std::ceil((double)(image_w - overlap_px) / (double)step));
    const int ny = std::max(1, (int)std::ceil((double)(image_h - overlap_px) / (double)step));
    *n_tiles = nx * ny;
    if (!h_xywh) return SR_OK;
    if (cap < nx * ny) return sr_set_error(SR_ERR_SHAPE, "sr_tile_plan: need room for %d tiles", nx * ny);
    int k = 0;
    for (int ty = 0; ty < ny; ++ty)
        for (int tx = 0; tx < nx; ++tx) {
            const int x = tx * step, y = ty * step;
            h_xywh[k++] = x;
            h_xywh[k++] = y;
            h_xywh[k++] = std::min(block_size, image_w - x);
            h_xywh[k++] = std::min(block_size, image_h - y);
        }
    return SR_OK;
}

// tiling_module.py:610-646, including the last row / column override.
int sr_tile_overlaps(int x, int y, int w, int h, int image_w, int image_h, int block_size, int overlap_px,
                     int *h_tblr)
{
    if (!h_tblr) return sr_set_error(SR_ERR_INVALID_ARG, "sr_tile_overlaps: output is null");
    int top = y > 0 ? overlap_px : 0;
    int left = x > 0 ? overlap_px : 0;
    int bottom = (y + h < image_h) ? overlap_px : 0;
    int right = (x + w < image_w) ? overlap_px : 0;
    if (y + block_size >= image_h) bottom = std::max(0, block_size - (image_h - y) - top);
    if (x + block_size >= image_w) right = std::max(0, block_size - (image_w - x) - left);
    h_tblr[0] = top;
    h_tblr[1] = bottom;
    h_tblr[2] = left;
    h_tblr[3] = right;
    return SR_OK;
}

// tiling_module.py:786-823: dict keyed by (global_x, global_y); later duplicates overwrite.
int sr_tile_neighbors(const int *h_xywh, int n, int block_size, int overlap_px, int *h_nbr)
{
    if (!h_xywh || !h_nbr || n < 0) return sr_set_error(SR_ERR_INVALID_ARG, "sr_tile_neighbors: bad args");
    const int step = block_size - overlap_px;
    std::map<std::pair<int, int>, int> index;
    for (int i = 0; i < n; ++i) index[{h_xywh[4 * i], h_xywh[4 * i + 1]}] = i;
    auto find = [&](int x, int y) {
        auto it = index.find({x, y});
        return it == index.end() ? -1 : it->second;
    };
    for (int i = 0; i < n; ++i) {
        const int x = h_xywh[4 * i], y = h_xywh[4 * i + 1];
        h_nbr[4 * i + 0] = find(x, y - step);
        h_nbr[4 * i + 1] = find(x, y + step);
        h_nbr[4 * i + 2] = find(x - step, y);
        h_nbr[4 * i + 3] = find(x + step, y);
    }
    return SR_OK;
}

// main.py:168-184 (float aspect compare, int() truncation)
int sr_target_size(int width, int height, int preset_mp, int *out_w, int *out_h)
{
    if (width <= 0 || height <= 0 || !out_w || !out_h)
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_target_size: bad args");
    int tw, th;
    switch (preset_mp) {
    case 100: tw = 12245; th = 8163; break;
    case 150: tw = 15000; th = 10000; break;
    case 200: tw = 17320; th = 11547; break;
    default: return sr_set_error(SR_ERR_INVALID_ARG, "sr_target_size: unknown preset %dMP", preset_mp);
    }
    const double aspect = (double)width / (double)height;
    if (aspect > (double)tw / (double)th) th = (int)((double)tw / aspect);
    else tw = (int)((double)th * aspect);
    *out_w = tw;
    *out_h = th;
    return SR_OK;
}

// blending_module.py:547-559 in float64, cast to fp32, tabulated by integer edge distance.
int sr_weight_lut(int fw, int weight_type, float *h_lut)
{
    if (fw < 1 || !h_lut)
        return sr_set_error(SR_ERR_INVALID_ARG,
                            "sr_weight_lut: feather width %d (tile min side < 8 gives the reference a "
                            "zero feather width and NaN weights)", fw);
    for (int d = 0; d <= fw; ++d) {
        double nd = (double)d / (double)fw;
        nd = std::min(1.0, std::max(0.0, nd));
        double v;
        if (weight_type == SR_W_ONES) v = 1.0;
        else if (weight_type == SR_W_COSINE) v = 0.5 * (1 - std::cos(M_PI * nd));
        else if (weight_type == SR_W_SIGMOID) v = 1 / (1 + std::exp(-10 * (nd - 0.5)));
        else v = nd;
        h_lut[d] = (float)v;
    }
    return SR_OK;
}

int sr_strip_tile_rows(const sr_tile_rect *h_tiles, int n, int levels, int canvas_h, int row_begin, int row_end,
                       int *h_rows)
{
    if (!h_tiles || !h_rows || n < 0 || levels < 1 || canvas_h < 1)
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_strip_tile_rows: bad arguments");
    row_begin = std::max(row_begin, 0);
    row_end = std::min(row_end, canvas_h);
    for (int t = 0; t < n; ++t) {
        const sr_tile_rect &r = h_tiles[t];
        if (r.w < 1 || r.h < 1 || r.x < 0 || r.y < 0)
            return sr_set_error(SR_ERR_INVALID_ARG, "sr_strip_tile_rows: tile %d has a bad rectangle", t);
        SrTileLevels lv;
        sr_plan_windows(r.h, r.w, r.y, std::min(levels, SR_MAX_LEVELS), row_begin, row_end, canvas_h, &lv);
        h_rows[2 * t] = lv.gw[0].a;
        h_rows[2 * t + 1] = lv.gw[0].b;
    }
    return SR_OK;
}

// Worst-case halo of sr_plan_windows, derived from its two recurrences (levels whose height is >= 8; smaller levels
// are taken whole).  Up chain, level i-1 -> i: rows [a, b] of R_{i-1} read rows floor(a/2) - 1 .. floor(b/2) + 1 of R_i;
// in level-0 units that is at most 1.5 * 2^i rows further below and 2^i further above.  Down chain, level i+1 -> i:
// rows [a, b] of G_{i+1} read rows 2a - 2 .. 2b + 2 of G_i: 2 * 2^i further on both sides.  Summed over a pyramid of
// L levels: below = 1.5 (2^L - 2) + 2 (2^(L-1) - 1) = 2.5 * 2^L - 5, above = (2^L - 2) + (2^L - 2) + 1 (half-open end).
// L = 6: 155 rows below, 125 above (the survey's "~128" counted the down chain only).
int sr_pyramid_halo(int levels, int *below, int *above)
{
    if (levels < 1 || levels > SR_MAX_LEVELS || !below || !above)
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_pyramid_halo: bad arguments");
    const long long p = 1LL << levels;
    *below = (int)((5 * p) / 2 - 5);
    *above = (int)(2 * p - 3);
    if (levels == 1) *below = *above = 0;
    return SR_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// Multi-GPU planning (host only, identical on every rank): the canvas is cut into `world` horizontal
// strips of equal WORK, every tile gets an owner, and sr_exchange_plan lists, per rank, the tile rows it must hold --
// what an owner sends and a strip owner receives.  The transfers themselves belong to the host (RCCL ncclSend /
// ncclRecv in one group, MPI, hipMemcpyPeer ...); device_pipeline.py posts them through torch.distributed.
// ---------------------------------------------------------------------------------------------------------------
// Relative cost of the stages per pixel, from the per-kernel times at 200 MP on MI355X (ps per pixel): what a canvas
// row costs its strip owner.  Only the ratios matter.
static const double COST_ASSESS = 10.5;    // per canvas pixel (fused PSNR + 3 x SSIM)
static const double COST_GATHER = 4.0;     // per tile pixel visited by the canvas gather
static const double COST_PYRAMID = 6.6;    // per tile pixel of the pyramid chains -- also paid for the halo rows

int sr_strip_bounds(const sr_tile_rect *h_tiles, int n, int levels, int canvas_h, int canvas_w, int world, int *h_bounds)
{
    if (!h_bounds || world < 1 || canvas_h < 1 || canvas_w < 1 || n < 0 || (n > 0 && !h_tiles) || levels < 1)
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_strip_bounds: bad arguments");
    if (world == 1 || n == 0) {
        for (int r = 0; r <= world; ++r) h_bounds[r] = (int)((long long)canvas_h * r / world);
        return SR_OK;
    }
    int below = 0, above = 0;
    sr_pyramid_halo(std::min(levels, SR_MAX_LEVELS), &below, &above);
    const int halo = std::max(below, above);
    std::vector<double> cover((size_t)canvas_h, 0.0), cum_can((size_t)canvas_h + 1, 0.0), cum_pyr((size_t)canvas_h + 1, 0.0);
    for (int t = 0; t < n; ++t) {
        const sr_tile_rect &r = h_tiles[t];
        for (int y = std::max(r.y, 0); y < std::min(r.y + r.h, canvas_h); ++y) cover[(size_t)y] += r.w;
    }
    for (int y = 0; y < canvas_h; ++y) {
        cum_can[(size_t)y + 1] = cum_can[(size_t)y] + (COST_ASSESS * canvas_w + COST_GATHER * cover[(size_t)y]);
        cum_pyr[(size_t)y + 1] = cum_pyr[(size_t)y] + COST_PYRAMID * cover[(size_t)y];
    }
    // a strip [a, b) costs the assessment and gather work of its own rows plus the pyramid work of rows a - halo ..
    // b + halo (clipped: the outer strips recompute a halo on one side only)
    auto cost = [&](int a, int b) {
        const int lo = std::max(a - halo, 0), hi = std::min(b + halo, canvas_h);
        return cum_can[(size_t)b] - cum_can[(size_t)a] + cum_pyr[(size_t)hi] - cum_pyr[(size_t)lo];
    };
    auto place = [&](double target, std::vector<int> &bounds) {
        bounds.assign(1, 0);
        for (int k = 0; k + 1 < world; ++k) {
            const int a = bounds.back();
            int lo = a + 2, hi = canvas_h;                     // smallest even b with cost(a, b) >= target
            while (lo < hi) {
                const int mid = (lo + hi) / 2;
                if (cost(a, std::min(mid, canvas_h)) >= target) hi = mid;
                else lo = mid + 1;
            }
            const int b = std::min(lo + (lo % 2), canvas_h);
            bounds.push_back(std::max(b, std::min(a + 2, canvas_h)));
        }
        bounds.push_back(canvas_h);
    };
    double t_lo = 0.0, t_hi = cost(0, canvas_h);
    std::vector<int> bnd;
    for (int it = 0; it < 60; ++it) {                          // bisection on the per-strip cost: the last strip absorbs the rest
        const double t = 0.5 * (t_lo + t_hi);
        place(t, bnd);
        if (cost(std::min(bnd[(size_t)world - 1], canvas_h), canvas_h) > t) t_lo = t;
        else t_hi = t;
    }
    place(t_hi, bnd);
    for (int r = 1; r <= world; ++r) bnd[(size_t)r] = std::min(std::max(bnd[(size_t)r], bnd[(size_t)r - 1]), canvas_h);
    for (int r = 0; r <= world; ++r) h_bounds[r] = bnd[(size_t)r];
    return SR_OK;
}

int sr_exchange_plan(const sr_tile_rect *h_tiles, int n, int cn, int levels, int canvas_h, int canvas_w, int world,
                     int metric_halo, int owner_policy, int *h_bounds, int *h_rows, int *h_need, int *h_owner)
{
    if (!h_tiles || !h_bounds || !h_rows || !h_need || !h_owner || n < 1 || world < 1 || cn < 1 || levels < 1 || metric_halo < 0 ||
        owner_policy < 0 || owner_policy > 2)
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_exchange_plan: bad arguments");
    int rc = sr_strip_bounds(h_tiles, n, levels, canvas_h, canvas_w, world, h_bounds);
    if (rc) return rc;
    for (int r = 0; r < world; ++r) {
        const int a = std::max(h_bounds[r] - (world > 1 ? metric_halo : 0), 0);
        const int b = std::min(h_bounds[r + 1] + (world > 1 ? metric_halo : 0), canvas_h);
        h_rows[2 * r] = a;
        h_rows[2 * r + 1] = b;
        rc = sr_strip_tile_rows(h_tiles, n, levels, canvas_h, a, b, h_need + (size_t)r * n * 2);
        if (rc) return rc;
    }
    if (owner_policy == SR_OWNER_ROUNDROBIN) {
        for (int t = 0; t < n; ++t) h_owner[t] = t % world;
        return SR_OK;
    }
    if (owner_policy == SR_OWNER_LOCALITY) {                   // the rank whose strip holds the tile's centre row
        for (int t = 0; t < n; ++t) {
            const int c = std::min(h_tiles[t].y + h_tiles[t].h / 2, h_bounds[world] - 1);
            int o = 0;
            while (o + 1 < world && h_bounds[o + 1] <= c) ++o;
            h_owner[t] = o;
        }
        return SR_OK;
    }
    // SR_OWNER_BALANCED: greedy, tile by tile, the owner that minimises the busiest rank-to-rank link after the assignment,
    // then the bytes added, then the owner's tile count (xGMI is point-to-point: the heaviest pair bounds the exchange)
    std::vector<long long> link((size_t)world * world, 0);
    std::vector<int> owned((size_t)world, 0);
    long long link_max = 0;
    for (int t = 0; t < n; ++t) {
        std::vector<long long> nbytes((size_t)world);
        for (int r = 0; r < world; ++r) {
            const int *nd = h_need + ((size_t)r * n + t) * 2;
            nbytes[(size_t)r] = (long long)std::max(nd[1] - nd[0], 0) * h_tiles[t].w * cn;
        }
        int best_o = -1;
        long long best_worst = 0, best_sum = 0;
        for (int o = 0; o < world; ++o) {
            long long worst = link_max, sum = 0;
            for (int r = 0; r < world; ++r) {
                const long long add = r == o ? 0 : nbytes[(size_t)r];
                worst = std::max(worst, link[(size_t)o * world + r] + add);
                sum += add;
            }
            const bool better = best_o < 0 || worst < best_worst || (worst == best_worst && (sum < best_sum ||
                                (sum == best_sum && owned[(size_t)o] < owned[(size_t)best_o])));
            if (better) { best_o = o; best_worst = worst; best_sum = sum; }
        }
        for (int r = 0; r < world; ++r)
            if (r != best_o) {
                link[(size_t)best_o * world + r] += nbytes[(size_t)r];
                link_max = std::max(link_max, link[(size_t)best_o * world + r]);
            }
        owned[(size_t)best_o] += 1;
        h_owner[t] = best_o;
    }
    return SR_OK;
}

// The batch of one rank's exchange as plain (peer, pointer, bytes) lists: what sr_comm_exchange_tile_rows posts to RCCL,
// for hosts that move the rows themselves (host only; nothing is dereferenced).
int sr_exchange_xfers(const sr_tile_rect *h_tiles, int n, int cn, int world, int me, const int *h_need, const int *h_owner,
                      const void *const *d_owned, const int64_t *owned_stride, void *const *d_recv, sr_xfer *sends, int cap_send,
                      int *n_send, sr_xfer *recvs, int cap_recv, int *n_recv)
{
    if (!h_tiles || !h_need || !h_owner || !d_owned || !owned_stride || !d_recv || !n_send || !n_recv || n < 1 || cn < 1 || world < 1 ||
        me < 0 || me >= world || cap_send < 0 || cap_recv < 0 || (cap_send && !sends) || (cap_recv && !recvs))
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_exchange_xfers: bad arguments");
    for (int t = 0; t < n; ++t)
        if (h_owner[t] < 0 || h_owner[t] >= world)
            return sr_set_error(SR_ERR_INVALID_ARG, "sr_exchange_xfers: tile %d owner %d of %d ranks", t, h_owner[t], world);
    int ns = 0, nr = 0;
    for (int r = 0; r < world; ++r) {                           // sends: reader-major, then tile
        if (r == me) continue;
        for (int t = 0; t < n; ++t) {
            const int a = h_need[((size_t)r * n + t) * 2], b = h_need[((size_t)r * n + t) * 2 + 1];
            if (a >= b || h_owner[t] != me) continue;
            const int64_t row = (int64_t)h_tiles[t].w * cn;     // u8 tiles (the SR output)
            if (!d_owned[t] || a < 0 || b > h_tiles[t].h)
                return sr_set_error(SR_ERR_INVALID_ARG, "sr_exchange_xfers: owned tile %d: pointer %p, rows [%d, %d) of %d", t, d_owned[t], a,
                                    b, h_tiles[t].h);
            if (owned_stride[t] != row)                         // the receiver posts ONE dense transfer per tile
                return sr_set_error(SR_ERR_UNSUPPORTED, "sr_exchange_xfers: tile %d is sent to rank %d and must be dense (stride %lld != "
                                    "%d * %d)", t, r, (long long)owned_stride[t], h_tiles[t].w, cn);
            if (ns < cap_send) sends[ns] = {r, (void *)((const char *)d_owned[t] + (int64_t)a * row), (uint64_t)((int64_t)(b - a) * row)};
            ++ns;
        }
    }
    for (int o = 0; o < world; ++o) {                           // receives: owner-major, then tile
        if (o == me) continue;
        for (int t = 0; t < n; ++t) {
            const int a = h_need[((size_t)me * n + t) * 2], b = h_need[((size_t)me * n + t) * 2 + 1];
            if (a >= b || h_owner[t] != o) continue;
            if (!d_recv[t])
                return sr_set_error(SR_ERR_INVALID_ARG, "sr_exchange_xfers: no receive buffer for rows [%d, %d) of tile %d", a, b, t);
            if (nr < cap_recv) recvs[nr] = {o, d_recv[t], (uint64_t)((int64_t)(b - a) * h_tiles[t].w * cn)};
            ++nr;
        }
    }
    *n_send = ns;
    *n_recv = nr;
    if (ns > cap_send || nr > cap_recv)
        return sr_set_error(SR_ERR_SHAPE, "sr_exchange_xfers: %d sends / %d receives, room for %d / %d", ns, nr, cap_send, cap_recv);
    return SR_OK;
}

// Tile base pointers of one rank as the strip blend wants them (host only; nothing is dereferenced): an owned tile as it
// is, a received row window moved up to its (virtual) row 0.  A receive buffer is dense -- w * cn bytes per row is what
// sr_exchange_xfers posts -- so that is the only stride a non-owned tile may carry.
int sr_sharded_tile_bases(const sr_tile_rect *h_tiles, int n, int cn, int world, int me, const int *h_need, const int *h_owner,
                          const void *const *d_owned, const int64_t *strides, void *const *d_recv, void **h_base)
{
    if (!h_tiles || !h_need || !h_owner || !d_owned || !strides || !d_recv || !h_base || n < 1 || cn < 1 || world < 1 || me < 0 ||
        me >= world)
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_sharded_tile_bases: bad arguments");
    for (int t = 0; t < n; ++t) {
        const int r0 = h_need[((size_t)me * n + t) * 2], r1 = h_need[((size_t)me * n + t) * 2 + 1];
        h_base[t] = nullptr;
        if (h_owner[t] < 0 || h_owner[t] >= world)
            return sr_set_error(SR_ERR_INVALID_ARG, "sr_sharded_tile_bases: tile %d owner %d of %d ranks", t, h_owner[t], world);
        if (r0 >= r1) continue;                                  // this strip does not read the tile
        if (r0 < 0 || r1 > h_tiles[t].h)
            return sr_set_error(SR_ERR_INVALID_ARG, "sr_sharded_tile_bases: tile %d rows [%d, %d) of %d", t, r0, r1, h_tiles[t].h);
        const int64_t dense = (int64_t)h_tiles[t].w * cn;
        if (h_owner[t] == me) {
            if (!d_owned[t] || strides[t] < dense)
                return sr_set_error(SR_ERR_INVALID_ARG, "sr_sharded_tile_bases: owned tile %d: pointer %p, stride %lld < %lld", t, d_owned[t],
                                    (long long)strides[t], (long long)dense);
            h_base[t] = (void *)d_owned[t];
        } else {
            if (strides[t] != dense)
                return sr_set_error(SR_ERR_SHAPE, "sr_sharded_tile_bases: tile %d is received into a dense buffer (%lld bytes per row) but "
                                                  "strides[%d] = %lld", t, (long long)dense, t, (long long)strides[t]);
            if (!d_recv[t])
                return sr_set_error(SR_ERR_INVALID_ARG, "sr_sharded_tile_bases: no receive buffer for rows [%d, %d) of tile %d", r0, r1, t);
            h_base[t] = (char *)d_recv[t] - (int64_t)r0 * dense;
        }
    }
    return SR_OK;
}

double sr_psnr_from_sse(uint64_t sse, uint64_t count, double data_range)
{
    if (count == 0) return NAN;
    const double mse = (double)sse / (double)count;
    if (mse == 0.0) return INFINITY;
    return 10.0 * std::log10((data_range * data_range) / mse);
}

}  // extern "C"

// build_gaussian_pyramid's stop rule (blending_module.py:248-252)
void sr_level_dims(int h, int w, int levels, int *nl, int *H, int *W)
{
    int k = 1;
    H[0] = h;
    W[0] = w;
    while (k < levels && k < SR_MAX_LEVELS && H[k - 1] >= 2 && W[k - 1] >= 2) {
        H[k] = (H[k - 1] + 1) / 2;
        W[k] = (W[k - 1] + 1) / 2;
        ++k;
    }
    *nl = k;
}

static SrWin hull(SrWin p, SrWin q)
{
    if (p.empty()) return q;
    if (q.empty()) return p;
    return {std::min(p.a, q.a), std::max(p.b, q.b)};
}

// destination rows d of a pyrUp -> source rows it reads (low edge reflect-101, high edge clamp)
static SrWin need_up(SrWin d, int hs)
{
    if (d.empty()) return {0, 0};
    if (hs < 8) return {0, hs};
    int lo = d.a / 2 - 1, hi = (d.b - 1) / 2 + 1;
    if (lo < 0) {
        lo = 0;
        hi = std::max(hi, 1);
    }
    hi = std::min(hi, hs - 1);
    return {lo, hi + 1};
}

// destination rows d of a pyrDown -> source rows it reads (reflect-101 both ends)
static SrWin need_down(SrWin d, int hs)
{
    if (d.empty()) return {0, 0};
    if (hs < 8) return {0, hs};
    int lo = 2 * d.a - 2, hi = 2 * (d.b - 1) + 2;
    if (lo < 0) {
        lo = 0;
        hi = std::max(hi, 2);
    }
    if (hi > hs - 1) {
        lo = std::min(lo, hs - 3);
        hi = hs - 1;
    }
    return {std::max(lo, 0), hi + 1};
}

void sr_plan_windows(int tile_h, int tile_w, int tile_y, int levels, int row_begin, int row_end,
                     int canvas_h, SrTileLevels *out)
{
    SrTileLevels &L = *out;
    sr_level_dims(tile_h, tile_w, levels, &L.nl, L.H, L.W);
    const int lo = std::max(std::max(row_begin, 0) - tile_y, 0);
    const int hi = std::min(std::min(row_end, canvas_h) - tile_y, tile_h);
    L.cw = {lo, hi};
    for (int i = 0; i < SR_MAX_LEVELS; ++i) L.gw[i] = L.rw[i] = {0, 0};
    if (L.cw.empty()) {
        L.cw = {0, 0};
        return;
    }
    L.rw[0] = L.cw;
    for (int i = 1; i < L.nl; ++i) L.rw[i] = need_up(L.rw[i - 1], L.H[i]);
    L.gw[L.nl - 1] = L.rw[L.nl - 1];
    for (int i = L.nl - 2; i >= 0; --i) L.gw[i] = hull(L.rw[i], need_down(L.gw[i + 1], L.H[i]));
}

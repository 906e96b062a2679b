// sr_engine.hip -- gfx950 (CDNA4) kernels and the device side of the C ABI in include/sr_hip.h.
//
// Numerics contract: every fp32 expression is evaluated in the order written in
// oracle/sr_oracle.c (build with -ffp-contract=off), so the blend agrees with the CPU
// restatement bit for bit.  fp64 is used only in the SSIM kernels.
//
// Data layout in HBM
//   * external images / tiles / canvas: row-major HWC, u8 (or fp32 tiles), byte strides --
//     exactly the reference's ndarrays.
//   * internal pyramid levels i >= 1 of every tile live in one arena: planar fp32, plane
//     c of level i at  off[i] + c * H_i * P_i,  row pitch P_i = round_up(W_i, 32) floats (whole 128-byte lines).
//     G_i = Gaussian level, R_i = collapsed weighted-Laplacian level, W_i = weight level
//     (one per distinct tile shape).
//   * the fp32 canvas accumulators of the reference are never materialised: the final kernel
//     is canvas-centric (gather) and sums the covering tiles in list order in registers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <type_traits>
#include <vector>

#include "sr_ctx.h"

// Wave priority of the chain's kernels beside the assessment (A/B builds: -DSR_CHAIN_PRIO=1..3 for the small launches --
// pyramid levels 2..5 down and up, the level-2 border columns -- -DSR_CHAIN_PRIO_BIG for tile extract and the level-1 + 2 march).
// Measured and rejected (profiles/r04_e_chain_prio.txt): the step is 3.15-3.23 ms at every setting -- the small launches are
// slow beside the assessment because ONE of their waves fits a SIMD next to its three (104 free registers), not because
// they lose the issue arbitration.
#ifdef SR_CHAIN_PRIO
#define SR_CHAIN_SETPRIO() __builtin_amdgcn_s_setprio(SR_CHAIN_PRIO)
#else
#define SR_CHAIN_SETPRIO() ((void)0)
#endif
#ifdef SR_CHAIN_PRIO_BIG
#define SR_CHAIN_SETPRIO_BIG() __builtin_amdgcn_s_setprio(SR_CHAIN_PRIO_BIG)
#else
#define SR_CHAIN_SETPRIO_BIG() ((void)0)
#endif

// ---------------------------------------------------------------------------------------------
// context (struct, Guard, ProfScope: sr_ctx.h)
// ---------------------------------------------------------------------------------------------
hipError_t upload_small(sr_ctx *c, void *d_dst, const void *h_src, size_t bytes)
{
    // bound what a caller that never synchronises can pile up: drain the stream once 32 MB of table copies are parked
    c->pending_bytes += bytes;
    if (c->pending_bytes > ((size_t)32 << 20)) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return e;
        c->pending_host.clear();
        c->pending_bytes = bytes;
    }
    c->pending_host.emplace_back((const char *)h_src, (const char *)h_src + bytes);
    return hipMemcpyAsync(d_dst, c->pending_host.back().data(), bytes, hipMemcpyHostToDevice, c->stream);
}

hipError_t upload_if_changed(sr_ctx *c, void *d_dst, const void *h_src, size_t bytes, std::vector<char> &shadow)
{
    if (shadow.size() == bytes && bytes > 0 && memcmp(shadow.data(), h_src, bytes) == 0) return hipSuccess;
    shadow.assign((const char *)h_src, (const char *)h_src + bytes);
    return upload_small(c, d_dst, h_src, bytes);
}

hipError_t upload_cached(sr_ctx *c, CachedTable &t, const void *h_src, size_t bytes)
{
    if (bytes > t.cap) {
        if (t.d) {
            hipError_t e = stream_sync(c);
            if (e != hipSuccess) return e;
            (void)hipFree(t.d);
            t.d = nullptr;
        }
        const size_t nb = std::max<size_t>((bytes + 4095) / 4096 * 4096, 4096);
        hipError_t e = hipMalloc(&t.d, nb);
        if (e != hipSuccess) { t.cap = 0; return e; }
        t.cap = nb;
        t.shadow.clear();
    }
    return upload_if_changed(c, t.d, h_src, bytes, t.shadow);
}

hipError_t stream_sync(sr_ctx *c)
{
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) {
        c->pending_host.clear();
        c->pending_bytes = 0;
    }
    return e;
}

// Live-handle registry: destroying a context destroys its plans; destroying (or using) a handle that is
// no longer live is a harmless no-op / SR_ERR_INVALID_ARG instead of a use-after-free (host languages with
// garbage collectors finalise objects in arbitrary order at shutdown).
struct sr_blend_plan;
static std::mutex g_reg_mu;
static std::set<const void *> g_live_ctx, g_live_plan;
bool ctx_is_live(const sr_ctx *c)
{
    std::lock_guard<std::mutex> lk(g_reg_mu);
    return c && g_live_ctx.count(c) != 0;
}
static bool plan_is_live(const sr_blend_plan *p)
{
    std::lock_guard<std::mutex> lk(g_reg_mu);
    return p && g_live_plan.count(p) != 0;
}

int ctx_scratch(sr_ctx *c, size_t bytes, void **out)
{
    if (bytes > c->scratch_bytes) {
        if (c->scratch) {
            HIPCHK(stream_sync(c));
            HIPCHK(hipFree(c->scratch));
            c->scratch = nullptr;
            c->scratch_bytes = 0;
        }
        size_t nb = std::max(bytes, (size_t)1 << 20);
        HIPCHK(hipMalloc(&c->scratch, nb));
        c->scratch_bytes = nb;
    }
    *out = c->scratch;
    return SR_OK;
}

hipEvent_t prof_event(sr_ctx *c)
{
    if (!c->ev_pool.empty()) {
        hipEvent_t e = c->ev_pool.back();
        c->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return sr_set_error(SR_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return SR_OK;
}

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
enum { PAD_MIRROR = 0, PAD_REPLICATE = 1, PAD_REFLECT = 2, PAD_CONSTANT = 3 };

__device__ __forceinline__ int border_index(int p, int n, int mode)
{
    if (p >= 0 && p < n) return p;
    if (mode == PAD_REPLICATE) return p < 0 ? 0 : n - 1;
    if (n == 1) return 0;
    const int delta = (mode == PAD_MIRROR) ? 1 : 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p - 1 + delta;
        else p = n - 1 - (p - n) - delta;
    }
    return p;
}

__device__ __forceinline__ int reflect101(int p, int n) { return border_index(p, n, PAD_MIRROR); }

// One axis of cv2.pyrUp, unnormalised (x8): value of destination column x from source row `row`.
__device__ __forceinline__ float up_h(const float *__restrict__ row, int ws, int x)
{
    const int sx = x >> 1;
    if (ws == 1) return (x & 1) ? row[0] * 8.0f : row[0] * 6.0f + row[0] * 2.0f;
    if (!(x & 1)) {
        if (sx == 0) return row[0] * 6.0f + row[1] * 2.0f;
        if (sx == ws - 1) return row[sx - 1] + row[sx] * 7.0f;
        return (row[sx - 1] + row[sx] * 6.0f) + row[sx + 1];
    }
    if (sx == ws - 1) return row[sx] * 8.0f;
    return (row[sx] + row[sx + 1]) * 4.0f;
}

// cv2.pyrUp sample at destination (y, x) from one planar fp32 source plane (hs x ws, pitch).
__device__ __forceinline__ float up_sample(const float *__restrict__ src, int hs, int ws, int pitch, int y,
                                           int x)
{
    const int sy = y >> 1;
    const int yp = min(sy + 1, hs - 1);
    const float r1 = up_h(src + (size_t)sy * pitch, ws, x);
    const float r2 = up_h(src + (size_t)yp * pitch, ws, x);
    if (!(y & 1)) {
        const int ym = (sy - 1 < 0) ? (hs > 1 ? 1 : 0) : sy - 1;
        const float r0 = up_h(src + (size_t)ym * pitch, ws, x);
        return ((r0 + r1 * 6.0f) + r2) * (1.0f / 64.0f);
    }
    return ((r1 + r2) * 4.0f) * (1.0f / 64.0f);
}

// ---------------------------------------------------------------------------------------------
// blend plan tables
// ---------------------------------------------------------------------------------------------
struct TileDev {
    int h, w, x, y;
    int nl, fw, lut_off, cls;
    int H[SR_MAX_LEVELS], W[SR_MAX_LEVELS], P[SR_MAX_LEVELS];
    long long g_off[SR_MAX_LEVELS], r_off[SR_MAX_LEVELS], w_off[SR_MAX_LEVELS];
    int g0[SR_MAX_LEVELS], g1[SR_MAX_LEVELS];  // G row windows
    int r0[SR_MAX_LEVELS], r1[SR_MAX_LEVELS];  // R row windows
};

static_assert(sizeof(TileDev) % 16 == 0, "k_final_blk reads the leading {h,w,x,y} as one int4");

struct TileSrc {
    const void *p;
    long long stride;
};

// What the final gather needs of one tile, 80 bytes (wave-uniform: read through the scalar cache).
struct FinalDesc {
    int x, y, w, h;
    int fw, lut_off, nl, pad0;
    int H1, W1, P1, pad1;
    long long g1, r1;      // float offsets of plane 0 of G_1 / R_1 in the arena
    const void *src;       // level-0 tile data (row 0, possibly virtual) and its row stride in bytes
    long long stride;
    int H2, W2, P2, pad2;  // level 2 (the fused gather builds R_1 from it on the fly)
    long long g2, r2;      // float offsets of plane 0 of G_2 / R_2
    long long w1;          // float offset of the weight level 1 of the tile's class
    long long pad3;
};
static_assert(sizeof(FinalDesc) == 128, "FinalDesc layout");

enum { SRC_U8 = 0, SRC_F32 = 1, SRC_PLANAR = 2, SRC_LUT = 3 };

// Source accessors for the pyrDown kernel -----------------------------------------------------
// Vector load / store helpers.  The *_aN typedefs carry a reduced alignment so the compiler may emit one wide
// global_load for an address that is only float- (or byte-) aligned; gfx950 handles those in hardware.
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef float f2_t __attribute__((ext_vector_type(2)));
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef unsigned u3_t __attribute__((ext_vector_type(3)));
typedef unsigned u2_t __attribute__((ext_vector_type(2)));
typedef f4_t f4_a4_t __attribute__((aligned(4)));    // 4 floats at any float-aligned address
typedef f2_t f2_a8_t __attribute__((aligned(8)));
typedef u3_t u3_a1_t __attribute__((aligned(1)));    // 12 bytes at any address
typedef u4_t u4_a1_t __attribute__((aligned(1)));    // 16 bytes at any address
typedef u4_t u4_a4_t __attribute__((aligned(4)));
typedef unsigned u1_a1_t __attribute__((aligned(1)));

__device__ __forceinline__ f4_t ld_f4_a4(const float *p) { return *(const f4_a4_t *)p; }
__device__ __forceinline__ f4_t ld_f4(const float *p) { return *(const f4_t *)p; }
__device__ __forceinline__ void st_f4(float *p, f4_t v) { *(f4_t *)p = v; }
__device__ __forceinline__ u3_t ld_u3_a1(const void *p) { return *(const u3_a1_t *)p; }
// Same through a global-address-space pointer: for addresses that come out of a descriptor table in memory (tile
// pointers), where the compiler would otherwise emit flat_load (both wait counters, aperture check).
__device__ __forceinline__ u3_t ld_u3_a1_g(const void *p)
{
    return *(const __attribute__((address_space(1))) u3_a1_t *)p;
}

// One output pixel (all planes) of level lvl+1 from level lvl -- the generic form with every border rule.
template <int SRC>
__device__ __forceinline__ void down_pixel(const TileDev &T, const TileSrc S, int lvl, int cn, int x, int y,
                                           float *__restrict__ arena, const float *__restrict__ luts, int c_only = -1)
{
    const int hs = T.H[lvl], ws = T.W[lvl];
    int xi[5], yi[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        xi[k] = reflect101(2 * x + k - 2, ws);
        yi[k] = reflect101(2 * y + k - 2, hs);
    }
    const int po = T.P[lvl + 1];
    float *dst = arena + T.g_off[lvl + 1] + (size_t)y * po + x;
    const size_t dplane = (size_t)T.H[lvl + 1] * po;
    // c_only >= 0: that plane alone (a thread per plane: its 25 loads are one round trip, no store of another plane between them)
    for (int c = c_only >= 0 ? c_only : 0; c < (c_only >= 0 ? c_only + 1 : cn); ++c) {
        float rowv[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            float s[5];
            if (SRC == SRC_U8) {
                const unsigned char *r = (const unsigned char *)S.p + (size_t)yi[k] * S.stride;
#pragma unroll
                for (int j = 0; j < 5; ++j) s[j] = (float)r[xi[j] * cn + c];
            } else if (SRC == SRC_F32) {
                const float *r = (const float *)((const char *)S.p + (size_t)yi[k] * S.stride);
#pragma unroll
                for (int j = 0; j < 5; ++j) s[j] = r[xi[j] * cn + c];
            } else if (SRC == SRC_PLANAR) {
                const float *r = arena + T.g_off[lvl] + (size_t)c * hs * T.P[lvl] + (size_t)yi[k] * T.P[lvl];
#pragma unroll
                for (int j = 0; j < 5; ++j) s[j] = r[xi[j]];
            } else {  // SRC_LUT: analytic weight map, level 0 of a weight class
                const int ry = yi[k];
                const int dy = min(ry, hs - 1 - ry);
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int dx = min(xi[j], ws - 1 - xi[j]);
                    s[j] = luts[T.lut_off + min(min(dy, dx), T.fw)];
                }
            }
            rowv[k] = ((s[2] * 6.0f + (s[1] + s[3]) * 4.0f) + s[0]) + s[4];
        }
        const float v = ((rowv[2] * 6.0f + (rowv[1] + rowv[3]) * 4.0f) + rowv[0]) + rowv[4];
        dst[c * dplane] = v * (1.0f / 256.0f);
    }
}

// level i -> i+1 of every tile (or weight class) in one launch.  One thread = one output pixel,
// all planes.  Rows limited to the G window of the destination level.  (Generic kernel: weight classes,
// fp32 HWC tiles, channel counts other than 1 / 3.)
template <int SRC>
__global__ __launch_bounds__(256) void k_down(const TileDev *__restrict__ tiles, const TileSrc *__restrict__ srcs,
                                              int lvl, int cn, float *__restrict__ arena,
                                              const float *__restrict__ luts)
{
    const TileDev &T = tiles[blockIdx.z];
    if (lvl + 1 >= T.nl) return;
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = T.g0[lvl + 1] + blockIdx.y * 4 + threadIdx.y;
    if (x >= T.W[lvl + 1] || y >= T.g1[lvl + 1]) return;
    TileSrc S;
    S.p = nullptr;
    S.stride = 0;
    if (SRC == SRC_U8 || SRC == SRC_F32) S = srcs[blockIdx.z];
    down_pixel<SRC>(T, S, lvl, cn, x, y, arena, luts);
}

// ---------------------------------------------------------------------------------------------
// Column-marching pyrDown.  One thread owns 4 output columns (a "column group": output x0 = 4 * cg, input columns
// 2 x0 - 2 .. 2 x0 + 8) and walks down seg_rows output rows: every input row is loaded once and its horizontal pass
// evaluated once; the five row-pass results an output row needs (rows 2y-2 .. 2y+2) live in registers.  Row indices
// go through REFLECT_101, so the top / bottom tile borders need no separate path.  Only "interior" column groups
// (whole window inside the row: cg = 1 .. ncg) take the march; the few border columns of a level are done by the
// trailing blocks of the same launch.
// Lanes are laid over (segment, column group) cells flattened per tile, so waves are full except the last one.
// seg_rows (output rows per segment, chosen per launch): longer segments amortise the 3-row prologue, shorter ones
// keep enough cells in flight on the small levels.
// ---------------------------------------------------------------------------------------------

// number of interior column groups of a level: cg = 1 .. ncg
__host__ __device__ __forceinline__ int down_ncg(int ws, int wo)
{
    const int a = ws >= 10 ? (ws - 10) / 8 : 0;     // 2 x0 + 9 <= ws - 1
    const int b = wo >= 4 ? (wo - 4) / 4 : 0;       // x0 + 3 <= wo - 1
    return a < b ? a : b;
}

struct TileDev;
__host__ __device__ __forceinline__ bool down2_takes(const TileDev &T);     // sr_down2.inc

// REFLECT_101 of a row index that leaves [0, n) by at most n - 1 (one bounce, no loop) -- the march's rows do by <= 2
__device__ __forceinline__ int reflect101_once(int p, int n)
{
    if (n == 1) return 0;
    p = p < 0 ? -p : p;
    return p >= n ? 2 * n - 2 - p : p;
}

// horizontal pass of one u8 row for the thread's 4 outputs x CN channels.  All values are integers below 2^24, so
// fp32 evaluates them exactly in any order: bit-identical to ((s2*6 + (s1+s3)*4) + s0) + s4 with two fmas.
template <int CN>
__device__ __forceinline__ void down_row_u8(const unsigned (&wds)[(CN == 3) ? 9 : 3], float (&h)[4 * CN])
{
    float s[11][CN];
#pragma unroll
    for (int b = 0; b < 11 * CN; ++b) s[b / CN][b % CN] = (float)((wds[b >> 2] >> (8 * (b & 3))) & 0xFFu);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < CN; ++c)
            h[k * CN + c] = fmaf(s[2 * k + 2][c], 6.0f, fmaf(s[2 * k + 1][c] + s[2 * k + 3][c], 4.0f, s[2 * k][c] + s[2 * k + 4][c]));
}

template <int CN>
__device__ __forceinline__ void down_load_u8(const unsigned char *__restrict__ base, long long stride, int row, int xb,
                                             unsigned (&wds)[(CN == 3) ? 9 : 3])
{
    // the tile pointer comes out of a table in memory: tell the compiler it is global memory (global_load, not flat_load)
    const unsigned char *p = base + (size_t)row * stride + (size_t)xb * CN;
    if (CN == 3) {
        // Dword-aligned loads + one funnel shift per dword instead of three byte-aligned 12-byte loads (the memory
        // pipeline was the limiter of this kernel: MemUnitStalled 70 %; -6 % run time).  m = byte offset of the window
        // inside its first dword (per lane: lanes of one wave can sit in different row segments); the 33 bytes needed lie
        // inside the nine aligned dwords [p - m, p - m + 36): no byte beyond what the unaligned loads touched is read.
        typedef u3_t u3_a4_t __attribute__((aligned(4)));
        const unsigned m = (unsigned)((size_t)p & 3u);
        const unsigned char *q = p - m;
        const u3_t q0 = *(const __attribute__((address_space(1))) u3_a4_t *)q, q1 = *(const __attribute__((address_space(1))) u3_a4_t *)(q + 12),
                   q2 = *(const __attribute__((address_space(1))) u3_a4_t *)(q + 24);
        const unsigned d[10] = {q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, q2.x, q2.y, q2.z, 0u};
#pragma unroll
        for (int i = 0; i < 9; ++i) wds[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], m);
    } else {
        const u3_t q0 = ld_u3_a1_g(p);
        wds[0] = q0.x; wds[1] = q0.y; wds[2] = q0.z;
    }
}

__device__ __forceinline__ void down_load_f32(const float *__restrict__ plane, int ps, int row, int xb, float (&s)[11])
{
    const float *p = plane + (size_t)row * ps + xb;          // xb = 8 cg - 2: 8-byte aligned, xb + 2 16-byte aligned
    const f2_t a = *(const f2_a8_t *)p;
    const f4_t b = ld_f4(p + 2), d = ld_f4(p + 6);
    s[0] = a.x; s[1] = a.y; s[2] = b.x; s[3] = b.y; s[4] = b.z; s[5] = b.w; s[6] = d.x; s[7] = d.y; s[8] = d.z; s[9] = d.w;
    s[10] = p[10];
}

__device__ __forceinline__ void down_row_f32(const float (&s)[11], float (&h)[4])
{
#pragma unroll
    for (int k = 0; k < 4; ++k) h[k] = ((s[2 * k + 2] * 6.0f + (s[2 * k + 1] + s[2 * k + 3]) * 4.0f) + s[2 * k]) + s[2 * k + 4];
}

template <int SRC, int CN>
__global__ __launch_bounds__(256) void k_down_march(const TileDev *__restrict__ tiles, const TileSrc *__restrict__ srcs,
                                                    int lvl, int seg_rows, int march_blocks, float *__restrict__ arena,
                                                    const float *__restrict__ luts, int skip_down2)
{
    SR_CHAIN_SETPRIO();
    const TileDev &T = tiles[blockIdx.z];
    if (lvl + 1 >= T.nl) return;
    if (skip_down2 && down2_takes(T)) return;                       // levels 1 and 2 of this tile come from k_down2_march
    const int ws = T.W[lvl], hs = T.H[lvl], wo = T.W[lvl + 1];
    const int ya = T.g0[lvl + 1], yb = T.g1[lvl + 1];
    const int ncg = down_ncg(ws, wo);
    if ((int)blockIdx.x >= march_blocks) {
        // The border columns the march leaves out: outputs 0 .. 3 and 4 (ncg + 1) .. wo - 1 (at most 12 columns;
        // every column when the level has no interior column group), one pixel per thread with the full border
        // rule -- the trailing blocks of the same launch, 16 columns x 16 rows each.
        const int tid = threadIdx.y * 64 + threadIdx.x;
        const int e = tid & 15;
        const int x = (ncg <= 0 || e < 4) ? e : 4 * (ncg + 1) + (e - 4);
        const int y = ya + ((int)blockIdx.x - march_blocks) * 16 + (tid >> 4);
        if (x >= wo || y >= yb) return;
        TileSrc S;
        S.p = nullptr;
        S.stride = 0;
        if (SRC == SRC_U8) S = srcs[blockIdx.z];
        down_pixel<SRC>(T, S, lvl, CN, x, y, arena, luts);
        return;
    }
    if (ncg <= 0 || yb <= ya) return;
    const int nseg = (yb - ya + seg_rows - 1) / seg_rows;
    const int cell = blockIdx.x * 256 + threadIdx.y * 64 + threadIdx.x;
    if (cell >= ncg * nseg) return;
    const int seg = cell / ncg, x0 = (1 + cell - seg * ncg) * 4;
    const int y_begin = ya + seg * seg_rows, y_end = min(y_begin + seg_rows, yb);
    const int po = T.P[lvl + 1];
    const size_t dplane = (size_t)T.H[lvl + 1] * po;
    float *dst = arena + T.g_off[lvl + 1] + x0;
    const int xb = 2 * x0 - 2;
    if (SRC == SRC_U8) {
        constexpr int NW = (CN == 3) ? 9 : 3, NV = 4 * CN;
        const TileSrc S = srcs[blockIdx.z];
        const unsigned char *base = (const unsigned char *)S.p;
        unsigned w0[NW], w1[NW], n0[NW], n1[NW];
        float e0[NV], o0[NV], e1[NV], o1[NV], e2[NV];
        down_load_u8<CN>(base, S.stride, reflect101_once(2 * y_begin - 2, hs), xb, w0);
        down_load_u8<CN>(base, S.stride, reflect101_once(2 * y_begin - 1, hs), xb, w1);
        down_load_u8<CN>(base, S.stride, 2 * y_begin, xb, n0);
        down_row_u8<CN>(w0, e0);
        down_row_u8<CN>(w1, o0);
        down_row_u8<CN>(n0, e1);
        down_load_u8<CN>(base, S.stride, reflect101_once(2 * y_begin + 1, hs), xb, w0);
        down_load_u8<CN>(base, S.stride, reflect101_once(2 * y_begin + 2, hs), xb, w1);
        for (int y = y_begin; y < y_end; ++y) {
            // the two rows of the NEXT output row are requested first and land in their own registers: a whole
            // iteration of arithmetic (row passes of the current rows, column pass, stores) covers their latency
            // (unconditional -- the last iteration re-reads its own rows -- so the compiler can keep exactly these
            // loads outstanding with a counted s_waitcnt instead of draining at a control-flow join)
            const int yn = min(y + 1, y_end - 1);
            down_load_u8<CN>(base, S.stride, reflect101_once(2 * yn + 1, hs), xb, n0);
            down_load_u8<CN>(base, S.stride, reflect101_once(2 * yn + 2, hs), xb, n1);
            down_row_u8<CN>(w0, o1);
            down_row_u8<CN>(w1, e2);
#pragma unroll
            for (int i = 0; i < NW; ++i) { w0[i] = n0[i]; w1[i] = n1[i]; }
#pragma unroll
            for (int c = 0; c < CN; ++c) {
                f4_t ov;
                float o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = k * CN + c;    // integers below 2^24 again: exact in any order
                    o[k] = fmaf(e1[i], 6.0f, fmaf(o0[i] + o1[i], 4.0f, e0[i] + e2[i])) * (1.0f / 256.0f);
                }
                ov.x = o[0]; ov.y = o[1]; ov.z = o[2]; ov.w = o[3];
                st_f4(dst + c * dplane + (size_t)y * po, ov);
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) { e0[i] = e1[i]; o0[i] = o1[i]; e1[i] = e2[i]; }
        }
    } else {  // SRC_PLANAR: fp32 rounds, the reference's evaluation order is kept
        const int ps = T.P[lvl];
        const size_t splane = (size_t)hs * ps;
#pragma unroll 1
        for (int c = 0; c < CN; ++c) {
            const float *plane = arena + T.g_off[lvl] + c * splane;
            float s0[11], s1[11], t0[11], t1[11];
            float e0[4], o0[4], e1[4], o1[4], e2[4];
            down_load_f32(plane, ps, reflect101_once(2 * y_begin - 2, hs), xb, s0);
            down_load_f32(plane, ps, reflect101_once(2 * y_begin - 1, hs), xb, s1);
            down_load_f32(plane, ps, 2 * y_begin, xb, t0);
            down_row_f32(s0, e0);
            down_row_f32(s1, o0);
            down_row_f32(t0, e1);
            down_load_f32(plane, ps, reflect101_once(2 * y_begin + 1, hs), xb, s0);
            down_load_f32(plane, ps, reflect101_once(2 * y_begin + 2, hs), xb, s1);
            for (int y = y_begin; y < y_end; ++y) {
                const int yn = min(y + 1, y_end - 1);   // next output row's rows first, into their own registers
                down_load_f32(plane, ps, reflect101_once(2 * yn + 1, hs), xb, t0);
                down_load_f32(plane, ps, reflect101_once(2 * yn + 2, hs), xb, t1);
                down_row_f32(s0, o1);
                down_row_f32(s1, e2);
#pragma unroll
                for (int i = 0; i < 11; ++i) { s0[i] = t0[i]; s1[i] = t1[i]; }
                f4_t ov;
                float o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    o[k] = ((((e1[k] * 6.0f + (o0[k] + o1[k]) * 4.0f) + e0[k]) + e2[k])) * (1.0f / 256.0f);
                ov.x = o[0]; ov.y = o[1]; ov.z = o[2]; ov.w = o[3];
                st_f4(dst + c * dplane + (size_t)y * po, ov);
#pragma unroll
                for (int k = 0; k < 4; ++k) { e0[k] = e1[k]; o0[k] = o1[k]; e1[k] = e2[k]; }
            }
        }
    }
}

// R_i for one level of every tile:  top level: G*W;  else up(R_{i+1}) + (G_i - up(G_{i+1})) * W_i
__global__ __launch_bounds__(256) void k_up_level(const TileDev *__restrict__ tiles, int lvl, int cn,
                                                  float *__restrict__ arena)
{
    const TileDev &T = tiles[blockIdx.z];
    if (lvl >= T.nl) return;
    const int w = T.W[lvl], h = T.H[lvl], p = T.P[lvl];
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = T.r0[lvl] + blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= T.r1[lvl]) return;
    const float wv = arena[T.w_off[lvl] + (size_t)y * p + x];
    const size_t plane = (size_t)h * p;
    const float *g = arena + T.g_off[lvl] + (size_t)y * p + x;
    float *r = arena + T.r_off[lvl] + (size_t)y * p + x;
    if (lvl == T.nl - 1) {
        for (int c = 0; c < cn; ++c) r[c * plane] = g[c * plane] * wv;
        return;
    }
    const int hs = T.H[lvl + 1], ws = T.W[lvl + 1], ps = T.P[lvl + 1];
    const size_t splane = (size_t)hs * ps;
    const float *gs = arena + T.g_off[lvl + 1];
    const float *rs = arena + T.r_off[lvl + 1];
    for (int c = 0; c < cn; ++c) {
        const float ug = up_sample(gs + c * splane, hs, ws, ps, y, x);
        const float ur = up_sample(rs + c * splane, hs, ws, ps, y, x);
        const float lap = g[c * plane] - ug;
        const float wl = lap * wv;
        r[c * plane] = ur + wl;
    }
}

// Final level, canvas-centric: for each canvas pixel sum the covering tiles in list order
// (acc += R_0, wacc += W_0), normalise, clip, truncate.  LAP == false: weighted_average_fusion.
template <int DT, bool LAP>
__global__ __launch_bounds__(256) void k_final(const TileDev *__restrict__ tiles, const TileSrc *__restrict__ srcs,
                                               int n, int cn, const float *__restrict__ arena,
                                               const float *__restrict__ luts, unsigned char *__restrict__ canvas,
                                               long long cstride, float *__restrict__ canvas_f32, int cw,
                                               int row_begin, int row_end)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = row_begin + blockIdx.y * 4 + threadIdx.y;
    if (x >= cw || y >= row_end) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float wacc = 0.f;
    for (int t = 0; t < n; ++t) {
        const TileDev &T = tiles[t];
        const int lx = x - T.x, ly = y - T.y;
        if (lx < 0 || ly < 0 || lx >= T.w || ly >= T.h) continue;
        const int d = min(min(ly, T.h - 1 - ly), min(lx, T.w - 1 - lx));
        const float w0 = luts[T.lut_off + min(d, T.fw)];
        const char *srow = (const char *)srcs[t].p + (size_t)ly * srcs[t].stride;
        for (int c = 0; c < cn; ++c) {
            float g0;
            if (DT == SRC_U8) g0 = (float)((const unsigned char *)srow)[lx * cn + c];
            else g0 = ((const float *)srow)[lx * cn + c];
            float r;
            if (LAP && T.nl > 1) {
                const int hs = T.H[1], ws = T.W[1], ps = T.P[1];
                const size_t splane = (size_t)hs * ps;
                const float ug = up_sample(arena + T.g_off[1] + c * splane, hs, ws, ps, ly, lx);
                const float ur = up_sample(arena + T.r_off[1] + c * splane, hs, ws, ps, ly, lx);
                const float lap = g0 - ug;
                const float wl = lap * w0;
                r = ur + wl;
            } else {
                r = g0 * w0;
            }
            acc[c] += r;
        }
        wacc += w0;
    }
    const float wv = wacc > 1e-6f ? wacc : 1e-6f;
    unsigned char *o = canvas + (size_t)y * cstride + (size_t)x * cn;
    for (int c = 0; c < cn; ++c) {
        const float v = acc[c] / wv;
        if (canvas_f32) canvas_f32[((size_t)y * cw + x) * cn + c] = v;
        const float cl = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
        o[c] = (unsigned char)cl;
    }
}

// ---------------------------------------------------------------------------------------------
// Final gather, register-blocked: one thread = 4 x 2 canvas pixels.  For every covering tile the
// thread loads ONE 3-row x 4-column neighbourhood of G_1 and R_1 per plane and evaluates the same
// per-pixel expressions as k_final from registers (12 loads per 8 pixels per plane per array
// instead of 6-9 per pixel).  x0 is a multiple of 4 and y0 - row_begin a multiple of 2, so the
// parity of the tile-local origin -- which selects the even/odd pyrUp phase of every pixel in the
// thread -- is uniform per tile across the launch (template XO / YO).
// ---------------------------------------------------------------------------------------------
template <int POS, bool ODD>
__device__ __forceinline__ float up_h_reg(const float (&v)[4], int ws, int sx)
{
    if (ws == 1) return ODD ? v[POS] * 8.0f : v[POS] * 6.0f + v[POS] * 2.0f;
    if (!ODD) {
        constexpr int PM = POS > 0 ? POS - 1 : 0;
        if (sx == 0) return v[POS] * 6.0f + v[POS + 1] * 2.0f;
        if (sx == ws - 1) return v[PM] + v[POS] * 7.0f;
        return (v[PM] + v[POS] * 6.0f) + v[POS + 1];
    }
    constexpr int PP = POS < 3 ? POS + 1 : 3;
    if (sx == ws - 1) return v[POS] * 8.0f;
    return (v[POS] + v[PP]) * 4.0f;
}

// h[r][k]: unnormalised horizontal pyrUp of loaded row r at thread pixel k
template <bool XO>
__device__ __forceinline__ void up_rows4(const float (&v)[3][4], int ws, int c0, float (&h)[3][4])
{
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        if (!XO) {
            h[r][0] = up_h_reg<1, false>(v[r], ws, c0 + 1);
            h[r][1] = up_h_reg<1, true>(v[r], ws, c0 + 1);
            h[r][2] = up_h_reg<2, false>(v[r], ws, c0 + 2);
            h[r][3] = up_h_reg<2, true>(v[r], ws, c0 + 2);
        } else {
            h[r][0] = up_h_reg<0, true>(v[r], ws, c0);
            h[r][1] = up_h_reg<1, false>(v[r], ws, c0 + 1);
            h[r][2] = up_h_reg<1, true>(v[r], ws, c0 + 1);
            h[r][3] = up_h_reg<2, false>(v[r], ws, c0 + 2);
        }
    }
}

// vertical combination for the thread's two rows (j = 0, 1) at pixel column k
template <bool YO>
__device__ __forceinline__ void up_cols2(const float (&h)[3][4], int r0, float (&u)[2][4])
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!YO) {
            // j = 0: even row, sy = r0 + 1;  j = 1: odd row, sy = r0 + 1
            const float top = (r0 + 1 == 0) ? h[2][k] : h[0][k];
            u[0][k] = ((top + h[1][k] * 6.0f) + h[2][k]) * (1.0f / 64.0f);
            u[1][k] = ((h[1][k] + h[2][k]) * 4.0f) * (1.0f / 64.0f);
        } else {
            // j = 0: odd row, sy = r0;  j = 1: even row, sy = r0 + 1
            const float top = (r0 + 1 == 0) ? h[2][k] : h[0][k];
            u[0][k] = ((h[0][k] + h[1][k]) * 4.0f) * (1.0f / 64.0f);
            u[1][k] = ((top + h[1][k] * 6.0f) + h[2][k]) * (1.0f / 64.0f);
        }
    }
}

template <bool XO, bool YO>
__device__ __forceinline__ void up_block(const float *__restrict__ plane, int hs, int ws, int ps, int r0, int c0,
                                         float (&u)[2][4])
{
    float v[3][4], h[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float *row = plane + (size_t)min(max(r0 + r, 0), hs - 1) * ps;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[r][c] = row[min(max(c0 + c, 0), ws - 1)];
    }
    up_rows4<XO>(v, ws, c0, h);
    up_cols2<YO>(h, r0, u);
}


// interior form of up_block: every pixel of the thread is away from the level-1 borders, so the
// border selects of up_h / the row clamps vanish; same expressions as the generic path otherwise
template <bool XO, bool YO>
__device__ __forceinline__ void up_block_interior(const float *__restrict__ plane, int ps, int r0, int c0,
                                                  float (&u)[2][4])
{
    float h[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const f4_t q = ld_f4_a4(plane + (size_t)(r0 + r) * ps + c0);
        // power-of-two factors of the odd phases are folded into the final constants (exact; see up_regs)
        if (!XO) {
            h[r][0] = (q.x + q.y * 6.0f) + q.z;
            h[r][1] = q.y + q.z;
            h[r][2] = (q.y + q.z * 6.0f) + q.w;
            h[r][3] = q.z + q.w;
        } else {
            h[r][0] = q.x + q.y;
            h[r][1] = (q.x + q.y * 6.0f) + q.z;
            h[r][2] = q.y + q.z;
            h[r][3] = (q.y + q.z * 6.0f) + q.w;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool kodd = XO ? ((k & 1) == 0) : ((k & 1) == 1);
        const float ce = kodd ? (1.0f / 16.0f) : (1.0f / 64.0f);
        const float co = kodd ? (1.0f / 4.0f) : (1.0f / 16.0f);
        const float ev = ((h[0][k] + h[1][k] * 6.0f) + h[2][k]) * ce;
        if (!YO) {
            u[0][k] = ev;
            u[1][k] = (h[1][k] + h[2][k]) * co;
        } else {
            u[0][k] = (h[0][k] + h[1][k]) * co;
            u[1][k] = ev;
        }
    }
}

// Is the thread's 4 x 2 rectangle (tile-local origin lx0, ly0; nx x ny of it on the canvas strip) an
// "interior" visit of tile D: all eight pixels inside the tile and every level-1 tap away from the borders?
// Interior visits run in the regular blocks of k_final_fast, everything else in its edge blocks; both evaluate this same test.
template <bool LAP>
__device__ __forceinline__ bool visit_is_interior(const FinalDesc &D, int lx0, int ly0, int nx, int ny)
{
    if (nx != 4 || ny != 2 || lx0 < 0 || ly0 < 0 || lx0 + 3 >= D.w || ly0 + 1 >= D.h) return false;
    if (LAP && D.nl > 1) {
        const int r0 = (ly0 - 1) >> 1, c0 = (lx0 - 1) >> 1;
        return c0 >= 0 && c0 + 3 <= D.W1 - 1 && r0 >= 0 && r0 + 2 <= D.H1 - 1;
    }
    return true;
}

__device__ __forceinline__ void tile_weights(const FinalDesc &D, const float *__restrict__ luts, int lx0, int ly0,
                                             float (&w0)[2][4])
{
    // edge distance of the whole 4 x 2 rectangle: beyond the feather width every weight is lut[fw]
    const int dmin = min(min(ly0, D.h - 2 - ly0), min(lx0, D.w - 4 - lx0));
    if (dmin >= D.fw) {
        const float wf = luts[D.lut_off + D.fw];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) w0[j][k] = wf;
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int lx = lx0 + k, ly = ly0 + j;
            const int d = min(min(ly, D.h - 1 - ly), min(lx, D.w - 1 - lx));
            w0[j][k] = luts[D.lut_off + min(max(d, 0), D.fw)];
        }
}

// pyrUp of one 3 x 4 register neighbourhood (rows q[0..2]) -> the thread's 2 x 4 pixels (interior form)
template <bool XO, bool YO>
__device__ __forceinline__ void up_regs(const f4_t (&q)[3], float (&u)[2][4])
{
    // Scaling by a power of two commutes with fp32 rounding, so the x4 of the odd phases ((a + b) * 4) is not
    // applied where the reference applies it but folded into the final constant: 1/64 (even,even), 1/16 (one odd
    // phase), 1/4 (odd,odd).  Bit-identical to the reference order, 10 multiplies fewer per plane.
    float h[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        if (!XO) {
            h[r][0] = (q[r].x + q[r].y * 6.0f) + q[r].z;
            h[r][1] = q[r].y + q[r].z;
            h[r][2] = (q[r].y + q[r].z * 6.0f) + q[r].w;
            h[r][3] = q[r].z + q[r].w;
        } else {
            h[r][0] = q[r].x + q[r].y;
            h[r][1] = (q[r].x + q[r].y * 6.0f) + q[r].z;
            h[r][2] = q[r].y + q[r].z;
            h[r][3] = (q[r].y + q[r].z * 6.0f) + q[r].w;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool kodd = XO ? ((k & 1) == 0) : ((k & 1) == 1);          // pixel k is an odd pyrUp column phase
        const float ce = kodd ? (1.0f / 16.0f) : (1.0f / 64.0f);          // even row phase
        const float co = kodd ? (1.0f / 4.0f) : (1.0f / 16.0f);           // odd row phase
        const float ev = ((h[0][k] + h[1][k] * 6.0f) + h[2][k]) * ce;
        if (!YO) {
            u[0][k] = ev;
            u[1][k] = (h[1][k] + h[2][k]) * co;
        } else {
            u[0][k] = (h[0][k] + h[1][k]) * co;
            u[1][k] = ev;
        }
    }
}

// weights of an interior visit (all eight pixels inside the tile).  The LUT is monotone in the edge distance, so
// lut[min(dy, dx)] == min(lut[dy], lut[dx]) exactly: 2 + 4 LUT reads and 8 v_min_f32 instead of 8 reads behind
// 8 three-way integer minima.
__device__ __forceinline__ void tile_weights_interior(const FinalDesc &D, const float *__restrict__ luts, int lx0, int ly0,
                                                      float (&w0)[2][4])
{
    const int dmin = min(min(ly0, D.h - 2 - ly0), min(lx0, D.w - 4 - lx0));
    const float *lut = luts + D.lut_off;
    if (dmin >= D.fw) {
        const float wf = lut[D.fw];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) w0[j][k] = wf;
        return;
    }
    float fy[2], fx[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) fy[j] = lut[min(min(ly0 + j, D.h - 1 - ly0 - j), D.fw)];
#pragma unroll
    for (int k = 0; k < 4; ++k) fx[k] = lut[min(min(lx0 + k, D.w - 1 - lx0 - k), D.fw)];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) w0[j][k] = __builtin_fminf(fy[j], fx[k]);
}

// interior visit: every global load of the visit is issued before the first use, then straight-line arithmetic.
// Plane bases are wave-uniform (SGPR pair); the per-lane part of an address is one 32-bit byte offset.
template <int DT, bool LAP, int CN, bool XO, bool YO>
__device__ __forceinline__ void gather_tile_fast(const FinalDesc &D, const float *__restrict__ arena,
                                                 const float *__restrict__ luts, int lx0, int ly0,
                                                 float (&acc)[2][4][CN], float (&wacc)[2][4])
{
    const bool pyr = LAP && D.nl > 1;
    // ---- loads of the level-0 pixels and the weights ------------------------------------------------------
    float g0[2][4][CN];
    u3_t qs[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const char *srow = (const char *)D.src + (size_t)(ly0 + j) * D.stride;
        if (DT == SRC_U8 && CN == 3) {
            qs[j] = ld_u3_a1_g(srow + (size_t)lx0 * 3);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int c = 0; c < CN; ++c) {
                    if (DT == SRC_U8) g0[j][k][c] = (float)((const unsigned char *)srow)[(lx0 + k) * CN + c];
                    else g0[j][k][c] = ((const float *)srow)[(lx0 + k) * CN + c];
                }
        }
    }
    float w0[2][4];
    tile_weights_interior(D, luts, lx0, ly0, w0);
    // u8 RGB: the pixels stay packed (6 dwords) and each plane's eight values are pulled out when that plane is
    // processed (v_cvt_f32_ubyteN, one instruction per value either way): 18 fewer live registers than unpacking up front
    constexpr bool LAZY = (DT == SRC_U8 && CN == 3);
    auto px = [&](int j, int k, int c) -> float {
        if (LAZY) {
            const int b = 3 * k + c;
            unsigned wd = (b >> 2) == 0 ? qs[j].x : ((b >> 2) == 1 ? qs[j].y : qs[j].z);
            asm volatile("" : "+v"(wd));      // keeps the conversion at its use: hoisted above the branch it would cost 24 live registers
            return (float)((wd >> (8 * (b & 3))) & 0xFFu);
        }
        return g0[j][k][c];
    };
    if (pyr) {                               // tile-uniform: a real branch, not a select per value
        // plane by plane: the six 16-byte loads of a plane are in flight together, the next plane's are issued
        // before this plane's arithmetic (two planes of level-1 data live at a time, not three)
        const int r0 = (ly0 - 1) >> 1, c0 = (lx0 - 1) >> 1;
        const size_t splane = (size_t)D.H1 * D.P1;
        const unsigned rowb = (unsigned)D.P1 * 4u;
        const unsigned o = (unsigned)r0 * rowb + (unsigned)c0 * 4u;      // byte offset inside a plane (planes < 4 GB)
        f4_t qg[3], qr[3], ng[3], nr[3];
        {
            const char *gb = (const char *)(arena + D.g1), *rb = (const char *)(arena + D.r1);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                qg[r] = ld_f4_a4((const float *)(gb + (o + r * rowb)));
                qr[r] = ld_f4_a4((const float *)(rb + (o + r * rowb)));
            }
        }
#pragma unroll
        for (int c = 0; c < CN; ++c) {
#ifndef SR_FINAL_NOPREF
            if (c + 1 < CN) {
                const char *gb = (const char *)(arena + D.g1 + (c + 1) * splane);
                const char *rb = (const char *)(arena + D.r1 + (c + 1) * splane);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    ng[r] = ld_f4_a4((const float *)(gb + (o + r * rowb)));
                    nr[r] = ld_f4_a4((const float *)(rb + (o + r * rowb)));
                }
            }
#else
            if (c > 0) {
                const char *gb = (const char *)(arena + D.g1 + c * splane);
                const char *rb = (const char *)(arena + D.r1 + c * splane);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    qg[r] = ld_f4_a4((const float *)(gb + (o + r * rowb)));
                    qr[r] = ld_f4_a4((const float *)(rb + (o + r * rowb)));
                }
            }
#endif
            float ug[2][4], ur[2][4];
            up_regs<XO, YO>(qg, ug);
            up_regs<XO, YO>(qr, ur);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float lap = px(j, k, c) - ug[j][k];
                    const float wl = lap * w0[j][k];
                    acc[j][k][c] += ur[j][k] + wl;
                }
#ifndef SR_FINAL_NOPREF
            if (c + 1 < CN) {
#pragma unroll
                for (int r = 0; r < 3; ++r) { qg[r] = ng[r]; qr[r] = nr[r]; }
            }
#endif
        }
    } else {
#pragma unroll
        for (int c = 0; c < CN; ++c)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[j][k][c] += px(j, k, c) * w0[j][k];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) wacc[j][k] += w0[j][k];
}

// any visit: clamped taps, border selects, per-pixel validity (tile / level borders, ragged canvas edges)
template <int DT, bool LAP, int CN, bool XO, bool YO>
__device__ __forceinline__ void gather_tile_generic(const FinalDesc &D, const float *__restrict__ arena,
                                                    const float *__restrict__ luts, int lx0, int ly0, unsigned valid,
                                                    float (&acc)[2][4][CN], float (&wacc)[2][4])
{
    float w0[2][4];
    tile_weights(D, luts, lx0, ly0, w0);
    const int r0 = (ly0 - 1) >> 1, c0 = (lx0 - 1) >> 1;
    const bool pyr = LAP && D.nl > 1;
    const int hs = D.H1, ws = D.W1, ps = D.P1;
    const size_t splane = (size_t)hs * ps;
#pragma unroll
    for (int c = 0; c < CN; ++c) {
        float ug[2][4], ur[2][4];
        if (pyr) {
            up_block<XO, YO>(arena + D.g1 + c * splane, hs, ws, ps, r0, c0, ug);
            up_block<XO, YO>(arena + D.r1 + c * splane, hs, ws, ps, r0, c0, ur);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!(valid & (1u << (j * 4 + k)))) continue;
                const char *srow = (const char *)D.src + (size_t)(ly0 + j) * D.stride;
                float g0;
                if (DT == SRC_U8) g0 = (float)((const unsigned char *)srow)[(lx0 + k) * CN + c];
                else g0 = ((const float *)srow)[(lx0 + k) * CN + c];
                float r;
                if (pyr) {
                    const float lap = g0 - ug[j][k];
                    const float wl = lap * w0[j][k];
                    r = ur[j][k] + wl;
                } else {
                    r = g0 * w0[j][k];
                }
                acc[j][k][c] += r;
            }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (valid & (1u << (j * 4 + k))) wacc[j][k] += w0[j][k];
}

// up_block_interior<false, YO> in packed fp32: the thread's 4 x 2 patch from rows r0 .. r0 + 2,
// columns c0 .. c0 + 3 of a planar level.  Pairs run over the columns of equal phase, (k0, k2) and (k1, k3): with
// A = the four taps and B = the two taps one column to the right (c0 is odd, so B is an aligned 8-byte load and A's halves
// are aligned register pairs) the horizontal pass is  (h0, h2) = (A01 + B01 * 6) + A23,  (h1, h3) = B01 + A23  -- the
// same operands in the same order as the scalar form, two results per instruction.  24 packed instead of 48 scalar.
template <bool YO>
__device__ __forceinline__ void up_block_interior_pk(const float *__restrict__ plane, int ps, int r0, int c0, float (&u)[2][4])
{
    f2_t h02[3], h13[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float *row = plane + (size_t)(r0 + r) * ps + c0;
        const f4_t a = ld_f4_a4(row);
        const f2_t b01 = *(const f2_a8_t *)(row + 1);
        f2_t a01, a23;
        a01.x = a.x; a01.y = a.y; a23.x = a.z; a23.y = a.w;
        h02[r] = (a01 + b01 * 6.0f) + a23;
        h13[r] = b01 + a23;
    }
    // columns k0, k2 are the even pyrUp phase (1/64 even rows, 1/16 odd rows), k1, k3 the odd one (1/16, 1/4)
    // (YO: the patch starts on an odd row, whose two taps are rows 0 and 1; the even row follows)
    const f2_t ev02 = ((h02[0] + h02[1] * 6.0f) + h02[2]) * (1.0f / 64.0f);
    const f2_t ev13 = ((h13[0] + h13[1] * 6.0f) + h13[2]) * (1.0f / 16.0f);
    const f2_t od02 = (YO ? (h02[0] + h02[1]) : (h02[1] + h02[2])) * (1.0f / 16.0f);
    const f2_t od13 = (YO ? (h13[0] + h13[1]) : (h13[1] + h13[2])) * (1.0f / 4.0f);
    constexpr int E = YO ? 1 : 0, O = YO ? 0 : 1;
    u[E][0] = ev02.x; u[E][2] = ev02.y; u[E][1] = ev13.x; u[E][3] = ev13.y;
    u[O][0] = od02.x; u[O][2] = od02.y; u[O][1] = od13.x; u[O][3] = od13.y;
}

// R_i for one level of every tile, register-blocked: one thread = 4 x 2 pixels of level i, all planes.
// Tile-local x0 is a multiple of 4 (so the pyrUp column phase is fixed: XO = false); the row phase follows
// the parity of the row-window start and is uniform per tile.  Same expressions as k_up_level.
template <int CN, bool YO>
__device__ __forceinline__ void up_level_thread(const TileDev &T, int lvl, float *__restrict__ arena, int x0, int y0,
                                                int ny)
{
    const int h = T.H[lvl], p = T.P[lvl];
    const size_t plane = (size_t)h * p;
    const int hs = T.H[lvl + 1], ws = T.W[lvl + 1], ps = T.P[lvl + 1];
    const size_t splane = (size_t)hs * ps;
    const int r0 = (y0 - 1) >> 1, c0 = (x0 - 1) >> 1;
    const bool interior = c0 >= 0 && c0 + 3 <= ws - 1 && r0 >= 0 && r0 + 2 <= hs - 1;
    const bool two = ny > 1;
    const float *wrow = arena + T.w_off[lvl] + (size_t)y0 * p + x0;
    const f4_t w0v = ld_f4(wrow);
    const f4_t w1v = two ? ld_f4(wrow + p) : w0v;
#pragma unroll
    for (int c = 0; c < CN; ++c) {
        const float *g = arena + T.g_off[lvl] + c * plane + (size_t)y0 * p + x0;
        float *r = arena + T.r_off[lvl] + c * plane + (size_t)y0 * p + x0;
        const f4_t g0v = ld_f4(g);
        const f4_t g1v = two ? ld_f4(g + p) : g0v;
        float ug[2][4], ur[2][4];
        const float *gs = arena + T.g_off[lvl + 1] + c * splane;
        const float *rs = arena + T.r_off[lvl + 1] + c * splane;
        if (interior) {
            up_block_interior_pk<YO>(gs, ps, r0, c0, ug);
            up_block_interior_pk<YO>(rs, ps, r0, c0, ur);
        } else {
            up_block<false, YO>(gs, hs, ws, ps, r0, c0, ug);
            up_block<false, YO>(rs, hs, ws, ps, r0, c0, ur);
        }
        f4_t o0, o1;
        o0.x = ur[0][0] + (g0v.x - ug[0][0]) * w0v.x;
        o0.y = ur[0][1] + (g0v.y - ug[0][1]) * w0v.y;
        o0.z = ur[0][2] + (g0v.z - ug[0][2]) * w0v.z;
        o0.w = ur[0][3] + (g0v.w - ug[0][3]) * w0v.w;
        o1.x = ur[1][0] + (g1v.x - ug[1][0]) * w1v.x;
        o1.y = ur[1][1] + (g1v.y - ug[1][1]) * w1v.y;
        o1.z = ur[1][2] + (g1v.z - ug[1][2]) * w1v.z;
        o1.w = ur[1][3] + (g1v.w - ug[1][3]) * w1v.w;
        st_f4(r, o0);                  // columns >= w land in the row padding (pitch is a multiple of 16)
        if (two) st_f4(r + p, o1);
    }
}

template <int CN>
__global__ __launch_bounds__(256) void k_up_level_blk(const TileDev *__restrict__ tiles, int lvl, float *__restrict__ arena)
{
    SR_CHAIN_SETPRIO();
    const TileDev &T = tiles[blockIdx.z];
    if (lvl >= T.nl) return;
    const int w = T.W[lvl], h = T.H[lvl], p = T.P[lvl];
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y0 = T.r0[lvl] + (blockIdx.y * 4 + threadIdx.y) * 2;
    if (x0 >= w || y0 >= T.r1[lvl]) return;
    const int ny = min(2, T.r1[lvl] - y0);
    if (lvl == T.nl - 1) {
        const size_t plane = (size_t)h * p;
        for (int j = 0; j < ny; ++j) {
            const f4_t wv = ld_f4(arena + T.w_off[lvl] + (size_t)(y0 + j) * p + x0);
#pragma unroll
            for (int c = 0; c < CN; ++c) {
                const f4_t gv = ld_f4(arena + T.g_off[lvl] + c * plane + (size_t)(y0 + j) * p + x0);
                st_f4(arena + T.r_off[lvl] + c * plane + (size_t)(y0 + j) * p + x0, gv * wv);
            }
        }
        return;
    }
    if (y0 & 1) up_level_thread<CN, true>(T, lvl, arena, x0, y0, ny);
    else up_level_thread<CN, false>(T, lvl, arena, x0, y0, ny);
}

// a / w for the channels of one pixel, IEEE-correct: exactly the fma chain the compiler emits for an fp32 division
// (rcp, one Newton step, quotient, two residual corrections) without the v_div_scale / v_div_fixup wrapping, which is
// the identity for these operands (w in [1e-6, n_tiles], |a| a few thousand at most) -- and with the reciprocal
// refined once per pixel instead of once per channel.
template <int CN>
__device__ __forceinline__ void div_shared(const float (&a)[CN], float w, float (&q)[CN])
{
    float r = __builtin_amdgcn_rcpf(w);
    r = fmaf(fmaf(-w, r, 1.0f), r, r);
    if (CN == 3) {
        // channels 0 and 1 as one packed pair (v_pk_mul / v_pk_fma: the same fma chain per element), channel 2 scalar
        f2_t a01, nw, rr;
        a01.x = a[0]; a01.y = a[1];
        nw.x = nw.y = -w;
        rr.x = rr.y = r;
        f2_t t = a01 * rr;
        t = __builtin_elementwise_fma(__builtin_elementwise_fma(nw, t, a01), rr, t);
        t = __builtin_elementwise_fma(__builtin_elementwise_fma(nw, t, a01), rr, t);
        q[0] = t.x;
        q[1] = t.y;
        float t2 = a[2] * r;
        t2 = fmaf(fmaf(-w, t2, a[2]), r, t2);
        q[2] = fmaf(fmaf(-w, t2, a[2]), r, t2);
        return;
    }
#pragma unroll
    for (int c = 0; c < CN; ++c) {
        float t = a[c] * r;
        t = fmaf(fmaf(-w, t, a[c]), r, t);
        q[c] = fmaf(fmaf(-w, t, a[c]), r, t);
    }
}

// normalise, clip, truncate and store the thread's pixels
template <int CN, int NR = 2>
__device__ __forceinline__ void store_pixels(float (&acc)[NR][4][CN], const float (&wacc)[NR][4],
                                             unsigned char *__restrict__ canvas, long long cstride,
                                             float *__restrict__ canvas_f32, int cw, int x0, int y0, int nx, int ny)
{
    const bool vec_ok = (CN == 3) && nx == 4 && ((cstride & 3) == 0) && ((((size_t)canvas) & 3) == 0);
    // x / 1.0f == x: where every pixel of the wave has sum-of-weights exactly 1 (single coverage beyond the
    // feather zone, about half of a grid canvas) the divisions are skipped -- wave-uniform branch
    bool ones = true;
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) ones = ones && (wacc[j][k] == 1.0f);
    if (__all(ones) == 0) {
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) div_shared<CN>(acc[j][k], __builtin_fmaxf(wacc[j][k], 1e-6f), acc[j][k]);
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        if (j >= ny) continue;
        unsigned ob[4 * CN];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int c = 0; c < CN; ++c) {
                const float v = acc[j][k][c];
                if (canvas_f32 && k < nx) canvas_f32[((size_t)(y0 + j) * cw + x0 + k) * CN + c] = v;
                ob[k * CN + c] = (unsigned)__builtin_amdgcn_fmed3f(v, 0.0f, 255.0f);     // clip, then truncate
            }
        }
        unsigned char *o = canvas + (size_t)(y0 + j) * cstride + (size_t)x0 * CN;
        if (vec_ok) {
            unsigned int *o32 = (unsigned int *)o;
#pragma unroll
            for (int q = 0; q < 3; ++q)
                o32[q] = ob[4 * q] | (ob[4 * q + 1] << 8) | (ob[4 * q + 2] << 16) | (ob[4 * q + 3] << 24);
        } else {
            for (int k = 0; k < nx; ++k)
#pragma unroll
                for (int c = 0; c < CN; ++c) o[k * CN + c] = (unsigned char)ob[k * CN + c];
        }
    }
}

// Candidate tiles of a 256 x 8 pixel block come from a table built on the host when the plan is made (the tile
// arrangement is fixed per plan): cand_off[blk] .. cand_off[blk + 1] index cand_idx, tiles in list order.  Block
// id, list entries and the 80-byte descriptors are wave-uniform, so they travel through the scalar cache into
// SGPRs: no LDS staging and no barrier before the first vector load.

// Final gather, border part: the cells the interior part leaves out (a border visit).  Runs over the blocks of the
// edge work list built on the host when the plan is made: 256 x 8 pixel blocks along horizontal tile edges (shape 0),
// 16 x 128 pixel blocks along vertical ones (shape 1).  These are the leading blocks of k_final_fast's launch (the
// slow, divergent ones are scheduled first, the kernel's tail is made of regular blocks).
template <int DT, bool LAP, int CN>
__device__ __forceinline__ void final_edge_block(const FinalDesc *__restrict__ descs, const int4 *__restrict__ edge_blocks,
                                                 int ebi, const int *__restrict__ cand_idx,
                                                 const float *__restrict__ arena, const float *__restrict__ luts,
                                                 unsigned char *__restrict__ canvas, long long cstride,
                                                 float *__restrict__ canvas_f32, int cw, int row_begin, int row_end)
{
    const int4 eb = edge_blocks[ebi];
    const int c_begin = eb.w, c_end = edge_blocks[ebi + 1].w;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const int x0 = eb.x + (eb.z ? (tid & 3) : (tid & 63)) * 4;
    const int y0 = eb.y + (eb.z ? (tid >> 2) : (tid >> 6)) * 2;
    if (x0 >= cw || y0 >= row_end) return;
    const int nx = min(4, cw - x0), ny = min(2, row_end - y0);
    bool edge = false;
    for (int i = c_begin; i < c_end; ++i) {
        const FinalDesc &D = descs[cand_idx[i]];
        const int lx0 = x0 - D.x, ly0 = y0 - D.y;
        if (lx0 + nx <= 0 || ly0 + ny <= 0 || lx0 >= D.w || ly0 >= D.h) continue;
        if (!visit_is_interior<LAP>(D, lx0, ly0, nx, ny)) edge = true;
    }
    if (!edge) return;
    float acc[2][4][CN], wacc[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            wacc[j][k] = 0.f;
#pragma unroll
            for (int c = 0; c < CN; ++c) acc[j][k][c] = 0.f;
        }
    for (int i = c_begin; i < c_end; ++i) {
        const FinalDesc &D = descs[cand_idx[i]];
        const int lx0 = x0 - D.x, ly0 = y0 - D.y;
        if (lx0 + nx <= 0 || ly0 + ny <= 0 || lx0 >= D.w || ly0 >= D.h) continue;
        unsigned valid = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (j < ny && k < nx && lx0 + k >= 0 && lx0 + k < D.w && ly0 + j >= 0 && ly0 + j < D.h)
                    valid |= 1u << (j * 4 + k);
        const bool xo = (D.x & 1) != 0;
        const bool yo = ((row_begin - D.y) & 1) != 0;
        if (!xo && !yo) gather_tile_generic<DT, LAP, CN, false, false>(D, arena, luts, lx0, ly0, valid, acc, wacc);
        else if (xo && !yo) gather_tile_generic<DT, LAP, CN, true, false>(D, arena, luts, lx0, ly0, valid, acc, wacc);
        else if (!xo && yo) gather_tile_generic<DT, LAP, CN, false, true>(D, arena, luts, lx0, ly0, valid, acc, wacc);
        else gather_tile_generic<DT, LAP, CN, true, true>(D, arena, luts, lx0, ly0, valid, acc, wacc);
    }
    store_pixels<CN>(acc, wacc, canvas, cstride, canvas_f32, cw, x0, y0, nx, ny);
}

// Final gather.  The first n_edge blocks of the (one-dimensional) grid work through the edge list (above); the others
// are the regular FIN_BW x FIN_BH pixel blocks: threads all of whose tile visits are interior compute here, threads with
// any border visit leave their pixels to the edge blocks.
#ifndef SR_FINAL_WAVES
#define SR_FINAL_WAVES 3
#endif
#ifndef FIN_TX
#define FIN_TX 64                 /* threads across a regular block */
#endif
#define FIN_TY (256 / FIN_TX)
#define FIN_BW (4 * FIN_TX)       /* canvas pixels per regular block */
#define FIN_BH (2 * FIN_TY)
template <int DT, bool LAP, int CN>
__global__ __launch_bounds__(256, SR_FINAL_WAVES) void k_final_fast(const FinalDesc *__restrict__ descs,
                                                       const int *__restrict__ cand_off, const int *__restrict__ cand_idx,
                                                       const int4 *__restrict__ edge_blocks, const int *__restrict__ edge_cand,
                                                       int n_edge, int nbx_r,
                                                       const float *__restrict__ arena, const float *__restrict__ luts,
                                                       unsigned char *__restrict__ canvas, long long cstride,
                                                       float *__restrict__ canvas_f32, int cw, int row_begin, int row_end)
{
    if ((int)blockIdx.x < n_edge) {
        final_edge_block<DT, LAP, CN>(descs, edge_blocks, (int)blockIdx.x, edge_cand, arena, luts, canvas, cstride, canvas_f32, cw,
                                      row_begin, row_end);
        return;
    }
    // regular blocks: FIN_BW x FIN_BH canvas pixels, FIN_TX x FIN_TY threads of 4 x 2 pixels (a wave covers
    // 64 / FIN_TX thread rows).  Squarer blocks re-read fewer level-1 halo rows: a block needs FIN_BH / 2 + 2 of them.
    const int blk = (int)blockIdx.x - n_edge;
    const int by = blk / nbx_r, bx = blk - by * nbx_r;
    const int c_begin = cand_off[blk], c_end = cand_off[blk + 1];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const int x0 = (bx * FIN_TX + (tid % FIN_TX)) * 4;
    const int y0 = row_begin + (by * FIN_TY + tid / FIN_TX) * 2;
    if (x0 >= cw || y0 >= row_end) return;
    const int nx = min(4, cw - x0), ny = min(2, row_end - y0);
    float acc[2][4][CN], wacc[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            wacc[j][k] = 0.f;
#pragma unroll
            for (int c = 0; c < CN; ++c) acc[j][k][c] = 0.f;
        }
    for (int i = c_begin; i < c_end; ++i) {
        const FinalDesc &D = descs[cand_idx[i]];
        const int lx0 = x0 - D.x, ly0 = y0 - D.y;
        if (lx0 + nx <= 0 || ly0 + ny <= 0 || lx0 >= D.w || ly0 >= D.h) continue;
        // one border visit sends the whole thread to the edge pass (which recomputes every visit)
        if (!visit_is_interior<LAP>(D, lx0, ly0, nx, ny)) return;
        const bool xo = (D.x & 1) != 0;                      // x0 is a multiple of 4
        const bool yo = ((row_begin - D.y) & 1) != 0;        // y0 - row_begin is a multiple of 2
        if (!xo && !yo) gather_tile_fast<DT, LAP, CN, false, false>(D, arena, luts, lx0, ly0, acc, wacc);
        else if (xo && !yo) gather_tile_fast<DT, LAP, CN, true, false>(D, arena, luts, lx0, ly0, acc, wacc);
        else if (!xo && yo) gather_tile_fast<DT, LAP, CN, false, true>(D, arena, luts, lx0, ly0, acc, wacc);
        else gather_tile_fast<DT, LAP, CN, true, true>(D, arena, luts, lx0, ly0, acc, wacc);
    }
    store_pixels<CN>(acc, wacc, canvas, cstride, canvas_f32, cw, x0, y0, nx, ny);
}

// ---------------------------------------------------------------------------------------------
// Fused final gather: levels 1 -> 0 -> canvas in ONE kernel.  R_1 (the collapsed level 1, reference
// blending_module.py:340-363) is never written to memory: for every (block, covering tile) the block first builds the
// window of R_1 it is about to sample -- each thread one 4 x 2 patch of one plane, exactly the expressions of
// k_up_level_blk, from G_1 / W_1 / G_2 / R_2 -- into LDS together with the G_1 values it was made from, then every
// thread gathers its 4 x 4 canvas pixels from those two LDS windows (blending_module.py:474-506: acc += R_0, wacc += W_0,
// normalise, clip, truncate).  Against k_up_level_blk(level 1) + k_final_fast this drops the R_1 round trip
// (12 B written + 12-18 B re-read per level-1 pixel), one launch, and the 3 x re-read of level-1 rows (a thread's 4 x 4
// pixels share 4 level-1 rows; a block's window is fetched once, coalesced).  Same fp32 expression order everywhere:
// bit-identical to the unfused path.
//   regular block: 128 x 32 canvas pixels, 512 threads of 4 x 2 pixels; level-1 window <= 20 rows x 72 columns per plane
//   edge blocks (cells with a border visit): 256 x 16 or 32 x 128 pixels, generic per-pixel border rules, same windows
// The kernels of this path issue at one VALU instruction per 4 cycles and wave whatever the type, so the lever is the
// instruction count: the two LDS windows are interleaved per pixel and every pyrUp step runs on (g, r) pairs in v_pk_*.
// ---------------------------------------------------------------------------------------------
#ifndef FU_THREADS
#define FU_THREADS 256        /* threads (= 4 x 2 cells) per block */
#endif
#ifndef FU_WAVES
#define FU_WAVES 4           /* waves per SIMD the register allocation is held to */
#endif
#define FU_BW 128
#define FU_BH (FU_THREADS / 16)             /* 32 cells across, FU_THREADS / 32 cell rows of 2 pixels */
#define FU_E0H (FU_THREADS / 32)            /* edge shape 0: 256 x FU_E0H pixels (64 cells across) */
#define FU_E1W 16                           /* edge shape 1: FU_E1W x FU_E1H pixels (4 cells across: a vertical tile edge
                                               makes 1 - 3 of them border cells) */
#define FU_E1H (FU_THREADS / 2)
#ifndef FU_DB
#define FU_DB 0                /* 1: two LDS windows per block taken in turn (one barrier per tile instead of two): measured
                                  and rejected, 0.21 -> 0.26 ms for the 200 MP grid's remainder -- three blocks per CU instead of four */
#endif
#define FU_LP 72              /* LDS pitch (floats) of a regular block's window: 18 patches of 4 columns */
/* pixels per plane window (two floats each): rows = block rows / 2 + 3, rounded up to even; 72 (regular), 136 (shape 0) or
   24 (shape 1) columns */
#define FU_ROWS_EVEN(bh) ((((bh) / 2 + 3) + 1) / 2 * 2)
#define FU_MAX3(a, b, c) ((a) > (b) ? ((a) > (c) ? (a) : (c)) : ((b) > (c) ? (b) : (c)))
#define FU_E1LP (((FU_E1W / 2 + 2 + 3) + 3) / 4 * 4)      /* window columns of shape 1, whole patches */
#define FU_PLANE FU_MAX3(FU_ROWS_EVEN(FU_BH) * 72, FU_ROWS_EVEN(FU_E0H) * 136, FU_ROWS_EVEN(FU_E1H) * FU_E1LP)

// The level-1 window (tile coordinates) a block of bw x bh canvas pixels at tile-local (lxa, lya) samples: rows R0 ..,
// columns C0 .. in patches of 2 rows x 4 columns (R0 even, C0 a multiple of 4: the alignment k_up_level_blk's threads have).
__device__ __forceinline__ bool fused_window(const FinalDesc &D, int lxa, int lya, int bw, int bh, int &R0, int &C0, int &npr, int &npc)
{
    const int c_lo = max((lxa - 1) >> 1, 0), r_lo = max((lya - 1) >> 1, 0);
    const int c_hi = min(((lxa + bw - 5) >> 1) + 3, D.W1 - 1), r_hi = min(((lya + bh - 3) >> 1) + 2, D.H1 - 1);
    C0 = c_lo & ~3;
    R0 = r_lo & ~1;
    if (c_hi < C0 || r_hi < R0) return false;
    npc = (c_hi - C0) / 4 + 1;
    npr = (r_hi - R0) / 2 + 1;
    return true;
}

// Stage 1: R_1 and G_1 of the window into LDS, interleaved per pixel as (g, r) pairs: lds[plane][row][col][2] -- stage 2
// then reads a pixel's two values as one aligned 8-byte pair and runs the pyrUp of both arrays in packed fp32 (v_pk_*,
// IEEE per element: same roundings as the scalar form).  One item = one 4 x 2 patch of one plane = up_level_thread's work.
template <int CN>
__device__ __forceinline__ void fused_stage1(const FinalDesc &D, const float *__restrict__ arena, float *lds, int R0, int C0,
                                             int npr, int npc, int LP, int tid)
{
    const int per_plane = npr * npc, n_items = per_plane * CN;
    const size_t plane1 = (size_t)D.H1 * D.P1, plane2 = (size_t)D.H2 * D.P2;
    // item -> (plane, patch row, patch column) with multiply-shift divisions (x < 1024, divisor d <= 256, m = ceil(2^20 / d):
    // exact because x * (m * d - 2^20) < 1024 * 256 < 2^20)
    const unsigned m_pp = ((1u << 20) + per_plane - 1) / per_plane, m_pc = ((1u << 20) + npc - 1) / npc;
    for (int item = tid; item < n_items; item += FU_THREADS) {
        const int c = (int)(((unsigned)item * m_pp) >> 20), rem = item - c * per_plane;
        const int pr = (int)(((unsigned)rem * m_pc) >> 20), pc = rem - pr * npc;
        const int px = C0 + 4 * pc, py = R0 + 2 * pr;
        if (px >= D.W1 || py >= D.H1) continue;
        // every load of the item is issued before the first use (no branch between them: a conditional second row would
        // put a full memory round trip between the two halves); a patch on the last odd row re-reads its own row
        const int row1 = (py + 1 < D.H1) ? D.P1 : 0;
        const float *g = arena + D.g1 + c * plane1 + (size_t)py * D.P1 + px;
        const float *wr = arena + D.w1 + (size_t)py * D.P1 + px;
        const f4_t g0v = ld_f4(g), g1v = ld_f4(g + row1);
        const f4_t w0v = ld_f4(wr), w1v = ld_f4(wr + row1);
        f4_t o0, o1;
        if (D.nl == 2) {                         // level 1 is the top of this tile's pyramid: R = G * W
            o0 = g0v * w0v;
            o1 = g1v * w1v;
        } else {
            const int r0 = (py - 1) >> 1, c0 = (px - 1) >> 1;
            const bool interior = c0 >= 0 && c0 + 3 <= D.W2 - 1 && r0 >= 0 && r0 + 2 <= D.H2 - 1;
            const float *gs = arena + D.g2 + c * plane2, *rs = arena + D.r2 + c * plane2;
            float ug[2][4], ur[2][4];
            if (interior) {
                up_block_interior_pk<false>(gs, D.P2, r0, c0, ug);
                up_block_interior_pk<false>(rs, D.P2, r0, c0, ur);
            } else {
                up_block<false, false>(gs, D.H2, D.W2, D.P2, r0, c0, ug);
                up_block<false, false>(rs, D.H2, D.W2, D.P2, r0, c0, ur);
            }
            o0.x = ur[0][0] + (g0v.x - ug[0][0]) * w0v.x;
            o0.y = ur[0][1] + (g0v.y - ug[0][1]) * w0v.y;
            o0.z = ur[0][2] + (g0v.z - ug[0][2]) * w0v.z;
            o0.w = ur[0][3] + (g0v.w - ug[0][3]) * w0v.w;
            o1.x = ur[1][0] + (g1v.x - ug[1][0]) * w1v.x;
            o1.y = ur[1][1] + (g1v.y - ug[1][1]) * w1v.y;
            o1.z = ur[1][2] + (g1v.z - ug[1][2]) * w1v.z;
            o1.w = ur[1][3] + (g1v.w - ug[1][3]) * w1v.w;
        }
        // (g, r) pairs, 32 contiguous bytes per patch row: 16-byte stores (dword stores at a 32-byte lane stride would hit
        // every LDS bank eight times)
        float *d0 = lds + c * (2 * FU_PLANE) + ((pr * 2) * LP + pc * 4) * 2;
        float *d1 = d0 + 2 * LP;
        f4_t s0, s1, s2, s3;
        s0.x = g0v.x; s0.y = o0.x; s0.z = g0v.y; s0.w = o0.y; s1.x = g0v.z; s1.y = o0.z; s1.z = g0v.w; s1.w = o0.w;
        s2.x = g1v.x; s2.y = o1.x; s2.z = g1v.y; s2.w = o1.y; s3.x = g1v.z; s3.y = o1.z; s3.z = g1v.w; s3.w = o1.w;
        st_f4(d0, s0);
        st_f4(d0 + 4, s1);
        st_f4(d1, s2);
        st_f4(d1 + 4, s3);
    }
}

// pyrUp of a 3-row x 4-column neighbourhood of (g, r) PAIRS -> the thread's 4 x 2 pixels, both arrays at once in packed
// fp32 (interior form; up_regs with pairs).  q[r][i] is the pair at row r0 + r, column c0 + i.
template <bool XO, bool YO>
__device__ __forceinline__ void up_pairs(const f2_t (&q)[3][4], f2_t (&u)[2][4])
{
    f2_t h[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        if (!XO) {
            h[r][0] = (q[r][0] + q[r][1] * 6.0f) + q[r][2];
            h[r][1] = q[r][1] + q[r][2];
            h[r][2] = (q[r][1] + q[r][2] * 6.0f) + q[r][3];
            h[r][3] = q[r][2] + q[r][3];
        } else {
            h[r][0] = q[r][0] + q[r][1];
            h[r][1] = (q[r][0] + q[r][1] * 6.0f) + q[r][2];
            h[r][2] = q[r][1] + q[r][2];
            h[r][3] = (q[r][1] + q[r][2] * 6.0f) + q[r][3];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool kodd = XO ? ((k & 1) == 0) : ((k & 1) == 1);
        const float ce = kodd ? (1.0f / 16.0f) : (1.0f / 64.0f);
        const float co = kodd ? (1.0f / 4.0f) : (1.0f / 16.0f);
        const f2_t ev = ((h[0][k] + h[1][k] * 6.0f) + h[2][k]) * ce;
        if (!YO) {
            u[0][k] = ev;
            u[1][k] = (h[1][k] + h[2][k]) * co;
        } else {
            u[0][k] = (h[0][k] + h[1][k]) * co;
            u[1][k] = ev;
        }
    }
}

// Stage 2, interior visit of a regular block: the thread's 4 x 2 pixels from the LDS window (pitch FU_LP pairs).  CODD: the
// first tap column is odd -- its 32 bytes per row are then 8-byte aligned only and read as 8 + 16 + 8.
// the level-0 pixels of an interior 4 x 2 visit: requested BEFORE the block's stage 1 so they arrive under it
template <int DT, int CN>
struct CellPixels {
    float g0[(DT == SRC_U8 && CN == 3) ? 1 : 2][(DT == SRC_U8 && CN == 3) ? 1 : 4][CN];
    u3_t qs[2];
};

template <int DT, int CN>
__device__ __forceinline__ void fused_load_pixels(const FinalDesc &D, int lx0, int ly0, CellPixels<DT, CN> &px)
{
    constexpr bool LAZY = (DT == SRC_U8 && CN == 3);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const char *srow = (const char *)D.src + (size_t)(ly0 + j) * D.stride;
        if (LAZY) {
            px.qs[j] = ld_u3_a1_g(srow + (size_t)lx0 * 3);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int c = 0; c < CN; ++c) {
                    if (DT == SRC_U8) px.g0[j][k][c] = (float)((const unsigned char *)srow)[(lx0 + k) * CN + c];
                    else px.g0[j][k][c] = ((const float *)srow)[(lx0 + k) * CN + c];
                }
        }
    }
}

template <int DT, int CN, bool XO, bool YO, bool CODD>
__device__ __forceinline__ void fused_gather_fast(const FinalDesc &D, const float *__restrict__ luts, const float *lds, int LP, int R0,
                                                  int C0, int lx0, int ly0, const CellPixels<DT, CN> &cp, float (&acc)[2][4][CN],
                                                  float (&wacc)[2][4])
{
    const bool pyr = D.nl > 1;
    constexpr bool LAZY = (DT == SRC_U8 && CN == 3);
    float w0[2][4];
    tile_weights_interior(D, luts, lx0, ly0, w0);
    auto px = [&](int j, int k, int c) -> float {
        if (LAZY) {
            const int b = 3 * k + c;
            unsigned wd = (b >> 2) == 0 ? cp.qs[j].x : ((b >> 2) == 1 ? cp.qs[j].y : cp.qs[j].z);
            asm volatile("" : "+v"(wd));      // keeps the conversion at its use (see gather_tile_fast)
            return (float)((wd >> (8 * (b & 3))) & 0xFFu);
        }
        return cp.g0[LAZY ? 0 : j][LAZY ? 0 : k][c];
    };
    // beyond the feather width every weight of the cell is lut[fw]; where that is exactly 1 (linear and cosine ramps)
    // for the whole wave, lap * w == lap and the multiplies are skipped -- wave-uniform branch, bit-identical
    const bool flat = min(min(ly0, D.h - 2 - ly0), min(lx0, D.w - 4 - lx0)) >= D.fw && luts[D.lut_off + D.fw] == 1.0f;
    const bool unit_w = __all(flat) != 0;
    if (pyr) {
        const int r0 = (ly0 - 1) >> 1, c0 = (lx0 - 1) >> 1;
        // (g, r) pairs of rows r0 .. r0 + 2, columns c0 .. c0 + 3: 32 contiguous bytes per row
        const float *base = lds + ((r0 - R0) * LP + (c0 - C0)) * 2;
#pragma unroll
        for (int c = 0; c < CN; ++c) {
            const float *p = base + c * (2 * FU_PLANE);
            f2_t q[3][4], u[2][4];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float *pr = p + r * (2 * LP);
                if (!CODD) {
                    const f4_t a = *(const f4_t *)pr, b = *(const f4_t *)(pr + 4);
                    q[r][0].x = a.x; q[r][0].y = a.y; q[r][1].x = a.z; q[r][1].y = a.w;
                    q[r][2].x = b.x; q[r][2].y = b.y; q[r][3].x = b.z; q[r][3].y = b.w;
                } else {
                    const f2_t a = *(const f2_t *)pr, d = *(const f2_t *)(pr + 6);
                    const f4_t b = *(const f4_t *)(pr + 2);
                    q[r][0] = a; q[r][1].x = b.x; q[r][1].y = b.y; q[r][2].x = b.z; q[r][2].y = b.w; q[r][3] = d;
                }
            }
            up_pairs<XO, YO>(q, u);
            if (unit_w) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float lap = px(j, k, c) - u[j][k].x;
                        acc[j][k][c] += u[j][k].y + lap;
                    }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float lap = px(j, k, c) - u[j][k].x;
                        const float wl = lap * w0[j][k];
                        acc[j][k][c] += u[j][k].y + wl;
                    }
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < CN; ++c)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[j][k][c] += px(j, k, c) * w0[j][k];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) wacc[j][k] += w0[j][k];
}

// up_block over an LDS window: same clamps and border rules, the window's origin subtracted from the clamped indices
template <bool XO, bool YO>
__device__ __forceinline__ void up_block_win(const float *win, int hs, int ws, int LP, int R0, int C0, int r0, int c0, float (&u)[2][4])
{
    float v[3][4], h[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int ro = (min(max(r0 + r, 0), hs - 1) - R0) * LP - C0;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[r][c] = win[(ro + min(max(c0 + c, 0), ws - 1)) * 2];      // (g, r) pairs: stride 2
    }
    up_rows4<XO>(v, ws, c0, h);
    up_cols2<YO>(h, r0, u);
}

// any 4 x 2 visit from the LDS windows (gather_tile_generic with G_1 / R_1 in LDS)
template <int DT, int CN, bool XO, bool YO>
__device__ __forceinline__ void fused_gather_generic(const FinalDesc &D, const float *__restrict__ luts, const float *lds, int LP,
                                                     int R0, int C0, int lx0, int ly0, unsigned valid, float (&acc)[2][4][CN],
                                                     float (&wacc)[2][4])
{
    float w0[2][4];
    tile_weights(D, luts, lx0, ly0, w0);
    const int r0 = (ly0 - 1) >> 1, c0 = (lx0 - 1) >> 1;
    const bool pyr = D.nl > 1;
#pragma unroll
    for (int c = 0; c < CN; ++c) {
        float ug[2][4], ur[2][4];
        if (pyr) {
            up_block_win<XO, YO>(lds + c * (2 * FU_PLANE), D.H1, D.W1, LP, R0, C0, r0, c0, ug);
            up_block_win<XO, YO>(lds + c * (2 * FU_PLANE) + 1, D.H1, D.W1, LP, R0, C0, r0, c0, ur);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!(valid & (1u << (j * 4 + k)))) continue;
                const char *srow = (const char *)D.src + (size_t)(ly0 + j) * D.stride;
                float g0;
                if (DT == SRC_U8) g0 = (float)((const unsigned char *)srow)[(lx0 + k) * CN + c];
                else g0 = ((const float *)srow)[(lx0 + k) * CN + c];
                float r;
                if (pyr) {
                    const float lap = g0 - ug[j][k];
                    const float wl = lap * w0[j][k];
                    r = ur[j][k] + wl;
                } else {
                    r = g0 * w0[j][k];
                }
                acc[j][k][c] += r;
            }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (valid & (1u << (j * 4 + k))) wacc[j][k] += w0[j][k];
}

// Edge blocks of the fused gather: the 4 x 2 cells with a border visit, generic per-pixel rules from the LDS window.
template <int DT, int CN>
__device__ __forceinline__ void fused_edge_block(const FinalDesc *__restrict__ descs, const int4 *__restrict__ edge_blocks, int ebi,
                                                 const int *__restrict__ cand_idx, const float *__restrict__ arena,
                                                 const float *__restrict__ luts, float *lds, unsigned char *__restrict__ canvas,
                                                 long long cstride, float *__restrict__ canvas_f32, int cw, int row_begin, int row_end)
{
    const int4 eb = edge_blocks[ebi];
    const int c_begin = eb.w, c_end = edge_blocks[ebi + 1].w;
    const int tid = threadIdx.x;
    const int shape = eb.z;                                   // 0: 256 x FU_E0H px (64 cells across), 1: FU_E1W x FU_E1H px (4 across)
    const int bw = shape ? FU_E1W : 256, bh = shape ? FU_E1H : FU_E0H, LP = shape ? FU_E1LP : 136;
    const int x0 = eb.x + (shape ? (tid & 3) : (tid & 63)) * 4;
    const int y0 = eb.y + (shape ? (tid >> 2) : (tid >> 6)) * 2;
    const bool inside = x0 < cw && y0 < row_end;
    const int nx = min(4, cw - x0), ny = min(2, row_end - y0);
    bool edge = false;
    if (inside)
        for (int i = c_begin; i < c_end; ++i) {
            const FinalDesc &D = descs[cand_idx[i]];
            const int lx0 = x0 - D.x, ly0 = y0 - D.y;
            if (lx0 + nx <= 0 || ly0 + ny <= 0 || lx0 >= D.w || ly0 >= D.h) continue;
            if (!visit_is_interior<true>(D, lx0, ly0, nx, ny)) edge = true;
        }
    float acc[2][4][CN], wacc[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            wacc[j][k] = 0.f;
#pragma unroll
            for (int c = 0; c < CN; ++c) acc[j][k][c] = 0.f;
        }
    float *const lds_all = lds;
    for (int i = c_begin; i < c_end; ++i) {
        const FinalDesc &D = descs[cand_idx[i]];
        if (FU_DB) lds = lds_all + ((i - c_begin) & 1) * (2 * CN * FU_PLANE);
        int R0 = 0, C0 = 0, npr = 0, npc = 0;
        const bool win = D.nl > 1 && fused_window(D, eb.x - D.x, eb.y - D.y, bw, bh, R0, C0, npr, npc);
        if (win) fused_stage1<CN>(D, arena, lds, R0, C0, npr, npc, LP, tid);
        __syncthreads();
        const int lx0 = x0 - D.x, ly0 = y0 - D.y;
        const bool touches = edge && !(lx0 + nx <= 0 || ly0 + ny <= 0 || lx0 >= D.w || ly0 >= D.h);
        // An edge cell has a border visit of SOME tile; its visits of the other tiles are mostly interior ones (the cell
        // lies on tile A's edge and well inside tile B): those take the regular blocks' packed path (same arithmetic, same
        // results), only the border visits pay for the per-pixel border rules.  Along a tile edge the split is uniform per
        // tile, so a wave rarely runs both.
        const bool inner = DT == SRC_U8 && touches && visit_is_interior<true>(D, lx0, ly0, nx, ny);    // (float tiles: generic path only)
        if constexpr (DT == SRC_U8) if (inner) {
            CellPixels<DT, CN> cp;
            fused_load_pixels<DT, CN>(D, lx0, ly0, cp);
            const bool xo = (D.x & 1) != 0;
            const bool yo = ((row_begin - D.y) & 1) != 0;
            const bool codd = D.nl > 1 && (((((eb.x - D.x) - 1) >> 1) - C0) & 1) != 0;
#define FE_CALL(XOV, YOV, CV) fused_gather_fast<DT, CN, XOV, YOV, CV>(D, luts, lds, LP, R0, C0, lx0, ly0, cp, acc, wacc)
            if (!codd) {
                if (!xo && !yo) FE_CALL(false, false, false);
                else if (xo && !yo) FE_CALL(true, false, false);
                else if (!xo && yo) FE_CALL(false, true, false);
                else FE_CALL(true, true, false);
            } else {
                if (!xo && !yo) FE_CALL(false, false, true);
                else if (xo && !yo) FE_CALL(true, false, true);
                else if (!xo && yo) FE_CALL(false, true, true);
                else FE_CALL(true, true, true);
            }
#undef FE_CALL
        }
        if (touches && !inner) {
            unsigned valid = 0;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (j < ny && k < nx && lx0 + k >= 0 && lx0 + k < D.w && ly0 + j >= 0 && ly0 + j < D.h)
                        valid |= 1u << (j * 4 + k);
            const bool xo = (D.x & 1) != 0;
            const bool yo = ((row_begin - D.y) & 1) != 0;
            if (!xo && !yo) fused_gather_generic<DT, CN, false, false>(D, luts, lds, LP, R0, C0, lx0, ly0, valid, acc, wacc);
            else if (xo && !yo) fused_gather_generic<DT, CN, true, false>(D, luts, lds, LP, R0, C0, lx0, ly0, valid, acc, wacc);
            else if (!xo && yo) fused_gather_generic<DT, CN, false, true>(D, luts, lds, LP, R0, C0, lx0, ly0, valid, acc, wacc);
            else fused_gather_generic<DT, CN, true, true>(D, luts, lds, LP, R0, C0, lx0, ly0, valid, acc, wacc);
        }
        if (!FU_DB) __syncthreads();                       // the next tile's stage 1 overwrites the window
    }
    if (edge) store_pixels<CN>(acc, wacc, canvas, cstride, canvas_f32, cw, x0, y0, nx, ny);
}

template <int DT, int CN>
__global__ __launch_bounds__(FU_THREADS, FU_WAVES) void k_final_fused(const FinalDesc *__restrict__ descs, const int *__restrict__ cand_off,
                                                        const int *__restrict__ cand_idx, const int4 *__restrict__ edge_blocks,
                                                        const int *__restrict__ edge_cand, int n_edge, int nbx_r,
                                                        const int *__restrict__ reg_list,
                                                        const float *__restrict__ arena, const float *__restrict__ luts,
                                                        unsigned char *__restrict__ canvas, long long cstride,
                                                        float *__restrict__ canvas_f32, int cw, int row_begin, int row_end)
{
    // FU_DB: two windows, taken in turn by the tiles of a block -- the barrier that kept a tile's stage 1 from overwriting the window
    // the previous tile's gather still reads is not needed then (one barrier per tile instead of two, 52 instead of 26 KB of LDS)
    __shared__ __attribute__((aligned(16))) float lds_all[(FU_DB ? 2 : 1) * 2 * CN * FU_PLANE];
    float *lds = lds_all;
    if ((int)blockIdx.x < n_edge) {
        fused_edge_block<DT, CN>(descs, edge_blocks, (int)blockIdx.x, edge_cand, arena, luts, lds, canvas, cstride, canvas_f32, cw,
                                 row_begin, row_end);
        return;
    }
    // (Measured and rejected: a workgroup marching down several blocks of a column -- one dispatch, halo rows still in
    // the CU's caches -- is slower, 1.37 -> 1.52 ms at 16 blocks: the blocks of a march run strictly one after the other
    // and each is a chain of dependent memory round trips; independent blocks overlap them.)
    // the regular blocks that are left once the marched zones (sr_march.inc) are taken out: block id and, per cell row of
    // the block, the mask of the cell columns a march item covers
    const int *rl = reg_list + 9 * ((int)blockIdx.x - n_edge);
    const int blk = rl[0];
    const int by = blk / nbx_r, bx = blk - by * nbx_r;
    const int c_begin = cand_off[blk], c_end = cand_off[blk + 1];
    const int tid = threadIdx.x;
    const int bx0 = bx * FU_BW, by0 = row_begin + by * FU_BH;
    // the cells that are left span a sub-rectangle of the block (thin bands along the marched zones, mostly): the level-1
    // window is built for that rectangle only, and the threads are dealt over ITS cells (a band of 3 x 8 cells is one
    // wave's work, not a few lanes of each of the four) -- block-uniform: scalar arithmetic on the mask words
    int sbx = bx0, sby = by0, sbw = FU_BW, sbh = FU_BH;
    {
        const int ncol = min(32, (cw - bx0 + 3) >> 2), nrow = min(FU_BH / 2, (row_end - by0 + 1) >> 1);
        const unsigned colmask = ncol >= 32 ? 0xFFFFFFFFu : ((1u << ncol) - 1u);
        unsigned any = 0u;
        int cy0 = FU_BH / 2, cy1 = -1;
#pragma unroll
        for (int cy = 0; cy < FU_BH / 2; ++cy) {
            const unsigned a = cy < nrow ? (~(unsigned)rl[1 + cy] & colmask) : 0u;
            if (a) {
                any |= a;
                cy0 = min(cy0, cy);
                cy1 = cy;
            }
        }
        if (any) {
            const int cx0 = __builtin_ctz(any), cx1 = 31 - __builtin_clz(any);
            sbx = bx0 + 4 * cx0;
            sby = by0 + 2 * cy0;
            sbw = 4 * (cx1 - cx0 + 1);
            sbh = 2 * (cy1 - cy0 + 1);
        }
    }
    const int scw = sbw >> 2;                                        // cells across the sub-rectangle
    const int scy = tid / scw, scx = tid - scy * scw;
    const int x0 = sbx + scx * 4, y0 = sby + scy * 2;
    const int nx = min(4, cw - x0), ny = min(2, row_end - y0);
    const int mrow = (y0 - by0) >> 1, mcol = (x0 - bx0) >> 2;         // the cell's place in the block's mask
    // a cell with any border visit belongs to the edge blocks (which recompute every visit of it): it stops
    // accumulating at its first border visit and stores nothing
    bool alive = scy < (sbh >> 1) && x0 < cw && y0 < row_end && !(((unsigned)rl[1 + min(mrow, FU_BH / 2 - 1)] >> mcol) & 1u);
    float acc[2][4][CN], wacc[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            wacc[j][k] = 0.f;
#pragma unroll
            for (int c = 0; c < CN; ++c) acc[j][k][c] = 0.f;
        }
    for (int i = c_begin; i < c_end; ++i) {
        const FinalDesc &D = descs[cand_idx[i]];
        if (FU_DB) lds = lds_all + ((i - c_begin) & 1) * (2 * CN * FU_PLANE);
        const int lxa = sbx - D.x, lya = sby - D.y;
        int R0 = 0, C0 = 0, npr = 0, npc = 0;
        const bool win = D.nl > 1 && fused_window(D, lxa, lya, sbw, sbh, R0, C0, npr, npc);        // block-uniform
        const int lx0 = x0 - D.x, ly0 = y0 - D.y;
        bool visit = alive && !(lx0 + nx <= 0 || ly0 + ny <= 0 || lx0 >= D.w || ly0 >= D.h);
        if (visit && !visit_is_interior<true>(D, lx0, ly0, nx, ny)) visit = alive = false;
        CellPixels<DT, CN> cp;
        if (visit) fused_load_pixels<DT, CN>(D, lx0, ly0, cp);                                     // in flight during stage 1
        if (win) fused_stage1<CN>(D, arena, lds, R0, C0, npr, npc, FU_LP, tid);
        __syncthreads();
        if (visit) {
            const bool xo = (D.x & 1) != 0;                      // x0 is a multiple of 4
            const bool yo = ((row_begin - D.y) & 1) != 0;        // y0 - row_begin is a multiple of 2
            const bool codd = D.nl > 1 && ((((lxa - 1) >> 1) - C0) & 1) != 0;   // parity of every cell's first tap column (block-uniform)
#define FU_CALL(XOV, YOV, CV) fused_gather_fast<DT, CN, XOV, YOV, CV>(D, luts, lds, FU_LP, R0, C0, lx0, ly0, cp, acc, wacc)
            if (!codd) {
                if (!xo && !yo) FU_CALL(false, false, false);
                else if (xo && !yo) FU_CALL(true, false, false);
                else if (!xo && yo) FU_CALL(false, true, false);
                else FU_CALL(true, true, false);
            } else {
                if (!xo && !yo) FU_CALL(false, false, true);
                else if (xo && !yo) FU_CALL(true, false, true);
                else if (!xo && yo) FU_CALL(false, true, true);
                else FU_CALL(true, true, true);
            }
#undef FU_CALL
        }
        if (!FU_DB) __syncthreads();                           // the next tile's stage 1 overwrites the window
    }
    if (alive) store_pixels<CN>(acc, wacc, canvas, cstride, canvas_f32, cw, x0, y0, nx, ny);   // ragged cells without a visit: zeros
}

// ---------------------------------------------------------------------------------------------
// Round 4: what the marched zones leave -- thin bands along the tiles' edges, 1.8 % of the cells of the 200 MP grid -- as
// RECTANGLES of cells (k_final_rect).  The 128 x 16 blocks above cut a vertical band 3-5 cells wide into 722 blocks per band
// with 30-40 of their 256 cells in use, and the cells with a border visit were visited a second time by the edge blocks
// (window built twice): 9 192 + 2 700 blocks, 12 rounds of dependent round trips, 0.21 ms for 3.5 MP.  A rectangle item is
// w x h cells with w * h <= 256 and a level-1 window that fits the LDS planes (a band 5 cells wide: 5 x 51 cells, a band 6
// rows high: 32 x 6); every cell of it is finished here, whatever its visits are -- interior visits through the packed
// path, border visits through the per-pixel rules (fused_edge_block's two paths; the same expressions, bit-identical) -- so
// the edge blocks are not launched at all beside a march.  Threads are dealt column-major over tall rectangles: along a
// vertical tile edge the kind of visit is then uniform per wave.
// ---------------------------------------------------------------------------------------------
struct RectItem {                    // 32 bytes, block-uniform
    int x, y;                        // canvas pixel of the first cell (y includes row_begin)
    int w, h;                        // cells across / down
    int cand, ncand;                 // candidate tiles: rect_cand[cand .. cand + ncand), list order
    int lp;                          // LDS pitch of the level-1 window (floats pairs per row)
    int colmajor;                    // 1: thread -> (column, row) with rows fastest
};
static_assert(sizeof(RectItem) == 32, "RectItem layout");

// window columns / rows fused_window can ask for, for a rectangle of w x h cells (host and static checks)
static inline int rect_lp(int w) { return 4 * ((2 * w + 4) / 4) + 4; }
static inline int rect_rows(int h) { return 2 * ((h + 2) / 2) + 2; }

#ifndef FR_WAVES
#define FR_WAVES FU_WAVES     /* waves per SIMD the register allocation of k_final_rect is held to */
#endif
template <int CN>
__global__ __launch_bounds__(FU_THREADS, FR_WAVES) void k_final_rect(const FinalDesc *__restrict__ descs, const RectItem *__restrict__ rects,
                                                        const int *__restrict__ rect_cand, const float *__restrict__ arena,
                                                        const float *__restrict__ luts, unsigned char *__restrict__ canvas,
                                                        long long cstride, float *__restrict__ canvas_f32, int cw, int row_begin,
                                                        int row_end)
{
    constexpr int DT = SRC_U8;
    __shared__ __attribute__((aligned(16))) float lds[2 * CN * FU_PLANE];
    const RectItem it = rects[blockIdx.x];
    const int tid = threadIdx.x;
    int scx, scy;
    if (it.colmajor) {
        scx = tid / it.h;
        scy = tid - scx * it.h;
    } else {
        scy = tid / it.w;
        scx = tid - scy * it.w;
    }
    const int x0 = it.x + 4 * scx, y0 = it.y + 2 * scy;
    const bool inside = scx < it.w && scy < it.h && x0 < cw && y0 < row_end;
    const int nx = min(4, cw - x0), ny = min(2, row_end - y0);
    const int LP = it.lp;
    float acc[2][4][CN], wacc[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            wacc[j][k] = 0.f;
#pragma unroll
            for (int c = 0; c < CN; ++c) acc[j][k][c] = 0.f;
        }
    for (int i = it.cand; i < it.cand + it.ncand; ++i) {
        const FinalDesc &D = descs[rect_cand[i]];
        const int lxa = it.x - D.x, lya = it.y - D.y;
        int R0 = 0, C0 = 0, npr = 0, npc = 0;
        const bool win = D.nl > 1 && fused_window(D, lxa, lya, 4 * it.w, 2 * it.h, R0, C0, npr, npc);     // block-uniform
        const int lx0 = x0 - D.x, ly0 = y0 - D.y;
        const bool touches = inside && !(lx0 + nx <= 0 || ly0 + ny <= 0 || lx0 >= D.w || ly0 >= D.h);
        const bool inner = touches && visit_is_interior<true>(D, lx0, ly0, nx, ny);
        CellPixels<DT, CN> cp;
        if (inner) fused_load_pixels<DT, CN>(D, lx0, ly0, cp);                                        // in flight during stage 1
        if (win) fused_stage1<CN>(D, arena, lds, R0, C0, npr, npc, LP, tid);
        __syncthreads();
        const bool xo = (D.x & 1) != 0;                      // x0 is a multiple of 4
        const bool yo = ((row_begin - D.y) & 1) != 0;        // y0 - row_begin is a multiple of 2
        if (inner) {
            const bool codd = D.nl > 1 && ((((lxa - 1) >> 1) - C0) & 1) != 0;   // parity of every cell's first tap column (block-uniform)
#define FR_CALL(XOV, YOV, CV) fused_gather_fast<DT, CN, XOV, YOV, CV>(D, luts, lds, LP, R0, C0, lx0, ly0, cp, acc, wacc)
            if (!codd) {
                if (!xo && !yo) FR_CALL(false, false, false);
                else if (xo && !yo) FR_CALL(true, false, false);
                else if (!xo && yo) FR_CALL(false, true, false);
                else FR_CALL(true, true, false);
            } else {
                if (!xo && !yo) FR_CALL(false, false, true);
                else if (xo && !yo) FR_CALL(true, false, true);
                else if (!xo && yo) FR_CALL(false, true, true);
                else FR_CALL(true, true, true);
            }
#undef FR_CALL
        }
        if (touches && !inner) {
            unsigned valid = 0;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (j < ny && k < nx && lx0 + k >= 0 && lx0 + k < D.w && ly0 + j >= 0 && ly0 + j < D.h)
                        valid |= 1u << (j * 4 + k);
            if (!xo && !yo) fused_gather_generic<DT, CN, false, false>(D, luts, lds, LP, R0, C0, lx0, ly0, valid, acc, wacc);
            else if (xo && !yo) fused_gather_generic<DT, CN, true, false>(D, luts, lds, LP, R0, C0, lx0, ly0, valid, acc, wacc);
            else if (!xo && yo) fused_gather_generic<DT, CN, false, true>(D, luts, lds, LP, R0, C0, lx0, ly0, valid, acc, wacc);
            else fused_gather_generic<DT, CN, true, true>(D, luts, lds, LP, R0, C0, lx0, ly0, valid, acc, wacc);
        }
        __syncthreads();                                       // the next tile's stage 1 overwrites the window
    }
    if (inside) store_pixels<CN>(acc, wacc, canvas, cstride, canvas_f32, cw, x0, y0, nx, ny);   // cells without a visit: zeros
}

#include "sr_march.inc"
#include "sr_down2.inc"

// ---------------------------------------------------------------------------------------------
// dense HWC pyramid primitives (API utilities for build_gaussian_pyramid & friends)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pyr_down_hwc(const float *__restrict__ src, int h, int w, int cn,
                                                      float *__restrict__ dst, int ho, int wo)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= wo || y >= ho) return;
    int xi[5], yi[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        xi[k] = reflect101(2 * x + k - 2, w);
        yi[k] = reflect101(2 * y + k - 2, h);
    }
    for (int c = 0; c < cn; ++c) {
        float rowv[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const float *r = src + (size_t)yi[k] * w * cn + c;
            rowv[k] = ((r[xi[2] * cn] * 6.0f + (r[xi[1] * cn] + r[xi[3] * cn]) * 4.0f) + r[xi[0] * cn]) + r[xi[4] * cn];
        }
        const float v = ((rowv[2] * 6.0f + (rowv[1] + rowv[3]) * 4.0f) + rowv[0]) + rowv[4];
        dst[((size_t)y * wo + x) * cn + c] = v * (1.0f / 256.0f);
    }
}

__device__ __forceinline__ float up_h_hwc(const float *__restrict__ row, int ws, int cn, int x)
{
    const int sx = x >> 1;
    if (ws == 1) return (x & 1) ? row[0] * 8.0f : row[0] * 6.0f + row[0] * 2.0f;
    if (!(x & 1)) {
        if (sx == 0) return row[0] * 6.0f + row[cn] * 2.0f;
        if (sx == ws - 1) return row[(sx - 1) * cn] + row[sx * cn] * 7.0f;
        return (row[(sx - 1) * cn] + row[sx * cn] * 6.0f) + row[(sx + 1) * cn];
    }
    if (sx == ws - 1) return row[sx * cn] * 8.0f;
    return (row[sx * cn] + row[(sx + 1) * cn]) * 4.0f;
}

// MODE 0: dst = up(src); 1: dst = a - up(src); 2: dst = up(src) + a
template <int MODE>
__global__ __launch_bounds__(256) void k_pyr_up_hwc(const float *__restrict__ src, int hs, int ws, int cn,
                                                    const float *__restrict__ a, float *__restrict__ dst, int hd,
                                                    int wd)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= wd || y >= hd) return;
    const int sy = y >> 1;
    const int yp = min(sy + 1, hs - 1);
    const int ym = (sy - 1 < 0) ? (hs > 1 ? 1 : 0) : sy - 1;
    for (int c = 0; c < cn; ++c) {
        const float r1 = up_h_hwc(src + (size_t)sy * ws * cn + c, ws, cn, x);
        const float r2 = up_h_hwc(src + (size_t)yp * ws * cn + c, ws, cn, x);
        float u;
        if (!(y & 1)) {
            const float r0 = up_h_hwc(src + (size_t)ym * ws * cn + c, ws, cn, x);
            u = ((r0 + r1 * 6.0f) + r2) * (1.0f / 64.0f);
        } else {
            u = ((r1 + r2) * 4.0f) * (1.0f / 64.0f);
        }
        const size_t o = ((size_t)y * wd + x) * cn + c;
        if (MODE == 0) dst[o] = u;
        else if (MODE == 1) dst[o] = a[o] - u;
        else dst[o] = u + a[o];
    }
}

// ---------------------------------------------------------------------------------------------
// tile extract
// ---------------------------------------------------------------------------------------------
struct ExtractDesc {
    int x, y, w, h;
    unsigned char *dst;
    long long dstride;
    int out_w, out_h;
};


// One thread = 16 consecutive bytes of one output row.  Inside the source rectangle that is a straight
// copy: one byte-aligned 16-byte load (the source offset x*cn is arbitrary) and one dword-aligned store;
// bytes in the padded band (and ragged tails) take the per-byte border rule.
__global__ __launch_bounds__(256) void k_tile_extract(const unsigned char *__restrict__ img, long long istride,
                                                      int cn, const ExtractDesc *__restrict__ descs, int pad_mode)
{
    SR_CHAIN_SETPRIO_BIG();
    const ExtractDesc D = descs[blockIdx.z];
    const int r = blockIdx.y * 4 + threadIdx.y;
    const long long b0 = ((long long)blockIdx.x * 64 + threadIdx.x) * 16;
    const long long row_bytes = (long long)D.out_w * cn;
    if (r >= D.out_h || b0 >= row_bytes) return;
    unsigned char *d = D.dst + (size_t)r * D.dstride + b0;
    if (r < D.h && b0 + 16 <= (long long)D.w * cn) {
        // 16 bytes at any source / destination alignment (tile x and width are arbitrary): the hardware splits an
        // unaligned access; a dword-aligned destination row gets the aligned store
        const unsigned char *sp = img + (size_t)(D.y + r) * istride + (size_t)D.x * cn + b0;
        const u4_t v = *(const u4_a1_t *)sp;
        if (((((size_t)D.dst) | (size_t)D.dstride) & 3) == 0) *(__attribute__((address_space(1))) u4_a4_t *)d = v;
        else *(__attribute__((address_space(1))) u4_a1_t *)d = v;
        return;
    }
    const int nb = (int)min((long long)16, row_bytes - b0);
    for (int i = 0; i < nb; ++i) {
        const long long bb = b0 + i;
        const int c = (int)(bb / cn), k = (int)(bb - (long long)c * cn);
        if (pad_mode == PAD_CONSTANT && (r >= D.h || c >= D.w)) {
            d[i] = 0;
            continue;
        }
        const int sr = border_index(r, D.h, pad_mode), sc = border_index(c, D.w, pad_mode);
        d[i] = img[(size_t)(D.y + sr) * istride + (size_t)(D.x + sc) * cn + k];
    }
}

// ---------------------------------------------------------------------------------------------
// metrics
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// contiguous fast path: both buffers dense and 16-byte aligned
__global__ __launch_bounds__(256) void k_sse_flat(const uint4 *__restrict__ a, const uint4 *__restrict__ b,
                                                  size_t nvec, const unsigned char *__restrict__ ta,
                                                  const unsigned char *__restrict__ tb, int ntail,
                                                  unsigned long long *__restrict__ out)
{
    unsigned long long s = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const uint4 va = a[i], vb = b[i];
        const unsigned int wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
        unsigned int p = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = (int)((wa[k] >> (8 * j)) & 0xFF) - (int)((wb[k] >> (8 * j)) & 0xFF);
                p += (unsigned int)(d * d);
            }
        s += p;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) {
        const int d = (int)ta[threadIdx.x] - (int)tb[threadIdx.x];
        s += (unsigned int)(d * d);
    }
    s = wave_sum_u64(s);
    __shared__ unsigned long long ws[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) ws[wid] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, ws[0] + ws[1] + ws[2] + ws[3]);
}

// strided path (cropped / non-dense images): one block row-chunk, byte loads
__global__ __launch_bounds__(256) void k_sse_rows(const unsigned char *__restrict__ a, long long sa,
                                                  const unsigned char *__restrict__ b, long long sb, int h,
                                                  long long rowlen, unsigned long long *__restrict__ out)
{
    unsigned long long s = 0;
    for (int y = blockIdx.y; y < h; y += gridDim.y) {
        const unsigned char *pa = a + (size_t)y * sa, *pb = b + (size_t)y * sb;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < rowlen;
             i += (long long)gridDim.x * blockDim.x) {
            const int d = (int)pa[i] - (int)pb[i];
            s += (unsigned int)(d * d);
        }
    }
    s = wave_sum_u64(s);
    __shared__ unsigned long long ws[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) ws[wid] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, ws[0] + ws[1] + ws[2] + ws[3]);
}

__device__ __forceinline__ int gray_rgb(int r, int g, int b, int shift)
{
    return shift == 15 ? (r * 9798 + g * 19235 + b * 3735 + (1 << 14)) >> 15
                       : (r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14;
}

// detect_seams (blending_module.py:765-853): one thread = one window of one tile.  Gray values (BGR2GRAY applied to
// RGB data, i.e. swapped R/B weights, as the reference does) are integers, so the five window sums are exact; the
// global-statistics SSIM of the window is finished in fp64 and windows below the threshold are appended.
struct SeamTile {
    const unsigned char *p;
    long long stride;
    int x, y, w, h;          // canvas position, size
    int roi_w, roi_h;        // part inside the canvas
    int nwx, nwy;            // windows per row / column
    long long first;         // index of this tile's first window in the flat window numbering
    long long bfirst;        // k_seam_scan_cells: index of this tile's first block, and its blocks per block row
    int nbx, pad;
};
struct SeamRec {
    int tile, x, y, pad;
    double score;
};

__global__ __launch_bounds__(256) void k_seam_scan(const unsigned char *__restrict__ canvas, long long cstride, int cn,
                                                   const SeamTile *__restrict__ tiles, int ntiles, long long nwin,
                                                   int window, int stride, int shift, double threshold, double c1,
                                                   double c2, SeamRec *__restrict__ out, int cap, int *__restrict__ count)
{
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nwin) return;
    int t = 0;
    while (t + 1 < ntiles && tiles[t + 1].first <= gid) ++t;
    const SeamTile T = tiles[t];
    const long long local = gid - T.first;
    const int wy = (int)(local / T.nwx), wx = (int)(local - (long long)wy * T.nwx);
    const int x0 = wx * stride, y0 = wy * stride;
    long long sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
    for (int r = 0; r < window; ++r) {
        const unsigned char *pt = T.p + (size_t)(y0 + r) * T.stride + (size_t)x0 * cn;
        const unsigned char *pc = canvas + (size_t)(T.y + y0 + r) * cstride + (size_t)(T.x + x0) * cn;
        for (int c = 0; c < window; ++c) {
            int a, b;
            if (cn == 1) {
                a = pt[c];
                b = pc[c];
            } else {      // BGR2GRAY on RGB data: first channel gets the blue weight
                a = gray_rgb(pt[3 * c + 2], pt[3 * c + 1], pt[3 * c], shift);
                b = gray_rgb(pc[3 * c + 2], pc[3 * c + 1], pc[3 * c], shift);
            }
            sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
        }
    }
    const double n = (double)window * (double)window;
    const double mu1 = (double)sx / n, mu2 = (double)sy / n;
    const double s1 = (double)sxx / n - mu1 * mu1, s2 = (double)syy / n - mu2 * mu2, s12 = (double)sxy / n - mu1 * mu2;
    const double score = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s1 + s2 + c2));
    if (score < threshold) {
        const int k = atomicAdd(count, 1);
        if (k < cap) {
            SeamRec rec;
            rec.tile = t; rec.x = T.x + x0; rec.y = T.y + y0; rec.pad = 0; rec.score = score;
            out[k] = rec;
        }
    }
}

// The default geometry (16 x 16 windows every 8 pixels: window_size 16, stride window_size // 2, blending_module.py:765-903)
// without the 4x redundancy of one thread per window: a window is 2 x 2 cells of 8 x 8 pixels.  One thread = one cell (gray of
// the 64 pixels of tile and canvas, five 32-bit sums: 64 x 255^2 fits), cells of a 32 x 8 block meet in LDS, then one thread
// = one window (four cells: 256 x 255^2 still fits 32 bits) and the reference's formula in fp64.  Same integers, same
// formula: identical scores.  3.3 -> 0.4 ms for the 4.75 M windows of the 200 MP workload.
#define SEAM_CX 32
#define SEAM_CY 8
template <int CN>
__global__ __launch_bounds__(256) void k_seam_scan_cells(const unsigned char *__restrict__ canvas, long long cstride,
                                                         const SeamTile *__restrict__ tiles, int ntiles, int shift, double threshold,
                                                         double c1, double c2, SeamRec *__restrict__ out, int cap, int *__restrict__ count)
{
    __shared__ unsigned cell[5][SEAM_CY][SEAM_CX + 1];
    int t = 0;
    while (t + 1 < ntiles && tiles[t + 1].bfirst <= (long long)blockIdx.x) ++t;
    const SeamTile T = tiles[t];
    const int lb = (int)((long long)blockIdx.x - T.bfirst), by = lb / T.nbx, bx = lb - by * T.nbx;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int cx = bx * (SEAM_CX - 1) + tx, cy = by * (SEAM_CY - 1) + ty;
    unsigned sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
    if (cx <= T.nwx && cy <= T.nwy) {                           // nwx + 1 cells per row: the last window ends at 8 (nwx + 1)
#pragma unroll 2
        for (int r = 0; r < 8; ++r) {
            const unsigned char *pt = T.p + (size_t)(8 * cy + r) * T.stride + (size_t)(8 * cx) * CN;
            const unsigned char *pc = canvas + (size_t)(T.y + 8 * cy + r) * cstride + (size_t)(T.x + 8 * cx) * CN;
            unsigned wa[6], wb[6];
            if (CN == 3) {
                const u3_t a0 = ld_u3_a1_g(pt), a1 = ld_u3_a1_g(pt + 12), b0 = ld_u3_a1_g(pc), b1 = ld_u3_a1_g(pc + 12);
                wa[0] = a0.x; wa[1] = a0.y; wa[2] = a0.z; wa[3] = a1.x; wa[4] = a1.y; wa[5] = a1.z;
                wb[0] = b0.x; wb[1] = b0.y; wb[2] = b0.z; wb[3] = b1.x; wb[4] = b1.y; wb[5] = b1.z;
            } else {
                wa[0] = *(const u1_a1_t *)pt; wa[1] = *(const u1_a1_t *)(pt + 4);
                wb[0] = *(const u1_a1_t *)pc; wb[1] = *(const u1_a1_t *)(pc + 4);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int a, b;
                if (CN == 3) {                                  // BGR2GRAY on RGB data: first channel gets the blue weight
                    const int i0 = 3 * k, i1 = 3 * k + 1, i2 = 3 * k + 2;
                    a = gray_rgb((int)((wa[i2 >> 2] >> (8 * (i2 & 3))) & 0xFFu), (int)((wa[i1 >> 2] >> (8 * (i1 & 3))) & 0xFFu),
                                 (int)((wa[i0 >> 2] >> (8 * (i0 & 3))) & 0xFFu), shift);
                    b = gray_rgb((int)((wb[i2 >> 2] >> (8 * (i2 & 3))) & 0xFFu), (int)((wb[i1 >> 2] >> (8 * (i1 & 3))) & 0xFFu),
                                 (int)((wb[i0 >> 2] >> (8 * (i0 & 3))) & 0xFFu), shift);
                } else {
                    a = (int)((wa[k >> 2] >> (8 * (k & 3))) & 0xFFu);
                    b = (int)((wb[k >> 2] >> (8 * (k & 3))) & 0xFFu);
                }
                sx += (unsigned)a; sy += (unsigned)b;
                sxx += (unsigned)__mul24(a, a); syy += (unsigned)__mul24(b, b); sxy += (unsigned)__mul24(a, b);
            }
        }
    }
    cell[0][ty][tx] = sx; cell[1][ty][tx] = sy; cell[2][ty][tx] = sxx; cell[3][ty][tx] = syy; cell[4][ty][tx] = sxy;
    __syncthreads();
    if (tx >= SEAM_CX - 1 || ty >= SEAM_CY - 1 || cx >= T.nwx || cy >= T.nwy) return;
    unsigned w[5];
#pragma unroll
    for (int m = 0; m < 5; ++m) w[m] = (cell[m][ty][tx] + cell[m][ty][tx + 1]) + (cell[m][ty + 1][tx] + cell[m][ty + 1][tx + 1]);
    const double n = 256.0;
    const double mu1 = (double)w[0] / n, mu2 = (double)w[1] / n;
    const double s1 = (double)w[2] / n - mu1 * mu1, s2 = (double)w[3] / n - mu2 * mu2, s12 = (double)w[4] / n - mu1 * mu2;
    const double score = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s1 + s2 + c2));
    if (score < threshold) {
        const int k = atomicAdd(count, 1);
        if (k < cap) {
            SeamRec rec;
            rec.tile = t; rec.x = T.x + 8 * cx; rec.y = T.y + 8 * cy; rec.pad = 0; rec.score = score;
            out[k] = rec;
        }
    }
}

// squared differences of two fp32 images (skimage's PSNR on float input: fp32 difference and square, fp64 mean)
__global__ __launch_bounds__(256) void k_sse_f32(const float *__restrict__ a, long long sa, const float *__restrict__ b,
                                                 long long sb, int h, long long rowlen, double *__restrict__ part)
{
    double s = 0.0;
    for (int y = blockIdx.y; y < h; y += gridDim.y) {
        const float *pa = (const float *)((const char *)a + (size_t)y * sa);
        const float *pb = (const float *)((const char *)b + (size_t)y * sb);
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < rowlen; i += (long long)gridDim.x * blockDim.x) {
            const float d = pa[i] - pb[i];
            s += (double)(d * d);
        }
    }
    s = wave_sum_f64(s);
    __shared__ double ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = ((ws[0] + ws[1]) + ws[2]) + ws[3];
}

// weighted_average_fusion with caller-supplied weight maps (blending_module.py:729-751): thread per canvas pixel
struct CustomW {
    const float *w;      // h x w fp32 weight map of the tile, row stride in bytes
    long long stride;
};
template <int DT>
__global__ __launch_bounds__(256) void k_weighted_custom(const TileDev *__restrict__ tiles, const TileSrc *__restrict__ srcs,
                                                         const CustomW *__restrict__ wts, int n, int cn,
                                                         unsigned char *__restrict__ canvas, long long cstride,
                                                         float *__restrict__ canvas_f32, int cw, int row_begin, int row_end)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = row_begin + blockIdx.y * 4 + threadIdx.y;
    if (x >= cw || y >= row_end) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float wacc = 0.f;
    for (int t = 0; t < n; ++t) {
        const TileDev &T = tiles[t];
        const int lx = x - T.x, ly = y - T.y;
        if (lx < 0 || ly < 0 || lx >= T.w || ly >= T.h) continue;
        const float w0 = ((const float *)((const char *)wts[t].w + (size_t)ly * wts[t].stride))[lx];
        const char *srow = (const char *)srcs[t].p + (size_t)ly * srcs[t].stride;
        for (int c = 0; c < cn; ++c) {
            const float g0 = DT == SRC_U8 ? (float)((const unsigned char *)srow)[lx * cn + c] : ((const float *)srow)[lx * cn + c];
            acc[c] += g0 * w0;
        }
        wacc += w0;
    }
    const float wv = wacc > 1e-6f ? wacc : 1e-6f;
    unsigned char *o = canvas + (size_t)y * cstride + (size_t)x * cn;
    for (int c = 0; c < cn; ++c) {
        const float v = acc[c] / wv;
        if (canvas_f32) canvas_f32[((size_t)y * cw + x) * cn + c] = v;
        const float cl = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
        o[c] = (unsigned char)cl;
    }
}

__device__ __forceinline__ int gray_of(const unsigned char *__restrict__ p, int cn, int shift)
{
    if (cn == 1) return p[0];
    const int r = p[0], g = p[1], b = p[2];
    return shift == 15 ? (r * 9798 + g * 19235 + b * 3735 + (1 << 14)) >> 15
                       : (r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14;
}

__global__ __launch_bounds__(256) void k_rgb2gray(const unsigned char *__restrict__ rgb, long long stride, int h,
                                                  int w, int shift, unsigned char *__restrict__ gray,
                                                  long long gstride)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    gray[(size_t)y * gstride + x] = (unsigned char)gray_of(rgb + (size_t)y * stride + (size_t)x * 3, 3, shift);
}

// ---------------------------------------------------------------------------------------------
// Fused assessment (k_assess_march below): one pass over both u8 images (6 B / pixel) for
//   * the sum of squared differences (PSNR),
//   * the Gaussian-11 SSIM map summed over the cropped (valid) region -> "gauss"  (branch A) and over the full
//     frame with REFLECT_101 borders -> "simple" (branch B): scipy's gaussian_filter(sigma 1.5, truncate 3.5) and
//     cv2.GaussianBlur((11,11), 1.5) are the same normalised kernel, the variants differ only in border and crop,
//   * the uniform 7x7 SSIM map, entirely in integers (window sums are exact), fp64 only for the final formula.
// fp64 throughout where the reference is (float64); the row pass works on exact integers: gray values, x^2 + y^2
// and x*y are ints, symmetric taps are pair-summed as ints and only 6 products per map are formed.  4 filtered maps
// (x, y, x^2 + y^2, x*y) replace the reference's 5: SSIM needs uxx and uyy only as their sum.
// ---------------------------------------------------------------------------------------------
// cv2.resize INTER_CUBIC, u8: per destination index the first source tap and four 11-bit fixed-point coefficients
struct CubicTab {
    int ofs;
    short c[4];
};

enum { ASSESS_SSE = 1, ASSESS_UNIFORM = 2, ASSESS_GAUSS = 4, ASSESS_SIMPLE = 8, ASSESS_ALL_BITS = 15 };

struct AssessParams {
    int h, w, shift, ry0, ry1, flags, same_c;
    int nch, ty;       // chunks of 11 rows a block marches, and the rows it produces (11 nch - 10)
    double c1a, c2a;   // constants for data_range (uniform / gauss)
    double c1b, c2b;   // constants for 255 (simple)
    double k1u, k2u;   // 49^2 c1a and 48*49 c2a: the uniform-7 variant in integer-scaled form
    double k[6];       // k[0] centre tap, k[j] the +-j taps
};

template <int CN>
__device__ __forceinline__ void load_gray_pair(const unsigned char *__restrict__ a, long long sa,
                                               const unsigned char *__restrict__ b, long long sb, int sy, int sx,
                                               int shift, int &ga, int &gb, unsigned &sq)
{
    const unsigned char *pa = a + (size_t)sy * sa + (size_t)sx * CN;
    const unsigned char *pb = b + (size_t)sy * sb + (size_t)sx * CN;
    if (CN == 1) {
        ga = pa[0];
        gb = pb[0];
        const int d = ga - gb;
        sq = (unsigned)(d * d);
    } else {
        const int r0 = pa[0], g0 = pa[1], b0 = pa[2], r1 = pb[0], g1 = pb[1], b1 = pb[2];
        if (shift == 15) {
            ga = (r0 * 9798 + g0 * 19235 + b0 * 3735 + (1 << 14)) >> 15;
            gb = (r1 * 9798 + g1 * 19235 + b1 * 3735 + (1 << 14)) >> 15;
        } else {
            ga = (r0 * 4899 + g0 * 9617 + b0 * 1868 + (1 << 13)) >> 14;
            gb = (r1 * 4899 + g1 * 9617 + b1 * 1868 + (1 << 13)) >> 14;
        }
        const int dr = r0 - r1, dg = g0 - g1, db = b0 - b1;
        sq = (unsigned)(dr * dr + dg * dg + db * db);
    }
}


// 1 / d to full double precision without the IEEE division sequence (d is a product of positive SSIM terms)
__device__ __forceinline__ double fast_recip(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

__device__ __forceinline__ double ssim_value(double ux, double uy, double spq, double dpq, double c1, double c2)
{
    // spq = uxx + uyy,  dpq = uxy
    const double uxuy = ux * uy, uu = ux * ux + uy * uy;
    const double a1 = 2.0 * uxuy + c1, a2 = 2.0 * (dpq - uxuy) + c2;
    const double b1 = uu + c1, b2 = (spq - uu) + c2;
    return (a1 * a2) * fast_recip(b1 * b2);
}

// ---------------------------------------------------------------------------------------------
// k_assess_march: all four metrics in ONE pass, column-marching.  A block is 256 columns wide (768 B of RGB per
// row: whole cache lines, ~4 % column halo) and walks down P.ty + 10 rows in chunks of 11.  Per chunk the block
// converts 11 rows of both images to gray once per pixel and leaves, per pixel, three dwords in LDS: x | y << 16,
// x*y and x^2 + y^2 (so no thread ever recomputes a neighbour's products, and one packed add pair-sums x and y
// together).  Then each thread owns one column: the row pass of its column (integer pair sums, 6 fp64 products per
// map) goes into an 11-deep register FIFO, the column pass reads the FIFO with static indices (the chunk loop body
// is the 11 unrolled rows), so the filtered maps never touch LDS.  The FIFO and the 7x7 window sums carry over
// from chunk to chunk: the only recomputed halo is the 10 rows at the top of a block (8 %).  The uniform-7 variant
// rides along: its per-row 7-tap integer sums go through a 7-slot per-column ring in LDS.
// ---------------------------------------------------------------------------------------------
#ifndef AM_TX
#define AM_TX 256                        /* columns (= threads) per block */
#endif
#define AM_R 5
#define AM_GP (AM_TX + 16)              /* row pitch in pixels: 10 halo columns, rounded up to groups of 4 */
#define AM_CH 11                        /* rows per chunk == FIFO depth */
#define AM_NCH_MAX 12                   /* chunks per block: P.nch <= 12, chosen per launch (rows / tail effect) */
/* a block marches 11 * nch rows and produces P.ty = 11 * nch - 10 of them; LDS 36 KB + 14 KB ring + 2 KB -> 3 blocks per CU */


// cv2.resize(INTER_CUBIC) sample of one destination pixel (all channels) -- the arithmetic of k_resize_cubic
template <int CN>
__device__ __forceinline__ void cubic_sample(const unsigned char *__restrict__ src, long long sstride, int sh, int sw,
                                             const CubicTab X, const CubicTab Y, int (&out)[CN])
{
    // 32-bit accumulators suffice: the cubic's taps (a = -0.75) have sum |c| <= 1.375, i.e. <= 2817 in 1/2048 units per
    // axis, so |acc| <= 255 * 2817^2 = 2.02e9 < 2^31 -- also after the rounding constant
    int acc[CN];
#pragma unroll
    for (int c = 0; c < CN; ++c) acc[c] = 0;
    const bool inner = X.ofs - 1 >= 0 && X.ofs + 2 <= sw - 1;
    const short yc[4] = {Y.c[0], Y.c[1], Y.c[2], Y.c[3]};
#pragma unroll 2
    for (int ky = 0; ky < 4; ++ky) {
        const unsigned char *r = src + (size_t)min(max(Y.ofs + ky - 1, 0), sh - 1) * sstride;
        int v[4][CN];
        if (inner && CN == 3) {
            const u3_t q = ld_u3_a1(r + (size_t)(X.ofs - 1) * 3);
            const unsigned wd[3] = {q.x, q.y, q.z};
#pragma unroll
            for (int b = 0; b < 12; ++b) v[b / 3][b % 3] = (int)((wd[b >> 2] >> (8 * (b & 3))) & 0xFFu);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int sx = min(max(X.ofs + k - 1, 0), sw - 1) * CN;
#pragma unroll
                for (int c = 0; c < CN; ++c) v[k][c] = (int)r[sx + c];
            }
        }
#pragma unroll
        for (int c = 0; c < CN; ++c) {
            const int hs = v[0][c] * X.c[0] + v[1][c] * X.c[1] + v[2][c] * X.c[2] + v[3][c] * X.c[3];
            acc[c] += hs * (int)yc[ky];
        }
    }
#pragma unroll
    for (int c = 0; c < CN; ++c) {
        const int t = (acc[c] + (1 << 21)) >> 22;
        out[c] = t < 0 ? 0 : (t > 255 ? 255 : t);
    }
}

// gray conversion + per-pixel products of 4-pixel groups of chunk `ch` into LDS; returns this thread's share of the
// squared differences of the block's own pixels
template <int CN>
__device__ __forceinline__ unsigned assess_load_chunk(const unsigned char *__restrict__ a, long long sa,
                                                      const unsigned char *__restrict__ b, long long sb,
                                                      const AssessParams &P, int bx0, int by0, int ch,
                                                      int rows_needed, unsigned (*XY)[AM_GP],
                                                      unsigned (*QQ)[AM_GP], unsigned (*PP)[AM_GP])
{
    // a thread squares at most 12 chunks x 3 groups x 4 pixels x 3 channels = 432 differences per block (< 2.9e7): 32 bits
    unsigned sse = 0;
    const bool want_sse = (P.flags & ASSESS_SSE) != 0;
    for (int i = threadIdx.x; i < AM_CH * (AM_GP / 4); i += AM_TX) {
        const int ly = i / (AM_GP / 4), lx = (i - ly * (AM_GP / 4)) * 4;
        const int lr = ch * AM_CH + ly;
        if (lr >= rows_needed) break;                       // rows grow with i
        const int gy = by0 - AM_R + lr, gx = bx0 - AM_R + lx;
        const int sy = reflect101(gy, P.h);
        int ga[4], gb[4];
        unsigned sq[4];
        if (gx >= 0 && gx + 3 < P.w) {
            const unsigned char *pa = a + (size_t)sy * sa + (size_t)gx * CN;
            const unsigned char *pb = b + (size_t)sy * sb + (size_t)gx * CN;
            if (CN == 3) {
                const u3_t qa = ld_u3_a1(pa), qb = ld_u3_a1(pb);
                const unsigned wa[3] = {qa.x, qa.y, qa.z}, wb[3] = {qb.x, qb.y, qb.z};
                int ca[12], cb[12];
#pragma unroll
                for (int t = 0; t < 12; ++t) {
                    ca[t] = (int)((wa[t >> 2] >> (8 * (t & 3))) & 0xFFu);
                    cb[t] = (int)((wb[t >> 2] >> (8 * (t & 3))) & 0xFFu);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ga[k] = gray_rgb(ca[3 * k], ca[3 * k + 1], ca[3 * k + 2], P.shift);
                    gb[k] = gray_rgb(cb[3 * k], cb[3 * k + 1], cb[3 * k + 2], P.shift);
                    const int dr = ca[3 * k] - cb[3 * k], dg = ca[3 * k + 1] - cb[3 * k + 1], db = ca[3 * k + 2] - cb[3 * k + 2];
                    sq[k] = (unsigned)(dr * dr + dg * dg + db * db);
                }
            } else {
                const unsigned qa = *(const u1_a1_t *)pa, qb = *(const u1_a1_t *)pb;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ga[k] = (int)((qa >> (8 * k)) & 0xFFu);
                    gb[k] = (int)((qb >> (8 * k)) & 0xFFu);
                    const int d = ga[k] - gb[k];
                    sq[k] = (unsigned)(d * d);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) load_gray_pair<CN>(a, sa, b, sb, sy, reflect101(gx + k, P.w), P.shift, ga[k], gb[k], sq[k]);
        }
        u4_t vxy, vq, vp;
        unsigned txy[4], tq[4], tp[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            txy[k] = (unsigned)ga[k] | ((unsigned)gb[k] << 14);
            tq[k] = (unsigned)__mul24(ga[k], gb[k]);
            tp[k] = (unsigned)(__mul24(ga[k], ga[k]) + __mul24(gb[k], gb[k]));
        }
        vxy.x = txy[0]; vxy.y = txy[1]; vxy.z = txy[2]; vxy.w = txy[3];
        vq.x = tq[0]; vq.y = tq[1]; vq.z = tq[2]; vq.w = tq[3];
        vp.x = tp[0]; vp.y = tp[1]; vp.z = tp[2]; vp.w = tp[3];
        *(u4_t *)&XY[ly][lx] = vxy;
        *(u4_t *)&QQ[ly][lx] = vq;
        *(u4_t *)&PP[ly][lx] = vp;
        if (want_sse && lr >= AM_R && lr < AM_R + P.ty && gy < P.ry1 && gy < P.h) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (lx + k >= AM_R && lx + k < AM_R + AM_TX && gx + k < P.w) sse += sq[k];
        }
    }
    return sse;
}

// An address the compiler cannot fold into its users: the eleven taps of a row are then read with ds_read2_b32 off ONE
// base register per array and row (the 8-bit dword offsets of ds_read2 do not reach across rows, and left alone the
// compiler materialises five bases per array and row with VALU adds).
typedef __attribute__((address_space(3))) const unsigned lds_cu32;
__device__ __forceinline__ lds_cu32 *lds_row_base(lds_cu32 *row0, int bytes)
{
    // one explicit VALU add per array and row off the thread's row-0 address: no per-row base registers kept alive
    lds_cu32 *q;
    asm volatile("v_add_u32 %0, %2, %1" : "=v"(q) : "v"(row0), "n"(bytes));     // literal goes in src0
    return q;
}

// 1 / d for the SSIM quotient: hardware estimate + one Newton step (relative error ~1e-15; the metric's bar is 1e-9
// against the oracle, 1e-4 against the reference)
__device__ __forceinline__ double ssim_recip(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    return fma(fma(-d, r, 1.0), r, r);
}

__device__ __forceinline__ double ssim_quot(double ux, double uy, double spq, double dpq, double c1, double c2)
{
    // spq = uxx + uyy,  dpq = uxy
    const double uxuy = ux * uy, uu = fma(ux, ux, uy * uy);
    const double a1 = fma(2.0, uxuy, c1), a2 = fma(2.0, dpq - uxuy, c2);
    const double b1 = uu + c1, b2 = (spq - uu) + c2;
    return (a1 * a2) * ssim_recip(b1 * b2);
}

// Compile-time variants: GAUSS (the two Gaussian-11 sums and their 88-register FIFO), UNIF (the uniform-7 sum and its LDS
// ring), SAMEC (data_range == 255: the cropped and the full-frame Gaussian variants share one SSIM value per pixel).
// What the march does per row, in instruction terms: 33 LDS dwords, 15 integer pair sums, 24 conversions + 24 fp64
// multiply-adds (row pass), 20 fp64 adds + 24 multiply-adds (column pass), ~20 fp64 operations per SSIM value; validity of
// a ROW is block-uniform (scalar branches), validity of a COLUMN is applied once, to the thread's sums, after the march
// (out-of-image columns hold reflected data, so their values are finite and simply dropped).
// In LDS x and y travel packed as x | y << 14: pair sums (<= 510), 7-tap sums (<= 1785) and 49-sample window sums
// (<= 12495 < 2^14) all stay inside their fields, so one integer add serves both images at every stage.
template <int CN, bool GAUSS, bool UNIF, bool SAMEC>
__global__ __launch_bounds__(AM_TX, 3) void k_assess_march(const unsigned char *__restrict__ a, long long sa,
                                                      const unsigned char *__restrict__ b, long long sb,
                                                      AssessParams P, double *__restrict__ part)
{
    // one array, so the march addresses all three maps off ONE per-thread base register
    __shared__ __attribute__((aligned(16))) unsigned L3[3][AM_CH][AM_GP];
    unsigned (*XY)[AM_GP] = L3[0];                                       // x | y << 14
    unsigned (*QQ)[AM_GP] = L3[1];                                       // x * y
    unsigned (*PP)[AM_GP] = L3[2];                                       // x^2 + y^2
    // per-row 7-tap sums of the last seven rows, two dwords per column: {sx:14 | sy:11 @14 | sq lo:7 @25}, {sp:20 | sq hi:12 @20}
    __shared__ unsigned U[UNIF ? 7 : 1][2][AM_TX];
    __shared__ double red[AM_TX / 64][4];
    const int c = threadIdx.x;
    const int bx0 = blockIdx.x * AM_TX, by0 = P.ry0 + blockIdx.y * P.ty;
    const int rows_needed = min(P.ty, P.ry1 - by0) + 2 * AM_R;          // block-uniform
    const int mx = bx0 + c;
    double f[GAUSS ? 4 : 1][11];
    if (GAUSS) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int i = 0; i < 11; ++i) f[m][i] = 0.0;
    }
    unsigned t_xy = 0, t_p = 0, t_q = 0;                                // 49-sample window sums (uniform-7)
    double sum_int = 0.0, sum_all = 0.0, sum_u = 0.0;
    unsigned sse = 0;                                                   // this thread's squared differences (fits: see the loader)
    int slot = 0;                                                       // row index mod 7
    const double k0 = P.k[0], k1 = P.k[1], k2 = P.k[2], k3 = P.k[3], k4 = P.k[4], k5 = P.k[5];
    lds_cu32 *xy0 = (lds_cu32 *)&L3[0][0][c];
    constexpr int MAPB = AM_CH * AM_GP * 4;                             // bytes between the maps
    // full-frame samples of the 5 top / bottom image rows: touched only by the first and last block rows, so the running
    // sum lives in LDS (2 KB) instead of two registers of every thread of every block
    __shared__ double EDGE[(GAUSS && SAMEC) ? AM_TX : 1];
    if (GAUSS && SAMEC) EDGE[c] = 0.0;                                  // own slot only: no barrier needed
#pragma unroll 1
    for (int ch = 0; ch < P.nch; ++ch) {
        if (ch * AM_CH >= rows_needed) break;
        __syncthreads();                                                // the previous chunk has been read
        sse += assess_load_chunk<CN>(a, sa, b, sb, P, bx0, by0, ch, rows_needed, XY, QQ, PP);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < AM_CH; ++s) {
            const int r = ch * AM_CH + s;
            if (r >= rows_needed) continue;                             // block-uniform (no break: the loop must unroll)
            lds_cu32 *rxy = lds_row_base(xy0, s * AM_GP * 4), *rq = lds_row_base(xy0, MAPB + s * AM_GP * 4),
                     *rp = lds_row_base(xy0, 2 * MAPB + s * AM_GP * 4);
            unsigned xy[11], qv[11], pv[11];
#pragma unroll
            for (int j = 0; j < 11; ++j) {
                xy[j] = rxy[j];
                qv[j] = rq[j];
                pv[j] = rp[j];
            }
            // symmetric pair sums, shared by the Gaussian row pass (all five) and the 7-tap box sums (the first three)
            unsigned sxy[6], sp[6], sq[6];
            sxy[0] = xy[5]; sp[0] = pv[5]; sq[0] = qv[5];
#pragma unroll
            for (int j = 1; j <= (GAUSS ? AM_R : 3); ++j) {
                sxy[j] = xy[5 - j] + xy[5 + j];                         // both images in one add
                sp[j] = pv[5 - j] + pv[5 + j];
                sq[j] = qv[5 - j] + qv[5 + j];
            }
            if (GAUSS) {
                const double kk[6] = {k0, k1, k2, k3, k4, k5};
                double hx = (double)(sxy[0] & 0x3FFFu) * kk[0], hy = (double)(sxy[0] >> 14) * kk[0];
                double hp = (double)sp[0] * kk[0], hq = (double)sq[0] * kk[0];
#pragma unroll
                for (int j = 1; j <= AM_R; ++j) {
                    hx = fma((double)(sxy[j] & 0x3FFFu), kk[j], hx);
                    hy = fma((double)(sxy[j] >> 14), kk[j], hy);
                    hp = fma((double)sp[j], kk[j], hp);
                    hq = fma((double)sq[j], kk[j], hq);
                }
                f[0][s] = hx; f[1][s] = hy; f[2][s] = hp; f[3][s] = hq;
            }
            if (UNIF) {
                const unsigned uxy = ((sxy[0] + sxy[1]) + sxy[2]) + sxy[3];
                const unsigned up = ((sp[0] + sp[1]) + sp[2]) + sp[3];
                const unsigned uq = ((sq[0] + sq[1]) + sq[2]) + sq[3];
                if (r >= 7) {                                           // block-uniform: the slot holds row r - 7
                    const unsigned o0 = U[slot][0][c], o1 = U[slot][1][c];
                    t_xy -= o0 & 0x1FFFFFFu;
                    t_p -= o1 & 0xFFFFFu;
                    t_q -= (o0 >> 25) | ((o1 >> 20) << 7);
                }
                t_xy += uxy; t_p += up; t_q += uq;
                U[slot][0][c] = uxy | (uq << 25);
                U[slot][1][c] = up | ((uq >> 7) << 20);
                slot = slot == 6 ? 0 : slot + 1;
                const int orow = r - 8, my = by0 + orow;                // window rows r-6 .. r, centre r-3
                if (orow >= 0 && orow < P.ty && my < P.ry1 && my >= 3 && my < P.h - 3) {        // block-uniform
                    // SSIM of the 49-sample window with both fractions scaled to integers: with S. the window sums,
                    //   (2 ux uy + C1) / (ux^2 + uy^2 + C1) = (2 Sx Sy + 49^2 C1) / (Sx^2 + Sy^2 + 49^2 C1)
                    //   (2 cov + C2) / (var_x + var_y + C2) = (2 (49 Sxy - Sx Sy) + 48*49 C2)
                    //                                         / (49 (Sxx + Syy) - (Sx^2 + Sy^2) + 48*49 C2)
                    // (sample covariance, N - 1 = 48).  Everything left of the constants is exact 32-bit integer
                    // arithmetic (|values| < 3.2e8, every factor below 2^24); fp64 enters with the constants.
                    const int sx = (int)(t_xy & 0x3FFFu), sy = (int)(t_xy >> 14);
                    const int sxsy = __mul24(sx, sy), ss = __mul24(sx, sx) + __mul24(sy, sy);
                    const int ncov = __mul24(49, (int)t_q) - sxsy, nvar = __mul24(49, (int)t_p) - ss;
                    const double a1 = fma(2.0, (double)sxsy, P.k1u), a2 = fma(2.0, (double)ncov, P.k2u);
                    const double b1 = (double)ss + P.k1u, b2 = (double)nvar + P.k2u;
                    sum_u += (a1 * a2) * ssim_recip(b1 * b2);
                }
            }
            if (GAUSS && r >= 2 * AM_R) {
                const int orow = r - 2 * AM_R, my = by0 + orow;         // rows r-10 .. r are in the FIFO, centre r-5
                if (orow < P.ty && my < P.ry1 && my < P.h) {            // block-uniform
                    const double kk[6] = {k0, k1, k2, k3, k4, k5};
                    double u[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        double acc = f[m][(s + 6) % 11] * kk[0];
#pragma unroll
                        for (int j = 1; j <= AM_R; ++j)
                            acc = fma(f[m][(s + 6 + 11 - j) % 11] + f[m][(s + 6 + j) % 11], kk[j], acc);
                        u[m] = acc;
                    }
                    const bool inner_row = my >= AM_R && my < P.h - AM_R;       // block-uniform
                    if (SAMEC) {
                        // one SSIM value serves both variants: the cropped sum is the full-frame sum minus the (at most
                        // ten) image rows outside the crop, which only the blocks at the top / bottom ever see
                        const double sv = ssim_quot(u[0], u[1], u[2], u[3], P.c1a, P.c2a);
                        sum_all += sv;
                        if (!inner_row) EDGE[c] += sv;
                    } else {
                        if (P.flags & ASSESS_SIMPLE) sum_all += ssim_quot(u[0], u[1], u[2], u[3], P.c1b, P.c2b);
                        if (inner_row && (P.flags & ASSESS_GAUSS)) sum_int += ssim_quot(u[0], u[1], u[2], u[3], P.c1a, P.c2a);
                    }
                }
            }
        }
    }
    // column validity, once: the full-frame variant counts every image column, the cropped ones lose 5 / 3 per side
    if (GAUSS && SAMEC) sum_int = sum_all - EDGE[c];
    if (!(mx < P.w)) sum_all = 0.0;
    if (!(mx >= AM_R && mx < P.w - AM_R)) sum_int = 0.0;
    if (!(mx >= 3 && mx < P.w - 3)) sum_u = 0.0;
    sum_int = wave_sum_f64(sum_int);
    sum_all = wave_sum_f64(sum_all);
    sum_u = wave_sum_f64(sum_u);
    const double dsse = wave_sum_f64((double)sse);
    if ((c & 63) == 0) {
        red[c >> 6][0] = sum_int;
        red[c >> 6][1] = sum_all;
        red[c >> 6][2] = dsse;
        red[c >> 6][3] = sum_u;
    }
    __syncthreads();
    if (c < 4) {
        const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        double t = red[0][c];
#pragma unroll
        for (int wv = 1; wv < AM_TX / 64; ++wv) t += red[wv][c];
        part[blk * 4 + c] = t;
    }
}

// Deterministic two-level sum of per-block partials laid out as part[i * ncomp + comp]:
// level 1: block j sums entries [j*1024, (j+1)*1024) in a fixed tree -> tmp[j * ncomp + comp];
// level 2 (one block): sums the level-1 results -> out[comp].
__global__ __launch_bounds__(256) void k_reduce_partials(const double *__restrict__ part, long long n, int ncomp,
                                                         double *__restrict__ out)
{
    __shared__ double sh[256];
    const long long base = (long long)blockIdx.x * 1024;
    for (int comp = 0; comp < ncomp; ++comp) {
        double s = 0.0;
        for (int k = 0; k < 4; ++k) {
            const long long i = base + k * 256 + threadIdx.x;
            if (i < n) s += part[i * ncomp + comp];
        }
        sh[threadIdx.x] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[(size_t)blockIdx.x * ncomp + comp] = sh[0];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// cv2.resize INTER_CUBIC, u8
// ---------------------------------------------------------------------------------------------

// RGB form: one thread = 4 consecutive destination pixels of a row (one row-table entry, 12 bytes stored as 3 dwords)
__global__ __launch_bounds__(256) void k_resize_cubic_rgb4(const unsigned char *__restrict__ src, long long sstride, int h,
                                                           int w, const CubicTab *__restrict__ xt,
                                                           const CubicTab *__restrict__ yt, int x0, int y0, int ww,
                                                           int wh, unsigned char *__restrict__ dst, long long dstride)
{
    const int x = (blockIdx.x * 64 + threadIdx.x) * 4, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= ww || y >= wh) return;
    const CubicTab Y = yt[y0 + y];
    const int nx = min(4, ww - x);
    unsigned ob[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int v[3] = {0, 0, 0};
        if (k < nx) cubic_sample<3>(src, sstride, h, w, xt[x0 + x + k], Y, v);
        ob[3 * k] = (unsigned)v[0]; ob[3 * k + 1] = (unsigned)v[1]; ob[3 * k + 2] = (unsigned)v[2];
    }
    unsigned char *o = dst + (size_t)y * dstride + (size_t)x * 3;
    if (nx == 4 && ((((size_t)o) & 3) == 0)) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
            ((unsigned *)o)[q] = ob[4 * q] | (ob[4 * q + 1] << 8) | (ob[4 * q + 2] << 16) | (ob[4 * q + 3] << 24);
    } else {
        for (int k = 0; k < nx; ++k) {
            o[3 * k] = (unsigned char)ob[3 * k]; o[3 * k + 1] = (unsigned char)ob[3 * k + 1]; o[3 * k + 2] = (unsigned char)ob[3 * k + 2];
        }
    }
}

// Upscaling form (destination rows >= source rows): consecutive destination rows read the same four source rows, so the
// horizontal pass is not repeated per destination row.  One thread owns 4 destination columns and marches down a segment of
// destination rows; it keeps the horizontal results of the four source rows of the current row window (4 x 4 x 3 ints) and
// computes ONE new source row when the window moves on (every dst_h / src_h rows); per destination row only the vertical
// pass remains (48 multiply-adds instead of 192 + 48 and sixteen 12-byte loads).  Same integers as cubic_sample:
// hs = sum v * xc, acc = sum hs * yc, (acc + 2^21) >> 22, clamped.
#define RUP_SEG 64
__device__ __forceinline__ void rup_row_pass(const unsigned char *__restrict__ src, long long sstride, int sh, int sw, int row,
                                             const CubicTab (&X)[4], int (&H)[4][3])
{
    const unsigned char *r = src + (size_t)min(max(row, 0), sh - 1) * sstride;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int v[4][3];
        if (X[k].ofs - 1 >= 0 && X[k].ofs + 2 <= sw - 1) {
            const u3_t q = ld_u3_a1(r + (size_t)(X[k].ofs - 1) * 3);
            const unsigned wd[3] = {q.x, q.y, q.z};
#pragma unroll
            for (int b = 0; b < 12; ++b) v[b / 3][b % 3] = (int)((wd[b >> 2] >> (8 * (b & 3))) & 0xFFu);
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int sx = min(max(X[k].ofs + t - 1, 0), sw - 1) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) v[t][c] = (int)r[sx + c];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) H[k][c] = v[0][c] * X[k].c[0] + v[1][c] * X[k].c[1] + v[2][c] * X[k].c[2] + v[3][c] * X[k].c[3];
    }
}

__global__ __launch_bounds__(256, 2) void k_resize_cubic_up_rgb(const unsigned char *__restrict__ src, long long sstride, int sh,
                                                             int sw, const CubicTab *__restrict__ xt,
                                                             const CubicTab *__restrict__ yt, int x0, int y0, int ww, int wh,
                                                             unsigned char *__restrict__ dst, long long dstride)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int ya = blockIdx.y * RUP_SEG, yb = min(ya + RUP_SEG, wh);
    if (x >= ww || ya >= yb) return;
    const int nx = min(4, ww - x);
    CubicTab X[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) X[k] = xt[x0 + min(x + k, ww - 1)];        // columns past the window repeat the last one, not stored
    // Source row r of the window lives in slot r & 3 (no copying when the window moves: the new row overwrites the slot of
    // the row that left); the vertical taps are matched to the slots instead -- the row window is the same for the whole
    // block, so that is scalar work.
    int H[4][4][3];                                                       // [slot][pixel][channel]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) H[j][k][c] = 0;
    int cur = yt[y0 + ya].ofs - 4;                                        // the window is filled by the step that advances it
#pragma unroll 1
    for (int y = ya; y < yb; ++y) {
        const CubicTab Y = yt[y0 + y];
#pragma unroll 1
        while (cur < Y.ofs) {                                             // the row window moves down by one source row
            ++cur;
            const int row = cur + 2;                                      // rows cur - 1 .. cur + 2 are held
            switch (row & 3) {
            case 0: rup_row_pass(src, sstride, sh, sw, row, X, H[0]); break;
            case 1: rup_row_pass(src, sstride, sh, sw, row, X, H[1]); break;
            case 2: rup_row_pass(src, sstride, sh, sw, row, X, H[2]); break;
            default: rup_row_pass(src, sstride, sh, sw, row, X, H[3]); break;
            }
        }
        // tap t belongs to row cur - 1 + t, which sits in slot (cur - 1 + t) & 3: rotate the taps onto the slots
        const int c0 = Y.c[0], c1 = Y.c[1], c2 = Y.c[2], c3 = Y.c[3];
        int yc[4];
        switch ((cur - 1) & 3) {
        case 0: yc[0] = c0; yc[1] = c1; yc[2] = c2; yc[3] = c3; break;
        case 1: yc[0] = c3; yc[1] = c0; yc[2] = c1; yc[3] = c2; break;
        case 2: yc[0] = c2; yc[1] = c3; yc[2] = c0; yc[3] = c1; break;
        default: yc[0] = c1; yc[1] = c2; yc[2] = c3; yc[3] = c0; break;
        }
        unsigned ob[12];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                // the reference adds the four products in tap order; integer addition is exact, so the slot order gives the same sum
                const int acc = (H[0][k][c] * yc[0] + H[1][k][c] * yc[1]) + (H[2][k][c] * yc[2] + H[3][k][c] * yc[3]);
                const int t = (acc + (1 << 21)) >> 22;
                ob[3 * k + c] = (unsigned)(t < 0 ? 0 : (t > 255 ? 255 : t));
            }
        unsigned char *o = dst + (size_t)y * dstride + (size_t)x * 3;
        if (nx == 4 && ((((size_t)o) & 3) == 0)) {
#pragma unroll
            for (int q = 0; q < 3; ++q)
                ((unsigned *)o)[q] = ob[4 * q] | (ob[4 * q + 1] << 8) | (ob[4 * q + 2] << 16) | (ob[4 * q + 3] << 24);
        } else {
            for (int k = 0; k < nx; ++k) {
                o[3 * k] = (unsigned char)ob[3 * k]; o[3 * k + 1] = (unsigned char)ob[3 * k + 1]; o[3 * k + 2] = (unsigned char)ob[3 * k + 2];
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_resize_cubic(const unsigned char *__restrict__ src, long long sstride,
                                                      int h, int w, int cn, const CubicTab *__restrict__ xt,
                                                      const CubicTab *__restrict__ yt, int x0, int y0, int ww,
                                                      int wh, unsigned char *__restrict__ dst, long long dstride)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= ww || y >= wh) return;
    const CubicTab X = xt[x0 + x], Y = yt[y0 + y];
    if (cn == 3 || cn == 1) {                    // one 12-byte load per tap row instead of 12 byte loads (cubic_sample)
        unsigned char *o = dst + (size_t)y * dstride + (size_t)x * cn;
        if (cn == 3) {
            int v[3];
            cubic_sample<3>(src, sstride, h, w, X, Y, v);
            o[0] = (unsigned char)v[0]; o[1] = (unsigned char)v[1]; o[2] = (unsigned char)v[2];
        } else {
            int v[1];
            cubic_sample<1>(src, sstride, h, w, X, Y, v);
            o[0] = (unsigned char)v[0];
        }
        return;
    }
    int sx[4], sy[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sx[k] = min(max(X.ofs + k - 1, 0), w - 1) * cn;
        sy[k] = min(max(Y.ofs + k - 1, 0), h - 1);
    }
    for (int c = 0; c < cn; ++c) {
        long long acc = 0;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const unsigned char *r = src + (size_t)sy[ky] * sstride + c;
            const int hs = (int)r[sx[0]] * X.c[0] + (int)r[sx[1]] * X.c[1] + (int)r[sx[2]] * X.c[2] +
                           (int)r[sx[3]] * X.c[3];
            acc += (long long)hs * Y.c[ky];
        }
        const long long v = (acc + (1 << 21)) >> 22;
        dst[(size_t)y * dstride + (size_t)x * cn + c] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

static void cubic_table(int n_src, int n_dst, std::vector<CubicTab> &tab)
{
    tab.resize(n_dst);
    const double scale = 1.0 / ((double)n_dst / (double)n_src);
    for (int d = 0; d < n_dst; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        const int s = (int)floorf(f);
        f -= (float)s;
        const float A = -0.75f;
        float c[4];
        c[0] = ((A * (f + 1.0f) - 5.0f * A) * (f + 1.0f) + 8.0f * A) * (f + 1.0f) - 4.0f * A;
        c[1] = ((A + 2.0f) * f - (A + 3.0f)) * f * f + 1.0f;
        const float f2 = 1.0f - f;
        c[2] = ((A + 2.0f) * f2 - (A + 3.0f)) * f2 * f2 + 1.0f;
        c[3] = 1.0f - c[0] - c[1] - c[2];
        tab[d].ofs = s;
        for (int k = 0; k < 4; ++k) {
            const float v = rintf(c[k] * 2048.0f);
            tab[d].c[k] = (short)(v < -32768.f ? -32768.f : (v > 32767.f ? 32767.f : v));
        }
    }
}


// ---------------------------------------------------------------------------------------------
// TilingModule.merge_tiles feather path (tiling_module.py:1074-1175), canvas-centric
// ---------------------------------------------------------------------------------------------
struct LinTab {
    int ofs;       // left / top source index (clamped)
    short a0, a1;  // 11-bit coefficients of cv::resize INTER_LINEAR (u8 data)
    float f;       // the fraction itself (float data: coefficients 1 - f and f)
};

struct MergeDev {
    int x, y, src_w, src_h, out_w, out_h;
    int ov_t, ov_b, ov_l, ov_r;
    int resize;          // 1: bilinear resize src -> out
    int xtab, ytab;      // offsets into the LinTab array
    double st, sb, sl, sr;  // np.linspace steps: +1/(ov-1) (top/left), -1/(ov-1) (bottom/right); 0 when ov == 1
};

template <int DT>
__global__ __launch_bounds__(256) void k_feather_merge(const MergeDev *__restrict__ tiles, const TileSrc *__restrict__ srcs,
                                                       const LinTab *__restrict__ tabs, int n, int blending,
                                                       unsigned char *__restrict__ canvas, long long cstride, int ch,
                                                       int cw)
{
    // tiles that touch this 256 x 4 pixel block, in list order (wave 0, ballot-compacted): a canvas pixel is covered by
    // 1-4 of the n tiles, so the per-pixel loop runs over this short list instead of all of them
    __shared__ int s_cnt;
    __shared__ int s_list[64];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const int bx0 = blockIdx.x * 256, by0 = blockIdx.y * 4;
    if (tid < 64) {
        int cnt = 0;
        for (int base = 0; base < n; base += 64) {
            const int t = base + tid;
            bool hit = false;
            if (t < n) {
                const MergeDev &T = tiles[t];
                hit = T.x < bx0 + 256 && T.x + T.out_w > bx0 && T.y < by0 + 4 && T.y + T.out_h > by0;
            }
            const unsigned long long m = __ballot(hit);
            if (hit) {
                const int pos = cnt + __popcll(m & ((1ull << tid) - 1ull));
                if (pos < 64) s_list[pos] = t;
            }
            cnt += __popcll(m);
        }
        if (tid == 0) s_cnt = cnt;
    }
    __syncthreads();
    const int ncand = s_cnt;
    const bool listed = ncand <= 64;                 // more than 64 tiles over one block: walk all of them
    // one thread = 4 consecutive canvas pixels of one row: the row part of the weight is formed once, unresized tile
    // pixels come in one 12-byte load
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4, y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= cw || y >= ch) return;
    const int nx = min(4, cw - x0);
    // Blocks that lie inside ONE unresized u8 tile, beyond its ramps (58 % of a 5 x 5 grid's canvas): weight exactly 1, so
    // acc = p * 1, wacc = 1, p / 1 = p -- the pixel itself: a copy (0.67 -> 0.58 ms for 25 tiles / 200 MP).  Building the
    // tile list once per 16 rows instead of 4, with the descriptors in LDS, was measured too: no change (0.60 ms).
    if (DT == SRC_U8 && ncand == 1) {
        const MergeDev &T = tiles[s_list[0]];
        const int fl = blending ? T.ov_l : 0, fr = blending ? T.ov_r : 0, ft = blending ? T.ov_t : 0, fb = blending ? T.ov_b : 0;
        if (!T.resize && bx0 >= T.x + fl && min(bx0 + 256, cw) <= T.x + T.out_w - fr && by0 >= T.y + ft &&
            min(by0 + 4, ch) <= T.y + T.out_h - fb) {
            const unsigned char *sp = (const unsigned char *)srcs[s_list[0]].p + (size_t)(y - T.y) * srcs[s_list[0]].stride +
                                      (size_t)(x0 - T.x) * 3;
            unsigned char *o = canvas + (size_t)y * cstride + (size_t)x0 * 3;
            if (nx == 4) {
                const u3_t q = ld_u3_a1_g(sp);
                *(__attribute__((address_space(1))) u3_a1_t *)o = q;
            } else {
                for (int i = 0; i < 3 * nx; ++i) o[i] = sp[i];
            }
            return;
        }
    }
    float acc[4][3], wacc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        wacc[k] = 0.f;
        acc[k][0] = acc[k][1] = acc[k][2] = 0.f;
    }
    for (int i = 0; i < (listed ? ncand : n); ++i) {
        const int t = listed ? s_list[i] : i;
        const MergeDev &T = tiles[t];
        const int lx0 = x0 - T.x, ly = y - T.y;
        if (lx0 + nx <= 0 || ly < 0 || lx0 >= T.out_w || ly >= T.out_h) continue;
        // weight[:t] *= linspace(0,1,t); weight[-b:] *= linspace(1,0,b); then columns -- each product in float64,
        // rounded to fp32 (NumPy's in-place multiply of an fp32 array by an fp64 ramp).  Rows first: shared by the 4 px.
        float wy = 1.0f;
        if (blending) {
            if (T.ov_t > 0 && ly < T.ov_t) {
                const double r = (ly == T.ov_t - 1 && T.ov_t > 1) ? 1.0 : (double)ly * T.st + 0.0;
                wy = (float)((double)wy * r);
            }
            if (T.ov_b > 0 && ly >= T.out_h - T.ov_b) {
                const int j = ly - (T.out_h - T.ov_b);
                const double r = (j == T.ov_b - 1 && T.ov_b > 1) ? 0.0 : (double)j * T.sb + 1.0;
                wy = (float)((double)wy * r);
            }
        }
        const unsigned char *base = (const unsigned char *)srcs[t].p;
        const long long st = srcs[t].stride;
        const bool whole = lx0 >= 0 && lx0 + 3 < T.out_w && nx == 4;
        unsigned pix[12];
        if (DT == SRC_U8 && !T.resize && whole) {
            const u3_t q = ld_u3_a1_g(base + (size_t)ly * st + (size_t)lx0 * 3);
            const unsigned wd[3] = {q.x, q.y, q.z};
#pragma unroll
            for (int b = 0; b < 12; ++b) pix[b] = (wd[b >> 2] >> (8 * (b & 3))) & 0xFFu;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int lx = lx0 + k;
            if (k >= nx || lx < 0 || lx >= T.out_w) continue;
            float w = wy;
            if (blending) {
                if (T.ov_l > 0 && lx < T.ov_l) {
                    const double r = (lx == T.ov_l - 1 && T.ov_l > 1) ? 1.0 : (double)lx * T.sl + 0.0;
                    w = (float)((double)w * r);
                }
                if (T.ov_r > 0 && lx >= T.out_w - T.ov_r) {
                    const int j = lx - (T.out_w - T.ov_r);
                    const double r = (j == T.ov_r - 1 && T.ov_r > 1) ? 0.0 : (double)j * T.sr + 1.0;
                    w = (float)((double)w * r);
                }
            }
            if (DT == SRC_F32) {
                // float tile data (tiling_module.py:1104-1109: astype(float32), or cv2.resize's float INTER_LINEAR path --
                // rows first, S[x0] * (1 - fx) + S[x1] * fx, then the same between the two rows; parity unpinned)
                if (T.resize) {
                    const LinTab X = tabs[T.xtab + lx], Y = tabs[T.ytab + ly];
                    const int x1 = min(X.ofs + 1, T.src_w - 1), y1 = min(Y.ofs + 1, T.src_h - 1);
                    const float *r0 = (const float *)(base + (size_t)Y.ofs * st), *r1 = (const float *)(base + (size_t)y1 * st);
                    const float ax0 = 1.0f - X.f, ax1 = X.f, ay0 = 1.0f - Y.f, ay1 = Y.f;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float s0 = r0[X.ofs * 3 + c] * ax0 + r0[x1 * 3 + c] * ax1;
                        const float s1 = r1[X.ofs * 3 + c] * ax0 + r1[x1 * 3 + c] * ax1;
                        acc[k][c] += (s0 * ay0 + s1 * ay1) * w;
                    }
                } else {
                    const float *r0 = (const float *)(base + (size_t)ly * st) + (size_t)lx * 3;
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[k][c] += r0[c] * w;
                }
            } else if (T.resize) {
                const LinTab X = tabs[T.xtab + lx], Y = tabs[T.ytab + ly];
                const int x1 = min(X.ofs + 1, T.src_w - 1), y1 = min(Y.ofs + 1, T.src_h - 1);
                const unsigned char *r0 = base + (size_t)Y.ofs * st, *r1 = base + (size_t)y1 * st;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int s0 = (int)r0[X.ofs * 3 + c] * X.a0 + (int)r0[x1 * 3 + c] * X.a1;
                    const int s1 = (int)r1[X.ofs * 3 + c] * X.a0 + (int)r1[x1 * 3 + c] * X.a1;
                    const int v = (((Y.a0 * (s0 >> 4)) >> 16) + ((Y.a1 * (s1 >> 4)) >> 16) + 2) >> 2;
                    acc[k][c] += (float)(unsigned char)v * w;
                }
            } else if (whole) {
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[k][c] += (float)pix[3 * k + c] * w;
            } else {
                const unsigned char *r0 = base + (size_t)ly * st + (size_t)lx * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[k][c] += (float)r0[c] * w;
            }
            wacc[k] += w;
        }
    }
    unsigned ob[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float wv = wacc[k] > 1e-6f ? wacc[k] : 1e-6f;
        float q[3];
        div_shared<3>(acc[k], wv, q);             // the IEEE quotients, one reciprocal per pixel (see the final gather)
#pragma unroll
        for (int c = 0; c < 3; ++c) ob[3 * k + c] = (unsigned)(unsigned char)(int)q[c];   // astype(uint8): truncation, no clip
    }
    unsigned char *o = canvas + (size_t)y * cstride + (size_t)x0 * 3;
    if (nx == 4 && ((cstride & 3) == 0) && ((((size_t)canvas) & 3) == 0)) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
            ((unsigned *)o)[q] = ob[4 * q] | (ob[4 * q + 1] << 8) | (ob[4 * q + 2] << 16) | (ob[4 * q + 3] << 24);
    } else {
        for (int k = 0; k < nx; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) o[3 * k + c] = (unsigned char)ob[3 * k + c];
    }
}

static void linear_table(int n_src, int n_dst, std::vector<LinTab> &tab)
{
    const size_t base = tab.size();
    tab.resize(base + n_dst);
    const double scale = 1.0 / ((double)n_dst / (double)n_src);
    for (int d = 0; d < n_dst; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= n_src - 1) { f = 0.f; s = n_src - 1; }
        LinTab &t = tab[base + d];
        t.ofs = s;
        t.a0 = (short)rintf((1.0f - f) * 2048.0f);
        t.a1 = (short)rintf(f * 2048.0f);
        t.f = f;
    }
}

// ---------------------------------------------------------------------------------------------
// blend plan (host object)
// ---------------------------------------------------------------------------------------------
struct sr_blend_plan {
    sr_ctx *ctx = nullptr;
    int n = 0, cn = 3, canvas_h = 0, canvas_w = 0, levels = 6, wtype = 1, row_begin = 0, row_end = 0;
    int max_nl = 1;
    std::vector<TileDev> tiles;     // per tile
    std::vector<TileDev> classes;   // per weight class (pseudo tiles, cn = 1, g_off = weight levels)
    std::vector<SrWin> tile_rows;   // rows of the input tiles that are read
    std::vector<float> luts;
    size_t arena_floats = 0;
    float *d_arena = nullptr;
    TileDev *d_tiles = nullptr, *d_classes = nullptr;
    TileSrc *d_srcs = nullptr;
    FinalDesc *d_fdesc = nullptr;
    std::vector<FinalDesc> fdesc;
    int4 *d_edge_blocks = nullptr;                      // edge pass work list: x0, y0, shape, first candidate
    int *d_edge_cand = nullptr;
    int n_edge_blocks = 0;
    int *d_cand_off = nullptr, *d_cand_idx = nullptr;   // per 256 x 8 block: candidate tiles (CSR, list order)
    bool weights_ready = false;                         // weight pyramids of the classes are in the arena
    // fused final gather (k_final_fused): its own block tables -- regular blocks of FU_BW x FU_BH, edge blocks of
    // 256 x 16 / 32 x 128 pixels holding every 4 x 4 cell that has a border visit
    bool fused = false;
    int *d_fcand_off = nullptr, *d_fcand_idx = nullptr, *d_fedge_cand = nullptr;
    int4 *d_fedge_blocks = nullptr;
    int n_fedge_blocks = 0;
    // marched zones (k_final_march): work items per tile count, and the regular blocks of k_final_fused that remain
    bool march = false;
    bool down2 = true;               // levels 1 and 2 of full-window u8 RGB tiles in one march (sr_down2.inc); SR_DOWN2=0: two launches
    MarchItem *d_march_items[MARCH_NT + 1] = {nullptr};
    int n_march_items[MARCH_NT + 1] = {0};
    long long n_march_total = 0;
    int *d_freg_list = nullptr, *d_freg_all = nullptr;      // without the marched zones / every regular block (float tiles): 9 ints each
    long long n_freg = 0, n_freg_all = 0;
    RectItem *d_rects = nullptr;                            // what the marched zones leave, as rectangles of cells (k_final_rect)
    int *d_rect_cand = nullptr;
    long long n_rects = 0;
    std::vector<char> sh_srcs, sh_fdesc;                // host shadows of d_srcs / d_fdesc (upload_if_changed)
    CachedTable subset_tabs[4];                         // compacted {TileDev, TileSrc} tables of recent tile subsets
    int subset_next = 0;
    float *d_luts = nullptr;
    // launch extents per level
    int max_w[SR_MAX_LEVELS] = {0}, max_grows[SR_MAX_LEVELS] = {0}, max_rrows[SR_MAX_LEVELS] = {0};
    int cmax_w[SR_MAX_LEVELS] = {0}, cmax_rows[SR_MAX_LEVELS] = {0};
};

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ---------------------------------------------------------------------------------------------
// Work lists of the marched gather (k_final_march, sr_march.inc).  A cell column (4 canvas pixels) / cell row (2 canvas
// rows) is, for one tile, 0 = not touched, 1 = marchable (the cell is an interior visit and neither the level-1 columns /
// rows the lane produces nor their level-2 taps meet a border rule), 2 = touched otherwise.  Between two consecutive class
// changes of any tile the classes are constant, so the canvas falls into rectangles with a fixed list of visiting tiles;
// a rectangle is marched when every tile that touches it is marchable in both directions.  Rectangles lose one cell
// column per side to the halo lanes and are cut into strips of <= MARCH_CELLS cells and segments of <= MARCH_SEG steps
// (an even number: an odd last cell row stays with the block kernel).
// reg_list: the regular blocks that still hold unmarched cells, 9 ints each: block id and, per cell row of the block,
// the mask of its marched cell columns.
// ---------------------------------------------------------------------------------------------
static void plan_march(const sr_blend_plan *P, int nbx_r, int nby_r, std::vector<MarchItem> (&items)[MARCH_NT + 1],
                       std::vector<int> &reg_list, std::vector<unsigned> &cover)
{
    const int rows = P->row_end - P->row_begin, cw = P->canvas_w, n = P->n;
    const int ncx = cw / 4, ncy = rows / 2;                    // whole cells only: ragged ends are border visits
    const int CPB = FU_BH / 2;                                 // cell rows per regular block
    cover.assign((size_t)nbx_r * nby_r * CPB, 0u);                         // [cell row][block column]: cell column bits (1 = marched)
    // the marched kernels address the arena and a tile's pixels with 32-bit byte offsets (buffer instructions)
    bool fits32 = P->arena_floats * sizeof(float) < 0xFFFF0000ull && (unsigned long long)cw * P->cn * 4ull * (2 * 64 + 2) < 0xFFFF0000ull;
    for (int t = 0; t < n && fits32; ++t) fits32 = (unsigned long long)P->tiles[t].h * P->tiles[t].w * P->cn * 4ull < 0x7FFF0000ull;
    // The march is four launches of long work items: it wins on a big canvas (200 MP: 1.12 against 1.22 ms for the block kernel
    // alone) and loses on a small one, where every launch is a few short items deep -- a rank's strip of a world of 4 / 8:
    // 0.39 / 0.27 ms against 0.33 / 0.18 (tools/virtual_scaling.py, profiles/r04_virtual_scaling.json); even at 100 MP.
    // SR_MARCH=2 marches whatever the size, SR_MARCH_MIN_MP moves the threshold (A/B runs).
    const char *env_force = std::getenv("SR_MARCH");               // read per plan: the tests switch it
    const char *env_min = std::getenv("SR_MARCH_MIN_MP");
    const double min_px = (env_min ? atof(env_min) : 90.0) * 1e6;
    const bool big = (env_force && env_force[0] == '2') || (double)rows * (double)cw >= min_px;
    const bool on = P->march && big && fits32 && n <= 128 && ncx >= 3 && ncy >= 2;
    if (on) {
        auto xclass = [&](const TileDev &T, int x0) -> unsigned char {
            const long long lx0 = (long long)x0 - T.x;
            if (lx0 + 4 <= 0 || lx0 >= T.w) return 0;
            if (T.nl < 3 || lx0 < 0 || lx0 + 3 >= T.w) return 2;
            const long long c0 = (lx0 - 1) >> 1, n0 = c0 >> 1;
            if (c0 < 0 || c0 + 3 > T.W[1] - 1 || n0 + 2 > T.W[2] - 1) return 2;
            return 1;
        };
        auto yclass = [&](const TileDev &T, int y0) -> unsigned char {
            const long long ly0 = (long long)y0 - T.y;
            if (ly0 + 2 <= 0 || ly0 >= T.h) return 0;
            if (T.nl < 3 || ly0 < 0 || ly0 + 1 >= T.h) return 2;
            const long long r0 = (ly0 - 1) >> 1;
            if (r0 < 1 || r0 + 2 > T.H[1] - 1 || ((r0 + 1) >> 1) + 2 > T.H[2] - 1) return 2;
            return 1;
        };
        std::vector<std::vector<unsigned char>> xc(n, std::vector<unsigned char>(ncx)), yc(n, std::vector<unsigned char>(ncy));
        std::vector<int> xb{0, ncx}, yb{0, ncy};
        for (int t = 0; t < n; ++t) {
            const TileDev &T = P->tiles[t];
            for (int c = 0; c < ncx; ++c) {
                xc[t][c] = xclass(T, 4 * c);
                if (c && xc[t][c] != xc[t][c - 1]) xb.push_back(c);
            }
            for (int c = 0; c < ncy; ++c) {
                yc[t][c] = yclass(T, P->row_begin + 2 * c);
                if (c && yc[t][c] != yc[t][c - 1]) yb.push_back(c);
            }
        }
        std::sort(xb.begin(), xb.end());
        xb.erase(std::unique(xb.begin(), xb.end()), xb.end());
        std::sort(yb.begin(), yb.end());
        yb.erase(std::unique(yb.begin(), yb.end()), yb.end());
        struct Strip { int ca, cb, ya, ye, nt, tile[4]; };
        std::vector<Strip> strips;
        std::vector<int> vis, run_vis;
        for (size_t iy = 0; iy + 1 < yb.size(); ++iy) {
            const int ya = yb[iy], ye = ya + (yb[iy + 1] - ya) / 2 * 2;      // an even number of steps (the loop is unrolled by two)
            if (ye - ya < 2) continue;
            // runs of consecutive x intervals with the same (valid) tile list
            int run_a = -1, run_b = -1;
            auto flush = [&]() {
                if (run_a < 0) return;
                const int ua = run_a + 1, ub = run_b - 1;          // one cell column per side goes to the halo lanes
                const int nu = ub - ua, nt = (int)run_vis.size();
                if (nu >= 1) {
                    const int ns = (nu + MARCH_CELLS - 1) / MARCH_CELLS;
                    for (int si = 0; si < ns; ++si) {
                        const int ca = ua + (int)((long long)nu * si / ns), cb = ua + (int)((long long)nu * (si + 1) / ns);
                        Strip st;
                        st.ca = ca; st.cb = cb; st.ya = ya; st.ye = ye; st.nt = nt;
                        for (int k = 0; k < 4; ++k) st.tile[k] = k < nt ? run_vis[k] : 0;
                        strips.push_back(st);
                    }
                    for (int cy = ya; cy < ye; ++cy)
                        for (int c = ua; c < ub; ++c) cover[(size_t)cy * nbx_r + c / 32] |= 1u << (c & 31);
                }
                run_a = -1;
            };
            for (size_t ix = 0; ix + 1 < xb.size(); ++ix) {
                vis.clear();
                bool ok = true;
                for (int t = 0; t < n && ok; ++t) {
                    const unsigned char cx = xc[t][xb[ix]], cy = yc[t][yb[iy]];
                    if (cx == 0 || cy == 0) continue;
                    if (cx == 1 && cy == 1) vis.push_back(t);
                    else ok = false;
                }
                ok = ok && !vis.empty() && (int)vis.size() <= MARCH_NT;
                if (ok && run_a >= 0 && vis == run_vis) {
                    run_b = xb[ix + 1];
                    continue;
                }
                flush();
                if (ok) {
                    run_a = xb[ix];
                    run_b = xb[ix + 1];
                    run_vis = vis;
                }
            }
            flush();
        }
        // Segment length per tile count: as long as MARCH_SEG steps where that still leaves every wave slot of the GPU a few
        // items (the warm-up of an item costs about two steps), shorter where a list is small -- a short list of long items
        // is a latency-bound launch of one or two rounds.
        long long steps_of[MARCH_NT + 1] = {0}, cells_of[MARCH_NT + 1] = {0};
        for (const Strip &st : strips) {
            steps_of[st.nt] += st.ye - st.ya;
            cells_of[st.nt] += (long long)(st.ye - st.ya) * (st.cb - st.ca);
        }
        for (int nt = 1; nt <= MARCH_NT; ++nt) {
            if (!steps_of[nt]) continue;
            // Rounds per list, measured (profiles/r04_e_gather_sweep.txt): four for the 1-tile list, three for the 2-tile list, two
            // for the short 3- / 4-tile lists (an item's warm-up is ~2.5 steps: at six rounds the 4-tile list was cut into 8-step
            // items); the tapered end of a list (below) is what makes long items affordable.
            // SR_MARCH_ROUNDS="r1,r2,r4" / SR_MARCH_SEG_MAX: A/B runs.
            double rounds = nt >= 3 ? MARCH_ROUNDS_N : (nt == 2 ? MARCH_ROUNDS_2 : MARCH_ROUNDS);
            if (const char *e_r = std::getenv("SR_MARCH_ROUNDS")) {
                double r[3] = {0, 0, 0};
                const int got = sscanf(e_r, "%lf,%lf,%lf", &r[0], &r[1], &r[2]);
                const int k = nt == 1 ? 0 : (nt == 2 ? 1 : 2);
                if (got >= 1 && r[std::min(k, got - 1)] > 0) rounds = r[std::min(k, got - 1)];
            }
            const char *e_s = std::getenv("SR_MARCH_SEG_MAX");
            const int seg_max = e_s && atoi(e_s) >= 8 ? std::min(atoi(e_s), 64) : MARCH_SEG;
            const long long want_items = std::max<long long>((long long)((double)(P->ctx->num_cu * 8 / nt) * rounds), 1);   // rounds at two waves per SIMD
            int seg = (int)std::min<long long>(seg_max, std::max<long long>(8, steps_of[nt] / want_items));
            seg = seg / 2 * 2;
            // The items of a list run in list order, a few rounds of them: the last round leaves the GPU emptier and emptier
            // while its long items finish (half an item's duration per launch, ~30 us of march1's 300).  So the list ends
            // with short items: the last `tail` strip-steps (about one round of full-length items) are cut into segments of
            // half the length, the last quarter of those into the shortest ones (8 steps: an item's warm-up is ~2).
            // SR_MARCH_TAIL=0: uniform segments (A/B runs).  Which segment a canvas row falls into changes no value.
            const int taper = std::getenv("SR_MARCH_TAIL") ? atoi(std::getenv("SR_MARCH_TAIL")) : 1;   // read per plan
            const long long slots = std::max<long long>((long long)P->ctx->num_cu * 8 / nt, 1);
            const long long tail = taper && seg > 8 ? std::min<long long>(slots * seg * (taper == 3 ? 2 : 1), steps_of[nt] / (taper == 3 ? 2 : 3)) : 0;
            long long done = 0;
            for (const Strip &st : strips) {
                if (st.nt != nt) continue;
                for (int sy = st.ya; sy < st.ye;) {
                    const long long left = steps_of[nt] - done;
                    int sg = seg;
                    if (taper == 2) {                                     // (A/B) three levels: 1/2, 1/4, shortest
                        if (left <= tail / 8) sg = 8;
                        else if (left <= tail / 2) sg = std::max(8, seg / 8 * 2);
                        else if (left <= tail) sg = std::max(8, seg / 4 * 2);
                    } else {
                        if (left <= tail / 4) sg = 8;
                        else if (left <= tail) sg = std::max(8, seg / 4 * 2);
                    }
                    MarchItem it;
                    memset(&it, 0, sizeof(it));
                    it.x0 = 4 * (st.ca - 1);
                    it.y0 = P->row_begin + 2 * sy;
                    it.ncell = st.cb - st.ca;
                    it.nstep = std::min(sg, st.ye - sy);
                    for (int k = 0; k < nt; ++k) it.tile[k] = st.tile[k];
                    items[nt].push_back(it);
                    sy += it.nstep;
                    done += it.nstep;
                }
            }
            if (std::getenv("SR_MARCH_STATS"))
                fprintf(stderr, "[march] %d-tile zones: %lld strip-steps, %lld cells (%.2f %% of %d x %d), segments of %d steps, %zu items\n", nt,
                        steps_of[nt], cells_of[nt], 100.0 * (double)cells_of[nt] / ((double)ncx * ncy), ncx, ncy, seg, items[nt].size());
        }
    }
    // the regular blocks that still hold unmarched cells: block id + the marched cell columns of each of its CPB cell rows
    static_assert(FU_BW == 128 && FU_BH == 16, "one mask word per cell row of a regular block, one bit per cell column");
    for (int by = 0; by < nby_r; ++by)
        for (int bx = 0; bx < nbx_r; ++bx) {
            const int cells = std::min(32, (cw - bx * FU_BW + 3) / 4);
            const unsigned all = cells >= 32 ? 0xFFFFFFFFu : ((1u << cells) - 1u);
            const int crows = std::min(CPB, (rows - by * FU_BH + 1) / 2);
            unsigned m[8];
            bool dead = true;
            for (int cy = 0; cy < CPB; ++cy) {
                m[cy] = cy < crows ? cover[(size_t)(by * CPB + cy) * nbx_r + bx] : 0xFFFFFFFFu;
                if (cy < crows && (m[cy] & all) != all) dead = false;
            }
            if (dead) continue;                                     // every cell of the block is marched
            reg_list.push_back(by * nbx_r + bx);
            for (int cy = 0; cy < CPB; ++cy) reg_list.push_back((int)m[cy]);
        }
    if (std::getenv("SR_MARCH_STATS")) {
        long long left = 0;
        for (size_t i = 0; i < reg_list.size(); i += 9)
            for (int cy = 0; cy < CPB; ++cy) left += 32 - __builtin_popcount((unsigned)reg_list[i + 1 + cy]);
        fprintf(stderr, "[march] block kernel: %zu regular blocks of %d x %d with %lld unmarched cell slots (%.2f %% of the canvas cells)\n",
                reg_list.size() / 9, nbx_r, nby_r, left, 100.0 * (double)left / ((double)ncx * ncy));
    }
}

// What the marched zones leave, cut into rectangles of cells for k_final_rect: every cell row's runs of unmarched cells --
// a narrow run (<= 8 cells: a band along a vertical tile edge) whole, a wide one in pieces that end on multiples of 32 cells
// -- stacked downwards while the next row holds the same piece and the rectangle still fits 256 threads and the LDS window.
// cover: plan_march's bitmap (row pitch nbx_r words); ragged cells at the right / bottom end are unmarched cells like any other.
static void plan_rects(const sr_blend_plan *P, int nbx_r, const std::vector<unsigned> &cover, std::vector<RectItem> &rects,
                       std::vector<int> &rcand)
{
    const int rows = P->row_end - P->row_begin, cw = P->canvas_w;
    const int ncxp = (cw + 3) / 4, ncyp = (rows + 1) / 2;
    const char *e_cells = std::getenv("SR_RECT_CELLS");                   // A/B runs: most cells per rectangle (<= 256 threads)
    const int max_cells = e_cells && atoi(e_cells) >= 32 ? std::min(atoi(e_cells), 256) : 256;
    const bool colmajor_on = !(std::getenv("SR_RECT_COLMAJOR") && std::getenv("SR_RECT_COLMAJOR")[0] == '0');
    // window pitch: two pixel pairs more than needed, so that consecutive rows start 36 (not 32) dwords apart for a band 5 cells
    // wide -- column-major lanes read the same columns of consecutive rows (SR_RECT_PAD=0: the bare pitch, A/B runs)
    const int pad = (std::getenv("SR_RECT_PAD") && std::getenv("SR_RECT_PAD")[0] == '0') ? 0 : 2;
    auto hmax = [max_cells, pad](int w) {
        int h = std::max(max_cells / w, 1);
        while (h > 1 && rect_rows(h) * (rect_lp(w) + pad) > FU_PLANE) --h;
        return h;
    };
    struct Open { int ya, h; };
    std::map<std::pair<int, int>, Open> open;                     // (first cell column, end) -> rectangle still growing
    auto emit = [&](int xa, int xb, int ya, int h) {
        RectItem it;
        it.x = 4 * xa;
        it.y = P->row_begin + 2 * ya;
        it.w = xb - xa;
        it.h = h;
        it.cand = (int)rcand.size();
        const long long bx0 = it.x, bx1 = std::min<long long>(bx0 + 4ll * it.w, cw);
        const long long by0 = it.y, by1 = std::min<long long>(by0 + 2ll * it.h, P->row_end);
        for (int t = 0; t < P->n; ++t) {
            const TileDev &T = P->tiles[t];
            if (T.x < bx1 && (long long)T.x + T.w > bx0 && T.y < by1 && (long long)T.y + T.h > by0) rcand.push_back(t);
        }
        it.ncand = (int)rcand.size() - it.cand;
        it.lp = rect_lp(it.w) + pad;
        it.colmajor = (colmajor_on && it.h > it.w) ? 1 : 0;
        rects.push_back(it);
    };
    std::vector<std::pair<int, int>> segs;
    std::map<std::pair<int, int>, std::pair<int, int>> wide;      // wide run (first cell, end) -> (piece width, last row seen)
    const bool wide_on = !(std::getenv("SR_RECT_WIDE") && std::getenv("SR_RECT_WIDE")[0] == '0');
    auto unmarched = [&](int cy, int cx) { return !((cover[(size_t)cy * nbx_r + (cx >> 5)] >> (cx & 31)) & 1u); };
    auto run_is = [&](int cy, int xa, int xe) {                   // row cy holds exactly the run [xa, xe) of unmarched cells
        if (xa > 0 && unmarched(cy, xa - 1)) return false;
        if (xe < ncxp && unmarched(cy, xe)) return false;
        for (int c = xa; c < xe; ++c)
            if (!unmarched(cy, c)) return false;
        return true;
    };
    for (int cy = 0; cy < ncyp; ++cy) {
        segs.clear();
        const unsigned *row = cover.data() + (size_t)cy * nbx_r;
        for (int cx = 0; cx < ncxp;) {
            const unsigned wd = row[cx >> 5];
            if ((cx & 31) == 0 && wd == 0xFFFFFFFFu) { cx += 32; continue; }
            if ((wd >> (cx & 31)) & 1u) { ++cx; continue; }
            int xe = cx;
            while (xe < ncxp && !((row[xe >> 5] >> (xe & 31)) & 1u)) ++xe;
            if (xe - cx <= 8) segs.emplace_back(cx, xe);
            else {
                // a wide run: pieces as wide as the band's height allows (a band 3-4 cell rows high along a horizontal tile
                // edge: 60 cells x 4 rows instead of 32 x 4 -- fewer, fuller items); the width is chosen where the band starts
                // and kept for its rows, so that the pieces of consecutive rows stack
                int pw = 32;
                auto rec = wide.find({cx, xe});
                if (rec != wide.end() && rec->second.second == cy - 1) {
                    pw = rec->second.first;
                    rec->second.second = cy;
                } else {
                    int H = 1;
                    while (H < 9 && cy + H < ncyp && run_is(cy + H, cx, xe)) ++H;
                    if (wide_on && H <= 8) {
                        pw = std::min(64, 256 / H) / 4 * 4;
                        while (pw > 32 && rect_rows(H) * (rect_lp(pw) + pad) > FU_PLANE) pw -= 4;
                        pw = std::max(pw, 32);
                    }
                    wide[{cx, xe}] = {pw, cy};
                }
                for (int a = cx; a < xe;) {
                    const int b = std::min(xe, (a / pw + 1) * pw);
                    segs.emplace_back(a, b);
                    a = b;
                }
            }
            cx = xe;
        }
        std::map<std::pair<int, int>, Open> next;
        for (const auto &sg : segs) {
            auto f = open.find(sg);
            if (f != open.end() && f->second.h < hmax(sg.second - sg.first)) {
                next[sg] = Open{f->second.ya, f->second.h + 1};
                open.erase(f);
            } else {
                next[sg] = Open{cy, 1};                               // (a full one stays in `open` and is closed below)
            }
        }
        for (const auto &o : open) emit(o.first.first, o.first.second, o.second.ya, o.second.h);
        open.swap(next);
    }
    for (const auto &o : open) emit(o.first.first, o.first.second, o.second.ya, o.second.h);
    if (std::getenv("SR_MARCH_STATS")) {
        long long cells = 0;
        for (const RectItem &r : rects) cells += (long long)r.w * r.h;
        fprintf(stderr, "[march] rectangles of the remainder: %zu items, %lld cells (%.1f per item), %zu candidate visits\n", rects.size(), cells,
                rects.empty() ? 0.0 : (double)cells / (double)rects.size(), rcand.size());
    }
}

bool plan_describe(const sr_blend_plan *p, sr_ctx **ctx, int *n, int *cn)
{
    if (!plan_is_live(p)) return false;
    if (ctx) *ctx = p->ctx;
    if (n) *n = p->n;
    if (cn) *cn = p->cn;
    return true;
}

extern "C" {

int sr_device_count(int *count)
{
    if (!count) return sr_set_error(SR_ERR_INVALID_ARG, "sr_device_count: null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return sr_set_error(SR_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return SR_OK;
}

static int ctx_create_impl(int device_id, void *stream, bool adopt, sr_ctx **out)
{
    if (!out) return sr_set_error(SR_ERR_INVALID_ARG, "sr_ctx_create: null out");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return sr_set_error(SR_ERR_HIP, "sr_ctx_create: no HIP device available (%s)",
                            e == hipSuccess ? "count is 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n)
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_ctx_create: device %d out of range (0..%d)", device_id, n - 1);
    int prev = 0;
    HIPCHK(hipGetDevice(&prev));
    HIPCHK(hipSetDevice(device_id));
    sr_ctx *c = new sr_ctx();
    c->device = device_id;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) c->num_cu = cus;
    }
    if (adopt) {
        c->stream = (hipStream_t)stream;
        c->own_stream = false;
    } else {
        hipError_t e2 = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e2 != hipSuccess) {
            delete c;
            (void)hipSetDevice(prev);
            return sr_set_error(SR_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e2));
        }
        c->own_stream = true;
    }
    (void)hipSetDevice(prev);
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        g_live_ctx.insert(c);
    }
    *out = c;
    return SR_OK;
}

int sr_ctx_create(int device_id, sr_ctx **out) { return ctx_create_impl(device_id, nullptr, false, out); }

int sr_ctx_create_on_stream(int device_id, void *hip_stream, sr_ctx **out)
{
    return ctx_create_impl(device_id, hip_stream, true, out);
}

static std::vector<sr_blend_plan *> plans_of(sr_ctx *ctx);

int sr_ctx_destroy(sr_ctx *ctx)
{
    if (!ctx_is_live(ctx)) return SR_OK;
    for (sr_blend_plan *p : plans_of(ctx)) sr_blend_plan_destroy(p);
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        g_live_ctx.erase(ctx);
    }
    {
        Guard g(ctx);
        (void)hipStreamSynchronize(ctx->stream);
        for (auto &p : ctx->prof_pairs) {
            (void)hipEventDestroy(p.a);
            (void)hipEventDestroy(p.b);
        }
        for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
        if (ctx->scratch) (void)hipFree(ctx->scratch);
        if (ctx->gray_planes) (void)hipFree(ctx->gray_planes);
        if (ctx->extract_tab.d) (void)hipFree(ctx->extract_tab.d);
        if (ctx->resize_tab.d) (void)hipFree(ctx->resize_tab.d);
        if (ctx->cubic_tab.d) (void)hipFree(ctx->cubic_tab.d);
        for (int i = 0; i < 2; ++i) {
            if (ctx->side[i]) {
                (void)hipStreamSynchronize(ctx->side[i]);
                (void)hipStreamDestroy(ctx->side[i]);
            }
            if (ctx->side_join[i]) (void)hipEventDestroy(ctx->side_join[i]);
        }
        if (ctx->side_fork) (void)hipEventDestroy(ctx->side_fork);
        if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
    return SR_OK;
}

int sr_ctx_sync(sr_ctx *ctx)
{
    CTX_ENTER(ctx);
    HIPCHK(stream_sync(ctx));
    return SR_OK;
}

int sr_dev_alloc(sr_ctx *ctx, size_t bytes, void **d_ptr)
{
    CTX_ENTER(ctx);
    if (!d_ptr) return sr_set_error(SR_ERR_INVALID_ARG, "sr_dev_alloc: null out");
    *d_ptr = nullptr;
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(d_ptr, bytes);
    if (e == hipErrorOutOfMemory) return sr_set_error(SR_ERR_OOM, "sr_dev_alloc: out of device memory (%zu B)", bytes);
    if (e != hipSuccess) return sr_set_error(SR_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e));
    return SR_OK;
}

int sr_dev_free(sr_ctx *ctx, void *d_ptr)
{
    CTX_ENTER(ctx);
    if (!d_ptr) return SR_OK;
    HIPCHK(stream_sync(ctx));
    HIPCHK(hipFree(d_ptr));
    return SR_OK;
}

int sr_memcpy_h2d(sr_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
    CTX_ENTER(ctx);
    if (bytes == 0) return SR_OK;
    HIPCHK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(stream_sync(ctx));
    return SR_OK;
}

int sr_memcpy_d2h(sr_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
    CTX_ENTER(ctx);
    if (bytes == 0) return SR_OK;
    HIPCHK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(stream_sync(ctx));
    return SR_OK;
}

int sr_memcpy_d2d(sr_ctx *ctx, void *d_dst, const void *d_src, size_t bytes)
{
    CTX_ENTER(ctx);
    if (bytes == 0) return SR_OK;
    HIPCHK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return SR_OK;
}

int sr_memset_d(sr_ctx *ctx, void *d_dst, int value, size_t bytes)
{
    CTX_ENTER(ctx);
    if (bytes == 0) return SR_OK;
    HIPCHK(hipMemsetAsync(d_dst, value, bytes, ctx->stream));
    return SR_OK;
}

int sr_prof_enable(sr_ctx *ctx, int on)
{
    CTX_ENTER(ctx);
    ctx->prof = on != 0;
    return SR_OK;
}

int sr_prof_select(sr_ctx *ctx, const char *name)
{
    CTX_ENTER(ctx);
    ctx->prof_only = name ? name : "";
    return SR_OK;
}

int sr_prof_reset(sr_ctx *ctx)
{
    CTX_ENTER(ctx);
    HIPCHK(stream_sync(ctx));
    for (auto &p : ctx->prof_pairs) {
        ctx->ev_pool.push_back(p.a);
        ctx->ev_pool.push_back(p.b);
    }
    ctx->prof_pairs.clear();
    return SR_OK;
}

int sr_prof_get(sr_ctx *ctx, sr_prof_record *h_records, int cap, int *n)
{
    CTX_ENTER(ctx);
    if (!n) return sr_set_error(SR_ERR_INVALID_ARG, "sr_prof_get: null n");
    HIPCHK(stream_sync(ctx));
    std::vector<double> ms(ctx->prof_names.size(), 0.0);
    std::vector<int64_t> cnt(ctx->prof_names.size(), 0);
    for (auto &p : ctx->prof_pairs) {
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, p.a, p.b));
        ms[p.name_id] += t;
        cnt[p.name_id] += 1;
    }
    int k = 0;
    for (size_t i = 0; i < ctx->prof_names.size(); ++i) {
        if (cnt[i] == 0) continue;
        if (h_records && k < cap) {
            memset(&h_records[k], 0, sizeof(sr_prof_record));
            strncpy(h_records[k].name, ctx->prof_names[i].c_str(), sizeof(h_records[k].name) - 1);
            h_records[k].ms = ms[i];
            h_records[k].launches = cnt[i];
        }
        ++k;
    }
    *n = k;
    return SR_OK;
}

// ---- tile extract ------------------------------------------------------------------------------
static int extract_impl(sr_ctx *ctx, const uint8_t *d_img, int img_h, int img_w, int cn, int64_t img_stride,
                        const std::vector<ExtractDesc> &descs, int pad_mode, const char *name)
{
    const int n = (int)descs.size();
    if (n == 0) return SR_OK;
    int mw = 0, mh = 0;
    for (auto &d : descs) {
        if (d.x < 0 || d.y < 0 || d.w <= 0 || d.h <= 0 || d.x + d.w > img_w || d.y + d.h > img_h)
            return sr_set_error(SR_ERR_SHAPE, "%s: tile (%d,%d,%d,%d) outside the %dx%d image", name, d.x, d.y, d.w,
                                d.h, img_w, img_h);
        mw = std::max(mw, d.out_w);
        mh = std::max(mh, d.out_h);
    }
    HIPCHK(upload_cached(ctx, ctx->extract_tab, descs.data(), sizeof(ExtractDesc) * n));
    const void *scr = ctx->extract_tab.d;
    {
        ProfScope ps(ctx, name);
        const long long chunks = ((long long)mw * cn + 15) / 16;
        dim3 grid((unsigned)((chunks + 63) / 64), (mh + 3) / 4, n), block(64, 4);
        hipLaunchKernelGGL(k_tile_extract, grid, block, 0, ctx->stream, d_img, (long long)img_stride, cn,
                           (const ExtractDesc *)scr, pad_mode);
    }
    return check_launch(name);
}

int sr_tile_extract_pad(sr_ctx *ctx, const uint8_t *d_img, int img_h, int img_w, int cn, int64_t img_stride,
                        const int *h_xywh, int n, int block_size, int pad_mode, uint8_t *d_tiles)
{
    CTX_ENTER(ctx);
    if (!d_img || !h_xywh || !d_tiles || n < 0 || cn < 1 || cn > 4 || block_size <= 0 || pad_mode < 0 || pad_mode > 3)
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_tile_extract_pad: bad arguments");
    std::vector<ExtractDesc> descs(n);
    for (int i = 0; i < n; ++i) {
        ExtractDesc &d = descs[i];
        d.x = h_xywh[4 * i];
        d.y = h_xywh[4 * i + 1];
        d.w = h_xywh[4 * i + 2];
        d.h = h_xywh[4 * i + 3];
        if (d.w > block_size || d.h > block_size)
            return sr_set_error(SR_ERR_SHAPE, "sr_tile_extract_pad: tile %d larger than block_size", i);
        d.dst = d_tiles + (size_t)i * block_size * block_size * cn;
        d.dstride = (long long)block_size * cn;
        d.out_w = d.out_h = block_size;
    }
    return extract_impl(ctx, d_img, img_h, img_w, cn, img_stride, descs, pad_mode, "tile_extract_pad");
}

int sr_tile_extract(sr_ctx *ctx, const uint8_t *d_img, int img_h, int img_w, int cn, int64_t img_stride,
                    const int *h_xywh, int n, void *const *h_d_tiles, const int64_t *h_tile_strides)
{
    CTX_ENTER(ctx);
    if (!d_img || !h_xywh || !h_d_tiles || !h_tile_strides || n < 0 || cn < 1 || cn > 4)
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_tile_extract: bad arguments");
    std::vector<ExtractDesc> descs(n);
    for (int i = 0; i < n; ++i) {
        ExtractDesc &d = descs[i];
        d.x = h_xywh[4 * i];
        d.y = h_xywh[4 * i + 1];
        d.w = h_xywh[4 * i + 2];
        d.h = h_xywh[4 * i + 3];
        d.dst = (unsigned char *)h_d_tiles[i];
        d.dstride = h_tile_strides[i];
        d.out_w = d.w;
        d.out_h = d.h;
    }
    return extract_impl(ctx, d_img, img_h, img_w, cn, img_stride, descs, PAD_REPLICATE, "tile_extract");
}

// ---- dense pyramid primitives --------------------------------------------------------------------
int sr_pyr_down(sr_ctx *ctx, const float *d_src, int h, int w, int cn, float *d_dst)
{
    CTX_ENTER(ctx);
    if (!d_src || !d_dst || h < 1 || w < 1 || cn < 1 || cn > 4)
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_pyr_down: bad arguments");
    const int ho = (h + 1) / 2, wo = (w + 1) / 2;
    {
        ProfScope ps(ctx, "pyr_down_hwc");
        dim3 grid((wo + 63) / 64, (ho + 3) / 4), block(64, 4);
        hipLaunchKernelGGL(k_pyr_down_hwc, grid, block, 0, ctx->stream, d_src, h, w, cn, d_dst, ho, wo);
    }
    return check_launch("pyr_down_hwc");
}

static int pyr_up_impl(sr_ctx *ctx, int mode, const float *d_src, int hs, int ws, int cn, const float *d_a,
                       float *d_dst, int hd, int wd)
{
    if (!d_src || !d_dst || hs < 1 || ws < 1 || cn < 1 || cn > 4 || (mode && !d_a))
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_pyr_up: bad arguments");
    if ((wd != 2 * ws && wd != 2 * ws - 1) || (hd != 2 * hs && hd != 2 * hs - 1) || wd < 1 || hd < 1)
        return sr_set_error(SR_ERR_SHAPE, "sr_pyr_up: dst %dx%d is not 2x(-1) of src %dx%d", wd, hd, ws, hs);
    {
        ProfScope ps(ctx, "pyr_up_hwc");
        dim3 grid((wd + 63) / 64, (hd + 3) / 4), block(64, 4);
        if (mode == 0) hipLaunchKernelGGL(k_pyr_up_hwc<0>, grid, block, 0, ctx->stream, d_src, hs, ws, cn, d_a, d_dst, hd, wd);
        else if (mode == 1) hipLaunchKernelGGL(k_pyr_up_hwc<1>, grid, block, 0, ctx->stream, d_src, hs, ws, cn, d_a, d_dst, hd, wd);
        else hipLaunchKernelGGL(k_pyr_up_hwc<2>, grid, block, 0, ctx->stream, d_src, hs, ws, cn, d_a, d_dst, hd, wd);
    }
    return check_launch("pyr_up_hwc");
}

int sr_pyr_up(sr_ctx *ctx, const float *d_src, int hs, int ws, int cn, float *d_dst, int hd, int wd)
{
    CTX_ENTER(ctx);
    return pyr_up_impl(ctx, 0, d_src, hs, ws, cn, nullptr, d_dst, hd, wd);
}

int sr_pyr_up_sub(sr_ctx *ctx, const float *d_a, int h, int w, int cn, const float *d_b, float *d_out)
{
    CTX_ENTER(ctx);
    return pyr_up_impl(ctx, 1, d_b, (h + 1) / 2, (w + 1) / 2, cn, d_a, d_out, h, w);
}

int sr_pyr_up_add(sr_ctx *ctx, const float *d_a, int h, int w, int cn, const float *d_b, float *d_out)
{
    CTX_ENTER(ctx);
    return pyr_up_impl(ctx, 2, d_b, (h + 1) / 2, (w + 1) / 2, cn, d_a, d_out, h, w);
}

// ---- blend plan ------------------------------------------------------------------------------------
static std::vector<sr_blend_plan *> plans_of(sr_ctx *ctx)
{
    std::lock_guard<std::mutex> lk(g_reg_mu);
    std::vector<sr_blend_plan *> out;
    for (const void *p : g_live_plan)
        if (((const sr_blend_plan *)p)->ctx == ctx) out.push_back((sr_blend_plan *)p);
    return out;
}

int sr_blend_plan_destroy(sr_blend_plan *plan)
{
    if (!plan) return SR_OK;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        if (!g_live_plan.erase(plan)) return SR_OK;          // already destroyed (with its context)
    }
    sr_ctx *ctx = plan->ctx;
    {
        Guard g(ctx);
        (void)hipStreamSynchronize(ctx->stream);
        if (plan->d_arena) (void)hipFree(plan->d_arena);
        if (plan->d_tiles) (void)hipFree(plan->d_tiles);
        if (plan->d_classes) (void)hipFree(plan->d_classes);
        if (plan->d_srcs) (void)hipFree(plan->d_srcs);
        if (plan->d_fdesc) (void)hipFree(plan->d_fdesc);
        if (plan->d_edge_blocks) (void)hipFree(plan->d_edge_blocks);
        if (plan->d_edge_cand) (void)hipFree(plan->d_edge_cand);
        if (plan->d_cand_off) (void)hipFree(plan->d_cand_off);
        if (plan->d_fcand_off) (void)hipFree(plan->d_fcand_off);
        if (plan->d_fcand_idx) (void)hipFree(plan->d_fcand_idx);
        if (plan->d_fedge_blocks) (void)hipFree(plan->d_fedge_blocks);
        if (plan->d_fedge_cand) (void)hipFree(plan->d_fedge_cand);
        if (plan->d_freg_list) (void)hipFree(plan->d_freg_list);
        if (plan->d_freg_all) (void)hipFree(plan->d_freg_all);
        if (plan->d_rects) (void)hipFree(plan->d_rects);
        if (plan->d_rect_cand) (void)hipFree(plan->d_rect_cand);
        for (int k = 0; k <= MARCH_NT; ++k)
            if (plan->d_march_items[k]) (void)hipFree(plan->d_march_items[k]);
        for (auto &t : plan->subset_tabs)
            if (t.d) (void)hipFree(t.d);
        if (plan->d_cand_idx) (void)hipFree(plan->d_cand_idx);
        if (plan->d_luts) (void)hipFree(plan->d_luts);
    }
    delete plan;
    return SR_OK;
}

int sr_blend_plan_create(sr_ctx *ctx, const sr_tile_rect *h_tiles, int n, int cn, int canvas_h, int canvas_w,
                         int levels, int weight_type, int row_begin, int row_end, sr_blend_plan **out)
{
    CTX_ENTER(ctx);
    if (!out) return sr_set_error(SR_ERR_INVALID_ARG, "sr_blend_plan_create: null out");
    *out = nullptr;
    if (!h_tiles || n < 1 || n > 65535) return sr_set_error(SR_ERR_INVALID_ARG, "sr_blend_plan_create: need 1..65535 tiles");
    if (cn < 1 || cn > 4) return sr_set_error(SR_ERR_INVALID_ARG, "sr_blend_plan_create: channels must be 1..4");
    if (canvas_h < 1 || canvas_w < 1) return sr_set_error(SR_ERR_SHAPE, "sr_blend_plan_create: empty canvas");
    if (levels < 1) return sr_set_error(SR_ERR_INVALID_ARG, "sr_blend_plan_create: levels must be >= 1");
    if (weight_type < 0 || weight_type > SR_W_ONES) return sr_set_error(SR_ERR_INVALID_ARG, "sr_blend_plan_create: bad weight type");
    row_begin = std::max(row_begin, 0);
    row_end = std::min(row_end, canvas_h);
    if (row_begin > row_end) row_begin = row_end;

    sr_blend_plan *P = new sr_blend_plan();
    P->ctx = ctx;
    P->n = n;
    P->cn = cn;
    P->canvas_h = canvas_h;
    P->canvas_w = canvas_w;
    P->levels = std::min(levels, SR_MAX_LEVELS);
    P->wtype = weight_type;
    P->row_begin = row_begin;
    P->row_end = row_end;
    P->tiles.resize(n);
    P->tile_rows.resize(n);
    {
        const char *env = std::getenv("SR_FUSED_FINAL");
        P->fused = (cn == 3 || cn == 1) && !(env && env[0] == '0');     // SR_FUSED_FINAL=0: the unfused pair (A/B runs)
        const char *env2 = std::getenv("SR_MARCH");
        P->march = P->fused && !(env2 && env2[0] == '0');               // SR_MARCH=0: every zone through k_final_fused (A/B runs)
        const char *env3 = std::getenv("SR_DOWN2");
        P->down2 = !(env3 && env3[0] == '0');
    }

    std::map<std::pair<int, int>, int> cls_of;
    std::vector<SrTileLevels> lv(n);
    size_t off = 0;
    for (int t = 0; t < n; ++t) {
        const sr_tile_rect &r = h_tiles[t];
        if (r.w < 1 || r.h < 1 || r.x < 0 || r.y < 0) {
            delete P;
            return sr_set_error(SR_ERR_INVALID_ARG, "sr_blend_plan_create: tile %d has bad rectangle (%d,%d,%d,%d)", t, r.x,
                                r.y, r.w, r.h);
        }
        if (std::min(r.w, r.h) < 8 && weight_type != SR_W_ONES) {
            delete P;
            return sr_set_error(SR_ERR_INVALID_ARG,
                                "sr_blend_plan_create: tile %d is %dx%d; min side < 8 gives the reference a zero "
                                "feather width (NaN weights)", t, r.w, r.h);
        }
        sr_plan_windows(r.h, r.w, r.y, P->levels, row_begin, row_end, canvas_h, &lv[t]);
        TileDev &T = P->tiles[t];
        memset(&T, 0, sizeof(T));
        T.h = r.h;
        T.w = r.w;
        T.x = r.x;
        T.y = r.y;
        T.nl = lv[t].nl;
        T.fw = std::max(std::min(r.w, r.h) / 8, 1);
        P->max_nl = std::max(P->max_nl, T.nl);
        auto key = std::make_pair(r.h, r.w);
        auto it = cls_of.find(key);
        if (it == cls_of.end()) {
            const int id = (int)P->classes.size();
            cls_of[key] = id;
            TileDev C;
            memset(&C, 0, sizeof(C));
            C.h = r.h;
            C.w = r.w;
            C.nl = T.nl;
            C.fw = T.fw;
            C.lut_off = (int)P->luts.size();
            P->luts.resize(P->luts.size() + T.fw + 1);
            int rc = sr_weight_lut(T.fw, weight_type, P->luts.data() + C.lut_off);
            if (rc) {
                delete P;
                return rc;
            }
            for (int i = 0; i < T.nl; ++i) {
                C.H[i] = lv[t].H[i];
                C.W[i] = lv[t].W[i];
                C.P[i] = round_up(C.W[i], 32);
                C.g0[i] = C.g1[i] = 0;
                if (i >= 1) {
                    C.g_off[i] = (long long)off;
                    off += (size_t)C.H[i] * C.P[i];
                }
            }
            P->classes.push_back(C);
            T.cls = id;
        } else {
            T.cls = it->second;
        }
        TileDev &C = P->classes[T.cls];
        T.lut_off = C.lut_off;
        for (int i = 0; i < T.nl; ++i) {
            T.H[i] = lv[t].H[i];
            T.W[i] = lv[t].W[i];
            T.P[i] = round_up(T.W[i], 32);
            T.g0[i] = lv[t].gw[i].a;
            T.g1[i] = lv[t].gw[i].b;
            T.r0[i] = lv[t].rw[i].a;
            T.r1[i] = lv[t].rw[i].b;
            T.w_off[i] = C.g_off[i];
            if (i >= 1) {
                // weight level i must cover every member tile's G window (hull)
                if (T.g0[i] < T.g1[i]) {
                    if (C.g0[i] >= C.g1[i]) {
                        C.g0[i] = T.g0[i];
                        C.g1[i] = T.g1[i];
                    } else {
                        C.g0[i] = std::min(C.g0[i], T.g0[i]);
                        C.g1[i] = std::max(C.g1[i], T.g1[i]);
                    }
                }
                const bool active = T.g0[i] < T.g1[i];
                T.g_off[i] = (long long)off;
                if (active) off += (size_t)cn * T.H[i] * T.P[i];
                T.r_off[i] = (long long)off;
                if (active && !(P->fused && i == 1)) off += (size_t)cn * T.H[i] * T.P[i];     // fused gather: R_1 lives in LDS only
            }
        }
        P->tile_rows[t] = lv[t].gw[0];
    }
    for (int i = 0; i < SR_MAX_LEVELS; ++i) {
        for (auto &T : P->tiles) {
            if (i >= T.nl) continue;
            P->max_w[i] = std::max(P->max_w[i], T.W[i]);
            P->max_grows[i] = std::max(P->max_grows[i], T.g1[i] - T.g0[i]);
            P->max_rrows[i] = std::max(P->max_rrows[i], T.r1[i] - T.r0[i]);
        }
        for (auto &C : P->classes) {
            if (i >= C.nl) continue;
            P->cmax_w[i] = std::max(P->cmax_w[i], C.W[i]);
            P->cmax_rows[i] = std::max(P->cmax_rows[i], C.g1[i] - C.g0[i]);
        }
    }
    P->arena_floats = off;

    auto fail = [&](hipError_t e, const char *what) {
        int code = (e == hipErrorOutOfMemory) ? SR_ERR_OOM : SR_ERR_HIP;
        sr_set_error(code, "sr_blend_plan_create: %s: %s", what, hipGetErrorString(e));
        {
            std::lock_guard<std::mutex> lk(g_reg_mu);
            g_live_plan.insert(P);
        }
        sr_blend_plan_destroy(P);
        return code;
    };
    hipError_t e;
    if ((e = hipMalloc((void **)&P->d_arena, std::max<size_t>(off, 4) * sizeof(float))) != hipSuccess) return fail(e, "arena");
    if ((e = hipMalloc((void **)&P->d_tiles, sizeof(TileDev) * n)) != hipSuccess) return fail(e, "tile table");
    if ((e = hipMalloc((void **)&P->d_classes, sizeof(TileDev) * P->classes.size())) != hipSuccess) return fail(e, "class table");
    if ((e = hipMalloc((void **)&P->d_srcs, sizeof(TileSrc) * n)) != hipSuccess) return fail(e, "src table");
    if ((e = hipMalloc((void **)&P->d_fdesc, sizeof(FinalDesc) * n)) != hipSuccess) return fail(e, "final table");
    P->fdesc.resize(n);
    for (int t = 0; t < n; ++t) {
        const TileDev &T = P->tiles[t];
        FinalDesc &D = P->fdesc[t];
        memset(&D, 0, sizeof(D));
        D.x = T.x; D.y = T.y; D.w = T.w; D.h = T.h;
        D.fw = T.fw; D.lut_off = T.lut_off; D.nl = T.nl;
        D.H1 = T.nl > 1 ? T.H[1] : 1; D.W1 = T.nl > 1 ? T.W[1] : 1; D.P1 = T.nl > 1 ? T.P[1] : 16;
        D.g1 = T.nl > 1 ? T.g_off[1] : 0; D.r1 = T.nl > 1 ? T.r_off[1] : 0;
        D.H2 = T.nl > 2 ? T.H[2] : 1; D.W2 = T.nl > 2 ? T.W[2] : 1; D.P2 = T.nl > 2 ? T.P[2] : 16;
        D.g2 = T.nl > 2 ? T.g_off[2] : 0; D.r2 = T.nl > 2 ? T.r_off[2] : 0;
        D.w1 = T.nl > 1 ? T.w_off[1] : 0;
    }

    if ((e = hipMalloc((void **)&P->d_luts, sizeof(float) * P->luts.size())) != hipSuccess) return fail(e, "luts");
    {
        // Edge work list: the cells (4 x 2 pixel rectangles of one thread) in which a visit can be a border visit lie
        // within 8 px of a tile edge line (or on the ragged right / bottom canvas edge).  Horizontal lines are
        // covered by 256 x 8 blocks of the fast pass's grid (shape 0), vertical lines by 16 x 128 blocks (shape 1:
        // 4 x 64 cells), so a wave of the edge pass is mostly border cells either way.  A cell reached through both
        // shapes is computed twice with identical results.
        const int rows = row_end - row_begin;
        const int nbx = (canvas_w + 255) / 256, nby = (rows + 7) / 8;
        const int nbx2 = (canvas_w + 15) / 16, nby2 = (rows + 127) / 128;
        std::vector<unsigned char> mark((size_t)std::max(nbx, 1) * std::max(nby, 1), 0);
        std::vector<unsigned char> mark2((size_t)std::max(nbx2, 1) * std::max(nby2, 1), 0);
        auto mark_rect = [&](std::vector<unsigned char> &m, int gx, int gy, int pitch, long long x0, long long y0,
                             long long x1, long long y1) {   // canvas px, half-open
            x0 = std::max<long long>(x0, 0); x1 = std::min<long long>(x1, canvas_w);
            y0 = std::max<long long>(y0, row_begin); y1 = std::min<long long>(y1, row_end);
            if (x0 >= x1 || y0 >= y1) return;
            for (long long by = (y0 - row_begin) / gy; by <= (y1 - 1 - row_begin) / gy; ++by)
                for (long long bx = x0 / gx; bx <= (x1 - 1) / gx; ++bx) m[(size_t)by * pitch + bx] = 1;
        };
        const int M = 8;
        for (int t = 0; t < n && rows > 0; ++t) {
            const TileDev &T = P->tiles[t];
            const long long xa = T.x, xb = (long long)T.x + T.w, ya = T.y, yb = (long long)T.y + T.h;
            mark_rect(mark2, 16, 128, nbx2, xa - M, ya - M, xa + M, yb + M);
            mark_rect(mark2, 16, 128, nbx2, xb - M, ya - M, xb + M, yb + M);
            mark_rect(mark, 256, 8, nbx, xa - M, ya - M, xb + M, ya + M);
            mark_rect(mark, 256, 8, nbx, xa - M, yb - M, xb + M, yb + M);
        }
        if (rows > 0) {
            if (canvas_w % 4) mark_rect(mark2, 16, 128, nbx2, canvas_w - 4, row_begin, canvas_w, row_end);
            if (rows % 2) mark_rect(mark, 256, 8, nbx, 0, row_end - 2, canvas_w, row_end);
        }
        std::vector<int4> eb;                       // x0, y0 (canvas), shape, first candidate
        for (int by = 0; by < nby; ++by)
            for (int bx = 0; bx < nbx; ++bx)
                if (mark[(size_t)by * nbx + bx]) eb.push_back(make_int4(bx * 256, row_begin + by * 8, 0, 0));
        for (int by = 0; by < nby2; ++by)
            for (int bx = 0; bx < nbx2; ++bx)
                if (mark2[(size_t)by * nbx2 + bx]) eb.push_back(make_int4(bx * 16, row_begin + by * 128, 1, 0));
        std::vector<int> ecand;
        for (auto &e4 : eb) {
            const long long bx0 = e4.x, by0 = e4.y;
            const long long bx1 = std::min<long long>(bx0 + (e4.z ? 16 : 256), canvas_w);
            const long long by1 = std::min<long long>(by0 + (e4.z ? 128 : 8), row_end);
            e4.w = (int)ecand.size();
            for (int t = 0; t < n; ++t) {
                const TileDev &T = P->tiles[t];
                if (T.x < bx1 && (long long)T.x + T.w > bx0 && T.y < by1 && (long long)T.y + T.h > by0) ecand.push_back(t);
            }
        }
        eb.push_back(make_int4(0, 0, 0, (int)ecand.size()));       // sentinel: end of the last list
        P->n_edge_blocks = (int)eb.size() - 1;
        if (P->n_edge_blocks > 0) {
            if ((e = hipMalloc((void **)&P->d_edge_blocks, sizeof(int4) * eb.size())) != hipSuccess) return fail(e, "edge blocks");
            if ((e = hipMemcpy(P->d_edge_blocks, eb.data(), sizeof(int4) * eb.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "upload");
            if ((e = hipMalloc((void **)&P->d_edge_cand, sizeof(int) * std::max<size_t>(ecand.size(), 1))) != hipSuccess) return fail(e, "edge candidates");
            if (!ecand.empty() && (e = hipMemcpy(P->d_edge_cand, ecand.data(), sizeof(int) * ecand.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "upload");
        }
        // candidate tiles per regular block, CSR, tiles in list order (the accumulation order of the reference)
        const int nbx_r = std::max((canvas_w + FIN_BW - 1) / FIN_BW, 1), nby_r = std::max((rows + FIN_BH - 1) / FIN_BH, 1);
        const size_t nblk = (size_t)nbx_r * nby_r;
        std::vector<int> coff(nblk + 1, 0);
        auto block_span = [&](const TileDev &T, int &bx_a, int &bx_b, int &by_a, int &by_b) {
            const long long x0 = std::max<long long>(T.x, 0), x1 = std::min<long long>((long long)T.x + T.w, canvas_w);
            const long long y0 = std::max<long long>(T.y, row_begin), y1 = std::min<long long>((long long)T.y + T.h, row_end);
            if (x0 >= x1 || y0 >= y1) return false;
            bx_a = (int)(x0 / FIN_BW); bx_b = (int)((x1 - 1) / FIN_BW);
            by_a = (int)((y0 - row_begin) / FIN_BH); by_b = (int)((y1 - 1 - row_begin) / FIN_BH);
            return true;
        };
        for (int t = 0; t < n && rows > 0; ++t) {
            int bxa, bxb, bya, byb;
            if (!block_span(P->tiles[t], bxa, bxb, bya, byb)) continue;
            for (int by = bya; by <= byb; ++by)
                for (int bx = bxa; bx <= bxb; ++bx) ++coff[(size_t)by * nbx_r + bx + 1];
        }
        for (size_t i = 0; i < nblk; ++i) coff[i + 1] += coff[i];
        std::vector<int> cidx((size_t)std::max(coff[nblk], 1), 0), fill(coff.begin(), coff.end() - 1);
        for (int t = 0; t < n && rows > 0; ++t) {
            int bxa, bxb, bya, byb;
            if (!block_span(P->tiles[t], bxa, bxb, bya, byb)) continue;
            for (int by = bya; by <= byb; ++by)
                for (int bx = bxa; bx <= bxb; ++bx) cidx[(size_t)fill[(size_t)by * nbx_r + bx]++] = t;
        }
        if ((e = hipMalloc((void **)&P->d_cand_off, sizeof(int) * coff.size())) != hipSuccess) return fail(e, "candidate table");
        if ((e = hipMalloc((void **)&P->d_cand_idx, sizeof(int) * cidx.size())) != hipSuccess) return fail(e, "candidate table");
        if ((e = hipMemcpy(P->d_cand_off, coff.data(), sizeof(int) * coff.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "upload");
        if ((e = hipMemcpy(P->d_cand_idx, cidx.data(), sizeof(int) * cidx.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "upload");
    }
    if (P->fused) {
        // ---- tables of the fused gather ---------------------------------------------------------------------------------
        const int rows = row_end - row_begin;
        // host mirror of visit_is_interior<true> (device): a 4 x 2 cell at tile-local (lx0, ly0), nx x ny of it on the strip
        auto cell_interior = [](const TileDev &T, long long lx0, long long ly0, int nx, int ny) {
            if (nx != 4 || ny != 2 || lx0 < 0 || ly0 < 0 || lx0 + 3 >= T.w || ly0 + 1 >= T.h) return false;
            if (T.nl > 1) {
                const long long r0 = (ly0 - 1) >> 1, c0 = (lx0 - 1) >> 1;
                return c0 >= 0 && c0 + 3 <= T.W[1] - 1 && r0 >= 0 && r0 + 2 <= T.H[1] - 1;
            }
            return true;
        };
        const int enbx = (canvas_w + 255) / 256, enby = (rows + FU_E0H - 1) / FU_E0H;     // shape 0: 256 x FU_E0H
        const int enbx2 = (canvas_w + FU_E1W - 1) / FU_E1W, enby2 = (rows + FU_E1H - 1) / FU_E1H;     // shape 1: FU_E1W x FU_E1H
        std::vector<unsigned char> m0((size_t)std::max(enbx, 1) * std::max(enby, 1), 0), m1((size_t)std::max(enbx2, 1) * std::max(enby2, 1), 0);
        // Every cell with a border visit lies within a few pixels of the edge line of the tile it visits (or on the
        // ragged right / bottom end of the strip): walk the cells of a 24-pixel frame around each tile's outline, test
        // them exactly, mark the block that holds them -- blocks along horizontal lines in shape 0, along vertical
        // lines in shape 1 (a corner cell may be marked in both: computed twice, identical bytes).
        auto scan = [&](const TileDev &T, long long xa, long long ya, long long xb, long long yb, int shape) {   // canvas px, half-open
            xa = std::max<long long>(xa, 0); xb = std::min<long long>(xb, canvas_w);
            ya = std::max<long long>(ya, row_begin); yb = std::min<long long>(yb, row_end);
            if (xa >= xb || ya >= yb) return;
            for (long long cy = (ya - row_begin) / 2; cy <= (yb - 1 - row_begin) / 2; ++cy)
                for (long long cx = xa / 4; cx <= (xb - 1) / 4; ++cx) {
                    const long long x0 = cx * 4, y0 = row_begin + cy * 2;
                    const int nx = (int)std::min<long long>(4, canvas_w - x0), ny = (int)std::min<long long>(2, row_end - y0);
                    const long long lx0 = x0 - T.x, ly0 = y0 - T.y;
                    if (lx0 + nx <= 0 || ly0 + ny <= 0 || lx0 >= T.w || ly0 >= T.h) continue;
                    if (cell_interior(T, lx0, ly0, nx, ny)) continue;
                    if (shape == 0) m0[(size_t)((y0 - row_begin) / FU_E0H) * enbx + x0 / 256] = 1;
                    else m1[(size_t)((y0 - row_begin) / FU_E1H) * enbx2 + x0 / FU_E1W] = 1;
                }
        };
        const long long F = 24;
        for (int t = 0; t < n && rows > 0; ++t) {
            const TileDev &T = P->tiles[t];
            const long long xa = T.x, xb = (long long)T.x + T.w, ya = T.y, yb = (long long)T.y + T.h;
            scan(T, xa - F, ya - F, xb + F, ya + F, 0);
            scan(T, xa - F, yb - F, xb + F, yb + F, 0);
            scan(T, xa - F, ya - F, xa + F, yb + F, 1);
            scan(T, xb - F, ya - F, xb + F, yb + F, 1);
            // the ragged ends of the strip (cells narrower / shorter than 4): border visits of every tile they touch
            if (canvas_w % 4) scan(T, canvas_w - (canvas_w % 4), row_begin, canvas_w, row_end, 1);
            if (rows % 2) scan(T, 0, row_end - 1, canvas_w, row_end, 0);
        }
        std::vector<int4> eb;
        for (int by = 0; by < enby; ++by)
            for (int bx = 0; bx < enbx; ++bx)
                if (m0[(size_t)by * enbx + bx]) eb.push_back(make_int4(bx * 256, row_begin + by * FU_E0H, 0, 0));
        for (int by = 0; by < enby2; ++by)
            for (int bx = 0; bx < enbx2; ++bx)
                if (m1[(size_t)by * enbx2 + bx]) eb.push_back(make_int4(bx * FU_E1W, row_begin + by * FU_E1H, 1, 0));
        std::vector<int> ecand;
        for (auto &e4 : eb) {
            const long long bx0 = e4.x, by0 = e4.y;
            const long long bx1 = std::min<long long>(bx0 + (e4.z ? FU_E1W : 256), canvas_w);
            const long long by1 = std::min<long long>(by0 + (e4.z ? FU_E1H : FU_E0H), row_end);
            e4.w = (int)ecand.size();
            for (int t = 0; t < n; ++t) {
                const TileDev &T = P->tiles[t];
                if (T.x < bx1 && (long long)T.x + T.w > bx0 && T.y < by1 && (long long)T.y + T.h > by0) ecand.push_back(t);
            }
        }
        eb.push_back(make_int4(0, 0, 0, (int)ecand.size()));
        P->n_fedge_blocks = (int)eb.size() - 1;
        if ((e = hipMalloc((void **)&P->d_fedge_blocks, sizeof(int4) * eb.size())) != hipSuccess) return fail(e, "fused edge blocks");
        if ((e = hipMemcpy(P->d_fedge_blocks, eb.data(), sizeof(int4) * eb.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "upload");
        if ((e = hipMalloc((void **)&P->d_fedge_cand, sizeof(int) * std::max<size_t>(ecand.size(), 1))) != hipSuccess) return fail(e, "fused edge candidates");
        if (!ecand.empty() && (e = hipMemcpy(P->d_fedge_cand, ecand.data(), sizeof(int) * ecand.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "upload");
        // candidate tiles per regular block (FU_BW x FU_BH), CSR, list order
        const int nbx_r = std::max((canvas_w + FU_BW - 1) / FU_BW, 1), nby_r = std::max((rows + FU_BH - 1) / FU_BH, 1);
        const size_t nblk = (size_t)nbx_r * nby_r;
        std::vector<int> coff(nblk + 1, 0);
        auto block_span = [&](const TileDev &T, int &bx_a, int &bx_b, int &by_a, int &by_b) {
            const long long x0 = std::max<long long>(T.x, 0), x1 = std::min<long long>((long long)T.x + T.w, canvas_w);
            const long long y0 = std::max<long long>(T.y, row_begin), y1 = std::min<long long>((long long)T.y + T.h, row_end);
            if (x0 >= x1 || y0 >= y1) return false;
            bx_a = (int)(x0 / FU_BW); bx_b = (int)((x1 - 1) / FU_BW);
            by_a = (int)((y0 - row_begin) / FU_BH); by_b = (int)((y1 - 1 - row_begin) / FU_BH);
            return true;
        };
        for (int t = 0; t < n && rows > 0; ++t) {
            int bxa, bxb, bya, byb;
            if (!block_span(P->tiles[t], bxa, bxb, bya, byb)) continue;
            for (int by = bya; by <= byb; ++by)
                for (int bx = bxa; bx <= bxb; ++bx) ++coff[(size_t)by * nbx_r + bx + 1];
        }
        for (size_t i = 0; i < nblk; ++i) coff[i + 1] += coff[i];
        std::vector<int> cidx((size_t)std::max(coff[nblk], 1), 0), fill(coff.begin(), coff.end() - 1);
        for (int t = 0; t < n && rows > 0; ++t) {
            int bxa, bxb, bya, byb;
            if (!block_span(P->tiles[t], bxa, bxb, bya, byb)) continue;
            for (int by = bya; by <= byb; ++by)
                for (int bx = bxa; bx <= bxb; ++bx) cidx[(size_t)fill[(size_t)by * nbx_r + bx]++] = t;
        }
        if ((e = hipMalloc((void **)&P->d_fcand_off, sizeof(int) * coff.size())) != hipSuccess) return fail(e, "fused candidate table");
        if ((e = hipMalloc((void **)&P->d_fcand_idx, sizeof(int) * cidx.size())) != hipSuccess) return fail(e, "fused candidate table");
        if ((e = hipMemcpy(P->d_fcand_off, coff.data(), sizeof(int) * coff.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "upload");
        if ((e = hipMemcpy(P->d_fcand_idx, cidx.data(), sizeof(int) * cidx.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "upload");
        // ---- marched zones and what is left for the regular blocks -----------------------------------------------------
        std::vector<MarchItem> mitems[MARCH_NT + 1];
        std::vector<int> reg_list;
        std::vector<unsigned> cover;
        plan_march(P, nbx_r, nby_r, mitems, reg_list, cover);
        {
            bool any_march = false;
            for (int k = 1; k <= MARCH_NT; ++k) any_march = any_march || !mitems[k].empty();
            if (any_march && (P->cn == 3 || P->cn == 1)) {
                std::vector<RectItem> rects;
                std::vector<int> rcand;
                plan_rects(P, nbx_r, cover, rects, rcand);
                P->n_rects = (long long)rects.size();
                if ((e = hipMalloc((void **)&P->d_rects, sizeof(RectItem) * std::max<size_t>(rects.size(), 1))) != hipSuccess) return fail(e, "rectangle items");
                if (!rects.empty() && (e = hipMemcpy(P->d_rects, rects.data(), sizeof(RectItem) * rects.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "upload");
                if ((e = hipMalloc((void **)&P->d_rect_cand, sizeof(int) * std::max<size_t>(rcand.size(), 1))) != hipSuccess) return fail(e, "rectangle candidates");
                if (!rcand.empty() && (e = hipMemcpy(P->d_rect_cand, rcand.data(), sizeof(int) * rcand.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "upload");
            }
        }
        {
            std::vector<int> all(nblk * 9, 0);
            for (size_t b = 0; b < nblk; ++b) all[b * 9] = (int)b;
            P->n_freg_all = (long long)nblk;
            if ((e = hipMalloc((void **)&P->d_freg_all, sizeof(int) * all.size())) != hipSuccess) return fail(e, "regular block list");
            if ((e = hipMemcpy(P->d_freg_all, all.data(), sizeof(int) * all.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "upload");
        }
        P->n_freg = (long long)(reg_list.size() / 9);
        if ((e = hipMalloc((void **)&P->d_freg_list, sizeof(int) * std::max<size_t>(reg_list.size(), 1))) != hipSuccess) return fail(e, "regular block list");
        if (!reg_list.empty() && (e = hipMemcpy(P->d_freg_list, reg_list.data(), sizeof(int) * reg_list.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "upload");
        for (int k = 1; k <= MARCH_NT; ++k) {
            P->n_march_items[k] = (int)mitems[k].size();
            P->n_march_total += (long long)mitems[k].size();
            if (mitems[k].empty()) continue;
            if ((e = hipMalloc((void **)&P->d_march_items[k], sizeof(MarchItem) * mitems[k].size())) != hipSuccess) return fail(e, "march items");
            if ((e = hipMemcpy(P->d_march_items[k], mitems[k].data(), sizeof(MarchItem) * mitems[k].size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "upload");
        }
    }
    if ((e = hipMemcpyAsync(P->d_tiles, P->tiles.data(), sizeof(TileDev) * n, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) return fail(e, "upload");
    if ((e = hipMemcpyAsync(P->d_classes, P->classes.data(), sizeof(TileDev) * P->classes.size(), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) return fail(e, "upload");
    if ((e = hipMemcpyAsync(P->d_luts, P->luts.data(), sizeof(float) * P->luts.size(), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) return fail(e, "upload");
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(e, "sync");
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        g_live_plan.insert(P);
    }
    *out = P;
    return SR_OK;
}

int sr_blend_plan_tile_rows(const sr_blend_plan *plan, int t, int *r0, int *r1)
{
    if (!plan_is_live(plan) || t < 0 || t >= plan->n || !r0 || !r1) return sr_set_error(SR_ERR_INVALID_ARG, "sr_blend_plan_tile_rows: bad args");
    *r0 = plan->tile_rows[t].a;
    *r1 = plan->tile_rows[t].b;
    return SR_OK;
}

int sr_blend_plan_workspace_bytes(const sr_blend_plan *plan, size_t *bytes)
{
    if (!plan_is_live(plan) || !bytes) return sr_set_error(SR_ERR_INVALID_ARG, "sr_blend_plan_workspace_bytes: bad args");
    *bytes = plan->arena_floats * sizeof(float);
    return SR_OK;
}

static int blend_check_tiles(sr_blend_plan *P, int dtype, void *const *h_d_tiles, const int64_t *h_strides)
{
    if (!h_d_tiles || !h_strides) return sr_set_error(SR_ERR_INVALID_ARG, "blend: null argument");
    if (dtype != SR_U8 && dtype != SR_F32) return sr_set_error(SR_ERR_INVALID_ARG, "blend: dtype must be SR_U8 or SR_F32");
    const int es = dtype == SR_U8 ? 1 : 4;
    for (int t = 0; t < P->n; ++t) {
        if (!h_d_tiles[t] && !P->tile_rows[t].empty()) return sr_set_error(SR_ERR_INVALID_ARG, "blend: tile %d pointer is null", t);
        if (h_strides[t] < (int64_t)P->tiles[t].w * P->cn * es) return sr_set_error(SR_ERR_SHAPE, "blend: tile %d stride too small", t);
    }
    return SR_OK;
}

// Stage A of the Laplacian blend: weight pyramids (when `first`) and the down / up chains of the listed tiles.
// The listed tiles' descriptors are compacted into a scratch table, so kernels are launched over exactly them.
static int blend_pyramids(sr_blend_plan *P, int dtype, void *const *h_d_tiles, const int64_t *h_strides,
                          const int *idx, int n_idx, bool first)
{
    sr_ctx *ctx = P->ctx;
    const int rows = P->row_end - P->row_begin;
    if (rows <= 0 || P->max_nl <= 1) return SR_OK;
    dim3 block(64, 4);
    // TIMING PROBE ONLY (wrong pixels): leaves out the small launches of the chain -- level-2 border columns, levels 2 -> 5 down,
    // the collapse 5 -> 2 -- to see what their stretched time beside the assessment costs the image stream
    static const bool probe_skip_small = std::getenv("SR_PROBE_SKIP_SMALL") != nullptr;
    if (first && !P->weights_ready) {
        // weight pyramids: level 0 analytic (LUT) -> 1, then planar chain.  They depend only on the tile shapes and
        // the row windows, both fixed per plan: built by the first blend, kept in the arena for every later one.
        P->weights_ready = true;
        for (int i = 0; i + 1 < P->max_nl; ++i) {
            if (P->cmax_rows[i + 1] <= 0) continue;
            ProfScope ps(ctx, "weight_down");
            dim3 grid((P->cmax_w[i + 1] + 63) / 64, (P->cmax_rows[i + 1] + 3) / 4, (unsigned)P->classes.size());
            if (i == 0) hipLaunchKernelGGL(k_down<SRC_LUT>, grid, block, 0, ctx->stream, P->d_classes, (const TileSrc *)nullptr, i, 1, P->d_arena, P->d_luts);
            else hipLaunchKernelGGL(k_down<SRC_PLANAR>, grid, block, 0, ctx->stream, P->d_classes, (const TileSrc *)nullptr, i, 1, P->d_arena, P->d_luts);
        }
    }
    if (n_idx <= 0) return check_launch("weight chain");
    // compacted tables for this subset (the full tables when the subset is everything, in order)
    const TileDev *d_tiles = P->d_tiles;
    const TileSrc *d_srcs = P->d_srcs;
    std::vector<TileDev> sub_t;
    std::vector<TileSrc> sub_s(n_idx);
    bool all = (n_idx == P->n);
    for (int k = 0; k < n_idx && all; ++k) all = idx[k] == k;
    for (int k = 0; k < n_idx; ++k) {
        const int t = idx[k];
        if (t < 0 || t >= P->n) return sr_set_error(SR_ERR_INVALID_ARG, "blend: tile index %d out of range", t);
        sub_s[k].p = h_d_tiles[t];
        sub_s[k].stride = h_strides[t];
    }
    int max_w[SR_MAX_LEVELS] = {0}, max_g[SR_MAX_LEVELS] = {0}, max_r[SR_MAX_LEVELS] = {0}, max_nl = 1;
    for (int k = 0; k < n_idx; ++k) {
        const TileDev &T = P->tiles[idx[k]];
        max_nl = std::max(max_nl, T.nl);
        for (int i = 0; i < T.nl; ++i) {
            max_w[i] = std::max(max_w[i], T.W[i]);
            max_g[i] = std::max(max_g[i], T.g1[i] - T.g0[i]);
            max_r[i] = std::max(max_r[i], T.r1[i] - T.r0[i]);
        }
    }
    if (all) {
        HIPCHK(upload_if_changed(ctx, P->d_srcs, sub_s.data(), sizeof(TileSrc) * n_idx, P->sh_srcs));
    } else {
        sub_t.resize(n_idx);
        for (int k = 0; k < n_idx; ++k) sub_t[k] = P->tiles[idx[k]];
        auto al = [](size_t v) { return (v + 255) / 256 * 256; };
        const size_t o_src = al(sizeof(TileDev) * n_idx), total = o_src + al(sizeof(TileSrc) * n_idx);
        std::vector<char> packed(total, 0);
        memcpy(packed.data(), sub_t.data(), sizeof(TileDev) * n_idx);
        memcpy(packed.data() + o_src, sub_s.data(), sizeof(TileSrc) * n_idx);
        // a subset seen recently (the held / arriving halves of a staged blend alternate) keeps its device table
        CachedTable *slot = nullptr;
        for (auto &t : P->subset_tabs)
            if (t.shadow.size() == total && memcmp(t.shadow.data(), packed.data(), total) == 0) slot = &t;
        if (!slot) {
            slot = &P->subset_tabs[P->subset_next];
            P->subset_next = (P->subset_next + 1) % 4;
        }
        HIPCHK(upload_cached(ctx, *slot, packed.data(), total));
        d_tiles = (const TileDev *)slot->d;
        d_srcs = (const TileSrc *)((const char *)slot->d + o_src);
    }
    // tiles whose levels 1 and 2 come out of one march (u8 RGB, 32-bit offsets inside a tile and the arena; a strip's row
    // windows included)
    int n_take = 0, max_take_h1 = 0, max_take_cols = 0;
    if (P->down2 && P->cn == 3 && dtype == SR_U8 && P->arena_floats * sizeof(float) < 0xFFFF0000ull) {
        bool ok = true;
        for (int k = 0; k < n_idx && ok; ++k)
            ok = sub_s[k].stride > 0 && (unsigned long long)sub_s[k].stride * (unsigned long long)(P->tiles[idx[k]].H[0] + 2) < 0x7FFF0000ull;
        for (int k = 0; k < n_idx && ok; ++k) {
            const TileDev &T = P->tiles[idx[k]];
            if (!down2_takes(T)) continue;
            ++n_take;
            max_take_h1 = std::max(max_take_h1, T.g1[1] - T.g0[1]);
            max_take_cols = std::max(max_take_cols, (T.g1[2] - T.g0[2]) * down2_cols_count(T));
        }
    }
    // Gaussian chain
    for (int i = 0; i + 1 < max_nl; ++i) {
        if (max_g[i + 1] <= 0) continue;
        ProfScope ps(ctx, i == 0 ? "down_l0" : "down_l1p");
        const bool blk = (P->cn == 3 || P->cn == 1) && !(i == 0 && dtype != SR_U8);
        int skip2 = 0;
        if (blk && i <= 1 && n_take > 0) {
            // levels 1 and 2 of the tiles with full row windows in one march (sr_down2.inc); the others keep the two launches
            if (i == 0) {
                // Segment length (level-2 rows per work item).  Every item pays three extra level-1 rows, yet short segments win:
                // measured at 200 MP (tools/down2_seg_sweep.sh) 6 .. 12 rows 0.51-0.54 ms, 16: 0.58, 32: 0.61, 64: 0.68 -- the
                // kernel is bound by its stores' way through the memory system, which many short items keep busier.  The longest
                // length up to 12 that still gives every wave slot four items, never below 6.
                int items = 0, seg2 = 12;
                {
                    const long long slots = (long long)ctx->num_cu * 16;    // four waves per SIMD
                    const char *envs = std::getenv("SR_DOWN2_SEG");     // timing runs: a fixed segment length
                    const int hi = envs ? std::max(2, atoi(envs)) : 12, lo = envs ? hi : 6;
                    for (int cand = hi; cand >= lo; --cand) {
                        long long total = 0;
                        items = 0;
                        seg2 = cand;
                        for (int k = 0; k < n_idx; ++k) {
                            const TileDev &T = P->tiles[idx[k]];
                            if (!down2_takes(T)) continue;
                            const int it = down2_nstrip(down_ncg(T.W[0], T.W[1])) * ((T.g1[2] - T.g0[2] + cand - 1) / cand);
                            items = std::max(items, it);
                            total += it;
                        }
                        if (total >= 4 * slots) break;
                    }
                }
                const bool probe_nb = std::getenv("SR_DOWN2_PROBE_NOBORDER") != nullptr;     // timing probe only (wrong border columns)
                dim3 grid2(items + (probe_nb ? 0 : (max_take_h1 + 3) / 4), 1, n_idx);
                hipLaunchKernelGGL((k_down2_march<3>), grid2, dim3(64), 0, ctx->stream, d_tiles, d_srcs, seg2, items, P->d_arena,
                                   (unsigned)(P->arena_floats * sizeof(float)), P->d_arena, P->d_luts);
            } else if (!probe_skip_small) {
                dim3 grid2((max_take_cols + 255) / 256, 1, n_idx);
                hipLaunchKernelGGL(k_down2_cols, grid2, dim3(256), 0, ctx->stream, d_tiles, P->d_arena);
            }
            if (n_take == n_idx) continue;
            skip2 = 1;
        }
        if (blk && probe_skip_small && i >= 2) continue;
        if (blk) {
            int max_cells = 0, seg_rows = 2;
            for (int cand = 32; cand >= 2; cand /= 2) {       // longest segments that still give ~8 blocks per CU;
                                                              // small levels end at 2 rows: short dependent chains
                long long total = 0;
                max_cells = 0;
                seg_rows = cand;
                for (int k = 0; k < n_idx; ++k) {
                    const TileDev &T = P->tiles[idx[k]];
                    if (i + 1 >= T.nl || T.g1[i + 1] <= T.g0[i + 1]) continue;
                    const int nseg = (T.g1[i + 1] - T.g0[i + 1] + cand - 1) / cand;
                    const int cells = down_ncg(T.W[i], T.W[i + 1]) * nseg;
                    max_cells = std::max(max_cells, cells);
                    total += cells;
                }
                if (total >= 512 * 1024) break;
            }
            {
                const int march_blocks = (max_cells + 255) / 256, cols_blocks = (max_g[i + 1] + 15) / 16;
                dim3 grid(march_blocks + cols_blocks, 1, n_idx);
                if (i == 0) {
                    if (P->cn == 3) hipLaunchKernelGGL((k_down_march<SRC_U8, 3>), grid, block, 0, ctx->stream, d_tiles, d_srcs, i, seg_rows, march_blocks, P->d_arena, P->d_luts, skip2);
                    else hipLaunchKernelGGL((k_down_march<SRC_U8, 1>), grid, block, 0, ctx->stream, d_tiles, d_srcs, i, seg_rows, march_blocks, P->d_arena, P->d_luts, skip2);
                } else {
                    if (P->cn == 3) hipLaunchKernelGGL((k_down_march<SRC_PLANAR, 3>), grid, block, 0, ctx->stream, d_tiles, d_srcs, i, seg_rows, march_blocks, P->d_arena, P->d_luts, skip2);
                    else hipLaunchKernelGGL((k_down_march<SRC_PLANAR, 1>), grid, block, 0, ctx->stream, d_tiles, d_srcs, i, seg_rows, march_blocks, P->d_arena, P->d_luts, skip2);
                }
            }
            continue;
        }
        dim3 grid((max_w[i + 1] + 63) / 64, (max_g[i + 1] + 3) / 4, n_idx);
        if (i == 0) {
            if (dtype == SR_U8) hipLaunchKernelGGL(k_down<SRC_U8>, grid, block, 0, ctx->stream, d_tiles, d_srcs, i, P->cn, P->d_arena, P->d_luts);
            else hipLaunchKernelGGL(k_down<SRC_F32>, grid, block, 0, ctx->stream, d_tiles, d_srcs, i, P->cn, P->d_arena, P->d_luts);
        } else {
            hipLaunchKernelGGL(k_down<SRC_PLANAR>, grid, block, 0, ctx->stream, d_tiles, d_srcs, i, P->cn, P->d_arena, P->d_luts);
        }
    }
    int rc = check_launch("down chain");
    if (rc) return rc;
    // collapse chain: levels max_nl-1 .. 1 (.. 2 when the gather is fused: it builds R_1 itself, in LDS)
    for (int i = max_nl - 1; i >= (P->fused ? 2 : 1); --i) {
        if (max_r[i] <= 0 || probe_skip_small) continue;
        ProfScope ps(ctx, "up_level");
        if (P->cn == 3 || P->cn == 1) {
            dim3 grid((max_w[i] + 255) / 256, (max_r[i] + 7) / 8, n_idx);
            if (P->cn == 3) hipLaunchKernelGGL(k_up_level_blk<3>, grid, block, 0, ctx->stream, d_tiles, i, P->d_arena);
            else hipLaunchKernelGGL(k_up_level_blk<1>, grid, block, 0, ctx->stream, d_tiles, i, P->d_arena);
        } else {
            dim3 grid((max_w[i] + 63) / 64, (max_r[i] + 3) / 4, n_idx);
            hipLaunchKernelGGL(k_up_level, grid, block, 0, ctx->stream, d_tiles, i, P->cn, P->d_arena);
        }
    }
    return check_launch("up chain");
}

// Stage B: the canvas gather over all tiles (LAP: from G_1 / R_1 of stage A; else plain weighted average).
static int blend_gather(sr_blend_plan *P, bool lap, int dtype, void *const *h_d_tiles, const int64_t *h_strides,
                        uint8_t *d_canvas, int64_t canvas_stride, float *d_canvas_f32)
{
    sr_ctx *ctx = P->ctx;
    if (!d_canvas) return sr_set_error(SR_ERR_INVALID_ARG, "blend: null canvas");
    if (canvas_stride < (int64_t)P->canvas_w * P->cn) return sr_set_error(SR_ERR_SHAPE, "blend: canvas stride too small");
    const int rows = P->row_end - P->row_begin;
    if (rows <= 0) return SR_OK;
    std::vector<TileSrc> srcs(P->n);
    for (int t = 0; t < P->n; ++t) {
        srcs[t].p = h_d_tiles[t];
        srcs[t].stride = h_strides[t];
        P->fdesc[t].src = h_d_tiles[t];
        P->fdesc[t].stride = h_strides[t];
    }
    HIPCHK(upload_if_changed(ctx, P->d_srcs, srcs.data(), sizeof(TileSrc) * P->n, P->sh_srcs));
    HIPCHK(upload_if_changed(ctx, P->d_fdesc, P->fdesc.data(), sizeof(FinalDesc) * P->n, P->sh_fdesc));
    dim3 block(64, 4);
    {
        ProfScope ps(ctx, lap ? "final_gather" : "weighted_gather");
        if (lap && P->fused) {
            const int nbx_r = std::max((P->canvas_w + FU_BW - 1) / FU_BW, 1), nby_r = std::max((rows + FU_BH - 1) / FU_BH, 1);
            const int n_edge = P->n_fedge_blocks;
            // the marched zones first (long work items), then the edge and regular blocks of what is left; float tiles take
            // the regular blocks everywhere
            const bool marched = dtype == SR_U8 && P->n_march_total > 0;
            // The marched zones and the block kernel write disjoint cells of the canvas: with SR_GATHER_STREAMS=1 the small,
            // latency-bound launches (3- / 4-tile zones, the block kernel) run on the context's side streams (forked off its
            // stream here, joined below) beside the large 1- and 2-tile ones; default:
            // one after the other on the context's stream (per-kernel timing).
            static const bool side_ok = std::getenv("SR_GATHER_STREAMS") && std::getenv("SR_GATHER_STREAMS")[0] == '1';
            bool forked = false;
            hipStream_t ms[3] = {ctx->stream, ctx->stream, ctx->stream};
            // (per-kernel timing of the parts wants them one after the other; timing the whole gather, or another family, does not)
            const bool want_parts = ctx->prof && (ctx->prof_only.empty() || ctx->prof_only.rfind("gather_", 0) == 0);
            if (marched && side_ok && !want_parts) {
                if (!ctx->side_fork) {
                    HIPCHK(hipEventCreateWithFlags(&ctx->side_fork, hipEventDisableTiming));
                    for (int i = 0; i < 2; ++i) {
                        HIPCHK(hipStreamCreateWithFlags(&ctx->side[i], hipStreamNonBlocking));
                        HIPCHK(hipEventCreateWithFlags(&ctx->side_join[i], hipEventDisableTiming));
                    }
                }
                HIPCHK(hipEventRecord(ctx->side_fork, ctx->stream));
                for (int i = 0; i < 2; ++i) HIPCHK(hipStreamWaitEvent(ctx->side[i], ctx->side_fork, 0));
                ms[1] = ctx->side[0];
                ms[2] = ctx->side[1];
                forked = true;
            }
            const unsigned arena_bytes = (unsigned)std::min<size_t>(P->arena_floats * sizeof(float), 0xFFFFFFFFu);
            if (dtype == SR_U8 && P->n_march_items[1] > 0) {
                ProfScope ps2(ctx, "gather_march1");
                if (P->cn == 3)
                    hipLaunchKernelGGL((k_final_march1<3>), dim3((unsigned)P->n_march_items[1]), dim3(64), 0, ms[0], P->d_march_items[1],
                                       P->d_fdesc, P->d_arena, arena_bytes, P->d_luts, d_canvas, (long long)canvas_stride, d_canvas_f32, P->canvas_w);
                else
                    hipLaunchKernelGGL((k_final_march1<1>), dim3((unsigned)P->n_march_items[1]), dim3(64), 0, ms[0], P->d_march_items[1],
                                       P->d_fdesc, P->d_arena, arena_bytes, P->d_luts, d_canvas, (long long)canvas_stride, d_canvas_f32, P->canvas_w);
            }
#define LAUNCH_MARCHN(CNV, NTV)                                                                                          \
    hipLaunchKernelGGL((k_final_marchn<CNV, NTV>), dim3((unsigned)P->n_march_items[NTV]), dim3(64 * NTV), 0, ms[NTV == 2 ? 0 : 1],  \
                       P->d_march_items[NTV], P->d_fdesc, P->d_arena, arena_bytes, P->d_luts, d_canvas,                      \
                       (long long)canvas_stride, d_canvas_f32, P->canvas_w)
#define LAUNCH_MARCHP(NTV)                                                                                               \
    hipLaunchKernelGGL((k_final_marchp<NTV>), dim3((unsigned)P->n_march_items[NTV]), dim3(192 * NTV), 0, ms[NTV == 2 ? 0 : 1],  \
                       P->d_march_items[NTV], P->d_fdesc, P->d_arena, arena_bytes, P->d_luts, d_canvas,                      \
                       (long long)canvas_stride, d_canvas_f32, P->canvas_w)
            static_assert(MARCH_NT == 4, "the tile counts launched here");
            for (int nt = 2; nt <= MARCH_NT && dtype == SR_U8; ++nt) {
                if (P->n_march_items[nt] <= 0) continue;
                ProfScope ps2(ctx, nt == 2 ? "gather_march2" : (nt == 3 ? "gather_march3" : "gather_march4"));
                // SR_MARCH_PLANES=1 (A/B runs): a wave per (tile, plane) (k_final_marchp) instead of a wave per tile with all planes.
                // Measured and rejected: two tiles 0.42 -> 0.59 ms, four 0.165 -> 0.216 -- the step waits on its requests, and
                // three waves that each fetch the shared pixels and W_1 issue 30 requests per strip and step where two issued 18.
                static const bool planes = std::getenv("SR_MARCH_PLANES") && std::getenv("SR_MARCH_PLANES")[0] == '1';
                if (P->cn == 3 && planes) { if (nt == 2) LAUNCH_MARCHP(2); else if (nt == 3) LAUNCH_MARCHP(3); else LAUNCH_MARCHP(4); }
                else if (P->cn == 3) { if (nt == 2) LAUNCH_MARCHN(3, 2); else if (nt == 3) LAUNCH_MARCHN(3, 3); else LAUNCH_MARCHN(3, 4); }
                else            { if (nt == 2) LAUNCH_MARCHN(1, 2); else if (nt == 3) LAUNCH_MARCHN(1, 3); else LAUNCH_MARCHN(1, 4); }
            }
#undef LAUNCH_MARCHN
#undef LAUNCH_MARCHP
            const long long n_reg = marched ? P->n_freg : P->n_freg_all;
            const int *reg_list = marched ? P->d_freg_list : P->d_freg_all;
            ProfScope ps3(ctx, "gather_rest");
            // beside a march the remainder runs as rectangles of cells (k_final_rect: every cell finished in one visit, no edge
            // blocks); SR_RECT=0 restores the masked 128 x 16 blocks + edge blocks (A/B runs; identical bytes)
            const bool rect_on = !(std::getenv("SR_RECT") && std::getenv("SR_RECT")[0] == '0');     // read per call: the tests switch it
            const bool use_rects = marched && rect_on && P->d_rects != nullptr;
            if (use_rects && P->n_rects > 0) {
                if (P->cn == 3)
                    hipLaunchKernelGGL((k_final_rect<3>), dim3((unsigned)P->n_rects), dim3(FU_THREADS), 0, ms[2], P->d_fdesc, P->d_rects,
                                       P->d_rect_cand, P->d_arena, P->d_luts, d_canvas, (long long)canvas_stride, d_canvas_f32, P->canvas_w,
                                       P->row_begin, P->row_end);
                else
                    hipLaunchKernelGGL((k_final_rect<1>), dim3((unsigned)P->n_rects), dim3(FU_THREADS), 0, ms[2], P->d_fdesc, P->d_rects,
                                       P->d_rect_cand, P->d_arena, P->d_luts, d_canvas, (long long)canvas_stride, d_canvas_f32, P->canvas_w,
                                       P->row_begin, P->row_end);
            }
            dim3 grid((unsigned)std::max<long long>(n_edge + n_reg, 1)), blk1(FU_THREADS);
#define LAUNCH_FUSED(DT, CNV)                                                                                          \
    hipLaunchKernelGGL((k_final_fused<DT, CNV>), grid, blk1, 0, ms[2], P->d_fdesc, P->d_fcand_off, P->d_fcand_idx,  \
                       P->d_fedge_blocks, P->d_fedge_cand, n_edge, nbx_r, reg_list, P->d_arena, P->d_luts, d_canvas,   \
                       (long long)canvas_stride, d_canvas_f32, P->canvas_w, P->row_begin, P->row_end)
            if (!use_rects && n_edge + n_reg > 0) {
                if (P->cn == 3) { if (dtype == SR_U8) LAUNCH_FUSED(SRC_U8, 3); else LAUNCH_FUSED(SRC_F32, 3); }
                else            { if (dtype == SR_U8) LAUNCH_FUSED(SRC_U8, 1); else LAUNCH_FUSED(SRC_F32, 1); }
            }
#undef LAUNCH_FUSED
            if (forked) {
                for (int i = 0; i < 2; ++i) HIPCHK(hipEventRecord(ctx->side_join[i], ctx->side[i]));
                for (int i = 0; i < 2; ++i) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->side_join[i], 0));
            }
        } else if (P->cn == 3 || P->cn == 1) {
            const int nbx_r = std::max((P->canvas_w + FIN_BW - 1) / FIN_BW, 1), nby_r = std::max((rows + FIN_BH - 1) / FIN_BH, 1);
            dim3 grid((unsigned)(P->n_edge_blocks + (long long)nbx_r * nby_r));     // edge blocks first, then the regular ones
#define LAUNCH_BLK(DT, LAPV, CNV)                                                                               \
    hipLaunchKernelGGL((k_final_fast<DT, LAPV, CNV>), grid, block, 0, ctx->stream, P->d_fdesc, P->d_cand_off,     \
                       P->d_cand_idx, P->d_edge_blocks, P->d_edge_cand, P->n_edge_blocks, nbx_r, P->d_arena, \
                       P->d_luts, d_canvas, (long long)canvas_stride, d_canvas_f32, P->canvas_w, P->row_begin,   \
                       P->row_end)
            if (P->cn == 3) {
                if (lap) { if (dtype == SR_U8) LAUNCH_BLK(SRC_U8, true, 3); else LAUNCH_BLK(SRC_F32, true, 3); }
                else     { if (dtype == SR_U8) LAUNCH_BLK(SRC_U8, false, 3); else LAUNCH_BLK(SRC_F32, false, 3); }
            } else {
                if (lap) { if (dtype == SR_U8) LAUNCH_BLK(SRC_U8, true, 1); else LAUNCH_BLK(SRC_F32, true, 1); }
                else     { if (dtype == SR_U8) LAUNCH_BLK(SRC_U8, false, 1); else LAUNCH_BLK(SRC_F32, false, 1); }
            }
#undef LAUNCH_BLK
        } else {
            dim3 grid((P->canvas_w + 63) / 64, (rows + 3) / 4);
#define LAUNCH_FINAL(DT, LAPV)                                                                                  \
    hipLaunchKernelGGL((k_final<DT, LAPV>), grid, block, 0, ctx->stream, P->d_tiles, P->d_srcs, P->n, P->cn,     \
                       P->d_arena, P->d_luts, d_canvas, (long long)canvas_stride, d_canvas_f32, P->canvas_w,     \
                       P->row_begin, P->row_end)
            if (lap) {
                if (dtype == SR_U8) LAUNCH_FINAL(SRC_U8, true);
                else LAUNCH_FINAL(SRC_F32, true);
            } else {
                if (dtype == SR_U8) LAUNCH_FINAL(SRC_U8, false);
                else LAUNCH_FINAL(SRC_F32, false);
            }
#undef LAUNCH_FINAL
        }
    }
    return check_launch("final gather");
}

static int blend_impl(sr_blend_plan *P, bool lap, int dtype, void *const *h_d_tiles, const int64_t *h_strides,
                      uint8_t *d_canvas, int64_t canvas_stride, float *d_canvas_f32)
{
    int rc = blend_check_tiles(P, dtype, h_d_tiles, h_strides);
    if (rc) return rc;
    if (lap) {
        std::vector<int> idx(P->n);
        for (int t = 0; t < P->n; ++t) idx[t] = t;
        rc = blend_pyramids(P, dtype, h_d_tiles, h_strides, idx.data(), P->n, true);
        if (rc) return rc;
    }
    return blend_gather(P, lap, dtype, h_d_tiles, h_strides, d_canvas, canvas_stride, d_canvas_f32);
}

int sr_blend_pyramids(sr_blend_plan *plan, int dtype, void *const *h_d_tiles, const int64_t *h_strides,
                      const int *h_tile_idx, int n_idx, int first)
{
    if (!plan_is_live(plan)) return sr_set_error(SR_ERR_INVALID_ARG, "sr_blend_pyramids: null or destroyed plan");
    CTX_ENTER(plan->ctx);
    if (n_idx < 0 || (n_idx > 0 && !h_tile_idx)) return sr_set_error(SR_ERR_INVALID_ARG, "sr_blend_pyramids: bad tile list");
    int rc = blend_check_tiles(plan, dtype, h_d_tiles, h_strides);
    if (rc) return rc;
    return blend_pyramids(plan, dtype, h_d_tiles, h_strides, h_tile_idx, n_idx, first != 0);
}

int sr_blend_gather(sr_blend_plan *plan, int dtype, void *const *h_d_tiles, const int64_t *h_strides,
                    uint8_t *d_canvas, int64_t canvas_stride, float *d_canvas_f32)
{
    if (!plan_is_live(plan)) return sr_set_error(SR_ERR_INVALID_ARG, "sr_blend_gather: null or destroyed plan");
    CTX_ENTER(plan->ctx);
    int rc = blend_check_tiles(plan, dtype, h_d_tiles, h_strides);
    if (rc) return rc;
    return blend_gather(plan, true, dtype, h_d_tiles, h_strides, d_canvas, canvas_stride, d_canvas_f32);
}

int sr_laplacian_blend(sr_blend_plan *plan, int dtype, void *const *h_d_tiles, const int64_t *h_strides,
                       uint8_t *d_canvas, int64_t canvas_stride, float *d_canvas_f32)
{
    if (!plan_is_live(plan)) return sr_set_error(SR_ERR_INVALID_ARG, "sr_laplacian_blend: null or destroyed plan");
    CTX_ENTER(plan->ctx);
    return blend_impl(plan, true, dtype, h_d_tiles, h_strides, d_canvas, canvas_stride, d_canvas_f32);
}

int sr_weighted_blend(sr_blend_plan *plan, int dtype, void *const *h_d_tiles, const int64_t *h_strides,
                      uint8_t *d_canvas, int64_t canvas_stride, float *d_canvas_f32)
{
    if (!plan_is_live(plan)) return sr_set_error(SR_ERR_INVALID_ARG, "sr_weighted_blend: null or destroyed plan");
    CTX_ENTER(plan->ctx);
    return blend_impl(plan, false, dtype, h_d_tiles, h_strides, d_canvas, canvas_stride, d_canvas_f32);
}

static int fusion_host_impl(sr_ctx *ctx, bool lap, int dtype, const void *const *h_tiles, const sr_tile_rect *h_rects,
                            int n, int cn, int canvas_h, int canvas_w, int levels, int weight_type, uint8_t *h_canvas,
                            float *h_canvas_f32)
{
    if (!h_tiles || !h_rects || !h_canvas || n < 1) return sr_set_error(SR_ERR_INVALID_ARG, "fusion_host: bad arguments");
    if (dtype != SR_U8 && dtype != SR_F32) return sr_set_error(SR_ERR_INVALID_ARG, "fusion_host: bad dtype");
    sr_blend_plan *plan = nullptr;
    int rc = sr_blend_plan_create(ctx, h_rects, n, cn, canvas_h, canvas_w, levels, weight_type, 0, canvas_h, &plan);
    if (rc) return rc;
    const int es = dtype == SR_U8 ? 1 : 4;
    std::vector<void *> d_tiles(n, nullptr);
    std::vector<int64_t> strides(n);
    void *d_canvas = nullptr, *d_f32 = nullptr;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(ctx->stream);
        for (auto p : d_tiles)
            if (p) (void)hipFree(p);
        if (d_canvas) (void)hipFree(d_canvas);
        if (d_f32) (void)hipFree(d_f32);
        sr_blend_plan_destroy(plan);
    };
    for (int t = 0; t < n && rc == SR_OK; ++t) {
        const size_t bytes = (size_t)h_rects[t].h * h_rects[t].w * cn * es;
        strides[t] = (int64_t)h_rects[t].w * cn * es;
        hipError_t e = hipMalloc(&d_tiles[t], bytes);
        if (e != hipSuccess) rc = sr_set_error(e == hipErrorOutOfMemory ? SR_ERR_OOM : SR_ERR_HIP, "fusion_host: hipMalloc: %s", hipGetErrorString(e));
        else if ((e = hipMemcpyAsync(d_tiles[t], h_tiles[t], bytes, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
            rc = sr_set_error(SR_ERR_HIP, "fusion_host: H2D: %s", hipGetErrorString(e));
    }
    const size_t cbytes = (size_t)canvas_h * canvas_w * cn;
    if (rc == SR_OK) {
        hipError_t e = hipMalloc(&d_canvas, cbytes);
        if (e != hipSuccess) rc = sr_set_error(e == hipErrorOutOfMemory ? SR_ERR_OOM : SR_ERR_HIP, "fusion_host: hipMalloc: %s", hipGetErrorString(e));
    }
    if (rc == SR_OK && h_canvas_f32) {
        hipError_t e = hipMalloc(&d_f32, cbytes * sizeof(float));
        if (e != hipSuccess) rc = sr_set_error(e == hipErrorOutOfMemory ? SR_ERR_OOM : SR_ERR_HIP, "fusion_host: hipMalloc: %s", hipGetErrorString(e));
    }
    if (rc == SR_OK)
        rc = blend_impl(plan, lap, dtype, d_tiles.data(), strides.data(), (uint8_t *)d_canvas, (int64_t)canvas_w * cn, (float *)d_f32);
    if (rc == SR_OK) {
        hipError_t e = hipMemcpyAsync(h_canvas, d_canvas, cbytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && h_canvas_f32) e = hipMemcpyAsync(h_canvas_f32, d_f32, cbytes * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = sr_set_error(SR_ERR_HIP, "fusion_host: D2H: %s", hipGetErrorString(e));
    }
    cleanup();
    return rc;
}

int sr_laplacian_fusion_host(sr_ctx *ctx, int dtype, const void *const *h_tiles, const sr_tile_rect *h_rects, int n,
                             int cn, int canvas_h, int canvas_w, int levels, int weight_type, uint8_t *h_canvas,
                             float *h_canvas_f32)
{
    CTX_ENTER(ctx);
    return fusion_host_impl(ctx, true, dtype, h_tiles, h_rects, n, cn, canvas_h, canvas_w, levels, weight_type, h_canvas,
                            h_canvas_f32);
}

int sr_weighted_fusion_host(sr_ctx *ctx, int dtype, const void *const *h_tiles, const sr_tile_rect *h_rects, int n,
                            int cn, int canvas_h, int canvas_w, int weight_type, uint8_t *h_canvas, float *h_canvas_f32)
{
    CTX_ENTER(ctx);
    return fusion_host_impl(ctx, false, dtype, h_tiles, h_rects, n, cn, canvas_h, canvas_w, 1, weight_type, h_canvas,
                            h_canvas_f32);
}

int sr_feather_merge(sr_ctx *ctx, const sr_merge_tile *h_tiles, int n, void *const *h_d_tiles,
                     const int64_t *h_strides, int blending, uint8_t *d_canvas, int64_t canvas_stride, int canvas_h,
                     int canvas_w)
{
    return sr_feather_merge_dt(ctx, SR_U8, h_tiles, n, h_d_tiles, h_strides, blending, d_canvas, canvas_stride, canvas_h, canvas_w);
}

int sr_feather_merge_dt(sr_ctx *ctx, int dtype, const sr_merge_tile *h_tiles, int n, void *const *h_d_tiles,
                        const int64_t *h_strides, int blending, uint8_t *d_canvas, int64_t canvas_stride, int canvas_h,
                        int canvas_w)
{
    CTX_ENTER(ctx);
    if (!h_tiles || !h_d_tiles || !h_strides || !d_canvas || n < 0 || canvas_h < 1 || canvas_w < 1)
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_feather_merge: bad arguments");
    if (dtype != SR_U8 && dtype != SR_F32) return sr_set_error(SR_ERR_INVALID_ARG, "sr_feather_merge: dtype must be SR_U8 or SR_F32");
    const int es = dtype == SR_U8 ? 1 : 4;
    if (canvas_stride < (int64_t)canvas_w * 3) return sr_set_error(SR_ERR_SHAPE, "sr_feather_merge: canvas stride too small");
    std::vector<MergeDev> md(n);
    std::vector<TileSrc> srcs(n);
    std::vector<LinTab> tabs;
    for (int t = 0; t < n; ++t) {
        const sr_merge_tile &m = h_tiles[t];
        if (m.src_w < 1 || m.src_h < 1 || m.out_w < 1 || m.out_h < 1 || m.x < 0 || m.y < 0 || !h_d_tiles[t])
            return sr_set_error(SR_ERR_INVALID_ARG, "sr_feather_merge: tile %d has a bad descriptor", t);
        if (blending && (m.ov_t > m.out_h || m.ov_b > m.out_h || m.ov_l > m.out_w || m.ov_r > m.out_w))
            return sr_set_error(SR_ERR_SHAPE,
                                "sr_feather_merge: tile %d: overlap ramp longer than the tile (NumPy cannot broadcast "
                                "the reference's ramp either)", t);
        MergeDev &D = md[t];
        D.x = m.x; D.y = m.y; D.src_w = m.src_w; D.src_h = m.src_h; D.out_w = m.out_w; D.out_h = m.out_h;
        D.ov_t = std::max(m.ov_t, 0); D.ov_b = std::max(m.ov_b, 0); D.ov_l = std::max(m.ov_l, 0); D.ov_r = std::max(m.ov_r, 0);
        D.resize = (m.src_w != m.out_w || m.src_h != m.out_h) ? 1 : 0;
        D.xtab = D.ytab = 0;
        if (D.resize) {
            D.xtab = (int)tabs.size();
            linear_table(m.src_w, m.out_w, tabs);
            D.ytab = (int)tabs.size();
            linear_table(m.src_h, m.out_h, tabs);
        }
        auto step = [](int ov, double delta) { return ov > 1 ? delta / (double)(ov - 1) : 0.0; };
        D.st = step(D.ov_t, 1.0); D.sb = step(D.ov_b, -1.0); D.sl = step(D.ov_l, 1.0); D.sr = step(D.ov_r, -1.0);
        srcs[t].p = h_d_tiles[t];
        srcs[t].stride = h_strides[t];
        if (h_strides[t] < (int64_t)m.src_w * 3 * es) return sr_set_error(SR_ERR_SHAPE, "sr_feather_merge: tile %d stride too small", t);
    }
    const size_t b0 = sizeof(MergeDev) * (size_t)n, b1 = sizeof(TileSrc) * (size_t)n, b2 = sizeof(LinTab) * tabs.size();
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    void *scr = nullptr;
    int rc = ctx_scratch(ctx, al(b0) + al(b1) + al(b2) + 256, &scr);
    if (rc) return rc;
    char *p0 = (char *)scr, *p1 = p0 + al(b0), *p2 = p1 + al(b1);
    if (n > 0) {
        HIPCHK(upload_small(ctx, p0, md.data(), b0));
        HIPCHK(upload_small(ctx, p1, srcs.data(), b1));
        if (b2) HIPCHK(upload_small(ctx, p2, tabs.data(), b2));
    }
    {
        ProfScope ps(ctx, "feather_merge");
        dim3 grid((canvas_w + 255) / 256, (canvas_h + 3) / 4), block(64, 4);
        if (dtype == SR_U8)
            hipLaunchKernelGGL(k_feather_merge<SRC_U8>, grid, block, 0, ctx->stream, (const MergeDev *)p0, (const TileSrc *)p1,
                               (const LinTab *)p2, n, blending ? 1 : 0, d_canvas, (long long)canvas_stride, canvas_h, canvas_w);
        else
            hipLaunchKernelGGL(k_feather_merge<SRC_F32>, grid, block, 0, ctx->stream, (const MergeDev *)p0, (const TileSrc *)p1,
                               (const LinTab *)p2, n, blending ? 1 : 0, d_canvas, (long long)canvas_stride, canvas_h, canvas_w);
    }
    return check_launch("feather_merge");
}

// ---- metrics ------------------------------------------------------------------------------------------
int sr_sse_u8_async(sr_ctx *ctx, const uint8_t *d_a, int64_t stride_a, const uint8_t *d_b, int64_t stride_b, int h,
                    int64_t rowlen, uint64_t *d_sse)
{
    CTX_ENTER(ctx);
    if (!d_a || !d_b || !d_sse || h < 0 || rowlen < 0) return sr_set_error(SR_ERR_INVALID_ARG, "sr_sse_u8: bad arguments");
    HIPCHK(hipMemsetAsync(d_sse, 0, sizeof(uint64_t), ctx->stream));
    if (h == 0 || rowlen == 0) return SR_OK;
    const bool dense = stride_a == rowlen && stride_b == rowlen && ((uintptr_t)d_a % 16 == 0) && ((uintptr_t)d_b % 16 == 0);
    {
        ProfScope ps(ctx, "psnr_sse");
        if (dense) {
            const size_t total = (size_t)h * (size_t)rowlen;
            const size_t nvec = total / 16;
            const int ntail = (int)(total - nvec * 16);
            const int blocks = (int)std::min<size_t>((nvec + 255) / 256 + 1, 256 * 16);
            hipLaunchKernelGGL(k_sse_flat, dim3(blocks), dim3(256), 0, ctx->stream, (const uint4 *)d_a, (const uint4 *)d_b, nvec,
                               d_a + nvec * 16, d_b + nvec * 16, ntail, (unsigned long long *)d_sse);
        } else {
            const int gx = (int)std::min<int64_t>((rowlen + 255) / 256, 64);
            const int gy = std::min(h, 1024);
            hipLaunchKernelGGL(k_sse_rows, dim3(gx, gy), dim3(256), 0, ctx->stream, d_a, (long long)stride_a, d_b,
                               (long long)stride_b, h, (long long)rowlen, (unsigned long long *)d_sse);
        }
    }
    return check_launch("psnr_sse");
}

int sr_seam_scan(sr_ctx *ctx, const uint8_t *d_canvas, int64_t canvas_stride, int canvas_h, int canvas_w, int cn,
                 const sr_tile_rect *h_rects, void *const *h_d_tiles, const int64_t *h_strides, int n, int window,
                 int stride, int gray_shift, double threshold, sr_seam_record *h_out, int cap, int *h_count)
{
    CTX_ENTER(ctx);
    if (!d_canvas || !h_rects || !h_d_tiles || !h_strides || !h_count || n < 0 || (cap > 0 && !h_out))
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_seam_scan: null argument");
    if ((cn != 1 && cn != 3) || window < 1 || stride < 1 || (gray_shift != 14 && gray_shift != 15))
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_seam_scan: bad channel count / window / stride / gray_shift");
    *h_count = 0;
    std::vector<SeamTile> st;
    long long nwin = 0, nblk_cells = 0;
    for (int t = 0; t < n; ++t) {
        const sr_tile_rect &r = h_rects[t];
        if (r.x < 0 || r.y < 0 || r.w < 1 || r.h < 1 || !h_d_tiles[t]) return sr_set_error(SR_ERR_INVALID_ARG, "sr_seam_scan: bad tile %d", t);
        SeamTile T;
        T.p = (const unsigned char *)h_d_tiles[t];
        T.stride = h_strides[t];
        T.x = r.x; T.y = r.y; T.w = r.w; T.h = r.h;
        T.roi_w = std::min(r.x + r.w, canvas_w) - r.x;
        T.roi_h = std::min(r.y + r.h, canvas_h) - r.y;
        T.nwx = T.roi_w >= window ? (T.roi_w - window) / stride + 1 : 0;
        T.nwy = T.roi_h >= window ? (T.roi_h - window) / stride + 1 : 0;
        if (T.roi_w <= 0 || T.roi_h <= 0) T.nwx = T.nwy = 0;
        T.first = nwin;
        nwin += (long long)T.nwx * T.nwy;
        T.nbx = (T.nwx + SEAM_CX - 2) / (SEAM_CX - 1);
        T.pad = 0;
        T.bfirst = nblk_cells;
        nblk_cells += (long long)T.nbx * ((T.nwy + SEAM_CY - 2) / (SEAM_CY - 1));
        st.push_back(T);
    }
    if (nwin == 0) return SR_OK;
    const size_t b_tiles = (sizeof(SeamTile) * st.size() + 255) / 256 * 256;
    const size_t b_out = sizeof(SeamRec) * (size_t)std::max(cap, 1);
    void *scr = nullptr;
    int rc = ctx_scratch(ctx, b_tiles + b_out + 512, &scr);
    if (rc) return rc;
    SeamTile *d_t = (SeamTile *)scr;
    int *d_cnt = (int *)((char *)scr + b_tiles);
    SeamRec *d_out = (SeamRec *)((char *)scr + b_tiles + 256);
    HIPCHK(upload_small(ctx, d_t, st.data(), sizeof(SeamTile) * st.size()));
    HIPCHK(hipMemsetAsync(d_cnt, 0, sizeof(int), ctx->stream));
    {
        ProfScope ps(ctx, "seam_scan");
        const double c1 = (0.01 * 255.0) * (0.01 * 255.0), c2 = (0.03 * 255.0) * (0.03 * 255.0);
        if (window == 16 && stride == 8 && nblk_cells < (1ll << 31)) {         // the reference's default geometry: cell kernel
            const dim3 block(SEAM_CX, SEAM_CY);
            if (cn == 3) hipLaunchKernelGGL(k_seam_scan_cells<3>, dim3((unsigned)nblk_cells), block, 0, ctx->stream, d_canvas, (long long)canvas_stride,
                                            (const SeamTile *)d_t, (int)st.size(), gray_shift, threshold, c1, c2, d_out, cap, d_cnt);
            else hipLaunchKernelGGL(k_seam_scan_cells<1>, dim3((unsigned)nblk_cells), block, 0, ctx->stream, d_canvas, (long long)canvas_stride,
                                    (const SeamTile *)d_t, (int)st.size(), gray_shift, threshold, c1, c2, d_out, cap, d_cnt);
        } else {
            const long long blocks = (nwin + 255) / 256;
            hipLaunchKernelGGL(k_seam_scan, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_canvas, (long long)canvas_stride, cn,
                               (const SeamTile *)d_t, (int)st.size(), nwin, window, stride, gray_shift, threshold, c1, c2, d_out, cap, d_cnt);
        }
    }
    rc = check_launch("seam_scan");
    if (rc) return rc;
    int cnt = 0;
    HIPCHK(hipMemcpyAsync(&cnt, d_cnt, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(stream_sync(ctx));
    *h_count = cnt;
    const int ncopy = std::min(cnt, cap);
    if (ncopy > 0) {
        static_assert(sizeof(SeamRec) == sizeof(sr_seam_record), "record layouts must match");
        HIPCHK(hipMemcpyAsync(h_out, d_out, sizeof(SeamRec) * (size_t)ncopy, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(stream_sync(ctx));
    }
    return SR_OK;
}

int sr_sse_f32(sr_ctx *ctx, const float *d_a, int64_t stride_a, const float *d_b, int64_t stride_b, int h,
               int64_t rowlen, double *h_sse)
{
    CTX_ENTER(ctx);
    if (!d_a || !d_b || !h_sse || h < 0 || rowlen < 0) return sr_set_error(SR_ERR_INVALID_ARG, "sr_sse_f32: bad arguments");
    *h_sse = 0.0;
    if (h == 0 || rowlen == 0) return SR_OK;
    const int gx = (int)std::min<int64_t>((rowlen + 255) / 256, 64), gy = std::min(h, 1024);
    void *scr = nullptr;
    int rc = ctx_scratch(ctx, (size_t)8 << 20, &scr);
    if (rc) return rc;
    double *part = (double *)scr, *r0 = part + (size_t)gx * gy + 32, *r1 = r0 + 4096;
    {
        ProfScope ps(ctx, "psnr_sse_f32");
        hipLaunchKernelGGL(k_sse_f32, dim3(gx, gy), dim3(256), 0, ctx->stream, d_a, (long long)stride_a, d_b, (long long)stride_b,
                           h, (long long)rowlen, part);
    }
    const double *res = reduce_partials(ctx, part, (long long)gx * gy, 1, r0, r1);
    rc = check_launch("psnr_sse_f32");
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(h_sse, res, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(stream_sync(ctx));
    return SR_OK;
}

int sr_weighted_blend_custom(sr_blend_plan *plan, int dtype, void *const *h_d_tiles, const int64_t *h_strides,
                             const float *const *h_d_weights, const int64_t *h_weight_strides, uint8_t *d_canvas,
                             int64_t canvas_stride, float *d_canvas_f32)
{
    if (!plan_is_live(plan)) return sr_set_error(SR_ERR_INVALID_ARG, "sr_weighted_blend_custom: null or destroyed plan");
    CTX_ENTER(plan->ctx);
    sr_blend_plan *P = plan;
    sr_ctx *c = P->ctx;
    int rc = blend_check_tiles(P, dtype, h_d_tiles, h_strides);
    if (rc) return rc;
    if (!h_d_weights || !h_weight_strides || !d_canvas) return sr_set_error(SR_ERR_INVALID_ARG, "sr_weighted_blend_custom: null argument");
    if (canvas_stride < (int64_t)P->canvas_w * P->cn) return sr_set_error(SR_ERR_SHAPE, "sr_weighted_blend_custom: canvas stride too small");
    std::vector<TileSrc> srcs(P->n);
    std::vector<CustomW> wts(P->n);
    for (int t = 0; t < P->n; ++t) {
        if (!h_d_weights[t]) return sr_set_error(SR_ERR_INVALID_ARG, "sr_weighted_blend_custom: weight map %d is null", t);
        if (h_weight_strides[t] < (int64_t)P->tiles[t].w * 4) return sr_set_error(SR_ERR_SHAPE, "sr_weighted_blend_custom: weight map %d stride too small", t);
        srcs[t].p = h_d_tiles[t];
        srcs[t].stride = h_strides[t];
        wts[t].w = h_d_weights[t];
        wts[t].stride = h_weight_strides[t];
    }
    const int rows = P->row_end - P->row_begin;
    if (rows <= 0) return SR_OK;
    void *scr = nullptr;
    rc = ctx_scratch(c, sizeof(CustomW) * P->n + 256, &scr);
    if (rc) return rc;
    HIPCHK(upload_if_changed(c, P->d_srcs, srcs.data(), sizeof(TileSrc) * P->n, P->sh_srcs));
    HIPCHK(upload_small(c, scr, wts.data(), sizeof(CustomW) * P->n));
    {
        ProfScope ps(c, "weighted_custom");
        dim3 grid((P->canvas_w + 63) / 64, (rows + 3) / 4), block(64, 4);
        if (dtype == SR_U8) hipLaunchKernelGGL(k_weighted_custom<SRC_U8>, grid, block, 0, c->stream, P->d_tiles, P->d_srcs, (const CustomW *)scr, P->n, P->cn, d_canvas, (long long)canvas_stride, d_canvas_f32, P->canvas_w, P->row_begin, P->row_end);
        else hipLaunchKernelGGL(k_weighted_custom<SRC_F32>, grid, block, 0, c->stream, P->d_tiles, P->d_srcs, (const CustomW *)scr, P->n, P->cn, d_canvas, (long long)canvas_stride, d_canvas_f32, P->canvas_w, P->row_begin, P->row_end);
    }
    return check_launch("weighted_custom");
}

int sr_sse_u8(sr_ctx *ctx, const uint8_t *d_a, int64_t stride_a, const uint8_t *d_b, int64_t stride_b, int h,
              int64_t rowlen, uint64_t *h_sse)
{
    CTX_ENTER(ctx);
    if (!h_sse) return sr_set_error(SR_ERR_INVALID_ARG, "sr_sse_u8: null result");
    void *scr = nullptr;
    int rc = ctx_scratch(ctx, 64, &scr);
    if (rc) return rc;
    rc = sr_sse_u8_async(ctx, d_a, stride_a, d_b, stride_b, h, rowlen, (uint64_t *)scr);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(h_sse, scr, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(stream_sync(ctx));
    return SR_OK;
}

// ---- fused assessment ---------------------------------------------------------------------------------------
}  // extern "C"

// Stage 1 of the resized assessment: one thread = 4 consecutive pixels of the RESIZED images; cv2.resize(INTER_CUBIC) sample of
// both images (cubic_sample: the arithmetic of k_resize_cubic), gray of each, the squared channel differences.  Writes the
// two gray planes and one SSE partial per block (exact: integers, < 2^53).
template <int CN>
__global__ __launch_bounds__(256) void k_resize_gray_pair(const unsigned char *__restrict__ a, long long sa,
                                                          const unsigned char *__restrict__ b, long long sb, int sh, int sw,
                                                          const CubicTab *__restrict__ xt, const CubicTab *__restrict__ yt, int dh,
                                                          int dw, int shift, unsigned char *__restrict__ ga,
                                                          unsigned char *__restrict__ gb, long long pitch, double *__restrict__ part)
{
    __shared__ double ws[4];
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4, y = blockIdx.y * 4 + threadIdx.y;
    unsigned sse = 0, pa = 0, pb = 0;
    if (y < dh && x0 < dw) {
        const CubicTab Y = yt[y];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = min(x0 + k, dw - 1);              // columns past the end repeat the last one and are not counted
            int va[CN], vb[CN];
            cubic_sample<CN>(a, sa, sh, sw, xt[x], Y, va);
            cubic_sample<CN>(b, sb, sh, sw, xt[x], Y, vb);
            const int g0 = CN == 3 ? gray_rgb(va[0], va[1], va[2], shift) : va[0];
            const int g1 = CN == 3 ? gray_rgb(vb[0], vb[1], vb[2], shift) : vb[0];
            pa |= (unsigned)g0 << (8 * k);
            pb |= (unsigned)g1 << (8 * k);
            if (x0 + k < dw) {
#pragma unroll
                for (int c = 0; c < CN; ++c) sse += (unsigned)((va[c] - vb[c]) * (va[c] - vb[c]));
            }
        }
        *(unsigned *)(ga + (size_t)y * pitch + x0) = pa;    // pitch is a multiple of 64: the row padding takes the tail
        *(unsigned *)(gb + (size_t)y * pitch + x0) = pb;
    }
    const double s = wave_sum_f64((double)sse);
    const int tid = threadIdx.y * 64 + threadIdx.x;
    if ((tid & 63) == 0) ws[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

// Down-sampling form of stage 1 (RGB, destination smaller than the source): the 4 x 4 taps of neighbouring destination pixels
// lie 1 / scale pixels apart, so the gather above issues 12-byte loads that share no cache line between lanes (TA-bound:
// 1.59 ms for the three scales of a 200 MP pair against 0.45 ms of HBM traffic).  Here a block of 64 x 4 destination pixels
// first copies the source window it needs -- the 4 source rows of each of its 4 destination rows, from the first to the last
// tap column, both images -- into LDS with coalesced 16-byte loads, and the taps are read from LDS.  Same integers as
// cubic_sample.  Launched when the window fits 64 KB of LDS (scales down to about 0.1).
__device__ __forceinline__ void lds_tap12(const unsigned char *__restrict__ row, int byte_off, unsigned (&wd)[3])
{
    // 12 bytes at any byte offset of an LDS row: four aligned dwords and a funnel shift
    const unsigned *p = (const unsigned *)(row + (byte_off & ~3));
    const unsigned d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], m = (unsigned)(byte_off & 3);
    wd[0] = __builtin_amdgcn_alignbyte(d1, d0, m);
    wd[1] = __builtin_amdgcn_alignbyte(d2, d1, m);
    wd[2] = __builtin_amdgcn_alignbyte(d3, d2, m);
}

__global__ __launch_bounds__(256) void k_resize_gray_pair_lds(const unsigned char *__restrict__ a, long long sa,
                                                              const unsigned char *__restrict__ b, long long sb, int sh, int sw,
                                                              const CubicTab *__restrict__ xt, const CubicTab *__restrict__ yt,
                                                              int dh, int dw, int shift, unsigned char *__restrict__ ga,
                                                              unsigned char *__restrict__ gb, long long pitch, int lds_pitch,
                                                              double *__restrict__ part)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char win[];      // [image 2][slot 16][lds_pitch]
    __shared__ double ws[4];
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * 64 + tx;
    const int bx0 = blockIdx.x * 64, by0 = blockIdx.y * 4;
    // window columns: first tap of the first pixel .. last tap of the last pixel, clamped to the image; in bytes, rounded
    // down to 16 at the start
    const int xl = bx0, xr = min(bx0 + 63, dw - 1);
    const int c_lo = max(xt[xl].ofs - 1, 0), c_hi = min(xt[xr].ofs + 2, sw - 1);
    const int byte0 = (c_lo * 3) & ~15, nbytes = (c_hi + 1) * 3 - byte0;       // nbytes <= lds_pitch - 16 (host sizes it)
    const int nchunk = (nbytes + 15) >> 4, rowbytes = sw * 3;
    for (int e = tid; e < 32 * nchunk; e += 256) {
        const int slot = e / nchunk, ck = e - slot * nchunk;                    // slot = image * 16 + dst row * 4 + tap
        const int img = slot >> 4, j = (slot >> 2) & 3, t = slot & 3;
        const int y = min(by0 + j, dh - 1);
        const int srow = min(max(yt[y].ofs - 1 + t, 0), sh - 1);
        const unsigned char *src = (img ? b + (size_t)srow * sb : a + (size_t)srow * sa);
        const int off = byte0 + 16 * ck;
        u4_t v;
        if (off + 16 <= rowbytes) {
            v = *(const __attribute__((address_space(1))) u4_a1_t *)(src + off);
        } else {                                                                // the row's last bytes: never read past it
            unsigned w4[4] = {0u, 0u, 0u, 0u};
            for (int i = 0; i < 16 && off + i < rowbytes; ++i) w4[i >> 2] |= (unsigned)src[off + i] << (8 * (i & 3));
            v.x = w4[0]; v.y = w4[1]; v.z = w4[2]; v.w = w4[3];
        }
        *(u4_t *)(win + (size_t)slot * lds_pitch + 16 * ck) = v;
    }
    __syncthreads();
    const int x = bx0 + tx, y = by0 + ty;
    unsigned sse = 0;
    if (x < dw && y < dh) {
        const CubicTab X = xt[x], Y = yt[y];
        const bool inner = X.ofs - 1 >= 0 && X.ofs + 2 <= sw - 1;
        int va[3], vb[3];
#pragma unroll
        for (int img = 0; img < 2; ++img) {
            int acc[3] = {0, 0, 0};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const unsigned char *row = win + (size_t)(img * 16 + ty * 4 + t) * lds_pitch;
                int v[4][3];
                if (inner) {
                    unsigned wd[3];
                    lds_tap12(row, (X.ofs - 1) * 3 - byte0, wd);
#pragma unroll
                    for (int q = 0; q < 12; ++q) v[q / 3][q % 3] = (int)((wd[q >> 2] >> (8 * (q & 3))) & 0xFFu);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int sx = min(max(X.ofs + k - 1, 0), sw - 1) * 3 - byte0;
#pragma unroll
                        for (int c = 0; c < 3; ++c) v[k][c] = (int)row[sx + c];
                    }
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int hs = v[0][c] * X.c[0] + v[1][c] * X.c[1] + v[2][c] * X.c[2] + v[3][c] * X.c[3];
                    acc[c] += hs * (int)Y.c[t];
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int r = (acc[c] + (1 << 21)) >> 22;
                (img ? vb : va)[c] = r < 0 ? 0 : (r > 255 ? 255 : r);
            }
        }
        ga[(size_t)y * pitch + x] = (unsigned char)gray_rgb(va[0], va[1], va[2], shift);
        gb[(size_t)y * pitch + x] = (unsigned char)gray_rgb(vb[0], vb[1], vb[2], shift);
#pragma unroll
        for (int c = 0; c < 3; ++c) sse += (unsigned)((va[c] - vb[c]) * (va[c] - vb[c]));
    }
    const double sred = wave_sum_f64((double)sse);
    if ((tid & 63) == 0) ws[tid >> 6] = sred;
    __syncthreads();
    if (tid == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

// Round 3: the sampler for gentle down-sampling as a COLUMN MARCH.  cv2's INTER_CUBIC on u8 is exact integer arithmetic and
// separable -- sum_t cy[t] * (sum_k cx[k] * s[t][k]) -- so a lane that owns one destination column walks down the source
// rows, forms the horizontal 4-tap sums of each row ONCE (both images, three channels; a row feeds the 1.6 destination rows
// whose windows contain it), keeps the last four rows' sums in registers and finishes a destination pixel whenever the row
// just pushed is a window's last.  The byte pairs are formed with v_perm_b32 and multiplied with v_dot2_i32_i16 (two taps
// per instruction: the coefficients are 16-bit): ~150 VALU instructions per destination pixel instead of ~270.
// A wave owns RGM_SEG destination rows of 64 columns and is autonomous -- no block barrier.  The source rows reach the lanes
// through a WAVE-PRIVATE LDS window: the wave copies the contiguous span of each row it needs with aligned 16-byte loads
// (five or six cache-line accesses per row) and every lane then picks its 12 bytes from LDS.  Letting each lane load its own
// unaligned 12 bytes from global memory was measured first: 49 L1 accesses per wave instruction (TCP_TOTAL_CACHE_ACCESSES /
// TA_FLAT_READ_WAVEFRONTS), the L1 tag pipeline 75 % busy, 0.49 ms at x0.4.  The next group's rows are requested before the
// current group is processed.  Same integers as cubic_sample.
#ifndef RGM_SEG
#define RGM_SEG 16
#endif
typedef short rg_s2_t __attribute__((ext_vector_type(2)));

// horizontal 4-tap sums of one row's 12 bytes (4 pixels x RGB): hs[c] = sum_k px[k][c] * cx[k]
__device__ __forceinline__ void rgm_hrow(unsigned q0, unsigned q1, unsigned q2, rg_s2_t c01, rg_s2_t c23, int (&hs)[3])
{
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // bytes c and 3 + c of (q0, q1); bytes 6 + c and 9 + c of (q1, q2), each zero-extended to 16 bits
        const unsigned pa = __builtin_amdgcn_perm(q1, q0, 0x0c000c00u | (unsigned)c | ((unsigned)(3 + c) << 16));
        const unsigned pb = __builtin_amdgcn_perm(q2, q1, 0x0c000c00u | (unsigned)(2 + c) | ((unsigned)(5 + c) << 16));
        hs[c] = __builtin_amdgcn_sdot2(__builtin_bit_cast(rg_s2_t, pa), c01,
                                       __builtin_amdgcn_sdot2(__builtin_bit_cast(rg_s2_t, pb), c23, 0, false), false);
    }
}

// one 16-byte chunk of a source row's window.  A chunk that crosses the end of its row reads on into the next row (bytes
// no tap uses); only on the image's LAST row would that leave the buffer, and there the bytes are fetched one by one.
__device__ __forceinline__ u4_t rgm_load_chunk(const unsigned char *__restrict__ row, int off, int rowbytes, bool last_row)
{
    if (!last_row || off + 16 <= rowbytes) return *(const __attribute__((address_space(1))) u4_a1_t *)(row + off);
    unsigned w4[4] = {0u, 0u, 0u, 0u};
#pragma unroll 1
    for (int i = 0; i < 16 && off + i < rowbytes; ++i) w4[i >> 2] |= (unsigned)row[off + i] << (8 * (i & 3));
    u4_t v;
    v.x = w4[0]; v.y = w4[1]; v.z = w4[2]; v.w = w4[3];
    return v;
}

#ifndef RGM_MINB
#define RGM_MINB 5
#endif
__global__ __launch_bounds__(256, RGM_MINB) void k_resize_gray_pair_march(const unsigned char *__restrict__ a, long long sa,
                                                                const unsigned char *__restrict__ b, long long sb, int sh, int sw,
                                                                const CubicTab *__restrict__ xt, const CubicTab *__restrict__ yt,
                                                                int dh, int dw, int shift, unsigned char *__restrict__ ga,
                                                                unsigned char *__restrict__ gb, long long pitch, int lds_pitch,
                                                                int cols, double *__restrict__ part)
{
    // cols = destination columns per wave: 64, or 32 for scales below 0.19 whose 64-column window would exceed the 64
    // chunks a wave loads per row (all 64 lanes still load; the upper 32 sample a repeated column and store nothing -- at
    // those scales the kernel is bound by its loads, there are 25 x fewer destination than source pixels)
    extern __shared__ __attribute__((aligned(16))) unsigned char win[];      // [wave 4][image 2][row 4][lds_pitch]
    __shared__ double ws[4];
    const int tx = threadIdx.x, tid = threadIdx.y * 64 + tx;
    const int bx0 = blockIdx.x * cols, x = tx < cols ? bx0 + tx : dw;
    const int ys = __builtin_amdgcn_readfirstlane((blockIdx.y * 4 + threadIdx.y) * RGM_SEG), ye = min(ys + RGM_SEG, dh);
    unsigned sse = 0;
    if (ys < dh) {                                                  // wave-uniform
        unsigned char *mine = win + (size_t)threadIdx.y * 8 * lds_pitch;
        const int c_lo = max(xt[bx0].ofs - 1, 0), c_hi = min(xt[min(bx0 + cols - 1, dw - 1)].ofs + 2, sw - 1);
        const int byte0 = (c_lo * 3) & ~15, nchunk = ((c_hi + 1) * 3 - byte0 + 15) >> 4, rowbytes = sw * 3;   // nchunk <= 64 (host)
        const CubicTab X = xt[min(x, min(bx0 + cols, dw) - 1)];     // lanes past the end repeat the wave's last column, store nothing
        const bool inner = X.ofs - 1 >= 0 && X.ofs + 2 <= sw - 1;
        const int loff = max(X.ofs - 1, 0) * 3 - byte0;             // this lane's 12 bytes inside the window
        rg_s2_t c01, c23;
        c01.x = X.c[0]; c01.y = X.c[1]; c23.x = X.c[2]; c23.y = X.c[3];
        int y = ys, yofs = yt[ys].ofs;
        const int r_end = yt[ye - 1].ofs + 2;
        int ha[4][3], hb[4][3];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int c = 0; c < 3; ++c) ha[p][c] = hb[p][c] = 0;
        u4_t va4[4], vb4[4];                                        // the rows in flight (lane = chunk)
        auto request = [&](int r) {                                 // unclamped row numbers: rows beyond the image repeat the edge
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int sr = min(max(r + p, 0), sh - 1);
                if (tx < nchunk) {
                    va4[p] = rgm_load_chunk(a + (size_t)sr * sa, byte0 + 16 * tx, rowbytes, sr == sh - 1);
                    vb4[p] = rgm_load_chunk(b + (size_t)sr * sb, byte0 + 16 * tx, rowbytes, sr == sh - 1);
                }
            }
        };
        request(yofs - 1);
        for (int r = yofs - 1, r_next; r <= r_end; r = r_next) {
            // where the next group of four rows starts: right below this one, or -- when the windows lie further apart than
            // four rows (scales below 0.25) -- at the first row of the next window still to be finished (scalar bookkeeping)
            {
                int yn = y, yo = yofs;
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if (yn < ye && yo + 2 == r + p) {
                        ++yn;
                        yo = yn < ye ? yt[yn].ofs : 0x3fffffff;
                    }
                r_next = yn < ye ? max(r + 4, yo - 1) : r_end + 1;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the previous group's LDS reads are done
            __builtin_amdgcn_wave_barrier();
            if (tx < nchunk) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    *(u4_t *)(mine + (size_t)p * lds_pitch + 16 * tx) = va4[p];
                    *(u4_t *)(mine + (size_t)(4 + p) * lds_pitch + 16 * tx) = vb4[p];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (r_next <= r_end) request(r_next);                   // flies under this group's arithmetic
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (r + p <= r_end) {                               // wave-uniform
                    const unsigned char *ra = mine + (size_t)p * lds_pitch, *rb = mine + (size_t)(4 + p) * lds_pitch;
                    unsigned wa[3], wb[3];
                    if (inner) {
                        lds_tap12(ra, loff, wa);
                        lds_tap12(rb, loff, wb);
                    } else {                                        // image border columns: taps clamped one by one
                        wa[0] = wa[1] = wa[2] = wb[0] = wb[1] = wb[2] = 0u;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int sx = min(max(X.ofs + k - 1, 0), sw - 1) * 3 - byte0;
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                const int i = 3 * k + c;
                                wa[i >> 2] |= (unsigned)ra[sx + c] << (8 * (i & 3));
                                wb[i >> 2] |= (unsigned)rb[sx + c] << (8 * (i & 3));
                            }
                        }
                    }
                    rgm_hrow(wa[0], wa[1], wa[2], c01, c23, ha[p]);
                    rgm_hrow(wb[0], wb[1], wb[2], c01, c23, hb[p]);
                    if (y < ye && yofs + 2 == r + p) {              // this row completes the window of destination row y
                        const CubicTab Y = yt[y];
                        int va[3], vb[3];
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            int s0 = 1 << 21, s1 = 1 << 21;
#pragma unroll
                            for (int t = 0; t < 4; ++t) {           // tap t = row r + p - 3 + t = ring slot (p + 1 + t) & 3
                                s0 += ha[(p + 1 + t) & 3][c] * (int)Y.c[t];
                                s1 += hb[(p + 1 + t) & 3][c] * (int)Y.c[t];
                            }
                            va[c] = min(max(s0 >> 22, 0), 255);
                            vb[c] = min(max(s1 >> 22, 0), 255);
                        }
                        if (x < dw) {
                            ga[(size_t)y * pitch + x] = (unsigned char)gray_rgb(va[0], va[1], va[2], shift);
                            gb[(size_t)y * pitch + x] = (unsigned char)gray_rgb(vb[0], vb[1], vb[2], shift);
#pragma unroll
                            for (int c = 0; c < 3; ++c) sse += (unsigned)((va[c] - vb[c]) * (va[c] - vb[c]));
                        }
                        ++y;
                        yofs = y < ye ? yt[y].ofs : 0x3fffffff;
                    }
                }
            }
        }
    }
    const double sred = wave_sum_f64((double)sse);
    if ((tid & 63) == 0) ws[tid >> 6] = sred;
    __syncthreads();
    if (tid == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

__global__ void k_store_sse(const double *__restrict__ src, sr_assess_sums *__restrict__ out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) out->sse = src[0];
}

// Final reduction of the assessment: ONE block sums the per-block partials part[i * 4 + comp] (fixed order: a
// strided serial sum per thread, then a fixed tree -- deterministic) and writes the four sums.
__global__ __launch_bounds__(256) void k_assess_finish(const double *__restrict__ part, long long n, int flags,
                                                       sr_assess_sums *__restrict__ out)
{
    __shared__ double sh[4][256];
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (long long i = threadIdx.x; i < n; i += 256) {
#pragma unroll
        for (int c = 0; c < 4; ++c) s[c] += part[i * 4 + c];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) sh[c][threadIdx.x] = s[c];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
#pragma unroll
            for (int c = 0; c < 4; ++c) sh[c][threadIdx.x] += sh[c][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out->ssim_gauss = (flags & ASSESS_GAUSS) ? sh[0][0] : 0.0;
        out->ssim_simple = (flags & ASSESS_SIMPLE) ? sh[1][0] : 0.0;
        out->sse = (flags & ASSESS_SSE) ? sh[2][0] : 0.0;
        out->ssim_uniform = (flags & ASSESS_UNIFORM) ? sh[3][0] : 0.0;
    }
}

__global__ void k_assess_store(const double *__restrict__ g, const double *__restrict__ u, int flags,
                               sr_assess_sums *__restrict__ out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    out->ssim_gauss = (g && (flags & ASSESS_GAUSS)) ? g[0] : 0.0;
    out->ssim_simple = (g && (flags & ASSESS_SIMPLE)) ? g[1] : 0.0;
    out->sse = (g && (flags & ASSESS_SSE)) ? g[2] : 0.0;
    out->ssim_uniform = (u && (flags & ASSESS_UNIFORM)) ? u[0] : 0.0;
}

// reduce part[n][ncomp] -> returns pointer (inside the two ping-pong buffers) holding ncomp results
// ---------------------------------------------------------------------------------------------
// SSIM on FLOAT images.  The reference hands skimage / cv2 whatever _preprocess_image returns: float arrays whose
// maximum exceeds 1 stay float (quality_assessment_module.py:169-195,351-417).  A plain separable float64 form of
// oracle_np.ssim, not a tuned kernel (API convenience path; the u8 march above is the hot one):
//   k_ssimf_gray : gray planes in float64 (float32 RGB: cv2's float cvtColor, ((R*0.299f) + G*0.587f) + B*0.114f in fp32)
//   k_ssimf_rows : horizontal window sums of x, y, x*x, y*y, x*y (five float64 planes)
//   k_ssimf_cols : vertical window sums, the SSIM value of the mode, validity, per-block partial sums
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_ssimf_gray(const T *__restrict__ img, long long stride_bytes, int h, int w, int cn,
                                                    double *__restrict__ out)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    const T *row = (const T *)((const char *)img + (size_t)y * stride_bytes);
    double g;
    if (cn == 1) g = (double)row[x];
    else {
        const float r = (float)row[3 * x], gg = (float)row[3 * x + 1], b = (float)row[3 * x + 2];
        g = (double)(((r * 0.299f) + gg * 0.587f) + b * 0.114f);
    }
    out[(size_t)y * w + x] = g;
}

struct SsimFParams {
    int h, w, mode, radius, bmode, crop, row_begin, row_end;
    double c1, c2, cov_norm;
    double k[11];
};

__global__ __launch_bounds__(256) void k_ssimf_rows(const double *__restrict__ ga, const double *__restrict__ gb, SsimFParams P,
                                                    double *__restrict__ tmp)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= P.w || y >= P.h) return;
    const size_t plane = (size_t)P.h * P.w;
    const double *ra = ga + (size_t)y * P.w, *rb = gb + (size_t)y * P.w;
    double sx = 0.0, sy = 0.0, sxx = 0.0, syy = 0.0, sxy = 0.0;
    for (int j = 0; j < 2 * P.radius + 1; ++j) {
        const int xi = border_index(x + j - P.radius, P.w, P.bmode);
        const double a = ra[xi], b = rb[xi], kj = P.k[j];
        sx += a * kj;
        sy += b * kj;
        sxx += (a * a) * kj;
        syy += (b * b) * kj;
        sxy += (a * b) * kj;
    }
    const size_t o = (size_t)y * P.w + x;
    tmp[o] = sx; tmp[plane + o] = sy; tmp[2 * plane + o] = sxx; tmp[3 * plane + o] = syy; tmp[4 * plane + o] = sxy;
}

__global__ __launch_bounds__(256) void k_ssimf_cols(const double *__restrict__ tmp, SsimFParams P, double *__restrict__ part)
{
    __shared__ double ws[4];
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    double v = 0.0;
    if (x < P.w && y < P.h && y >= max(P.crop, P.row_begin) && y < min(P.h - P.crop, P.row_end) && x >= P.crop && x < P.w - P.crop) {
        const size_t plane = (size_t)P.h * P.w;
        double u[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
        for (int j = 0; j < 2 * P.radius + 1; ++j) {
            const size_t o = (size_t)border_index(y + j - P.radius, P.h, P.bmode) * P.w + x;
            const double kj = P.k[j];
#pragma unroll
            for (int m = 0; m < 5; ++m) u[m] += tmp[m * plane + o] * kj;
        }
        const double ux = u[0], uy = u[1], uxx = u[2], uyy = u[3], uxy = u[4];
        if (P.mode == SR_SSIM_SIMPLE) {
            const double m1 = ux * ux, m2 = uy * uy, m12 = ux * uy;
            const double s1 = uxx - m1, s2 = uyy - m2, s12 = uxy - m12;
            v = ((2.0 * m12 + P.c1) * (2.0 * s12 + P.c2)) / ((m1 + m2 + P.c1) * (s1 + s2 + P.c2));
        } else {
            const double vx = P.cov_norm * (uxx - ux * ux), vy = P.cov_norm * (uyy - uy * uy), vxy = P.cov_norm * (uxy - ux * uy);
            const double a1 = 2.0 * ux * uy + P.c1, a2 = 2.0 * vxy + P.c2;
            const double b1 = ux * ux + uy * uy + P.c1, b2 = vx + vy + P.c2;
            v = (a1 * a2) / (b1 * b2);
        }
    }
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const double sred = wave_sum_f64(v);
    if ((tid & 63) == 0) ws[tid >> 6] = sred;
    __syncthreads();
    if (tid == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

const double *reduce_partials(sr_ctx *ctx, const double *part, long long n, int ncomp, double *buf0, double *buf1)
{
    const double *src = part;
    double *dst = buf0;
    while (true) {
        const long long nb = (n + 1023) / 1024;
        hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)nb), dim3(256), 0, ctx->stream, src, n, ncomp, dst);
        if (nb == 1) return dst;
        src = dst;
        dst = (dst == buf0) ? buf1 : buf0;
        n = nb;
    }
}

extern "C" {

static void gauss_taps(double *k6)
{
    double k[11], sum = 0.0;
    for (int i = 0; i < 11; ++i) {
        const double x = i - 5;
        k[i] = std::exp(-0.5 / (1.5 * 1.5) * x * x);     // scipy.ndimage._gaussian_kernel1d(sigma=1.5, radius=5)
        sum += k[i];
    }
    for (int j = 0; j <= 5; ++j) k6[j] = k[5 + j] / sum;
}

int sr_ssim_count(int h, int w, int mode, int row_begin, int row_end, uint64_t *count)
{
    if (!count || h < 1 || w < 1) return sr_set_error(SR_ERR_INVALID_ARG, "sr_ssim_count: bad arguments");
    int pad;
    if (mode == SR_SSIM_UNIFORM7) pad = 3;
    else if (mode == SR_SSIM_GAUSS11) pad = 5;
    else if (mode == SR_SSIM_SIMPLE) pad = 0;
    else return sr_set_error(SR_ERR_INVALID_ARG, "sr_ssim_count: unknown mode %d", mode);
    const long long y0 = std::max(pad, row_begin), y1 = std::min(h - pad, row_end), nx = (long long)w - 2 * pad;
    *count = (y1 > y0 && nx > 0) ? (uint64_t)((y1 - y0) * nx) : 0;
    return SR_OK;
}

// Shared body of sr_assess_u8_async / sr_assess_resized_u8_async (which hands it the resized gray planes).
static int assess_impl(sr_ctx *ctx, const uint8_t *d_a, int64_t stride_a, const uint8_t *d_b, int64_t stride_b, int h,
                       int w, int cn, int gray_shift, double data_range, int row_begin, int row_end, int flags,
                       sr_assess_sums *d_out, const char *scope)
{
    if (!d_a || !d_b || !d_out) return sr_set_error(SR_ERR_INVALID_ARG, "%s: null argument", scope);
    if (h < 1 || w < 1 || (cn != 1 && cn != 3)) return sr_set_error(SR_ERR_INVALID_ARG, "%s: need h,w >= 1 and 1 or 3 channels", scope);
    if (gray_shift != 14 && gray_shift != 15) return sr_set_error(SR_ERR_INVALID_ARG, "%s: gray_shift must be 14 or 15", scope);
    const int64_t min_stride = (int64_t)w * cn;
    if (stride_a < min_stride || stride_b < min_stride) return sr_set_error(SR_ERR_SHAPE, "%s: stride smaller than a row", scope);
    row_begin = std::max(row_begin, 0);
    row_end = std::min(row_end, h);
    AssessParams P;
    memset(&P, 0, sizeof(P));
    P.h = h; P.w = w; P.shift = gray_shift; P.ry0 = row_begin; P.ry1 = row_end; P.flags = flags;
    P.c1a = (0.01 * data_range) * (0.01 * data_range);
    P.c2a = (0.03 * data_range) * (0.03 * data_range);
    P.c1b = (0.01 * 255.0) * (0.01 * 255.0);
    P.c2b = (0.03 * 255.0) * (0.03 * 255.0);
    P.same_c = (P.c1a == P.c1b && P.c2a == P.c2b) ? 1 : 0;
    P.k1u = 2401.0 * P.c1a;
    P.k2u = 2352.0 * P.c2a;
    gauss_taps(P.k);
    const int rows = row_end - row_begin;
    if (rows > 0 && (flags & ASSESS_ALL_BITS)) {
        // Blocks are equal work, 3 resident per CU: pick the chunk count that minimises (rounds of blocks) x (rows a
        // block marches) -- long blocks amortise the 10-row halo, short ones avoid a mostly empty last round on strips.
        const long long gbx = (w + AM_TX - 1) / AM_TX;
        const long long slots = (long long)std::max(ctx->num_cu, 1) * 3;
        // Measured (profiles/r02_assess_nch.json): longer blocks do NOT pay although they amortise the 10-row halo and
        // can fill the chip in exact rounds -- 25 / 33 / 49 chunks run 1.60 / 1.64 / 1.70 ms against 1.55 ms for 12: blocks
        // that start together stay in lockstep, so every wave of a CU sits in its load phase (or its fp64 march) at
        // the same time; many short blocks drift apart and overlap the two.
        long long best_cost = -1;
        for (int n = 2; n <= AM_NCH_MAX; ++n) {
            const int ty = AM_CH * n - 2 * AM_R;
            const long long blocks = gbx * ((rows + ty - 1) / ty);
            const long long cost = ((blocks + slots - 1) / slots) * (AM_CH * n);
            if (best_cost < 0 || cost <= best_cost) { best_cost = cost; P.nch = n; P.ty = ty; }
        }
        const long long gby = (rows + P.ty - 1) / P.ty;
        const size_t nblk = (size_t)(gbx * gby);
        void *scr = nullptr;
        int rc = ctx_scratch(ctx, std::max<size_t>(nblk * 4 * 8 + 512, (size_t)8 << 20), &scr);
        if (rc) return rc;
        double *part = (double *)scr;
        {
            ProfScope ps(ctx, scope);
            const dim3 grid((unsigned)gbx, (unsigned)gby), block(AM_TX);
            static const size_t lds_pad = std::getenv("SR_ASSESS_PAD") ? (size_t)atoi(std::getenv("SR_ASSESS_PAD")) : 0;   // experiments: fewer blocks per CU
#define LAUNCH_ASSESS(CNV, GS, US, SC)                                                                               \
    hipLaunchKernelGGL((k_assess_march<CNV, GS, US, SC>), grid, block, lds_pad, ctx->stream, d_a, (long long)stride_a, d_b,   \
                       (long long)stride_b, P, part)
#define LAUNCH_ASSESS_V(CNV)                                                                                            \
    do {                                                                                                                \
        if (gauss && unif) { if (P.same_c) LAUNCH_ASSESS(CNV, true, true, true); else LAUNCH_ASSESS(CNV, true, true, false); } \
        else if (gauss) { if (P.same_c) LAUNCH_ASSESS(CNV, true, false, true); else LAUNCH_ASSESS(CNV, true, false, false); } \
        else if (unif) LAUNCH_ASSESS(CNV, false, true, true);                                                           \
        else LAUNCH_ASSESS(CNV, false, false, true);                                                                    \
    } while (0)
            const bool gauss = (flags & (ASSESS_GAUSS | ASSESS_SIMPLE)) != 0, unif = (flags & ASSESS_UNIFORM) != 0;
            if (cn == 3) LAUNCH_ASSESS_V(3);
            else LAUNCH_ASSESS_V(1);
#undef LAUNCH_ASSESS_V
#undef LAUNCH_ASSESS
            hipLaunchKernelGGL(k_assess_finish, dim3(1), dim3(256), 0, ctx->stream, part, (long long)nblk, flags, d_out);
        }
        return check_launch("assess");
    }
    hipLaunchKernelGGL(k_assess_store, dim3(1), dim3(64), 0, ctx->stream, (const double *)nullptr, (const double *)nullptr, flags, d_out);
    return check_launch("assess");
}

int sr_assess_u8_async(sr_ctx *ctx, const uint8_t *d_a, int64_t stride_a, const uint8_t *d_b, int64_t stride_b, int h,
                       int w, int cn, int gray_shift, double data_range, int row_begin, int row_end, int flags,
                       sr_assess_sums *d_out)
{
    CTX_ENTER(ctx);
    return assess_impl(ctx, d_a, stride_a, d_b, stride_b, h, w, cn, gray_shift, data_range, row_begin, row_end, flags,
                       d_out, "assess_all");
}

int sr_assess_resized_u8_async(sr_ctx *ctx, const uint8_t *d_a, int64_t stride_a, const uint8_t *d_b, int64_t stride_b,
                               int h, int w, int cn, int dst_h, int dst_w, int gray_shift, double data_range, int flags,
                               sr_assess_sums *d_out)
{
    CTX_ENTER(ctx);
    if (h < 1 || w < 1 || dst_h < 1 || dst_w < 1)
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_assess_resized_u8: need positive source and destination sizes");
    if (!d_a || !d_b || !d_out) return sr_set_error(SR_ERR_INVALID_ARG, "sr_assess_resized_u8: null argument");
    if (cn != 1 && cn != 3) return sr_set_error(SR_ERR_INVALID_ARG, "sr_assess_resized_u8: need 1 or 3 channels");
    if (gray_shift != 14 && gray_shift != 15) return sr_set_error(SR_ERR_INVALID_ARG, "sr_assess_resized_u8: gray_shift must be 14 or 15");
    if (stride_a < (int64_t)w * cn || stride_b < (int64_t)w * cn) return sr_set_error(SR_ERR_SHAPE, "sr_assess_resized_u8: stride smaller than a row");
    // Stage 1: every resized pixel of both images is sampled ONCE by a plain map kernel (8 independent 12-byte loads per
    // pixel, full occupancy), its channel differences squared and summed, and only its two gray values kept: two u8 planes of
    // the resized size (the RGB intermediates of cv2.resize are never materialised).  Stage 2: the ordinary one-channel
    // assessment march over those planes.  (Sampling inside the march's loader -- round 1 -- kept the 88-register filter
    // FIFO alive across a rolled, latency-bound loop: 1.98 ms for the three scales of a 200 MP pair.)
    std::vector<CubicTab> xt, yt;
    cubic_table(w, dst_w, xt);
    cubic_table(h, dst_h, yt);
    xt.insert(xt.end(), yt.begin(), yt.end());
    HIPCHK(upload_cached(ctx, ctx->resize_tab, xt.data(), sizeof(CubicTab) * xt.size()));
    const CubicTab *d_xt = (const CubicTab *)ctx->resize_tab.d, *d_yt = d_xt + dst_w;
    const int64_t pitch = ((int64_t)dst_w + 63) / 64 * 64;
    // down-sampling RGB: the LDS-staged kernel when its source window (64 destination columns wide) fits
    int lds_pitch = 0, march_cols = 0, march_pitch = 0;
    if (cn == 3 && dst_w < w && dst_h < h) {
        auto window_pitch = [&](int cols) {
            int span = 0;
            for (int x0b = 0; x0b < dst_w; x0b += cols) {
                const int lo = std::max(xt[(size_t)x0b].ofs - 1, 0), hi = std::min(xt[(size_t)std::min(x0b + cols - 1, dst_w - 1)].ofs + 2, w - 1);
                span = std::max(span, (hi + 1) * 3 - ((lo * 3) & ~15));
            }
            return (span + 15) / 16 * 16 + 16;                                     // + one chunk: lds_tap12 reads 16 aligned bytes
        };
        const int lp = window_pitch(64);
        if (32 * lp <= 64 * 1024) lds_pitch = lp;
        for (int cols = 64; cols >= 32 && !march_cols; cols >>= 1) {
            const int mp = cols == 64 ? lp : window_pitch(cols);
            if (mp <= 64 * 16 + 16) { march_cols = cols; march_pitch = mp; }
        }
    }
    // since round 3 the wave-autonomous column march (k_resize_gray_pair_march) takes every down-sampling whose source window
    // (64 destination columns wide, or 32) is at most 64 chunks of 16 bytes, i.e. scales down to about 0.095
    // (SR_RESIZE_MARCH=0: the block-staged kernel)
    const bool march = march_cols > 0 && !(std::getenv("SR_RESIZE_MARCH") && std::getenv("SR_RESIZE_MARCH")[0] == '0');
    const dim3 block(64, 4), grid(march ? (unsigned)((dst_w + march_cols - 1) / march_cols)
                                 : lds_pitch ? (unsigned)((dst_w + 63) / 64) : (unsigned)((dst_w + 255) / 256),
                                 march ? (unsigned)((dst_h + 4 * RGM_SEG - 1) / (4 * RGM_SEG)) : (unsigned)((dst_h + 3) / 4));
    const size_t nblk = (size_t)grid.x * grid.y, plane = (size_t)pitch * dst_h;
    const size_t off_part = (2 * plane + 255) / 256 * 256, need = off_part + (nblk + 2 * (nblk / 1024 + 2)) * sizeof(double);
    if (need > ctx->gray_planes_bytes) {
        if (ctx->gray_planes) {
            HIPCHK(stream_sync(ctx));
            HIPCHK(hipFree(ctx->gray_planes));
            ctx->gray_planes = nullptr;
            ctx->gray_planes_bytes = 0;
        }
        HIPCHK(hipMalloc(&ctx->gray_planes, need));
        ctx->gray_planes_bytes = need;
    }
    uint8_t *ga = (uint8_t *)ctx->gray_planes, *gb = ga + plane;
    double *part = (double *)((char *)ctx->gray_planes + off_part), *buf0 = part + nblk, *buf1 = buf0 + nblk / 1024 + 2;
    const bool want_sse = (flags & ASSESS_SSE) != 0;
    const double *sse_ptr = nullptr;
    {
        ProfScope ps(ctx, "resize_gray");
        if (march) {
            hipLaunchKernelGGL(k_resize_gray_pair_march, grid, block, (size_t)32 * march_pitch, ctx->stream, d_a, (long long)stride_a, d_b,
                               (long long)stride_b, h, w, d_xt, d_yt, dst_h, dst_w, gray_shift, ga, gb, (long long)pitch, march_pitch,
                               march_cols, part);
        } else if (lds_pitch) {
            {   // once per device (the attribute belongs to the function ON a device), under a lock: contexts of several
                // devices / threads reach this concurrently
                static std::mutex mu;
                static std::set<int> done;
                std::lock_guard<std::mutex> lk(mu);
                if (!done.count(ctx->device)) {
                    HIPCHK(hipFuncSetAttribute((const void *)k_resize_gray_pair_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
                    done.insert(ctx->device);
                }
            }
            hipLaunchKernelGGL(k_resize_gray_pair_lds, grid, block, (size_t)32 * lds_pitch, ctx->stream, d_a, (long long)stride_a, d_b,
                               (long long)stride_b, h, w, d_xt, d_yt, dst_h, dst_w, gray_shift, ga, gb, (long long)pitch, lds_pitch, part);
        } else if (cn == 3) hipLaunchKernelGGL(k_resize_gray_pair<3>, grid, block, 0, ctx->stream, d_a, (long long)stride_a, d_b, (long long)stride_b, h, w, d_xt, d_yt, dst_h, dst_w, gray_shift, ga, gb, (long long)pitch, part);
        else hipLaunchKernelGGL(k_resize_gray_pair<1>, grid, block, 0, ctx->stream, d_a, (long long)stride_a, d_b, (long long)stride_b, h, w, d_xt, d_yt, dst_h, dst_w, gray_shift, ga, gb, (long long)pitch, part);
        if (want_sse) sse_ptr = reduce_partials(ctx, part, (long long)nblk, 1, buf0, buf1);
    }
    int rc = check_launch("resize_gray");
    if (rc) return rc;
    rc = assess_impl(ctx, ga, pitch, gb, pitch, dst_h, dst_w, 1, gray_shift, data_range, 0, dst_h, flags & ~ASSESS_SSE, d_out, "assess_resized");
    if (rc) return rc;
    if (want_sse) {
        hipLaunchKernelGGL(k_store_sse, dim3(1), dim3(64), 0, ctx->stream, sse_ptr, d_out);
        return check_launch("assess_resized sse");
    }
    return SR_OK;
}

int sr_assess_resized_u8(sr_ctx *ctx, const uint8_t *d_a, int64_t stride_a, const uint8_t *d_b, int64_t stride_b, int h,
                         int w, int cn, int dst_h, int dst_w, int gray_shift, double data_range, int flags,
                         sr_assess_sums *h_out)
{
    CTX_ENTER(ctx);
    if (!h_out) return sr_set_error(SR_ERR_INVALID_ARG, "sr_assess_resized_u8: null result");
    void *res = nullptr;
    HIPCHK(hipMalloc(&res, sizeof(sr_assess_sums)));
    int rc = sr_assess_resized_u8_async(ctx, d_a, stride_a, d_b, stride_b, h, w, cn, dst_h, dst_w, gray_shift,
                                        data_range, flags, (sr_assess_sums *)res);
    if (rc == SR_OK) {
        hipError_t e = hipMemcpyAsync(h_out, res, sizeof(sr_assess_sums), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = stream_sync(ctx);
        if (e != hipSuccess) rc = sr_set_error(SR_ERR_HIP, "sr_assess_resized_u8: %s", hipGetErrorString(e));
    }
    (void)hipFree(res);
    return rc;
}

int sr_assess_u8(sr_ctx *ctx, const uint8_t *d_a, int64_t stride_a, const uint8_t *d_b, int64_t stride_b, int h, int w,
                 int cn, int gray_shift, double data_range, int row_begin, int row_end, int flags, sr_assess_sums *h_out)
{
    CTX_ENTER(ctx);
    if (!h_out) return sr_set_error(SR_ERR_INVALID_ARG, "sr_assess_u8: null result");
    void *res = nullptr;
    HIPCHK(hipMalloc(&res, sizeof(sr_assess_sums)));
    int rc = sr_assess_u8_async(ctx, d_a, stride_a, d_b, stride_b, h, w, cn, gray_shift, data_range, row_begin, row_end,
                                flags, (sr_assess_sums *)res);
    if (rc == SR_OK) {
        hipError_t e = hipMemcpyAsync(h_out, res, sizeof(sr_assess_sums), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = stream_sync(ctx);
        if (e != hipSuccess) rc = sr_set_error(SR_ERR_HIP, "sr_assess_u8: D2H: %s", hipGetErrorString(e));
    }
    (void)stream_sync(ctx);
    (void)hipFree(res);
    return rc;
}

static int ssim_mode_check(int h, int w, int mode, int *flag, size_t *field_off)
{
    int pad;
    if (mode == SR_SSIM_UNIFORM7) { pad = 3; *flag = ASSESS_UNIFORM; *field_off = offsetof(sr_assess_sums, ssim_uniform); }
    else if (mode == SR_SSIM_GAUSS11) { pad = 5; *flag = ASSESS_GAUSS; *field_off = offsetof(sr_assess_sums, ssim_gauss); }
    else if (mode == SR_SSIM_SIMPLE) { pad = 0; *flag = ASSESS_SIMPLE; *field_off = offsetof(sr_assess_sums, ssim_simple); }
    else return sr_set_error(SR_ERR_INVALID_ARG, "sr_ssim_u8: unknown mode %d", mode);
    if (h <= 2 * pad || w <= 2 * pad)
        return sr_set_error(SR_ERR_SHAPE, "sr_ssim_u8: image %dx%d smaller than the %d-tap window", w, h, 2 * pad + 1);
    return SR_OK;
}

int sr_ssim_u8_async(sr_ctx *ctx, const uint8_t *d_a, int64_t stride_a, const uint8_t *d_b, int64_t stride_b, int h,
                     int w, int cn, int mode, int gray_shift, double data_range, int row_begin, int row_end,
                     double *d_sum, uint64_t *h_count)
{
    CTX_ENTER(ctx);
    if (!d_a || !d_b || !d_sum || !h_count) return sr_set_error(SR_ERR_INVALID_ARG, "sr_ssim_u8: null argument");
    int flag = 0;
    size_t off = 0;
    int rc = ssim_mode_check(h, w, mode, &flag, &off);
    if (rc) return rc;
    rc = sr_ssim_count(h, w, mode, row_begin, row_end, h_count);
    if (rc) return rc;
    void *scr = nullptr;
    rc = ctx_scratch(ctx, (size_t)8 << 20, &scr);
    if (rc) return rc;
    // the result record lives in the last 256 bytes of the (>= 8 MiB) scratch, clear of the partial buffers
    sr_assess_sums *rec = (sr_assess_sums *)((char *)ctx->scratch + ctx->scratch_bytes - 256);
    rc = sr_assess_u8_async(ctx, d_a, stride_a, d_b, stride_b, h, w, cn, gray_shift, data_range, row_begin, row_end, flag, rec);
    if (rc) return rc;
    rec = (sr_assess_sums *)((char *)ctx->scratch + ctx->scratch_bytes - 256);    // scratch may have grown
    HIPCHK(hipMemcpyAsync(d_sum, (const char *)rec + off, sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return SR_OK;
}

int sr_ssim_u8(sr_ctx *ctx, const uint8_t *d_a, int64_t stride_a, const uint8_t *d_b, int64_t stride_b, int h, int w,
               int cn, int mode, int gray_shift, double data_range, int row_begin, int row_end, double *h_sum,
               uint64_t *h_count)
{
    CTX_ENTER(ctx);
    if (!h_sum || !h_count) return sr_set_error(SR_ERR_INVALID_ARG, "sr_ssim_u8: null result");
    int flag = 0;
    size_t off = 0;
    int rc = ssim_mode_check(h, w, mode, &flag, &off);
    if (rc) return rc;
    rc = sr_ssim_count(h, w, mode, row_begin, row_end, h_count);
    if (rc) return rc;
    sr_assess_sums sums;
    rc = sr_assess_u8(ctx, d_a, stride_a, d_b, stride_b, h, w, cn, gray_shift, data_range, row_begin, row_end, flag, &sums);
    if (rc) return rc;
    *h_sum = *(const double *)((const char *)&sums + off);
    return SR_OK;
}

int sr_ssim_float(sr_ctx *ctx, int dtype, const void *d_a, int64_t stride_a, const void *d_b, int64_t stride_b, int h, int w,
                  int cn, int mode, double data_range, int row_begin, int row_end, double *h_sum, uint64_t *h_count)
{
    CTX_ENTER(ctx);
    if (!d_a || !d_b || !h_sum || !h_count) return sr_set_error(SR_ERR_INVALID_ARG, "sr_ssim_float: null argument");
    if (dtype != SR_F32 && dtype != SR_F64) return sr_set_error(SR_ERR_INVALID_ARG, "sr_ssim_float: dtype must be SR_F32 or SR_F64");
    if (cn != 1 && !(cn == 3 && dtype == SR_F32))
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_ssim_float: 1 channel, or 3 channels of float32 (cv2.cvtColor rejects float64 RGB)");
    int flag = 0;
    size_t off = 0;
    int rc = ssim_mode_check(h, w, mode, &flag, &off);
    if (rc) return rc;
    const int es = dtype == SR_F32 ? 4 : 8;
    if (stride_a < (int64_t)w * cn * es || stride_b < (int64_t)w * cn * es) return sr_set_error(SR_ERR_SHAPE, "sr_ssim_float: stride smaller than a row");
    row_begin = std::max(row_begin, 0);
    row_end = std::min(row_end, h);
    rc = sr_ssim_count(h, w, mode, row_begin, row_end, h_count);
    if (rc) return rc;
    SsimFParams P;
    memset(&P, 0, sizeof(P));
    P.h = h; P.w = w; P.mode = mode; P.row_begin = row_begin; P.row_end = row_end;
    P.c1 = (0.01 * data_range) * (0.01 * data_range);
    P.c2 = (0.03 * data_range) * (0.03 * data_range);
    P.cov_norm = 1.0;
    if (mode == SR_SSIM_UNIFORM7) {
        P.radius = 3; P.bmode = PAD_REFLECT; P.crop = 3; P.cov_norm = 49.0 / 48.0;
        for (int j = 0; j < 7; ++j) P.k[j] = 1.0 / 7.0;
    } else {
        double k6[6];
        gauss_taps(k6);                                  // scipy's and cv2's normalised 11-tap kernels coincide
        P.radius = 5;
        for (int j = 0; j <= 5; ++j) P.k[5 + j] = P.k[5 - j] = k6[j];
        if (mode == SR_SSIM_GAUSS11) { P.bmode = PAD_REFLECT; P.crop = 5; }
        else { P.bmode = PAD_MIRROR; P.crop = 0; P.c1 = (0.01 * 255.0) * (0.01 * 255.0); P.c2 = (0.03 * 255.0) * (0.03 * 255.0); }
    }
    const size_t plane = (size_t)h * w;
    const dim3 block(64, 4), grid((unsigned)((w + 63) / 64), (unsigned)((h + 3) / 4));
    const size_t nblk = (size_t)grid.x * grid.y;
    double *buf = nullptr;
    {
        hipError_t e = hipMalloc((void **)&buf, (7 * plane + nblk + 2 * (nblk / 1024 + 2)) * sizeof(double));
        if (e != hipSuccess) return sr_set_error(e == hipErrorOutOfMemory ? SR_ERR_OOM : SR_ERR_HIP, "sr_ssim_float: %s", hipGetErrorString(e));
    }
    double *ga = buf, *gb = buf + plane, *tmp = buf + 2 * plane, *part = buf + 7 * plane, *b0 = part + nblk, *b1 = b0 + nblk / 1024 + 2;
    {
        ProfScope ps(ctx, "ssim_float");
        if (dtype == SR_F32) {
            hipLaunchKernelGGL(k_ssimf_gray<float>, grid, block, 0, ctx->stream, (const float *)d_a, (long long)stride_a, h, w, cn, ga);
            hipLaunchKernelGGL(k_ssimf_gray<float>, grid, block, 0, ctx->stream, (const float *)d_b, (long long)stride_b, h, w, cn, gb);
        } else {
            hipLaunchKernelGGL(k_ssimf_gray<double>, grid, block, 0, ctx->stream, (const double *)d_a, (long long)stride_a, h, w, cn, ga);
            hipLaunchKernelGGL(k_ssimf_gray<double>, grid, block, 0, ctx->stream, (const double *)d_b, (long long)stride_b, h, w, cn, gb);
        }
        hipLaunchKernelGGL(k_ssimf_rows, grid, block, 0, ctx->stream, (const double *)ga, (const double *)gb, P, tmp);
        hipLaunchKernelGGL(k_ssimf_cols, grid, block, 0, ctx->stream, (const double *)tmp, P, part);
    }
    const double *res = reduce_partials(ctx, part, (long long)nblk, 1, b0, b1);
    rc = check_launch("ssim_float");
    hipError_t e = hipMemcpyAsync(h_sum, res, sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    hipError_t es2 = stream_sync(ctx);
    (void)hipFree(buf);
    if (rc) return rc;
    if (e != hipSuccess || es2 != hipSuccess) return sr_set_error(SR_ERR_HIP, "sr_ssim_float: %s", hipGetErrorString(e != hipSuccess ? e : es2));
    return SR_OK;
}

int sr_rgb2gray_u8(sr_ctx *ctx, const uint8_t *d_rgb, int64_t stride, int h, int w, int gray_shift, uint8_t *d_gray,
                   int64_t gray_stride)
{
    CTX_ENTER(ctx);
    if (!d_rgb || !d_gray || h < 1 || w < 1 || (gray_shift != 14 && gray_shift != 15))
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_rgb2gray_u8: bad arguments");
    {
        ProfScope ps(ctx, "rgb2gray");
        dim3 grid((w + 63) / 64, (h + 3) / 4), block(64, 4);
        hipLaunchKernelGGL(k_rgb2gray, grid, block, 0, ctx->stream, d_rgb, (long long)stride, h, w, gray_shift, d_gray,
                           (long long)gray_stride);
    }
    return check_launch("rgb2gray");
}

int sr_resize_cubic_window_u8(sr_ctx *ctx, const uint8_t *d_src, int64_t src_stride, int h, int w, int cn, int dh,
                              int dw, int x0, int y0, int ww, int wh, uint8_t *d_dst, int64_t dst_stride)
{
    CTX_ENTER(ctx);
    if (!d_src || !d_dst || h < 1 || w < 1 || dh < 1 || dw < 1 || cn < 1 || cn > 4)
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_resize_cubic_u8: bad arguments");
    if (x0 < 0 || y0 < 0 || ww < 1 || wh < 1 || x0 + ww > dw || y0 + wh > dh)
        return sr_set_error(SR_ERR_SHAPE, "sr_resize_cubic_u8: window outside the %dx%d result", dw, dh);
    std::vector<CubicTab> xt, yt;
    cubic_table(w, dw, xt);
    cubic_table(h, dh, yt);
    xt.insert(xt.end(), yt.begin(), yt.end());          // both axes in one cached device table (re-used per geometry)
    HIPCHK(upload_cached(ctx, ctx->cubic_tab, xt.data(), sizeof(CubicTab) * xt.size()));
    CubicTab *dx = (CubicTab *)ctx->cubic_tab.d, *dy = dx + dw;
    {
        ProfScope ps(ctx, "resize_cubic");
        dim3 grid((ww + 63) / 64, (wh + 3) / 4), block(64, 4);
        if (cn == 3 && dh >= h) {                        // rows are reused: the marching kernel
            dim3 gridu((ww + 1023) / 1024, (wh + RUP_SEG - 1) / RUP_SEG);
            hipLaunchKernelGGL(k_resize_cubic_up_rgb, gridu, dim3(256), 0, ctx->stream, d_src, (long long)src_stride, h, w,
                               (const CubicTab *)dx, (const CubicTab *)dy, x0, y0, ww, wh, d_dst, (long long)dst_stride);
        } else if (cn == 3) {
            dim3 grid4((ww + 255) / 256, (wh + 3) / 4);
            hipLaunchKernelGGL(k_resize_cubic_rgb4, grid4, block, 0, ctx->stream, d_src, (long long)src_stride, h, w,
                               (const CubicTab *)dx, (const CubicTab *)dy, x0, y0, ww, wh, d_dst, (long long)dst_stride);
        } else
        hipLaunchKernelGGL(k_resize_cubic, grid, block, 0, ctx->stream, d_src, (long long)src_stride, h, w, cn,
                           (const CubicTab *)dx, (const CubicTab *)dy, x0, y0, ww, wh, d_dst, (long long)dst_stride);
    }
    return check_launch("resize_cubic");
}

int sr_resize_cubic_u8(sr_ctx *ctx, const uint8_t *d_src, int64_t src_stride, int h, int w, int cn, uint8_t *d_dst,
                       int64_t dst_stride, int dh, int dw)
{
    return sr_resize_cubic_window_u8(ctx, d_src, src_stride, h, w, cn, dh, dw, 0, 0, dw, dh, d_dst, dst_stride);
}

}  // extern "C"

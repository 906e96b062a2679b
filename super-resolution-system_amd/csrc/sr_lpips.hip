// sr_lpips.hip -- LPIPS (quality_assessment_module.py:419-465 calculate_lpips, :197-224 _to_lpips_tensor) on gfx950.
//
// The reference delegates to the third-party package `lpips` (>= 0.1.4, requirements.txt:16): AlexNet / VGG16 feature
// stacks, unit-normalised channel vectors, squared difference, learned 1x1 `lin` weights, spatial mean, sum over five
// taps.  This is the one dense contraction of the tile -> blend -> assess path, so -- unlike everything in
// sr_engine.hip -- it runs on the matrix cores: every convolution with Cin >= 64 is an implicit GEMM on
// v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate: the reference's arithmetic is torch fp32, there is no reduced
// precision anywhere).  The two 3-channel stem convolutions are a direct VALU kernel that also applies the u8 -> [-1, 1]
// -> ScalingLayer preprocessing, so the fp32 input tensor is never materialised.
//
// Memory: activations are planar fp32 [C][rows][pitch].  A 200 MP image does not fit (relu1_1 alone is 51 GB per
// image), so the image is streamed in square tiles of the INPUT; each tile recomputes the receptive-field halo its
// outputs need (90 px for VGG16), and zero padding is applied per layer at the true image border only, so every
// feature value equals the untiled forward's.  Per tap layer a tile contributes the sum over its own (disjoint)
// part of the feature map; the sums are fp64 and deterministic (fixed block order).
//
// Weights are caller-supplied (sr_lpips_create); nothing is fetched.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <set>
#include <vector>

#include "sr_ctx.h"

namespace {

enum { LP_CONV = 0, LP_POOL = 1, LP_TAP = 2 };

struct LpLayer {
    int kind;
    int cout, cin, k, s, p;   // conv / pool geometry
    int tap;                  // LP_TAP: LPIPS layer index 0..4
    int widx;                 // LP_CONV: index into the weight arrays
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------
// Stem convolution (Cin = 3) straight from the u8 image: one thread = one output pixel x 64 output channels.
// value(c, y, x) = ((u8 / 255) * 2 - 1 - shift_c) / scale_c inside the image, 0 outside (Conv2d zero padding acts on
// the ScalingLayer output).  The 3 x 256 possible values are tabulated in LDS once per block with exactly those fp32
// operations.  Weights are laid out [c][tap][64] so the 64 multipliers of one input value are wave-uniform
// (scalar loads), the inner loop is 64 v_fmac with an SGPR operand.
// ---------------------------------------------------------------------------------------------------------------
template <int KS, int S, int P>
__global__ __launch_bounds__(256) void k_lp_conv_stem(const unsigned char *__restrict__ img, long long stride, int cn,
                                                      int H, int W, const float *__restrict__ wt,
                                                      const float *__restrict__ bias, float sh0, float sh1, float sh2,
                                                      float sc0, float sc1, float sc2, float *__restrict__ out, int ya,
                                                      int xa, int rows, int cols, int pitch, long long plane)
{
    __shared__ float lut[3][256];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    for (int i = tid; i < 768; i += 256) {
        const int c = i >> 8, v = i & 255;
        const float t = (float)v / 255.0f;
        const float u = t * 2.0f - 1.0f;
        const float sh = c == 0 ? sh0 : (c == 1 ? sh1 : sh2), sc = c == 0 ? sc0 : (c == 1 ? sc1 : sc2);
        lut[c][v] = (u - sh) / sc;
    }
    __syncthreads();
    const int lx = blockIdx.x * 64 + threadIdx.x, ly = blockIdx.y * 4 + threadIdx.y;
    if (lx >= cols || ly >= rows) return;
    const int oy = ya + ly, ox = xa + lx;
    float acc[64];
#pragma unroll
    for (int co = 0; co < 64; ++co) acc[co] = bias[co];
    constexpr int T = KS * KS;
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        const int cs = cn == 1 ? 0 : c;              // gray is repeated, alpha (cn == 4) is dropped
#pragma unroll 1
        for (int ky = 0; ky < KS; ++ky) {
            const int gy = oy * S - P + ky;
            const bool yok = gy >= 0 && gy < H;
            const unsigned char *row = img + (size_t)(yok ? gy : 0) * stride;
#pragma unroll 1
            for (int kx = 0; kx < KS; ++kx) {
                const int gx = ox * S - P + kx;
                float v = 0.0f;
                if (yok && gx >= 0 && gx < W) v = lut[c][row[(size_t)gx * cn + cs]];
                const float *wp = wt + ((size_t)c * T + ky * KS + kx) * 64;
#pragma unroll
                for (int co = 0; co < 64; ++co) acc[co] = fmaf(wp[co], v, acc[co]);
            }
        }
    }
    float *o = out + (size_t)ly * pitch + lx;
#pragma unroll
    for (int co = 0; co < 64; ++co) o[(size_t)co * plane] = fmaxf(acc[co], 0.0f);
}

// ---------------------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution + bias + ReLU on v_mfma_f32_32x32x2_f32, stride 1, KS x KS, zero padding P = KS / 2.
//   GEMM view: D[cout][pixel] = sum_k W[cout][k] * X[k][pixel], k = (cin, tap).
//   A operand = weights (lane l: cout l & 31, k-half l >> 5), B operand = input pixels (lane l: pixel l & 31, k-half
//   l >> 5); the accumulator then holds pixel l & 31 on the lane and 16 couts in its registers, so every store
//   instruction writes two contiguous 128-byte row segments of two output planes.
//   Block = 4 waves = 8 output rows x 32 columns x 64 couts; wave w owns rows 2w, 2w + 1 and both 32-cout halves
//   (four 32 x 32 accumulators, 64 VGPRs).  Cin is walked in chunks of CC channels: the input patch
//   [CC][8 + KS - 1][32 + KS - 1] and the weight slab [CC][KS*KS][64] are staged in LDS, the k-half of a lane selects
//   the channel parity, so the per-step LDS addresses are lane base + compile-time immediates.
//   Per k-step a wave issues 2 + 2 LDS dword reads for 4 MFMAs (256 matrix-pipe cycles): the kernel is matrix-pipe
//   bound by a wide margin; global loads of the next chunk are in flight during the MFMAs of the current one.
// The input is a planar buffer covering rows [in_ya, ..) x cols [in_xa, ..) of the layer's global index space; taps
// outside the image extent (H_in, W_in) read as zero -- that is the layer's own zero padding, also in the interior
// of a tiled forward where the buffer holds real neighbour data instead.
// ---------------------------------------------------------------------------------------------------------------
template <int KS, int CC>
__global__ __launch_bounds__(256) void k_lp_conv_mfma(const float *__restrict__ in, long long in_plane, int in_pitch,
                                                      int in_ya, int in_xa, int H_in, int W_in, int cin,
                                                      const float *__restrict__ wslab, const float *__restrict__ bias,
                                                      float *__restrict__ out, long long out_plane, int out_pitch,
                                                      int out_ya, int out_xa, int rows, int cols)
{
    constexpr int T = KS * KS, P = KS / 2;
    constexpr int PH = 8 + KS - 1, PW = 32 + KS - 1;
    constexpr int NPATCH = CC * PH * PW, NW4 = CC * T * 16;            // patch floats, weight float4s per chunk
    constexpr int PE = (NPATCH + 255) / 256, WE = (NW4 + 255) / 256;   // per-thread staging counts
    __shared__ __attribute__((aligned(16))) float s_patch[NPATCH];
    __shared__ __attribute__((aligned(16))) float s_w[CC * T * 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, l32 = lane & 31, half = lane >> 5;
    const int ox0 = blockIdx.x * 32, oy0 = blockIdx.y * 8;              // tile origin inside the output range
    const int ct = blockIdx.z;                                          // 64-cout tile
    const int nchunk = cin / CC;

    // staging map of this thread: patch element e -> (channel, row, col) is the same for every chunk
    int p_off[PE];
    unsigned p_ok = 0;
#pragma unroll
    for (int i = 0; i < PE; ++i) {
        const int e = tid + i * 256;
        const int c = e / (PH * PW), r = (e / PW) % PH, x = e % PW;
        const int gy = out_ya + oy0 - P + r, gx = out_xa + ox0 - P + x;   // global index in the input layer
        const bool ok = e < NPATCH && gy >= 0 && gy < H_in && gx >= 0 && gx < W_in;
        p_off[i] = ok ? (int)((long long)c * in_plane + (long long)(gy - in_ya) * in_pitch + (gx - in_xa)) : 0;
        if (ok) p_ok |= 1u << i;
    }
    const f4v *wsrc = (const f4v *)(wslab + (size_t)ct * nchunk * (CC * T * 64));

    f32x16 acc[2][2];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float b = bias[ct * 64 + c2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
            acc[c2][0][r] = b;
            acc[c2][1][r] = b;
        }

    float pv[PE];
    f4v wv[WE];
    auto load_chunk = [&](int ch) {
        const float *ib = in + (size_t)ch * CC * in_plane;
#pragma unroll
        for (int i = 0; i < PE; ++i) pv[i] = (p_ok >> i) & 1u ? ib[p_off[i]] : 0.0f;
        const f4v *wb = wsrc + (size_t)ch * NW4;
#pragma unroll
        for (int i = 0; i < WE; ++i) {
            const int e = tid + i * 256;
            wv[i] = e < NW4 ? wb[e] : f4v{0.f, 0.f, 0.f, 0.f};
        }
    };
    load_chunk(0);
    // lane bases: the k-half selects the odd channel of a pair
    const float *a_base = s_w + half * (T * 64) + l32;
    const float *b_base = s_patch + half * (PH * PW) + (2 * wave) * PW + l32;
#pragma unroll 1
    for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();                                   // the previous chunk has been consumed
#pragma unroll
        for (int i = 0; i < PE; ++i) {
            const int e = tid + i * 256;
            if (e < NPATCH) s_patch[e] = pv[i];
        }
#pragma unroll
        for (int i = 0; i < WE; ++i) {
            const int e = tid + i * 256;
            if (e < NW4) ((f4v *)s_w)[e] = wv[i];
        }
        __syncthreads();
        if (ch + 1 < nchunk) load_chunk(ch + 1);           // in flight under the MFMAs below
#pragma unroll
        for (int cp = 0; cp < CC / 2; ++cp)
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int dy = t / KS, dx = t % KS;
                const float a0 = a_base[(2 * cp * T + t) * 64], a1 = a_base[(2 * cp * T + t) * 64 + 32];
                const float b0 = b_base[2 * cp * PH * PW + dy * PW + dx], b1 = b_base[2 * cp * PH * PW + (dy + 1) * PW + dx];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
    }
    // epilogue: ReLU, masked store
    const int col = ox0 + l32;
    if (col >= cols) return;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        const int row = oy0 + 2 * wave + pr;
        if (row >= rows) continue;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = ct * 64 + c2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                out[(size_t)co * out_plane + (size_t)row * out_pitch + col] = fmaxf(acc[c2][pr][r], 0.0f);
            }
    }
}

// MaxPool2d(K, S), no padding, floor mode: every window lies inside the input range by construction.
template <int K, int S>
__global__ __launch_bounds__(256) void k_lp_pool(const float *__restrict__ in, long long in_plane, int in_pitch, int in_ya,
                                                 int in_xa, float *__restrict__ out, long long out_plane, int out_pitch,
                                                 int out_ya, int out_xa, int rows, int cols)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, c = blockIdx.z;
    if (x >= cols || y >= rows) return;
    const float *p = in + (size_t)c * in_plane + (size_t)((out_ya + y) * S - in_ya) * in_pitch + ((out_xa + x) * S - in_xa);
    float m = p[0];
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
        for (int i = 0; i < K; ++i) m = fmaxf(m, p[(size_t)j * in_pitch + i]);
    out[(size_t)c * out_plane + (size_t)y * out_pitch + x] = m;
}

__device__ __forceinline__ double lp_wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// One LPIPS tap over the tile's own part of the feature map: per pixel
//   sum_c lin_c * (a_c / (|a| + 1e-10) - b_c / (|b| + 1e-10))^2        (normalize_tensor, lpips.py)
// in fp32 like the package, summed over pixels in fp64 -> one partial per block.
__global__ __launch_bounds__(256) void k_lp_tap(const float *__restrict__ fa, const float *__restrict__ fb, long long plane,
                                                int pitch, int C, const float *__restrict__ lin, int y0, int x0, int rows,
                                                int cols, double *__restrict__ part)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    double v = 0.0;
    if (x < cols && y < rows) {
        const size_t o = (size_t)(y0 + y) * pitch + (x0 + x);
        float sa = 0.0f, sb = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float a = fa[(size_t)c * plane + o], b = fb[(size_t)c * plane + o];
            sa += a * a;
            sb += b * b;
        }
        const float na = sqrtf(sa) + 1e-10f, nb = sqrtf(sb) + 1e-10f;
        float s = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float a = fa[(size_t)c * plane + o] / na, b = fb[(size_t)c * plane + o] / nb;
            const float d = a - b;
            s += lin[c] * (d * d);
        }
        v = (double)s;
    }
    v = lp_wave_sum(v);
    __shared__ double ws[4];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    if ((tid & 63) == 0) ws[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = ((ws[0] + ws[1]) + ws[2]) + ws[3];
}

// acc += sum(part[0..n)) in a fixed order (one block)
__global__ __launch_bounds__(256) void k_lp_accum(const double *__restrict__ part, long long n, double *__restrict__ acc)
{
    __shared__ double sh[256];
    double s = 0.0;
    for (long long i = threadIdx.x; i < n; i += 256) s += part[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *acc += sh[0];
}

struct Range {
    int a = 0, b = 0;                 // [a, b)
    bool empty() const { return a >= b; }
    int n() const { return b > a ? b - a : 0; }
};

Range hull(Range p, Range q)
{
    if (p.empty()) return q;
    if (q.empty()) return p;
    Range r;
    r.a = std::min(p.a, q.a);
    r.b = std::max(p.b, q.b);
    return r;
}

}  // namespace

struct sr_lpips_model {
    sr_ctx *ctx = nullptr;
    int net = 0;
    std::vector<LpLayer> layers;
    std::vector<float *> d_w, d_b;    // per convolution: weights in the kernel's layout, bias
    float *d_lin[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int tap_c[5] = {0, 0, 0, 0, 0};
    float shift[3], scale[3];
    float *buf[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [image][ping-pong] activation buffers
    size_t buf_floats = 0;
    double *d_acc = nullptr;          // 5 layer sums
    double *d_part = nullptr;         // per-block partials of one tap launch
    size_t part_cap = 0;
};

static std::mutex g_lp_mu;
static std::set<const void *> g_lp_live;

static bool lp_is_live(const sr_lpips_model *m)
{
    std::lock_guard<std::mutex> lk(g_lp_mu);
    return m && g_lp_live.count(m) != 0;
}

static void lp_arch(int net, std::vector<LpLayer> &L)
{
    auto conv = [&](int cout, int cin, int k, int s, int p) {
        int widx = 0;
        for (auto &l : L) widx += l.kind == LP_CONV;
        L.push_back({LP_CONV, cout, cin, k, s, p, -1, widx});
    };
    auto pool = [&](int k, int s) { L.push_back({LP_POOL, 0, 0, k, s, 0, -1, -1}); };
    auto tap = [&](int i) { L.push_back({LP_TAP, 0, 0, 1, 1, 0, i, -1}); };
    if (net == SR_LPIPS_ALEX) {          // torchvision AlexNet.features, lpips/pretrained_networks.py alexnet
        conv(64, 3, 11, 4, 2); tap(0);
        pool(3, 2); conv(192, 64, 5, 1, 2); tap(1);
        pool(3, 2); conv(384, 192, 3, 1, 1); tap(2);
        conv(256, 384, 3, 1, 1); tap(3);
        conv(256, 256, 3, 1, 1); tap(4);
    } else {                             // torchvision VGG16.features, taps relu1_2 .. relu5_3
        conv(64, 3, 3, 1, 1); conv(64, 64, 3, 1, 1); tap(0);
        pool(2, 2); conv(128, 64, 3, 1, 1); conv(128, 128, 3, 1, 1); tap(1);
        pool(2, 2); conv(256, 128, 3, 1, 1); conv(256, 256, 3, 1, 1); conv(256, 256, 3, 1, 1); tap(2);
        pool(2, 2); conv(512, 256, 3, 1, 1); conv(512, 512, 3, 1, 1); conv(512, 512, 3, 1, 1); tap(3);
        pool(2, 2); conv(512, 512, 3, 1, 1); conv(512, 512, 3, 1, 1); conv(512, 512, 3, 1, 1); tap(4);
    }
}

// Per-activation geometry of an h x w input: activation 0 is the image, activation j the output of the j-th conv / pool.
struct LpGeom {
    std::vector<int> H, W, C, S;      // extent, channels and cumulative stride per activation
    std::vector<int> act_of_layer;    // activation a layer reads (conv / pool) or taps
};

static bool lp_geometry(const std::vector<LpLayer> &L, int h, int w, LpGeom &G)
{
    G.H = {h}; G.W = {w}; G.C = {3}; G.S = {1};
    G.act_of_layer.clear();
    for (auto &l : L) {
        G.act_of_layer.push_back((int)G.H.size() - 1);
        if (l.kind == LP_TAP) continue;
        const int hi = G.H.back(), wi = G.W.back();
        const int ho = (hi + 2 * l.p - l.k) / l.s + 1, wo = (wi + 2 * l.p - l.k) / l.s + 1;
        if (hi + 2 * l.p < l.k || wi + 2 * l.p < l.k || ho < 1 || wo < 1) return false;
        G.H.push_back(ho); G.W.push_back(wo);
        G.C.push_back(l.kind == LP_CONV ? l.cout : G.C.back());
        G.S.push_back(G.S.back() * l.s);
    }
    return true;
}

// Ranges (one axis) every activation of a tile must cover: its own part of every tapped activation plus what the
// deeper layers need of it.  t0 / t1: the tile's span in input pixels (t1 < 0: to the end).
static void lp_ranges(const std::vector<LpLayer> &L, const LpGeom &G, const std::vector<int> &ext, int t0, int t1,
                      std::vector<Range> &need, std::vector<Range> &own)
{
    const int na = (int)ext.size();
    need.assign(na, Range());
    own.assign(na, Range());
    for (int j = 0; j < na; ++j) {
        own[j].a = std::min(t0 / G.S[j], ext[j]);
        own[j].b = t1 < 0 ? ext[j] : std::min(t1 / G.S[j], ext[j]);
    }
    for (int li = (int)L.size() - 1; li >= 0; --li) {
        const LpLayer &l = L[li];
        const int ai = G.act_of_layer[li];
        if (l.kind == LP_TAP) {
            need[ai] = hull(need[ai], own[ai]);
            continue;
        }
        const Range o = need[ai + 1];
        if (o.empty()) continue;
        Range r;
        r.a = std::max(o.a * l.s - l.p, 0);
        r.b = std::min((o.b - 1) * l.s - l.p + l.k, ext[ai]);
        need[ai] = hull(need[ai], r);
    }
}

extern "C" {

int sr_lpips_create(sr_ctx *ctx, int net, const float *const *h_conv_w, const float *const *h_conv_b, int n_conv,
                    const float *const *h_lin_w, int n_lin, const float *h_shift, const float *h_scale,
                    sr_lpips_model **out)
{
    CTX_ENTER(ctx);
    if (!out) return sr_set_error(SR_ERR_INVALID_ARG, "sr_lpips_create: null out");
    *out = nullptr;
    if (net != SR_LPIPS_ALEX && net != SR_LPIPS_VGG) return sr_set_error(SR_ERR_INVALID_ARG, "sr_lpips_create: net must be SR_LPIPS_ALEX or SR_LPIPS_VGG");
    sr_lpips_model *M = new sr_lpips_model();
    M->ctx = ctx;
    M->net = net;
    lp_arch(net, M->layers);
    int want_conv = 0;
    for (auto &l : M->layers) want_conv += l.kind == LP_CONV;
    if (!h_conv_w || !h_conv_b || !h_lin_w || n_conv != want_conv || n_lin != 5) {
        delete M;
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_lpips_create: net %d needs %d convolutions and 5 lin layers", net, want_conv);
    }
    const float def_shift[3] = {-0.030f, -0.088f, -0.188f}, def_scale[3] = {0.458f, 0.448f, 0.450f};
    for (int c = 0; c < 3; ++c) {
        M->shift[c] = h_shift ? h_shift[c] : def_shift[c];
        M->scale[c] = h_scale ? h_scale[c] : def_scale[c];
    }
    {
        std::lock_guard<std::mutex> lk(g_lp_mu);
        g_lp_live.insert(M);
    }
    auto fail = [&](int code, const char *what) {
        sr_set_error(code, "sr_lpips_create: %s", what);
        sr_lpips_destroy(M);
        return code;
    };
    int tapc = 3, ti = 0;
    for (auto &l : M->layers) {
        if (l.kind == LP_CONV) tapc = l.cout;
        if (l.kind == LP_TAP) M->tap_c[ti++] = tapc;
    }
    for (auto &l : M->layers) {
        if (l.kind != LP_CONV) continue;
        const float *w = h_conv_w[l.widx], *b = h_conv_b[l.widx];
        if (!w || !b) return fail(SR_ERR_INVALID_ARG, "null weight array");
        const int T = l.k * l.k;
        std::vector<float> arranged((size_t)l.cout * l.cin * T);
        if (l.cin == 3) {                                   // stem: [c][tap][cout]
            if (l.cout != 64) return fail(SR_ERR_UNSUPPORTED, "stem convolution must have 64 outputs");
            for (int co = 0; co < l.cout; ++co)
                for (int c = 0; c < 3; ++c)
                    for (int t = 0; t < T; ++t) arranged[((size_t)c * T + t) * 64 + co] = w[((size_t)co * 3 + c) * T + t];
        } else {                                            // MFMA: [cout tile][chunk][c in chunk][tap][64]
            const int CC = l.k == 3 ? 8 : 4;
            if (l.cout % 64 || l.cin % CC || l.s != 1 || (l.k != 3 && l.k != 5) || l.p != l.k / 2)
                return fail(SR_ERR_UNSUPPORTED, "convolution shape outside the MFMA kernel's cases");
            const int nch = l.cin / CC;
            for (int ct = 0; ct < l.cout / 64; ++ct)
                for (int ch = 0; ch < nch; ++ch)
                    for (int c = 0; c < CC; ++c)
                        for (int t = 0; t < T; ++t)
                            for (int co = 0; co < 64; ++co)
                                arranged[((((size_t)ct * nch + ch) * CC + c) * T + t) * 64 + co] =
                                    w[((size_t)(ct * 64 + co) * l.cin + ch * CC + c) * T + t];
        }
        float *dw = nullptr, *db = nullptr;
        if (hipMalloc((void **)&dw, arranged.size() * sizeof(float)) != hipSuccess) return fail(SR_ERR_OOM, "weights");
        M->d_w.push_back(dw);
        if (hipMalloc((void **)&db, (size_t)l.cout * sizeof(float)) != hipSuccess) { M->d_b.push_back(nullptr); return fail(SR_ERR_OOM, "bias"); }
        M->d_b.push_back(db);
        if (hipMemcpy(dw, arranged.data(), arranged.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(db, b, (size_t)l.cout * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            return fail(SR_ERR_HIP, "weight upload");
    }
    for (int i = 0; i < 5; ++i) {
        if (!h_lin_w[i]) return fail(SR_ERR_INVALID_ARG, "null lin array");
        if (hipMalloc((void **)&M->d_lin[i], (size_t)M->tap_c[i] * sizeof(float)) != hipSuccess) return fail(SR_ERR_OOM, "lin");
        if (hipMemcpy(M->d_lin[i], h_lin_w[i], (size_t)M->tap_c[i] * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            return fail(SR_ERR_HIP, "lin upload");
    }
    if (hipMalloc((void **)&M->d_acc, 5 * sizeof(double)) != hipSuccess) return fail(SR_ERR_OOM, "accumulators");
    *out = M;
    return SR_OK;
}

int sr_lpips_destroy(sr_lpips_model *m)
{
    if (!m) return SR_OK;
    {
        std::lock_guard<std::mutex> lk(g_lp_mu);
        if (!g_lp_live.erase(m)) return SR_OK;
    }
    if (ctx_is_live(m->ctx)) {
        Guard g(m->ctx);
        (void)hipStreamSynchronize(m->ctx->stream);
        for (auto p : m->d_w) if (p) (void)hipFree(p);
        for (auto p : m->d_b) if (p) (void)hipFree(p);
        for (auto p : m->d_lin) if (p) (void)hipFree(p);
        for (auto &im : m->buf) for (auto p : im) if (p) (void)hipFree(p);
        if (m->d_acc) (void)hipFree(m->d_acc);
        if (m->d_part) (void)hipFree(m->d_part);
    }
    delete m;
    return SR_OK;
}

int sr_lpips_layer_sizes(int net, int h, int w, int *h_hw)
{
    if (!h_hw || (net != SR_LPIPS_ALEX && net != SR_LPIPS_VGG)) return sr_set_error(SR_ERR_INVALID_ARG, "sr_lpips_layer_sizes: bad arguments");
    std::vector<LpLayer> L;
    lp_arch(net, L);
    LpGeom G;
    if (h < 1 || w < 1 || !lp_geometry(L, h, w, G)) return sr_set_error(SR_ERR_SHAPE, "sr_lpips: %dx%d image is too small for the network", w, h);
    for (size_t li = 0; li < L.size(); ++li)
        if (L[li].kind == LP_TAP) {
            h_hw[2 * L[li].tap] = G.H[G.act_of_layer[li]];
            h_hw[2 * L[li].tap + 1] = G.W[G.act_of_layer[li]];
        }
    return SR_OK;
}

int sr_lpips_tile_count(int h, int w, int tile, int *n_tiles)
{
    if (!n_tiles || h < 1 || w < 1) return sr_set_error(SR_ERR_INVALID_ARG, "sr_lpips_tile_count: bad arguments");
    if (tile <= 0) { *n_tiles = 1; return SR_OK; }
    if (tile % 16) return sr_set_error(SR_ERR_INVALID_ARG, "sr_lpips: tile size must be a multiple of 16 (the networks' total stride)");
    *n_tiles = ((h + tile - 1) / tile) * ((w + tile - 1) / tile);
    return SR_OK;
}

int sr_lpips_u8(sr_lpips_model *m, const uint8_t *d_a, int64_t stride_a, const uint8_t *d_b, int64_t stride_b, int h,
                int w, int cn, int tile, int tile_begin, int tile_end, double *h_layer_sums)
{
    if (!lp_is_live(m)) return sr_set_error(SR_ERR_INVALID_ARG, "sr_lpips_u8: null or destroyed model");
    sr_ctx *ctx = m->ctx;
    CTX_ENTER(ctx);
    if (!d_a || !d_b || !h_layer_sums) return sr_set_error(SR_ERR_INVALID_ARG, "sr_lpips_u8: null argument");
    if (cn != 1 && cn != 3 && cn != 4) return sr_set_error(SR_ERR_INVALID_ARG, "sr_lpips_u8: 1, 3 or 4 channels");
    if (stride_a < (int64_t)w * cn || stride_b < (int64_t)w * cn) return sr_set_error(SR_ERR_SHAPE, "sr_lpips_u8: stride smaller than a row");
    int ntiles = 0;
    int rc = sr_lpips_tile_count(h, w, tile, &ntiles);
    if (rc) return rc;
    LpGeom G;
    if (!lp_geometry(m->layers, h, w, G)) return sr_set_error(SR_ERR_SHAPE, "sr_lpips_u8: %dx%d image is too small for the network", w, h);
    if (tile <= 0) tile = std::max(h, w) + 16;
    const int tiles_x = tile >= w ? 1 : (w + tile - 1) / tile, tiles_y = tile >= h ? 1 : (h + tile - 1) / tile;
    ntiles = tiles_x * tiles_y;
    if (tile_end < 0 || tile_end > ntiles) tile_end = ntiles;
    tile_begin = std::max(tile_begin, 0);
    HIPCHK(hipMemsetAsync(m->d_acc, 0, 5 * sizeof(double), ctx->stream));
    const std::vector<LpLayer> &L = m->layers;
    const int na = (int)G.H.size();
    const dim3 blk(64, 4);
    for (int ti = tile_begin; ti < tile_end; ++ti) {
        const int ty = ti / tiles_x, tx = ti % tiles_x;
        std::vector<Range> ny, oy, nx, ox;
        lp_ranges(L, G, G.H, ty * tile, ty + 1 < tiles_y ? (ty + 1) * tile : -1, ny, oy);
        lp_ranges(L, G, G.W, tx * tile, tx + 1 < tiles_x ? (tx + 1) * tile : -1, nx, ox);
        // buffer geometry per activation (activation 0 is read from the image itself)
        std::vector<int> pitch(na, 0);
        std::vector<long long> plane(na, 0);
        size_t need_floats = 0;
        for (int j = 1; j < na; ++j) {
            pitch[j] = (nx[j].n() + 3) / 4 * 4;
            plane[j] = (long long)ny[j].n() * pitch[j];
            need_floats = std::max(need_floats, (size_t)plane[j] * G.C[j]);
        }
        if (need_floats > m->buf_floats) {
            HIPCHK(stream_sync(ctx));
            for (auto &im : m->buf)
                for (auto &p : im) {
                    if (p) (void)hipFree(p);
                    p = nullptr;
                }
            m->buf_floats = 0;
            for (auto &im : m->buf)
                for (auto &p : im) {
                    hipError_t e = hipMalloc((void **)&p, need_floats * sizeof(float));
                    if (e != hipSuccess)
                        return sr_set_error(e == hipErrorOutOfMemory ? SR_ERR_OOM : SR_ERR_HIP,
                                            "sr_lpips_u8: activation buffers (4 x %zu MB; use a smaller tile): %s",
                                            need_floats * 4 >> 20, hipGetErrorString(e));
                }
            m->buf_floats = need_floats;
        }
        int cur = 0;                       // ping-pong index of the current activation (both images in step)
        for (size_t li = 0; li < L.size(); ++li) {
            const LpLayer &l = L[li];
            const int ai = G.act_of_layer[li];
            if (l.kind == LP_TAP) {
                const Range ry = oy[ai], rx = ox[ai];
                if (ry.empty() || rx.empty()) continue;
                const dim3 grid((rx.n() + 63) / 64, (ry.n() + 3) / 4);
                const size_t nblk = (size_t)grid.x * grid.y;
                if (nblk > m->part_cap) {
                    HIPCHK(stream_sync(ctx));
                    if (m->d_part) (void)hipFree(m->d_part);
                    m->d_part = nullptr;
                    m->part_cap = 0;
                    HIPCHK(hipMalloc((void **)&m->d_part, nblk * sizeof(double)));
                    m->part_cap = nblk;
                }
                ProfScope ps(ctx, "lpips_tap");
                hipLaunchKernelGGL(k_lp_tap, grid, blk, 0, ctx->stream, m->buf[0][cur], m->buf[1][cur], plane[ai], pitch[ai],
                                   G.C[ai], m->d_lin[l.tap], ry.a - ny[ai].a, rx.a - nx[ai].a, ry.n(), rx.n(), m->d_part);
                hipLaunchKernelGGL(k_lp_accum, dim3(1), dim3(256), 0, ctx->stream, m->d_part, (long long)nblk, m->d_acc + l.tap);
                continue;
            }
            const int ao = ai + 1;
            const int rows = ny[ao].n(), cols = nx[ao].n();
            if (rows <= 0 || cols <= 0) { cur ^= (ai > 0); continue; }
            const int nxt = ai == 0 ? 0 : cur ^ 1;
            for (int im = 0; im < 2; ++im) {
                const float *src = ai == 0 ? nullptr : m->buf[im][cur];
                float *dst = m->buf[im][nxt];
                if (l.kind == LP_POOL) {
                    ProfScope ps(ctx, "lpips_pool");
                    const dim3 grid((cols + 63) / 64, (rows + 3) / 4, G.C[ao]);
                    if (l.k == 2) hipLaunchKernelGGL((k_lp_pool<2, 2>), grid, blk, 0, ctx->stream, src, plane[ai], pitch[ai], ny[ai].a, nx[ai].a, dst, plane[ao], pitch[ao], ny[ao].a, nx[ao].a, rows, cols);
                    else hipLaunchKernelGGL((k_lp_pool<3, 2>), grid, blk, 0, ctx->stream, src, plane[ai], pitch[ai], ny[ai].a, nx[ai].a, dst, plane[ao], pitch[ao], ny[ao].a, nx[ao].a, rows, cols);
                } else if (ai == 0) {
                    ProfScope ps(ctx, "lpips_conv_stem");
                    const uint8_t *img = im == 0 ? d_a : d_b;
                    const long long st = im == 0 ? stride_a : stride_b;
                    const dim3 grid((cols + 63) / 64, (rows + 3) / 4);
#define STEM_ARGS img, st, cn, h, w, m->d_w[l.widx], m->d_b[l.widx], m->shift[0], m->shift[1], m->shift[2], m->scale[0], \
                  m->scale[1], m->scale[2], dst, ny[ao].a, nx[ao].a, rows, cols, pitch[ao], plane[ao]
                    if (l.k == 3) hipLaunchKernelGGL((k_lp_conv_stem<3, 1, 1>), grid, blk, 0, ctx->stream, STEM_ARGS);
                    else hipLaunchKernelGGL((k_lp_conv_stem<11, 4, 2>), grid, blk, 0, ctx->stream, STEM_ARGS);
#undef STEM_ARGS
                } else {
                    ProfScope ps(ctx, "lpips_conv_mfma");
                    const dim3 grid((cols + 31) / 32, (rows + 7) / 8, l.cout / 64);
#define CONV_ARGS src, plane[ai], pitch[ai], ny[ai].a, nx[ai].a, G.H[ai], G.W[ai], l.cin, m->d_w[l.widx], m->d_b[l.widx], dst, \
                  plane[ao], pitch[ao], ny[ao].a, nx[ao].a, rows, cols
                    if (l.k == 3) hipLaunchKernelGGL((k_lp_conv_mfma<3, 8>), grid, dim3(256), 0, ctx->stream, CONV_ARGS);
                    else hipLaunchKernelGGL((k_lp_conv_mfma<5, 4>), grid, dim3(256), 0, ctx->stream, CONV_ARGS);
#undef CONV_ARGS
                }
            }
            cur = nxt;
        }
        rc = check_launch("lpips");
        if (rc) return rc;
    }
    HIPCHK(hipMemcpyAsync(h_layer_sums, m->d_acc, 5 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(stream_sync(ctx));
    return SR_OK;
}

}  // extern "C"

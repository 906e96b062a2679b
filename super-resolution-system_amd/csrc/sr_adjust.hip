// sr_adjust.hip -- BlendingModule.color_correction (blending_module.py:969-1146) on gfx950.
//
//   _histogram_matching : per-channel 256-bin histograms on the GPU (LDS bins per block, one atomic per bin per block);
//                         the CDF / argmin table is 256-entry host bookkeeping (Python mirror) and comes back as a table.
//   _mean_std_matching  : the affine map evaluated per source value -> the same kind of table.
//   _simple_guided_filter(guide = corrected, src = image): five cv2.blur box means and fp32 element-wise algebra, as two
//                         passes: k_cc_coeff (box means of g, s, g*s, g*g -> a, b) and k_cc_apply (box means of a, b ->
//                         mean_a * g + mean_b -> clip -> truncate).  Box sums are accumulated in fp64 like cv::boxFilter
//                         does for CV_32F data (sum type CV_64F) in a fixed order: each window row left to right, then the
//                         row sums top to bottom; mean = (float)(sum * (1.0 / (r * r))).  Separable through LDS: a block
//                         owns 64 x 16 output pixels of one channel at a time.  Since round 3 the reference's setting (radius 8,
//                         integer-valued guide table) is ONE kernel, k_cc_fused8: exact 32-bit sliding sums for the first
//                         stage, a and b kept in LDS (9.9 -> 3.4 ms at 200 MP); round 4: k_cc_fused8f does the same with
//                         fp64 sliding sums for a float (mean_std) table whose box sums are provably exact (cc_table_class;
//                         9.9 -> 4.3 ms); the passes above remain for the other float tables and other radii.
//   cv2.ximgproc.guidedFilter (the branch a requirements-complete install takes; parity unpinned): k_gfx_coeff17 (all 21 first-
//                         stage means + the 3 x 3 inverse in one kernel, integer tables) and k_gf_box17_out (second stage +
//                         output per channel); k_gf_box / k_gf_box17 / k_gf_coeff / k_gf_out for float tables, gray images
//                         and other radii.
// Byte / fp32 / ordered-fp64 stencil work, bound by VALU issue and latency rather than by HBM once a / b stay on chip; no MFMA.
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sr_ctx.h"

namespace {

// experiments / A-B runs: NAME=0 switches a default-on path off
bool env_flag_off(const char *name)
{
    const char *e = std::getenv(name);
    return e && e[0] == '0';
}

__device__ __forceinline__ int cc_reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// 12 bytes per thread and step (a multiple of every channel count 1..4, so byte j of a group is channel j % cn), bins
// replicated four times per block (lane & 3): neighbouring lanes read neighbouring, mostly equal pixels, and equal values in
// one LDS atomic serialise.
typedef unsigned adj_u3_t __attribute__((ext_vector_type(3)));
typedef adj_u3_t adj_u3_a1_t __attribute__((aligned(1)));

__global__ __launch_bounds__(256) void k_hist_u8(const unsigned char *__restrict__ img, long long stride, int h, int w, int cn,
                                                 unsigned long long *__restrict__ hist)
{
    __shared__ unsigned bins[4][4][256];                       // [replica][channel][value]
    const int tid = threadIdx.x, rep = tid & 3;
    for (int i = tid; i < 4 * 4 * 256; i += 256) (&bins[0][0][0])[i] = 0u;
    __syncthreads();
    const long long rowlen = (long long)w * cn, ngroups = rowlen / 12;
    int ch[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) ch[j] = j % cn;
    for (int y = blockIdx.y; y < h; y += gridDim.y) {
        const unsigned char *row = img + (size_t)y * stride;
        for (long long g = (long long)blockIdx.x * 256 + tid; g < ngroups; g += (long long)gridDim.x * 256) {
            const adj_u3_t q = *(const __attribute__((address_space(1))) adj_u3_a1_t *)(row + g * 12);
            const unsigned wd[3] = {q.x, q.y, q.z};
#pragma unroll
            for (int j = 0; j < 12; ++j) atomicAdd(&bins[rep][ch[j]][(wd[j >> 2] >> (8 * (j & 3))) & 0xFFu], 1u);
        }
        if (blockIdx.x == 0)                                    // the row's tail (< 12 bytes)
            for (long long i = ngroups * 12 + tid; i < rowlen; i += 256) atomicAdd(&bins[rep][(int)(i % cn)][row[i]], 1u);
    }
    __syncthreads();
    for (int i = tid; i < cn * 256; i += 256) {
        const unsigned v = ((&bins[0][0][0])[i] + (&bins[1][0][0])[i]) + ((&bins[2][0][0])[i] + (&bins[3][0][0])[i]);
        if (v) atomicAdd(&hist[i], (unsigned long long)v);
    }
}

// TilingModule.split_image's complexity_score = np.std(cv2.cvtColor(tile, COLOR_BGR2GRAY)) (tiling_module.py:746-749: the
// BGR constants applied to RGB data, i.e. R and B weights swapped): exact integer sums of g and g^2 per tile.
__global__ __launch_bounds__(256) void k_gray_moments(const unsigned char *__restrict__ tiles, long long tile_bytes,
                                                      long long stride, int h, int w, int shift, int swap_rb,
                                                      unsigned long long *__restrict__ sums)
{
    const unsigned char *base = tiles + (size_t)blockIdx.z * tile_bytes;
    unsigned long long s1 = 0, s2 = 0;
#pragma unroll 4
    for (int y = blockIdx.y; y < h; y += gridDim.y) {
        const unsigned char *row = base + (size_t)y * stride;
        auto gray = [&](int c0, int c1, int c2) {
            const int r = swap_rb ? c2 : c0, b = swap_rb ? c0 : c2;
            return shift == 15 ? (r * 9798 + c1 * 19235 + b * 3735 + (1 << 14)) >> 15 : (r * 4899 + c1 * 9617 + b * 1868 + (1 << 13)) >> 14;
        };
        const int ngroups = w / 4;                              // four pixels = 12 bytes per thread and step
        for (int gi = blockIdx.x * 256 + threadIdx.x; gi < ngroups; gi += gridDim.x * 256) {
            const adj_u3_t q = *(const __attribute__((address_space(1))) adj_u3_a1_t *)(row + (size_t)gi * 12);
            const unsigned wd[3] = {q.x, q.y, q.z};
            unsigned t1 = 0, t2 = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i0 = 3 * k, i1 = 3 * k + 1, i2 = 3 * k + 2;
                const int g = gray((int)((wd[i0 >> 2] >> (8 * (i0 & 3))) & 0xFFu), (int)((wd[i1 >> 2] >> (8 * (i1 & 3))) & 0xFFu),
                                   (int)((wd[i2 >> 2] >> (8 * (i2 & 3))) & 0xFFu));
                t1 += (unsigned)g;
                t2 += (unsigned)(g * g);
            }
            s1 += t1;
            s2 += t2;
        }
        if (blockIdx.x == 0)
            for (int x = ngroups * 4 + threadIdx.x; x < w; x += 256) {
                const int g = gray(row[3 * x], row[3 * x + 1], row[3 * x + 2]);
                s1 += (unsigned)g;
                s2 += (unsigned)(g * g);
            }
    }
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_down(s1, o, 64);
        s2 += __shfl_down(s2, o, 64);
    }
    __shared__ unsigned long long ws[4][2];
    if ((threadIdx.x & 63) == 0) { ws[threadIdx.x >> 6][0] = s1; ws[threadIdx.x >> 6][1] = s2; }
    __syncthreads();
    if (threadIdx.x < 2) {                                      // one atomic per block and moment (a few addresses take them all)
        const unsigned long long t = (ws[0][threadIdx.x] + ws[1][threadIdx.x]) + (ws[2][threadIdx.x] + ws[3][threadIdx.x]);
        if (t) atomicAdd(&sums[2 * blockIdx.z + threadIdx.x], t);
    }
}

#define CC_TW 64
#define CC_TH 16

// pass 1: a = cov(g, s) / (var(g) + eps), b = mean_s - a * mean_g per pixel and channel (fp32 HWC planes a, b)
__global__ __launch_bounds__(256) void k_cc_coeff(const unsigned char *__restrict__ img, long long stride, int h, int w, int cn,
                                                  const float *__restrict__ glut, int R, float eps, double scale,
                                                  float *__restrict__ a_out, float *__restrict__ b_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int PH = CC_TH + R - 1, PW = CC_TW + R - 1, anchor = R / 2;
    float *g_p = (float *)smem;                               // [PH][PW]
    float *s_p = g_p + PH * PW;
    double *hs = (double *)(smem + (((size_t)2 * PH * PW * sizeof(float) + 15) & ~(size_t)15));   // [PH][TW][4]
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * CC_TW, y0 = blockIdx.y * CC_TH;
    for (int c = 0; c < cn; ++c) {
        for (int e = tid; e < PH * PW; e += 256) {
            const int py = e / PW, px = e - py * PW;
            const int gy = cc_reflect101(y0 + py - anchor, h), gx = cc_reflect101(x0 + px - anchor, w);
            const int v = img[(size_t)gy * stride + (size_t)gx * cn + c];
            g_p[e] = glut[c * 256 + v];
            s_p[e] = (float)v;
        }
        __syncthreads();
        for (int e = tid; e < PH * CC_TW; e += 256) {
            const int py = e / CC_TW, ox = e - py * CC_TW;
            double sg = 0.0, ss = 0.0, sgs = 0.0, sgg = 0.0;
            for (int k = 0; k < R; ++k) {
                const float g = g_p[py * PW + ox + k], s = s_p[py * PW + ox + k];
                sg += (double)g;
                ss += (double)s;
                sgs += (double)(g * s);
                sgg += (double)(g * g);
            }
            double *o = hs + (size_t)e * 4;
            o[0] = sg; o[1] = ss; o[2] = sgs; o[3] = sgg;
        }
        __syncthreads();
        for (int e = tid; e < CC_TH * CC_TW; e += 256) {
            const int oy = e / CC_TW, ox = e - oy * CC_TW;
            const int y = y0 + oy, x = x0 + ox;
            if (y >= h || x >= w) continue;
            double t[4] = {0.0, 0.0, 0.0, 0.0};
            for (int k = 0; k < R; ++k) {
                const double *r = hs + ((size_t)(oy + k) * CC_TW + ox) * 4;
                t[0] += r[0]; t[1] += r[1]; t[2] += r[2]; t[3] += r[3];
            }
            const float mg = (float)(t[0] * scale), ms = (float)(t[1] * scale);
            const float mgs = (float)(t[2] * scale), mgg = (float)(t[3] * scale);
            const float cov = mgs - mg * ms, var = mgg - mg * mg;
            const float a = cov / (var + eps);
            const float b = ms - a * mg;
            const size_t o = ((size_t)y * w + x) * cn + c;
            a_out[o] = a;
            b_out[o] = b;
        }
        __syncthreads();
    }
}

// pass 2: out = u8(clip(mean_a * g + mean_b, 0, 255))
__global__ __launch_bounds__(256) void k_cc_apply(const unsigned char *__restrict__ img, long long stride, int h, int w, int cn,
                                                  const float *__restrict__ glut, int R, double scale,
                                                  const float *__restrict__ a_in, const float *__restrict__ b_in,
                                                  unsigned char *__restrict__ out, long long ostride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int PH = CC_TH + R - 1, PW = CC_TW + R - 1, anchor = R / 2;
    float *a_p = (float *)smem;
    float *b_p = a_p + PH * PW;
    double *hs = (double *)(smem + (((size_t)2 * PH * PW * sizeof(float) + 15) & ~(size_t)15));   // [PH][TW][2]
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * CC_TW, y0 = blockIdx.y * CC_TH;
    for (int c = 0; c < cn; ++c) {
        for (int e = tid; e < PH * PW; e += 256) {
            const int py = e / PW, px = e - py * PW;
            const int gy = cc_reflect101(y0 + py - anchor, h), gx = cc_reflect101(x0 + px - anchor, w);
            const size_t o = ((size_t)gy * w + gx) * cn + c;
            a_p[e] = a_in[o];
            b_p[e] = b_in[o];
        }
        __syncthreads();
        for (int e = tid; e < PH * CC_TW; e += 256) {
            const int py = e / CC_TW, ox = e - py * CC_TW;
            double sa = 0.0, sb = 0.0;
            for (int k = 0; k < R; ++k) {
                sa += (double)a_p[py * PW + ox + k];
                sb += (double)b_p[py * PW + ox + k];
            }
            hs[(size_t)e * 2] = sa;
            hs[(size_t)e * 2 + 1] = sb;
        }
        __syncthreads();
        for (int e = tid; e < CC_TH * CC_TW; e += 256) {
            const int oy = e / CC_TW, ox = e - oy * CC_TW;
            const int y = y0 + oy, x = x0 + ox;
            if (y >= h || x >= w) continue;
            double ta = 0.0, tb = 0.0;
            for (int k = 0; k < R; ++k) {
                ta += hs[((size_t)(oy + k) * CC_TW + ox) * 2];
                tb += hs[((size_t)(oy + k) * CC_TW + ox) * 2 + 1];
            }
            const float ma = (float)(ta * scale), mb = (float)(tb * scale);
            const float g = glut[c * 256 + img[(size_t)y * stride + (size_t)x * cn + c]];
            const float r = ma * g + mb;
            const float cl = r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r);
            out[(size_t)y * ostride + (size_t)x * cn + c] = (unsigned char)cl;
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------------------------
// The guided filter at its reference setting (8 x 8 box means, blending_module.py:1110-1146) register-blocked: a thread
// forms FOUR neighbouring window sums from eleven values it reads once (row pass: four columns of one patch row; column
// pass: four rows of one column), and the four maps go through LDS two at a time (24 KB instead of 47 KB of row sums: four
// blocks per CU).  Every sum still adds its eight terms in the oracle's order (left to right, top to bottom, in fp64), so
// the result is bit-identical to the generic kernels below, which remain for other window sizes.  a / b are PLANAR
// [c][h][w] here (coalesced in both passes).  200 MP image: 10.5 + 6.4 ms -> see DESIGN.md.
// ---------------------------------------------------------------------------------------------------------------
#define CC8_R 8
#define CC8_PH (CC_TH + CC8_R - 1)        /* 23 patch rows */
#define CC8_PW 72                         /* 71 patch columns, padded */
#define CC8_LDS (2 * CC8_PH * CC8_PW * 4 + CC8_PH * CC_TW * 2 * 8 + 1024)

// four outputs j = 0..3, each the sum of v[j .. j + 7] in that order
__device__ __forceinline__ void cc8_sums(const double (&v)[11], double (&o)[4])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < CC8_R; ++k) t += v[j + k];
        o[j] = t;
    }
}

// column pass over the row sums of one pair of maps: thread (gy, ox) -> rows 4 gy .. 4 gy + 3
__device__ __forceinline__ void cc8_cols(const double *__restrict__ hs, int gy, int ox, double (&t0)[4], double (&t1)[4])
{
    double v0[11], v1[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        const double *r = hs + ((size_t)(4 * gy + k) * CC_TW + ox) * 2;
        v0[k] = r[0];
        v1[k] = r[1];
    }
    cc8_sums(v0, t0);
    cc8_sums(v1, t1);
}

__global__ __launch_bounds__(256) void k_cc_coeff8(const unsigned char *__restrict__ img, long long stride, int h, int w, int cn,
                                                   const float *__restrict__ glut, float eps, double scale,
                                                   float *__restrict__ a_out, float *__restrict__ b_out)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[CC8_LDS];
    float *g_p = (float *)smem;                                   // [PH][PW]
    float *s_p = g_p + CC8_PH * CC8_PW;
    double *hs = (double *)(s_p + CC8_PH * CC8_PW);               // [PH][TW][2]
    float *lut = (float *)(hs + CC8_PH * CC_TW * 2);              // [256]
    const int tid = threadIdx.x, anchor = CC8_R / 2;
    const int x0 = blockIdx.x * CC_TW, y0 = blockIdx.y * CC_TH;
    const size_t plane = (size_t)h * w;
    for (int c = 0; c < cn; ++c) {
        lut[tid] = glut[c * 256 + tid];
        __syncthreads();                                          // also: the previous channel's passes are done
        for (int e = tid; e < CC8_PH * CC8_PW; e += 256) {
            const int py = e / CC8_PW, px = e - py * CC8_PW;
            const int gy = cc_reflect101(y0 + py - anchor, h), gx = cc_reflect101(x0 + px - anchor, w);
            const int v = img[(size_t)gy * stride + (size_t)gx * cn + c];
            g_p[e] = lut[v];
            s_p[e] = (float)v;
        }
        __syncthreads();
        double keep[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};   // column sums of g and s (first pair) of this thread
#pragma unroll
        for (int pair = 0; pair < 2; ++pair) {
            // row pass: item (py, gx) -> columns 4 gx .. 4 gx + 3 of patch row py; pair 0: g, s   pair 1: g * s, g * g
            for (int e = tid; e < CC8_PH * (CC_TW / 4); e += 256) {
                const int py = e / (CC_TW / 4), ox = (e - py * (CC_TW / 4)) * 4;
                double v0[11], v1[11];
#pragma unroll
                for (int k = 0; k < 11; ++k) {
                    const float g = g_p[py * CC8_PW + ox + k], sv = s_p[py * CC8_PW + ox + k];
                    v0[k] = pair == 0 ? (double)g : (double)(g * sv);
                    v1[k] = pair == 0 ? (double)sv : (double)(g * g);
                }
                double o0[4], o1[4];
                cc8_sums(v0, o0);
                cc8_sums(v1, o1);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    double *o = hs + ((size_t)py * CC_TW + ox + j) * 2;
                    o[0] = o0[j];
                    o[1] = o1[j];
                }
            }
            __syncthreads();
            const int gy = tid >> 6, ox = tid & 63;               // column pass: 4 x 64 items = the block
            double t0[4], t1[4];
            cc8_cols(hs, gy, ox, t0, t1);
            if (pair == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { keep[0][j] = t0[j]; keep[1][j] = t1[j]; }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int y = y0 + 4 * gy + j, x = x0 + ox;
                    if (y >= h || x >= w) continue;
                    const float mg = (float)(keep[0][j] * scale), ms = (float)(keep[1][j] * scale);
                    const float mgs = (float)(t0[j] * scale), mgg = (float)(t1[j] * scale);
                    const float cov = mgs - mg * ms, var = mgg - mg * mg;
                    const float a = cov / (var + eps);
                    const float b = ms - a * mg;
                    a_out[c * plane + (size_t)y * w + x] = a;
                    b_out[c * plane + (size_t)y * w + x] = b;
                }
            }
            __syncthreads();                                      // hs is free for the next pair / channel
        }
    }
}

__global__ __launch_bounds__(256) void k_cc_apply8(const unsigned char *__restrict__ img, long long stride, int h, int w, int cn,
                                                   const float *__restrict__ glut, double scale, const float *__restrict__ a_in,
                                                   const float *__restrict__ b_in, unsigned char *__restrict__ out, long long ostride)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[CC8_LDS];
    float *a_p = (float *)smem;
    float *b_p = a_p + CC8_PH * CC8_PW;
    double *hs = (double *)(b_p + CC8_PH * CC8_PW);
    float *lut = (float *)(hs + CC8_PH * CC_TW * 2);
    const int tid = threadIdx.x, anchor = CC8_R / 2;
    const int x0 = blockIdx.x * CC_TW, y0 = blockIdx.y * CC_TH;
    const size_t plane = (size_t)h * w;
    for (int c = 0; c < cn; ++c) {
        lut[tid] = glut[c * 256 + tid];
        __syncthreads();
        for (int e = tid; e < CC8_PH * CC8_PW; e += 256) {
            const int py = e / CC8_PW, px = e - py * CC8_PW;
            const int gy = cc_reflect101(y0 + py - anchor, h), gx = cc_reflect101(x0 + px - anchor, w);
            const size_t o = c * plane + (size_t)gy * w + gx;
            a_p[e] = a_in[o];
            b_p[e] = b_in[o];
        }
        __syncthreads();
        for (int e = tid; e < CC8_PH * (CC_TW / 4); e += 256) {
            const int py = e / (CC_TW / 4), ox = (e - py * (CC_TW / 4)) * 4;
            double v0[11], v1[11];
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                v0[k] = (double)a_p[py * CC8_PW + ox + k];
                v1[k] = (double)b_p[py * CC8_PW + ox + k];
            }
            double o0[4], o1[4];
            cc8_sums(v0, o0);
            cc8_sums(v1, o1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                double *o = hs + ((size_t)py * CC_TW + ox + j) * 2;
                o[0] = o0[j];
                o[1] = o1[j];
            }
        }
        __syncthreads();
        const int gy = tid >> 6, ox = tid & 63;
        double ta[4], tb[4];
        cc8_cols(hs, gy, ox, ta, tb);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int y = y0 + 4 * gy + j, x = x0 + ox;
            if (y >= h || x >= w) continue;
            const float ma = (float)(ta[j] * scale), mb = (float)(tb[j] * scale);
            const float g = lut[img[(size_t)y * stride + (size_t)x * cn + c]];
            const float r = ma * g + mb;
            const float cl = r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r);
            out[(size_t)y * ostride + (size_t)x * cn + c] = (unsigned char)cl;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 3: the guided filter in ONE kernel -- a and b never leave the CU (the two-pass form above writes and re-reads 24
// bytes per pixel and spends most of its time waiting for those loads).  Used when the guide table is integer-valued
// (histogram matching and method 'none': every entry of the table is a whole number 0..255), because then the four
// first-stage box sums are sums of integers < 2^24 -- exact in any order, so they run as 32-bit sliding sums (the sums of
// g and s share one word, 16 bits each) and mean = (float)S * (1 / 64) is the float the oracle's fp64 sum rounds to.
// The second stage (box means of a and b, arbitrary floats) keeps the oracle's fp64 order: row sums left to right, then
// top to bottom.
//
// A block of 512 threads owns 64 x 32 output pixels and takes the channels in turn:
//   load   the (64 + 14) x (32 + 14) input window, all channels, once (dword loads; border blocks reflect per byte);
//   B      column sums of 8 rows of (g | s << 16), g * s, g * g: item = (column, 7-row segment), sliding;
//   C      row sums of 8 columns of those, sliding, then a = cov / (var + eps), b = mean_s - a * mean_g at the
//          (64 + 7) x (32 + 7) positions the second stage reads; positions outside the image take the value of their
//          BORDER_REFLECT_101 mirror position (cv2.blur's border rule applied to the a / b images: a plain copy);
//   D, E   the ordered fp64 row / column sums of a and b (four neighbouring sums from eleven values, as above), then
//          u8(clip(mean_a * g + mean_b)) written over the input byte it came from;
//   store  the window's interior, now the output tile, with dword stores.
// LDS: 39 KB (column sums, later the fp64 row sums) + 22 KB (a, b) + 1 KB table + 11 KB window = 73 KB, two blocks per CU.
// ---------------------------------------------------------------------------------------------------------------
#define FG_TH 32
#define FG_TW 64
#define FG_NT 512
#define FG_AH (FG_TH + 7)                  /* rows of a / b */
#define FG_AW (FG_TW + 7)                  /* columns of a / b */
#define FG_AP 72                           /* their pitch */
#define FG_IH (FG_TH + 14)                 /* input window */
#define FG_IW (FG_TW + 14)
#define FG_VP 80                           /* pitch of the column-sum planes */
#ifndef FG_SEG
#define FG_SEG 7                           /* rows per item of pass B */
#endif
#define FG_NSEG ((FG_AH + FG_SEG - 1) / FG_SEG)   /* 6 x 7 rows; 4 x 10 and 3 x 13 issue 8 % fewer instructions and are 2-3 % slower: latency, not issue */
#define FG_Y_BYTES (FG_AH * FG_TW * 16)    /* fp64 row sums of (a, b): 39936 >= 3 * 39 * 80 * 4 */
#define FG_X_BYTES (2 * FG_AH * FG_AP * 4)
static_assert(FG_Y_BYTES >= 3 * FG_AH * FG_VP * 4, "column-sum planes must fit the row-sum region");

typedef unsigned fg_u32_a1_t __attribute__((aligned(1)));
typedef float fg_f2_t __attribute__((ext_vector_type(2)));

// four outputs j = 0..3, each v[j] + v[j + 1] + ... + v[j + 7] in that order (cc8_sums without the leading 0.0 + v[j]: that
// addition only turns a -0.0 into +0.0, which no stored byte can tell apart)
__device__ __forceinline__ void fg_sums(const double (&v)[11], double (&o)[4])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double t = v[j];
#pragma unroll
        for (int k = 1; k < CC8_R; ++k) t += v[j + k];
        o[j] = t;
    }
}

// cov / den for two positions at once, IEEE-correct: the fma chain hipcc emits for an fp32 division (rcp, one Newton step,
// quotient, two residual corrections) without the v_div_scale / v_div_fixup wrapping, which is the identity here -- den =
// var + eps lies in [0.006, 65026] (var >= -0.004: the rounding of mean_g * mean_g) and cov is 0 or 2^-13 <= |cov| <= 65025,
// so nothing is scaled, nothing over- or underflows.  v_pk_mul_f32 / v_pk_fma_f32 round per element like the scalar forms.
__device__ __forceinline__ fg_f2_t fg_div2(fg_f2_t num, fg_f2_t den)
{
    fg_f2_t r;
    r.x = __builtin_amdgcn_rcpf(den.x);
    r.y = __builtin_amdgcn_rcpf(den.y);
    const fg_f2_t nd = -den, one = {1.0f, 1.0f};
    r = __builtin_elementwise_fma(__builtin_elementwise_fma(nd, r, one), r, r);
    fg_f2_t t = num * r;
    t = __builtin_elementwise_fma(__builtin_elementwise_fma(nd, t, num), r, t);
    return __builtin_elementwise_fma(__builtin_elementwise_fma(nd, t, num), r, t);
}

template <int CN>
__global__ __launch_bounds__(FG_NT) void k_cc_fused8(const unsigned char *__restrict__ img, long long stride, int h, int w,
                                                     const unsigned char *__restrict__ glutb, float eps,
                                                     unsigned char *__restrict__ out, long long ostride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int rowb = CN * FG_IW, rawp = (rowb + 3) & ~3;
    constexpr int LUT0 = FG_Y_BYTES + FG_X_BYTES, RAW0 = LUT0 + 1024;
    unsigned *V1 = (unsigned *)smem, *V2 = V1 + FG_AH * FG_VP, *V3 = V2 + FG_AH * FG_VP;
    double *hs = (double *)smem;                                  // [AH][64 positions][2] once V1..V3 are consumed
    float *a_p = (float *)(smem + FG_Y_BYTES), *b_p = a_p + FG_AH * FG_AP;
    unsigned char *lutb = smem + LUT0;                            // [CN][256]
    unsigned char *raw = smem + RAW0;                             // [IH][rawp], pixels interleaved as in the image
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * FG_TW, y0 = blockIdx.y * FG_TH;

    for (int i = tid; i < CN * 256; i += FG_NT) lutb[i] = glutb[i];
    if (x0 >= 8 && x0 - 8 + FG_IW <= w && y0 >= 8 && y0 - 8 + FG_IH <= h) {
        constexpr int ndw = rowb >> 2, tail = rowb & 3, per = ndw + (tail ? 1 : 0);
        for (int e = tid; e < FG_IH * per; e += FG_NT) {
            const int i = e / per, d = e - i * per;
            const unsigned char *src = img + (size_t)(y0 - 8 + i) * stride + (size_t)(x0 - 8) * CN + 4 * d;
            if (d < ndw) *(unsigned *)(raw + i * rawp + 4 * d) = *(const __attribute__((address_space(1))) fg_u32_a1_t *)src;
            else for (int t = 0; t < tail; ++t) raw[i * rawp + 4 * d + t] = src[t];
        }
    } else {
        for (int e = tid; e < FG_IH * rowb; e += FG_NT) {
            const int i = e / rowb, r = e - i * rowb, j = r / CN, c = r - j * CN;
            raw[i * rawp + r] = img[(size_t)cc_reflect101(y0 - 8 + i, h) * stride + (size_t)cc_reflect101(x0 - 8 + j, w) * CN + c];
        }
    }
    const bool edge = y0 < 4 || y0 - 4 + FG_AH > h || x0 < 4 || x0 - 4 + FG_AW > w;   // some a / b position lies outside
    __syncthreads();

#pragma unroll
    for (int c = 0; c < CN; ++c) {                                // unrolled: table and window offsets become immediates
        const unsigned char *lt = lutb + c * 256;
        // ---- B: column sums ----
        if (tid < FG_NSEG * FG_IW) {
            // the last segment ends with row 38 and repeats rows of the one before (same values): no row guards
            const int seg = tid / FG_IW, col = tid - seg * FG_IW, r0 = min(seg * FG_SEG, FG_AH - FG_SEG);
            const unsigned char *rp = raw + r0 * rawp + col * CN + c;
            unsigned p1[FG_SEG + 7], p2[FG_SEG + 7], p3[FG_SEG + 7];
#pragma unroll
            for (int k = 0; k < FG_SEG + 7; ++k) {
                const unsigned s = rp[k * rawp], g = lt[s];
                p1[k] = g | (s << 16);
                // opaque products: hipcc 7.2 folds sums of byte products into v_perm_b32 + v_dot4_u32_u8 and gets them wrong
                asm("v_mul_u32_u24_e32 %0, %1, %2" : "=v"(p2[k]) : "v"(g), "v"(s));
                asm("v_mul_u32_u24_e32 %0, %1, %1" : "=v"(p3[k]) : "v"(g));
            }
            unsigned s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) { s1 += p1[k]; s2 += p2[k]; s3 += p3[k]; }
#pragma unroll
            for (int k = 0; k < FG_SEG; ++k) {
                if (k > 0) {
                    s1 += p1[k + 7] - p1[k - 1];
                    s2 += p2[k + 7] - p2[k - 1];
                    s3 += p3[k + 7] - p3[k - 1];
                }
                V1[(r0 + k) * FG_VP + col] = s1;
                V2[(r0 + k) * FG_VP + col] = s2;
                V3[(r0 + k) * FG_VP + col] = s3;
            }
        }
        __syncthreads();
        // ---- C: row sums, a and b (two positions per packed instruction) ----
        if (tid < FG_AH * 12) {
            const int row = tid / 12, q0 = (tid - row * 12) * 6;
            const unsigned *v1 = V1 + row * FG_VP + q0, *v2 = V2 + row * FG_VP + q0, *v3 = V3 + row * FG_VP + q0;
            unsigned p1[13], p2[13], p3[13];
#pragma unroll
            for (int k = 0; k < 13; ++k) { p1[k] = v1[k]; p2[k] = v2[k]; p3[k] = v3[k]; }
            unsigned s1[6], s2[6], s3[6];
            s1[0] = s2[0] = s3[0] = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) { s1[0] += p1[k]; s2[0] += p2[k]; s3[0] += p3[k]; }
#pragma unroll
            for (int t = 1; t < 6; ++t) {
                s1[t] = s1[t - 1] + (p1[t + 7] - p1[t - 1]);
                s2[t] = s2[t - 1] + (p2[t + 7] - p2[t - 1]);
                s3[t] = s3[t - 1] + (p3[t + 7] - p3[t - 1]);
            }
            const fg_f2_t k64 = {0.015625f, 0.015625f}, e2 = {eps, eps};
#pragma unroll
            for (int t = 0; t < 6; t += 2) {
                fg_f2_t mg, ms, mgs, mgg;
                mg.x = (float)(s1[t] & 0xFFFFu);      mg.y = (float)(s1[t + 1] & 0xFFFFu);
                ms.x = (float)(s1[t] >> 16);          ms.y = (float)(s1[t + 1] >> 16);
                mgs.x = (float)s2[t];                 mgs.y = (float)s2[t + 1];
                mgg.x = (float)s3[t];                 mgg.y = (float)s3[t + 1];
                mg = mg * k64; ms = ms * k64; mgs = mgs * k64; mgg = mgg * k64;
                const fg_f2_t cov = mgs - mg * ms, var = mgg - mg * mg;
                const fg_f2_t a = fg_div2(cov, var + e2);
                const fg_f2_t b = ms - a * mg;
                if (q0 + t < FG_AW) {                              // q0 + t is even, FG_AW odd: the pair's second half may be beyond
                    a_p[row * FG_AP + q0 + t] = a.x;
                    b_p[row * FG_AP + q0 + t] = b.x;
                    if (q0 + t + 1 < FG_AW) {
                        a_p[row * FG_AP + q0 + t + 1] = a.y;
                        b_p[row * FG_AP + q0 + t + 1] = b.y;
                    }
                }
            }
        }
        __syncthreads();
        if (edge) {                                               // block-uniform
            for (int e = tid; e < FG_AH * FG_AW; e += FG_NT) {
                const int r = e / FG_AW, q = e - r * FG_AW;
                const int py = y0 - 4 + r, px = x0 - 4 + q;
                if (py < 0 || py >= h || px < 0 || px >= w) {
                    const int sr = cc_reflect101(py, h) - (y0 - 4), sq = cc_reflect101(px, w) - (x0 - 4);
                    if (sr >= 0 && sr < FG_AH && sq >= 0 && sq < FG_AW) {   // else: a position no stored pixel reads
                        a_p[r * FG_AP + q] = a_p[sr * FG_AP + sq];
                        b_p[r * FG_AP + q] = b_p[sr * FG_AP + sq];
                    }
                }
            }
            __syncthreads();
        }
        // ---- D: ordered row sums of a and b; the sums of output column 4 gx + j go to position j * 16 + gx of their row ----
        for (int e = tid; e < FG_AH * 16; e += FG_NT) {
            const int py = e >> 4, gx = e & 15;
            const float4 *ar = (const float4 *)(a_p + py * FG_AP + 4 * gx), *br = (const float4 *)(b_p + py * FG_AP + 4 * gx);
            const float4 a0 = ar[0], a1 = ar[1], a2 = ar[2], b0 = br[0], b1 = br[1], b2 = br[2];
            const double v0[11] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z};
            const double v1[11] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z};
            double o0[4], o1[4];
            fg_sums(v0, o0);
            fg_sums(v1, o1);
#pragma unroll
            for (int j = 0; j < 4; ++j) *(double2 *)(hs + ((size_t)py * FG_TW + j * 16 + gx) * 2) = make_double2(o0[j], o1[j]);
        }
        __syncthreads();
        // ---- E: ordered column sums, output ----
        {
            const int gy = tid >> 6, l = tid & 63, ox = 4 * (l & 15) + (l >> 4);
            double v0[11], v1[11];
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const double2 t = *(const double2 *)(hs + ((size_t)(4 * gy + k) * FG_TW + l) * 2);
                v0[k] = t.x;
                v1[k] = t.y;
            }
            double ta[4], tb[4];
            fg_sums(v0, ta);
            fg_sums(v1, tb);
            unsigned char *px = raw + (4 * gy + 8) * rawp + (ox + 8) * CN + c;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float ma = (float)(ta[j] * 0.015625), mb = (float)(tb[j] * 0.015625);
                const float g = (float)lt[px[j * rawp]];
                const float r = ma * g + mb;                       // finite: clip = median of (r, 0, 255)
                px[j * rawp] = (unsigned char)__builtin_amdgcn_fmed3f(r, 0.0f, 255.0f);
            }
        }
        __syncthreads();                                          // hs becomes V1..V3 again
    }
    constexpr int tileb = FG_TW * CN;                             // bytes of one output row of the block
    if (x0 + FG_TW <= w && y0 + FG_TH <= h) {
        constexpr int ndw = tileb >> 2;                           // 64 CN is a multiple of 4
        for (int e = tid; e < FG_TH * ndw; e += FG_NT) {
            const int oy = e / ndw, d = e - oy * ndw;
            *(__attribute__((address_space(1))) fg_u32_a1_t *)(out + (size_t)(y0 + oy) * ostride + (size_t)x0 * CN + 4 * d) =
                *(const unsigned *)(raw + (oy + 8) * rawp + 8 * CN + 4 * d);
        }
    } else {
        const int vw = min(FG_TW, w - x0) * CN, vh = min(FG_TH, h - y0);
        for (int e = tid; e < vh * tileb; e += FG_NT) {
            const int oy = e / tileb, r = e - oy * tileb;
            if (r < vw) out[(size_t)(y0 + oy) * ostride + (size_t)x0 * CN + r] = raw[(oy + 8) * rawp + 8 * CN + r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 4: the same kernel for a FLOAT guide table (method 'mean_std': table[v] = (v - mean_s) * gain + mean_r, any float).
// The first-stage maps are functions of the source byte alone -- g = T[v], s = v, g * s = fl32(T[v] * v), g * g =
// fl32(T[v]^2) -- so the host can look at all 3 x 256 values a channel can ever add up and decide whether a box sum of
// them is EXACT in fp64 (cc_table_class below: every value a multiple of 2^lo, |value| < 2^hi, hi + 6 - lo <= 53 for the
// 64 addends).  Then the order of the additions is free, the sums slide (add the entering value, subtract the leaving one:
// every intermediate is again an exactly representable multiple of 2^lo), and float(S * (1 / 64)) is the float the
// oracle's ordered fp64 sum rounds to.  A table that fails the test (an entry very close to zero next to large ones) keeps
// the ordered kernels (k_cc_coeff8 / k_cc_apply8).
// fp64 column sums are 8 bytes where the integer kernel's are 4, so the maps go through the column-sum region one after
// the other -- (g as fp64, s as u32), then g * s, then g * g: three B / C rounds per channel instead of one, the means of
// a position kept in its thread's registers in between -- and the tile, the LDS footprint (two blocks per CU) and the
// second stage stay as they are.  a = cov / (var + eps) is the IEEE division (a float table has no bound on var).
// ---------------------------------------------------------------------------------------------------------------
static_assert(FG_Y_BYTES >= FG_AH * FG_VP * 12, "fp64 + u32 column-sum planes must fit the row-sum region");

template <int CN>
__global__ __launch_bounds__(FG_NT) void k_cc_fused8f(const unsigned char *__restrict__ img, long long stride, int h, int w,
                                                      const float *__restrict__ glutf, float eps,
                                                      unsigned char *__restrict__ out, long long ostride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int rowb = CN * FG_IW, rawp = (rowb + 3) & ~3;
    constexpr int LUT0 = FG_Y_BYTES + FG_X_BYTES, RAW0 = LUT0 + CN * 1024;
    double *Vd = (double *)smem;                                  // [AH][VP] fp64 column sums of the round's map
    unsigned *Vu = (unsigned *)(smem + FG_AH * FG_VP * 8);        // [AH][VP] column sums of s (round 0)
    double *hs = (double *)smem;                                  // [AH][64 positions][2] in the second stage
    float *a_p = (float *)(smem + FG_Y_BYTES), *b_p = a_p + FG_AH * FG_AP;
    float *lutf = (float *)(smem + LUT0);                         // [CN][256]
    unsigned char *raw = smem + RAW0;                             // [IH][rawp], pixels interleaved as in the image
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * FG_TW, y0 = blockIdx.y * FG_TH;

    for (int i = tid; i < CN * 256; i += FG_NT) lutf[i] = glutf[i];
    if (x0 >= 8 && x0 - 8 + FG_IW <= w && y0 >= 8 && y0 - 8 + FG_IH <= h) {
        constexpr int ndw = rowb >> 2, tail = rowb & 3, per = ndw + (tail ? 1 : 0);
        for (int e = tid; e < FG_IH * per; e += FG_NT) {
            const int i = e / per, d = e - i * per;
            const unsigned char *src = img + (size_t)(y0 - 8 + i) * stride + (size_t)(x0 - 8) * CN + 4 * d;
            if (d < ndw) *(unsigned *)(raw + i * rawp + 4 * d) = *(const __attribute__((address_space(1))) fg_u32_a1_t *)src;
            else for (int t = 0; t < tail; ++t) raw[i * rawp + 4 * d + t] = src[t];
        }
    } else {
        for (int e = tid; e < FG_IH * rowb; e += FG_NT) {
            const int i = e / rowb, r = e - i * rowb, j = r / CN, c = r - j * CN;
            raw[i * rawp + r] = img[(size_t)cc_reflect101(y0 - 8 + i, h) * stride + (size_t)cc_reflect101(x0 - 8 + j, w) * CN + c];
        }
    }
    const bool edge = y0 < 4 || y0 - 4 + FG_AH > h || x0 < 4 || x0 - 4 + FG_AW > w;   // some a / b position lies outside
    const bool bitem = tid < FG_NSEG * FG_IW, citem = tid < FG_AH * 12;
    const int bseg = tid / FG_IW, bcol = tid - bseg * FG_IW, br0 = min(bseg * FG_SEG, FG_AH - FG_SEG);
    const int crow = tid / 12, cq0 = (tid - crow * 12) * 6;
    __syncthreads();

#pragma unroll
    for (int c = 0; c < CN; ++c) {
        const float *lt = lutf + c * 256;
        float gf[FG_SEG + 7], sf[FG_SEG + 7];                     // pass B's column segment: guide and source values
        float mg[6], ms[6], mgs[6];                               // pass C's positions: means of the earlier rounds
#pragma unroll
        for (int round = 0; round < 3; ++round) {
            // ---- B: column sums of the round's map (round 0: g and, as integers, s) ----
            if (bitem) {
                double p[FG_SEG + 7];
                unsigned su = 0;
                if (round == 0) {
                    const unsigned char *rp = raw + br0 * rawp + bcol * CN + c;
                    unsigned sv[FG_SEG + 7];
#pragma unroll
                    for (int k = 0; k < FG_SEG + 7; ++k) {
                        sv[k] = rp[k * rawp];
                        gf[k] = lt[sv[k]];
                        sf[k] = (float)sv[k];
                        p[k] = (double)gf[k];
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) su += sv[k];
#pragma unroll
                    for (int k = 0; k < FG_SEG; ++k) {
                        if (k > 0) su += sv[k + 7] - sv[k - 1];
                        Vu[(br0 + k) * FG_VP + bcol] = su;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < FG_SEG + 7; ++k) {
                        const float pr = round == 1 ? gf[k] * sf[k] : gf[k] * gf[k];   // the reference's float32 products
                        p[k] = (double)pr;
                    }
                }
                double sd = 0.0;
#pragma unroll
                for (int k = 0; k < 8; ++k) sd += p[k];
#pragma unroll
                for (int k = 0; k < FG_SEG; ++k) {
                    if (k > 0) sd += p[k + 7] - p[k - 1];
                    Vd[(br0 + k) * FG_VP + bcol] = sd;
                }
            }
            __syncthreads();
            // ---- C: row sums; after the last round a and b ----
            if (citem) {
                const double *vd = Vd + crow * FG_VP + cq0;
                double q[13], sd[6];
#pragma unroll
                for (int k = 0; k < 13; ++k) q[k] = vd[k];
                sd[0] = 0.0;
#pragma unroll
                for (int k = 0; k < 8; ++k) sd[0] += q[k];
#pragma unroll
                for (int t = 1; t < 6; ++t) sd[t] = sd[t - 1] + (q[t + 7] - q[t - 1]);
                if (round == 0) {
                    const unsigned *vu = Vu + crow * FG_VP + cq0;
                    unsigned u[13], su[6];
#pragma unroll
                    for (int k = 0; k < 13; ++k) u[k] = vu[k];
                    su[0] = 0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) su[0] += u[k];
#pragma unroll
                    for (int t = 1; t < 6; ++t) su[t] = su[t - 1] + (u[t + 7] - u[t - 1]);
#pragma unroll
                    for (int t = 0; t < 6; ++t) {
                        mg[t] = (float)(sd[t] * 0.015625);
                        ms[t] = (float)su[t] * 0.015625f;          // an integer below 2^24: exact
                    }
                } else if (round == 1) {
#pragma unroll
                    for (int t = 0; t < 6; ++t) mgs[t] = (float)(sd[t] * 0.015625);
                } else {
#pragma unroll
                    for (int t = 0; t < 6; ++t) {
                        const float mgg = (float)(sd[t] * 0.015625);
                        const float cov = mgs[t] - mg[t] * ms[t], var = mgg - mg[t] * mg[t];
                        const float a = cov / (var + eps);
                        const float b = ms[t] - a * mg[t];
                        if (cq0 + t < FG_AW) {
                            a_p[crow * FG_AP + cq0 + t] = a;
                            b_p[crow * FG_AP + cq0 + t] = b;
                        }
                    }
                }
            }
            __syncthreads();
        }
        if (edge) {                                               // block-uniform
            for (int e = tid; e < FG_AH * FG_AW; e += FG_NT) {
                const int r = e / FG_AW, q = e - r * FG_AW;
                const int py = y0 - 4 + r, px = x0 - 4 + q;
                if (py < 0 || py >= h || px < 0 || px >= w) {
                    const int sr = cc_reflect101(py, h) - (y0 - 4), sq = cc_reflect101(px, w) - (x0 - 4);
                    if (sr >= 0 && sr < FG_AH && sq >= 0 && sq < FG_AW) {   // else: a position no stored pixel reads
                        a_p[r * FG_AP + q] = a_p[sr * FG_AP + sq];
                        b_p[r * FG_AP + q] = b_p[sr * FG_AP + sq];
                    }
                }
            }
            __syncthreads();
        }
        // ---- D: ordered row sums of a and b (as in k_cc_fused8) ----
        for (int e = tid; e < FG_AH * 16; e += FG_NT) {
            const int py = e >> 4, gx = e & 15;
            const float4 *ar = (const float4 *)(a_p + py * FG_AP + 4 * gx), *br = (const float4 *)(b_p + py * FG_AP + 4 * gx);
            const float4 a0 = ar[0], a1 = ar[1], a2 = ar[2], b0 = br[0], b1 = br[1], b2 = br[2];
            const double v0[11] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z};
            const double v1[11] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z};
            double o0[4], o1[4];
            fg_sums(v0, o0);
            fg_sums(v1, o1);
#pragma unroll
            for (int j = 0; j < 4; ++j) *(double2 *)(hs + ((size_t)py * FG_TW + j * 16 + gx) * 2) = make_double2(o0[j], o1[j]);
        }
        __syncthreads();
        // ---- E: ordered column sums, output ----
        {
            const int gy = tid >> 6, l = tid & 63, ox = 4 * (l & 15) + (l >> 4);
            double v0[11], v1[11];
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const double2 t = *(const double2 *)(hs + ((size_t)(4 * gy + k) * FG_TW + l) * 2);
                v0[k] = t.x;
                v1[k] = t.y;
            }
            double ta[4], tb[4];
            fg_sums(v0, ta);
            fg_sums(v1, tb);
            unsigned char *px = raw + (4 * gy + 8) * rawp + (ox + 8) * CN + c;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float ma = (float)(ta[j] * 0.015625), mb = (float)(tb[j] * 0.015625);
                const float g = lt[px[j * rawp]];
                const float r = ma * g + mb;
                const float cl = r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r);     // k_cc_apply8's clip, also for a NaN
                px[j * rawp] = (unsigned char)cl;
            }
        }
        __syncthreads();                                          // hs becomes the column-sum planes again
    }
    constexpr int tileb = FG_TW * CN;
    if (x0 + FG_TW <= w && y0 + FG_TH <= h) {
        constexpr int ndw = tileb >> 2;
        for (int e = tid; e < FG_TH * ndw; e += FG_NT) {
            const int oy = e / ndw, d = e - oy * ndw;
            *(__attribute__((address_space(1))) fg_u32_a1_t *)(out + (size_t)(y0 + oy) * ostride + (size_t)x0 * CN + 4 * d) =
                *(const unsigned *)(raw + (oy + 8) * rawp + 8 * CN + 4 * d);
        }
    } else {
        const int vw = min(FG_TW, w - x0) * CN, vh = min(FG_TH, h - y0);
        for (int e = tid; e < vh * tileb; e += FG_NT) {
            const int oy = e / tileb, r = e - oy * tileb;
            if (r < vw) out[(size_t)(y0 + oy) * ostride + (size_t)x0 * CN + r] = raw[(oy + 8) * rawp + 8 * CN + r];
        }
    }
}

// What the first stage of the radius-8 guided filter may do with a guide table (host only):
//   1  every entry a whole number 0..255: the integer kernel (k_cc_fused8),
//   2  a float table whose box sums of `terms` addends are exact in fp64 (see k_cc_fused8f): sliding fp64 sums,
//   0  neither: ordered sums (k_cc_coeff8 / k_cc_apply8).
static int cc_table_class(const float *tab, int cn, int terms)
{
    bool whole = true;
    for (int i = 0; i < cn * 256 && whole; ++i) {
        const float v = tab[i];
        whole = v >= 0.0f && v <= 255.0f && v == (float)(int)v;
    }
    if (whole) return 1;
    int tbits = 0;
    while ((1 << tbits) < terms) ++tbits;
    for (int c = 0; c < cn; ++c) {
        for (int m = 0; m < 3; ++m) {
            int lo = INT_MAX, hi = INT_MIN;
            for (int v = 0; v < 256; ++v) {
                const float g = tab[c * 256 + v];
                volatile float x = m == 0 ? g : (m == 1 ? g * (float)v : g * g);      // rounded to float32, as on the device
                const float ax = fabsf(x);
                if (!(ax <= FLT_MAX)) return 0;                                        // inf / NaN
                if (ax == 0.0f) continue;
                const int e = ilogbf(ax);                                              // 2^e <= |x| < 2^(e + 1)
                lo = std::min(lo, std::max(e, -126) - 23);
                hi = std::max(hi, e + 1);
            }
            if (lo != INT_MAX && hi + tbits - lo > 53) return 0;
        }
    }
    return 2;
}


// ---------------------------------------------------------------------------------------------------------------
// The OTHER branch of BlendingModule._guided_filter (blending_module.py:1108-1111): cv2.ximgproc.guidedFilter(guide, src,
// radius, eps), which is what runs when opencv-contrib is installed (requirements.txt:6).  He et al.'s guided filter with a
// (2 r + 1)^2 box window and, for a 3-channel guide, the colour form: per pixel the 3 x 3 covariance of the guide (+ eps on
// the diagonal) is inverted and every source channel gets a 3-vector a and an offset b.  PARITY UNPINNED: restated from the
// published algorithm and OpenCV-contrib's guided_filter.cpp as remembered (float32 planes, box means through
// cv::boxFilter with float64 sums, BORDER_REFLECT, cofactor inverse); oracle_np.guided_filter_ximgproc spells out the same
// expression order and is what the GPU result is compared with.  A correctness-first layout -- one launch per box mean over
// planar float32 temporaries -- not a tuned kernel; the default stays the _simple_guided_filter branch (sr_color_correct_u8
// local_filter = 1).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int gf_reflect(int p, int n)           // BORDER_REFLECT: fedcba|abcdefgh|hgfedcb
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p - 1 : 2 * n - 1 - p;
    return p;
}

// I_c = glut[c][v], p_c = v as float32 planes: planes[c] = I_c, planes[cn + c] = p_c
__global__ __launch_bounds__(256) void k_gf_planes(const unsigned char *__restrict__ img, long long stride, int h, int w, int cn,
                                                   const float *__restrict__ glut, float *__restrict__ planes)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    const size_t plane = (size_t)h * w, o = (size_t)y * w + x;
    for (int c = 0; c < cn; ++c) {
        const int v = img[(size_t)y * stride + (size_t)x * cn + c];
        planes[c * plane + o] = glut[c * 256 + v];
        planes[(cn + c) * plane + o] = (float)v;
    }
}

// dst = boxFilter(A * B  or  A, (R, R), normalize, BORDER_REFLECT), sums in float64: each window row left to right, then the
// row sums top to bottom (cv::boxFilter's order), mean = (float)(sum * scale).  Block = 64 x 16 outputs.
__global__ __launch_bounds__(256) void k_gf_box(const float *__restrict__ A, const float *__restrict__ B, int h, int w, int R,
                                                double scale, float *__restrict__ dst)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int PH = CC_TH + R - 1, PW = CC_TW + R - 1, anchor = R / 2;
    float *pt = (float *)smem;                                            // [PH][PW]
    double *hs = (double *)(smem + (((size_t)PH * PW * sizeof(float) + 15) & ~(size_t)15));   // [PH][TW]
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * CC_TW, y0 = blockIdx.y * CC_TH;
    for (int e = tid; e < PH * PW; e += 256) {
        const int py = e / PW, px = e - py * PW;
        const size_t o = (size_t)gf_reflect(y0 + py - anchor, h) * w + gf_reflect(x0 + px - anchor, w);
        pt[e] = B ? A[o] * B[o] : A[o];
    }
    __syncthreads();
    for (int e = tid; e < PH * CC_TW; e += 256) {
        const int py = e / CC_TW, ox = e - py * CC_TW;
        double sacc = 0.0;
        for (int k = 0; k < R; ++k) sacc += (double)pt[py * PW + ox + k];
        hs[e] = sacc;
    }
    __syncthreads();
    for (int e = tid; e < CC_TH * CC_TW; e += 256) {
        const int oy = e / CC_TW, ox = e - oy * CC_TW;
        const int y = y0 + oy, x = x0 + ox;
        if (y >= h || x >= w) continue;
        double t = 0.0;
        for (int k = 0; k < R; ++k) t += hs[(size_t)(oy + k) * CC_TW + ox];
        dst[(size_t)y * w + x] = (float)(t * scale);
    }
}

// The same box mean at the reference's setting (radius 8: a 17 x 17 window), register-blocked like the 8 x 8 kernels above: a
// block owns 64 x 32 outputs (80 x 48 patch: 1.9 x the outputs instead of 2.5 x at 64 x 16), the row pass forms FOUR neighbouring
// 17-term sums from twenty values read with 16-byte LDS loads, the column pass EIGHT from twenty-four row sums; every sum still
// adds its terms in cv::boxFilter's order (left to right, then top to bottom, in fp64).  Row sums of output column 4 gx + j sit at
// position j * 16 + gx of their row (conflict-free stores, lane l of the column pass owns column 4 (l % 16) + l / 16).
// 200 MP plane: 1.42 -> see DESIGN.md.
#define GB_TW 64
#define GB_TH 32
#define GB_R 17
#define GB_PH (GB_TH + GB_R - 1)          /* 48 */
#define GB_PW (GB_TW + GB_R - 1)          /* 80 */

template <int NOUT, int NIN>
__device__ __forceinline__ void gb_sums(const double (&v)[NIN], double (&o)[NOUT])
{
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < GB_R; ++k) t += v[j + k];
        o[j] = t;
    }
}

__global__ __launch_bounds__(256) void k_gf_box17(const float *__restrict__ A, const float *__restrict__ B, int h, int w, double scale,
                                                  float *__restrict__ dst)
{
    __shared__ __attribute__((aligned(16))) float pt[GB_PH * GB_PW];                 // 15 KB
    __shared__ __attribute__((aligned(16))) double hs[GB_PH * GB_TW];                // 24 KB
    const int tid = threadIdx.x, anchor = GB_R / 2;
    const int x0 = blockIdx.x * GB_TW, y0 = blockIdx.y * GB_TH;
    if ((w & 3) == 0 && x0 >= anchor && x0 - anchor + GB_PW <= w && y0 >= anchor && y0 - anchor + GB_PH <= h) {
        // interior: 16-byte loads (x0 - 8 is a multiple of 4 and so is w: every row segment is 16-byte aligned)
        for (int e = tid; e < GB_PH * (GB_PW / 4); e += 256) {
            const int py = e / (GB_PW / 4), q = e - py * (GB_PW / 4);
            const size_t o = (size_t)(y0 + py - anchor) * w + (x0 - anchor) + 4 * q;
            float4 v = *(const float4 *)(A + o);
            if (B) {
                const float4 u = *(const float4 *)(B + o);
                v.x *= u.x; v.y *= u.y; v.z *= u.z; v.w *= u.w;
            }
            *(float4 *)(pt + py * GB_PW + 4 * q) = v;
        }
    } else {
        for (int e = tid; e < GB_PH * GB_PW; e += 256) {
            const int py = e / GB_PW, px = e - py * GB_PW;
            const size_t o = (size_t)gf_reflect(y0 + py - anchor, h) * w + gf_reflect(x0 + px - anchor, w);
            pt[e] = B ? A[o] * B[o] : A[o];
        }
    }
    __syncthreads();
    for (int e = tid; e < GB_PH * 16; e += 256) {
        const int py = e >> 4, gx = e & 15;
        const float4 *r = (const float4 *)(pt + py * GB_PW + 4 * gx);
        const float4 q0 = r[0], q1 = r[1], q2 = r[2], q3 = r[3], q4 = r[4];
        const double v[20] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w, q4.x, q4.y, q4.z, q4.w};
        double o[4];
        gb_sums<4, 20>(v, o);
#pragma unroll
        for (int j = 0; j < 4; ++j) hs[py * GB_TW + j * 16 + gx] = o[j];
    }
    __syncthreads();
    {
        const int l = tid & 63, g = tid >> 6, ox = 4 * (l & 15) + (l >> 4);         // rows 8 g .. 8 g + 7 of column ox
        double v[24], o[8];
#pragma unroll
        for (int k = 0; k < 24; ++k) v[k] = hs[(8 * g + k) * GB_TW + l];
        gb_sums<8, 24>(v, o);
        const int x = x0 + ox;
        if (x < w) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int y = y0 + 8 * g + j;
                if (y < h) dst[(size_t)y * w + x] = (float)(o[j] * scale);
            }
        }
    }
}

// First stage of the colour guided filter in ONE launch, for an integer-valued guide table (histogram matching, 'none'): the
// 21 box means (m_i, m_ij, mp_c, mIp_ic) are means of integers whose 289-term sums stay below 2^25 -- exact in any order -- so
// they run as 32-bit sliding sums straight from the u8 image and (float)((double)S * (1 / 289)) is the float cv::boxFilter's
// ordered fp64 sum rounds to.  A block owns 64 x 16 positions; its (64 + 16) x (16 + 16) window sits in LDS as six byte
// planes (I_0 I_1 I_2 p_0 p_1 p_2, BORDER_REFLECT).  The maps are taken in five groups of at most five (LDS: 5 column-sum
// planes): pass V -- item = (column, map): the 32 products of its column, sliding 17-row sums; pass H -- a thread owns four
// neighbouring positions of one row: sliding 17-column sums.  After the two covariance groups the thread inverts its four
// 3 x 3 matrices (k_gf_coeff's expressions in k_gf_coeff's order); after each source channel's group it writes a_c0..2, b_c.
// The 21 mean planes, the 6 float planes and 21 box launches + k_gf_coeff (20 ms of the 30) are gone.
#define GC_TW 64
#define GC_TH 16
#define GC_WW (GC_TW + 16)                 /* window columns */
#define GC_WH (GC_TH + 16)                 /* window rows */
#define GC_NG 5                            /* maps per group */

// map m of the 21: which two of the six window bytes multiply (second = -1: the byte itself)
__device__ __forceinline__ void gc_map_bytes(int m, int &b0, int &b1)
{
    // [0..2] I_i   [3..8] I_i I_j (00 01 02 11 12 22)   [9..11] p_c   [12..20] I_i p_c (c * 3 + i)
    if (m < 3) { b0 = m; b1 = -1; }
    else if (m < 9) {
        const int k = m - 3;
        b0 = k < 3 ? 0 : (k < 5 ? 1 : 2);
        b1 = k < 3 ? k : (k < 5 ? k - 2 : 2);
    } else if (m < 12) { b0 = 3 + (m - 9); b1 = -1; }
    else { const int k = m - 12; b0 = k % 3; b1 = 3 + k / 3; }
}

#ifndef GC_MINB
#define GC_MINB 1
#endif
__global__ __launch_bounds__(256, GC_MINB) void k_gfx_coeff17(const unsigned char *__restrict__ img, long long stride, int h, int w,
                                                     const unsigned char *__restrict__ glutb, float eps, double scale,
                                                     float *__restrict__ ab, size_t plane)
{
    __shared__ __attribute__((aligned(16))) unsigned char win[GC_WH * GC_WW * 6];          // 15 KB
    __shared__ __attribute__((aligned(16))) unsigned V[GC_NG][GC_TH][GC_WW];               // 25 KB: 40 KB in all, four blocks per CU
    unsigned char *lutb = (unsigned char *)&V[0][0][0];            // the table is only needed while the window is filled
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * GC_TW, y0 = blockIdx.y * GC_TH;
    for (int i = tid; i < 768; i += 256) lutb[i] = glutb[i];
    __syncthreads();
    for (int e = tid; e < GC_WH * GC_WW; e += 256) {
        const int py = e / GC_WW, px = e - py * GC_WW;
        const unsigned char *s = img + (size_t)gf_reflect(y0 + py - 8, h) * stride + (size_t)gf_reflect(x0 + px - 8, w) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {                              // six byte PLANES: the lanes of pass V read consecutive bytes
            const unsigned char v = s[c];                          // (interleaved pixels: 1.0 G bank-conflict cycles per image)
            win[c * (GC_WH * GC_WW) + e] = lutb[c * 256 + v];
            win[(3 + c) * (GC_WH * GC_WW) + e] = v;
        }
    }
    const int r = tid >> 4, q = tid & 15;                          // pass H / the algebra: row r, columns 4 q .. 4 q + 3
    float mean[4][9];                                              // m_0..2, then the six second moments (later: the inverse)
    float inv[4][6];
    // groups: 0 = maps 0..4, 1 = maps 5..8, 2 + c = maps 9 + c, 12 + 3 c .. 14 + 3 c
#pragma unroll
    for (int grp = 0; grp < 5; ++grp) {
        const int nmap = grp == 0 ? 5 : 4;
        __syncthreads();                                           // window ready / the previous group's V consumed
#pragma unroll 1
        for (int e = tid; e < nmap * GC_WW; e += 256) {
            const int k = e / GC_WW, col = e - k * GC_WW;
            const int m = grp == 0 ? k : (grp == 1 ? 5 + k : (k == 0 ? 9 + (grp - 2) : 12 + 3 * (grp - 2) + (k - 1)));
            int b0, b1;
            gc_map_bytes(m, b0, b1);
            const unsigned char *cp0 = win + b0 * (GC_WH * GC_WW) + col, *cp1 = win + max(b1, 0) * (GC_WH * GC_WW) + col;
            unsigned pr[GC_WH];
#pragma unroll
            for (int i = 0; i < GC_WH; ++i) {
                const unsigned a = cp0[i * GC_WW];
                const unsigned b = b1 >= 0 ? (unsigned)cp1[i * GC_WW] : 1u;
                // opaque product: hipcc 7.2 folds sums of byte products into v_perm_b32 + v_dot4_u32_u8 and gets them wrong
                asm("v_mul_u32_u24_e32 %0, %1, %2" : "=v"(pr[i]) : "v"(a), "v"(b));
                if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // eight rows' bytes in flight, not all 32 (registers)
            }
            unsigned sacc = 0;
#pragma unroll
            for (int i = 0; i < 17; ++i) sacc += pr[i];
#pragma unroll
            for (int o = 0; o < GC_TH; ++o) {
                if (o > 0) sacc += pr[o + 16] - pr[o - 1];
                V[k][o][col] = sacc;
            }
        }
        __syncthreads();
        float val[GC_NG][4];
#pragma unroll
        for (int k = 0; k < GC_NG; ++k) {
            if (k < nmap) {
                const uint4 *vp = (const uint4 *)&V[k][r][4 * q];
                const uint4 u0 = vp[0], u1 = vp[1], u2 = vp[2], u3 = vp[3], u4 = vp[4];
                const unsigned t[20] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w, u2.x, u2.y, u2.z, u2.w, u3.x, u3.y, u3.z, u3.w,
                                        u4.x, u4.y, u4.z, u4.w};
                unsigned sacc = 0;
#pragma unroll
                for (int i = 0; i < 17; ++i) sacc += t[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j > 0) sacc += t[j + 16] - t[j - 1];
                    val[k][j] = (float)((double)sacc * scale);
                    // pinned here: hipcc otherwise sinks the sums to their first use -- below the next group's passes -- and keeps
                    // the hundred values loaded from V alive meanwhile (225 VGPRs, two waves per SIMD)
                    asm volatile("" : "+v"(val[k][j]));
                }
                __builtin_amdgcn_sched_barrier(0);                 // one map's twenty values at a time (registers)
            }
        }
        if (grp == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < 5; ++k) mean[j][k] = val[k][j];
        } else if (grp == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int k = 0; k < 4; ++k) mean[j][5 + k] = val[k][j];
                const float m0 = mean[j][0], m1 = mean[j][1], m2 = mean[j][2];
                const float c00 = (mean[j][3] - m0 * m0) + eps, c01 = mean[j][4] - m0 * m1, c02 = mean[j][5] - m0 * m2;
                const float c11 = (mean[j][6] - m1 * m1) + eps, c12 = mean[j][7] - m1 * m2, c22 = (mean[j][8] - m2 * m2) + eps;
                const float A00 = c11 * c22 - c12 * c12, A01 = c02 * c12 - c01 * c22, A02 = c01 * c12 - c02 * c11;
                const float A11 = c00 * c22 - c02 * c02, A12 = c01 * c02 - c00 * c12, A22 = c00 * c11 - c01 * c01;
                const float det = (c00 * A00 + c01 * A01) + c02 * A02;
                inv[j][0] = A00 / det; inv[j][1] = A01 / det; inv[j][2] = A02 / det;
                inv[j][3] = A11 / det; inv[j][4] = A12 / det; inv[j][5] = A22 / det;
                __builtin_amdgcn_sched_barrier(0);                 // one matrix at a time (registers)
            }
        } else {
            const int c = grp - 2;
            float4 o0, o1, o2, ob;
            float a0v[4], a1v[4], a2v[4], bv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float m0 = mean[j][0], m1 = mean[j][1], m2 = mean[j][2];
                const float mp = val[0][j];
                const float v0 = val[1][j] - m0 * mp, v1 = val[2][j] - m1 * mp, v2 = val[3][j] - m2 * mp;
                const float i00 = inv[j][0], i01 = inv[j][1], i02 = inv[j][2], i11 = inv[j][3], i12 = inv[j][4], i22 = inv[j][5];
                a0v[j] = (i00 * v0 + i01 * v1) + i02 * v2;
                a1v[j] = (i01 * v0 + i11 * v1) + i12 * v2;
                a2v[j] = (i02 * v0 + i12 * v1) + i22 * v2;
                bv[j] = mp - ((a0v[j] * m0 + a1v[j] * m1) + a2v[j] * m2);
            }
            o0.x = a0v[0]; o0.y = a0v[1]; o0.z = a0v[2]; o0.w = a0v[3];
            o1.x = a1v[0]; o1.y = a1v[1]; o1.z = a1v[2]; o1.w = a1v[3];
            o2.x = a2v[0]; o2.y = a2v[1]; o2.z = a2v[2]; o2.w = a2v[3];
            ob.x = bv[0]; ob.y = bv[1]; ob.z = bv[2]; ob.w = bv[3];
            const int y = y0 + r, x = x0 + 4 * q;
            if (y < h && x < w) {
                const size_t o = (size_t)y * w + x;
                float *p0 = ab + (size_t)(3 * c) * plane + o, *p1 = p0 + plane, *p2 = p1 + plane, *pb = ab + (size_t)(9 + c) * plane + o;
                if (x + 3 < w && (w & 3) == 0) {
                    *(float4 *)p0 = o0; *(float4 *)p1 = o1; *(float4 *)p2 = o2; *(float4 *)pb = ob;
                } else {
                    for (int j = 0; j < 4 && x + j < w; ++j) { p0[j] = a0v[j]; p1[j] = a1v[j]; p2[j] = a2v[j]; pb[j] = bv[j]; }
                }
            }
        }
    }
}

// Second stage of the colour guided filter for one source channel in ONE launch: the 17 x 17 box means of a_0, a_1, a_2 and b
// (the planes ab[first], ab[first + 1], ab[first + 2], ab[bplane]) formed one after the other with k_gf_box17's passes, the
// eight means per map of a thread kept in registers, then q = ((mean_a0 * I_0 + mean_a1 * I_1) + mean_a2 * I_2) + mean_b ->
// clip -> truncate.  Replaces four box launches + a quarter of k_gf_out: the twelve mean planes are never written.
__global__ __launch_bounds__(256) void k_gf_box17_out(const float *__restrict__ ab, const float *__restrict__ planes, size_t plane,
                                                      int first, int bplane, int h, int w, double scale, int c,
                                                      unsigned char *__restrict__ out, long long ostride,
                                                      const unsigned char *__restrict__ img, long long stride,
                                                      const float *__restrict__ glut)
{
    // the guide values I_m come from the float planes, or (planes == nullptr: the fused first stage never made them) from the
    // image through the guide table -- the same floats
    __shared__ __attribute__((aligned(16))) float pt[GB_PH * GB_PW];
    __shared__ __attribute__((aligned(16))) double hs[GB_PH * GB_TW];
    const int tid = threadIdx.x, anchor = GB_R / 2;
    const int x0 = blockIdx.x * GB_TW, y0 = blockIdx.y * GB_TH;
    const bool inner = (w & 3) == 0 && x0 >= anchor && x0 - anchor + GB_PW <= w && y0 >= anchor && y0 - anchor + GB_PH <= h;
    const int l = tid & 63, g = tid >> 6, ox = 4 * (l & 15) + (l >> 4);
    const int x = x0 + ox;
    float acc[8];                                                 // the output expression, built left to right as the means arrive
#pragma unroll 1
    for (int m = 0; m < 4; ++m) {
        const float *A = ab + (size_t)(m < 3 ? first + m : bplane) * plane;
        if (inner) {
            for (int e = tid; e < GB_PH * (GB_PW / 4); e += 256) {
                const int py = e / (GB_PW / 4), q = e - py * (GB_PW / 4);
                *(float4 *)(pt + py * GB_PW + 4 * q) = *(const float4 *)(A + (size_t)(y0 + py - anchor) * w + (x0 - anchor) + 4 * q);
            }
        } else {
            for (int e = tid; e < GB_PH * GB_PW; e += 256) {
                const int py = e / GB_PW, px = e - py * GB_PW;
                pt[e] = A[(size_t)gf_reflect(y0 + py - anchor, h) * w + gf_reflect(x0 + px - anchor, w)];
            }
        }
        __syncthreads();
        for (int e = tid; e < GB_PH * 16; e += 256) {
            const int py = e >> 4, gx = e & 15;
            const float4 *r = (const float4 *)(pt + py * GB_PW + 4 * gx);
            const float4 q0 = r[0], q1 = r[1], q2 = r[2], q3 = r[3], q4 = r[4];
            const double v[20] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w, q4.x, q4.y, q4.z, q4.w};
            double o[4];
            gb_sums<4, 20>(v, o);
#pragma unroll
            for (int j = 0; j < 4; ++j) hs[py * GB_TW + j * 16 + gx] = o[j];
        }
        __syncthreads();
        {
            double v[24], o[8];
#pragma unroll
            for (int k = 0; k < 24; ++k) v[k] = hs[(8 * g + k) * GB_TW + l];
            gb_sums<8, 24>(v, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float mean = (float)(o[j] * scale);
                const int y = min(y0 + 8 * g + j, h - 1);
                const int xc = min(x, w - 1);
                const float gi = m >= 3 ? 0.0f : planes ? planes[(size_t)m * plane + (size_t)y * w + xc]
                                                        : glut[m * 256 + img[(size_t)y * stride + (size_t)xc * 3 + m]];
                acc[j] = m == 0 ? mean * gi : (m < 3 ? acc[j] + mean * gi : acc[j] + mean);
            }
        }
        __syncthreads();                                          // pt / hs are free for the next map
    }
    if (x >= w) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int y = y0 + 8 * g + j;
        if (y >= h) break;
        const float r = acc[j];
        const float cl = r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r);
        out[(size_t)y * ostride + (size_t)x * 3 + c] = (unsigned char)cl;
    }
}

// per pixel: the linear coefficients.  means: [0..2] mI, [3..8] mII (00 01 02 11 12 22), [9..11] mp, [12..20] mIp (c * 3 + i).
// out: [c * 3 + i] a_ci, [9 + c] b_c.  cn == 1: means [0] mI, [1] mII, [2] mp, [3] mIp -> out [0] a, [1] b.
__global__ __launch_bounds__(256) void k_gf_coeff(const float *__restrict__ means, size_t plane, long long n, int cn, float eps,
                                                  float *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    auto M = [&](int m) { return means[(size_t)m * plane + i]; };
    if (cn == 1) {
        const float mI = M(0), mp = M(2);
        const float var = (M(1) - mI * mI) + eps, cov = M(3) - mI * mp;
        const float a = cov / var;
        out[i] = a;
        out[plane + i] = mp - a * mI;
        return;
    }
    const float m0 = M(0), m1 = M(1), m2 = M(2);
    const float c00 = (M(3) - m0 * m0) + eps, c01 = M(4) - m0 * m1, c02 = M(5) - m0 * m2;
    const float c11 = (M(6) - m1 * m1) + eps, c12 = M(7) - m1 * m2, c22 = (M(8) - m2 * m2) + eps;
    const float A00 = c11 * c22 - c12 * c12, A01 = c02 * c12 - c01 * c22, A02 = c01 * c12 - c02 * c11;
    const float A11 = c00 * c22 - c02 * c02, A12 = c01 * c02 - c00 * c12, A22 = c00 * c11 - c01 * c01;
    const float det = (c00 * A00 + c01 * A01) + c02 * A02;
    const float i00 = A00 / det, i01 = A01 / det, i02 = A02 / det, i11 = A11 / det, i12 = A12 / det, i22 = A22 / det;
    for (int c = 0; c < 3; ++c) {
        const float mp = M(9 + c);
        const float v0 = M(12 + c * 3) - m0 * mp, v1 = M(13 + c * 3) - m1 * mp, v2 = M(14 + c * 3) - m2 * mp;
        const float a0 = (i00 * v0 + i01 * v1) + i02 * v2;
        const float a1 = (i01 * v0 + i11 * v1) + i12 * v2;
        const float a2 = (i02 * v0 + i12 * v1) + i22 * v2;
        out[(size_t)(c * 3) * plane + i] = a0;
        out[(size_t)(c * 3 + 1) * plane + i] = a1;
        out[(size_t)(c * 3 + 2) * plane + i] = a2;
        out[(size_t)(9 + c) * plane + i] = mp - ((a0 * m0 + a1 * m1) + a2 * m2);
    }
}

// q_c = (mean_a_c . I) + mean_b_c -> clip -> truncate (blending_module.py:1017)
__global__ __launch_bounds__(256) void k_gf_out(const float *__restrict__ planes, const float *__restrict__ mab, size_t plane, int h,
                                                int w, int cn, unsigned char *__restrict__ out, long long ostride)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    const size_t o = (size_t)y * w + x;
    for (int c = 0; c < cn; ++c) {
        float r;
        if (cn == 1) r = mab[o] * planes[o] + mab[plane + o];
        else
            r = ((mab[(size_t)(c * 3) * plane + o] * planes[o] + mab[(size_t)(c * 3 + 1) * plane + o] * planes[plane + o]) +
                 mab[(size_t)(c * 3 + 2) * plane + o] * planes[2 * plane + o]) + mab[(size_t)(9 + c) * plane + o];
        const float cl = r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r);
        out[(size_t)y * ostride + (size_t)x * cn + c] = (unsigned char)cl;
    }
}

// no local filter: out = u8(clip(glut[c][v], 0, 255))
__global__ __launch_bounds__(256) void k_cc_map(const unsigned char *__restrict__ img, long long stride, int h, long long rowlen,
                                                int cn, const float *__restrict__ glut, unsigned char *__restrict__ out,
                                                long long ostride)
{
    __shared__ unsigned char tab[4 * 256];
    for (int i = threadIdx.x; i < cn * 256; i += 256) {
        const float r = glut[i];
        const float cl = r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r);
        tab[i] = (unsigned char)cl;
    }
    __syncthreads();
    const long long ngroups = rowlen / 12;                     // 12 bytes per thread and step: byte j of a group is channel j % cn
    int ch[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) ch[j] = (j % cn) * 256;
    for (int y = blockIdx.y; y < h; y += gridDim.y) {
        const unsigned char *row = img + (size_t)y * stride;
        unsigned char *orow = out + (size_t)y * ostride;
        for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < ngroups; g += (long long)gridDim.x * 256) {
            const adj_u3_t q = *(const __attribute__((address_space(1))) adj_u3_a1_t *)(row + g * 12);
            const unsigned wd[3] = {q.x, q.y, q.z};
            unsigned o[3] = {0u, 0u, 0u};
#pragma unroll
            for (int j = 0; j < 12; ++j) o[j >> 2] |= (unsigned)tab[ch[j] + ((wd[j >> 2] >> (8 * (j & 3))) & 0xFFu)] << (8 * (j & 3));
            adj_u3_t v;
            v.x = o[0]; v.y = o[1]; v.z = o[2];
            *(__attribute__((address_space(1))) adj_u3_a1_t *)(orow + g * 12) = v;
        }
        if (blockIdx.x == 0)
            for (long long i = ngroups * 12 + threadIdx.x; i < rowlen; i += 256) orow[i] = tab[(int)(i % cn) * 256 + row[i]];
    }
}

}  // namespace

extern "C" {

int sr_histogram_u8(sr_ctx *ctx, const uint8_t *d_img, int64_t stride, int h, int w, int cn, uint64_t *h_hist)
{
    CTX_ENTER(ctx);
    if (!d_img || !h_hist || h < 1 || w < 1 || cn < 1 || cn > 4) return sr_set_error(SR_ERR_INVALID_ARG, "sr_histogram_u8: bad arguments");
    if (stride < (int64_t)w * cn) return sr_set_error(SR_ERR_SHAPE, "sr_histogram_u8: stride smaller than a row");
    void *scr = nullptr;
    int rc = ctx_scratch(ctx, 4 * 256 * sizeof(unsigned long long), &scr);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(scr, 0, 4 * 256 * sizeof(unsigned long long), ctx->stream));
    {
        ProfScope ps(ctx, "histogram");
        const long long rowlen = (long long)w * cn;
        dim3 grid((unsigned)std::min<long long>((rowlen + 255) / 256, 32), (unsigned)std::min(h, 512));
        hipLaunchKernelGGL(k_hist_u8, grid, dim3(256), 0, ctx->stream, d_img, (long long)stride, h, w, cn, (unsigned long long *)scr);
    }
    rc = check_launch("histogram");
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(h_hist, scr, (size_t)cn * 256 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(stream_sync(ctx));
    return SR_OK;
}

int sr_gray_moments_u8(sr_ctx *ctx, const uint8_t *d_tiles, int n, int64_t tile_bytes, int64_t stride, int h, int w,
                       int gray_shift, int swap_rb, uint64_t *h_sums)
{
    CTX_ENTER(ctx);
    if (!d_tiles || !h_sums || n < 0 || h < 1 || w < 1 || (gray_shift != 14 && gray_shift != 15))
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_gray_moments_u8: bad arguments");
    if (stride < (int64_t)w * 3) return sr_set_error(SR_ERR_SHAPE, "sr_gray_moments_u8: stride smaller than a row");
    if (n == 0) return SR_OK;
    void *scr = nullptr;
    int rc = ctx_scratch(ctx, (size_t)n * 2 * sizeof(unsigned long long), &scr);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(scr, 0, (size_t)n * 2 * sizeof(unsigned long long), ctx->stream));
    {
        ProfScope ps(ctx, "gray_moments");
        // four pixels per thread and step, one pair of atomics per block
        dim3 grid((unsigned)std::max(std::min((w / 4 + 255) / 256, 8), 1), (unsigned)std::min(h, 128), (unsigned)n);
        hipLaunchKernelGGL(k_gray_moments, grid, dim3(256), 0, ctx->stream, d_tiles, (long long)tile_bytes, (long long)stride, h, w,
                           gray_shift, swap_rb ? 1 : 0, (unsigned long long *)scr);
    }
    rc = check_launch("gray_moments");
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(h_sums, scr, (size_t)n * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(stream_sync(ctx));
    return SR_OK;
}

int sr_color_table_class(const float *h_glut, int cn, int terms, int *cls)
{
    if (!h_glut || !cls || cn < 1 || cn > 4 || terms < 1 || terms > 4096)
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_color_table_class: bad arguments");
    *cls = cc_table_class(h_glut, cn, terms);
    return SR_OK;
}

int sr_color_correct_u8(sr_ctx *ctx, const uint8_t *d_img, int64_t stride, int h, int w, int cn, const float *h_glut,
                        int local_filter, int radius, float eps, uint8_t *d_out, int64_t out_stride)
{
    CTX_ENTER(ctx);
    if (!d_img || !h_glut || !d_out || h < 1 || w < 1 || cn < 1 || cn > 4) return sr_set_error(SR_ERR_INVALID_ARG, "sr_color_correct_u8: bad arguments");
    if (stride < (int64_t)w * cn || out_stride < (int64_t)w * cn) return sr_set_error(SR_ERR_SHAPE, "sr_color_correct_u8: stride smaller than a row");
    if (local_filter && (radius < 1 || radius > 16)) return sr_set_error(SR_ERR_INVALID_ARG, "sr_color_correct_u8: radius must be 1..16");
    const size_t npx = (size_t)h * w * cn;
    const size_t tab_bytes = 4 * 256 * sizeof(float);
    // the fused kernels read a block's input halo while other blocks store their output: never in place
    const bool in_place = !(d_out + (size_t)h * out_stride <= d_img || d_img + (size_t)h * stride <= d_out);
    void *scr = nullptr;
    int rc = ctx_scratch(ctx, tab_bytes + 1024 + 256, &scr);
    if (rc) return rc;
    float *d_glut = (float *)scr;
    HIPCHK(upload_small(ctx, d_glut, h_glut, (size_t)cn * 256 * sizeof(float)));
    if (!local_filter) {
        ProfScope ps(ctx, "color_map");
        const long long rowlen = (long long)w * cn;
        dim3 grid((unsigned)std::min<long long>((rowlen + 255) / 256, 64), (unsigned)std::min(h, 1024));
        hipLaunchKernelGGL(k_cc_map, grid, dim3(256), 0, ctx->stream, d_img, (long long)stride, h, rowlen, cn, (const float *)d_glut,
                           d_out, (long long)out_stride);
        return check_launch("color_map");
    }
    if (local_filter == 2) {
        // cv2.ximgproc.guidedFilter semantics (parity unpinned; see k_gf_*): window 2 r + 1, colour guide
        if (cn != 1 && cn != 3) return sr_set_error(SR_ERR_INVALID_ARG, "sr_color_correct_u8: the ximgproc guided filter takes 1 or 3 channels");
        const int R = 2 * radius + 1, nmean = cn == 3 ? 21 : 4, nab = cn == 3 ? 12 : 2;
        const size_t plane = (size_t)h * w;
        if (cn == 3 && R == GB_R && !in_place && !env_flag_off("SR_GF_FUSED")) {
            // the reference's setting with an integer-valued guide table: first stage in one launch (exact 32-bit sliding sums),
            // second stage + output in one launch per channel; the only temporaries are the twelve a / b planes
            unsigned char tabb[3 * 256];
            bool whole = true;
            for (int i = 0; i < 3 * 256 && whole; ++i) {
                const float v = h_glut[i];
                whole = v >= 0.0f && v <= 255.0f && v == (float)(int)v;
                tabb[i] = (unsigned char)(whole ? (int)v : 0);
            }
            if (whole) {
                unsigned char *d_glutb = (unsigned char *)scr + tab_bytes;
                HIPCHK(upload_small(ctx, d_glutb, tabb, sizeof(tabb)));
                float *ab2 = nullptr;
                hipError_t e = hipMalloc((void **)&ab2, (size_t)12 * plane * sizeof(float));
                if (e != hipSuccess)
                    return sr_set_error(e == hipErrorOutOfMemory ? SR_ERR_OOM : SR_ERR_HIP, "sr_color_correct_u8: %s", hipGetErrorString(e));
                const double sc = 1.0 / ((double)R * (double)R);
                {
                    ProfScope ps(ctx, "guided_ximgproc");
                    hipLaunchKernelGGL(k_gfx_coeff17, dim3((w + GC_TW - 1) / GC_TW, (h + GC_TH - 1) / GC_TH), dim3(256), 0, ctx->stream,
                                       d_img, (long long)stride, h, w, (const unsigned char *)d_glutb, eps, sc, ab2, plane);
                    const dim3 g17((w + GB_TW - 1) / GB_TW, (h + GB_TH - 1) / GB_TH);
                    for (int c = 0; c < 3; ++c)
                        hipLaunchKernelGGL(k_gf_box17_out, g17, dim3(256), 0, ctx->stream, (const float *)ab2, (const float *)nullptr, plane,
                                           3 * c, 9 + c, h, w, sc, c, d_out, (long long)out_stride, d_img, (long long)stride,
                                           (const float *)d_glut);
                }
                int rcf = check_launch("guided_ximgproc");
                hipError_t esf = stream_sync(ctx);
                (void)hipFree(ab2);
                if (rcf) return rcf;
                if (esf != hipSuccess) return sr_set_error(SR_ERR_HIP, "sr_color_correct_u8: %s", hipGetErrorString(esf));
                return SR_OK;
            }
        }
        float *buf = nullptr;
        {
            // (the fused second stage of the 3-channel radius-8 case never writes the mean planes of a / b)
            const int nmab = (cn == 3 && R == GB_R) ? 0 : nab;
            hipError_t e = hipMalloc((void **)&buf, (size_t)(2 * cn + nmean + nab + nmab) * plane * sizeof(float));
            if (e != hipSuccess)
                return sr_set_error(e == hipErrorOutOfMemory ? SR_ERR_OOM : SR_ERR_HIP, "sr_color_correct_u8: %s", hipGetErrorString(e));
        }
        float *planes = buf, *means = planes + 2 * cn * plane, *ab = means + (size_t)nmean * plane, *mab = ab + (size_t)nab * plane;
        const int PH = CC_TH + R - 1, PW = CC_TW + R - 1;
        const size_t lds = ((((size_t)PH * PW * sizeof(float)) + 15) & ~(size_t)15) + (size_t)PH * CC_TW * sizeof(double);
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)k_gf_box, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const double scale = 1.0 / ((double)R * (double)R);
        const dim3 g4((w + 63) / 64, (h + 3) / 4), b4(64, 4), gt((w + CC_TW - 1) / CC_TW, (h + CC_TH - 1) / CC_TH);
        {
            ProfScope ps(ctx, "guided_ximgproc");
            hipLaunchKernelGGL(k_gf_planes, g4, b4, 0, ctx->stream, d_img, (long long)stride, h, w, cn, (const float *)d_glut, planes);
            const dim3 gt17((w + GB_TW - 1) / GB_TW, (h + GB_TH - 1) / GB_TH);
            auto box = [&](const float *A, const float *B, float *dst) {
                if (R == GB_R) hipLaunchKernelGGL(k_gf_box17, gt17, dim3(256), 0, ctx->stream, A, B, h, w, scale, dst);
                else hipLaunchKernelGGL(k_gf_box, gt, dim3(256), lds, ctx->stream, A, B, h, w, R, scale, dst);
            };
            auto I = [&](int c) { return planes + (size_t)c * plane; };
            auto Pp = [&](int c) { return planes + (size_t)(cn + c) * plane; };
            int m = 0;
            if (cn == 1) {
                box(I(0), nullptr, means);
                box(I(0), I(0), means + plane);
                box(Pp(0), nullptr, means + 2 * plane);
                box(I(0), Pp(0), means + 3 * plane);
            } else {
                for (int c = 0; c < 3; ++c) box(I(c), nullptr, means + (size_t)(m++) * plane);
                for (int a = 0; a < 3; ++a)
                    for (int b2 = a; b2 < 3; ++b2) box(I(a), I(b2), means + (size_t)(m++) * plane);
                for (int c = 0; c < 3; ++c) box(Pp(c), nullptr, means + (size_t)(m++) * plane);
                for (int c = 0; c < 3; ++c)
                    for (int a = 0; a < 3; ++a) box(I(a), Pp(c), means + (size_t)(m++) * plane);
            }
            hipLaunchKernelGGL(k_gf_coeff, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, ctx->stream, (const float *)means, plane,
                               (long long)plane, cn, eps, ab);
            if (cn == 3 && R == GB_R) {                            // the reference's setting: second stage + output per channel
                for (int c = 0; c < 3; ++c)
                    hipLaunchKernelGGL(k_gf_box17_out, gt17, dim3(256), 0, ctx->stream, (const float *)ab, (const float *)planes, plane,
                                       3 * c, 9 + c, h, w, scale, c, d_out, (long long)out_stride, (const unsigned char *)nullptr, 0ll,
                                       (const float *)nullptr);
            } else {
                for (int k = 0; k < nab; ++k) box(ab + (size_t)k * plane, nullptr, mab + (size_t)k * plane);
                hipLaunchKernelGGL(k_gf_out, g4, b4, 0, ctx->stream, (const float *)planes, (const float *)mab, plane, h, w, cn, d_out,
                                   (long long)out_stride);
            }
        }
        int rc2 = check_launch("guided_ximgproc");
        hipError_t es2 = stream_sync(ctx);
        (void)hipFree(buf);
        if (rc2) return rc2;
        if (es2 != hipSuccess) return sr_set_error(SR_ERR_HIP, "sr_color_correct_u8: %s", hipGetErrorString(es2));
        return SR_OK;
    }
    // (the fused kernel's division leaves out v_div_scale / v_div_fixup, the identity while var + eps stays in [0.006, 65026]:
    // true for the reference's eps = 0.01, not for an arbitrary one -- a smaller eps takes the pass-structured kernels and
    // their IEEE division, so results never depend on which kernels run)
    if (radius == CC8_R && h >= 16 && w >= 16 && !in_place && !env_flag_off("SR_CC_FUSED")) {
        // the reference's radius with a table whose first-stage sums may slide: one fused kernel
        //   class 1 (histogram matching, method 'none': whole numbers) -> k_cc_fused8, exact 32-bit sums;
        //   class 2 (a mean_std table whose fp64 box sums are exact)    -> k_cc_fused8f, exact fp64 sums, IEEE division
        // (an integer table with eps < 0.005 is exact as a float table too and takes the second kernel)
        int cls = cc_table_class(h_glut, cn, CC8_R * CC8_R);
        if (cls == 1 && eps < 0.005f) cls = 2;
        if (cls == 2 && env_flag_off("SR_CC_FUSED_F")) cls = 0;
        if (cls == 1) {
            unsigned char tabb[4 * 256];
            for (int i = 0; i < cn * 256; ++i) tabb[i] = (unsigned char)(int)h_glut[i];
            unsigned char *d_glutb = (unsigned char *)scr + tab_bytes;
            HIPCHK(upload_small(ctx, d_glutb, tabb, (size_t)cn * 256));
            const size_t lds = (size_t)FG_Y_BYTES + FG_X_BYTES + 1024 + (size_t)FG_IH * ((cn * FG_IW + 3) & ~3);
            const void *kf = cn == 1 ? (const void *)k_cc_fused8<1> : cn == 2 ? (const void *)k_cc_fused8<2>
                             : cn == 3 ? (const void *)k_cc_fused8<3> : (const void *)k_cc_fused8<4>;
            (void)hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            {
                ProfScope ps(ctx, "guided_fused");
                const dim3 grid((w + FG_TW - 1) / FG_TW, (h + FG_TH - 1) / FG_TH);
                const unsigned char *lb = d_glutb;
#define FG_LAUNCH(N) hipLaunchKernelGGL(k_cc_fused8<N>, grid, dim3(FG_NT), lds, ctx->stream, d_img, (long long)stride, h, w, lb, eps, \
                                        d_out, (long long)out_stride)
                switch (cn) {
                case 1: FG_LAUNCH(1); break;
                case 2: FG_LAUNCH(2); break;
                case 3: FG_LAUNCH(3); break;
                default: FG_LAUNCH(4); break;
                }
#undef FG_LAUNCH
            }
            return check_launch("guided_fused");
        }
        if (cls == 2) {
            const size_t lds = (size_t)FG_Y_BYTES + FG_X_BYTES + (size_t)cn * 1024 + (size_t)FG_IH * ((cn * FG_IW + 3) & ~3);
            const void *kf = cn == 1 ? (const void *)k_cc_fused8f<1> : cn == 2 ? (const void *)k_cc_fused8f<2>
                             : cn == 3 ? (const void *)k_cc_fused8f<3> : (const void *)k_cc_fused8f<4>;
            (void)hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            {
                ProfScope ps(ctx, "guided_fused_f");
                const dim3 grid((w + FG_TW - 1) / FG_TW, (h + FG_TH - 1) / FG_TH);
                const float *lf = d_glut;
#define FG_LAUNCH(N) hipLaunchKernelGGL(k_cc_fused8f<N>, grid, dim3(FG_NT), lds, ctx->stream, d_img, (long long)stride, h, w, lf, eps, \
                                        d_out, (long long)out_stride)
                switch (cn) {
                case 1: FG_LAUNCH(1); break;
                case 2: FG_LAUNCH(2); break;
                case 3: FG_LAUNCH(3); break;
                default: FG_LAUNCH(4); break;
                }
#undef FG_LAUNCH
            }
            return check_launch("guided_fused_f");
        }
    }
    // a / b planes (8 bytes per sample): an allocation of their own, released when the call is done
    float *d_a = nullptr;
    {
        hipError_t e = hipMalloc((void **)&d_a, 2 * npx * sizeof(float));
        if (e != hipSuccess)
            return sr_set_error(e == hipErrorOutOfMemory ? SR_ERR_OOM : SR_ERR_HIP, "sr_color_correct_u8: %s", hipGetErrorString(e));
    }
    float *d_b = d_a + npx;
    const int PH = CC_TH + radius - 1, PW = CC_TW + radius - 1;
    const size_t patch = (((size_t)2 * PH * PW * sizeof(float)) + 15) & ~(size_t)15;
    const size_t lds1 = patch + (size_t)PH * CC_TW * 4 * sizeof(double), lds2 = patch + (size_t)PH * CC_TW * 2 * sizeof(double);
    const double scale = 1.0 / ((double)radius * (double)radius);
    const dim3 grid((w + CC_TW - 1) / CC_TW, (h + CC_TH - 1) / CC_TH);
    if (lds1 > 48 * 1024) {
        (void)hipFuncSetAttribute((const void *)k_cc_coeff, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        (void)hipFuncSetAttribute((const void *)k_cc_apply, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    }
    if (radius == CC8_R) {                      // the reference's setting: register-blocked kernels, planar a / b
        {
            ProfScope ps(ctx, "guided_coeff");
            hipLaunchKernelGGL(k_cc_coeff8, grid, dim3(256), 0, ctx->stream, d_img, (long long)stride, h, w, cn, (const float *)d_glut, eps,
                               scale, d_a, d_b);
        }
        {
            ProfScope ps(ctx, "guided_apply");
            hipLaunchKernelGGL(k_cc_apply8, grid, dim3(256), 0, ctx->stream, d_img, (long long)stride, h, w, cn, (const float *)d_glut, scale,
                               (const float *)d_a, (const float *)d_b, d_out, (long long)out_stride);
        }
    } else {
        {
            ProfScope ps(ctx, "guided_coeff");
            hipLaunchKernelGGL(k_cc_coeff, grid, dim3(256), lds1, ctx->stream, d_img, (long long)stride, h, w, cn, (const float *)d_glut,
                               radius, eps, scale, d_a, d_b);
        }
        {
            ProfScope ps(ctx, "guided_apply");
            hipLaunchKernelGGL(k_cc_apply, grid, dim3(256), lds2, ctx->stream, d_img, (long long)stride, h, w, cn, (const float *)d_glut,
                               radius, scale, (const float *)d_a, (const float *)d_b, d_out, (long long)out_stride);
        }
    }
    rc = check_launch("color_correct");
    hipError_t es = stream_sync(ctx);
    (void)hipFree(d_a);
    if (rc) return rc;
    if (es != hipSuccess) return sr_set_error(SR_ERR_HIP, "sr_color_correct_u8: %s", hipGetErrorString(es));
    return SR_OK;
}

}  // extern "C"

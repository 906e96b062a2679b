/*
 * sr_oracle.c -- plain-C CPU restatement of the reference's tile -> blend -> assess
 * arithmetic.  TEST INFRASTRUCTURE ONLY: it is the checker the HIP path is compared
 * against (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  The product
 * never links, loads or calls it.
 *
 * Parity status: PSNR/SSIM pinned against scikit-image 0.18.3 fixtures (tests/golden);
 * tile bookkeeping pinned by the survey's known answers; everything the reference
 * delegates to OpenCV (pyrDown/pyrUp/copyMakeBorder/cvtColor/resize/GaussianBlur) is
 * "parity unpinned" -- cv2 cannot be installed here and the reference's own tests hold
 * no pixel values.  Semantics follow SURVEY.md Appendix A.
 *
 * Every fp32 expression is written in the evaluation order oracle_np.py and the HIP
 * kernels use; build with -ffp-contract=off so no FMA is formed.
 *
 * Reference call sites restated (paths relative to /root/reference):
 *   blending_module.py:217-269 build_gaussian_pyramid      -> orc_pyr_down_f32
 *   blending_module.py:271-363 laplacian build / collapse  -> orc_pyr_up_f32
 *   blending_module.py:369-506 laplacian_fusion            -> orc_laplacian_fusion
 *   blending_module.py:508-561 _create_distance_weight_map -> orc_weight_lut
 *   blending_module.py:661-760 weighted_average_fusion     -> orc_weighted_fusion
 *   tiling_module.py:522-570,713-724 extract + pad         -> orc_tile_extract_pad
 *   quality_assessment_module.py:277-320 calculate_psnr    -> orc_psnr_u8
 *   quality_assessment_module.py:322-417 calculate_ssim    -> orc_ssim_u8
 *   quality_assessment_module.py:359-360 RGB2GRAY          -> orc_rgb2gray_u8
 *   quality_assessment_module.py:226-253 downsample_bicubic-> orc_resize_cubic_u8
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ---- border rules (cv2.borderInterpolate) ------------------------------------- */
enum { ORC_REFLECT101 = 0, ORC_REPLICATE = 1, ORC_REFLECT = 2, ORC_CONSTANT = 3 };

static inline int border_index(int p, int n, int mode)
{
    if (p >= 0 && p < n) return p;
    if (mode == ORC_REPLICATE) return p < 0 ? 0 : n - 1;
    if (n == 1) return 0;
    int delta = (mode == ORC_REFLECT101) ? 1 : 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p - 1 + delta;
        else       p = n - 1 - (p - n) - delta;
    }
    return p;
}

/* ---- cv2.pyrDown, fp32, interleaved cn channels ---------------------------------- */
ORC_API void orc_pyr_down_f32(const float *src, int h, int w, int cn, float *dst)
{
    const int ho = (h + 1) / 2, wo = (w + 1) / 2;
    float *row = (float *)malloc((size_t)h * wo * cn * sizeof(float));
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const float *s = src + (size_t)y * w * cn;
        float *r = row + (size_t)y * wo * cn;
        for (int x = 0; x < wo; ++x) {
            int i0 = border_index(2 * x - 2, w, ORC_REFLECT101) * cn;
            int i1 = border_index(2 * x - 1, w, ORC_REFLECT101) * cn;
            int i2 = (2 * x) * cn;
            int i3 = border_index(2 * x + 1, w, ORC_REFLECT101) * cn;
            int i4 = border_index(2 * x + 2, w, ORC_REFLECT101) * cn;
            for (int c = 0; c < cn; ++c) {
                float a = s[i2 + c] * 6.0f;
                float b = (s[i1 + c] + s[i3 + c]) * 4.0f;
                r[x * cn + c] = ((a + b) + s[i0 + c]) + s[i4 + c];
            }
        }
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < ho; ++y) {
        const float *r0 = row + (size_t)border_index(2 * y - 2, h, ORC_REFLECT101) * wo * cn;
        const float *r1 = row + (size_t)border_index(2 * y - 1, h, ORC_REFLECT101) * wo * cn;
        const float *r2 = row + (size_t)(2 * y) * wo * cn;
        const float *r3 = row + (size_t)border_index(2 * y + 1, h, ORC_REFLECT101) * wo * cn;
        const float *r4 = row + (size_t)border_index(2 * y + 2, h, ORC_REFLECT101) * wo * cn;
        float *d = dst + (size_t)y * wo * cn;
        for (int i = 0; i < wo * cn; ++i) {
            float a = r2[i] * 6.0f;
            float b = (r1[i] + r3[i]) * 4.0f;
            float v = ((a + b) + r0[i]) + r4[i];
            d[i] = v * (1.0f / 256.0f);
        }
    }
    free(row);
}

/* one source row -> unnormalised (x8) upsampled row of wd entries */
static void up_row(const float *s, int ws, int cn, float *r, int wd)
{
    for (int c = 0; c < cn; ++c) {
        if (ws == 1) {
            float e = s[c] * 6.0f + s[c] * 2.0f;
            float o = s[c] * 8.0f;
            r[c] = e;
            if (wd > 1) r[cn + c] = o;
            continue;
        }
        for (int x = 0; x < ws; ++x) {
            float e, o;
            if (x == 0) {
                e = s[c] * 6.0f + s[cn + c] * 2.0f;
                o = (s[c] + s[cn + c]) * 4.0f;
            } else if (x == ws - 1) {
                e = s[(x - 1) * cn + c] + s[x * cn + c] * 7.0f;
                o = s[x * cn + c] * 8.0f;
            } else {
                e = (s[(x - 1) * cn + c] + s[x * cn + c] * 6.0f) + s[(x + 1) * cn + c];
                o = (s[x * cn + c] + s[(x + 1) * cn + c]) * 4.0f;
            }
            r[(2 * x) * cn + c] = e;
            if (2 * x + 1 < wd) r[(2 * x + 1) * cn + c] = o;
        }
    }
}

/* ---- cv2.pyrUp(src, dstsize=(wd,hd)), wd in {2ws-1, 2ws} ------------------------- */
ORC_API void orc_pyr_up_f32(const float *src, int hs, int ws, int cn, float *dst, int hd, int wd)
{
    float *rows = (float *)malloc((size_t)hs * wd * cn * sizeof(float));
#pragma omp parallel for schedule(static)
    for (int y = 0; y < hs; ++y)
        up_row(src + (size_t)y * ws * cn, ws, cn, rows + (size_t)y * wd * cn, wd);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < hs; ++y) {
        int ym = (y - 1 < 0) ? (hs > 1 ? 1 : 0) : y - 1;       /* reflect-101 at the top   */
        int yp = (y + 1 > hs - 1) ? hs - 1 : y + 1;            /* clamp at the bottom      */
        const float *r0 = rows + (size_t)ym * wd * cn;
        const float *r1 = rows + (size_t)y * wd * cn;
        const float *r2 = rows + (size_t)yp * wd * cn;
        float *d0 = dst + (size_t)(2 * y) * wd * cn;
        float *d1 = (2 * y + 1 < hd) ? dst + (size_t)(2 * y + 1) * wd * cn : NULL;
        for (int i = 0; i < wd * cn; ++i) {
            float e = (r0[i] + r1[i] * 6.0f) + r2[i];
            d0[i] = e * (1.0f / 64.0f);
            if (d1) {
                float o = (r1[i] + r2[i]) * 4.0f;
                d1[i] = o * (1.0f / 64.0f);
            }
        }
    }
    free(rows);
}

/* ---- weight LUT by integer edge distance d = 0..fw (blending_module.py:547-559) --- */
enum { ORC_W_LINEAR = 0, ORC_W_COSINE = 1, ORC_W_SIGMOID = 2 };

ORC_API void orc_weight_lut(int fw, int type, float *lut /* fw+1 */)
{
    for (int d = 0; d <= fw; ++d) {
        double nd = (double)d / (double)fw;
        if (nd < 0) nd = 0;
        if (nd > 1) nd = 1;
        double wv;
        if (type == ORC_W_COSINE)       wv = 0.5 * (1 - cos(M_PI * nd));
        else if (type == ORC_W_SIGMOID) wv = 1 / (1 + exp(-10 * (nd - 0.5)));
        else                            wv = nd;
        lut[d] = (float)wv;
    }
}

static void weight_map(int h, int w, int fw, const float *lut, float *out)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        int dy = y < h - 1 - y ? y : h - 1 - y;
        for (int x = 0; x < w; ++x) {
            int dx = x < w - 1 - x ? x : w - 1 - x;
            int d = dy < dx ? dy : dx;
            out[(size_t)y * w + x] = lut[d < fw ? d : fw];
        }
    }
}

static void level_dims(int h, int w, int levels, int *hs, int *ws, int *n)
{
    int k = 0;
    hs[0] = h; ws[0] = w; k = 1;
    while (k < levels && hs[k - 1] >= 2 && ws[k - 1] >= 2) {
        hs[k] = (hs[k - 1] + 1) / 2;
        ws[k] = (ws[k - 1] + 1) / 2;
        ++k;
    }
    *n = k;
}

/*
 * laplacian_fusion (blending_module.py:369-506).
 * tiles: n pointers to h_i x w_i x cn images, dtype u8 (is_f32 == 0) or fp32.
 * pos: n x (y, x).  out_u8: H x W x cn.  out_f32 (nullable): normalised value before
 * clip/truncation.  Returns 0, or -1 on a tile whose min side < 8 (reference: 0-width feather).
 */
ORC_API int orc_laplacian_fusion(const void *const *tiles, const int *hw, const int *pos, int n,
                                 int cn, int is_f32, int H, int W, int levels, int wtype,
                                 uint8_t *out_u8, float *out_f32)
{
    float *acc = (float *)calloc((size_t)H * W * cn, sizeof(float));
    float *wacc = (float *)calloc((size_t)H * W, sizeof(float));
    for (int t = 0; t < n; ++t) {
        const int h = hw[2 * t], w = hw[2 * t + 1], ty = pos[2 * t], tx = pos[2 * t + 1];
        const int fw = (h < w ? h : w) / 8;
        if (fw < 1) { free(acc); free(wacc); return -1; }
        int hs[32], ws[32], nl;
        level_dims(h, w, levels, hs, ws, &nl);
        float *g[32], *wp[32], *r[32];
        for (int i = 0; i < nl; ++i) {
            g[i] = (float *)malloc((size_t)hs[i] * ws[i] * cn * sizeof(float));
            wp[i] = (float *)malloc((size_t)hs[i] * ws[i] * sizeof(float));
            r[i] = (float *)malloc((size_t)hs[i] * ws[i] * cn * sizeof(float));
        }
        size_t n0 = (size_t)h * w * cn;
        if (is_f32) memcpy(g[0], tiles[t], n0 * sizeof(float));
        else {
            const uint8_t *s = (const uint8_t *)tiles[t];
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < n0; ++i) g[0][i] = (float)s[i];
        }
        float *lut = (float *)malloc((size_t)(fw + 1) * sizeof(float));
        orc_weight_lut(fw, wtype, lut);
        weight_map(h, w, fw, lut, wp[0]);
        for (int i = 1; i < nl; ++i) {
            orc_pyr_down_f32(g[i - 1], hs[i - 1], ws[i - 1], cn, g[i]);
            orc_pyr_down_f32(wp[i - 1], hs[i - 1], ws[i - 1], 1, wp[i]);
        }
        /* R[last] = G[last] * W[last];  R[i] = up(R[i+1]) + (G[i] - up(G[i+1])) * W[i] */
        {
            const int i = nl - 1;
            const size_t np = (size_t)hs[i] * ws[i];
#pragma omp parallel for schedule(static)
            for (size_t p = 0; p < np; ++p)
                for (int c = 0; c < cn; ++c) r[i][p * cn + c] = g[i][p * cn + c] * wp[i][p];
        }
        for (int i = nl - 2; i >= 0; --i) {
            const size_t np = (size_t)hs[i] * ws[i];
            float *ug = (float *)malloc(np * cn * sizeof(float));
            float *ur = (float *)malloc(np * cn * sizeof(float));
            orc_pyr_up_f32(g[i + 1], hs[i + 1], ws[i + 1], cn, ug, hs[i], ws[i]);
            orc_pyr_up_f32(r[i + 1], hs[i + 1], ws[i + 1], cn, ur, hs[i], ws[i]);
#pragma omp parallel for schedule(static)
            for (size_t p = 0; p < np; ++p)
                for (int c = 0; c < cn; ++c) {
                    float lap = g[i][p * cn + c] - ug[p * cn + c];
                    float wl = lap * wp[i][p];
                    r[i][p * cn + c] = ur[p * cn + c] + wl;
                }
            free(ug); free(ur);
        }
        int ye = ty + h < H ? ty + h : H, xe = tx + w < W ? tx + w : W;
#pragma omp parallel for schedule(static)
        for (int y = (ty < 0 ? 0 : ty); y < ye; ++y)
            for (int x = (tx < 0 ? 0 : tx); x < xe; ++x) {
                size_t cp = (size_t)y * W + x, tp = (size_t)(y - ty) * w + (x - tx);
                for (int c = 0; c < cn; ++c) acc[cp * cn + c] += r[0][tp * cn + c];
                wacc[cp] += wp[0][tp];
            }
        for (int i = 0; i < nl; ++i) { free(g[i]); free(wp[i]); free(r[i]); }
        free(lut);
    }
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < (size_t)H * W; ++p) {
        float wv = wacc[p] > 1e-6f ? wacc[p] : 1e-6f;
        for (int c = 0; c < cn; ++c) {
            float v = acc[p * cn + c] / wv;
            if (out_f32) out_f32[p * cn + c] = v;
            float cl = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
            out_u8[p * cn + c] = (uint8_t)cl;          /* truncation, as ndarray.astype(uint8) */
        }
    }
    free(acc); free(wacc);
    return 0;
}

/* weighted_average_fusion (blending_module.py:661-760), generated weights only */
ORC_API int orc_weighted_fusion(const void *const *tiles, const int *hw, const int *pos, int n,
                                int cn, int is_f32, int H, int W, int wtype,
                                uint8_t *out_u8, float *out_f32)
{
    float *acc = (float *)calloc((size_t)H * W * cn, sizeof(float));
    float *wacc = (float *)calloc((size_t)H * W, sizeof(float));
    for (int t = 0; t < n; ++t) {
        const int h = hw[2 * t], w = hw[2 * t + 1], ty = pos[2 * t], tx = pos[2 * t + 1];
        const int fw = (h < w ? h : w) / 8;
        if (fw < 1) { free(acc); free(wacc); return -1; }
        float *lut = (float *)malloc((size_t)(fw + 1) * sizeof(float));
        orc_weight_lut(fw, wtype, lut);
        int ye = ty + h < H ? ty + h : H, xe = tx + w < W ? tx + w : W;
#pragma omp parallel for schedule(static)
        for (int y = (ty < 0 ? 0 : ty); y < ye; ++y) {
            int ly = y - ty, dy = ly < h - 1 - ly ? ly : h - 1 - ly;
            for (int x = (tx < 0 ? 0 : tx); x < xe; ++x) {
                int lx = x - tx, dx = lx < w - 1 - lx ? lx : w - 1 - lx;
                int d = dy < dx ? dy : dx;
                float wv = lut[d < fw ? d : fw];
                size_t cp = (size_t)y * W + x, tp = (size_t)ly * w + lx;
                for (int c = 0; c < cn; ++c) {
                    float v = is_f32 ? ((const float *)tiles[t])[tp * cn + c]
                                     : (float)((const uint8_t *)tiles[t])[tp * cn + c];
                    acc[cp * cn + c] += v * wv;
                }
                wacc[cp] += wv;
            }
        }
        free(lut);
    }
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < (size_t)H * W; ++p) {
        float wv = wacc[p] > 1e-6f ? wacc[p] : 1e-6f;
        for (int c = 0; c < cn; ++c) {
            float v = acc[p * cn + c] / wv;
            if (out_f32) out_f32[p * cn + c] = v;
            float cl = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
            out_u8[p * cn + c] = (uint8_t)cl;
        }
    }
    free(acc); free(wacc);
    return 0;
}

/* ---- tile extract + bottom/right pad (tiling_module.py:713-724, 522-570) ---------- */
ORC_API void orc_tile_extract_pad(const uint8_t *img, int H, int W, int cn, int x, int y, int w, int h,
                                  int block, int mode, uint8_t *dst /* block x block x cn */)
{
    (void)H;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < block; ++r) {
        for (int c = 0; c < block; ++c) {
            uint8_t *d = dst + ((size_t)r * block + c) * cn;
            if (mode == ORC_CONSTANT && (r >= h || c >= w)) {
                for (int k = 0; k < cn; ++k) d[k] = 0;
                continue;
            }
            int sr = border_index(r, h, mode), sc = border_index(c, w, mode);
            const uint8_t *s = img + ((size_t)(y + sr) * W + (x + sc)) * cn;
            for (int k = 0; k < cn; ++k) d[k] = s[k];
        }
    }
}

/* ---- quality metrics -------------------------------------------------------------- */
ORC_API void orc_rgb2gray_u8(const uint8_t *rgb, size_t npix, int shift, uint8_t *gray)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < npix; ++i) {
        int r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2];
        gray[i] = (shift == 15) ? (uint8_t)((r * 9798 + g * 19235 + b * 3735 + (1 << 14)) >> 15)
                                : (uint8_t)((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14);
    }
}

/* PSNR over all elements of two u8 arrays given as h rows of `rowlen` elements with strides */
ORC_API double orc_psnr_u8(const uint8_t *a, size_t stride_a, const uint8_t *b, size_t stride_b,
                           int h, size_t rowlen, double data_range)
{
    double sum = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : sum)
    for (int y = 0; y < h; ++y) {
        const uint8_t *pa = a + (size_t)y * stride_a, *pb = b + (size_t)y * stride_b;
        uint64_t s = 0;
        for (size_t i = 0; i < rowlen; ++i) {
            int d = (int)pa[i] - (int)pb[i];
            s += (uint64_t)(d * d);
        }
        sum += (double)s;
    }
    double mse = sum / ((double)h * (double)rowlen);
    if (mse == 0.0) return INFINITY;
    return 10.0 * log10((data_range * data_range) / mse);
}

enum { ORC_SSIM_UNIFORM7 = 0, ORC_SSIM_GAUSS11 = 1, ORC_SSIM_SIMPLE = 2 };

static void sep_filter(const double *in, int h, int w, const double *k, int klen, int bmode, double *tmp,
                       double *out)
{
    const int r = klen / 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            double s = 0.0;
            for (int j = 0; j < klen; ++j)
                s += in[(size_t)border_index(y + j - r, h, bmode) * w + x] * k[j];
            tmp[(size_t)y * w + x] = s;
        }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            double s = 0.0;
            for (int j = 0; j < klen; ++j)
                s += tmp[(size_t)y * w + border_index(x + j - r, w, bmode)] * k[j];
            out[(size_t)y * w + x] = s;
        }
}

/* SSIM of two gray u8 images (quality_assessment_module.py:365-417; skimage 0.18.3
 * _structural_similarity.py:127-214).  mode: uniform7 / gauss11 (branch A, cropped mean),
 * simple (branch B, cv2.GaussianBlur REFLECT_101, full-map mean). */
ORC_API double orc_ssim_u8(const uint8_t *g1, const uint8_t *g2, int h, int w, int mode, double data_range)
{
    const size_t n = (size_t)h * w;
    double k[11];
    int klen, pad, bmode;
    double cov_norm;
    if (mode == ORC_SSIM_UNIFORM7) {
        klen = 7; pad = 3; cov_norm = 49.0 / 48.0; bmode = ORC_REFLECT;
        for (int i = 0; i < 7; ++i) k[i] = 1.0 / 7.0;
    } else if (mode == ORC_SSIM_GAUSS11) {
        klen = 11; pad = 5; cov_norm = 1.0; bmode = ORC_REFLECT;
        double s = 0;
        for (int i = 0; i < 11; ++i) { double x = i - 5; k[i] = exp(-0.5 / (1.5 * 1.5) * x * x); s += k[i]; }
        for (int i = 0; i < 11; ++i) k[i] /= s;
    } else {
        klen = 11; pad = 0; cov_norm = 1.0; bmode = ORC_REFLECT101; data_range = 255.0;
        double s = 0;
        for (int i = 0; i < 11; ++i) { double x = i - 5.0; k[i] = exp(-(x * x) / (2.0 * 1.5 * 1.5)); s += k[i]; }
        for (int i = 0; i < 11; ++i) k[i] /= s;
    }
    if (h <= 2 * pad || w <= 2 * pad) return NAN;
    const double c1 = (0.01 * data_range) * (0.01 * data_range);
    const double c2 = (0.03 * data_range) * (0.03 * data_range);
    double *x = (double *)malloc(n * sizeof(double)), *y = (double *)malloc(n * sizeof(double));
    double *p = (double *)malloc(n * sizeof(double)), *tmp = (double *)malloc(n * sizeof(double));
    double *ux = (double *)malloc(n * sizeof(double)), *uy = (double *)malloc(n * sizeof(double));
    double *uxx = (double *)malloc(n * sizeof(double)), *uyy = (double *)malloc(n * sizeof(double));
    double *uxy = (double *)malloc(n * sizeof(double));
    for (size_t i = 0; i < n; ++i) { x[i] = g1[i]; y[i] = g2[i]; }
    sep_filter(x, h, w, k, klen, bmode, tmp, ux);
    sep_filter(y, h, w, k, klen, bmode, tmp, uy);
    for (size_t i = 0; i < n; ++i) p[i] = x[i] * x[i];
    sep_filter(p, h, w, k, klen, bmode, tmp, uxx);
    for (size_t i = 0; i < n; ++i) p[i] = y[i] * y[i];
    sep_filter(p, h, w, k, klen, bmode, tmp, uyy);
    for (size_t i = 0; i < n; ++i) p[i] = x[i] * y[i];
    sep_filter(p, h, w, k, klen, bmode, tmp, uxy);
    double total = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : total)
    for (int yy = pad; yy < h - pad; ++yy) {
        double rs = 0.0;
        for (int xx = pad; xx < w - pad; ++xx) {
            size_t i = (size_t)yy * w + xx;
            double vx = cov_norm * (uxx[i] - ux[i] * ux[i]);
            double vy = cov_norm * (uyy[i] - uy[i] * uy[i]);
            double vxy = cov_norm * (uxy[i] - ux[i] * uy[i]);
            double a1 = 2 * ux[i] * uy[i] + c1, a2 = 2 * vxy + c2;
            double b1 = ux[i] * ux[i] + uy[i] * uy[i] + c1, b2 = vx + vy + c2;
            rs += (a1 * a2) / (b1 * b2);
        }
        total += rs;
    }
    free(x); free(y); free(p); free(tmp); free(ux); free(uy); free(uxx); free(uyy); free(uxy);
    return total / ((double)(h - 2 * pad) * (double)(w - 2 * pad));
}

/* ---- cv2.resize INTER_CUBIC on u8 (Appendix A 11) ---------------------------------- */
static void cubic_table(int n_src, int n_dst, int *ofs, short *coef /* n_dst*4 */)
{
    double scale = 1.0 / ((double)n_dst / (double)n_src);
    for (int d = 0; d < n_dst; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        const float A = -0.75f;
        float c[4];
        c[0] = ((A * (f + 1.0f) - 5.0f * A) * (f + 1.0f) + 8.0f * A) * (f + 1.0f) - 4.0f * A;
        c[1] = ((A + 2.0f) * f - (A + 3.0f)) * f * f + 1.0f;
        float f2 = 1.0f - f;
        c[2] = ((A + 2.0f) * f2 - (A + 3.0f)) * f2 * f2 + 1.0f;
        c[3] = 1.0f - c[0] - c[1] - c[2];
        ofs[d] = s;
        for (int k = 0; k < 4; ++k) {
            float v = rintf(c[k] * 2048.0f);
            coef[d * 4 + k] = (short)(v < -32768.f ? -32768.f : (v > 32767.f ? 32767.f : v));
        }
    }
}

ORC_API void orc_resize_cubic_u8(const uint8_t *src, int h, int w, int cn, uint8_t *dst, int dh, int dw)
{
    int *xo = (int *)malloc(sizeof(int) * dw), *yo = (int *)malloc(sizeof(int) * dh);
    short *xa = (short *)malloc(sizeof(short) * 4 * dw), *ya = (short *)malloc(sizeof(short) * 4 * dh);
    cubic_table(w, dw, xo, xa);
    cubic_table(h, dh, yo, ya);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y) {
        for (int x = 0; x < dw; ++x)
            for (int c = 0; c < cn; ++c) {
                int64_t acc = 0;
                for (int ky = 0; ky < 4; ++ky) {
                    int sy = yo[y] + ky - 1;
                    sy = sy < 0 ? 0 : (sy > h - 1 ? h - 1 : sy);
                    int hsum = 0;
                    for (int kx = 0; kx < 4; ++kx) {
                        int sx = xo[x] + kx - 1;
                        sx = sx < 0 ? 0 : (sx > w - 1 ? w - 1 : sx);
                        hsum += (int)src[((size_t)sy * w + sx) * cn + c] * xa[x * 4 + kx];
                    }
                    acc += (int64_t)hsum * ya[y * 4 + ky];
                }
                int64_t v = (acc + (1 << 21)) >> 22;
                dst[((size_t)y * dw + x) * cn + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
    }
    free(xo); free(yo); free(xa); free(ya);
}

#ifdef _OPENMP
#include <omp.h>
ORC_API void orc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
ORC_API int orc_get_threads(void) { return omp_get_max_threads(); }
#else
ORC_API void orc_set_threads(int n) { (void)n; }
ORC_API int orc_get_threads(void) { return 1; }
#endif

ORC_API int orc_version(void) { return 1; }

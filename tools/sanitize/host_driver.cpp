// Host-side sanitizer driver ("-fsanitize=address host build of the C ABI"): every host-only entry point of
// libsrhip (csrc/sr_host.cpp, csrc/sr_encode.cpp) is called over a spread of geometries with EXACT-size output buffers, built
// with -fsanitize=address,undefined -fno-sanitize-recover=all, so an overrun, a signed overflow or a misaligned access ends
// the run.  Built and run by tests/test_sanitizers.py (CPU); GPU sanitizers are not available on the pool.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "sr_hip.h"

static const int MAX_LEVELS = 16;     // SR_MAX_LEVELS of csrc/sr_internal.h

static uint32_t rng_state = 20260313u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }
static int rint_(int lo, int hi) { return lo + (int)(rnd() % (uint32_t)(hi - lo + 1)); }

#define CHECK(cond)                                                                      \
    do {                                                                                 \
        if (!(cond)) { std::fprintf(stderr, "CHECK failed line %d: %s (%s)\n", __LINE__, #cond, sr_last_error()); std::exit(1); } \
    } while (0)

int main(int argc, char **argv)
{
    const std::string tmp = argc > 1 ? argv[1] : "/tmp";
    int calls = 0;
    // ---- bookkeeping ---------------------------------------------------------------------------------------------
    for (int it = 0; it < 300; ++it) {
        const int W = rint_(1, 9000), H = rint_(1, 9000), block = rint_(2, 4096), ov = rint_(0, block - 1);
        int n = 0;
        int rc = sr_tile_plan(W, H, block, ov, &n, nullptr, 0);
        CHECK(rc == SR_OK || rc == SR_ERR_SHAPE);
        CHECK(n >= 1);
        std::vector<int> xywh((size_t)n * 4);
        CHECK(sr_tile_plan(W, H, block, ov, &n, xywh.data(), n) == SR_OK);
        if (n > 1) { int m = 0; CHECK(sr_tile_plan(W, H, block, ov, &m, xywh.data(), n - 1) == SR_ERR_SHAPE && m == n); }
        for (int t = 0; t < n; t += (n > 64 ? n / 32 : 1)) {
            int tblr[4];
            CHECK(sr_tile_overlaps(xywh[4 * t], xywh[4 * t + 1], xywh[4 * t + 2], xywh[4 * t + 3], W, H, block, ov, tblr) == SR_OK);
        }
        if (n <= 4096) {
            std::vector<int> nbr((size_t)n * 4);
            CHECK(sr_tile_neighbors(xywh.data(), n, block, ov, nbr.data()) == SR_OK);
            for (int v : nbr) CHECK(v >= -1 && v < n);
        }
        calls += 3;
    }
    for (int mp : {100, 150, 200})
        for (int it = 0; it < 50; ++it) { int ow, oh; CHECK(sr_target_size(rint_(1, 8000), rint_(1, 8000), mp, &ow, &oh) == SR_OK && ow > 0 && oh > 0); }
    { int ow, oh; CHECK(sr_target_size(100, 100, 123, &ow, &oh) != SR_OK); CHECK(sr_target_size(0, 5, 100, &ow, &oh) != SR_OK); }
    for (int wt = 0; wt <= 3; ++wt)
        for (int fw : {1, 2, 7, 371, 4096}) { std::vector<float> lut((size_t)fw + 1); CHECK(sr_weight_lut(fw, wt, lut.data()) == SR_OK); }
    CHECK(sr_psnr_from_sse(0, 10, 255.0) > 1e300);           // MSE 0 -> inf (skimage)
    CHECK(sr_psnr_from_sse(650250, 10, 255.0) == 0.0);
    // ---- strip planner -------------------------------------------------------------------------------------------
    for (int it = 0; it < 120; ++it) {
        const int rows = rint_(1, 6), cols = rint_(1, 6), tw = rint_(16, 900), th = rint_(16, 900);
        const int ovx = rint_(0, tw / 2), ovy = rint_(0, th / 2), levels = rint_(1, MAX_LEVELS), cn = (it & 1) ? 3 : 1;
        std::vector<sr_tile_rect> rects;
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) rects.push_back({c * (tw - ovx), r * (th - ovy), tw, th});
        const int n = (int)rects.size();
        const int CW = (cols - 1) * (tw - ovx) + tw, CH = (rows - 1) * (th - ovy) + th - (it % 5 == 0 ? rint_(0, th / 3) : 0);
        for (int world : {1, 2, 3, 8}) {
            std::vector<int> bounds((size_t)world + 1), rws((size_t)2 * world), need((size_t)2 * world * n), owner((size_t)n);
            CHECK(sr_strip_bounds(rects.data(), n, levels, CH, CW, world, bounds.data()) == SR_OK);
            CHECK(bounds[0] == 0 && bounds[(size_t)world] == CH);
            for (int pol = 0; pol <= 2; ++pol) {
                CHECK(sr_exchange_plan(rects.data(), n, cn, levels, CH, CW, world, 5, pol, bounds.data(), rws.data(), need.data(), owner.data()) == SR_OK);
                for (int o : owner) CHECK(o >= 0 && o < world);
            }
            std::vector<int> rr((size_t)2 * n);
            CHECK(sr_strip_tile_rows(rects.data(), n, levels, CH, rws[0], rws[1], rr.data()) == SR_OK);
            for (int rank = 0; rank < world; ++rank) {
                std::vector<const void *> owned((size_t)n, nullptr);
                std::vector<void *> recv((size_t)n, nullptr);
                std::vector<int64_t> stride((size_t)n);
                for (int t = 0; t < n; ++t) {
                    stride[(size_t)t] = (int64_t)rects[(size_t)t].w * cn;
                    if (owner[(size_t)t] == rank) owned[(size_t)t] = (const void *)(uintptr_t)(0x10000000u + 0x100000u * (unsigned)t);
                    else recv[(size_t)t] = (void *)(uintptr_t)(0x70000000u + 0x100000u * (unsigned)t);
                }
                std::vector<sr_xfer> sends((size_t)n * world), recvs((size_t)n);
                int ns = 0, nr = 0;
                CHECK(sr_exchange_xfers(rects.data(), n, cn, world, rank, need.data(), owner.data(), owned.data(), stride.data(), recv.data(),
                                        sends.data(), n * world, &ns, recvs.data(), n, &nr) == SR_OK);
                if (ns > 0) { int a = 0, b = 0; CHECK(sr_exchange_xfers(rects.data(), n, cn, world, rank, need.data(), owner.data(), owned.data(), stride.data(),
                                                                      recv.data(), sends.data(), ns - 1, &a, recvs.data(), n, &b) == SR_ERR_SHAPE && a == ns); }
            }
            calls += 6;
        }
    }
    for (int l = 1; l <= MAX_LEVELS; ++l) { int b, a; CHECK(sr_pyramid_halo(l, &b, &a) == SR_OK && b >= 0 && a >= 0); }
    // ---- encoders --------------------------------------------------------------------------------------------------
    const int sizes[][2] = {{1, 1}, {1, 17}, {17, 1}, {7, 9}, {16, 16}, {33, 65}, {257, 131}, {600, 811}};
    for (auto &sz : sizes)
        for (int cn : {1, 3, 4}) {
            const int h = sz[0], w = sz[1];
            const int64_t stride = (int64_t)w * cn + (h % 3);          // padded rows too
            std::vector<uint8_t> img((size_t)(stride * h));
            for (size_t i = 0; i < img.size(); ++i) img[i] = (uint8_t)((i % 7 == 0) ? rnd() : (i / 5));
            const std::string base = tmp + "/san_" + std::to_string(h) + "x" + std::to_string(w) + "_" + std::to_string(cn);
            for (int threads : {1, 3}) {
                CHECK(sr_encode_tiff_lzw(img.data(), h, w, cn, stride, (base + ".tif").c_str(), threads) == SR_OK);
                CHECK(sr_encode_png(img.data(), h, w, cn, stride, 3, (base + ".png").c_str(), threads) == SR_OK);
                if (cn != 4) CHECK(sr_encode_jpeg(img.data(), h, w, cn, stride, 95, (base + ".jpg").c_str(), threads) == SR_OK);
                calls += 3;
            }
            std::remove((base + ".tif").c_str()); std::remove((base + ".png").c_str()); std::remove((base + ".jpg").c_str());
        }
    // LZW worst cases: constant data (longest matches) and noise (table clears), big enough for several clears per strip
    for (int mode = 0; mode < 2; ++mode) {
        const int h = 64, w = 4096;
        std::vector<uint8_t> img((size_t)h * w * 3);
        for (auto &v : img) v = mode ? (uint8_t)rnd() : (uint8_t)77;
        const std::string p = tmp + "/san_lzw.tif";
        CHECK(sr_encode_tiff_lzw(img.data(), h, w, 3, (int64_t)w * 3, p.c_str(), 2) == SR_OK);
        std::remove(p.c_str());
    }
    CHECK(sr_encode_png(nullptr, 4, 4, 3, 12, 3, (tmp + "/x.png").c_str(), 1) != SR_OK);
    CHECK(sr_encode_jpeg(nullptr, 4, 4, 3, 12, 95, (tmp + "/x.jpg").c_str(), 1) != SR_OK);
    std::printf("sanitizer-driver-ok %d\n", calls);
    return 0;
}

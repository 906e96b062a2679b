// sr_encode.cpp -- stage 5 of the pipeline (main.py:399-404): the output writers, multi-threaded on the host.
//
// The reference hands the fused canvas to Pillow: TIFF with LZW, PNG with compress_level 3, JPEG with quality 95 -- all
// single-threaded there.  With the blend at a few milliseconds per 200 MP image the writer is the wall clock of the
// pipeline, so the three formats are written here with every core:
//   * TIFF / LZW   : strips of 16 rows, each strip an independent LZW stream (TIFF 6.0, MSB-first codes, early change),
//                    strips compressed in parallel.
//   * PNG          : adaptively filtered rows (libpng's minimum-sum-of-absolute-differences choice among the five
//                    filters), raw deflate of independent ~1 MiB row chunks in parallel, each ended by a full flush
//                    (byte-aligned, no back-reference leaves a chunk), concatenated into one zlib stream; the chunks'
//                    Adler-32 values are combined.
//   * JPEG         : baseline, YCbCr 4:2:0, libjpeg's quality scaling of the Annex K tables, its fixed-point colour
//                    conversion, h2v2 down-sampling with alternating bias, jfdctint (islow) forward DCT and rounding
//                    division, Annex K Huffman tables -- the coefficients libjpeg produces for the same pixels -- with a
//                    restart marker after every MCU row so the rows are entropy-coded in parallel.
// Lossless formats decode to the input bytes; the JPEG decodes to what Pillow's own file decodes to (tests/).
// No GPU work here: plain C++ threads over host memory.
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "sr_internal.h"

namespace {

int pick_threads(int threads, size_t jobs)
{
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    if (threads <= 0) threads = 4;
    threads = std::min<size_t>((size_t)threads, std::max<size_t>(jobs, 1));
    return std::min(threads, 64);
}

// run fn(job) for job = 0 .. jobs-1 on `threads` workers (dynamic distribution)
template <class F>
void parallel_for(size_t jobs, int threads, F fn)
{
    threads = pick_threads(threads, jobs);
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (;;) {
            const size_t j = next.fetch_add(1);
            if (j >= jobs) return;
            fn(j);
        }
    };
    if (threads <= 1) { worker(); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(worker);
    for (auto &t : pool) t.join();
}

bool write_file(const char *path, const std::vector<const std::vector<uint8_t> *> &parts)
{
    FILE *f = fopen(path, "wb");
    if (!f) return false;
    bool ok = true;
    for (auto p : parts)
        if (!p->empty() && fwrite(p->data(), 1, p->size(), f) != p->size()) { ok = false; break; }
    if (fclose(f) != 0) ok = false;
    return ok;
}

// ------------------------------------------------------------------------------------------------------------------
// TIFF 6.0, LZW (compression 5), no predictor, 8-bit samples, chunky
// ------------------------------------------------------------------------------------------------------------------
struct LzwTable {
    // open-addressed hash of (prefix code, byte) -> code; 4096 codes at most in 16384 slots.  A slot is
    // generation << 20 | prefix << 8 | byte: a CLEAR code bumps the generation instead of wiping 64 KB (on photographic
    // data the table fills every ~6 KB of input, so wiping it cost more than the coding itself).
    static constexpr int HBITS = 14, HSIZE = 1 << HBITS;
    uint32_t key[HSIZE];
    uint16_t val[HSIZE];
    uint32_t gen = 0;
    LzwTable() { std::fill(key, key + HSIZE, 0u); }
    void clear()
    {
        if (++gen == (1u << 12)) {              // generation field exhausted: really wipe
            std::fill(key, key + HSIZE, 0u);
            gen = 1;
        }
    }
};

void lzw_encode_strip(const uint8_t *src, size_t n, std::vector<uint8_t> &out, LzwTable &tab)
{
    const int CLEAR = 256, EOI = 257;
    // worst case 12 bits per input byte plus the CLEAR codes (one per 3836 codes) and CLEAR / EOI / padding
    out.resize(n + n / 2 + n / 2048 + 16);
    uint8_t *dst = out.data();
    uint64_t acc = 0;                           // MSB-first bit buffer: the nbits valid bits are its low bits
    int nbits = 0, width = 9, next = 258;
    auto put = [&](int code) {
        acc = (acc << width) | (uint64_t)(unsigned)code;
        nbits += width;
        if (nbits >= 32) {                      // flush four bytes at a time
            const uint32_t w = (uint32_t)(acc >> (nbits - 32));
            dst[0] = (uint8_t)(w >> 24); dst[1] = (uint8_t)(w >> 16); dst[2] = (uint8_t)(w >> 8); dst[3] = (uint8_t)w;
            dst += 4;
            nbits -= 32;
        }
    };
    auto finish = [&]() {
        while (nbits >= 8) { *dst++ = (uint8_t)(acc >> (nbits - 8)); nbits -= 8; }
        if (nbits) *dst++ = (uint8_t)(acc << (8 - nbits));
        out.resize((size_t)(dst - out.data()));
    };
    tab.clear();
    put(CLEAR);
    if (n == 0) { put(EOI); finish(); return; }
    uint32_t tag = tab.gen << 20;
    int prefix = src[0];
    for (size_t i = 1; i < n; ++i) {
        const int c = src[i];
        const uint32_t k = tag | ((uint32_t)prefix << 8) | (uint32_t)c;
        uint32_t h = (((uint32_t)prefix << 8 | (uint32_t)c) * 2654435761u) >> (32 - LzwTable::HBITS);
        int found = -1;
        while ((tab.key[h] >> 20) == tab.gen) {             // a slot of an older generation is empty
            if (tab.key[h] == k) { found = tab.val[h]; break; }
            h = (h + 1) & (LzwTable::HSIZE - 1);
        }
        if (found >= 0) { prefix = found; continue; }
        put(prefix);
        tab.key[h] = k;
        tab.val[h] = (uint16_t)next++;
        // "early change" (TIFF 6.0 section 13, libtiff's rule): the width grows as soon as entry 511 / 1023 / 2047 has been
        // added -- one entry before a plain LZW would need it; the table is cleared once entry 4093 has been added
        if (next == 512) width = 10;
        else if (next == 1024) width = 11;
        else if (next == 2048) width = 12;
        else if (next == 4094) {
            put(CLEAR);
            tab.clear();
            tag = tab.gen << 20;
            next = 258;
            width = 9;
        }
        prefix = c;
    }
    put(prefix);
    // the last code counts like any other: the decoder adds an entry for it, so the width rule runs once more before EOI
    ++next;
    if (next == 4094) { put(CLEAR); width = 9; }
    else if (next == 512) width = 10;
    else if (next == 1024) width = 11;
    else if (next == 2048) width = 12;
    put(EOI);
    finish();
}

void put16(std::vector<uint8_t> &v, uint16_t x) { v.push_back((uint8_t)(x & 255)); v.push_back((uint8_t)(x >> 8)); }
void put32(std::vector<uint8_t> &v, uint32_t x) { for (int i = 0; i < 4; ++i) v.push_back((uint8_t)(x >> (8 * i))); }
void put64(std::vector<uint8_t> &v, uint64_t x) { for (int i = 0; i < 8; ++i) v.push_back((uint8_t)(x >> (8 * i))); }

// ------------------------------------------------------------------------------------------------------------------
// PNG
// ------------------------------------------------------------------------------------------------------------------
void be32(std::vector<uint8_t> &v, uint32_t x) { for (int i = 3; i >= 0; --i) v.push_back((uint8_t)(x >> (8 * i))); }

void png_chunk(std::vector<uint8_t> &out, const char *type, const uint8_t *data, size_t n)
{
    be32(out, (uint32_t)n);
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    if (n) out.insert(out.end(), data, data + n);
    be32(out, (uint32_t)crc32(0L, out.data() + at, (uInt)(n + 4)));
}

// ------------------------------------------------------------------------------------------------------------------
// JPEG (baseline sequential, YCbCr 4:2:0 or gray), libjpeg-compatible coefficients
// ------------------------------------------------------------------------------------------------------------------
const uint8_t kStdLumQ[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
                              14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                              49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kStdChrQ[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
                              47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                              99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
                             28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61,
                             54, 47, 55, 62, 63};
// Annex K.3 - K.6 Huffman tables (bits[1..16], values)
const uint8_t kDcLumBits[17] = {0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcChrBits[17] = {0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumBits[17] = {0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kAcLumVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91,
    0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a,
    0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53,
    0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79,
    0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
    0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9,
    0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2,
    0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kAcChrBits[17] = {0, 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kAcChrVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14,
    0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17,
    0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a,
    0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78,
    0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7,
    0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2,
    0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct HuffEnc {
    uint16_t code[256];
    uint8_t len[256];
    void build(const uint8_t *bits, const uint8_t *vals)
    {
        memset(len, 0, sizeof(len));
        memset(code, 0, sizeof(code));
        int k = 0;
        unsigned c = 0;
        for (int l = 1; l <= 16; ++l) {
            for (int i = 0; i < bits[l]; ++i) {
                code[vals[k]] = (uint16_t)c++;
                len[vals[k]] = (uint8_t)l;
                ++k;
            }
            c <<= 1;
        }
    }
};

struct BitSink {
    std::vector<uint8_t> &out;
    uint32_t acc = 0;
    int n = 0;
    explicit BitSink(std::vector<uint8_t> &o) : out(o) {}
    void put(unsigned code, int len)
    {
        acc = (acc << len) | (code & ((1u << len) - 1u));
        n += len;
        while (n >= 8) {
            const uint8_t b = (uint8_t)(acc >> (n - 8));
            out.push_back(b);
            if (b == 0xFF) out.push_back(0);
            n -= 8;
        }
        acc &= (1u << n) - 1u;
    }
    void flush()
    {
        if (n) put(0x7F, 8 - n);             // pad with one bits
    }
};

// jfdctint.c (accurate integer DCT, CONST_BITS 13, PASS1_BITS 2): output scaled up by 8
void fdct_islow(int *data)
{
    const int CONST_BITS = 13, PASS1_BITS = 2;
    const int F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633, F_1_501 = 12299,
              F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;
    auto descale = [](long x, int n) { return (int)((x + (1L << (n - 1))) >> n); };
    int *p = data;
    for (int r = 0; r < 8; ++r, p += 8) {
        const long t0 = p[0] + p[7], t7 = p[0] - p[7], t1 = p[1] + p[6], t6 = p[1] - p[6];
        const long t2 = p[2] + p[5], t5 = p[2] - p[5], t3 = p[3] + p[4], t4 = p[3] - p[4];
        const long t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
        p[0] = (int)((t10 + t11) * (1 << PASS1_BITS));
        p[4] = (int)((t10 - t11) * (1 << PASS1_BITS));
        long z1 = (t12 + t13) * F_0_541;
        p[2] = descale(z1 + t13 * F_0_765, CONST_BITS - PASS1_BITS);
        p[6] = descale(z1 + t12 * (-F_1_847), CONST_BITS - PASS1_BITS);
        z1 = t4 + t7;
        long z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
        const long z5 = (z3 + z4) * F_1_175;
        long a4 = t4 * F_0_298, a5 = t5 * F_2_053, a6 = t6 * F_3_072, a7 = t7 * F_1_501;
        z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
        z3 += z5; z4 += z5;
        p[7] = descale(a4 + z1 + z3, CONST_BITS - PASS1_BITS);
        p[5] = descale(a5 + z2 + z4, CONST_BITS - PASS1_BITS);
        p[3] = descale(a6 + z2 + z3, CONST_BITS - PASS1_BITS);
        p[1] = descale(a7 + z1 + z4, CONST_BITS - PASS1_BITS);
    }
    p = data;
    for (int c = 0; c < 8; ++c, ++p) {
        const long t0 = p[0] + p[56], t7 = p[0] - p[56], t1 = p[8] + p[48], t6 = p[8] - p[48];
        const long t2 = p[16] + p[40], t5 = p[16] - p[40], t3 = p[24] + p[32], t4 = p[24] - p[32];
        const long t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
        p[0] = descale(t10 + t11, PASS1_BITS);
        p[32] = descale(t10 - t11, PASS1_BITS);
        long z1 = (t12 + t13) * F_0_541;
        p[16] = descale(z1 + t13 * F_0_765, CONST_BITS + PASS1_BITS);
        p[48] = descale(z1 + t12 * (-F_1_847), CONST_BITS + PASS1_BITS);
        z1 = t4 + t7;
        long z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
        const long z5 = (z3 + z4) * F_1_175;
        long a4 = t4 * F_0_298, a5 = t5 * F_2_053, a6 = t6 * F_3_072, a7 = t7 * F_1_501;
        z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
        z3 += z5; z4 += z5;
        p[56] = descale(a4 + z1 + z3, CONST_BITS + PASS1_BITS);
        p[40] = descale(a5 + z2 + z4, CONST_BITS + PASS1_BITS);
        p[24] = descale(a6 + z2 + z3, CONST_BITS + PASS1_BITS);
        p[8] = descale(a7 + z1 + z4, CONST_BITS + PASS1_BITS);
    }
}

struct JpegTables {
    uint16_t q[2][64];        // natural order
    HuffEnc dc[2], ac[2];
};

// one 8x8 block: level shift, DCT, quantise (libjpeg: (|x| + q/2) / q on the x8-scaled output with divisor 8q),
// Huffman-code into the sink; returns the new DC predictor
int encode_block(const uint8_t *px, int pitch, const uint16_t *q, const HuffEnc &dc, const HuffEnc &ac, int pred, BitSink &bs)
{
    int blk[64];
    for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x) blk[y * 8 + x] = (int)px[y * pitch + x] - 128;
    fdct_islow(blk);
    int coef[64];
    for (int i = 0; i < 64; ++i) {
        const int qv = (int)q[i] << 3;
        int t = blk[i];
        if (t < 0) { t = -t; t += qv >> 1; t = t >= qv ? t / qv : 0; t = -t; }
        else { t += qv >> 1; t = t >= qv ? t / qv : 0; }
        coef[i] = t;
    }
    // DC
    int diff = coef[0] - pred, a = diff < 0 ? -diff : diff, nb = 0;
    while (a) { ++nb; a >>= 1; }
    bs.put(dc.code[nb], dc.len[nb]);
    if (nb) bs.put((unsigned)(diff < 0 ? diff - 1 : diff), nb);
    // AC
    int run = 0;
    for (int k = 1; k < 64; ++k) {
        int v = coef[kZigzag[k]];
        if (v == 0) { ++run; continue; }
        while (run > 15) { bs.put(ac.code[0xF0], ac.len[0xF0]); run -= 16; }
        int av = v < 0 ? -v : v, n2 = 0;
        while (av) { ++n2; av >>= 1; }
        const int sym = (run << 4) | n2;
        bs.put(ac.code[sym], ac.len[sym]);
        bs.put((unsigned)(v < 0 ? v - 1 : v), n2);
        run = 0;
    }
    if (run > 0) bs.put(ac.code[0], ac.len[0]);
    return coef[0];
}

}  // namespace

extern "C" {

int sr_encode_tiff_lzw(const uint8_t *h_img, int h, int w, int cn, int64_t stride, const char *path, int threads)
{
    if (!h_img || !path || h < 1 || w < 1 || (cn != 1 && cn != 3 && cn != 4) || stride < (int64_t)w * cn)
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_encode_tiff_lzw: bad arguments");
    const int rows_per_strip = 16;
    const size_t nstrips = ((size_t)h + rows_per_strip - 1) / rows_per_strip;
    const size_t rowlen = (size_t)w * cn;
    std::vector<std::vector<uint8_t>> strips(nstrips);
    parallel_for(nstrips, threads, [&](size_t s) {
        const int y0 = (int)s * rows_per_strip, y1 = std::min(h, y0 + rows_per_strip);
        std::vector<uint8_t> raw;
        const uint8_t *src = h_img + (size_t)y0 * stride;
        if ((size_t)stride != rowlen) {
            raw.resize((size_t)(y1 - y0) * rowlen);
            for (int y = y0; y < y1; ++y) memcpy(raw.data() + (size_t)(y - y0) * rowlen, h_img + (size_t)y * stride, rowlen);
            src = raw.data();
        }
        static thread_local LzwTable tab;
        strips[s].reserve((size_t)(y1 - y0) * rowlen / 2 + 64);
        lzw_encode_strip(src, (size_t)(y1 - y0) * rowlen, strips[s], tab);
        if (strips[s].size() & 1) strips[s].push_back(0);           // keep every strip offset even
    });
    uint64_t data_bytes = 0;
    for (auto &s : strips) data_bytes += s.size();
    const bool big = 16 + data_bytes + nstrips * 16 + 512 > 0xFFFF0000ull;     // classic TIFF offsets are 32-bit
    std::vector<uint8_t> head, tail;
    const uint64_t first = big ? 16 : 8;
    std::vector<uint64_t> offs(nstrips), cnts(nstrips);
    uint64_t pos = first;
    for (size_t s = 0; s < nstrips; ++s) { offs[s] = pos; cnts[s] = strips[s].size(); pos += strips[s].size(); }
    const uint64_t ifd_at = pos;                                                 // even: all parts are even-sized
    const int photometric = cn == 1 ? 1 : 2;
    struct Entry { uint16_t tag, type; uint64_t count; std::vector<uint64_t> vals; };
    std::vector<Entry> ents;
    const uint16_t LONGT = big ? 16 : 4;                                         // LONG8 in BigTIFF
    ents.push_back({256, 4, 1, {(uint64_t)w}});
    ents.push_back({257, 4, 1, {(uint64_t)h}});
    ents.push_back({258, 3, (uint64_t)cn, std::vector<uint64_t>((size_t)cn, 8)});
    ents.push_back({259, 3, 1, {5}});
    ents.push_back({262, 3, 1, {(uint64_t)photometric}});
    ents.push_back({273, LONGT, nstrips, offs});
    ents.push_back({277, 3, 1, {(uint64_t)cn}});
    ents.push_back({278, 4, 1, {(uint64_t)rows_per_strip}});
    ents.push_back({279, LONGT, nstrips, cnts});
    ents.push_back({284, 3, 1, {1}});
    if (cn == 4) ents.push_back({338, 3, 1, {2}});                               // ExtraSamples: unassociated alpha
    auto tsize = [](uint16_t t) { return t == 3 ? 2u : (t == 16 ? 8u : 4u); };
    // header
    if (big) { head = {'I', 'I', 43, 0, 8, 0, 0, 0}; put64(head, ifd_at); }
    else { head = {'I', 'I', 42, 0}; put32(head, (uint32_t)ifd_at); }
    // IFD + out-of-line values
    const size_t entry_bytes = big ? 20 : 12, inl = big ? 8 : 4;
    const uint64_t ifd_bytes = (big ? 8 : 2) + ents.size() * entry_bytes + (big ? 8 : 4);
    uint64_t extra_at = ifd_at + ifd_bytes;
    std::vector<uint8_t> extra;
    if (big) put64(tail, ents.size()); else put16(tail, (uint16_t)ents.size());
    for (auto &e : ents) {
        put16(tail, e.tag);
        put16(tail, e.type);
        if (big) put64(tail, e.count); else put32(tail, (uint32_t)e.count);
        const size_t bytes = (size_t)e.count * tsize(e.type);
        std::vector<uint8_t> v;
        for (auto x : e.vals) {
            if (e.type == 3) put16(v, (uint16_t)x);
            else if (e.type == 16) put64(v, x);
            else put32(v, (uint32_t)x);
        }
        if (bytes <= inl) {
            v.resize(inl, 0);
            tail.insert(tail.end(), v.begin(), v.end());
        } else {
            if (big) put64(tail, extra_at + extra.size()); else put32(tail, (uint32_t)(extra_at + extra.size()));
            extra.insert(extra.end(), v.begin(), v.end());
            if (extra.size() & 1) extra.push_back(0);
        }
    }
    if (big) put64(tail, 0); else put32(tail, 0);
    tail.insert(tail.end(), extra.begin(), extra.end());
    std::vector<const std::vector<uint8_t> *> parts{&head};
    for (auto &s : strips) parts.push_back(&s);
    parts.push_back(&tail);
    if (!write_file(path, parts)) return sr_set_error(SR_ERR_INVALID_ARG, "sr_encode_tiff_lzw: cannot write %s", path);
    return SR_OK;
}

int sr_encode_png(const uint8_t *h_img, int h, int w, int cn, int64_t stride, int level, const char *path, int threads)
{
    if (!h_img || !path || h < 1 || w < 1 || (cn != 1 && cn != 3 && cn != 4) || stride < (int64_t)w * cn || level < 0 || level > 9)
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_encode_png: bad arguments");
    const size_t rowlen = (size_t)w * cn, frow = rowlen + 1;
    // chunks of whole rows, about 1 MiB of filtered data each
    const int rows_per_chunk = (int)std::max<size_t>(1, ((size_t)1 << 20) / frow);
    const size_t nchunks = ((size_t)h + rows_per_chunk - 1) / rows_per_chunk;
    std::vector<std::vector<uint8_t>> comp(nchunks);
    std::vector<uLong> adler(nchunks), piece_crc(nchunks);
    std::vector<size_t> raw_len(nchunks);
    std::atomic<int> failed{0};
    parallel_for(nchunks, threads, [&](size_t c) {
        const int y0 = (int)c * rows_per_chunk, y1 = std::min(h, y0 + rows_per_chunk);
        std::vector<uint8_t> raw((size_t)(y1 - y0) * frow), cand(4 * rowlen);
        const std::vector<uint8_t> zero(rowlen, 0);
        for (int y = y0; y < y1; ++y) {
            // Adaptive filtering with libpng's default heuristic: try None / Sub / Up / Average / Paeth and keep the
            // filter whose output has the smallest sum of |signed byte| (the previous ROW is raw image data, so chunks
            // stay independent).  Level 0 (stored) skips the search.
            uint8_t *d = raw.data() + (size_t)(y - y0) * frow;
            const uint8_t *cur = h_img + (size_t)y * stride;
            const uint8_t *up = y > 0 ? h_img + (size_t)(y - 1) * stride : zero.data();
            int best = 0;
            if (level > 0) {
                uint64_t sum[5] = {0, 0, 0, 0, 0};
                uint8_t *f1 = cand.data(), *f2 = f1 + rowlen, *f3 = f2 + rowlen, *f4 = f3 + rowlen;
                // the first pixel has no left neighbour (a = c = 0); the rest runs in four simple, branch-free loops the
                // compiler vectorises (one fused loop with the Paeth selection in it stayed scalar: half the writer's time)
                const size_t c0 = (size_t)cn;
                for (size_t i = 0; i < c0 && i < rowlen; ++i) {
                    const int x = cur[i], b = up[i];
                    f1[i] = (uint8_t)x; f2[i] = (uint8_t)(x - b); f3[i] = (uint8_t)(x - (b >> 1)); f4[i] = (uint8_t)(x - b);
                }
                const uint8_t *ca = cur, *cb = up + c0, *cc = up, *cx = cur + c0;      // a, b, c, x of byte i = c0 + j
                const size_t m = rowlen > c0 ? rowlen - c0 : 0;
                uint8_t *o1 = f1 + c0, *o2 = f2 + c0, *o3 = f3 + c0, *o4 = f4 + c0;
                for (size_t j = 0; j < m; ++j) o1[j] = (uint8_t)(cx[j] - ca[j]);
                for (size_t j = 0; j < m; ++j) o2[j] = (uint8_t)(cx[j] - cb[j]);
                for (size_t j = 0; j < m; ++j) o3[j] = (uint8_t)(cx[j] - (uint8_t)(((unsigned)ca[j] + (unsigned)cb[j]) >> 1));
                for (size_t j = 0; j < m; ++j) {
                    const int a = ca[j], b = cb[j], c = cc[j];
                    const int pa = abs(b - c), pb = abs(a - c), pc = abs(a + b - 2 * c);
                    const int pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    o4[j] = (uint8_t)(cx[j] - pr);
                }
                auto cost = [rowlen](const uint8_t *v) {
                    uint32_t t = 0;                                            // <= 128 * rowlen: fits for rows below 2^25 bytes
                    for (size_t i = 0; i < rowlen; ++i) t += (uint32_t)(v[i] < 128 ? v[i] : 256 - v[i]);
                    return (uint64_t)t;
                };
                sum[0] = cost(cur); sum[1] = cost(f1); sum[2] = cost(f2); sum[3] = cost(f3); sum[4] = cost(f4);
                for (int k = 1; k < 5; ++k)
                    if (sum[k] < sum[best]) best = k;
            }
            d[0] = (uint8_t)best;
            memcpy(d + 1, best == 0 ? cur : cand.data() + (size_t)(best - 1) * rowlen, rowlen);
        }
        raw_len[c] = raw.size();
        adler[c] = adler32(adler32(0L, Z_NULL, 0), raw.data(), (uInt)raw.size());
        // Strategy per chunk: on filtered photographic data string matching finds almost nothing and its short matches cost
        // more bits than the literals they replace -- run-length matching + Huffman (Z_RLE) is 3x faster AND smaller there
        // (0.42 vs 0.47 on the benchmark's image); flat or repetitive content is the opposite.  A 64 KB sample from the middle
        // of the chunk is deflated both ways and the smaller one decides.
        int strategy = Z_DEFAULT_STRATEGY;
        if (level > 0 && raw.size() >= (size_t)8 << 10) {
            const size_t sn = std::min<size_t>(raw.size(), (size_t)64 << 10), so = (raw.size() - sn) / 2;
            std::vector<uint8_t> tmp;
            size_t got[2] = {0, 0};
            const int strat[2] = {Z_DEFAULT_STRATEGY, Z_RLE};
            for (int k = 0; k < 2; ++k) {
                z_stream t;
                memset(&t, 0, sizeof(t));
                if (deflateInit2(&t, level, Z_DEFLATED, -15, 8, strat[k]) != Z_OK) { failed = 1; return; }
                tmp.resize(deflateBound(&t, (uLong)sn) + 16);
                t.next_in = raw.data() + so;
                t.avail_in = (uInt)sn;
                t.next_out = tmp.data();
                t.avail_out = (uInt)tmp.size();
                if (deflate(&t, Z_FINISH) != Z_STREAM_END) failed = 1;
                got[k] = t.total_out;
                deflateEnd(&t);
            }
            if (got[1] <= got[0]) strategy = Z_RLE;
        }
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy) != Z_OK) { failed = 1; return; }   // raw deflate
        comp[c].resize(deflateBound(&zs, (uLong)raw.size()) + 16);
        zs.next_in = raw.data();
        zs.avail_in = (uInt)raw.size();
        zs.next_out = comp[c].data();
        zs.avail_out = (uInt)comp[c].size();
        const bool last = c + 1 == nchunks;
        const int rc = deflate(&zs, last ? Z_FINISH : Z_FULL_FLUSH);     // FULL flush: byte-aligned, no back-references across chunks
        if ((last && rc != Z_STREAM_END) || (!last && rc != Z_OK)) failed = 1;
        comp[c].resize(zs.total_out);
        deflateEnd(&zs);
        piece_crc[c] = crc32(crc32(0L, (const Bytef *)"IDAT", 4), comp[c].data(), (uInt)comp[c].size());   // CRC of the piece's own IDAT chunk
    });
    if (failed) return sr_set_error(SR_ERR_INVALID_ARG, "sr_encode_png: deflate failed");
    uLong ad = adler32(0L, Z_NULL, 0);
    for (size_t c = 0; c < nchunks; ++c) ad = c == 0 ? adler[0] : adler32_combine(ad, adler[c], (z_off_t)raw_len[c]);
    std::vector<uint8_t> head = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    uint8_t ihdr[13];
    for (int i = 0; i < 4; ++i) { ihdr[i] = (uint8_t)(w >> (24 - 8 * i)); ihdr[4 + i] = (uint8_t)(h >> (24 - 8 * i)); }
    ihdr[8] = 8;
    ihdr[9] = cn == 1 ? 0 : (cn == 3 ? 2 : 6);
    ihdr[10] = ihdr[11] = ihdr[12] = 0;
    png_chunk(head, "IHDR", ihdr, 13);
    // IDAT chunks (their boundaries are free: the zlib stream is the concatenation of all IDAT data): the 2-byte zlib header,
    // then ONE chunk per deflate piece -- its CRC was taken by the thread that compressed it, and the piece is written from
    // where it lies, so nothing of the compressed image is copied or re-read serially -- then the Adler-32.
    std::vector<uint8_t> zhead, ztail;
    const uint8_t zh[2] = {0x78, (uint8_t)(level >= 7 ? 0xDA : (level >= 6 ? 0x9C : (level >= 2 ? 0x5E : 0x01)))};
    png_chunk(zhead, "IDAT", zh, 2);
    const uint8_t adl[4] = {(uint8_t)(ad >> 24), (uint8_t)(ad >> 16), (uint8_t)(ad >> 8), (uint8_t)ad};
    png_chunk(ztail, "IDAT", adl, 4);
    png_chunk(ztail, "IEND", nullptr, 0);
    std::vector<std::vector<uint8_t>> pre(nchunks), post(nchunks);
    std::vector<const std::vector<uint8_t> *> parts = {&head, &zhead};
    for (size_t c = 0; c < nchunks; ++c) {
        if (comp[c].empty()) continue;
        if (comp[c].size() >= ((size_t)1 << 31)) return sr_set_error(SR_ERR_UNSUPPORTED, "sr_encode_png: a deflate piece of 2 GiB");
        be32(pre[c], (uint32_t)comp[c].size());
        pre[c].insert(pre[c].end(), {'I', 'D', 'A', 'T'});
        be32(post[c], (uint32_t)piece_crc[c]);
        parts.push_back(&pre[c]);
        parts.push_back(&comp[c]);
        parts.push_back(&post[c]);
    }
    parts.push_back(&ztail);
    if (!write_file(path, parts)) return sr_set_error(SR_ERR_INVALID_ARG, "sr_encode_png: cannot write %s", path);
    return SR_OK;
}

int sr_encode_jpeg(const uint8_t *h_img, int h, int w, int cn, int64_t stride, int quality, const char *path, int threads)
{
    if (!h_img || !path || h < 1 || w < 1 || (cn != 1 && cn != 3) || stride < (int64_t)w * cn || quality < 1 || quality > 100 ||
        h > 65535 || w > 65535)
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_encode_jpeg: bad arguments (1 or 3 channels, sides <= 65535, quality 1..100)");
    JpegTables T;
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;          // jpeg_quality_scaling
    for (int i = 0; i < 64; ++i) {
        long a = ((long)kStdLumQ[i] * scale + 50L) / 100L, b = ((long)kStdChrQ[i] * scale + 50L) / 100L;
        T.q[0][i] = (uint16_t)std::min(std::max(a, 1L), 255L);                    // force_baseline
        T.q[1][i] = (uint16_t)std::min(std::max(b, 1L), 255L);
    }
    T.dc[0].build(kDcLumBits, kDcVals); T.dc[1].build(kDcChrBits, kDcVals);
    T.ac[0].build(kAcLumBits, kAcLumVals); T.ac[1].build(kAcChrBits, kAcChrVals);
    const bool color = cn == 3;
    const int mcu = color ? 16 : 8;
    const int mcux = (w + mcu - 1) / mcu, mcuy = (h + mcu - 1) / mcu;
    std::vector<std::vector<uint8_t>> rows((size_t)mcuy);
    parallel_for((size_t)mcuy, threads, [&](size_t my) {
        const int pw = mcux * mcu;                                 // padded width (right edge replicated)
        std::vector<uint8_t> Y((size_t)mcu * pw), Cb, Cr, cb_f, cr_f;
        if (color) { cb_f.resize((size_t)16 * pw); cr_f.resize((size_t)16 * pw); Cb.resize((size_t)8 * (pw / 2)); Cr.resize((size_t)8 * (pw / 2)); }
        for (int r = 0; r < mcu; ++r) {
            const int sy = std::min((int)my * mcu + r, h - 1);      // bottom edge replicated
            const uint8_t *src = h_img + (size_t)sy * stride;
            for (int x = 0; x < pw; ++x) {
                const int sx = std::min(x, w - 1);
                if (!color) { Y[(size_t)r * pw + x] = src[sx]; continue; }
                const int R = src[3 * sx], G = src[3 * sx + 1], B = src[3 * sx + 2];
                // jccolor.c, SCALEBITS 16: FIX(x) = (int)(x * 65536 + 0.5)
                Y[(size_t)r * pw + x] = (uint8_t)((19595 * R + 38470 * G + 7471 * B + 32768) >> 16);
                cb_f[(size_t)r * pw + x] = (uint8_t)((-11059 * R - 21709 * G + 32768 * B + (128 << 16) + 32767) >> 16);
                cr_f[(size_t)r * pw + x] = (uint8_t)((32768 * R - 27439 * G - 5329 * B + (128 << 16) + 32767) >> 16);
            }
        }
        if (color) {
            const int cw = pw / 2;
            for (int r = 0; r < 8; ++r) {                            // jcsample.c h2v2_downsample: bias 1, 2, 1, 2, ...
                int bias = 1;
                for (int x = 0; x < cw; ++x) {
                    const size_t a = (size_t)(2 * r) * pw + 2 * x, b = a + pw;
                    Cb[(size_t)r * cw + x] = (uint8_t)((cb_f[a] + cb_f[a + 1] + cb_f[b] + cb_f[b + 1] + bias) >> 2);
                    Cr[(size_t)r * cw + x] = (uint8_t)((cr_f[a] + cr_f[a + 1] + cr_f[b] + cr_f[b + 1] + bias) >> 2);
                    bias ^= 3;
                }
            }
        }
        std::vector<uint8_t> &out = rows[my];
        out.reserve((size_t)pw * mcu / 4 + 64);
        BitSink bs(out);
        int pd[3] = {0, 0, 0};
        for (int mx = 0; mx < mcux; ++mx) {
            if (color) {
                for (int by = 0; by < 2; ++by)
                    for (int bx = 0; bx < 2; ++bx)
                        pd[0] = encode_block(Y.data() + (size_t)(by * 8) * pw + mx * 16 + bx * 8, pw, T.q[0], T.dc[0], T.ac[0], pd[0], bs);
                pd[1] = encode_block(Cb.data() + mx * 8, pw / 2, T.q[1], T.dc[1], T.ac[1], pd[1], bs);
                pd[2] = encode_block(Cr.data() + mx * 8, pw / 2, T.q[1], T.dc[1], T.ac[1], pd[2], bs);
            } else {
                pd[0] = encode_block(Y.data() + mx * 8, pw, T.q[0], T.dc[0], T.ac[0], pd[0], bs);
            }
        }
        bs.flush();
    });
    std::vector<uint8_t> hd;
    auto m16 = [&](int v) { hd.push_back((uint8_t)(v >> 8)); hd.push_back((uint8_t)(v & 255)); };
    hd.push_back(0xFF); hd.push_back(0xD8);
    const uint8_t app0[] = {0xFF, 0xE0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
    hd.insert(hd.end(), app0, app0 + sizeof(app0));
    for (int t = 0; t < (color ? 2 : 1); ++t) {
        hd.push_back(0xFF); hd.push_back(0xDB); m16(67); hd.push_back((uint8_t)t);
        for (int k = 0; k < 64; ++k) hd.push_back((uint8_t)T.q[t][kZigzag[k]]);
    }
    hd.push_back(0xFF); hd.push_back(0xC0); m16(8 + 3 * (color ? 3 : 1)); hd.push_back(8); m16(h); m16(w);
    hd.push_back((uint8_t)(color ? 3 : 1));
    if (color) { const uint8_t c[] = {1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1}; hd.insert(hd.end(), c, c + 9); }
    else { const uint8_t c[] = {1, 0x11, 0}; hd.insert(hd.end(), c, c + 3); }
    auto dht = [&](int cls, int id, const uint8_t *bits, const uint8_t *vals) {
        int n = 0;
        for (int l = 1; l <= 16; ++l) n += bits[l];
        hd.push_back(0xFF); hd.push_back(0xC4); m16(2 + 1 + 16 + n); hd.push_back((uint8_t)((cls << 4) | id));
        hd.insert(hd.end(), bits + 1, bits + 17);
        hd.insert(hd.end(), vals, vals + n);
    };
    dht(0, 0, kDcLumBits, kDcVals); dht(1, 0, kAcLumBits, kAcLumVals);
    if (color) { dht(0, 1, kDcChrBits, kDcVals); dht(1, 1, kAcChrBits, kAcChrVals); }
    hd.push_back(0xFF); hd.push_back(0xDD); m16(4); m16(mcux);                   // DRI: one restart interval = one MCU row
    hd.push_back(0xFF); hd.push_back(0xDA); m16(6 + 2 * (color ? 3 : 1)); hd.push_back((uint8_t)(color ? 3 : 1));
    if (color) { const uint8_t c[] = {1, 0x00, 2, 0x11, 3, 0x11}; hd.insert(hd.end(), c, c + 6); }
    else { const uint8_t c[] = {1, 0x00}; hd.insert(hd.end(), c, c + 2); }
    hd.push_back(0); hd.push_back(63); hd.push_back(0);
    std::vector<std::vector<uint8_t>> rst((size_t)mcuy);
    std::vector<const std::vector<uint8_t> *> parts{&hd};
    for (int my = 0; my < mcuy; ++my) {
        parts.push_back(&rows[(size_t)my]);
        if (my + 1 < mcuy) rst[(size_t)my] = {0xFF, (uint8_t)(0xD0 + (my & 7))};
        else rst[(size_t)my] = {0xFF, 0xD9};
        parts.push_back(&rst[(size_t)my]);
    }
    if (!write_file(path, parts)) return sr_set_error(SR_ERR_INVALID_ARG, "sr_encode_jpeg: cannot write %s", path);
    return SR_OK;
}

}  // extern "C"

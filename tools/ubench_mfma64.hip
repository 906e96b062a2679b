// v_mfma_f64_16x16x4_f64 on gfx950: operand layout check, issue rate, and how far it overlaps with fp64 VALU work of the same
// wave / of another wave on the same SIMD (the question behind moving the assessment's Gaussian passes to the matrix pipe).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o ubench_mfma64 tools/ubench_mfma64.hip   (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4_t __attribute__((ext_vector_type(4)));
#define N_IT 2048
#define REP 8

// MODE 0: 4 MFMA per rep (4 independent accumulators); 1: NV fp64 FMAs per rep; 2: both in one wave, interleaved;
// 3: waves 0-3 of a 512-thread block issue the MFMAs, waves 4-7 the FMAs (one of each per SIMD)
template <int MODE, int NV>
__global__ __launch_bounds__(512) void k(double *out, int seed)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a = 1.0 + 1e-9 * lane, b = 1.0 - 1e-9 * lane;
    d4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = seed + lane + i;
    const double kk = 0.999999, cc = 1e-9;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (do_m) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
                if (do_v && MODE == 2) {
#pragma unroll
                    for (int i = 0; i < NV / 4; ++i) v[i & 7] = fma(v[i & 7], kk, cc);
                }
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
                if (do_v && MODE == 2) {
#pragma unroll
                    for (int i = 0; i < NV / 4; ++i) v[(i + 2) & 7] = fma(v[(i + 2) & 7], kk, cc);
                }
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
                if (do_v && MODE == 2) {
#pragma unroll
                    for (int i = 0; i < NV / 4; ++i) v[(i + 4) & 7] = fma(v[(i + 4) & 7], kk, cc);
                }
                c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
                if (do_v && MODE == 2) {
#pragma unroll
                    for (int i = 0; i < NV / 4; ++i) v[(i + 6) & 7] = fma(v[(i + 6) & 7], kk, cc);
                }
            } else if (do_v) {
#pragma unroll
                for (int i = 0; i < NV; ++i) v[i & 7] = fma(v[i & 7], kk, cc);
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s + c0.x + c1.y + c2.z + c3.w;
}

template <int MODE, int NV> void run(const char *name, double *d, int threads)
{
    printf("%-44s", name);
    for (int w = 1; w <= 2; ++w) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int blocks = 256 * w;
        hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(threads), 0, 0, d, 1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(threads), 0, 0, d, 1);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        // SIMD cycles per rep (= 4 MFMA and / or NV FMAs of one wave [pair])
        const double reps_per_simd = (double)w * N_IT * REP;
        printf("  %d blk/CU: %7.3f ms  %7.1f cyc/rep", w, ms, ms * 1e-3 * 2.4e9 / reps_per_simd);
    }
    printf("\n");
}

// layout check: D = A(16x4) B(4x16) with A[i][k] = i + 0.25 k, B[k][j] = (k + 1) * (j + 1)
__global__ void k_layout(double *out)
{
    const int lane = threadIdx.x;
    const double a = (lane & 15) + 0.25 * (lane >> 4);                 // A: row = lane & 15, k = lane >> 4
    const double b = ((lane >> 4) + 1.0) * ((lane & 15) + 1.0);        // B: col = lane & 15, k = lane >> 4
    d4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) out[v * 64 + lane] = c[v];
}

int main()
{
    double *d; (void)hipMalloc(&d, (size_t)512 * 512 * 8);
    double h[256];
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int v = 0; v < 4; ++v)
        for (int lane = 0; lane < 64; ++lane) {
            const int row = (lane >> 4) + 4 * v, col = lane & 15;     // the guide's map: col = lane & 15, row = (lane >> 4) + 4 reg
            double e = 0;
            for (int kq = 0; kq < 4; ++kq) e += (row + 0.25 * kq) * ((kq + 1.0) * (col + 1.0));
            if (h[v * 64 + lane] != e) ++bad;
        }
    printf("layout D[row = (lane >> 4) + 4 reg][col = lane & 15], A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15]: %s (%d mismatches)\n",
           bad ? "WRONG" : "confirmed", bad);
    run<0, 0>("4 mfma_f64_16x16x4 / rep, 4 waves/blk", d, 256);
    run<1, 8>("8 v_fma_f64 / rep, 4 waves/blk", d, 256);
    run<1, 16>("16 v_fma_f64 / rep, 4 waves/blk", d, 256);
    run<1, 32>("32 v_fma_f64 / rep, 4 waves/blk", d, 256);
    run<2, 8>("same wave: 4 mfma + 8 fma / rep", d, 256);
    run<2, 16>("same wave: 4 mfma + 16 fma / rep", d, 256);
    run<2, 32>("same wave: 4 mfma + 32 fma / rep", d, 256);
    run<2, 48>("same wave: 4 mfma + 48 fma / rep", d, 256);
    run<3, 16>("two waves/SIMD: one 4 mfma, one 16 fma", d, 512);
    run<3, 32>("two waves/SIMD: one 4 mfma, one 32 fma", d, 512);
    run<3, 48>("two waves/SIMD: one 4 mfma, one 48 fma", d, 512);
    run<3, 64>("two waves/SIMD: one 4 mfma, one 64 fma", d, 512);
    return 0;
}

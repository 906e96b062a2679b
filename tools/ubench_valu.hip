// instruction-throughput microbenchmarks (gfx950): cycles per wave-instruction per SIMD
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o ubench_valu tools/ubench_valu.hip   (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define N_IT 4096
#define REP 16
template <int OP>
__global__ __launch_bounds__(256) void k(double *out, int seed)
{
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, kk = 0.999999, c = 1e-9;
    int i0 = seed + threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
    __shared__ unsigned char sm[4096];
    sm[threadIdx.x] = (unsigned char)seed;
    __syncthreads();
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (OP == 0) { a0 = fma(a0, kk, c); a1 = fma(a1, kk, c); a2 = fma(a2, kk, c); a3 = fma(a3, kk, c); }
            if (OP == 1) { a0 = a0 + c; a1 = a1 + c; a2 = a2 + c; a3 = a3 + c; }
            if (OP == 2) { a0 = a0 * kk; a1 = a1 * kk; a2 = a2 * kk; a3 = a3 * kk; }
            if (OP == 3) { asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %1\n v_mul_lo_u32 %3, %3, %1\n v_mul_lo_u32 %4, %4, %1" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(seed | 3)); i1 ^= 0; }
            if (OP == 4) { asm volatile("v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %2, %2, %1\n v_mul_u32_u24 %3, %3, %1\n v_mul_u32_u24 %4, %4, %1" : "+v"(i0), "+v"(i2), "+v"(i1), "+v"(i3) : "v"(seed | 3)); }
            if (OP == 5) { asm volatile("v_cvt_f64_i32 %0, %4\n v_cvt_f64_i32 %1, %5\n v_cvt_f64_i32 %2, %6\n v_cvt_f64_i32 %3, %7" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(i0), "v"(i1), "v"(i2), "v"(i3)); i0 += 1; }
            if (OP == 6) { asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
            if (OP == 7) { asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(seed)); }
            if (OP == 8) { int t0, t1, t2, t3; asm volatile("ds_read_u8 %0, %4\n ds_read_u8 %1, %4 offset:1\n ds_read_u8 %2, %4 offset:2\n ds_read_u8 %3, %4 offset:3\n s_waitcnt lgkmcnt(0)" : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "v"((int)threadIdx.x + (r & 7) * 4)); i0 += t0; i1 += t1; i2 += t2; i3 += t3; }
            if (OP == 9) { int t0, t1, t2, t3; asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:4\n ds_read_b32 %2, %4 offset:8\n ds_read_b32 %3, %4 offset:12\n s_waitcnt lgkmcnt(0)" : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "v"((int)threadIdx.x * 4 + (r & 7) * 16)); i0 += t0; i1 += t1; i2 += t2; i3 += t3; }
            if (OP == 10) { asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(kk)); }
            if (OP == 11) { float *f = (float *)&a0; asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(0.5f)); }
            if (OP == 12) { asm volatile("v_mad_u32_u24 %0, %0, %1, %1\n v_mad_u32_u24 %2, %2, %1, %1\n v_mad_u32_u24 %3, %3, %1, %1\n v_mad_u32_u24 %4, %4, %1, %1" : "+v"(i0), "+v"(i2), "+v"(i1), "+v"(i3) : "v"(seed | 3)); }
            if (OP == 13) { asm volatile("v_mad_u64_u32 %0, vcc, %4, %4, %0\n v_mad_u64_u32 %1, vcc, %4, %4, %1\n v_mad_u64_u32 %2, vcc, %4, %4, %2\n v_mad_u64_u32 %3, vcc, %4, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed | 3) : "vcc"); }
            if (OP == 14) { asm volatile("v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3)); }
            if (OP == 15) { asm volatile("v_dot4_u32_u8 %0, %0, %1, %0\n v_dot4_u32_u8 %2, %2, %1, %2\n v_dot4_u32_u8 %3, %3, %1, %3\n v_dot4_u32_u8 %4, %4, %1, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(seed | 3)); }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + (double)(i0 + i1 + i2 + i3);
}
template <int OP> void run(const char *name, double *d)
{
    // one launch per occupancy: w blocks of 4 waves per CU = w waves per SIMD (1, 2, 4, 8); cycles of a SIMD per wave-instruction
    // issued to it (N_IT x REP x 4 instructions per wave).  4.0 at every occupancy = a wave's VALU instruction holds its SIMD
    // for four cycles whatever the type (the guide's `v_fma_f32 2 cyc` is a rate this stream of dependent-free
    // instructions does not reach on gfx950).
    printf("%-16s", name);
    for (int w = 1; w <= 8; w *= 2) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int blocks = 256 * w;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double winstr = (double)blocks * 4 * N_IT * REP * 4;       // wave-instructions
        const double per_simd = winstr / 1024.0;
        printf("  %d w/SIMD: %7.3f ms %5.2f cyc", w, ms, ms * 1e-3 * 2.4e9 / per_simd);
    }
    printf("   (cycles per wave-instruction and SIMD at 2.4 GHz)\n");
}
int main()
{
    double *d; (void)hipMalloc(&d, 256 * 8 * 256 * 8);
    run<0>("v_fma_f64", d); run<1>("v_add_f64", d); run<2>("v_mul_f64", d); run<3>("v_mul_lo_u32", d);
    run<4>("v_mul_u32_u24", d); run<5>("v_cvt_f64_i32", d); run<6>("v_rcp_f64", d); run<7>("v_add_u32", d);
    run<8>("ds_read_u8", d); run<9>("ds_read_b32", d); run<10>("v_pk_fma_f32", d); run<11>("v_fma_f32", d);
    run<12>("v_mad_u32_u24", d); run<13>("v_mad_u64_u32", d); run<14>("v_cvt_f32_i32", d); run<15>("v_dot4_u32_u8", d);
    return 0;
}

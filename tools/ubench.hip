// Micro-benchmarks: issue cost (cycles per wave-instruction per SIMD) of the fp64 instructions the
// lnprob kernels are made of, on gfx950.  8 independent chains per lane, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

enum { I_FMA, I_ADD, I_MUL, I_RCP, I_RSQ, I_SQRT, I_LDEXP, I_CVT_F64_I32, I_MIN, I_FMA32, I_AND, I_BFE, I_RNDNE, I_FREXP, I_CVT_I32_F64, I_NCASE };
const char* NAMES[] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_ldexp_f64", "v_cvt_f64_i32", "v_min_f64", "v_fma_f32", "v_and_b32", "v_bfe_u32", "v_rndne_f64", "v_frexp_mant_f64", "v_cvt_i32_f64"};

template <int I>
__global__ __launch_bounds__(256) void k(double* out, int iters, unsigned long long* clk) {
    double a[8];
    for (int j = 0; j < 8; ++j) a[j] = 1.0 + 1e-3 * (threadIdx.x + j);
    float f[8];
    for (int j = 0; j < 8; ++j) f[j] = 1.0f + 1e-3f * (threadIdx.x + j);
    int n[8];
    for (int j = 0; j < 8; ++j) n[j] = threadIdx.x + j;
    const double c = 0.999999, d = 1e-7;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (I == I_FMA) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(c), "v"(d));
            if (I == I_ADD) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[j]) : "v"(d));
            if (I == I_MUL) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[j]) : "v"(c));
            if (I == I_RCP) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[j]));
            if (I == I_RSQ) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[j]));
            if (I == I_SQRT) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a[j]));
            if (I == I_LDEXP) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a[j]) : "v"(n[j] & 1));
            if (I == I_CVT_F64_I32) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[j]) : "v"(n[j]));
            if (I == I_MIN) asm volatile("v_min_f64 %0, %0, %1" : "+v"(a[j]) : "v"(c));
            if (I == I_FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[j]) : "v"(0.999f), "v"(1e-6f));
            if (I == I_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(n[j]) : "v"(0x7fffffff));
            if (I == I_BFE) asm volatile("v_bfe_u32 %0, %0, 1, 30" : "+v"(n[j]));
            if (I == I_RNDNE) asm volatile("v_rndne_f64 %0, %0" : "+v"(a[j]));
            if (I == I_FREXP) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(a[j]));
            if (I == I_CVT_I32_F64) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n[j]) : "v"(a[j]));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int j = 0; j < 8; ++j) s += a[j] + f[j] + n[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int I>
int run(int wavesPerSimd) {
    const int blocks = 256 * wavesPerSimd;   // 256-thread blocks = 4 waves = 1 per SIMD
    const int iters = 20000;
    double* out; unsigned long long* clk;
    CHK(hipMalloc(&out, (size_t)blocks * 256 * 8)); CHK(hipMalloc(&clk, (size_t)blocks * 16));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<I>, dim3(blocks), dim3(256), 0, 0, out, 100, clk);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<I>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * blocks);
    CHK(hipMemcpy(h.data(), clk, (size_t)blocks * 16, hipMemcpyDeviceToHost));
    double ghz = 0; for (int b = 0; b < blocks; ++b) ghz += (double)h[2 * b] / (double)h[2 * b + 1] * 0.1; ghz /= blocks;
    double cyc = 0; for (int b = 0; b < blocks; ++b) cyc += (double)h[2 * b]; cyc /= blocks;
    // per SIMD: wavesPerSimd waves each issue iters*8 instructions during `cyc` shader cycles
    double per = cyc / ((double)iters * 8 * wavesPerSimd);
    printf("%-18s waves/SIMD %d  %.3f ms  clock %.2f GHz  %.2f cycles/wave-instr/SIMD\n", NAMES[I], wavesPerSimd, ms, ghz, per);
    hipFree(out); hipFree(clk);
    return 0;
}

template <int I> int all() { return run<I>(1) || run<I>(2) || run<I>(4); }

int main() {
    if (all<I_FMA>() || all<I_ADD>() || all<I_MUL>() || all<I_RCP>() || all<I_RSQ>() || all<I_SQRT>() || all<I_LDEXP>() ||
        all<I_CVT_F64_I32>() || all<I_MIN>() || all<I_FMA32>() || all<I_AND>() || all<I_BFE>() || all<I_RNDNE>() || all<I_FREXP>() || all<I_CVT_I32_F64>()) return 1;
    return 0;
}

// How many workgroups of a given LDS / VGPR footprint does a CU really hold?  Each workgroup records the CU it ran on
// (HW_ID) and spins until all have started or a timeout: the maximum number seen together per CU is the residency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int LDSB>
__global__ __launch_bounds__(256) void census(int* cu_of, unsigned long long* t0, unsigned long long* t1) {
    __shared__ char buf[LDSB];
    buf[threadIdx.x] = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        cu_of[blockIdx.x] = (int)(((xcc & 0xf) << 16) | (id & 0xffff0));      // se/sh/cu bits + xcc
        t0[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        unsigned long long t = t0[blockIdx.x];
        while (__builtin_amdgcn_s_memrealtime() - t < 2000) {}                // 20 us at 100 MHz
        t1[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    }
    __syncthreads();
    if (buf[threadIdx.x] == 0) cu_of[0] = -1;
}
template <int LDSB>
void run(const char* name) {
    const int n = 4096;
    int* d; unsigned long long *a, *b;
    hipMalloc(&d, n * 4); hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
    hipLaunchKernelGGL(census<LDSB>, dim3(n), dim3(256), 0, 0, d, a, b);
    hipDeviceSynchronize();
    std::vector<int> cu(n); std::vector<unsigned long long> s(n), e(n);
    hipMemcpy(cu.data(), d, n * 4, hipMemcpyDeviceToHost); hipMemcpy(s.data(), a, n * 8, hipMemcpyDeviceToHost); hipMemcpy(e.data(), b, n * 8, hipMemcpyDeviceToHost);
    // max overlap per CU id
    int best = 0;
    for (int i = 0; i < n; i += 7) {
        int c = 0;
        for (int j = 0; j < n; ++j) if (cu[j] == cu[i] && s[j] <= s[i] && e[j] > s[i]) ++c;
        if (c > best) best = c;
    }
    int occ = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, census<LDSB>, 256, 0);
    printf("%s: LDS %d B per workgroup: occupancy API %d, observed resident together on one CU %d\n", name, LDSB, occ, best);
    hipFree(d); hipFree(a); hipFree(b);
}
int main() {
    run<16384>("16K"); run<40000>("40K"); run<53000>("53K"); run<65536>("64K"); run<75136>("75K"); run<81920>("80K");
    return 0;
}

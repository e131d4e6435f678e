// Does this box's firmware preload kernel arguments into SGPRs (gfx950 "kernarg preload")?  Built twice: with and without
// -mllvm -amdgpu-kernarg-preload-count=4.  Each wave stamps s_memtime at its start, loads one double through the pointer that
// is the kernel's FIRST argument, and stamps again when the value has arrived; then reads the rest of its (cold) arguments.
//   hipcc -O3 --offload-arch=gfx950 [-mllvm -amdgpu-kernarg-preload-count=4] -o probe preload_probe.hip && ./probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
struct Big { double a[64]; };
__global__ void probe(const double* __restrict__ p, int n, int m, Big b, unsigned long long* out, double* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    double v = p[(blockIdx.x * 64 + threadIdx.x) % n];
    asm volatile("" : "+v"(v));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const double w = v + b.a[m];
    asm volatile("" :: "v"(w));
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = t1 - t0;
        out[2 * blockIdx.x + 1] = t2 - t0;
    }
    if (w == 12345.678) sink[0] = w;
}
__global__ void touch(double* p, int n, double x) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = x + i;
}
int main() {
    const int n = 1 << 16, nb = 512, reps = 200;
    double *p, *sink;
    unsigned long long* out;
    hipMalloc(&p, n * 8); hipMalloc(&sink, 8); hipMalloc(&out, nb * 16);
    Big b; for (int i = 0; i < 64; ++i) b.a[i] = i;
    std::vector<unsigned long long> h(2 * nb), a, c;
    for (int r = 0; r < reps; ++r) {
        touch<<<n / 256, 256>>>(p, n, (double)r);
        probe<<<nb, 64>>>(p, n, r % 64, b, out, sink);
        hipMemcpy(h.data(), out, nb * 16, hipMemcpyDeviceToHost);
        if (r < 10) continue;
        for (int i = 0; i < nb; ++i) { a.push_back(h[2 * i]); c.push_back(h[2 * i + 1]); }
    }
    std::sort(a.begin(), a.end()); std::sort(c.begin(), c.end());
    printf("first-argument load arrived: median %llu p10 %llu p90 %llu cycles; + rest of the arguments: median %llu\n",
           a[a.size() / 2], a[a.size() / 10], a[a.size() * 9 / 10], c[c.size() / 2]);
    return 0;
}

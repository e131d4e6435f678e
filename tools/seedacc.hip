// accuracy of the raw v_rcp_f64 / v_rsq_f64 seeds and of the refined versions
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* rcp, double* rsq, double* rcp3, double* rsq3, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = x[i];
    double y0 = __builtin_amdgcn_rcp(d);
    rcp[i] = y0;
    double e = fma(-d, y0, 1.0);
    rcp3[i] = fma(y0, fma(e, e, e), y0);          // cubic Newton
    double s = 1.0 + d * 1e3;
    double z0 = __builtin_amdgcn_rsq(s);
    rsq[i] = z0;
    double t = s * z0;
    double ee = fma(-t, z0, 1.0);
    double p = fma(0.375, ee, 0.5);
    rsq3[i] = fma(z0, p * ee, z0);                // cubic
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n);
    for (int i = 0; i < n; ++i) x[i] = 1e-3 + (double)i / n * 0.999 + 1e-9 * (i % 7);
    double *dx, *a, *b, *c, *d;
    hipMalloc(&dx, n * 8); hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&c, n * 8); hipMalloc(&d, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, a, b, c, d, n);
    std::vector<double> ha(n), hb(n), hc(n), hd(n);
    hipMemcpy(ha.data(), a, n * 8, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), c, n * 8, hipMemcpyDeviceToHost); hipMemcpy(hd.data(), d, n * 8, hipMemcpyDeviceToHost);
    double m1 = 0, m2 = 0, m3 = 0, m4 = 0;
    for (int i = 0; i < n; ++i) {
        long double dd = x[i], s = 1.0L + dd * 1e3L;
        m1 = fmax(m1, fabs((double)(ha[i] * dd - 1.0L)));
        m3 = fmax(m3, fabs((double)(hc[i] * dd - 1.0L)));
        long double r = 1.0L / sqrtl(s);
        m2 = fmax(m2, fabs((double)(hb[i] / r - 1.0L)));
        m4 = fmax(m4, fabs((double)(hd[i] / r - 1.0L)));
    }
    printf("raw rcp rel err %.3e (2^%.1f)  raw rsq %.3e (2^%.1f)  cubic rcp %.3e  cubic rsq %.3e\n", m1, log2(m1), m2, log2(m2), m3, m4);
    return 0;
}

// developer micro-test (GPU box): accuracy of v_rsq_f64 and of one / two Newton steps on it, against a long-double host reference
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_rsq.hip -o tools/bin/ubench_rsq
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double* x, double* y0, double* y1, double* y2, double* g1, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double y = __builtin_amdgcn_rsq(v);
    y0[i] = y;
    // one Newton step: y <- y (1.5 - 0.5 v y^2), written as y + y * (0.5 - 0.5 v y^2)
    double h = 0.5 * y, g = v * y;
    double r = fma(-h, g, 0.5);
    double ya = fma(y, r, y);
    y1[i] = ya;
    // Goldschmidt pair once + the companion (what one iteration of fast_sqrt_rsqrt's loop gives for 1/sqrt: 2 h1)
    double gg = fma(g, r, g), hh = fma(h, r, h);
    g1[i] = hh + hh;
    double r2 = fma(-hh, gg, 0.5);
    double yb = fma(hh + hh, r2, hh + hh);
    y2[i] = yb;
}
int main() {
    const int n = 1 << 22;
    std::vector<double> x(n);
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> u(-40.0, 40.0);
    for (auto& v : x) v = std::exp2(u(rng)) * (1.0 + (rng() >> 11) * 0x1.0p-53);
    double *dx, *d0, *d1, *d2, *d3;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&d3, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, d3, n);
    std::vector<double> y0(n), y1(n), y2(n), g1(n);
    hipMemcpy(y0.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(y1.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(y2.data(), d2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(g1.data(), d3, n * 8, hipMemcpyDeviceToHost);
    long double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
    for (int i = 0; i < n; ++i) {
        const long double ref = 1.0L / sqrtl((long double)x[i]);
        e0 = fmaxl(e0, fabsl(y0[i] - ref) / ref); e1 = fmaxl(e1, fabsl(y1[i] - ref) / ref);
        e2 = fmaxl(e2, fabsl(y2[i] - ref) / ref); e3 = fmaxl(e3, fabsl(g1[i] - ref) / ref);
    }
    printf("max relative error: v_rsq_f64 %.3Le, + one Newton step %.3Le, Goldschmidt 2 h1 %.3Le, + second step %.3Le (eps/2 = 1.11e-16)\n", e0, e1, e3, e2);
    return 0;
}

// developer tool: in-situ cycles of the NP = 32 factorisation (factor_tiles32: fp64 MFMA tiles), 1..8 waves per CU, 1..8 waves per CU
#include <hip/hip_runtime.h>
#include <cstdio>
#include "osot_qp_core.h"
using namespace osot;

template <int WHICH>
__global__ void __launch_bounds__(64, 2) factor_bench(long long* out, double* sink, int n, int reps) {
    OSOT_DYNAMIC_LDS(smem);
    double* base = reinterpret_cast<double*>(smem);
    constexpr int S = WaveCtx<32>::S;
    WaveCtx<32> w;
    w.c = threadIdx.x & 31; w.h = threadIdx.x >> 5; w.n = n;
    w.M1 = base; w.M2 = base + 32 * S; w.V = base + 2 * 32 * S;
    double acc = 0.0;
    long long total = 0;
    long long tt[4] = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        double Hc[16];
#pragma unroll
        for (int ii = 0; ii < 16; ++ii) {
            const int i = 2 * ii + w.h;
            Hc[ii] = (i < n && w.c < n) ? ((i == w.c) ? 40.0 + r : 1.0 / (1.0 + (i > w.c ? i - w.c : w.c - i))) : 0.0;
        }
        if (WHICH == 1) {
            const int ta = threadIdx.x & 15, tq = threadIdx.x >> 4;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int i = 16 * (t >> 3) + tq + 4 * (t & 3), cc = 16 * ((t >> 2) & 1) + ta;
                Hc[t] = (i == cc) ? 40.0 + r : 1.0 / (1.0 + (i > cc ? i - cc : cc - i));
            }
        }
        double x;
        wave_sync();
        const long long t0 = clock64();
        const int st = factor_tiles32<true>(w, Hc, 1.0 + w.c, x, tt);
        const long long t1 = clock64();
        total += t1 - t0;
        acc += x + st;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = total / reps;
    if (threadIdx.x == 0 && blockIdx.x == 0 && WHICH == 1) for (int i = 0; i < 4; ++i) out[gridDim.x + i] = tt[i] / reps;
    sink[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main() {
    long long* out; double* sink;
    const int maxb = 256 * 8;
    hipMalloc(&out, (maxb + 8) * sizeof(long long));
    hipMalloc(&sink, maxb * 64 * sizeof(double));
    const size_t lds = (2 * 32 * 33 + 4 * 32) * sizeof(double);
    for (int wpc : {1, 2, 4, 8}) {
        const int grid = 256 * wpc;
        static long long h[maxb];
        for (int which = 1; which < 2; ++which) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const int reps = 64;
            for (int it = 0; it < 2; ++it) {
                if (it == 1) hipEventRecord(e0);
                if (which) factor_bench<1><<<grid, 64, lds>>>(out, sink, 32, reps);
                else factor_bench<0><<<grid, 64, lds>>>(out, sink, 32, reps);
                if (it == 1) hipEventRecord(e1);
            }
            hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, out, grid * sizeof(long long), hipMemcpyDeviceToHost);
            double m = 0; for (int i = 0; i < grid; ++i) m += h[i];
            if (which) { long long t4[4]; hipMemcpy(t4, out + grid, sizeof(t4), hipMemcpyDeviceToHost);
                printf("    panels %lld  L^-1 store %lld  forward %lld  backward %lld\n", t4[0], t4[1], t4[2], t4[3]); }
            printf("waves/CU %d: %s (n = 32) %.0f cycles = %.0f per column; kernel %.1f us for %d factorisations per wave -> >= %.2f G ticks/s\n", wpc,
                   which ? "factor_tiles32" : "factor_loop32 ", m / grid, m / grid / 32, ms * 1e3, reps, (m / grid) * reps / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}

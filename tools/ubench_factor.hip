// developer tool: in-situ cycles of the NP = 32 building blocks (factor_tiles32, nullspace_equalities32), 1..8 waves per CU
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 -Iopensot_amd/csrc -Iinclude tools/ubench_factor.hip -o tools/bin/ubench_factor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include "osot_qp_core.h"
using namespace osot;

constexpr int kRowsCap = 64;
__device__ inline void make_ctx(WaveCtx<32>& w, char* smem, int n) {
    double* base = reinterpret_cast<double*>(smem);
    constexpr int S = WaveCtx<32>::S;
    w.c = threadIdx.x & 31; w.h = threadIdx.x >> 5; w.n = n;
    w.M1 = base; w.M2 = base + WaveCtx<32>::M1_DOUBLES; w.V = w.M2 + 32 * S;
    w.rlo = w.V + 4 * 32; w.rup = w.rlo + kRowsCap;
    w.rptr = reinterpret_cast<unsigned long long*>(w.rup + kRowsCap);
    w.rowstate = reinterpret_cast<int*>(w.rptr + kRowsCap);
    w.eqlist = w.rowstate + kRowsCap;
    w.rsrc = reinterpret_cast<signed char*>(w.eqlist + kRowsCap);
}
constexpr size_t kLds = (WaveCtx<32>::M1_DOUBLES + 32 * 33 + 4 * 32 + 2 * kRowsCap + kRowsCap) * 8 + kRowsCap * 9 + 64;

__global__ void __launch_bounds__(64, 2) factor_bench(long long* out, double* sink, int n, int reps) {
    OSOT_DYNAMIC_LDS(smem);
    WaveCtx<32> w; make_ctx(w, smem, n);
    double acc = 0.0;
    long long total = 0;
    long long tt[4] = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        double Hc[16];
        const int ta = threadIdx.x & 15, tq = threadIdx.x >> 4;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int i = 16 * (t >> 3) + tq + 4 * (t & 3), cc = 16 * ((t >> 2) & 1) + ta;
            Hc[t] = (i == cc) ? 40.0 + r : 1.0 / (1.0 + (i > cc ? i - cc : cc - i));
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
    if (threadIdx.x == 0 && blockIdx.x == 0) for (int i = 0; i < 4; ++i) out[gridDim.x + i] = tt[i] / reps;
    sink[blockIdx.x * 64 + threadIdx.x] = acc;
}

// E: [n_eq][32] rows in global memory (one copy per workgroup so that the loads behave like the solver's)
__global__ void __launch_bounds__(64, 2) nullspace_bench(long long* out, double* sink, const double* E, int n_eq, int n, int reps) {
    OSOT_DYNAMIC_LDS(smem);
    WaveCtx<32> w; make_ctx(w, smem, n);
    const double* Eb = E + (size_t)blockIdx.x * n_eq * 32;
    for (int r = threadIdx.x; r < n_eq; r += 64) { w.rptr[r] = reinterpret_cast<unsigned long long>(Eb + r * 32); w.eqlist[r] = r; }
    w.safe_row = reinterpret_cast<unsigned long long>(Eb);
    wave_sync();
    double acc = 0.0;
    long long total = 0;
    long long prof[PH_COUNT];
    for (int i = 0; i < PH_COUNT; ++i) prof[i] = 0;
    for (int r = 0; r < reps; ++r) {
        double x;
        wave_sync();
        const long long t0 = clock64();
        const int me = nullspace_equalities32<true>(w, n_eq, 1.0 + 0.01 * w.c, 0.1 * w.c, 0.5, x, prof);
        const long long t1 = clock64();
        total += t1 - t0;
        acc += x + me;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = total / reps;
    if (threadIdx.x == 0 && blockIdx.x == 0) for (int i = 0; i < 4; ++i) out[gridDim.x + i] = prof[PH_EQ_D + i] / reps;
    sink[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main() {
    long long* out; double* sink; double* E;
    const int maxb = 256 * 8;
    const int n_eq = 27;
    hipMalloc(&out, (maxb + 8) * sizeof(long long));
    hipMalloc(&sink, maxb * 64 * sizeof(double));
    hipMalloc(&E, (size_t)maxb * n_eq * 32 * sizeof(double));
    {
        std::vector<double> h((size_t)maxb * n_eq * 32);
        std::mt19937_64 g(1); std::normal_distribution<double> nd;
        for (auto& v : h) v = nd(g);
        hipMemcpy(E, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice);
    }
    for (int wpc : {1, 4, 8}) {
        const int grid = 256 * wpc;
        static long long h[maxb];
        for (int which = 0; which < 2; ++which) {
            const int reps = 32;
            for (int it = 0; it < 2; ++it) {
                if (which) factor_bench<<<grid, 64, kLds>>>(out, sink, 32, reps);
                else nullspace_bench<<<grid, 64, kLds>>>(out, sink, E, n_eq, 32, reps);
            }
            hipDeviceSynchronize();
            hipMemcpy(h, out, grid * sizeof(long long), hipMemcpyDeviceToHost);
            double m = 0; for (int i = 0; i < grid; ++i) m += h[i];
            long long t4[4]; hipMemcpy(t4, out + grid, sizeof(t4), hipMemcpyDeviceToHost);
            if (which) printf("waves/CU %d: factor_tiles32 %.0f cycles  (panels %lld  L^-1 store %lld  backward %lld)\n", wpc, m / grid, t4[0], t4[1], t4[3]);
            else printf("waves/CU %d: nullspace_equalities32 (27 rows) %.0f cycles  (load %lld  Gauss-Jordan %lld  Z+MGS %lld  projection %lld)\n", wpc, m / grid, t4[0], t4[1], t4[2], t4[3]);
        }
    }
    return 0;
}

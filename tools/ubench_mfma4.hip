// developer tool (GPU box): lane layout and dependent latency of v_mfma_f64_4x4x4_4b_f64 (four 4x4x4 blocks, 4 passes), the candidate
// for a SHORT-LATENCY sum network (the kernels are bound by their dependent chains; a 32-lane fp64 sum through DPP is ~137 cycles).
// Probe 1: a = 2^lane, b = 1 -> D(lane) = sum of the A lanes of its row: which four lanes make up a row.
// Probe 2: a = 1, b = 2^lane -> D(lane) = sum of the B lanes of its column.
// Then dependent-chain latencies: DPP 16-lane sum, mfma4x4x4 pair, mixed forms.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Iopensot_amd/csrc tools/ubench_mfma4.hip -o tools/bin/ubench_mfma4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "osot_team.h"
using namespace osot;

__global__ void __launch_bounds__(64) probe(double* out) {
    const int l = threadIdx.x;
    const double p = ldexp(1.0, l % 52);      // (2^l, folded into the mantissa range per 52: lanes l and l + 52 collide -- disambiguated by probe on l/2)
    out[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(ldexp(1.0, l & 31), 1.0, 0.0, 0, 0, 0);          // rows, lanes mod 32
    out[64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(ldexp(1.0, l >> 1), 1.0, 0.0, 0, 0, 0);     // rows, lanes / 2
    out[128 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, ldexp(1.0, l & 31), 0.0, 0, 0, 0);    // columns, lanes mod 32
    out[192 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, ldexp(1.0, l >> 1), 0.0, 0, 0, 0);    // columns, lanes / 2
    (void)p;
}

#define REP 64
#define TIME(idx, ...)                                                            \
    {                                                                             \
        __builtin_amdgcn_s_waitcnt(0);                                            \
        __builtin_amdgcn_sched_barrier(0);                                        \
        const long long t0 = clock64();                                           \
        __builtin_amdgcn_sched_barrier(0);                                        \
        _Pragma("unroll") for (int r = 0; r < REP; ++r) { __VA_ARGS__; }          \
        __builtin_amdgcn_sched_barrier(0);                                        \
        asm volatile("s_nop 0" ::"v"(x));                                         \
        __builtin_amdgcn_s_waitcnt(0);                                            \
        const long long t1 = clock64();                                           \
        if (threadIdx.x == 0) out[idx] = (double)(t1 - t0) / REP;                 \
    }

__global__ void __launch_bounds__(64) lat(double* out, double seed) {
    double x = seed + threadIdx.x * 1e-3;
    TIME(0, asm volatile("" ::: "memory"));
    TIME(1, x = row16_sum(x) * 0.0625);
    TIME(2, x = colsum<32>(x) * 0.03125);
    TIME(3, x = __builtin_amdgcn_mfma_f64_4x4x4f64(x, 0.25, 0.0, 0, 0, 0));
    TIME(4, x = __builtin_amdgcn_mfma_f64_4x4x4f64(__builtin_amdgcn_mfma_f64_4x4x4f64(x, 0.25, 0.0, 0, 0, 0), 0.25, 0.0, 0, 0, 0));
    TIME(5, x = __builtin_amdgcn_mfma_f64_4x4x4f64(0.25, x, 0.0, 0, 0, 0));
    TIME(6, x = __builtin_amdgcn_mfma_f64_4x4x4f64(0.25, __builtin_amdgcn_mfma_f64_4x4x4f64(x, 0.25, 0.0, 0, 0, 0), 0.0, 0, 0, 0));
    TIME(7, { v4f64 z = {0.0, 0.0, 0.0, 0.0}; v4f64 d = __builtin_amdgcn_mfma_f64_16x16x4f64(x, 0.015625, z, 0, 0, 0); x = d[0] + d[1] + d[2] + d[3]; });
    TIME(8, { double a, b; swap16_pair(x, a, b); x = (a + b) * 0.5; });
    TIME(9, { double a, b; swap32_pair(x, a, b); x = (a + b) * 0.5; });
    TIME(10, x = quad_sum(x) * 0.25);
    out[15] = x;
}

int main() {
    double* d; hipMalloc(&d, sizeof(double) * 512);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // decode: rows -- lanes mod 32 from probe 0, lanes / 2 from probe 1 (a lane l is in the set iff bit (l & 31) of the first AND bit (l >> 1) of the second)
    for (int which = 0; which < 2; ++which) {
        printf("%s\n", which == 0 ? "A lanes summed into D(lane) with b = 1 [row sets]:" : "B lanes summed into D(lane) with a = 1 [column sets]:");
        for (int l = 0; l < 64; ++l) {
            const unsigned long long m0 = (unsigned long long)h[128 * which + l], m1 = (unsigned long long)h[128 * which + 64 + l];
            printf("  D[%2d] <-", l);
            for (int s = 0; s < 64; ++s) if (((m0 >> (s & 31)) & 1ull) && ((m1 >> (s >> 1)) & 1ull)) printf(" %d", s);
            printf("\n");
        }
    }
    hipLaunchKernelGGL(lat, dim3(1), dim3(64), 0, 0, d + 256, 1.0);
    double t[16]; hipMemcpy(t, d + 256, sizeof(t), hipMemcpyDeviceToHost);
    const char* nm[] = {"empty", "row16_sum (4 DPP stages)", "colsum<32> (DPP + permlane16)", "mfma4x4x4 (a = x)", "mfma4x4x4 twice (a, a)", "mfma4x4x4 (b = x)",
                        "mfma4x4x4 twice (a then b)", "mfma16x16x4 + 3 adds", "swap16 + add", "swap32 + add", "quad_sum (2 DPP stages)"};
    for (int i = 0; i < 11; ++i) printf("%-34s %.1f cycles\n", nm[i], t[i] - (i ? t[0] : 0.0));
    return 0;
}

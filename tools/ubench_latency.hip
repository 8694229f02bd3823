// developer tool: dependent-chain latencies (shader cycles, s_memtime) of the primitives the QP kernels are
// built from, one wavefront alone on a CU.  Build + run:  hipcc --offload-arch=gfx950 -O3 -std=c++17
//   -Iopensot_amd/csrc tools/ubench_latency.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include "osot_team.h"

using namespace osot;

#define REP 64
#define TIME(name, idx, ...)                                                      \
    {                                                                             \
        __builtin_amdgcn_s_waitcnt(0);                                            \
        __builtin_amdgcn_sched_barrier(0);                                        \
        const long long t0 = clock64();                                           \
        __builtin_amdgcn_sched_barrier(0);                                        \
        _Pragma("unroll") for (int r = 0; r < REP; ++r) { __VA_ARGS__; }                 \
        __builtin_amdgcn_sched_barrier(0);                                        \
        asm volatile("s_nop 0" ::"v"(x), "v"(y));                                 \
        __builtin_amdgcn_s_waitcnt(0);                                            \
        const long long t1 = clock64();                                           \
        if (threadIdx.x == 0) out[idx] = (double)(t1 - t0) / REP;                 \
    }

__global__ void __launch_bounds__(64) ubench(double* out, double* sink, double seed) {
    __shared__ double lds[64 * 33];
    const int lane = threadIdx.x;
    double x = seed + lane * 1e-3, y = 1.0 + lane * 1e-4;
    lds[lane] = x;
    lds[lane + 64] = y;
    __syncthreads();
    // 0: empty timing overhead
    TIME("empty", 0, asm volatile("" ::: "memory"));
    // 1: dependent v_fma_f64
    TIME("fma_f64 dep", 1, x = fma(x, 1.0000001, y));
    // 2: 4 independent fma chains (per-instruction issue cost)
    {
        double a0 = x, a1 = y, a2 = x + 1, a3 = y + 1;
        TIME("fma_f64 x4 indep", 2, a0 = fma(a0, 1.0000001, y); a1 = fma(a1, 1.0000001, y); a2 = fma(a2, 1.0000001, y); a3 = fma(a3, 1.0000001, y));
        x = a0 + a1 + a2 + a3;
    }
    // 3: v_rsq_f64 dependent
    TIME("rsq_f64 dep", 3, x = __builtin_amdgcn_rsq(x) + 2.0);
    // 4: v_rcp_f64 dependent
    TIME("rcp_f64 dep", 4, x = __builtin_amdgcn_rcp(x) + 2.0);
    // 5: fast_sqrt_rsqrt dependent
    TIME("fast_sqrt_rsqrt", 5, { double s, rs; fast_sqrt_rsqrt(x, s, rs); x = s + rs + 1.0; });
    // 6: fast_rcp dependent
    TIME("fast_rcp", 6, x = fast_rcp(x) + 2.0);
    // 7: bcast (2 x v_readlane) -> VALU
    TIME("bcast readlane+use", 7, x = bcast(x, 17) * 1.0000001 + y);
    // 8: permlane32 swap (from_half)
    TIME("from_half<32>", 8, x = from_half<32>(x, 1) + y);
    // 9: DPP quad xor + add
    TIME("dpp xor1 + add", 9, x += dpp_f64<DPP_XOR1>(x));
    // 10: colsum<32>
    TIME("colsum<32>", 10, x = colsum<32>(x) * 0.03 + y);
    // 11: colsum<64>
    TIME("colsum<64>", 11, x = colsum<64>(x) * 0.015 + y);
    // 12: colargmin<32>
    {
        int p = lane;
        TIME("colargmin<32>", 12, { colargmin<32>(x, p); x += y; p += lane; });
        x += p;
    }
    // 13: LDS write -> wave_sync -> read (other lane's slot)
    TIME("ds_write->sync->ds_read", 13, { lds[lane] = x; wave_sync(); x = lds[(lane + 1) & 63] + y; wave_sync(); });
    // 14: LDS uniform read dependent through address
    {
        int idx = lane & 1;
        TIME("ds_read dep (addr chain)", 14, { idx = (int)lds[idx] & 1; });
        x += idx;
    }
    // 15: __shfl (ds_bpermute) dependent
    TIME("__shfl f64", 15, x = __shfl(x, (lane + 5) & 63) + y);
    // 16: 16 independent LDS reads + 16 fma (batched)
    {
        TIME("16 ds_read + 16 fma", 16, {
            double a[16];
            _Pragma("unroll") for (int t = 0; t < 16; ++t) a[t] = lds[33 * t + lane];
            _Pragma("unroll") for (int t = 0; t < 16; ++t) x = fma(a[t], 1e-9, x);
        });
    }
    // 17: one factor step's arithmetic chain (pivot broadcast -> sqrt/rsqrt -> scale -> LDS write -> sync -> 16 reads -> 32 fma)
    {
        double H[16], L[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) { H[t] = x + t; L[t] = y + t; }
        TIME("factor step (model)", 17, {
            double piv = bcast(H[r & 15], r & 31);
            double sq, rs;
            fast_sqrt_rsqrt(fabs(piv) + 1.0, sq, rs);
            const double hjc = from_half<32>(H[r & 15], r & 1);
            const double lcj = hjc * rs;
            lds[(lane & 31) * 33 + (r & 31)] = lcj;
            const double lin = from_half<32>(L[r & 15], r & 1) * rs;
            wave_sync();
            double li[16];
            _Pragma("unroll") for (int t = 0; t < 16; ++t) li[t] = lds[(2 * t + (lane >> 5)) * 33 + (r & 31)];
            _Pragma("unroll") for (int t = 0; t < 16; ++t) { H[t] = fma(-li[t], lcj, H[t]); L[t] = fma(-li[t], lin, L[t]); }
        });
#pragma unroll
        for (int t = 0; t < 16; ++t) x += H[t] + L[t];
    }
    // ---- throughput (independent operations; 16 accumulators kept live through the asm fence) ----
    {
        double a[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) a[t] = x + t;
#define KEEP16() asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), \
                             "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]))
        TIME("16 indep fma_f64", 19, { _Pragma("unroll") for (int t = 0; t < 16; ++t) a[t] = fma(a[t], 1.0000001, y); KEEP16(); });
        TIME("16 indep v_mov_b64 (rotate)", 20, { const double e = a[0]; _Pragma("unroll") for (int t = 0; t < 15; ++t) a[t] = a[t + 1]; a[15] = e; KEEP16(); });
        TIME("16 indep cndmask f64", 21, { _Pragma("unroll") for (int t = 0; t < 16; ++t) a[t] = (lane == r + t) ? y : a[t]; KEEP16(); });
        TIME("16 indep __shfl f64", 22, { const int src = (r & 31) + (lane & 32); _Pragma("unroll") for (int t = 0; t < 16; ++t) a[t] = __shfl(a[t], src, 64); KEEP16(); });
        TIME("16 indep bcast (readlane x2 + use)", 23, { _Pragma("unroll") for (int t = 0; t < 16; ++t) a[t] = bcast(a[t], r & 63) + y; KEEP16(); });
        TIME("16 ds_read_b64 uniform + sync", 24, { _Pragma("unroll") for (int t = 0; t < 16; ++t) a[t] = lds[33 * t + (r & 31)]; KEEP16(); });
        TIME("16 ds_write_b64 + sync", 25, { _Pragma("unroll") for (int t = 0; t < 16; ++t) lds[33 * t + lane] = a[t]; wave_sync(); });
        TIME("16 indep from_half", 26, { _Pragma("unroll") for (int t = 0; t < 16; ++t) a[t] = from_half<32>(a[t], r & 1); KEEP16(); });
        TIME("colmax<32> + first_lane_equal", 27, { const double m = colmax<32>(a[0]); a[1] += first_lane_equal(a[0], m); a[0] += y; KEEP16(); });
#pragma unroll
        for (int t = 0; t < 16; ++t) x += a[t];
    }
    // 18: global load dependent (L2 hit), pointer chase through sink
    {
        const double* p = sink;
        long long off = 0;
        TIME("global_load dep (L2)", 18, { off = (long long)p[off] & 7; });
        x += off;
    }
    sink[lane + 64] = x + y;
}

int main() {
    double *out, *sink;
    hipMalloc(&out, 64 * sizeof(double));
    hipMalloc(&sink, 256 * sizeof(double));
    hipMemset(sink, 0, 256 * sizeof(double));
    hipMemset(out, 0, 64 * sizeof(double));
    for (int it = 0; it < 3; ++it) ubench<<<1, 64>>>(out, sink, 3.0);
    hipDeviceSynchronize();
    double h[64];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"empty", "fma_f64 dependent", "4 independent fma_f64", "rsq_f64 dependent", "rcp_f64 dependent",
                           "fast_sqrt_rsqrt", "fast_rcp", "bcast (readlane x2) + use", "from_half<32> (permlane32_swap)",
                           "dpp xor1 + add", "colsum<32>", "colsum<64>", "colargmin<32>", "ds_write->sync->ds_read",
                           "ds_read dependent", "__shfl f64 (ds_bpermute)", "16 ds_read + 16 dependent fma",
                           "factor step (model)", "global_load dependent (L2)", "16 independent fma_f64", "16 v_mov_b64 (rotate)",
                           "16 cndmask f64", "16 independent __shfl f64", "16 independent bcast + add", "16 ds_read_b64 uniform",
                           "16 ds_write_b64 + sync", "16 independent from_half", "colmax<32> + first_lane_equal"};
    for (int i = 0; i < 28; ++i) printf("%-36s %8.1f cycles\n", names[i], h[i]);
    return 0;
}

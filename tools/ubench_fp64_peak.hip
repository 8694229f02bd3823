// developer tool (GPU box): the FP64 roof of THIS chip, measured -- SURVEY 8(d) quotes AMD's public 78.6 TFLOP/s (vector = matrix) and
// says "verify with a micro-benchmark"; /opt/skills/guides/MI355X_MICROARCH.md has no FP64 row.  Two dependent-free loops on every CU:
//   v_fma_f64 ............. 16 independent accumulator chains per lane (2 flop per lane and instruction)
//   v_mfma_f64_16x16x4 .... 8 independent accumulator tiles per wavefront (2 * 16 * 16 * 4 flop per instruction)
// each at 1, 2, 4 and 8 wavefronts per SIMD, timed with HIP events over ~50 ms; prints a JSON line with the best of each.
// Build + run:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_fp64_peak.hip -o tools/bin/ubench_fp64_peak && tools/bin/ubench_fp64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double v4f64 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) fma_loop(double* out, int iters, double b, double c) {
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = 1.0 + 1e-3 * (threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = __builtin_fma(a[i], b, c);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    if (s == 123.456) out[blockIdx.x * blockDim.x + threadIdx.x] = s;      // (never true: keeps the chains alive)
}

__global__ void __launch_bounds__(256) mfma_loop(double* out, int iters, double x, double y) {
    v4f64 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = {0.0, 0.0, 0.0, 0.0};
    const double a = x + 1e-6 * threadIdx.x, b = y;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
static double run(K kernel, int blocks, int iters, double flop_per_thread_iter, double* d_out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, iters / 8, 0.999999, 1e-9);      // warm-up
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, iters, 0.999999, 1e-9);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return (double)blocks * 256.0 * iters * flop_per_thread_iter / (best * 1e-3) / 1e12;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    double* d_out;
    hipMalloc(&d_out, sizeof(double) * 256 * (size_t)cus * 8 * 4);
    double best_fma = 0.0, best_mfma = 0.0;
    int wf = 0, wm = 0;
    std::printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"runs\": [", p.gcnArchName, cus, p.clockRate / 1000);
    bool first = true;
    for (int waves_per_simd : {1, 2, 4, 8}) {
        const int blocks = cus * waves_per_simd;        // one block = 4 wavefronts = one per SIMD of a CU
        // fma: 64 fma per iteration and lane, 2 flop each; mfma: 32 instructions per iteration and wavefront, 2048 flop each = 32 * 2048 / 64 per lane
        const double tf = run(fma_loop, blocks, 40000, 64.0 * 2.0, d_out);
        const double tm = run(mfma_loop, blocks, 8000, 32.0 * 2048.0 / 64.0, d_out);
        std::printf("%s{\"waves_per_simd\": %d, \"v_fma_f64_tflops\": %.2f, \"v_mfma_f64_16x16x4_tflops\": %.2f}", first ? "" : ", ", waves_per_simd, tf, tm);
        first = false;
        if (tf > best_fma) { best_fma = tf; wf = waves_per_simd; }
        if (tm > best_mfma) { best_mfma = tm; wm = waves_per_simd; }
    }
    std::printf("], \"v_fma_f64_tflops\": %.2f, \"v_fma_f64_waves_per_simd\": %d, \"v_mfma_f64_16x16x4_tflops\": %.2f, \"v_mfma_waves_per_simd\": %d, "
                "\"spec_tflops\": 78.6}\n", best_fma, wf, best_mfma, wm);
    hipFree(d_out);
    return 0;
}

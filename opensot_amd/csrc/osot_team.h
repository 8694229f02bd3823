// osot_team.h -- team-of-lanes primitives for gfx950 (wave64).
//
// One QP instance is solved by a TEAM of T lanes (T = 32: two instances per wavefront, T = 64: one).
// Lane t of a team owns element t of every length-n vector; matrices live in the team's LDS slice.
// A team never spans wavefronts, so every collective below is a wave-level data movement
// (ds_bpermute / DPP) and the LDS hand-offs between lanes need no s_barrier: LDS instructions of one
// wave are executed in order, the fences only stop the compiler from reordering them.
//
// tests/emu/ provides a host-side lock-step emulation of exactly this interface (same names) so the
// kernel bodies in osot_qp_core.h can be exercised on a machine without a GPU; that header is test
// infrastructure and is never part of the product build.
#pragma once
#include <hip/hip_runtime.h>

// LDS declarations (the emulation in tests/emu maps these onto host memory)
#define OSOT_DYNAMIC_LDS(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define OSOT_STATIC_LDS(type, name, count) __shared__ type name[count]

namespace osot {

// make this team's earlier LDS writes visible to its later LDS reads (other lanes of the same wave)
__device__ __forceinline__ void team_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int T>
__device__ __forceinline__ double team_sum(double v) {
#pragma unroll
    for (int m = T / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, T);
    return v;
}

template <int T>
__device__ __forceinline__ double team_bcast(double v, int src) {
    return __shfl(v, src, T);
}

template <int T>
__device__ __forceinline__ int team_bcast_i(int v, int src) {
    return __shfl(v, src, T);
}

// minimum of v over the team with the payload of the (lowest-lane) minimiser; every lane gets both
template <int T>
__device__ __forceinline__ void team_argmin(double& v, int& payload) {
#pragma unroll
    for (int m = T / 2; m >= 1; m >>= 1) {
        double ov = __shfl_xor(v, m, T);
        int op = __shfl_xor(payload, m, T);
        bool take = (ov < v) || (ov == v && op < payload);
        v = take ? ov : v;
        payload = take ? op : payload;
    }
}

template <int T>
__device__ __forceinline__ bool team_any(bool p) {
    int v = p ? 1 : 0;
#pragma unroll
    for (int m = T / 2; m >= 1; m >>= 1) v |= __shfl_xor(v, m, T);
    return v != 0;
}

// value held by lane (t+1) of the team (lane T-1 receives its own)
template <int T>
__device__ __forceinline__ double team_shift_down(double v) { return __shfl_down(v, 1, T); }
template <int T>
__device__ __forceinline__ int team_shift_down_i(int v) { return __shfl_down(v, 1, T); }

}  // namespace osot

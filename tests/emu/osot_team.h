// tests/emu/osot_team.h -- TEST INFRASTRUCTURE ONLY: the host lock-step twin of
// opensot_amd/csrc/osot_team.h (same names, same semantics), picked up by include order when the
// kernel headers are compiled for the emulator (tests/emu/emu_driver.cpp).
#pragma once
#include <hip/hip_runtime.h>

#define OSOT_DYNAMIC_LDS(name) char* name = emu::dyn_smem_ptr()
#define OSOT_STATIC_LDS(type, name, count) static type name[count]

namespace osot {
inline void team_sync() { int z = 0; emu::exchange(&z, nullptr, sizeof(int), 0, emu::S().team_width); }
template <int T> inline double team_sum(double v) { for (int m = T / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, T); return v; }
template <int T> inline double team_bcast(double v, int src) { return __shfl(v, src, T); }
template <int T> inline int team_bcast_i(int v, int src) { return __shfl(v, src, T); }
template <int T> inline void team_argmin(double& v, int& payload) {
    for (int m = T / 2; m >= 1; m >>= 1) {
        double ov = __shfl_xor(v, m, T);
        int op = __shfl_xor(payload, m, T);
        bool take = (ov < v) || (ov == v && op < payload);
        v = take ? ov : v; payload = take ? op : payload;
    }
}
template <int T> inline bool team_any(bool p) { int v = p; for (int m = T / 2; m >= 1; m >>= 1) v |= __shfl_xor(v, m, T); return v != 0; }
template <int T> inline double team_shift_down(double v) { return __shfl_down(v, 1, T); }
template <int T> inline int team_shift_down_i(int v) { return __shfl_down(v, 1, T); }
}  // namespace osot

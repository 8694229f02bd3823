// tests/emu/osot_team.h -- TEST INFRASTRUCTURE ONLY: the host lock-step twin of
// opensot_amd/csrc/osot_team.h (same names, same semantics), picked up through the include path when the
// kernel headers are compiled for the emulator (tests/emu/emu_driver.cpp).  Every collective is a
// rendezvous of the whole 64-lane wave, so a missing wave_sync() between an LDS write and another lane's
// read produces a wrong result here even though real hardware executes the wave in lock-step.
// What this emulation does NOT model: the exec mask.  A cross-lane read (ds_bpermute / __shfl, DPP) issued inside a branch
// that only some lanes take reads ZEROS (bpermute) or stale registers (DPP, v_readlane) from the lanes that are masked off on
// the hardware, while the rendezvous here hands over every lane's value -- `x = cond ? shift_down(v) : 0` compiled as two
// masked arms cost a GPU-only failure in sym_eig32 once.  Rule for the kernels: cross-lane reads are issued by ALL lanes,
// outside any lane-dependent select or branch; the GPU suite (tests -m gpu) is what checks it.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include <cstdio>
#include <cstdlib>

#define OSOT_DYNAMIC_LDS(name) char* name = emu::dyn_smem_ptr()
#define OSOT_STATIC_LDS(type, name, count) static type name[count]
#define OSOT_KERNARG_PTR(type, first_param) (&(first_param))
#define OSOT_ALWAYS_INLINE_CALL
#define OSOT_KEEP16(a) do { } while (0)
#define OSOT_KEEP12(a, o) do { } while (0)
#define OSOT_GLOBAL_F64(addr) (reinterpret_cast<const double*>(addr))

namespace osot {
inline int emu_lane() { return emu::S().cur; }
inline int phys_lane() { return emu_lane(); }
// lanes per half of a padded size (32 -> 32; 56 and 64 -> 64: see WaveCtx in osot_qp_core.h)
constexpr int LWOF(int np) { return np <= 32 ? 32 : 64; }
inline void wave_sync() { int z = 0, out[64]; emu::allgather(&z, out, sizeof(int)); }
inline void sched_fence() {}
inline void workgroup_fence() {}
inline void wave_priority_by_rank(unsigned, unsigned) {}
inline int launder_i(int v) { return v; }
inline int launder_s(int v) { return v; }
inline int uniform_i(int v) { int out[64]; emu::allgather(&v, out, sizeof(int)); return out[0]; }
// uniform_d / uniform_b DECLARE a value wave-uniform (on the device: v_readfirstlane, i.e. lane 0's value for everybody, silently).
// Here every lane's value is at hand, so the declaration is CHECKED: a lane that disagrees with lane 0 (bit pattern; NaNs of
// different payload count as different) aborts the test run with a message.  uniform_i is also used as "take the first
// lane's value" on purpose (colargmin payloads), so it is not checked.
inline void emu_uniform_violation(const char* what, int lane, double a, double b) {
    fprintf(stderr, "emu: %s declared wave-uniform, but lane %d holds %.17g and lane 0 holds %.17g\n", what, lane, a, b);
    abort();
}
inline double uniform_d(double v) {
    double out[64]; emu::allgather(&v, out, sizeof(double));
    for (int i = 1; i < 64; ++i) if (std::memcmp(&out[i], &out[0], sizeof(double)) != 0) emu_uniform_violation("uniform_d", i, out[i], out[0]);
    return out[0];
}
inline bool uniform_b(bool p) {
    int v = p ? 1 : 0; int out[64]; emu::allgather(&v, out, sizeof(int));
    for (int i = 1; i < 64; ++i) if (out[i] != out[0]) emu_uniform_violation("uniform_b", i, out[i], out[0]);
    return out[0] != 0;
}

inline unsigned long long wave_ballot(bool p) {
    int v = p ? 1 : 0, all[64]; emu::allgather(&v, all, sizeof(int));
    unsigned long long m = 0; for (int i = 0; i < 64; ++i) if (all[i]) m |= (1ull << i);
    return m;
}
inline int lanes_below(unsigned long long mask) {
    const int l = emu_lane();
    return __builtin_popcountll(l ? (mask & ((~0ull) >> (64 - l))) : 0ull);
}
struct v4f64 {
    double v[4];
    double& operator[](int i) { return v[i]; }
    const double& operator[](int i) const { return v[i]; }
};
inline v4f64 mfma_f64_16x16x4(double a, double b, v4f64 c) {
    double aa[64], bb[64];
    emu::allgather(&a, aa, sizeof(double));
    emu::allgather(&b, bb, sizeof(double));
    const int l = emu_lane(), col = l & 15;
    v4f64 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = (l >> 4) + 4 * r;
        double acc = c[r];
        for (int k = 0; k < 4; ++k) acc = std::fma(aa[row + 16 * k], bb[col + 16 * k], acc);
        d[r] = acc;
    }
    return d;
}
inline double rowgroup_sum(double v) {
    double all[64]; emu::allgather(&v, all, sizeof(double));
    const int a = emu_lane() & 15;
    return (all[a] + all[a + 16]) + (all[a + 32] + all[a + 48]);
}
inline double quad_sum(double v) {
    double all[64]; emu::allgather(&v, all, sizeof(double));
    const int q0 = emu_lane() & ~3;
    // (the order of the device's two butterfly stages: (l + l^1) + (l^2 + l^3))
    const int l = emu_lane();
    return (all[l] + all[l ^ 1]) + (all[l ^ 2] + all[l ^ 3]);
    (void)q0;
}
struct Quad { double a, b, c, d; };
inline Quad rowgroup_gather4(double v) {
    double all[64]; emu::allgather(&v, all, sizeof(double));
    const int a = emu_lane() & 15;
    return Quad{all[a], all[a + 16], all[a + 32], all[a + 48]};
}
inline double permute_f64(double v, int byteaddr) {
    double all[64]; emu::allgather(&v, all, sizeof(double));
    return all[(byteaddr >> 2) & 63];
}
inline unsigned row16_max_u32(unsigned v) {
    unsigned all[64]; emu::allgather(&v, all, sizeof(unsigned));
    const int r0 = emu_lane() & ~15;
    unsigned m = all[r0];
    for (int i = 1; i < 16; ++i) m = all[r0 + i] > m ? all[r0 + i] : m;
    return m;
}
inline unsigned f32_bits(float v) { unsigned b; std::memcpy(&b, &v, 4); return b; }
inline unsigned umax(unsigned a, unsigned b) { return a > b ? a : b; }
inline unsigned bcast_u32(unsigned v, int lane) { unsigned all[64]; emu::allgather(&v, all, sizeof(unsigned)); return all[lane]; }
inline unsigned uniform_u32(unsigned v) { return bcast_u32(v, 0); }
template <int NP> inline double colsum(double v) {
    double all[64]; emu::allgather(&v, all, sizeof(double));
    const int h0 = (emu_lane() / LWOF(NP)) * LWOF(NP);
    double s = 0.0;
    for (int i = 0; i < LWOF(NP); ++i) s += all[h0 + i];
    return s;
}
template <int NP> inline double colmax(double v) {
    double all[64]; emu::allgather(&v, all, sizeof(double));
    const int h0 = (emu_lane() / LWOF(NP)) * LWOF(NP);
    double m = all[h0];
    for (int i = 1; i < LWOF(NP); ++i) m = std::fmax(m, all[h0 + i]);
    return m;
}
template <int NP> inline double colmin(double v) {
    double all[64]; emu::allgather(&v, all, sizeof(double));
    const int h0 = (emu_lane() / LWOF(NP)) * LWOF(NP);
    double m = all[h0];
    for (int i = 1; i < LWOF(NP); ++i) m = std::fmin(m, all[h0 + i]);
    return m;
}
template <int NP> inline float colmax_f32(float v) {
    float all[64]; emu::allgather(&v, all, sizeof(float));
    const int h0 = (emu_lane() / LWOF(NP)) * LWOF(NP);
    float m = all[h0];
    for (int i = 1; i < LWOF(NP); ++i) m = std::fmax(m, all[h0 + i]);
    return m;
}
inline int first_lane_equal_f32(float v, float m) {
    const unsigned long long mask = wave_ballot(v == m);
    return mask ? __builtin_ctzll(mask) : 64;
}
inline int first_lane_equal(double v, double m) {
    const unsigned long long mask = wave_ballot(v == m);
    return mask ? __builtin_ctzll(mask) : 64;
}
template <int NP> inline void colsum2(double va, double vb, double& ra, double& rb) {
    ra = colsum<NP>(va); rb = colsum<NP>(vb);
}
template <int NP> inline double halfsum(double v) {
    double all[64]; emu::allgather(&v, all, sizeof(double));
    if (NP > 32) return v;
    const int c = emu_lane() % LWOF(NP);
    return all[c] + all[c + LWOF(NP)];
}
template <int NP> inline double from_half(double v, int hsel) {
    double all[64]; emu::allgather(&v, all, sizeof(double));
    if (NP > 32) return v;
    return all[emu_lane() % LWOF(NP) + LWOF(NP) * hsel];
}
template <int NP> inline void colargmin(double& v, int& p) {
    double av[64]; int ap[64];
    emu::allgather(&v, av, sizeof(double)); emu::allgather(&p, ap, sizeof(int));
    const int h0 = (emu_lane() / LWOF(NP)) * LWOF(NP);
    double bv = av[h0]; int bp = ap[h0];
    for (int i = 1; i < LWOF(NP); ++i)
        if (av[h0 + i] < bv || (av[h0 + i] == bv && ap[h0 + i] < bp)) { bv = av[h0 + i]; bp = ap[h0 + i]; }
    v = bv; p = bp;
}
inline double bcast(double v, int lane) { double all[64]; emu::allgather(&v, all, sizeof(double)); return all[lane]; }
inline float bcast_f32(float v, int lane) { float all[64]; emu::allgather(&v, all, sizeof(float)); return all[lane]; }
inline int bcast_i(int v, int lane) { int all[64]; emu::allgather(&v, all, sizeof(int)); return all[lane]; }
template <int NP> inline double shift_down(double v) {
    double all[64]; emu::allgather(&v, all, sizeof(double));
    const int l = emu_lane(), c = l % LWOF(NP);
    return (c + 1 < LWOF(NP)) ? all[l + 1] : all[l];
}
template <int NP> inline int shift_down_i(int v) {
    int all[64]; emu::allgather(&v, all, sizeof(int));
    const int l = emu_lane(), c = l % LWOF(NP);
    return (c + 1 < LWOF(NP)) ? all[l + 1] : all[l];
}
inline int frexp_exponent(double x) { int e = 0; if (x != 0.0) std::frexp(x, &e); return e; }
inline double scale_pow2(double x, int e) { return std::ldexp(x, e); }
#ifndef OSOT_EMU_HW_ROUNDING
inline double fast_rcp(double x) { return 1.0 / x; }
inline double fast_rcp1(double x) { return 1.0 / x; }
inline double fast_div(double a, double b) { return a / b; }
inline void fast_sqrt_rsqrt(double x, double& s, double& rs) { s = std::sqrt(x); rs = 1.0 / s; }
#else
// a SECOND round-off pattern for the robustness fixtures: every reciprocal / division / square root is moved one ulp up or
// down (by a bit of its argument), i.e. results that are faithfully but not correctly rounded -- what the hardware's
// v_rcp_f64 / v_rsq_f64 seeds + Newton steps deliver.  Instances that fail on the GPU only do not reproduce under exact
// IEEE division; some do under this one.
inline double ulp_jitter(double v, double key) {
    unsigned long long b; std::memcpy(&b, &key, 8);
    b ^= b >> 17; b *= 0x9E3779B97F4A7C15ull; b ^= b >> 29;
    const int sel = (int)(b % 3);
    return sel == 0 ? v : std::nextafter(v, sel == 1 ? INFINITY : -INFINITY);
}
inline double fast_rcp(double x) { return ulp_jitter(1.0 / x, x); }
inline double fast_rcp1(double x) { return ulp_jitter(1.0 / x, x + 1.0); }
inline double fast_div(double a, double b) { return ulp_jitter(a / b, a + b); }
inline void fast_sqrt_rsqrt(double x, double& s, double& rs) {
    s = std::sqrt(x);
    rs = (s * s == x) ? 1.0 / s : ulp_jitter(1.0 / s, x);     // exact roots stay exact (see the product's routine)
    if (s * s != x) s = ulp_jitter(s, x * 3.0);
}
#endif
}  // namespace osot

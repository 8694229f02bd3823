// osot_team.h -- wavefront primitives for gfx950 (wave64).
//
// ONE QP instance is solved by ONE wavefront.  With NP = padded problem size (32, 64, or 56 = 64 lanes on a shorter LDS
// layout: osot_qp_core.h) the 64 lanes are
// laid out as  lane = c + LW*h :  c = column/element index, h = "half" (LW = 32 or 64 lanes per half).  Every length-n
// vector is held as one element per lane, REPLICATED across the halves; matrices live in the wave's LDS
// slice.  For NP = 32 the two halves split the inner (k) range of every mat-vec and rank-1 update, so the
// serial trip counts are n/2.  All control flow is wave-uniform: loop counters and working-set sizes live
// in SGPRs, broadcasts are v_readlane, reductions are DPP row operations + v_permlane{16,32}_swap (no
// ds_bpermute round trips), and the LDS hand-offs between lanes need no s_barrier: LDS instructions of one
// wave execute in order, the fences below only stop the compiler from reordering them.
//
// tests/emu/osot_team.h is the host lock-step twin of this interface (same names, same semantics) used to
// execute the kernel bodies without a GPU; it is test infrastructure, never part of the product build.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

// LDS declarations (the emulation in tests/emu maps these onto host memory)
#define OSOT_DYNAMIC_LDS(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define OSOT_STATIC_LDS(type, name, count) __shared__ type name[count]
// "these sixteen values are needed now": an empty asm that reads and rewrites them.  Placed after a batch of loads
// it stops the compiler from sinking each load into the branch that uses it (which would serialise the batch into
// sixteen round trips) and makes it wait once, for all of them.
#define OSOT_KEEP16(a) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), \
                                         "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]))
// the same for twelve values a[o] .. a[o + 11]
#define OSOT_KEEP12(a, o) asm volatile("" : "+v"(a[(o) + 0]), "+v"(a[(o) + 1]), "+v"(a[(o) + 2]), "+v"(a[(o) + 3]), "+v"(a[(o) + 4]), "+v"(a[(o) + 5]), \
                                            "+v"(a[(o) + 6]), "+v"(a[(o) + 7]), "+v"(a[(o) + 8]), "+v"(a[(o) + 9]), "+v"(a[(o) + 10]), "+v"(a[(o) + 11]))
// the FIRST kernel parameter as memory: a pointer into the kernarg segment (taking the address of a by-value kernel
// parameter would make the compiler copy it to scratch)
#define OSOT_KERNARG_PTR(type, first_param) ((const type*)(__builtin_amdgcn_kernarg_segment_ptr()))
// call-site inlining (statement attribute): used where ONE instantiation of a template must be inlined
#define OSOT_ALWAYS_INLINE_CALL [[clang::always_inline]]
// a 64-bit integer that holds an HBM address -> pointer in the global address space (global_load, not flat_load)
#define OSOT_GLOBAL_F64(addr) (reinterpret_cast<const __attribute__((address_space(1))) double*>(addr))

namespace osot {

// make this wave's earlier LDS writes visible to its later LDS reads (other lanes of the same wave)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// release/acquire fence at workgroup scope: global-memory stores of this workgroup become visible to its later loads
__device__ __forceinline__ void workgroup_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// forbid the instruction scheduler from moving anything across this point (keeps the register pressure of
// fully unrolled step sequences local to one step)
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }

// returns v, but opaque to the optimiser: values derived from the result cannot be hoisted out of the
// enclosing loop / kept live across phases (used on the lane index to stop loop-invariant code motion
// from parking hundreds of per-lane addresses and masks in registers for the whole kernel)
__device__ __forceinline__ int launder_i(int v) { asm volatile("" : "+v"(v)); return v; }
// the same for a wave-uniform value (stays in a scalar register)
__device__ __forceinline__ int launder_s(int v) { asm volatile("" : "+s"(v)); return v; }

__device__ __forceinline__ int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
// The same for a double / a predicate that IS wave-uniform but that the compiler cannot know to be: the result of a DPP /
// permlane reduction network, or a value loaded through a pointer that may be global.  Branching on such a value makes the
// branch an exec-mask branch and -- when it leaves or continues a loop -- every loop-carried scalar (working-set size,
// iteration count, mode) a vector register; two v_readfirstlane put the control flow back on the scalar unit.
__device__ __forceinline__ double uniform_d(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ bool uniform_b(bool p) { return __builtin_amdgcn_readfirstlane(p ? 1 : 0) != 0; }

// this lane's position in the wavefront ("lane = row" loops; not the column index c of osot_qp_core.h's WaveCtx)
__device__ __forceinline__ int phys_lane() { return (int)(threadIdx.x & 63u); }

// 64-bit mask of the lanes whose predicate is true, and the number of true lanes below this one
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __ballot(p ? 1 : 0); }
__device__ __forceinline__ int lanes_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

__device__ __forceinline__ unsigned f32_bits(float v) { return __float_as_uint(v); }
__device__ __forceinline__ unsigned umax(unsigned a, unsigned b) { return a > b ? a : b; }

// Issue priority of this wavefront among the waves of its SIMD (s_setprio, 0..3), by its position in a dispatch list
// that is sorted by descending expected cost: the launch ends with its longest jobs, so those get the SIMD's issue slots
// first and their co-resident (shorter) waves absorb the delay.  g = position, total = list length.
#ifndef OSOT_PRIO_SH3
#define OSOT_PRIO_SH3 5
#define OSOT_PRIO_SH2 3
#define OSOT_PRIO_SH1 2
#endif
__device__ __forceinline__ void wave_priority_by_rank(unsigned g, unsigned total) {
#ifndef OSOT_PRIO_OFF
    if (g < (total >> OSOT_PRIO_SH3)) __builtin_amdgcn_s_setprio(3);
    else if (g < (total >> OSOT_PRIO_SH2)) __builtin_amdgcn_s_setprio(2);
    else if (g < (total >> OSOT_PRIO_SH1)) __builtin_amdgcn_s_setprio(1);
#endif
}

// ---- DPP helpers on doubles (two 32-bit halves) ------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);   // (every lane has a source under the controls used here: no
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);   //  "old" operand, hence no register copy in front of the DPP move)
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true); }

constexpr int DPP_XOR1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141;  // lane i <-> 7-i inside each 8
constexpr int DPP_MIRROR = 0x140;       // lane i <-> 15-i inside each row of 16

// v_max_f64 / v_min_f64 as they are (IEEE mode: a NaN operand loses, a signalling NaN is quieted by the instruction
// itself).  fmax()/fmin() on values that went through a DPP move or a permlane swap cost a v_max_f64 x, x, x in front
// of every operand: the compiler cannot see that they are canonical.
__device__ __forceinline__ double max_raw(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double min_raw(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// sum over the 16 lanes of each DPP row; every lane of the row gets it
__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_f64<DPP_XOR1>(v);
    v += dpp_f64<DPP_XOR2>(v);
    v += dpp_f64<DPP_HALF_MIRROR>(v);
    v += dpp_f64<DPP_MIRROR>(v);
    return v;
}
// sum over the four lanes of each quad (lanes 4k .. 4k+3); every lane of the quad gets it
__device__ __forceinline__ double quad_sum(double v) {
    v += dpp_f64<DPP_XOR1>(v);
    v += dpp_f64<DPP_XOR2>(v);
    return v;
}
// v_permlane16_swap with both operands = v returns {even rows duplicated, odd rows duplicated}
__device__ __forceinline__ void swap16_pair(double v, double& a, double& b) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto pl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto ph = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    a = __hiloint2double((int)ph[0], (int)pl[0]);
    b = __hiloint2double((int)ph[1], (int)pl[1]);
}
__device__ __forceinline__ void swap32_pair(double v, double& a, double& b) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto pl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto ph = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    a = __hiloint2double((int)ph[0], (int)pl[0]);
    b = __hiloint2double((int)ph[1], (int)pl[1]);
}
__device__ __forceinline__ void swap16_pair_i(int v, int& a, int& b) {
    auto p = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    a = (int)p[0]; b = (int)p[1];
}
__device__ __forceinline__ void swap32_pair_i(int v, int& a, int& b) {
    auto p = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    a = (int)p[0]; b = (int)p[1];
}

// ---- fp64 matrix core: D = A(16x4) B(4x16) + C, one wavefront ------------------------------------------
// v_mfma_f64_16x16x4_f64.  Lane l supplies A[m = l & 15][k = l >> 4] and B[k = l >> 4][n = l & 15]; the
// accumulator holds four results per lane: D[row = (l >> 4) + 4 r][col = l & 15] in element r.
typedef double v4f64 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f64 mfma_f64_16x16x4(double a, double b, v4f64 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// sum over the four rows of 16 lanes (lanes l, l^16, l^32, l^48); every lane gets it
__device__ __forceinline__ double rowgroup_sum(double v) {
    double a, b;
    swap16_pair(v, a, b);
    v = a + b;
    swap32_pair(v, a, b);
    return a + b;
}

// ---- LDS-crossbar permutes with a precomputed BYTE address (4 * source lane): no per-call lane arithmetic ----------
__device__ __forceinline__ double permute_f64(double v, int byteaddr) {
    const int lo = __builtin_amdgcn_ds_bpermute(byteaddr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(byteaddr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
// all-gather over the four rows of 16 lanes: the values of lanes (l & 15) + 16 q, q = 0..3, in every lane.  Eight
// ds_bpermute_b32 (~10 cycles of issue each, one ~80-cycle latency for the lot); the v_permlane16/32_swap form needs twelve
// instructions at ~16 cycles each (tools/ubench_latency.hip).
struct Quad { double a, b, c, d; };
__device__ __forceinline__ Quad rowgroup_gather4(double v) {
    const int a4 = (int)(threadIdx.x & 15u) << 2;
    Quad g;
    g.a = permute_f64(v, a4);
    g.b = permute_f64(v, a4 + 64);
    g.c = permute_f64(v, a4 + 128);
    g.d = permute_f64(v, a4 + 192);
    return g;
}
// maximum of an unsigned over the 16 lanes of each DPP row; every lane of the row gets it
__device__ __forceinline__ unsigned row16_max_u32(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, DPP_XOR1, 0xF, 0xF, true));
    v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, DPP_XOR2, 0xF, 0xF, true));
    v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, DPP_HALF_MIRROR, 0xF, 0xF, true));
    v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, DPP_MIRROR, 0xF, 0xF, true));
    return v;
}
__device__ __forceinline__ unsigned bcast_u32(unsigned v, int lane) { return (unsigned)__builtin_amdgcn_readlane((int)v, lane); }
__device__ __forceinline__ unsigned uniform_u32(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }

// ---- reductions over the NP columns (lanes with equal h); every lane gets the result ------------------
// Round 6, MEASURED AND NOT THE DEFAULT (OSOT_X_MFMA_SUM builds it): SUMS ON THE MATRIX CORE.  A sum over the wavefront as two
// v_mfma_f64_16x16x4 with a constant operand and three additions:
//   D1 = V B1 with A[m][k] = v(lane m + 16 k): D1[m][n] = sum_k v(m + 16 k) B1[k][n]; lane l holds D1[(l >> 4) + 4 r][l & 15], r = 0..3;
//   t  = D1[0] + D1[1] + D1[2] + D1[3]      (the partial sums of the rows congruent to l >> 4 mod 4, at column l & 15);
//   D2 = T 1 with A[m][k] = t(lane m + 16 k): D2[m][n] = sum_k t(m + 16 k) = the total of column m of D1, in EVERY lane.
// With B1 = ones the total is the sum over all 64 lanes; with B1[k][n] = [(n < 8) == (k < 2)] the columns n < 8 of D1 collect the
// lanes 0..31 and the columns n >= 8 the lanes 32..63, so D2's rows 0..7 (elements 0, 1 of every lane) carry the sum over the first
// half of the wavefront and its rows 8..15 (elements 2, 3) the sum over the second: two sums for the price of one.
// The idea: the DPP network of a 32-lane sum is 8 v_mov_dpp + 4 v_add_f64 + 2 v_permlane16_swap + 1 add on the VALU the two co-resident
// wavefronts of a SIMD share, ~110 such sums per instance of BASELINE config 3 = 1.4 k of its 10.5 k VALU instructions, while the matrix
// core is 7 % busy; here the VALU sees three additions.  The measurement (A/B on one box, tools/ab_headline.py, same answers to
// round-off, all 136 GPU tests green): config 3 32.7 against 33.9 M solves/s at 4096, 32.2 against 33.6 M at 32768; COMAN35 S3 / S4
// 6.39 / 4.53 against 6.58 / 4.67 M.  1.4 k fewer VALU instructions per instance made the kernel 4 % SLOWER: a sum this way is two
// dependent 16-pass MFMAs (2 x 64 cycles + the read-after-MFMA wait states) where the network is ~137 cycles, and what bounds these
// kernels is the length of each wavefront's dependent chain, not the VALU's issue rate ("Round 6").
#ifdef OSOT_X_MFMA_SUM
__device__ __forceinline__ double mfma_ones() { return 1.0; }
// B1[k = lane >> 4][n = lane & 15] of the two-halves form: bit 3 and bit 5 of the lane agree
__device__ __forceinline__ double mfma_half_selector() {
    const unsigned l = threadIdx.x;
    return (((l >> 3) ^ (l >> 5)) & 1u) ? 0.0 : 1.0;
}
// (sum over lanes 0..31, sum over lanes 32..63) of v, both in every lane
__device__ __forceinline__ void wave_sum_halves(double v, double& s0, double& s1) {
    const v4f64 z = {0.0, 0.0, 0.0, 0.0};
    const v4f64 d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(v, mfma_half_selector(), z, 0, 0, 0);
    const double t = (d1[0] + d1[1]) + (d1[2] + d1[3]);
    const v4f64 d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(t, mfma_ones(), z, 0, 0, 0);
    s0 = d2[0]; s1 = d2[2];
}
// sum over all 64 lanes
__device__ __forceinline__ double wave_sum64(double v) {
    const v4f64 z = {0.0, 0.0, 0.0, 0.0};
    const v4f64 d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(v, mfma_ones(), z, 0, 0, 0);
    const double t = (d1[0] + d1[1]) + (d1[2] + d1[3]);
    const v4f64 d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(t, mfma_ones(), z, 0, 0, 0);
    return d2[0];
}
template <int NP>
__device__ __forceinline__ double colsum(double v) {
    if (NP > 32) return wave_sum64(v);
    double s0, s1;
    wave_sum_halves(v, s0, s1);
    return (threadIdx.x & 32u) ? s1 : s0;      // (each half its own sum: the contract of the DPP form, whatever the halves hold)
}
#else
template <int NP>
__device__ __forceinline__ double colsum(double v) {
    v = row16_sum(v);
    double a, b;
    swap16_pair(v, a, b);
    v = a + b;                       // 32 lanes of a half
    if (NP > 32) { swap32_pair(v, a, b); v = a + b; }
    return v;
}
#endif
// maximum over the NP columns (lanes with equal h); every lane gets it
template <int NP>
__device__ __forceinline__ double colmax(double v) {
    v = max_raw(v, dpp_f64<DPP_XOR1>(v));
    v = max_raw(v, dpp_f64<DPP_XOR2>(v));
    v = max_raw(v, dpp_f64<DPP_HALF_MIRROR>(v));
    v = max_raw(v, dpp_f64<DPP_MIRROR>(v));
    double a, b;
    swap16_pair(v, a, b);
    v = max_raw(a, b);
    if (NP > 32) { swap32_pair(v, a, b); v = max_raw(a, b); }
    return v;
}
// (inputs >= 0 or -0: non-negative floats order like unsigned integers, and v_max_u32 takes its DPP operand directly)
template <int NP>
__device__ __forceinline__ float colmax_f32(float v) {
    unsigned u = __float_as_uint(v) & 0x7fffffffu;
    u = umax(u, (unsigned)__builtin_amdgcn_mov_dpp((int)u, DPP_XOR1, 0xF, 0xF, true));
    u = umax(u, (unsigned)__builtin_amdgcn_mov_dpp((int)u, DPP_XOR2, 0xF, 0xF, true));
    u = umax(u, (unsigned)__builtin_amdgcn_mov_dpp((int)u, DPP_HALF_MIRROR, 0xF, 0xF, true));
    u = umax(u, (unsigned)__builtin_amdgcn_mov_dpp((int)u, DPP_MIRROR, 0xF, 0xF, true));
    int a, b;
    swap16_pair_i((int)u, a, b);
    u = umax((unsigned)a, (unsigned)b);
    if (NP > 32) { swap32_pair_i((int)u, a, b); u = umax((unsigned)a, (unsigned)b); }
    return __uint_as_float(u);
}
__device__ __forceinline__ int first_lane_equal_f32(float v, float m) {
    const unsigned long long mask = wave_ballot(v == m);
    return mask ? __builtin_ctzll(mask) : 64;
}
// lowest lane whose value equals the (already reduced) extremum m; 64 if none (NaN)
__device__ __forceinline__ int first_lane_equal(double v, double m) {
    const unsigned long long mask = wave_ballot(v == m);
    return mask ? __builtin_ctzll(mask) : 64;
}
// two column sums for the price of one reduction network: half 0 reduces va, half 1 reduces vb, then the
// halves exchange their totals (NP = 32; plain two reductions for NP = 64)
template <int NP>
__device__ __forceinline__ void colsum2(double va, double vb, double& ra, double& rb) {
#ifdef OSOT_X_MFMA_SUM
    if (NP > 32) {
        // two sums over all 64 lanes: one first stage each, ONE second stage -- lane (m, k) hands over t_a for m < 8 and t_b otherwise,
        // so rows 0..7 of the product carry sum(va) and rows 8..15 sum(vb)
        const v4f64 z = {0.0, 0.0, 0.0, 0.0};
        const v4f64 da = __builtin_amdgcn_mfma_f64_16x16x4f64(va, mfma_ones(), z, 0, 0, 0);
        const v4f64 db = __builtin_amdgcn_mfma_f64_16x16x4f64(vb, mfma_ones(), z, 0, 0, 0);
        const double ta = (da[0] + da[1]) + (da[2] + da[3]), tb = (db[0] + db[1]) + (db[2] + db[3]);
        const v4f64 d2 = __builtin_amdgcn_mfma_f64_16x16x4f64((threadIdx.x & 8u) ? tb : ta, mfma_ones(), z, 0, 0, 0);
        ra = d2[0]; rb = d2[2];
        return;
    }
    wave_sum_halves((threadIdx.x >= 32) ? vb : va, ra, rb);
#else
    if (NP > 32) { ra = colsum<64>(va); rb = colsum<64>(vb); return; }
    double v = (threadIdx.x >= 32) ? vb : va;
    v = row16_sum(v);
    double a, b;
    swap16_pair(v, a, b);
    v = a + b;
    swap32_pair(v, ra, rb);
#endif
}
// v(c,0) + v(c,1): combines the partial results of the two halves (identity for NP = 64)
template <int NP>
__device__ __forceinline__ double halfsum(double v) {
    if (NP > 32) return v;
    double a, b;
    swap32_pair(v, a, b);
    return a + b;
}

// value held by lane (c, hsel) delivered to both halves (hsel is wave-uniform; identity for NP = 64)
template <int NP>
__device__ __forceinline__ double from_half(double v, int hsel) {
    if (NP > 32) return v;
    double a, b;
    swap32_pair(v, a, b);
    return hsel ? b : a;
}

// minimum over the columns with the payload of the minimiser (ties: smaller payload); every lane gets both.
// Two plain min networks (the value, then the payload among the lanes that hold the minimum) instead of one
// network that carries (value, payload) pairs through compare-and-select stages: ~270 cycles instead of ~760
// (tools/ubench_latency.hip); the selected pair is the same.
template <int NP>
__device__ __forceinline__ double colmin(double v) {
    v = min_raw(v, dpp_f64<DPP_XOR1>(v));
    v = min_raw(v, dpp_f64<DPP_XOR2>(v));
    v = min_raw(v, dpp_f64<DPP_HALF_MIRROR>(v));
    v = min_raw(v, dpp_f64<DPP_MIRROR>(v));
    double a, b;
    swap16_pair(v, a, b);
    v = min_raw(a, b);
    if (NP > 32) { swap32_pair(v, a, b); v = min_raw(a, b); }
    return v;
}
template <int NP>
__device__ __forceinline__ int colmin_i(int v) {
    v = min(v, dpp_i32<DPP_XOR1>(v));
    v = min(v, dpp_i32<DPP_XOR2>(v));
    v = min(v, dpp_i32<DPP_HALF_MIRROR>(v));
    v = min(v, dpp_i32<DPP_MIRROR>(v));
    int a, b;
    swap16_pair_i(v, a, b);
    v = min(a, b);
    if (NP > 32) { swap32_pair_i(v, a, b); v = min(a, b); }
    return v;
}
// NP = 32: BOTH HALVES MUST HOLD THE SAME CANDIDATES (they do wherever the solver reduces over 32 columns: vectors are
// replicated over the halves).  One lane holds the minimum (the usual case): its payload is a v_readlane away; several do
// (exact ties): the second network picks the smallest payload among them -- the same pair either way.
template <int NP>
__device__ __forceinline__ void colargmin(double& v, int& p) {
    const double m = colmin<NP>(v);
    unsigned long long tie = wave_ballot(v == m);
    if (NP <= 32) tie &= 0xffffffffull;
    if (__builtin_popcountll(tie) == 1) p = __builtin_amdgcn_readlane(p, __builtin_ctzll(tie));
    else p = colmin_i<NP>((v == m) ? p : 0x7fffffff);
    v = m;
}

// value held by lane `lane` (wave-uniform index) -> every lane (v_readlane: the result lives in SGPRs)
__device__ __forceinline__ double bcast(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int bcast_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float bcast_f32(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

// value of lane c+1 of the same half (lane NP-1 keeps its own): used to shift the working-set bookkeeping
template <int NP>
__device__ __forceinline__ double shift_down(double v) { return __shfl_down(v, 1, NP <= 32 ? 32 : 64); }
template <int NP>
__device__ __forceinline__ int shift_down_i(int v) { return __shfl_down(v, 1, NP <= 32 ? 32 : 64); }

// ---- fp64 reciprocal / square root without the IEEE corner-case sequences -----------------------------
// v_rcp_f64 / v_rsq_f64 seeds + Newton steps: ~1 ulp for normal arguments, far cheaper than the compiler's
// div_scale/div_fmas/div_fixup expansion.  Arguments here are pivots/norms that are checked > 0 first.
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    return r;
}
// one Newton step only (~2 ulp): for counts and comparisons, not for values that are kept (sym_bisect_32's Sturm sequences)
__device__ __forceinline__ double fast_rcp1(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return fma(r, fma(-x, r, 1.0), r);
}
// binary exponent of x as frexp returns it (0 for x = 0), one instruction; x 2^e, one instruction
__device__ __forceinline__ int frexp_exponent(double x) { return __builtin_amdgcn_frexp_exp(x); }
__device__ __forceinline__ double scale_pow2(double x, int e) { return __builtin_amdgcn_ldexp(x, e); }
__device__ __forceinline__ double fast_div(double a, double b) {
    const double r = fast_rcp(b);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}
// returns sqrt(x) and 1/sqrt(x) for x > 0.  The reciprocal is taken from the corrected root so that
// exact inputs stay exact (sqrt(1) = 1, 1/1 = 1): the reference's integer-valued known-answer tests on
// rank-deficient H rely on exact cancellation in the eps-pivots (TestQPOases.cpp:274-340).
__device__ __forceinline__ void fast_sqrt_rsqrt(double x, double& s, double& rs) {
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    const double d = fma(-g, g, x);
    g = fma(d, h, g);
    s = g;
    // 1/s from the Goldschmidt companion h ~ 1/(2s) with one Newton step against the corrected root (no
    // v_rcp_f64 + two refinements on the critical path of every factorisation step); exact for exact roots
    double q = h + h;
    const double e = fma(-g, q, 1.0);
    rs = fma(q, e, q);
}

}  // namespace osot

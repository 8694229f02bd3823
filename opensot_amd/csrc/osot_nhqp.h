// opensot_amd/csrc/osot_nhqp.h -- the NULL-SPACE front-end of the reference, OpenSoT::solvers::nHQP (src/solvers/nHQP.cpp),
// for B instances: per priority level the task is projected into the cumulated null space of the levels above
// (AN = A N, b0 = b - A q0), an SVD of AN drives the A/b regularisation and yields the level's own null space, a QP in the
// nf free coordinates is solved by the batched back-end kernel (osot_qp_kernel), and q0 += N z, N <- N V2.
//
//   osot_nhqp_prepare_kernel ..... compute_cost + compute_contraints (nHQP.cpp:357-390, 236-279, 282-317) of one level:
//                                  H, g, constraint rows / bounds in z-coordinates and V2 -> per-instance scratch in HBM
//   osot_nhqp_prepare64_kernel ... the same for 33 .. 64 variables (round 4); osot_nhqp_prepare_wide_kernel: levels beyond 32 on BOTH
//                                  sides (round 5: the reference's one-level COMAN stack S1)
//   osot_qp_kernel<NP> ........... the level's QP (BackEnd convention; NP = 32 / 40 / 56 / 64 by its free variables)
//   osot_nhqp_accumulate_kernel .. solution += N z; N <- N V2 (nHQP.cpp:182-196)
// Round 5: the GEMM-shaped pieces -- Gram matrix, H = AN'W AN + sv_max V2 V2', the triplets' A N V, N V2 -- run on the fp64 matrix
// core (nhqp_tile_gram, acc_tile_product); the eigen-decompositions and the Householder completions stay on the vector unit.
//
// One wavefront per instance, lane = c + 32 h like the QP core (the 32-wide kernel: n <= 32, m <= 64 rows per level).
// The SVD: AN is m x nf with k = min(m, nf) <= 32.  The SYMMETRIC eigenproblem of the SMALL Gram matrix (AN AN' when
// m <= nf, AN'AN otherwise) is solved in LDS by Householder tridiagonalisation + implicit QL (sym_eig32 below) and the other
// factor follows from one product with AN.  Squaring
// costs accuracy only in singular values below ~1e-8 sv_max, which the reference's own regularisation overwrites
// (anything below min_sv_ratio * sv_max = 0.05 sv_max is lifted, nHQP.cpp:262-266).  A level's null-space basis V2 need
// not be the SVD's: every orthonormal basis gives the same q (the QP is posed in its coordinates and both
// regularisations are basis-invariant); for m <= nf it is the orthogonal complement of the leading right singular
// vectors, built with Householder reflections.
#pragma once
#include <osot_team.h>
#include <osot_mi355x.h>

namespace osot {

struct DevNhqp {
    int B, n, nc, nc_stored_all;     // nc: global constraint rows (all stored: unit-row blocks are refused for nHQP)
    int level, m, ma;                // this level: task rows, stored rows (the rest: implicit identity rows)
    int nf, ns;                      // free variables of this level, null-space dimension handed to the next (0: none)
    int has_box;
    int ab_reg, sel_reg;             // perform_A_b_regularization, perform_selective_null_space_regularization
    double thr;                      // min_sv_ratio
    const double* A; const double* b; const double* w;      // level data [B][ma][n], [B][m], [B][m] (w may be null)
    const double* C; const double* lo; const double* up; const double* l; const double* u;
    const double* N;                 // [B][n][n] cumulated null space (n x nf, row stride n); unused at level 0
    const double* q0;                // [B][n] solution so far; unused at level 0
    double* H; double* g;            // out [B][nf][nf], [B][nf]
    double* R; double* rlo; double* rup;   // out (levels > 0) [B][nr][nf], [B][nr], nr = nc + (has_box ? n : 0)
    double* V2;                      // out [B][n][n] (nf x ns, row stride n)
    const int* status;               // [B] status so far (instances that failed above are skipped)
    const double* Wd;                // [B][m][m] the level's FULL weight matrix (Task::getWeight() of the level: block diagonal over its tasks,
                                     // symmetric), or null = diagonal weights w.  Round 5: osot_nhqp_options.level_W
    unsigned long long zero_rows;    // bit r: row r of the level belongs to an INACTIVE task (Task::setActive(false), Task.h:383-387: A is
                                     // zeroed, b stays): the row of A N is a zero row
};

constexpr int kNS = 33;              // LDS row stride of the 32-column work matrices
#ifndef OSOT_QL_TOL
#define OSOT_QL_TOL 1.0
#endif

// sum over the 32 lanes of a half AND over the two halves; every lane gets it
__device__ __forceinline__ double sum64(double v) { return halfsum<32>(colsum<32>(v)); }

// sum over i = h, h + 2, ... < 2 TRIPS of a[i sa] b[i sb]: a FIXED, fully unrolled trip count, so that the LDS reads of the
// whole product are in flight together (a loop with a run-time trip count pays one LDS round trip, ~100 cycles, per
// element).  Both arrays must be readable and ZERO beyond their logical length (the work matrices are zero-padded to 32).
template <int TRIPS>
__device__ __forceinline__ double dot_half(const double* a, int sa, const double* b, int sb, int h) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int t = 0; t < TRIPS; t += 2) {
        const int i0 = 2 * t + h, i1 = 2 * (t + 1) + h;
        s0 = fma(a[i0 * sa], b[i0 * sb], s0);
        s1 = fma(a[i1 * sa], b[i1 * sb], s1);
    }
    return s0 + s1;
}

// Symmetric eigenproblem of K (k x k, LDS [32][33], zero beyond k) -> K[c][c] = eigenvalue c, E[:, c] = its eigenvector.
// Householder tridiagonalisation with accumulation of the transformations, then the implicit QL iteration with shifts
// (the classic EISPACK pair tred2 / tql2).  Why not Jacobi: the parallel two-sided Jacobi this replaces moved ~60 k LDS
// words per lane for a 24 x 24 matrix (14 sweeps x 31 steps, column AND row phase through LDS) and the kernel was bound by
// the CU's LDS bandwidth (4.1 ms per level for 4096 instances); this form moves ~7 k: the reduction keeps rows in their
// own lanes (lane j = row j, the inner index split over the two halves), and the QL sweep is a scalar recurrence on (d, e)
// -- held one entry per lane, read with v_readlane -- whose plane rotations touch two entries of every lane's OWN row of
// the eigenvector matrix.
// (force-inlined, and k / the reduction results said to be wave-uniform: as a CALL every argument arrives in a vector register, so
//  the size, every loop bound derived from it and with them the whole QL recurrence were compiled as divergent control flow
//  -- loop counters in VGPRs, exec-mask branches, a null check in front of every access through the generic pointers)
// phase 1 + 2 of sym_eig32: K (k x k symmetric) -> tridiagonal (d, e: one entry per lane, e[c] coupling c - 1 and c as tred2 leaves
// it) and K <- the accumulated orthogonal transformation Q (A = Q T Q')
// ACCUM = false: the tridiagonal form only (d, e), K is left in pieces -- for callers that want eigenVALUES alone
template <bool ACCUM = true>
__device__ __forceinline__ void sym_tred2_32(double* K, double* E, int k, int c, int h, double& d_out, double& e_out) {
    constexpr double kEps = 2.220446049250313e-16;
    double d = 0.0, e = 0.0;                 // lane j: d[j], e[j]
    const bool wr = (h == 0);
    // ---- one exact scaling of the whole matrix (a power of two, largest entry into [0.5, 1)) instead of tred2's scaling of every
    // row by its 1-norm -- a wave reduction, a reciprocal and two multiplies on the dependent chain of each of the k - 1 steps
    int kexp = 0;
    {
        double mx[16], rmax = 0.0;
#pragma unroll
        for (int t = 0; t < 16; ++t) { mx[t] = K[c * kNS + 2 * t + h]; rmax = fmax(rmax, fabs(mx[t])); }
        rmax = fmax(rmax, from_half<32>(rmax, 1 - h));
        kexp = frexp_exponent(uniform_d(colmax<32>(rmax)));
#pragma unroll
        for (int t = 0; t < 16; ++t) K[c * kNS + 2 * t + h] = scale_pow2(mx[t], -kexp);
        wave_sync();
    }
    // ---- reduction to tridiagonal form (tred2): for i = k-1 .. 1 the row i is reflected onto e_{i-1}
    for (int i = k - 1; i >= 1; --i) {
        const int l = i - 1;
        double hv = 0.0, e_i;
        if (l > 0) {
            const double kic = K[i * kNS + c];       // (loaded by every lane, then masked: no exec-mask branch around the load)
            double ai = (c <= l) ? kic : 0.0;
            hv = uniform_d(colsum<32>(ai * ai));
            if (hv < 1.0e-30) {
                // the part of row i still to be reduced is below 1e-15 of the matrix's largest entry: round-off (the Gram matrix of an
                // orthonormal N at a Postural level is I + noise).  Dropped -- a backward error at the level of the arithmetic --
                // instead of spending a full reflection on it and leaving QL a cluster of noise-level couplings to iterate on.
                e_i = 0.0;
                hv = 0.0;
                wave_sync();
                if (wr && c <= l) { K[i * kNS + c] = 0.0; K[c * kNS + i] = 0.0; }     // (no reflector: row / column i say so to whoever
                wave_sync();                                                           //  applies the stored reflectors later)
            } else {
                const double f0 = bcast(ai, l);
                double sq, rs;
                fast_sqrt_rsqrt(hv, sq, rs);
                const double g0 = (f0 >= 0.0) ? -sq : sq;
                e_i = g0;
                hv -= f0 * g0;
                if (c == l) ai = f0 - g0;
                const double ih = fast_rcp(hv);
                wave_sync();
                if (wr && c <= l) { K[i * kNS + c] = ai; K[c * kNS + i] = ai * ih; }    // row i <- u, column i <- u / H
                wave_sync();
                // p = A u / H (full symmetric rows are kept up to date).  The inner index runs in CHUNKS of 8 (4 per half) with the
                // reads of a chunk in flight together and the entries beyond l masked: element by element the loop paid one LDS round
                // trip per multiply-add (see dot_half; the chunk keeps the LDS traffic near the triangular minimum, which the fixed
                // 32-wide form would triple).  Chunk bases are multiples of 8 and l <= 31, so every index read is a valid column.
                double pj = 0.0;
                for (int base = 0; base <= l; base += 8) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int kk = base + 2 * t + h;
                        const double a = K[c * kNS + kk], u = K[i * kNS + kk];
                        pj = fma((kk <= l) ? a : 0.0, u, pj);
                    }
                }
                pj = halfsum<32>(pj) * ih;
                const double f1 = uniform_d(colsum<32>((c <= l) ? pj * ai : 0.0));
                const double hh2 = 0.5 * f1 * ih;
                const double qj = (c <= l) ? pj - hh2 * ai : 0.0;      // q = p - (u'p / 2H) u
                if (wr) E[c] = qj;           // (E is free until the end: its first row carries q)
                wave_sync();
                for (int base = 0; base <= l; base += 8) {        // A <- A - u q' - q u' (my row; same chunks)
                    double nv[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int kk = base + 2 * t + h;
                        nv[t] = K[c * kNS + kk] - fma(ai, E[kk], qj * K[i * kNS + kk]);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int kk = base + 2 * t + h;
                        if (c <= l && kk <= l) K[c * kNS + kk] = nv[t];
                    }
                }
                wave_sync();
            }
        } else {
            e_i = K[i * kNS + l];
        }
        if (c == i) { e = e_i; d = hv; }
    }
    if constexpr (!ACCUM) {       // the diagonal of the tridiagonal form sits on K's diagonal (the accumulation below does not move it)
        const double kcc = K[c * kNS + c];
        d_out = scale_pow2((c < k) ? kcc : 0.0, kexp); e_out = scale_pow2(e, kexp);
        return;
    }
    // ---- accumulate the transformations: K becomes the orthogonal Q
    for (int i = 0; i < k; ++i) {
        const int l = i - 1;
        if (bcast(d, i) != 0.0) {
            double gj = 0.0;
            for (int base = 0; base <= l; base += 8) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int kk = base + 2 * t + h;
                    const double u = K[i * kNS + kk], a = K[kk * kNS + c];
                    gj = fma((kk <= l) ? u : 0.0, a, gj);
                }
            }
            gj = halfsum<32>(gj);
            for (int base = 0; base <= l; base += 8) {
                double nv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int kk = base + 2 * t + h;
                    nv[t] = fma(-gj, K[kk * kNS + i], K[kk * kNS + c]);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int kk = base + 2 * t + h;
                    if (c <= l && kk <= l) K[kk * kNS + c] = nv[t];
                }
            }
            wave_sync();
        }
        if (c == i) d = K[i * kNS + i];
        wave_sync();
        if (wr) {
            if (c == i) K[i * kNS + i] = 1.0;
            else if (c <= l) { K[c * kNS + i] = 0.0; K[i * kNS + c] = 0.0; }
        }
        wave_sync();
    }
    d_out = scale_pow2(d, kexp); e_out = scale_pow2(e, kexp);
}

// Round 6: sym_tred2_32<false> with the matrix in REGISTERS.  The LDS form above pays, per reflector step, two chunked passes over the
// lane's row (8 reads in flight, ~150 clocks per chunk and dependent on the one before) with two synchronisations around them: 4.5 k
// clocks per step, 100 k of the 270 k clocks of a 24 x 29 level (BASELINE config 3's level 1).  Here lane c keeps row c of the
// symmetric work matrix in 32 registers (both halves the same row: nothing is split, nothing is exchanged through LDS), the
// reflector's entries u_kk and q_kk reach the other lanes as v_readlane broadcasts (scalar operands of the fma), the steps are unrolled
// with compile-time bounds (steps at or beyond k are skipped by a scalar branch), and p = A u is a lane-local dot product -- no
// reduction.  The arithmetic is the LDS form's operation by operation (the even / odd partial sums of p are the two halves' chunks) except
// that a step takes row i of the symmetric matrix from the lanes' own COLUMN-i entries (the two triangles differ in the last bit: the
// rank-two update rounds u_c q_k + q_c u_k differently on either side of the diagonal): (d, e) and the stored reflectors (row i of K = u,
// column i = u / H: what sym_bisect_32<2> back-transforms through) agree with the LDS form's to round-off.
template <int P, int N, class F>
__device__ __forceinline__ void nh_static_for(F& f) {
    if constexpr (P < N) { f(std::integral_constant<int, P>{}); nh_static_for<P + 1, N>(f); }
}
#ifndef OSOT_X_TRED2_LDS
constexpr bool kTred2Regs = true;
#else
constexpr bool kTred2Regs = false;      // (A/B builds: the LDS form)
#endif
__device__ __forceinline__ void sym_tred2_32_regs(double* K, int k, int c, int h, double& d_out, double& e_out) {
    double e = 0.0;
    const bool wr = (h == 0);
    double a[32];                        // row c of the work matrix
    int kexp = 0;
    {
        double rmax = 0.0;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) { a[kk] = K[c * kNS + kk]; rmax = fmax(rmax, fabs(a[kk])); }
        kexp = frexp_exponent(uniform_d(colmax<32>(rmax)));
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) a[kk] = scale_pow2(a[kk], -kexp);
    }
    auto step = [&](auto ic) {
        constexpr int I = 31 - decltype(ic)::value;      // i = 31 .. 1
        constexpr int L = I - 1;
        if (I >= k) return;                              // (uniform)
        double e_i;
        if constexpr (L > 0) {
            double ai = (c <= L) ? a[I] : 0.0;           // row I at my column = my row at column I (the rows are kept symmetric)
            double hv = uniform_d(colsum<32>(ai * ai));
            if (hv < 1.0e-30) {                          // (round-off: no reflector -- see the LDS form)
                e_i = 0.0;
                if (wr && c <= L) { K[I * kNS + c] = 0.0; K[c * kNS + I] = 0.0; }
            } else {
                const double f0 = bcast(ai, L);
                double sq, rs;
                fast_sqrt_rsqrt(hv, sq, rs);
                const double g0 = (f0 >= 0.0) ? -sq : sq;
                e_i = g0;
                hv -= f0 * g0;
                if (c == L) ai = f0 - g0;
                const double ih = fast_rcp(hv);
                if (wr && c <= L) { K[I * kNS + c] = ai; K[c * kNS + I] = ai * ih; }    // row I <- u, column I <- u / H
                double u[L + 1];
#pragma unroll
                for (int kk = 0; kk <= L; ++kk) u[kk] = bcast(ai, kk);
                double pe = 0.0, po = 0.0;               // (the two halves' partial sums of the LDS form)
#pragma unroll
                for (int kk = 0; kk <= L; ++kk) {
                    if (kk & 1) po = fma(a[kk], u[kk], po);
                    else pe = fma(a[kk], u[kk], pe);
                }
                const double pj = (h ? po + pe : pe + po) * ih;
                const double f1 = uniform_d(colsum<32>((c <= L) ? pj * ai : 0.0));
                const double hh2 = 0.5 * f1 * ih;
                const double qj = (c <= L) ? pj - hh2 * ai : 0.0;
#pragma unroll
                for (int kk = 0; kk <= L; ++kk) {
                    const double nv = a[kk] - fma(ai, bcast(qj, kk), qj * u[kk]);
                    a[kk] = (c <= L) ? nv : a[kk];
                }
            }
        } else {
            e_i = bcast(a[1], 0);                        // K[1][0]
        }
        if (c == I) e = e_i;
    };
    nh_static_for<0, 31>(step);
    double dd = 0.0;
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) dd = (kk == c) ? a[kk] : dd;
    wave_sync();
    d_out = scale_pow2((c < k) ? dd : 0.0, kexp); e_out = scale_pow2(e, kexp);
}

// phase 3 of sym_eig32: implicit QL with shifts on (d, e) (lane c: d[c]; e[c] couples c - 1 and c on entry), rotations into the columns of K
__device__ __forceinline__ void sym_ql_32(double* K, int k, int c, int h, double& d, double& e) {
    constexpr double kEps = 2.220446049250313e-16;
    // ---- implicit QL with shifts on (d, e); the rotations go into the columns of K (tql2)
    {   // e[j] <- e[j + 1]  (the shuffle OUTSIDE the select: a ds_bpermute under a partial exec mask reads zeros from the
        // lanes that are masked off, and lane k - 2 needs lane k - 1)
        const double es = shift_down<32>(e);
        e = (c + 1 < k) ? es : 0.0;
    }
    // (deflation test against the norm of the whole tridiagonal matrix, as in EISPACK's tql2 -- not against the two
    //  neighbouring diagonal entries: the Gram matrices here are often rank deficient, and a cluster of round-off-level
    //  eigenvalues never passes a purely local test)
    const double anorm = uniform_d(colmax<32>((c < k) ? fabs(d) + fabs(e) : 0.0));
    const double etol = OSOT_QL_TOL * kEps * anorm;
    for (int l = 0; l < k; ++l) {
        for (int iter = 0; iter < 60; ++iter) {
            const bool small = (c >= l && c < k - 1) && (fabs(e) <= etol);
            const unsigned long long mk = wave_ballot(small) & 0xffffffffull;
            const int m = mk ? __builtin_ctzll(mk) : k - 1;
            if (m == l) break;
            const double dl = bcast(d, l), dl1 = bcast(d, l + 1), el = bcast(e, l), dm = bcast(d, m);
            double g = (dl1 - dl) / (2.0 * el);
            double r = sqrt(fma(g, g, 1.0));
            g = dm - dl + el / (g + (g >= 0.0 ? r : -r));
            double sn = 1.0, cs = 1.0, p = 0.0;
            bool underflow = false;
            double di1 = dm;                 // d[i + 1] of the step about to run (d[i] of a step is not written by it)
            // my row of the eigenvector matrix: the entry of column i + 1 travels in a register from step to step (it is
            // the one the previous step produced), the entry of column i is fetched one step ahead: the LDS round trip is
            // off the recurrence's chain.  All 64 lanes run it (rows >= k are zero rows, the halves write the same values).
            double zc = K[c * kNS + m];
            double z0n = K[c * kNS + m - 1];
            for (int i = m - 1; i >= l; --i) {
                const double z0 = z0n;
                if (i > l) z0n = K[c * kNS + i - 1];
                const double ei = bcast(e, i), di = bcast(d, i);
                const double f = sn * ei, b = cs * ei;
                const double rr2 = fma(f, f, g * g);
                double ir;
                if (rr2 > 0.0) fast_sqrt_rsqrt(rr2, r, ir);          // r and 1 / r from one Goldschmidt chain
                else { r = 0.0; ir = 0.0; }
                if (c == i + 1) e = r;
                if (r == 0.0) {
                    if (c == i + 1) d -= p;
                    if (c == m) e = 0.0;
                    K[c * kNS + i + 1] = zc;
                    underflow = true;
                    break;
                }
                sn = f * ir; cs = g * ir;
                g = di1 - p;
                r = fma(di - g, sn, 2.0 * cs * b);
                p = sn * r;
                if (c == i + 1) d = g + p;
                g = fma(cs, r, -b);
                di1 = di;
                K[c * kNS + i + 1] = fma(sn, z0, cs * zc);
                zc = fma(cs, z0, -sn * zc);
            }
            if (underflow) continue;
            K[c * kNS + l] = zc;
            if (c == l) { d -= p; e = g; }
            if (c == m) e = 0.0;
        }
    }
}

// results where the callers expect them: E = eigenvectors (columns), diag(K) = eigenvalues (lane c: d), K otherwise zero
__device__ __forceinline__ void sym_eig_finish_32(double* K, double* E, int k, int c, int h, double d) {
    const bool wr = (h == 0);
    wave_sync();
    for (int t = 0; t < 16; ++t) { const int i = 2 * t + h; E[i * kNS + c] = (i < k && c < k) ? K[i * kNS + c] : 0.0; }
    wave_sync();
    for (int t = 0; t < 16; ++t) { const int i = 2 * t + h; K[i * kNS + c] = 0.0; }
    wave_sync();
    if (wr && c < k) K[c * kNS + c] = d;
    wave_sync();
}


// ---------------------------------------------------------------------------------------------------------------------
// Round 4: the eigen-decomposition of the tridiagonal form WITHOUT the QL recurrence (sym_eig32_fast; the nHQP level preparation).
// tql2 is a scalar recurrence -- rotation i + 1 needs the result of rotation i, ~650 rotations for a 24 x 24 matrix, each followed by
// two LDS columns per lane -- and was 54-73 % of the level-1 launch.  Here every lane works on ITS OWN eigenpair, all of them at once:
//   eigenvalue j (lane j) by bisection on the Sturm count of T - x I (56 halvings of the Gershgorin interval; d and e^2 are read as
//     LDS broadcasts, the recurrence q_i = (d_i - x) - e_(i-1)^2 / q_(i-1) is lane-local);
//   eigenvector j by the twisted factorisation of T - lambda_j I (forward pivots D+ into the lane's column of E, backward pivots D-
//     in registers, the twist index r where |gamma_i| = |D+_i + D-_i - (d_i - lambda)| is smallest, y_r = 1 and the two-term
//     recurrences up and down from r);
//   V = Q Y row by row into K (Q from tred2).
// Eigenvalues closer than 1e-7 |T| to a neighbour (a rank-deficient Gram matrix: the zero cluster; accidental near-multiplicities)
// make the routine return false BEFORE anything is overwritten, and the caller runs the QL iteration on the same (d, e, Q).
// d, e as sym_tred2_32 leaves them (e[c] couples c - 1 and c).  On true: K = eigenvectors (columns), d = this lane's eigenvalue.
// MODE 1: the full decomposition (above).  MODE 0: the eigenvalues alone (d <- eigenvalue number c, ascending; always succeeds:
// clusters are no obstacle to counting).  MODE 2 (K holds what sym_tred2_32<false> left: the reflectors, not Q): the eigenvalues
// and the eigenVECTORS of the nl smallest ones only -- those below thr2 times the largest -- as columns 0 .. nl - 1 of E (twisted
// factorisation, then back-transformed through the stored reflectors); false if the matrix is rank deficient at noise level or
// one of those eigenvalues sits in a cluster (the caller then takes the full route).
template <int MODE = 1>
__device__ __forceinline__ bool sym_bisect_32(double* K, double* E, int k, int c, int h, double& d, double e, double thr2 = 0.0, int* nl_out = nullptr) {
    constexpr bool VECTORS = MODE != 0;
    const bool in = c < k;
    // d[i] -> K[i][32], e2[i] = (coupling i, i + 1)^2 and e[i] -> E[i][32] / E[i][...]: the padding column of the two matrices
    const double ec = shift_down<32>(e);                 // e[c] <- e[c + 1]: couples c and c + 1
    const double eu = (c + 1 < k) ? ec : 0.0;
    const double el = (c > 0 && in) ? e : 0.0;           // couples c - 1 and c
    if (h == 0) { K[c * kNS + 32] = in ? d : 0.0; E[c * kNS + 32] = in ? eu : 0.0; }
    wave_sync();
    const double rad = fabs(el) + fabs(eu);
    const double gl = -uniform_d(colmax<32>(in ? -(d - rad) : -INFINITY));
    const double gu = uniform_d(colmax<32>(in ? d + rad : -INFINITY));
    const double anorm = fmax(fabs(gl), fabs(gu));
    if (!(anorm > 0.0)) { if (!VECTORS) d = 0.0; return !VECTORS; }   // the zero matrix: one k-fold cluster (its eigenvalues: zeros)
    const double e2max = uniform_d(colmax<32>(eu * eu));
    const double pivmin = 2.2250738585072014e-308 * fmax(1.0, e2max) * 4.0;
    // (a tridiagonal form that is already diagonal -- the Gram matrix of an orthonormal N at a Postural level is the identity -- is QL's
    //  trivial case and usually one cluster: not worth 15 sweeps to find that out)
    if (MODE == 1 && e2max <= 1.0e-28 * anorm * anorm) return false;
    // ---- bisection: lane c looks for eigenvalue number c (ascending); count(x) = eigenvalues below x
    // The two halves of the wave test two abscissae of the same interval per sweep (it shrinks by 3: 36 sweeps for 2^-56).  The count
    // is taken from the signs of the leading principal minors p_i = (d_i - x) p_(i-1) - e_i^2 p_(i-2) (d, e^2 of the matrix scaled
    // to norm 1, in registers; rescaled by a power of two every 4 rows): the pivot form q_i = (d_i - x) - e_i^2 / q_(i-1) has a
    // reciprocal on a dependent chain of ~110 clocks per row at the 1.5 waves per SIMD this kernel runs with, and reading d / e
    // from LDS row by row another round trip -- the fixed 56 x k rows of the first version were slower than QL for small k.
    // A minor that vanishes is given the sign opposite to its predecessor's (the pivot form's q = -pivmin).
    const double inorm = fast_rcp(anorm);
    // (rows beyond k: d = 4 > every abscissa and no coupling -- the minor keeps its sign, no flip is counted, no per-row predicate.
    //  Couplings get a floor of 1e-30 of the norm: a perturbation far below round-off that keeps a vanished minor from zeroing its
    //  successors, p_(i+1) = -e^2 p_(i-1) != 0, so zeros need no special case: +0 counts as positive and the flip shows up one row on.)
    double ds[32], es2[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const double di = K[i * kNS + 32] * inorm;
        const double ei = E[(i > 0 ? i - 1 : 0) * kNS + 32] * inorm;
        ds[i] = (i < k) ? di : 4.0;
        es2[i] = (i > 0 && i < k) ? fmax(ei * ei, 1.0e-60) : 0.0;
    }
    // (two abscissae per lane and sweep -- a second, independent chain, the interval shrinking by 5 -- measured slower: 100 k clocks
    //  instead of 90 k at k = 24)
    double lo = (gl - 2.0e-16 * anorm * k) * inorm - 1.0e-300, hi = (gu + 2.0e-16 * anorm * k) * inorm + 1.0e-300;
    const int want = in ? c : 0;
    for (int it = 0; it < 36; ++it) {
        const double w = hi - lo;
        const double x1 = fma(w, 1.0 / 3.0, lo), x2 = fma(w, 2.0 / 3.0, lo);
        const double x = h ? x2 : x1;
        double p1 = 1.0, p2 = 0.0;
        int cnt = 0;
#pragma unroll
        for (int base = 0; base < 32; base += 4) {
            if (base < k) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = base + j;
                    const double pn = fma(ds[i] - x, p1, -(es2[i] * p2));
                    cnt += ((pn < 0.0) != (p1 < 0.0)) ? 1 : 0;
                    p2 = p1; p1 = pn;
                }
                const int ex = frexp_exponent(fabs(p1) + fabs(p2));
                p1 = scale_pow2(p1, -ex); p2 = scale_pow2(p2, -ex);
            }
        }
        const unsigned long long below = wave_ballot(cnt > want);      // bit c: lambda_c < x1; bit 32 + c: lambda_c < x2
        const bool b1 = (below >> c) & 1ull, b2 = (below >> (32 + c)) & 1ull;
        if (b1) hi = x1;
        else if (b2) { lo = x1; hi = x2; }
        else lo = x2;
        if (MODE == 1 && it == 14) {
            // an early look for clusters: two neighbours still in the same interval (3^-15 of the spectrum's width, ~1e-7) after 15
            // sweeps will not separate to the 1e-7 the twisted factorisation needs -- leave now instead of after 36
            const double lo_next = shift_down<32>(lo);
            if (wave_ballot(c + 1 < k && lo_next == lo) != 0ull) return false;
        }
    }
    lo *= anorm; hi *= anorm;
    const double lam = 0.5 * (lo + hi);
    if constexpr (!VECTORS) { d = in ? lam : 0.0; return true; }
    int nl = k;                                          // eigenvectors wanted: all (MODE 1) or the nl smallest (MODE 2)
    if constexpr (MODE == 2) {
        constexpr double kNoise2 = 1.0e-14;              // (singular values below 1e-7 of the largest: rank deficient at the Gram route's noise level)
        const double lmax = bcast(lam, k - 1), lmin = bcast(lam, 0);
        d = in ? lam : 0.0;
        if (!(lmin > 0.0) || lmin < kNoise2 * lmax) return false;
        nl = __builtin_popcountll(wave_ballot(in && h == 0 && lam < thr2 * lmax));
        if (nl_out) *nl_out = nl;
        if (nl == 0) return true;
    }
    // ---- clusters: the twisted factorisation needs separated eigenvalues
    {
        const double nxt = shift_down<32>(lam);
        const bool close = (c + 1 < k) && (c < nl) && (nxt - lam < 1.0e-7 * anorm);
        if (wave_ballot(close) != 0ull) return false;
    }
    // ---- eigenvector of T for lam: twisted factorisation, lane-local (column c of E holds D+ and then y)
    double Dm[32];
    double nrm2 = 0.0;
    {
        const int col = in ? c : 0;                      // (idle lanes shadow lane 0: in-bounds addresses, results discarded)
        double dp = K[32] - lam;
        if (fabs(dp) < pivmin) dp = -pivmin;
        if (in && h == 0) E[col] = dp;
        for (int i = 1; i < k; ++i) {                    // forward: D+_i
            const double ei = E[(i - 1) * kNS + 32];
            dp = (K[i * kNS + 32] - lam) - ei * ei * fast_rcp(dp);
            if (fabs(dp) < pivmin) dp = -pivmin;
            if (in && h == 0) E[i * kNS + col] = dp;
        }
        wave_sync();
        // backward: D-_i (registers, static indices), gamma_i and the twist index
        double dmv = K[(k - 1) * kNS + 32] - lam;
        if (fabs(dmv) < pivmin) dmv = -pivmin;
        double gbest = INFINITY;
        int r = k - 1;
#pragma unroll
        for (int i = 31; i >= 0; --i) {
            if (i < k) {
                if (i < k - 1) {
                    const double ei = E[i * kNS + 32];
                    dmv = (K[i * kNS + 32] - lam) - ei * ei * fast_rcp(dmv);
                    if (fabs(dmv) < pivmin) dmv = -pivmin;
                }
                Dm[i] = dmv;
                const double gam = fabs(E[i * kNS + col] + dmv - (K[i * kNS + 32] - lam));
                if (gam < gbest) { gbest = gam; r = i; }
            } else {
                Dm[i] = 1.0;
            }
        }
        // y_r = 1; upwards y_i = -(e_i / D+_i) y_(i+1); downwards y_(i+1) = -(e_i / D-_(i+1)) y_i.  y overwrites D+ in place.
        double yv = 1.0;
        nrm2 = 1.0;
        for (int i = k - 2; i >= 0; --i) {               // (rows above the twist; the others are skipped by the select)
            const double dpi = E[i * kNS + col];
            const double ynew = -(E[i * kNS + 32] * fast_rcp(dpi)) * yv;
            const bool up = i < r;
            yv = up ? ynew : 1.0;                        // at i >= r the running value is re-seeded: y_r = 1
            if (up) nrm2 = fma(yv, yv, nrm2);
            if (up && in && h == 0) E[i * kNS + col] = yv;
        }
        wave_sync();
        if (in && h == 0) E[r * kNS + col] = 1.0;
        yv = 1.0;
#pragma unroll
        for (int i = 0; i < 31; ++i) {
            if (i < k - 1) {
                const double ynew = -(E[i * kNS + 32] * fast_rcp(Dm[i + 1])) * yv;
                const bool down = i >= r;
                yv = down ? ynew : 1.0;
                if (down) nrm2 = fma(yv, yv, nrm2);
                if (down && in && h == 0) E[(i + 1) * kNS + col] = yv;
            }
        }
        wave_sync();
    }
    // ---- V = Q Y, row by row into K (Q's row i is read by every lane before it is overwritten); the columns are normalised on the way
    // (the norm is the first half's: the second half runs the same recurrences on the same column, but its reads of D+ race with
    //  the first half's in-place writes of y -- harmless in lock-step, and nothing of the second half's is used)
    double sq, rs;
    {   // a pivot clamped to -pivmin can drive the y recurrence to Inf / NaN: such a vector is no eigenvector -- the caller's other
        // route (QL iteration / the full decomposition) takes the matrix instead
        const double n2 = from_half<32>(nrm2, 0);
        if (wave_ballot(in && !(n2 > 0.0 && n2 < INFINITY)) != 0ull) return false;
    }
    fast_sqrt_rsqrt(from_half<32>(nrm2, 0), sq, rs);
    if constexpr (MODE == 2) {
        // my column of Y normalised in place, then V = H_(k-1) .. H_2 Y for every column at once (lane = column, its entries split
        // over the halves): tred2 left reflector i as row i of K (u) and column i of K (u / H); what Q would have cost to build
        // and to multiply is k - 2 dot products and updates of a column here.  Only the first nl columns are used afterwards.
        if constexpr (kTred2Regs) {
            // (round 6) my column of Y in 32 registers, both halves the same column: the reflector's two vectors are uniform-address LDS
            // reads in flight together, the dot product is lane-local, no synchronisation between the steps (a column belongs to its lane)
            double y[32];
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) { const double v = E[kk * kNS + c]; y[kk] = (kk < k) ? v * rs : 0.0; }
            auto bstep = [&](auto ic) {
                constexpr int I = decltype(ic)::value;       // 2 .. 31
                constexpr int L = I - 1;
                if (I >= k) return;                          // (uniform)
                double u[L + 1], uh[L + 1];
#pragma unroll
                for (int kk = 0; kk <= L; ++kk) { u[kk] = K[I * kNS + kk]; uh[kk] = K[kk * kNS + I]; }
                double de = 0.0, dq = 0.0;                   // (the two halves' partial sums of the LDS form)
#pragma unroll
                for (int kk = 0; kk <= L; ++kk) {
                    if (kk & 1) dq = fma(u[kk], y[kk], dq);
                    else de = fma(u[kk], y[kk], de);
                }
                const double dot = h ? dq + de : de + dq;
#pragma unroll
                for (int kk = 0; kk <= L; ++kk) y[kk] = fma(-dot, uh[kk], y[kk]);
            };
            nh_static_for<2, 32>(bstep);
            wave_sync();
            if (in && h == 0) {
#pragma unroll
                for (int kk = 0; kk < 32; ++kk) if (kk < k) E[kk * kNS + c] = y[kk];
            }
            wave_sync();
            return true;
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) { const int r = 2 * t + h; const double v = E[r * kNS + c]; if (r < k && in) E[r * kNS + c] = v * rs; }
        wave_sync();
        for (int i = 2; i < k; ++i) {
            const int l = i - 1;
            double dot = 0.0;
            for (int base = 0; base <= l; base += 8) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int kk = base + 2 * t + h;
                    const double u = K[i * kNS + kk], y = E[kk * kNS + c];
                    dot = fma((kk <= l) ? u : 0.0, y, dot);
                }
            }
            dot = halfsum<32>(dot);
            for (int base = 0; base <= l; base += 8) {
                double nv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int kk = base + 2 * t + h;
                    nv[t] = fma(-dot, K[kk * kNS + i], E[kk * kNS + c]);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int kk = base + 2 * t + h;
                    if (kk <= l && in) E[kk * kNS + c] = nv[t];
                }
            }
            wave_sync();
        }
        return true;
    }
    // (my column of Y in registers; Q is zero beyond column k and E holds finite numbers there: fixed trip counts, no masks; two rows
    //  of Q per trip -- independent chains)
    {
        double yc[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) yc[t] = E[(2 * t + h) * kNS + c] * rs;
        for (int i0 = 0; i0 < k; i0 += 2) {
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                a0 = fma(K[i0 * kNS + 2 * t + h], yc[t], a0);
                a1 = fma(K[(i0 + 1) * kNS + 2 * t + h], yc[t], a1);
            }
            a0 = halfsum<32>(a0); a1 = halfsum<32>(a1);
            wave_sync();
            if (in && h == 0) { K[i0 * kNS + c] = a0; if (i0 + 1 < k) K[(i0 + 1) * kNS + c] = a1; }
        }
    }
    wave_sync();
    d = in ? lam : 0.0;
    return true;
}

// sym_eig32 with the bisection / twisted-factorisation phase where the spectrum allows it (see sym_bisect_32), QL otherwise
__device__ __forceinline__ void sym_eig32_fast(double* K, double* E, int k_in, int c, int h) {
    const int k = uniform_i(k_in);
    double d, e;
#ifdef OSOT_NHQP_PHASES
    long long t0_ = (long long)clock64();
#endif
    sym_tred2_32(K, E, k, c, h, d, e);
#ifdef OSOT_NHQP_PHASES
    if (blockIdx.x == 0 && threadIdx.x == 0) printf("PHASE    tred2 k=%d %lld\n", k, (long long)clock64() - t0_);
    t0_ = (long long)clock64();
#endif
#ifndef OSOT_X_NO_BISECT
    // (the bisection has a floor of ~20 k clocks -- 36 sweeps whatever the size -- and QL costs ~470 k^2: they cross near k = 10)
    const bool done = (k >= 10) && uniform_b(sym_bisect_32(K, E, k, c, h, d, e));
#else
    const bool done = false;
#endif
    if (!done) sym_ql_32(K, k, c, h, d, e);
#ifdef OSOT_NHQP_PHASES
    if (blockIdx.x == 0 && threadIdx.x == 0) printf("PHASE    bisect/ql done=%d %lld\n", (int)done, (long long)clock64() - t0_);
#endif
    sym_eig_finish_32(K, E, k, c, h, d);
    if (h == 0) { K[c * kNS + 32] = 0.0; E[c * kNS + 32] = 0.0; }      // (the padding columns carried d and e)
    wave_sync();
}

// The eigenVALUES of K alone (k >= 2): lane c < k (both halves) gets number c in ascending order.  Tridiagonal form without the
// accumulation of Q, then the bisection sweeps: ~60 % of the full decomposition's clocks at k = 24.  K is destroyed, E's first row
// and the two padding columns are scratch.  For the level preparation's common case -- a full-rank level from which nothing is
// lifted -- the singular values are all it needs: the null space then comes from a Householder QR of the level's own rows.
// ... and with the eigenvectors of the eigenvalues below thr2 times the largest as well (the nl smallest, columns 0 .. nl - 1 of E):
// what regularize_A_b needs of a full-rank level whose smallest singular values it lifts.  false: rank deficient at noise level or
// a cluster among the lifted ones -- rebuild the Gram matrix and take sym_eig32_fast.  lam: eigenvalue number c (ascending).
__device__ __forceinline__ bool sym_eig_selected32(double* K, double* E, int k_in, int c, int h, double thr2, double& lam, int& nl) {
    const int k = uniform_i(k_in);
    double d, e;
    if constexpr (kTred2Regs) sym_tred2_32_regs(K, k, c, h, d, e);
    else sym_tred2_32<false>(K, E, k, c, h, d, e);
    int nsel = 0;
    const bool ok = uniform_b(sym_bisect_32<2>(K, E, k, c, h, d, e, thr2, &nsel));
    wave_sync();
    if (h == 0) { K[c * kNS + 32] = 0.0; E[c * kNS + 32] = 0.0; }
    wave_sync();
    lam = d; nl = uniform_i(nsel);
    return ok;
}
__device__ __forceinline__ double sym_eigvals32(double* K, double* E, int k_in, int c, int h) {
    const int k = uniform_i(k_in);
    double d, e;
    if constexpr (kTred2Regs) sym_tred2_32_regs(K, k, c, h, d, e);
    else sym_tred2_32<false>(K, E, k, c, h, d, e);
    sym_bisect_32<0>(K, E, k, c, h, d, e);
    wave_sync();
    if (h == 0) { K[c * kNS + 32] = 0.0; E[c * kNS + 32] = 0.0; }
    wave_sync();
    return d;
}

__device__ __forceinline__ void sym_eig32(double* K, double* E, int k_in, int c, int h) {
    const int k = uniform_i(k_in);
    double d, e;
    sym_tred2_32(K, E, k, c, h, d, e);
    sym_ql_32(K, k, c, h, d, e);
    sym_eig_finish_32(K, E, k, c, h, d);
}

// X'diag(w) X of the LDS matrix X ([rows][S], zero rows beyond m up to the next multiple of four) on the fp64 matrix core: T x T tiles
// of 16 x 16 (upper triangle), one v_mfma_f64_16x16x4 per tile per four rows of X; a lane's operand X[r0 + (lane >> 4)][16 J + (lane & 15)]
// serves as A AND as B (the tile (I, J) is x_I' x_J), so a k-step costs T LDS reads.  Columns >= the logical width read the row's padding /
// the next row: finite numbers that only reach tile entries nobody stores.  w: LDS weights per row (null: ones).  Optional second
// operand set: `ns` columns idx2[0 .. ns) (idx2 null: columns 0 .. ns - 1) of the matrix Y ([yrows][SY], lane's ROW 16 J + (lane & 15), rows
// beyond yrows read as zero) scaled by sqrt(s2): + s2 Y2 Y2'.
// acc[4 tile + r] = entry (16 I + (lane >> 4) + 4 r, 16 J + (lane & 15)) of tile (I, J), tiles in row-major upper-triangle order.
template <int T>
__device__ __forceinline__ void nhqp_tile_gram(const double* X, int S, int m, const double* w, const double* Y, int SY, int yrows, const int* idx2, int ns, double s2,
                                               int lane, double (&acc)[2 * T * (T + 1)]) {
    const int q = lane >> 4, a = lane & 15;
    v4f64 t[T * (T + 1) / 2];
#pragma unroll
    for (int u = 0; u < T * (T + 1) / 2; ++u) { t[u][0] = 0.0; t[u][1] = 0.0; t[u][2] = 0.0; t[u][3] = 0.0; }
    for (int r0 = 0; r0 < m; r0 += 4) {
        double x[T];
#pragma unroll
        for (int J = 0; J < T; ++J) x[J] = X[(r0 + q) * S + 16 * J + a];
        const double wr = w ? w[r0 + q] : 1.0;
        int u = 0;
#pragma unroll
        for (int I = 0; I < T; ++I) {
            const double xa = wr * x[I];
#pragma unroll
            for (int J = I; J < T; ++J) { t[u] = mfma_f64_16x16x4(xa, x[J], t[u]); ++u; }
        }
    }
    if (ns > 0) {
        const double rs2 = sqrt(s2);
        for (int t0 = 0; t0 < ns; t0 += 4) {
            const bool live = t0 + q < ns;
            const int tq = live ? t0 + q : 0;
            const int ec = idx2 ? idx2[tq] : tq;
            double y[T];
#pragma unroll
            for (int J = 0; J < T; ++J) {
                const int yr = 16 * J + a;
                const double v = Y[((yr < yrows) ? yr : 0) * SY + ec];
                y[J] = (live && yr < yrows) ? rs2 * v : 0.0;
            }
            int u = 0;
#pragma unroll
            for (int I = 0; I < T; ++I)
#pragma unroll
                for (int J = I; J < T; ++J) { t[u] = mfma_f64_16x16x4(y[I], y[J], t[u]); ++u; }
        }
    }
#pragma unroll
    for (int u = 0; u < T * (T + 1) / 2; ++u) { acc[4 * u] = t[u][0]; acc[4 * u + 1] = t[u][1]; acc[4 * u + 2] = t[u][2]; acc[4 * u + 3] = t[u][3]; }
}
// the tiles of nhqp_tile_gram -> a symmetric k x k matrix with row stride ld (LDS or HBM), both triangles
template <int T>
__device__ __forceinline__ void nhqp_tile_store(const double (&acc)[2 * T * (T + 1)], double* M, int ld, int k, int lane) {
    const int q = lane >> 4, a = lane & 15;
    int u = 0;
#pragma unroll
    for (int I = 0; I < T; ++I)
#pragma unroll
        for (int J = I; J < T; ++J) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * I + q + 4 * r, j = 16 * J + a;
                if (i < k && j < k) { M[i * ld + j] = acc[4 * u + r]; if (I != J) M[j * ld + i] = acc[4 * u + r]; }
            }
            ++u;
        }
}

// H (k x k, row stride ld) = X'diag(w) X + s2 Y2 Y2' through the tile products above, T = ceil(k / 16) tiles a side
// TMAX (round 6): the largest tile count the CALLER can meet.  The 32-column preparation kernel never has k > 32, but the T = 3 / 4 branches
// it could not take still cost it their accumulators -- 80 AGPRs beside 177 VGPRs: ONE register over the 256 that let two wavefronts share a
// SIMD (profiles/r05_v4_kernel_resources.txt: every preparation kernel at occupancy 1).  With TMAX = 2 it keeps 24.
template <int TMAX = 4>
__device__ __forceinline__ void nhqp_gram_to(double* M, int ld, int k, const double* X, int S, int m, const double* w,
                                             const double* Y, int SY, int yrows, const int* idx2, int ns, double s2, int lane) {
    const int T = uniform_i((k + 15) >> 4);
    if (T <= 1) { double acc[4]; nhqp_tile_gram<1>(X, S, m, w, Y, SY, yrows, idx2, ns, s2, lane, acc); nhqp_tile_store<1>(acc, M, ld, k, lane); }
    else if (T == 2 || TMAX <= 2) { double acc[12]; nhqp_tile_gram<2>(X, S, m, w, Y, SY, yrows, idx2, ns, s2, lane, acc); nhqp_tile_store<2>(acc, M, ld, k, lane); }
    else if constexpr (TMAX >= 3) {
        if (T == 3 || TMAX == 3) { double acc[24]; nhqp_tile_gram<3>(X, S, m, w, Y, SY, yrows, idx2, ns, s2, lane, acc); nhqp_tile_store<3>(acc, M, ld, k, lane); }
        else if constexpr (TMAX >= 4) { double acc[40]; nhqp_tile_gram<4>(X, S, m, w, Y, SY, yrows, idx2, ns, s2, lane, acc); nhqp_tile_store<4>(acc, M, ld, k, lane); }
    }
}

// Sixteen rows 16 I .. 16 I + 15 of a row-major HBM matrix M (ld n, `rows` of them) against the LDS matrix Nl (n x nf, stride S, zero rows
// beyond n) on the fp64 matrix core: t[J] = tile (I, J) of M Nl.  With q0v (LDS, n entries, zero beyond) the product's column nf is M q0
// (the B operand of that one column comes from q0v instead of Nl: needs nf < 16 T).  A lane's A operand M[16 I + (lane & 15)][k0 + (lane >> 4)]
// is loaded straight from HBM / L2 (clamped, masked), its B operands Nl[k0 + (lane >> 4)][16 J + (lane & 15)] from LDS.
template <int T>
__device__ __forceinline__ void nhqp_rows16_times_N(const double* M, int rows, int n, const double* Nl, int S, int nf, const double* q0v,
                                                    int I, int lane, v4f64 (&t)[T]) {
    const int q = lane >> 4, a = lane & 15;
#pragma unroll
    for (int J = 0; J < T; ++J) { t[J][0] = 0.0; t[J][1] = 0.0; t[J][2] = 0.0; t[J][3] = 0.0; }
    const int row = 16 * I + a;
    const double* Mr = M + (long long)((row < rows) ? row : rows - 1) * n;
    for (int k0 = 0; k0 < n; k0 += 8) {             // two k-steps per trip: 2 HBM + 2 T LDS reads in flight
        double xa[2], xb[2][T];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = k0 + 4 * u + q;
            const bool live = k < n && row < rows;
            const double va = Mr[(k < n) ? k : n - 1];
            xa[u] = live ? va : 0.0;
            const int kc = (k < n) ? k : 0;
#pragma unroll
            for (int J = 0; J < T; ++J) {
                const int col = 16 * J + a;
                const double vb = Nl[kc * S + ((col < S) ? col : 0)];
                const double vq = q0v ? q0v[kc] : 0.0;
                xb[u][J] = (k < n) ? ((q0v && col == nf) ? vq : vb) : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int J = 0; J < T; ++J) t[J] = mfma_f64_16x16x4(xa[u], xb[u][J], t[J]);
    }
}

// MR = row capacity of A N (32 or 64).  LDS: three 32 x 33 work matrices + A N = 25.3 KB (MR = 32: six wavefronts per CU;
// the first version held five matrices and a 64-row A N, 50 KB, three per CU).  The buffers are re-used as the level goes:
//   NE : N (until the constraints are written, right after A N)  ->  eigenvectors E  ->  V2 on the row side
//   K  : Gram matrix -> its eigenvalues on the diagonal -> the reflectors of the complement / V2 on the column side
// developer knob (tools/build_variant.sh NAME -DOSOT_NHQP_PHASES): instance 0 prints the clock count of every phase of its level
// A level with a NON-DIAGONAL weight matrix (Task::setWeight(W), nHQP.cpp:381-382: H = AN'W AN, g = -AN'W b0 with the REGULARISED AN and
// b0 -- which is why W itself is needed: W u of a lifted null triplet is not in range(W A)).  The kernels below run their own H / g stage on
// diag(W); this adds the rest, H += AN'(W - diag W) AN and g -= AN'(W - diag W) b0, column c of H and entry c of g by the lane that
// stored them (its own stores, read back and rewritten: no other lane touches them).  A plain loop -- m (m + nf) steps per lane with a
// read-modify-write of HBM in the inner one: a weight matrix is a rare configuration, not a benchmark.
__device__ inline void nhqp_dense_weight_correction(const DevNhqp& Q, long long inst, const double* AN, int S, const double* b0, int c) {
    const int m = Q.m, nf = Q.nf;
    if (c < 0 || c >= nf) return;
    const double* Wg = Q.Wd + inst * (long long)m * m;
    double* Hg = Q.H + inst * (long long)nf * nf;
    double gcor = 0.0;
    for (int r = 0; r < m; ++r) {
        double u = 0.0;                               // ((W - diag W) AN)[r][c]
        for (int s2 = 0; s2 < m; ++s2) if (s2 != r) u = fma(Wg[r * m + s2], AN[s2 * S + c], u);
        gcor = fma(-u, b0[r], gcor);                  // (W symmetric: AN'(W_off b0) = (W_off AN)'b0)
        for (int i = 0; i < nf; ++i) Hg[i * nf + c] = fma(AN[r * S + i], u, Hg[i * nf + c]);
    }
    Q.g[inst * nf + c] += gcor;
}

// preparation (s_memtime deltas of one wave among the CU's six)
#ifdef OSOT_NHQP_PHASES
#define NHQP_PHASE(tag) do { const long long t_ = (long long)clock64(); if (inst == 0 && lane == 0) printf("PHASE L%d " tag " %lld\n", Q.level, t_ - ph_t_); ph_t_ = (long long)clock64(); } while (0)
#else
#define NHQP_PHASE(tag) do { } while (0)
#endif
template <int MR>
__global__ void __launch_bounds__(64) osot_nhqp_prepare_kernel(const DevNhqp Q) {
    OSOT_STATIC_LDS(double, AN, MR * kNS);     // A N (m x nf)
    OSOT_STATIC_LDS(double, NE, 32 * kNS);     // N (n x nf), then E, then V2 (row side)
    OSOT_STATIC_LDS(double, K, 32 * kNS);      // Gram matrix -> diag(lambda) -> V1 scratch / V2 (column side)
    double* Nl = NE;
    double* E = NE;
    OSOT_STATIC_LDS(double, b0, 64);
    OSOT_STATIC_LDS(double, vec, 64);          // staging vector
    OSOT_STATIC_LDS(double, sig, 32);          // singular values, sorted descending
    OSOT_STATIC_LDS(int, idx, 32);             // idx[pos] = eigen-column holding the pos-th largest
    OSOT_STATIC_LDS(double, refl_beta, 32);    // 2 / |v|^2 of the Householder reflectors of the right singular vectors
    const long long inst = blockIdx.x;
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    if (inst >= Q.B) return;
    if (Q.status && Q.status[inst] != 0) return;
    const int n = Q.n, m = Q.m, ma = Q.ma, nf = Q.nf, ns = Q.ns;
    const bool first = Q.level == 0;
#ifdef OSOT_NHQP_PHASES
    long long ph_t_ = (long long)clock64();
#endif
    const double* A = Q.A ? Q.A + inst * (long long)ma * n : nullptr;
    // ---- N -> LDS (identity at the first level)
    for (int e = lane; e < 32 * kNS; e += 64) { Nl[e] = 0.0; K[e] = 0.0; }
    for (int e = lane; e < MR * kNS; e += 64) AN[e] = 0.0;
    wave_sync();
    if (first) { if (h == 0 && c < n) Nl[c * kNS + c] = 1.0; }
    else {
        // (fixed trip count, clamped addresses: the 16 loads of a lane are in flight together; a run-time loop waited for each)
        const double* Ng = Q.N + inst * (long long)n * n;
        const int cn = (c < nf) ? c : nf - 1;
        double nv[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) { const int i = 2 * t + h; nv[t] = Ng[((i < n) ? i : n - 1) * n + cn]; }
#pragma unroll
        for (int t = 0; t < 16; ++t) { const int i = 2 * t + h; if (i < n && c < nf) Nl[i * kNS + c] = nv[t]; }
    }
    if (lane < 64) vec[lane] = (!first && lane < n) ? Q.q0[inst * n + lane] : 0.0;     // q0 staged
    wave_sync();
    NHQP_PHASE("loadN");
    // ---- AN = A N (stored rows: a row of A against the columns of N, i split over the halves; identity rows: rows of N)
    // (A goes through the Gram buffer, free until the Gram matrix is formed, 32 rows at a time with coalesced loads: read
    //  element by element inside the product it was a uniform-address HBM load per multiply-add)
    double aq_row = 0.0;      // lane = stored row r: (A q0)_r
    for (int rb = 0; rb < ma; rb += 32) {
        const int nr = (ma - rb < 32) ? ma - rb : 32;
        wave_sync();
        {
            const int ca = (c < n) ? c : n - 1;
            double av[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) { const int r = 2 * t + h; av[t] = A[(rb + ((r < nr) ? r : nr - 1)) * n + ca]; }
#pragma unroll
            for (int t = 0; t < 16; ++t) { const int r = 2 * t + h; if (r < nr) K[r * kNS + c] = (c < n) ? av[t] : 0.0; }
        }
        wave_sync();
        for (int r = 0; r < nr; ++r) {
            const double acc = halfsum<32>(dot_half<16>(K + r * kNS, 1, Nl + c, kNS, h));
            if (h == 0 && c < nf) AN[(rb + r) * kNS + c] = acc;
            // (A q0)_r for b0 = b - A q0, from the staged row (it was a loop of dependent HBM loads per row)
            const double aq = halfsum<32>(dot_half<16>(K + r * kNS, 1, vec, 1, h));
            if (lane == rb + r) aq_row = aq;
        }
    }
    wave_sync();
    for (int r = ma + h; r < m; r += 2) if (c < nf) AN[r * kNS + c] = Nl[(r - ma) * kNS + c];
    // ---- b0 = b - A q0 (lane = row)
    {
        double v = 0.0;
        if (lane < m) {
            v = Q.b[inst * m + lane];
            if (!first) {
                if (lane < ma) v -= aq_row;
                else v -= vec[lane - ma];
            }
        }
        if ((Q.zero_rows >> lane) & 1ull) v = (lane < m) ? Q.b[inst * m + lane] : 0.0;     // inactive task: A = 0, so b0 = b
        b0[lane] = v;              // (zero beyond m: the fixed-trip products below read all 64 entries)
    }
    if (Q.zero_rows) {             // rows of inactive tasks are zero rows of A N
        wave_sync();
        for (int r = 0; r < m; ++r) if (((Q.zero_rows >> r) & 1ull) && h == 0) AN[r * kNS + c] = 0.0;
    }
    wave_sync();
    NHQP_PHASE("AN+b0");
    // ---- constraints in z-coordinates (levels below the first): rows [C N; N], bounds shifted by q0
    if (!first) {
        const int nr = Q.nc + (Q.has_box ? n : 0);
        double* Rg = Q.R + inst * (long long)nr * nf;
        if (lane < 64) vec[lane] = (lane < n) ? Q.q0[inst * n + lane] : 0.0;
        wave_sync();
        // (the rows of C go through the Gram buffer like the rows of A above -- 32 at a time, coalesced; element by element inside
        //  the product it was a uniform-address HBM load per multiply-add -- and the bounds of a block are loaded together, lane = row)
        const double* Cg = Q.C + inst * (long long)Q.nc * n;
        for (int rb = 0; rb < Q.nc; rb += 32) {
            const int nr32 = (Q.nc - rb < 32) ? Q.nc - rb : 32;
            wave_sync();
            {
                const int ca = (c < n) ? c : n - 1;
                double cv[16];
#pragma unroll
                for (int t = 0; t < 16; ++t) { const int r = 2 * t + h; cv[t] = Cg[(rb + ((r < nr32) ? r : nr32 - 1)) * n + ca]; }
#pragma unroll
                for (int t = 0; t < 16; ++t) { const int r = 2 * t + h; if (r < nr32) K[r * kNS + c] = (c < n) ? cv[t] : 0.0; }
            }
            wave_sync();
            double cq_row = 0.0;               // lane = row r of the block: (C q0)_r
            for (int r = 0; r < nr32; ++r) {
                const double acc = halfsum<32>(dot_half<16>(K + r * kNS, 1, Nl + c, kNS, h));
                if (h == 0 && c < nf) Rg[(rb + r) * nf + c] = acc;
                const double cq = halfsum<32>(dot_half<16>(K + r * kNS, 1, vec, 1, h));
                if (lane == r) cq_row = cq;
            }
            if (lane < nr32) {
                const int r = rb + lane;
                const double lo = Q.lo[inst * Q.nc + r], up = Q.up[inst * Q.nc + r];
                Q.rlo[inst * nr + r] = (lo <= -1.0e20) ? -1.0e20 : lo - cq_row;
                Q.rup[inst * nr + r] = (up >= 1.0e20) ? 1.0e20 : up - cq_row;
            }
        }
        if (Q.has_box) {
            for (int i = h; i < n; i += 2) if (c < nf) Rg[(Q.nc + i) * nf + c] = Nl[i * kNS + c];
            if (lane < n) {
                const double l = Q.l[inst * n + lane], u = Q.u[inst * n + lane];
                Q.rlo[inst * nr + Q.nc + lane] = (l <= -1.0e20) ? -1.0e20 : l - vec[lane];
                Q.rup[inst * nr + Q.nc + lane] = (u >= 1.0e20) ? 1.0e20 : u - vec[lane];
            }
        }
    }
    wave_sync();      // N is dead from here on: its buffer becomes E (and later V2)
    for (int e = lane; e < 32 * kNS; e += 64) K[e] = 0.0;       // (the staging of A / C rows is over: the Gram matrix starts from zero)
    wave_sync();
    NHQP_PHASE("constr");
    // ---- Gram matrix of the small side
    const bool rowside = m <= nf;
    const int k = rowside ? m : nf;
    constexpr double kSvNoise = 1.0e-7;
    auto build_gram = [&]() {
    if (rowside) {          // K[a][c] = <row a, row c> of AN
        const int cm = (c < m) ? c : 0;
        for (int a = 0; a < m; a += 2) {       // (two rows per trip: independent chains; A N is zero beyond column nf and row m)
            const double acc0 = halfsum<32>(dot_half<16>(AN + a * kNS, 1, AN + cm * kNS, 1, h));
            const double acc1 = halfsum<32>(dot_half<16>(AN + (a + 1) * kNS, 1, AN + cm * kNS, 1, h));
            if (h == 0 && c < m) { K[a * kNS + c] = acc0; if (a + 1 < m) K[(a + 1) * kNS + c] = acc1; }
        }
    } else {                // K[a][c] = <column a, column c>
        const int cf = (c < nf) ? c : 0;
        for (int a = 0; a < nf; a += 2) {
            const double acc0 = halfsum<32>(dot_half<MR / 2>(AN + a, kNS, AN + cf, kNS, h));
            const double acc1 = halfsum<32>(dot_half<MR / 2>(AN + a + 1, kNS, AN + cf, kNS, h));
            if (h == 0 && c < nf) { K[a * kNS + c] = acc0; if (a + 1 < nf) K[(a + 1) * kNS + c] = acc1; }
        }
    }
    wave_sync();
    };
    build_gram();
    NHQP_PHASE("gram");
    // ---- the common case first: a row-side level of full rank from which nothing is lifted needs its singular VALUES only
    // (sv_max for the selective regularisation, the smallest against the lifting threshold) -- the right singular vectors served
    // to span the row space of A N, and the rows themselves do that: the Householder orthonormalisation below then starts from
    // V1 = (A N)' and the completion is the same null space.  Eigenvalues without the accumulation of Q, the twisted
    // factorisations and V = Q Y: ~60 % of the decomposition's clocks.  If a singular value turns out to be at noise level or
    // below the lifting threshold, the Gram matrix is rebuilt and the full decomposition runs (the rare case pays twice).
    // Tried only with the A / b regularisation OFF: with it on, a level whose smallest singular value is under min_sv_ratio of its
    // largest needs the vectors of the lifted triplets after all -- 65 % of the instances of BASELINE config 3's level 1 (ratios
    // 0.02 .. 0.09 against the default 0.05) -- and paying twice there cost more than the others gained (nHQP 4.18 -> 3.82 M).
    bool values_only = false;
    if (rowside && k >= 10 && nf - ns >= k) {
        double lamv; int nlift = 0;
        values_only = sym_eig_selected32(K, E, k, c, h, Q.ab_reg ? Q.thr * Q.thr * (1.0 + 1.0e-9) : 0.0, lamv, nlift);
        const double svc = sqrt(lamv > 0.0 ? lamv : 0.0);
        if (values_only) {
            if (h == 0 && c < k) { idx[k - 1 - c] = c; sig[k - 1 - c] = svc; }
            wave_sync();
        } else {
            for (int e = lane; e < 32 * kNS; e += 64) K[e] = 0.0;
            wave_sync();
            build_gram();
        }
    }
    if (!values_only) {
    sym_eig32_fast(K, E, k, c, h);
    NHQP_PHASE("eig");
    // ---- singular values, sorted descending: pos = number of eigenvalues ahead of mine
    {
        const double lam = (c < k) ? K[c * kNS + c] : -1.0;
        if (h == 0) vec[c] = lam;
        wave_sync();
        int pos = 0;
        for (int d = 0; d < k; ++d) { const double ld = vec[d]; pos += (ld > lam || (ld == lam && d < c)) ? 1 : 0; }
        if (h == 0 && c < k) { idx[pos] = c; sig[pos] = sqrt(lam > 0.0 ? lam : 0.0); }
        wave_sync();
    }
    }
    const double sv_max = sig[0];
    // ---- right singular vectors on the ROW side (m <= nf).  v_i = AN'u_i / |AN'u_i| only exists for sv_i > 0: for a singular
    // value at round-off level (a rank-deficient level: duplicated or dependent task rows) that product is noise inside the
    // row space, while the reference's full V (Eigen, nHQP.cpp:373) holds a NULL-SPACE vector there -- and regularize_A_b
    // lifts exactly those triplets, so a noise v_i would put the lifted singular value on a genuine task direction (ADVICE
    // r2: 2-4e-2 off on a stack with a duplicated row).  So: the rho well-defined v_i (sv_i >= kSvNoise sv_max; the Gram
    // route resolves singular values down to ~1e-8 sv_max) are orthonormalised by Householder reflections H_1 .. H_r'
    // (r' = min(rho, nf - ns)), and every other column of V is taken from the completion Q = H_1 .. H_r' : column j >= r' is
    // Q e_j, orthogonal to v_1 .. v_r'.  The lifted null triplets use Q e_i, the next level's null space Q e_(r + t).
    int rho = 0;
    for (int i = 0; i < k; ++i) rho += (sig[i] >= kSvNoise * sv_max && sig[i] > 0.0) ? 1 : 0;
    const int r_next = nf - ns;
    const int nrefl = rowside ? (rho < r_next ? rho : r_next) : 0;
    const bool need_refl = rowside && (ns > 0 || (Q.ab_reg && rho < k));
    double* V1 = K;          // K is free now (its diagonal went into sig[]): V1[t][i] = component t of v_i, then reflector i
    if (need_refl) {
        if (values_only) {          // V1 = (A N)': column i = row i of A N (nrefl = m here); the reflections orthonormalise them
            double t16[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) t16[t] = AN[(2 * t + h) * kNS + c];
            wave_sync();
#pragma unroll
            for (int t = 0; t < 16; ++t) V1[c * kNS + 2 * t + h] = (2 * t + h < nrefl && c < nf) ? t16[t] : 0.0;
        }
        for (int i = 0; i < (values_only ? 0 : nrefl); ++i) {
            const int ec = idx[i];
            // (row side: m <= 32; A N is zero beyond row m and E finite there -- fixed-trip product, reads in flight together)
            double vv = halfsum<32>(dot_half<16>(AN + c, kNS, E + ec, kNS, h));
            const double nrm2 = colsum<32>((c < nf) ? vv * vv : 0.0);
            double nsq, nrs;
            fast_sqrt_rsqrt(nrm2 > 0.0 ? nrm2 : 1.0, nsq, nrs);
            vv = (nrm2 > 0.0 && c < nf) ? vv * nrs : 0.0;
            if (h == 0) V1[c * kNS + i] = vv;                  // (zero beyond nf: the products below run over all 32 rows)
        }
        wave_sync();
        // Householder vectors: for column i, reflect x = V1[i:, i] onto alpha e_i; apply to the later columns.  The update of the
        // later columns is done with lane = COLUMN j (its 32 components split over the halves, fixed trip counts): with lane =
        // component it was one wave reduction and two barriers per (i, j) pair -- 276 of them for 24 reflectors.
        // (round 6: lane j keeping column j in 32 registers through all the steps -- norms lane-local, the reflector as v_readlane
        //  broadcasts, no LDS round trip inside the loop -- was built and measured 1.5-2 % SLOWER at config 3: twice the vector
        //  instructions, both halves carrying all 32 components, for a shorter chain that the other sub-batches' kernels hide anyway)
        for (int i = 0; i < nrefl; ++i) {
            const double v1ci = V1[c * kNS + i];
            const double x = (c >= i && c < nf) ? v1ci : 0.0;
            const double nrm2 = colsum<32>(x * x);
            const double xi = bcast(x, i);
            double nsq, nrs;
            fast_sqrt_rsqrt(nrm2 > 0.0 ? nrm2 : 1.0, nsq, nrs);
            if (!(nrm2 > 0.0)) nsq = 0.0;
            const double alpha = (xi > 0.0) ? -nsq : nsq;
            const double hv = (c == i) ? x - alpha : x;       // reflector v (zero above i)
            // |v|^2 = |x|^2 - x_i^2 + (x_i - alpha)^2 = 2 (|x|^2 + |x_i| |x|): no second reduction
            const double vn2h = fma(fabs(xi), nsq, nrm2);
            const double beta = (vn2h > 0.0) ? fast_rcp(vn2h) : 0.0;
            wave_sync();
            if (h == 0) { V1[c * kNS + i] = hv; vec[c] = hv; }        // keep the reflector in place of the column
            wave_sync();
            {
                const int j = (c > i && c < nrefl) ? c : i;     // (idle lanes shadow the reflector's own column; nothing stored)
                double y[16], dot = 0.0;
#pragma unroll
                for (int t = 0; t < 16; ++t) { y[t] = V1[(2 * t + h) * kNS + j]; dot = fma(vec[2 * t + h], y[t], dot); }
                dot = halfsum<32>(dot) * beta;
                if (c > i && c < nrefl) {
#pragma unroll
                    for (int t = 0; t < 16; ++t) V1[(2 * t + h) * kNS + j] = fma(-dot, vec[2 * t + h], y[t]);
                }
            }
            if (h == 0 && c == 0) refl_beta[i] = beta;
            wave_sync();
        }
    }
    NHQP_PHASE("V1refl");
    auto completion_column = [&](int j) -> double {           // (Q e_j)[c], j >= nrefl
        double y = (c == j) ? 1.0 : 0.0;
        for (int i = nrefl - 1; i >= 0; --i) {
            const double hv = (c >= i && c < nf) ? V1[c * kNS + i] : 0.0;
            const double dot = colsum<32>(hv * y);
            y -= refl_beta[i] * dot * hv;
        }
        return y;
    };
    // ---- A / b regularisation (regularize_A_b, nHQP.cpp:236-279).  For singular triplet i (sorted): the known factor is a
    // column of E, the other one is a normalised product with AN.  AN <- AN + sum_i (sv'_i - sv_i) u_i v_i'  over the
    // lifted ones;  b0 <- sum_i d_i (u_i'b0) u_i  (+ the part of b0 in the complement of range(U_k) only when U_k is all
    // of U, i.e. never when m > nf: "b0(i) = 0 for i >= sv.size()").
    if (Q.ab_reg) {
        double bnew = 0.0;                       // lane = row r (< m): accumulates the new b0 when it is rebuilt (m > nf)
        const bool rebuild = !rowside;           // m > nf: b0 is the sum over the nf triplets
        for (int i = 0; i < k; ++i) {
            const double sv = sig[i];
            const bool lift = sv < Q.thr * sv_max;
            if (!lift && !rebuild) continue;
            const int ec = idx[i];
            // u (m entries, lane = row via vec) and v (nf entries, lane c)
            double uu = 0.0, vv = 0.0;
            if (rowside) {
                uu = (lane < m) ? E[lane * kNS + ec] : 0.0;                       // u_i = column of E
                if (i >= rho) {                                                   // null triplet: v_i from the completion
                    const double qc = completion_column(i);                       // (a wave collective: every lane calls it)
                    vv = (c < nf) ? qc : 0.0;
                } else {
                    double acc = 0.0;                                             // v_i ~ AN' u_i
                    if (c < nf) for (int r = h; r < m; r += 2) acc = fma(AN[r * kNS + c], E[r * kNS + ec], acc);
                    vv = halfsum<32>(acc);
                    const double nrm2 = colsum<32>((c < nf) ? vv * vv : 0.0);
                    vv = (nrm2 > 0.0) ? vv / sqrt(nrm2) : 0.0;
                }
            } else {
                vv = (c < nf) ? E[c * kNS + ec] : 0.0;                            // v_i = column of E
                double acc = 0.0;                                                 // u_i ~ AN v_i (lane = row, all 64 lanes)
                if (lane < m) for (int t = 0; t < nf; ++t) acc = fma(AN[lane * kNS + t], E[t * kNS + ec], acc);
                const double nrm2 = sum64(acc * acc);
                uu = (nrm2 > 0.0) ? acc / sqrt(nrm2) : 0.0;
            }
            const double ub = sum64((lane < m) ? uu * b0[lane] : 0.0);           // u_i' b0
            double d = 1.0, svn = sv;
            if (lift) { d = sv / (Q.thr * sv_max); svn = (Q.thr * sv_max) * (Q.thr * sv_max) / (sv + Q.thr / 100.0); }
            if (rebuild) bnew = fma(d * ub, uu, bnew);
            else if (lane < m) b0[lane] -= (1.0 - d) * ub * uu;
            if (lift) {         // AN += (sv' - sv) u v'
                if (lane < 64) vec[lane] = (lane < m) ? uu : 0.0;
                wave_sync();
                const double dl = svn - sv;
                for (int r = h; r < m; r += 2) if (c < nf) AN[r * kNS + c] = fma(dl * vec[r], vv, AN[r * kNS + c]);
                wave_sync();
            }
        }
        if (rebuild && lane < m) b0[lane] = bnew;
        wave_sync();
    }
    NHQP_PHASE("ABreg");
    // ---- null-space basis V2 (nf x ns) for the next level and for the selective regularisation
    double* V2 = rowside ? NE : K;    // (E's last use on the row side is the construction of V1 below; K's on the column side is sig[])
    if (ns > 0) {
        if (!rowside) {
            // columns of E for the ns smallest eigenvalues
            for (int t = 0; t < ns; ++t) { const int ec = idx[k - ns + t]; if (h == 0 && c < nf) V2[c * kNS + t] = E[c * kNS + ec]; }
            wave_sync();
        } else {
            // the last ns columns of the completion built above: Q e_(r + t), r = nf - ns (orthogonal to the leading
            // well-defined right singular vectors)
            // All ns columns at once, lane = column t with its 32 components in registers (split over the halves): one column
            // at a time (completion_column) is a wave reduction per reflector and column.
            {
                double y[16];
#pragma unroll
                for (int s = 0; s < 16; ++s) y[s] = (2 * s + h == r_next + c) ? 1.0 : 0.0;
                for (int i = nrefl - 1; i >= 0; --i) {
                    double hv[16], dot = 0.0;
#pragma unroll
                    for (int s = 0; s < 16; ++s) {             // (the reflector: zero above i, V1 is zero beyond nf)
                        const int r = 2 * s + h;
                        const double v = V1[r * kNS + i];
                        hv[s] = (r >= i) ? v : 0.0;
                        dot = fma(hv[s], y[s], dot);
                    }
                    dot = halfsum<32>(dot) * refl_beta[i];
#pragma unroll
                    for (int s = 0; s < 16; ++s) y[s] = fma(-dot, hv[s], y[s]);
                }
                wave_sync();
                if (c < ns) {
#pragma unroll
                    for (int s = 0; s < 16; ++s) if (2 * s + h < nf) V2[(2 * s + h) * kNS + c] = y[s];
                }
            }
            wave_sync();
        }
    }
    NHQP_PHASE("V2");
    // ---- H = AN' W AN (+ sv_max V2 V2'), g = -AN' W b0   ->  HBM, nf x nf row-major
    {
        const double* w = Q.w ? Q.w + inst * m : nullptr;
        // (my column of W A N in registers, fixed trip counts: see dot_half)
        vec[lane] = (lane < m) ? (Q.Wd ? Q.Wd[inst * (long long)m * m + lane * (m + 1)] : (w ? w[lane] : 1.0)) : 0.0;   // (dense W: its diagonal here, the rest below)
        wave_sync();
        double* Hg = Q.H + inst * (long long)nf * nf;
        const int cc = (c < nf) ? c : 0;
        double wan[MR / 2];                         // w_r (A N)[r][c] for r = 2 t + h
        double gacc = 0.0;
#pragma unroll
        for (int t = 0; t < MR / 2; ++t) {
            const int r = 2 * t + h;
            wan[t] = vec[r] * AN[r * kNS + cc];
            gacc = fma(-wan[t], b0[r], gacc);       // (b0 beyond m: multiplied by w = 0)
        }
        gacc = halfsum<32>(gacc);
        if (h == 0 && c < nf) Q.g[inst * nf + c] = gacc;
        const bool sel = ns > 0 && Q.sel_reg;
#ifndef OSOT_NHQP_H_VALU
        nhqp_gram_to<2>(Hg, nf, nf, AN, kNS, m, vec, V2, kNS, nf, nullptr, sel ? ns : 0, sv_max, lane);     // (round 5: on the matrix core; nf <= 32 here)
#else
        double v2c[16];                             // my row of V2 (zero beyond ns)
#pragma unroll
        for (int t = 0; t < 16; ++t) { const double v = V2[cc * kNS + 2 * t + h]; v2c[t] = (sel && 2 * t + h < ns) ? sv_max * v : 0.0; }
        // (two rows of H per trip: their products are independent chains; row nf of an odd count is computed on row nf - 1 again)
        for (int i0 = 0; i0 < nf; i0 += 2) {
            const int i1 = (i0 + 1 < nf) ? i0 + 1 : i0;
            double a0 = 0.0, a1 = 0.0, c0 = 0.0, c1 = 0.0;
#pragma unroll
            for (int t = 0; t < MR / 2; t += 2) {
                a0 = fma(wan[t], AN[(2 * t + h) * kNS + i0], a0);
                a1 = fma(wan[t + 1], AN[(2 * (t + 1) + h) * kNS + i0], a1);
                c0 = fma(wan[t], AN[(2 * t + h) * kNS + i1], c0);
                c1 = fma(wan[t + 1], AN[(2 * (t + 1) + h) * kNS + i1], c1);
            }
            if (sel) {
#pragma unroll
                for (int t = 0; t < 16; t += 2) {
                    // (BOTH factors masked beyond ns: V2 shares its buffer with the twisted factorisation's work, whose unused part
                    //  may hold Inf after a clamped pivot, and 0 * Inf would poison H -- ADVICE r4)
                    const bool m0 = 2 * t + h < ns, m1 = 2 * (t + 1) + h < ns;
                    const double f00 = V2[i0 * kNS + 2 * t + h], f01 = V2[i0 * kNS + 2 * (t + 1) + h];
                    const double f10 = V2[i1 * kNS + 2 * t + h], f11 = V2[i1 * kNS + 2 * (t + 1) + h];
                    a0 = fma(v2c[t], m0 ? f00 : 0.0, a0);
                    a1 = fma(v2c[t + 1], m1 ? f01 : 0.0, a1);
                    c0 = fma(v2c[t], m0 ? f10 : 0.0, c0);
                    c1 = fma(v2c[t + 1], m1 ? f11 : 0.0, c1);
                }
            }
            const double acc0 = halfsum<32>(a0 + a1), acc1 = halfsum<32>(c0 + c1);
            if (h == 0 && c < nf) { Hg[i0 * nf + c] = acc0; if (i1 != i0) Hg[i1 * nf + c] = acc1; }
        }
#endif
    }
    NHQP_PHASE("Hg");
    if (Q.Wd) { wave_sync(); nhqp_dense_weight_correction(Q, inst, AN, kNS, b0, (h == 0) ? c : -1); }
    // ---- V2 -> HBM (row stride n)
    if (ns > 0 && Q.V2) {
        double* Vg = Q.V2 + inst * (long long)n * n;
        for (int i = h; i < nf; i += 2) if (c < ns) Vg[i * n + c] = V2[i * kNS + c];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: the same level preparation for 32 < n <= 64 variables (the reference's own COMAN has 35 coordinates, BASELINE config 5
// has 50).  One lane per COLUMN of N / A N (64 lanes, no halves); the small-side Gram matrix and its eigen-decomposition stay
// 32-wide (sym_eig32_fast), i.e. min(rows of the level, free variables) <= 32 at every level -- what every stack of the
// reference's example satisfies below its first level and with up to 32 task rows at the first (nhqp_validate refuses the rest).
// Same arithmetic as osot_nhqp_prepare_kernel, block by block.
// LDS, sized by the plan's n at launch (nhqp_prepare64_lds_bytes): rows padded to a multiple of 8 (every product reads its operands
// in chunks of 8 with the reads of a chunk in flight together; the padding is zero), odd row stride RN + 1.  With the fixed 64 x 65
// layout of the first version the COMAN35 plans held 71 KB per wavefront -- two wavefronts per CU, and this kernel is bound by the
// latency of one wavefront's instruction stream; at n = 35 it is 38 KB, four per CU.
constexpr int nhqp64_rows(int n) { return (n + 7) & ~7; }
constexpr int nhqp64_stride(int n) { return nhqp64_rows(n) + 1; }
inline size_t nhqp_prepare64_lds_bytes(int MR, int n) {
    const size_t RN = (size_t)nhqp64_rows(n), S = (size_t)nhqp64_stride(n);
    return sizeof(double) * ((size_t)MR * S + RN * S + RN * kNS + 8 * 64 + 64 + 64 + 32 + 32) + sizeof(int) * 32;
}

template <int MR>
__global__ void __launch_bounds__(64) osot_nhqp_prepare64_kernel(const DevNhqp Q) {
    OSOT_DYNAMIC_LDS(nh_smem);
    const int n = Q.n, m = Q.m, ma = Q.ma, nf = Q.nf, ns = Q.ns;
    const int RN = nhqp64_rows(n), S = nhqp64_stride(n), hr = RN >> 1;
    double* AN = reinterpret_cast<double*>(nh_smem);   // A N (m x nf), MR rows, stride S
    double* NE = AN + MR * S;                          // N (n x nf, RN rows, stride S) -> E (32 x 33 view) -> V2 on the row side (stride S)
    double* K = NE + RN * S;                           // [RN][33]: Gram matrix (k x k) -> V1 / reflectors (nf rows) -> V2 on the column side
    double* stage = K + RN * kNS;                      // [8][64]: eight rows of A / C staged for the products
    double* b0 = stage + 8 * 64;                       // [64]
    double* vec = b0 + 64;                             // [64]
    double* sig = vec + 64;                            // [32]
    double* refl_beta = sig + 32;                      // [32]
    int* idx = reinterpret_cast<int*>(refl_beta + 32); // [32]
    double* Nl = NE;
    double* E = NE;                                    // (stride kNS inside the same buffer: N is dead by then)
    const long long inst = blockIdx.x;
    const int lane = threadIdx.x, c = lane;
    const int c32 = lane & 31, h32 = lane >> 5;        // the lane coordinates sym_eig32 works in (and the split products below)
    if (inst >= Q.B) return;
    if (Q.status && Q.status[inst] != 0) return;
    const bool first = Q.level == 0;
#ifdef OSOT_NHQP_PHASES
    long long ph_t_ = (long long)clock64();
#endif
    const double* A = Q.A ? Q.A + inst * (long long)ma * n : nullptr;
    const int cc = (c < nf) ? c : 0;                   // my column where it exists (reads of idle lanes are discarded)
    const int ln = (lane < n) ? lane : n - 1;
    for (int e = lane; e < RN * S; e += 64) Nl[e] = 0.0;
    for (int e = lane; e < RN * kNS; e += 64) K[e] = 0.0;
    for (int e = lane; e < MR * S; e += 64) AN[e] = 0.0;
    wave_sync();
    if (first) { if (c < n) Nl[c * S + c] = 1.0; }
    else {
        const double* Ng = Q.N + inst * (long long)n * n;
        for (int i0 = 0; i0 < n; i0 += 8) {
            double nv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u; nv[u] = Ng[((i < n) ? i : n - 1) * n + cc]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u; if (i < n && c < nf) Nl[i * S + c] = nv[u]; }
        }
    }
    vec[lane] = (!first && lane < n) ? Q.q0[inst * n + ln] : 0.0;     // q0 staged
    wave_sync();
    NHQP_PHASE("loadN");
    // ---- eight stored rows of A (or C) against N: lane i loads M[r][i] (coalesced), the rows go through LDS, lane c accumulates its
    // column for all eight with N's entries read once per chunk of 8.  Lane 63 runs the same product against q0 instead of a column
    // of N (below the first level nf < n <= 64, so it has no column of its own): (M q0)_r comes out of the same instructions instead
    // of a wave reduction per row.  Returns my column of the eight rows; mq[u] = (M q0)_(rb+u) for everybody.
    auto rows8_times_N = [&](const double* M, int rows, int rb, double (&acc8)[8], double (&mq)[8]) {
        double a8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const double v = M[(long long)((rb + u < rows) ? rb + u : rows - 1) * n + ln]; a8[u] = (rb + u < rows && lane < n) ? v : 0.0; }
        wave_sync();
#pragma unroll
        for (int u = 0; u < 8; ++u) stage[u * 64 + lane] = a8[u];
        wave_sync();
        const bool q0col = !first && nf < 64;          // (nf = 64 below the first level: every level above was empty -- the reduction then)
        const bool q0lane = q0col && lane == 63;
#pragma unroll
        for (int u = 0; u < 8; ++u) acc8[u] = 0.0;
        for (int i0 = 0; i0 < n; i0 += 8) {
            double nl[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const double a = Nl[(i0 + j) * S + cc], b = vec[i0 + j]; nl[j] = q0lane ? b : a; }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc8[u] = fma(stage[u * 64 + i0 + j], nl[j], acc8[u]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) mq[u] = first ? 0.0 : (q0col ? bcast(acc8[u], 63) : colsum<64>(a8[u] * vec[lane]));
    };
    // (round 5) the same products on the fp64 matrix core, sixteen rows at a time (nhqp_rows16_times_N: the A operand straight from HBM / L2,
    // M q0 as column nf of the product -- which needs a spare tile column: nf < 64 below the first level; OSOT_NHQP_ROWS8 keeps the staged form)
#ifndef OSOT_NHQP_ROWS8
    const bool use_tiles = first || nf < 64;
#else
    const bool use_tiles = false;
#endif
    const int Tt = uniform_i((nf + (first ? 0 : 1) + 15) >> 4);
    auto rows16_times_N = [&](const double* M, int rows, int I, auto&& sink) {      // sink(row, col, value) for every entry of the 16 x 16 T tiles
        auto run = [&](auto tc) {
            constexpr int T = decltype(tc)::value;
            v4f64 t[T];
            nhqp_rows16_times_N<T>(M, rows, n, Nl, S, nf, first ? nullptr : vec, I, lane, t);
            const int q4 = lane >> 4, a16 = lane & 15;
#pragma unroll
            for (int J = 0; J < T; ++J)
#pragma unroll
                for (int r = 0; r < 4; ++r) sink(16 * I + q4 + 4 * r, 16 * J + a16, t[J][r]);
        };
        if (Tt <= 1) run(std::integral_constant<int, 1>{});
        else if (Tt == 2) run(std::integral_constant<int, 2>{});
        else if (Tt == 3) run(std::integral_constant<int, 3>{});
        else run(std::integral_constant<int, 4>{});
    };
    double aq_row = 0.0;      // lane = stored row r: (A q0)_r
    if (use_tiles) {
        for (int I = 0; 16 * I < ma; ++I)
            rows16_times_N(A, ma, I, [&](int row, int col, double v) {
                if (row < ma) { if (col < nf) AN[row * S + col] = v; else if (!first && col == nf) stage[row] = v; }
            });
        wave_sync();
        if (!first && lane < ma) aq_row = stage[lane];
    } else
    for (int rb = 0; rb < ma; rb += 8) {
        double acc8[8], mq[8];
        rows8_times_N(A, ma, rb, acc8, mq);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (rb + u < ma) {
                if (c < nf) AN[(rb + u) * S + c] = acc8[u];
                if (lane == rb + u) aq_row = mq[u];
            }
        }
    }
    wave_sync();
    for (int r0 = ma; r0 < m; r0 += 8) {          // identity rows of the task: rows of N
        double t8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int r = r0 + u - ma; t8[u] = Nl[((r < RN) ? r : RN - 1) * S + cc]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (r0 + u < m && c < nf) AN[(r0 + u) * S + c] = t8[u];
    }
    {
        double v = 0.0;
        if (lane < m) {
            v = Q.b[inst * m + lane];
            if (!first) { if (lane < ma) v -= aq_row; else v -= vec[lane - ma]; }
        }
        if ((Q.zero_rows >> lane) & 1ull) v = (lane < m) ? Q.b[inst * m + lane] : 0.0;     // inactive task: A = 0, so b0 = b
        b0[lane] = v;
    }
    if (Q.zero_rows) {             // rows of inactive tasks are zero rows of A N
        wave_sync();
        for (int r = 0; r < m; ++r) if (((Q.zero_rows >> r) & 1ull) && c < nf) AN[r * S + c] = 0.0;
    }
    wave_sync();
    NHQP_PHASE("AN+b0");
    // ---- constraints in z-coordinates (levels below the first): rows [C N; N], bounds shifted by q0
    if (!first) {
        const int nr = Q.nc + (Q.has_box ? n : 0);
        double* Rg = Q.R + inst * (long long)nr * nf;
        const double* Cg = Q.C + inst * (long long)Q.nc * n;
        if (use_tiles) {
            for (int I = 0; 16 * I < Q.nc; ++I) {
                wave_sync();
                rows16_times_N(Cg, Q.nc, I, [&](int row, int col, double v) {
                    if (row < Q.nc) { if (col < nf) Rg[row * nf + col] = v; else if (col == nf) stage[row - 16 * I] = v; }
                });
                wave_sync();
                if (lane < 16 && 16 * I + lane < Q.nc) {       // lane u < 16: the bounds of row 16 I + u, shifted by (C q0)
                    const int r = 16 * I + lane;
                    const double myq = stage[lane];
                    const double lo = Q.lo[inst * Q.nc + r], up = Q.up[inst * Q.nc + r];
                    Q.rlo[inst * nr + r] = (lo <= -1.0e20) ? -1.0e20 : lo - myq;
                    Q.rup[inst * nr + r] = (up >= 1.0e20) ? 1.0e20 : up - myq;
                }
            }
        } else
        for (int rb = 0; rb < Q.nc; rb += 8) {
            double acc8[8], mq[8];
            rows8_times_N(Cg, Q.nc, rb, acc8, mq);
            double myq = 0.0;                  // lane u < 8: (C q0) of row rb + u; the eight rows' bounds are loaded together
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (rb + u < Q.nc && c < nf) Rg[(rb + u) * nf + c] = acc8[u];
                if (lane == u) myq = mq[u];
            }
            if (lane < 8 && rb + lane < Q.nc) {
                const int r = rb + lane;
                const double lo = Q.lo[inst * Q.nc + r], up = Q.up[inst * Q.nc + r];
                Q.rlo[inst * nr + r] = (lo <= -1.0e20) ? -1.0e20 : lo - myq;
                Q.rup[inst * nr + r] = (up >= 1.0e20) ? 1.0e20 : up - myq;
            }
        }
        if (Q.has_box) {
            for (int i0 = 0; i0 < n; i0 += 8) {
                double t8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t8[u] = Nl[(i0 + u) * S + cc];
#pragma unroll
                for (int u = 0; u < 8; ++u) if (i0 + u < n && c < nf) Rg[(Q.nc + i0 + u) * nf + c] = t8[u];
            }
            if (lane < n) {
                const double l = Q.l[inst * n + lane], u = Q.u[inst * n + lane];
                Q.rlo[inst * nr + Q.nc + lane] = (l <= -1.0e20) ? -1.0e20 : l - vec[lane];
                Q.rup[inst * nr + Q.nc + lane] = (u >= 1.0e20) ? 1.0e20 : u - vec[lane];
            }
        }
    }
    wave_sync();      // N is dead from here on: its buffer becomes E (and later V2)
    NHQP_PHASE("constr");
    // ---- Gram matrix of the small side (k <= 32), stride kNS: lane = (row c32 of the result, half h32 of the inner index),
    // two rows of the result per trip
    const bool rowside = m <= nf;
    const int k = rowside ? m : nf;
    constexpr double kSvNoise = 1.0e-7;
    auto build_gram = [&]() {
    if (rowside) {
        const int cm = (c32 < m) ? c32 : 0;
        for (int a = 0; a < m; a += 2) {
            double s0 = 0.0, s1 = 0.0;
            for (int t0 = 0; t0 < nf; t0 += 8) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = t0 + 2 * j + h32;
                    const double own = AN[cm * S + t];
                    s0 = fma(AN[a * S + t], own, s0);
                    s1 = fma(AN[(a + 1) * S + t], own, s1);
                }
            }
            s0 = halfsum<32>(s0); s1 = halfsum<32>(s1);
            if (h32 == 0 && c32 < m) { K[a * kNS + c32] = s0; if (a + 1 < m) K[(a + 1) * kNS + c32] = s1; }
        }
    } else {
        const int cf = (c32 < nf) ? c32 : 0;
        for (int a = 0; a < nf; a += 2) {
            double s0 = 0.0, s1 = 0.0;
            for (int r0 = 0; r0 < m; r0 += 8) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = r0 + 2 * j + h32;
                    const double own = AN[r * S + cf];
                    s0 = fma(AN[r * S + a], own, s0);
                    s1 = fma(AN[r * S + a + 1], own, s1);
                }
            }
            s0 = halfsum<32>(s0); s1 = halfsum<32>(s1);
            if (h32 == 0 && c32 < nf) { K[a * kNS + c32] = s0; if (a + 1 < nf) K[(a + 1) * kNS + c32] = s1; }
        }
    }
    wave_sync();
    };
    build_gram();
    for (int e = lane; e < 32 * kNS; e += 64) E[e] = 0.0;
    wave_sync();
    NHQP_PHASE("gram");
    // (the values-only case first: see the 32-wide kernel)
    bool values_only = false;
    if (rowside && k >= 10 && nf - ns >= k) {
        double lamv; int nlift = 0;
        values_only = sym_eig_selected32(K, E, k, c32, h32, Q.ab_reg ? Q.thr * Q.thr * (1.0 + 1.0e-9) : 0.0, lamv, nlift);
        const double svc = sqrt(lamv > 0.0 ? lamv : 0.0);
        if (values_only) {
            if (c < k) { idx[k - 1 - c] = c; sig[k - 1 - c] = svc; }
            wave_sync();
        } else {
            for (int e = lane; e < 32 * kNS; e += 64) { K[e] = 0.0; E[e] = 0.0; }
            wave_sync();
            build_gram();
        }
    }
    if (!values_only) {
    sym_eig32_fast(K, E, k, c32, h32);
    {
        const double kcc = K[c32 * kNS + c32];
        const double lam = (c < k) ? kcc : -1.0;
        vec[lane] = lam;
        wave_sync();
        int pos = 0;
        for (int d = 0; d < k; ++d) { const double ld = vec[d]; pos += (ld > lam || (ld == lam && d < c)) ? 1 : 0; }
        if (c < k) { idx[pos] = c; sig[pos] = sqrt(lam > 0.0 ? lam : 0.0); }
        wave_sync();
    }
    }
    const double sv_max = sig[0];
    int rho = 0;
    for (int i = 0; i < k; ++i) rho += (sig[i] >= kSvNoise * sv_max && sig[i] > 0.0) ? 1 : 0;
    const int r_next = nf - ns;
    const int nrefl = rowside ? (rho < r_next ? rho : r_next) : 0;
    const bool need_refl = rowside && (ns > 0 || (Q.ab_reg && rho < k));
    double* V1 = K;          // V1[t][i]: component t (< nf <= 64) of v_i (i < 32), then reflector i
    // AN' e (a column of E, rows < m <= 32 on the row side) for my component: chunks of 8 rows
    auto ANt_times_Ecol = [&](int ec) -> double {
        double vv = 0.0;
        for (int q0 = 0; q0 < m; q0 += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) vv = fma(AN[(q0 + j) * S + cc], E[(q0 + j) * kNS + ec], vv);
        }
        return vv;
    };
    if (need_refl) {
        // (K still holds the eigenvalues on its diagonal: they are in sig[] now; clear what V1 uses)
        wave_sync();
        for (int e = lane; e < RN * kNS; e += 64) K[e] = 0.0;
        wave_sync();
        if (values_only) {          // V1 = (A N)': column i = row i of A N
            for (int i0 = 0; i0 < nrefl; i0 += 8) {
                double t8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t8[u] = AN[(i0 + u) * S + cc];
#pragma unroll
                for (int u = 0; u < 8; ++u) if (i0 + u < nrefl && c < nf) V1[c * kNS + i0 + u] = t8[u];
            }
        }
        for (int i = 0; i < (values_only ? 0 : nrefl); ++i) {
            double vv = ANt_times_Ecol(idx[i]);
            const double nrm2 = colsum<64>((c < nf) ? vv * vv : 0.0);
            double nsq, nrs;
            fast_sqrt_rsqrt(nrm2 > 0.0 ? nrm2 : 1.0, nsq, nrs);
            vv = (nrm2 > 0.0) ? vv * nrs : 0.0;
            if (c < nf) V1[c * kNS + i] = vv;
        }
        wave_sync();
        // Householder orthonormalisation; the later columns are updated with lane = (column c32, half h32 of the components), see
        // the 32-wide kernel
        for (int i = 0; i < nrefl; ++i) {
            const double v1ci = V1[cc * kNS + i];
            const double x = (c >= i && c < nf) ? v1ci : 0.0;
            const double nrm2 = colsum<64>(x * x);
            const double xi = bcast(x, i);
            double nsq, nrs;
            fast_sqrt_rsqrt(nrm2 > 0.0 ? nrm2 : 1.0, nsq, nrs);
            if (!(nrm2 > 0.0)) nsq = 0.0;
            const double alpha = (xi > 0.0) ? -nsq : nsq;
            const double hv = (c == i) ? x - alpha : x;
            const double vn2h = fma(fabs(xi), nsq, nrm2);             // |v|^2 / 2
            const double beta = (vn2h > 0.0) ? fast_rcp(vn2h) : 0.0;
            wave_sync();
            if (c < nf) V1[c * kNS + i] = hv;
            vec[lane] = hv;
            wave_sync();
            {
                const bool mine = c32 > i && c32 < nrefl;
                const int j = mine ? c32 : i;
                // (components 0 .. 39 exist for every n > 32: their reads carry no branch; the other 24 sit behind ONE, rows clamped)
                double y[32], hq[32], dot = 0.0;
#pragma unroll
                for (int t = 0; t < 20; ++t) { const int r = 2 * t + h32; y[t] = V1[r * kNS + j]; hq[t] = vec[r]; }
#pragma unroll
                for (int t = 20; t < 32; ++t) { y[t] = 0.0; hq[t] = 0.0; }
                if (hr > 20) {
#pragma unroll
                    for (int t = 20; t < 32; ++t) {
                        const int r = 2 * t + h32;
                        const double v = V1[((r < RN) ? r : RN - 1) * kNS + j];
                        y[t] = (r < RN) ? v : 0.0; hq[t] = vec[r];
                    }
                }
#pragma unroll
                for (int t = 0; t < 32; ++t) dot = fma(hq[t], y[t], dot);
                dot = halfsum<32>(dot) * beta;
                if (mine) {
#pragma unroll
                    for (int t = 0; t < 32; ++t) { const int r = 2 * t + h32; if (t < hr) V1[r * kNS + j] = fma(-dot, hq[t], y[t]); }
                }
            }
            if (c == 0) refl_beta[i] = beta;
            wave_sync();
        }
    }
    NHQP_PHASE("eig+V1refl");
    auto completion_column = [&](int j) -> double {           // (Q e_j)[c], j >= nrefl
        double y = (c == j) ? 1.0 : 0.0;
        for (int i = nrefl - 1; i >= 0; --i) {
            const double v = V1[cc * kNS + i];
            const double hv = (c >= i && c < nf) ? v : 0.0;
            const double dot = colsum<64>(hv * y);
            y -= refl_beta[i] * dot * hv;
        }
        return y;
    };
    if (Q.ab_reg) {
        double bnew = 0.0;
        const bool rebuild = !rowside;
        const int lr = (lane < m) ? lane : 0;
        for (int i = 0; i < k; ++i) {
            const double sv = sig[i];
            const bool lift = sv < Q.thr * sv_max;
            if (!lift && !rebuild) continue;
            const int ec = idx[i];
            double uu = 0.0, vv = 0.0;
            if (rowside) {
                const double e_l = E[(lane & 31) * kNS + ec];
                uu = (lane < m) ? e_l : 0.0;
                if (i >= rho) {
                    const double qc = completion_column(i);
                    vv = (c < nf) ? qc : 0.0;
                } else {
                    vv = ANt_times_Ecol(ec);
                    const double nrm2 = colsum<64>((c < nf) ? vv * vv : 0.0);
                    vv = (nrm2 > 0.0 && c < nf) ? vv / sqrt(nrm2) : 0.0;
                }
            } else {
                const double e_c = E[c32 * kNS + ec];
                vv = (c < nf) ? e_c : 0.0;
                double acc = 0.0;                                     // u_i ~ AN v_i: lane = row, chunks of 8 columns (nf <= 32)
                for (int t0 = 0; t0 < nf; t0 += 8) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc = fma(AN[lr * S + t0 + j], E[(t0 + j) * kNS + ec], acc);
                }
                if (lane >= m) acc = 0.0;
                const double nrm2 = colsum<64>(acc * acc);
                uu = (nrm2 > 0.0) ? acc / sqrt(nrm2) : 0.0;
            }
            const double ub = colsum<64>((lane < m) ? uu * b0[lane] : 0.0);
            double d = 1.0, svn = sv;
            if (lift) { d = sv / (Q.thr * sv_max); svn = (Q.thr * sv_max) * (Q.thr * sv_max) / (sv + Q.thr / 100.0); }
            if (rebuild) bnew = fma(d * ub, uu, bnew);
            else if (lane < m) b0[lane] -= (1.0 - d) * ub * uu;
            if (lift) {
                wave_sync();
                vec[lane] = (lane < m) ? uu : 0.0;
                wave_sync();
                const double dl = svn - sv;
                for (int r0 = 0; r0 < m; r0 += 8) {
                    double t8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) t8[j] = fma(dl * vec[r0 + j], vv, AN[(r0 + j) * S + cc]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (r0 + j < m && c < nf) AN[(r0 + j) * S + c] = t8[j];
                }
                wave_sync();
            }
        }
        if (rebuild && lane < m) b0[lane] = bnew;
        wave_sync();
    }
    NHQP_PHASE("ABreg");
    // ---- null-space basis V2 (nf x ns): row side in NE (stride S), column side in K (stride 33)
    double* V2 = rowside ? NE : K;
    const int v2s = rowside ? S : kNS;
    if (ns > 0) {
        if (!rowside) {
            for (int t0 = 0; t0 < ns; t0 += 8) {
                double t8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t8[u] = E[c32 * kNS + idx[(t0 + u < ns) ? k - ns + t0 + u : 0]];
                wave_sync();
#pragma unroll
                for (int u = 0; u < 8; ++u) if (t0 + u < ns && c < nf) V2[c * v2s + t0 + u] = t8[u];
            }
            wave_sync();
        } else {
            // the completion columns Q e_(r_next + t), 32 at a time: lane = (column c32 of the pass, half h32 of the components),
            // my column in registers; they read V1 = K only and are written over E / N in NE
            for (int p0 = 0; p0 < ns; p0 += 32) {
                double y[32];
#pragma unroll
                for (int s = 0; s < 32; ++s) y[s] = (2 * s + h32 == r_next + p0 + c32) ? 1.0 : 0.0;
                for (int i = nrefl - 1; i >= 0; --i) {
                    double hvv[32], dot = 0.0;
#pragma unroll
                    for (int s = 0; s < 20; ++s) hvv[s] = V1[(2 * s + h32) * kNS + i];
#pragma unroll
                    for (int s = 20; s < 32; ++s) hvv[s] = 0.0;
                    if (hr > 20) {
#pragma unroll
                        for (int s = 20; s < 32; ++s) {
                            const int r = 2 * s + h32;
                            const double v = V1[((r < RN) ? r : RN - 1) * kNS + i];
                            hvv[s] = (r < RN) ? v : 0.0;
                        }
                    }
#pragma unroll
                    for (int s = 0; s < 32; ++s) dot = fma(hvv[s], y[s], dot);
                    dot = halfsum<32>(dot) * refl_beta[i];
#pragma unroll
                    for (int s = 0; s < 32; ++s) y[s] = fma(-dot, hvv[s], y[s]);
                }
                wave_sync();
                if (p0 + c32 < ns) {
#pragma unroll
                    for (int s = 0; s < 32; ++s) if (s < hr && 2 * s + h32 < nf) V2[(2 * s + h32) * v2s + p0 + c32] = y[s];
                }
            }
            wave_sync();
        }
    }
    NHQP_PHASE("V2");
    // ---- H = AN' W AN (+ sv_max V2 V2'), g = -AN' W b0   ->  HBM, nf x nf row-major.  Two rows of H per trip; the inner index in
    // chunks of 8 (w_r (A N)[r][c] is rebuilt per chunk: 16 reads of my own column against 16 broadcast reads)
    {
        const double* w = Q.w ? Q.w + inst * m : nullptr;
        wave_sync();
        vec[lane] = (lane < m) ? (Q.Wd ? Q.Wd[inst * (long long)m * m + lane * (m + 1)] : (w ? w[lane] : 1.0)) : 0.0;   // (dense W: its diagonal here, the rest below)
        wave_sync();
        double* Hg = Q.H + inst * (long long)nf * nf;
        double gacc = 0.0;
        for (int r0 = 0; r0 < m; r0 += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) gacc = fma(-(vec[r0 + j] * AN[(r0 + j) * S + cc]), b0[r0 + j], gacc);
        }
        if (c < nf) Q.g[inst * nf + c] = gacc;
        const bool sel = ns > 0 && Q.sel_reg;
#ifndef OSOT_NHQP_H_VALU
        // (round 5) on the matrix core: nhqp_tile_gram -- was two rows of H per trip through the vector unit, 36-58 k clocks a level
        nhqp_gram_to(Hg, nf, nf, AN, S, m, vec, V2, v2s, nf, nullptr, sel ? ns : 0, sv_max, lane);
#else
        for (int i0 = 0; i0 < nf; i0 += 2) {
            const int i1 = (i0 + 1 < nf) ? i0 + 1 : i0;
            double a0 = 0.0, a1 = 0.0;
            for (int r0 = 0; r0 < m; r0 += 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = r0 + j;
                    const double wan = vec[r] * AN[r * S + cc];
                    a0 = fma(wan, AN[r * S + i0], a0);
                    a1 = fma(wan, AN[r * S + i1], a1);
                }
            }
            if (sel) {
                for (int t0 = 0; t0 < ns; t0 += 8) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int t = t0 + j;
                        const double v2ct = V2[cc * v2s + t];               // (loaded, then masked: no branch between the reads of a chunk)
                        const double own = (t < ns) ? sv_max * v2ct : 0.0;
                        const double f0 = V2[i0 * v2s + t], f1 = V2[i1 * v2s + t];
                        a0 = fma(own, (t < ns) ? f0 : 0.0, a0);
                        a1 = fma(own, (t < ns) ? f1 : 0.0, a1);
                    }
                }
            }
            if (c < nf) { Hg[i0 * nf + c] = a0; if (i1 != i0) Hg[i1 * nf + c] = a1; }
        }
#endif
    }
    NHQP_PHASE("Hg");
    if (Q.Wd) { wave_sync(); nhqp_dense_weight_correction(Q, inst, AN, S, b0, c); }
    if (ns > 0 && Q.V2) {
        double* Vg = Q.V2 + inst * (long long)n * n;
        const int cs = (c < ns) ? c : 0;
        for (int i0 = 0; i0 < nf; i0 += 8) {
            double t8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t8[u] = V2[((i0 + u < nf) ? i0 + u : 0) * v2s + cs];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (i0 + u < nf && c < ns) Vg[(i0 + u) * n + c] = t8[u];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the level preparation for levels whose SMALL side exceeds 32 (min(rows, free variables) > 32, n <= 64): the reference's
// own stack S1 -- (0.1 l_wrist + r_wrist + com + 1e-4 postural) on the 35-coordinate COMAN, one level of 50 rows
// (examples/cpp/coman_ik.cpp:425-431) -- was refused until now.  Lane = column / row, every loop
// generic in n and m; what it does differently from the two kernels above is the decomposition --
//   * always the COLUMN-side Gram matrix G = (A N)'(A N) (nf x nf): its eigenvectors are ALL of V, the null space of A N included
//     (no Householder completion), whatever the rank;
//   * Householder tridiagonalisation + implicit QL over all 64 lanes (sym_eig_wide) -- until late in round 5 a parallel cyclic Jacobi
//     iteration (kept under OSOT_NHQP_WIDE_JACOBI): ~280 rounds of 2 x 18 dependent LDS hand-offs, 93 % of a 7.4 ms launch at S1;
//     the Gram matrix, H and the triplets' u = A N v on the fp64 matrix core (nhqp_tile_gram; T = A N V in one go);
//   * U = A N V Sigma^-1 explicitly for the min(m, nf) triplets (null triplets: completed by Gram-Schmidt of unit vectors), so that
//     regularize_A_b is the reference's own formula, b0 <- U diag(d) U'b0, A N <- A N + sum (sv' - sv) u v' (nHQP.cpp:236-279).
// LDS (dynamic, nhqp_prepare_wide_lds_bytes): A N [RM][S], N -> V [RN][S], G -> U [RM][S], S = RN + 1, RN = n rounded up to 8,
// RM = max(m, n) rounded up to 8: 50 KB at S1.
// Symmetric eigenproblem of the wide path, G (k x k, k <= 64, LDS stride S, lane = column / row) -> G[c][c] = eigenvalue c, V[:, c] =
// its eigenvector (V: LDS, stride S; both buffers ZERO beyond k up to the next multiple of eight, rows and columns).  Householder tridiagonalisation
// (one reflector per column, lane = column for the matrix - vector product and the rank-two update, lane = row for V <- V H) and the
// implicit QL iteration of sym_ql_32 over all 64 lanes: (d, e) one entry per lane read with v_readlane, a lane's own row of V carried
// through the rotations.  Replaces the cyclic Jacobi iteration (kept under OSOT_NHQP_WIDE_JACOBI for A/B), which moved ~12 LDS words
// per pair per round, ~280 rounds, and was 93 % of the launch.  vv / ww: 64 doubles of LDS scratch each.
__device__ __forceinline__ bool sym_eig_wide(double* G, double* V, double* vv, double* ww, int k_in, int S_in, int lane) {
    constexpr double kEps = 2.220446049250313e-16;
    const int k = uniform_i(k_in), S = uniform_i(S_in);
    const int cl = (lane < k) ? lane : 0;
    // ---- V = I
    for (int i = 0; i < k; ++i) if (lane < k) V[i * S + lane] = (i == lane) ? 1.0 : 0.0;
    double d = 0.0, e = 0.0;                       // lane c: T[c][c], T[c][c + 1]
    wave_sync();
    for (int j = 0; j + 2 < k; ++j) {
        const bool in = lane > j && lane < k;
        const double x = in ? G[j * S + cl] : 0.0;                   // column j below the diagonal (= row j: G is symmetric)
        const double tail = uniform_d(colsum<64>((lane > j + 1) ? x * x : 0.0));
        const double x1 = bcast(x, j + 1);
        if (!(tail > 0.0)) { if (lane == j) e = x1; continue; }      // already tridiagonal in this column
        // scaled against over- / underflow of the squares: the power of two of the column's largest entry
        const int ex = frexp_exponent(uniform_d(colmax<64>(fabs(x))));
        const double xs = scale_pow2(x, -ex), x1s = scale_pow2(x1, -ex);
        const double sigma = uniform_d(colsum<64>(xs * xs));
        double rt, irt;
        fast_sqrt_rsqrt(sigma, rt, irt);
        const double alpha = (x1s >= 0.0) ? -rt : rt;
        const double v = xs - ((lane == j + 1) ? alpha : 0.0);       // v = x - alpha e_(j+1), |v|^2 = 2 (sigma - alpha x1)
        const double beta = 1.0 / (sigma - alpha * x1s);             // H = I - beta v v'
        if (lane == j) e = scale_pow2(alpha, ex);
        vv[lane] = v;
        wave_sync();
        // p = beta G v (lane = column; rows j + 1 .. k - 1), w = p - (beta / 2)(v'p) v.  Every loop below runs over whole chunks of eight
        // rows / columns from (j + 1) rounded down, eight LDS reads in flight (a loop with a run-time trip count pays one LDS round trip
        // per element otherwise): v and w are ZERO outside (j, k), the rows and columns of G and V beyond k are zero, and the stale
        // rows <= j of G are finite -- so the extra terms are exact zeros.
        const int r0 = (j + 1) & ~7, r1 = (k + 7) & ~7;
        double p0 = 0.0, p1 = 0.0, t0 = 0.0, t1 = 0.0;
        for (int i = r0; i < r1; i += 8) {
            double gg[8], v8[8], tt[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { gg[u] = G[(i + u) * S + cl]; v8[u] = vv[i + u]; tt[u] = V[cl * S + i + u]; }
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                p0 = fma(gg[u], v8[u], p0); p1 = fma(gg[u + 1], v8[u + 1], p1);
                t0 = fma(tt[u], v8[u], t0); t1 = fma(tt[u + 1], v8[u + 1], t1);          // my row of V times v (lane = row)
            }
        }
        const double p = in ? beta * (p0 + p1) : 0.0;
        const double kap = 0.5 * beta * uniform_d(colsum<64>(v * p));
        const double w = fma(-kap, v, p);
        ww[lane] = w;
        const double bt = beta * (t0 + t1);
        wave_sync();
        // G <- G - v w' - w v' on the trailing block (lane = column), V <- V - (beta V v) v' (lane = row)
        if (lane < k) for (int r = r0; r < r1; r += 8) {
            double gg[8], v8[8], w8[8], tt[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { gg[u] = G[(r + u) * S + lane]; v8[u] = vv[r + u]; w8[u] = ww[r + u]; tt[u] = V[lane * S + r + u]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                G[(r + u) * S + lane] = fma(-v8[u], w, fma(-w8[u], v, gg[u]));
                V[lane * S + r + u] = fma(-bt, v8[u], tt[u]);
            }
        }
        wave_sync();
    }
    if (lane < k) d = G[lane * S + lane];
    if (k >= 2) { const double el = G[(k - 2) * S + (k - 1)]; if (lane == k - 2) e = el; }
    wave_sync();
#ifdef OSOT_NHQP_PHASES
    if (blockIdx.x == 0 && lane == 0) printf("PHASE w:tridiagonal done at %lld\n", (long long)clock64());
#endif
    // ---- implicit QL with shifts (tql2; see sym_ql_32): e[c] couples c and c + 1, e[k - 1] = 0
    const double anorm = uniform_d(colmax<64>((lane < k) ? fabs(d) + fabs(e) : 0.0));
    const double etol = OSOT_QL_TOL * kEps * anorm;
    bool converged = true;      // (uniform) every eigenvalue's QL iteration ended on a negligible coupling within its 60 sweeps
    for (int l = 0; l < k; ++l) {
        bool done = false;
        for (int iter = 0; iter < 60; ++iter) {
            const bool small = (lane >= l && lane < k - 1) && (fabs(e) <= etol);
            const unsigned long long mk = wave_ballot(small);
            const int m = mk ? __builtin_ctzll(mk) : k - 1;
            if (m == l) { done = true; break; }
            const double dl = bcast(d, l), dl1 = bcast(d, l + 1), el = bcast(e, l), dm = bcast(d, m);
            double g = (dl1 - dl) / (2.0 * el);
            double r = sqrt(fma(g, g, 1.0));
            g = dm - dl + el / (g + (g >= 0.0 ? r : -r));
            double sn = 1.0, cs = 1.0, p = 0.0;
            bool underflow = false;
            double di1 = dm;
            double zc = V[cl * S + m];
            double z0n = V[cl * S + m - 1];
            for (int i = m - 1; i >= l; --i) {
                const double z0 = z0n;
                if (i > l) z0n = V[cl * S + i - 1];
                const double ei = bcast(e, i), di = bcast(d, i);
                const double f = sn * ei, b = cs * ei;
                const double rr2 = fma(f, f, g * g);
                double ir;
                if (rr2 > 0.0) fast_sqrt_rsqrt(rr2, r, ir);
                else { r = 0.0; ir = 0.0; }
                if (lane == i + 1) e = r;
                if (r == 0.0) {
                    if (lane == i + 1) d -= p;
                    if (lane == m) e = 0.0;
                    if (lane < k) V[lane * S + i + 1] = zc;
                    underflow = true;
                    break;
                }
                sn = f * ir; cs = g * ir;
                g = di1 - p;
                r = fma(di - g, sn, 2.0 * cs * b);
                p = sn * r;
                if (lane == i + 1) d = g + p;
                g = fma(cs, r, -b);
                di1 = di;
                if (lane < k) V[lane * S + i + 1] = fma(sn, z0, cs * zc);
                zc = fma(cs, z0, -sn * zc);
            }
            if (underflow) continue;
            if (lane < k) V[lane * S + l] = zc;
            if (lane == l) { d -= p; e = g; }
            if (lane == m) e = 0.0;
        }
        // the sweep budget ran out (or the last sweep ended on an underflow restart): the eigenpairs from here on are not to be
        // trusted -- said to the caller, which fails the instance instead of building a null space from them
        if (!done) {
            const bool small = (lane >= l && lane < k - 1) && (fabs(e) <= etol);
            const unsigned long long mk = wave_ballot(small);
            if ((mk ? __builtin_ctzll(mk) : k - 1) != l) converged = false;
        }
    }
    wave_sync();
    if (lane < k) G[lane * S + lane] = d;
    wave_sync();
    // (a non-finite eigenvalue -- NaN in G -- never passes `fabs(e) <= etol`: covered by the same flag)
    return converged;
}

inline size_t nhqp_prepare_wide_lds_bytes(int m, int n) {
    const size_t RN = (size_t)nhqp64_rows(n), S = RN + 1, RM = (size_t)nhqp64_rows(m > n ? m : n);
    return sizeof(double) * (2 * RM * S + RN * S + 5 * 64 + 2 * 32) + sizeof(int) * (64 + 2 * 32);
}

__global__ void __launch_bounds__(64) osot_nhqp_prepare_wide_kernel(const DevNhqp Q) {
    OSOT_DYNAMIC_LDS(nw_smem);
    const int n = Q.n, m = Q.m, ma = Q.ma, nf = Q.nf, ns = Q.ns;
    const int RN = nhqp64_rows(n), S = RN + 1, RM = nhqp64_rows(m > n ? m : n);
    double* AN = reinterpret_cast<double*>(nw_smem);   // [RM][S]  A N (m x nf)
    double* NV = AN + RM * S;                          // [RN][S]  N (n x nf), then V (nf x nf)
    double* G = NV + RN * S;                           // [RM][S]  Gram matrix (nf x nf), then U (m x ksv)
    double* b0 = G + RM * S;                           // [64]
    double* vec = b0 + 64;                             // [64]
    double* sig = vec + 64;                            // [64] singular values, descending (zero beyond min(m, nf))
    double* ub = sig + 64;                             // [64] u_i'b0
    double* dl = ub + 64;                              // [64] sv'_i - sv_i of the lifted triplets (0: not lifted)
    double* rc = dl + 64;                              // [32] rotation cosines of a round
    double* rs = rc + 32;                              // [32] ... sines
    int* idx = reinterpret_cast<int*>(rs + 32);        // [64] idx[pos] = eigen-column of the pos-th largest
    int* pp = idx + 64;                                // [32] the pairs of a round
    int* pq = pp + 32;
    const long long inst = blockIdx.x;
    const int lane = threadIdx.x;
    if (inst >= Q.B) return;
    if (Q.status && Q.status[inst] != 0) return;
    const bool first = Q.level == 0;
#ifdef OSOT_NHQP_PHASES
    long long ph_t_ = (long long)clock64();
#endif
    const double* A = Q.A ? Q.A + inst * (long long)ma * n : nullptr;
    for (int e = lane; e < RM * S; e += 64) { AN[e] = 0.0; G[e] = 0.0; }
    for (int e = lane; e < RN * S; e += 64) NV[e] = 0.0;
    wave_sync();
    // ---- N, q0
    if (first) { if (lane < n) NV[lane * S + lane] = 1.0; }
    else {
        const double* Ng = Q.N + inst * (long long)n * n;
        if (lane < nf) for (int i = 0; i < n; ++i) NV[i * S + lane] = Ng[i * n + lane];
    }
    vec[lane] = (!first && lane < n) ? Q.q0[inst * n + lane] : 0.0;
    wave_sync();
    NHQP_PHASE("w:loadN");
    // ---- A N (lane = column), b0 = b - A q0 (lane = row)
    for (int r = 0; r < m; ++r) {
        const bool zr = (Q.zero_rows >> r) & 1ull;
        double acc = 0.0;
        if (lane < nf && !zr) {
            if (r < ma) {
                if (first) acc = A[r * n + lane];                      // N = I: A N = A
                else for (int i = 0; i < n; ++i) acc = fma(A[r * n + i], NV[i * S + lane], acc);
            }
            else acc = NV[(r - ma) * S + lane];
        }
        if (lane < nf) AN[r * S + lane] = acc;
    }
    {
        double v = 0.0;
        if (lane < m) {
            v = Q.b[inst * m + lane];
            if (!first && !((Q.zero_rows >> lane) & 1ull)) {
                if (lane < ma) { double aq = 0.0; for (int i = 0; i < n; ++i) aq = fma(A[lane * n + i], vec[i], aq); v -= aq; }
                else v -= vec[lane - ma];
            }
        }
        b0[lane] = v;
    }
    wave_sync();
    NHQP_PHASE("w:AN+b0");
    // ---- constraints in z-coordinates (levels below the first): rows [C N; N], bounds shifted by q0 (compute_contraints, :282-317)
    if (!first) {
        const int nr = Q.nc + (Q.has_box ? n : 0);
        double* Rg = Q.R + inst * (long long)nr * nf;
        const double* Cg = Q.C + inst * (long long)Q.nc * n;
        for (int r = 0; r < Q.nc; ++r) {
            double acc = 0.0;
            if (lane < nf) for (int i = 0; i < n; ++i) acc = fma(Cg[r * n + i], NV[i * S + lane], acc);
            if (lane < nf) Rg[r * nf + lane] = acc;
        }
        if (lane < Q.nc) {
            double cq = 0.0;
            for (int i = 0; i < n; ++i) cq = fma(Cg[lane * n + i], vec[i], cq);
            const double lo = Q.lo[inst * Q.nc + lane], up = Q.up[inst * Q.nc + lane];
            Q.rlo[inst * nr + lane] = (lo <= -1.0e20) ? -1.0e20 : lo - cq;
            Q.rup[inst * nr + lane] = (up >= 1.0e20) ? 1.0e20 : up - cq;
        }
        if (Q.has_box) {
            if (lane < nf) for (int i = 0; i < n; ++i) Rg[(Q.nc + i) * nf + lane] = NV[i * S + lane];
            if (lane < n) {
                const double l = Q.l[inst * n + lane], u = Q.u[inst * n + lane];
                Q.rlo[inst * nr + Q.nc + lane] = (l <= -1.0e20) ? -1.0e20 : l - vec[lane];
                Q.rup[inst * nr + Q.nc + lane] = (u >= 1.0e20) ? 1.0e20 : u - vec[lane];
            }
        }
    }
    wave_sync();      // N is dead: NV becomes V
    NHQP_PHASE("w:constr");
    // ---- G = (A N)'(A N) on the matrix core (nhqp_tile_gram: was eight rows of G per pass over A N through the vector unit, 53 k clocks
    // of a 670 k-clock level at S1), V = I
    nhqp_gram_to(G, S, nf, AN, S, m, nullptr, nullptr, 0, 0, nullptr, 0, 0.0, lane);
    for (int e = lane; e < RN * S; e += 64) NV[e] = 0.0;
    wave_sync();
    if (lane < nf) NV[lane * S + lane] = 1.0;
    wave_sync();
    NHQP_PHASE("w:gram");
#ifndef OSOT_NHQP_WIDE_JACOBI
    // ---- eigen-decomposition of G: tridiagonalisation + implicit QL (sym_eig_wide); eigenvalues to the diagonal, V in NV
    const bool eig_ok = sym_eig_wide(G, NV, vec, ub, nf, S, lane);
#else
    const bool eig_ok = true;
    // ---- cyclic Jacobi on G, rotations accumulated in V.  kk = nf rounded up to even (a phantom index pairs with nobody); round r of
    // a sweep: (kk - 1, r) and ((r + t) mod (kk - 1), (r - t) mod (kk - 1)), t = 1 .. kk / 2 - 1 -- every pair once per sweep.
    {
        const int kk = (nf + 1) & ~1, half = kk >> 1, md = kk - 1;
        double tr = 0.0;
        if (lane < nf) tr = G[lane * S + lane];
        const double scale = uniform_d(colsum<64>(fabs(tr)));            // trace: the eigenvalues' scale
        for (int sweep = 0; sweep < 14; ++sweep) {
            double off = 0.0;
            if (lane < nf) for (int a = 0; a < nf; ++a) if (a != lane) { const double v = G[a * S + lane]; off = fma(v, v, off); }
            off = uniform_d(colsum<64>(off));
            if (!(off > 1.0e-30 * scale * scale)) break;
            for (int r = 0; r < md; ++r) {
                if (lane < half) {
                    int p, q;
                    if (lane == 0) { p = md; q = r; }
                    else { p = (r + lane) % md; q = (r - lane + md) % md; }
                    if (p > q) { const int t = p; p = q; q = t; }
                    double cs = 1.0, sn = 0.0;
                    if (q < nf) {
                        const double gpq = G[p * S + q], gpp = G[p * S + p], gqq = G[q * S + q];
                        if (fabs(gpq) > 1.0e-300 && fabs(gpq) > 1.0e-17 * sqrt(fabs(gpp * gqq))) {
                            const double tau = (gqq - gpp) / (2.0 * gpq);
                            const double t = ((tau >= 0.0) ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                            cs = 1.0 / sqrt(1.0 + t * t); sn = t * cs;
                        }
                    }
                    pp[lane] = p; pq[lane] = (q < nf) ? q : p; rc[lane] = cs; rs[lane] = sn;     // (phantom partner: identity on p alone)
                }
                wave_sync();
                // rows p, q of G <- J'G (lane = column): (g_p, g_q) <- (c g_p - s g_q, s g_p + c g_q)
                if (lane < nf) {
                    for (int t = 0; t < half; ++t) {
                        const int p = pp[t], q = pq[t];
                        if (p == q) continue;
                        const double cs = rc[t], sn = rs[t];
                        const double gp = G[p * S + lane], gq = G[q * S + lane];
                        G[p * S + lane] = cs * gp - sn * gq;
                        G[q * S + lane] = sn * gp + cs * gq;
                    }
                }
                wave_sync();
                // columns p, q of G <- G J and of V <- V J (lane = row)
                if (lane < nf) {
                    for (int t = 0; t < half; ++t) {
                        const int p = pp[t], q = pq[t];
                        if (p == q) continue;
                        const double cs = rc[t], sn = rs[t];
                        const double gp = G[lane * S + p], gq = G[lane * S + q];
                        G[lane * S + p] = cs * gp - sn * gq;
                        G[lane * S + q] = sn * gp + cs * gq;
                        const double vp = NV[lane * S + p], vq = NV[lane * S + q];
                        NV[lane * S + p] = cs * vp - sn * vq;
                        NV[lane * S + q] = sn * vp + cs * vq;
                    }
                }
                wave_sync();
            }
        }
    }
#endif
    // ---- singular values, descending; ksv = min(m, nf) of them exist (svd.singularValues(), Eigen's thin count)
    NHQP_PHASE("w:jacobi");
    const int ksv = (m < nf) ? m : nf;
    {
        const double lam = (lane < nf) ? G[lane * S + lane] : -1.0;
        wave_sync();
        vec[lane] = lam;
        wave_sync();
        int pos = 0;
        for (int d = 0; d < nf; ++d) { const double ld = vec[d]; pos += (ld > lam || (ld == lam && d < lane)) ? 1 : 0; }
        sig[lane] = 0.0;
        wave_sync();
        if (lane < nf) { idx[pos] = lane; if (pos < ksv) sig[pos] = sqrt(lam > 0.0 ? lam : 0.0); }
        wave_sync();
    }
    const double sv_max = sig[0];
    constexpr double kSvNoise = 1.0e-7;
    NHQP_PHASE("w:sort");
    // ---- regularize_A_b (nHQP.cpp:236-279)
    if (Q.ab_reg) {
        double* U = G;                                 // (the Gram matrix is spent: its diagonal went into sig[])
        wave_sync();
        for (int e = lane; e < RM * S; e += 64) U[e] = 0.0;
        wave_sync();
        // U[:, i] = A N v_i / |A N v_i| where the triplet is genuine; a null triplet's u_i completes the basis: the first unit
        // vector with a usable component outside the span of the u's found so far (modified Gram-Schmidt, lane = row)
#ifndef OSOT_NHQP_WIDE_U_VALU
        // (round 5) all genuine triplets at once: T = A N V on the fp64 matrix core with the columns of V gathered in singular-value
        // order (column j of T belongs to idx[j]), then lane = triplet for the norms, u_j'b0, the scaling and the lifting rule; the
        // null triplets (none on a full-rank level) are completed one by one afterwards as before.  Was one triplet after the other:
        // a matrix - vector product, two wave reductions, a square root and a division each, 3.1 k clocks x 35 at S1.
        {
            const int q4 = lane >> 4, a16 = lane & 15;
            for (int J = 0; 16 * J < ksv; ++J) {
                const int pos = 16 * J + a16;
                const int ec = idx[(pos < nf) ? pos : 0];
                v4f64 t[4];
#pragma unroll
                for (int I = 0; I < 4; ++I) { t[I][0] = 0.0; t[I][1] = 0.0; t[I][2] = 0.0; t[I][3] = 0.0; }
                for (int k0 = 0; k0 < nf; k0 += 4) {
                    const double xb = NV[(k0 + q4) * S + ec];
                    double xa[4];
#pragma unroll
                    for (int I = 0; I < 4; ++I) { const int row = 16 * I + a16; xa[I] = AN[((row < RM) ? row : 0) * S + k0 + q4]; }
#pragma unroll
                    for (int I = 0; I < 4; ++I) t[I] = mfma_f64_16x16x4(xa[I], xb, t[I]);
                }
#pragma unroll
                for (int I = 0; I < 4; ++I)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * I + q4 + 4 * r;
                        if (row < m && pos < ksv) U[row * S + pos] = t[I][r];
                    }
            }
        }
        wave_sync();
        unsigned long long done;
        {
            const int j = (lane < ksv) ? lane : 0;
            double n0 = 0.0, n1 = 0.0, d0 = 0.0, d1 = 0.0;
            for (int r = 0; r < m; r += 8) {            // (rows of U beyond m are zero, b0 too)
                double u8[8], b8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { u8[q] = U[(r + q) * S + j]; b8[q] = b0[(r + q) & 63]; }
#pragma unroll
                for (int q = 0; q < 8; q += 2) {
                    n0 = fma(u8[q], u8[q], n0); n1 = fma(u8[q + 1], u8[q + 1], n1);
                    d0 = fma(u8[q], b8[q], d0); d1 = fma(u8[q + 1], b8[q + 1], d1);
                }
            }
            const double nrm2 = n0 + n1, sv = sig[j];
            const bool genuine = lane < ksv && sv >= kSvNoise * sv_max && sv > 0.0 && nrm2 > 0.0;
            const double scale = genuine ? 1.0 / sqrt(nrm2) : 0.0;
            const double dotb = (d0 + d1) * scale;
            for (int r = 0; r < m; r += 8) {            // u_j = T[:, j] / |T[:, j]|  (a null triplet's column: zeros until it is completed below)
                double u8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) u8[q] = U[(r + q) * S + j];
#pragma unroll
                for (int q = 0; q < 8; ++q) if (lane < ksv && r + q < m) U[(r + q) * S + j] = scale * u8[q];
            }
            const bool lift = sv < Q.thr * sv_max;
            double d = 1.0, svn = sv;
            if (lift) { d = sv / (Q.thr * sv_max); svn = (Q.thr * sv_max) * (Q.thr * sv_max) / (sv + Q.thr / 100.0); }
            if (genuine) { ub[lane] = d * dotb; dl[lane] = lift ? svn - sv : 0.0; }
            done = wave_ballot(genuine);
        }
        wave_sync();
        int next_unit = 0;
        for (int i = 0; i < ksv; ++i) {
            if ((done >> i) & 1ull) continue;
            // a null triplet's u_i completes the basis: the first unit vector with a usable component outside the span of the u's so far
            double u = 0.0;
            bool have = false;
            while (!have && next_unit < m) {
                u = (lane == next_unit) ? 1.0 : 0.0;
                next_unit++;
                for (int pass = 0; pass < 2; ++pass)
                    for (int j = 0; j < ksv; ++j) {
                        if (!((done >> j) & 1ull)) continue;
                        const double uj = (lane < m) ? U[lane * S + j] : 0.0;
                        const double dot = uniform_d(colsum<64>(uj * u));
                        u = fma(-dot, uj, u);
                    }
                const double nrm2 = uniform_d(colsum<64>(u * u));
                if (nrm2 > 0.25) { u = u / sqrt(nrm2); have = true; }
            }
            if (!have) u = 0.0;
            wave_sync();
            if (lane < m) U[lane * S + i] = u;
            const double dotb = uniform_d(colsum<64>((lane < m) ? u * b0[lane] : 0.0));
            const double sv = sig[i];
            const bool lift = sv < Q.thr * sv_max;
            double d = 1.0, svn = sv;
            if (lift) { d = sv / (Q.thr * sv_max); svn = (Q.thr * sv_max) * (Q.thr * sv_max) / (sv + Q.thr / 100.0); }
            if (lane == 0) { ub[i] = d * dotb; dl[i] = lift ? svn - sv : 0.0; }
            done |= 1ull << i;
            wave_sync();
        }
#else
        int next_unit = 0;
        for (int i = 0; i < ksv; ++i) {
            const int ec = idx[i];
            double u = 0.0;
            bool have = false;
            if (sig[i] >= kSvNoise * sv_max && sig[i] > 0.0) {
                {   // (columns of A N and rows of V beyond nf are zero up to the next multiple of eight: whole chunks, eight reads in flight)
                    const int rl = (lane < m) ? lane : 0;
                    double u0 = 0.0, u1 = 0.0;
                    for (int t = 0; t < nf; t += 8) {
                        double a8[8], v8[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) { a8[q] = AN[rl * S + t + q]; v8[q] = NV[(t + q) * S + ec]; }
#pragma unroll
                        for (int q = 0; q < 8; q += 2) { u0 = fma(a8[q], v8[q], u0); u1 = fma(a8[q + 1], v8[q + 1], u1); }
                    }
                    u = (lane < m) ? u0 + u1 : 0.0;
                }
                const double nrm2 = uniform_d(colsum<64>(u * u));
                if (nrm2 > 0.0) { u = u / sqrt(nrm2); have = true; }
            }
            while (!have && next_unit < m) {
                u = (lane == next_unit) ? 1.0 : 0.0;
                next_unit++;
                for (int pass = 0; pass < 2; ++pass)
                    for (int j = 0; j < i; ++j) {
                        const double uj = (lane < m) ? U[lane * S + j] : 0.0;
                        const double dot = uniform_d(colsum<64>(uj * u));
                        u = fma(-dot, uj, u);
                    }
                const double nrm2 = uniform_d(colsum<64>(u * u));
                if (nrm2 > 0.25) { u = u / sqrt(nrm2); have = true; }
            }
            if (!have) u = 0.0;
            wave_sync();
            if (lane < m) U[lane * S + i] = u;
            const double dotb = uniform_d(colsum<64>((lane < m) ? u * b0[lane] : 0.0));
            const double sv = sig[i];
            const bool lift = sv < Q.thr * sv_max;
            double d = 1.0, svn = sv;
            if (lift) { d = sv / (Q.thr * sv_max); svn = (Q.thr * sv_max) * (Q.thr * sv_max) / (sv + Q.thr / 100.0); }
            if (lane == 0) { ub[i] = d * dotb; dl[i] = lift ? svn - sv : 0.0; }
            wave_sync();
        }
#endif
        // b0 <- sum_i d_i (u_i'b0) u_i  (U diag(d) U'b0 with b0_rot(i) = 0 beyond the ksv singular values)
        if (lane < m) {
            double v = 0.0;
            for (int i = 0; i < ksv; ++i) v = fma(ub[i], U[lane * S + i], v);
            b0[lane] = v;
        }
        // A N <- A N + sum over the lifted triplets (sv' - sv) u v'   (lane = column)
        for (int i = 0; i < ksv; ++i) {
            const double del = dl[i];
            if (del == 0.0) continue;
            const int ec = idx[i];
            if (lane < nf) {
                const double vv = NV[lane * S + ec];
                for (int r = 0; r < m; ++r) AN[r * S + lane] = fma(del * U[r * S + i], vv, AN[r * S + lane]);
            }
        }
        wave_sync();
    }
    // ---- H = AN' W AN (+ sv_max V2 V2'), g = -AN' W b0, V2 = the columns of V of the ns smallest eigenvalues (svd.matrixV().rightCols)
    NHQP_PHASE("w:ABreg");
    {
        const double* w = Q.w ? Q.w + inst * m : nullptr;
        wave_sync();
        vec[lane] = (lane < m) ? (Q.Wd ? Q.Wd[inst * (long long)m * m + lane * (m + 1)] : (w ? w[lane] : 1.0)) : 0.0;   // (dense W: its diagonal here, the rest below)
        wave_sync();
        double* Hg = Q.H + inst * (long long)nf * nf;
        const bool sel = ns > 0 && Q.sel_reg;
        {
            const int cl = (lane < nf) ? lane : 0;
            double gacc = 0.0;
            for (int r = 0; r < m; r += 8) {             // (rows of A N and entries of vec beyond m are zero: whole chunks of eight)
                double a8[8], w8[8], b8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { a8[u] = AN[(r + u) * S + cl]; w8[u] = vec[r + u]; b8[u] = b0[(r + u) & 63]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) gacc = fma(-(w8[u] * a8[u]), b8[u], gacc);
            }
            if (lane < nf) Q.g[inst * nf + lane] = gacc;
            // H on the matrix core: rows of A N weighted by w, plus sv_max V2 V2' as extra k-steps (its columns scaled by sqrt(sv_max))
            const int nsel = sel ? ns : 0;
            nhqp_gram_to(Hg, nf, nf, AN, S, m, vec, NV, S, RN, idx + (nf - ns), nsel, sv_max, lane);
            // an eigen-decomposition that did not converge (sym_eig_wide's flag): the level's QP is handed a NaN pivot, ends as NOT_PD, and
            // the accumulation marks the instance failed (dq = 0, like every other failure) -- sigma, U, V2 of an unconverged iteration
            // never reach the lower levels silently.  (Same wavefront, same address, program order: the store above lands first.)
            if (!eig_ok && lane == 0) Hg[0] = __builtin_nan("");
        }
        if (Q.Wd) { wave_sync(); nhqp_dense_weight_correction(Q, inst, AN, S, b0, lane); }
        if (ns > 0 && Q.V2) {
            double* Vg = Q.V2 + inst * (long long)n * n;
            if (lane < ns) { const int ec = idx[nf - ns + lane]; for (int i = 0; i < nf; ++i) Vg[i * n + lane] = NV[i * S + ec]; }
        }
    }
}

struct DevNhqpAcc {
    int B, n, nf, ns, first, last;
    const double* z;       // [B][nf] the level's QP solution
    const int* qp_status;  // [B] status of the level's QP
    double* q0;            // [B][n] in/out
    const double* N;       // [B][n][n] (n x nf); identity at the first level
    const double* V2;      // [B][n][n] (nf x ns)
    double* Nnext;         // [B][n][n] (n x ns)
    int* status;           // [B] in/out: OSOT_STATUS_* of the instance (first failure sticks)
    double* dq;            // [B][n] written at the last level
};

// Nn (n x ns, row stride n) = Ng (n x nf) Vg (nf x ns), all three in HBM with row stride n: 16 x 16 tiles, MB row blocks held at once,
// one column block after the other; a lane's operands are Ng[16 I + (lane & 15)][k0 + (lane >> 4)] and Vg[k0 + (lane >> 4)][16 J + (lane & 15)]
// (addresses clamped into the instance's block; the inner index masked beyond nf; rows / columns beyond n / ns are not stored).
template <int MB>
__device__ __forceinline__ void acc_tile_product(const double* Ng, const double* Vg, double* Nn, int n, int nf, int ns, int lane) {
    const int q = lane >> 4, a = lane & 15;
    for (int J = 0; 16 * J < ns; ++J) {
        v4f64 t[MB];
#pragma unroll
        for (int I = 0; I < MB; ++I) { t[I][0] = 0.0; t[I][1] = 0.0; t[I][2] = 0.0; t[I][3] = 0.0; }
        const int col = 16 * J + a, colc = (col < ns) ? col : ns - 1;
        for (int k0 = 0; k0 < nf; k0 += 8) {            // two k-steps per trip: 2 (MB + 1) loads in flight
            double xa[2][MB], xb[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = k0 + 4 * u + q;
                const bool live = k < nf;
                const int kc = live ? k : 0;
                const double vb = Vg[kc * n + colc];
                xb[u] = live ? vb : 0.0;
#pragma unroll
                for (int I = 0; I < MB; ++I) {
                    const int row = 16 * I + a;
                    const double va = Ng[((row < n) ? row : n - 1) * n + kc];
                    xa[u][I] = live ? va : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int I = 0; I < MB; ++I) t[I] = mfma_f64_16x16x4(xa[u][I], xb[u], t[I]);
        }
#pragma unroll
        for (int I = 0; I < MB; ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * I + q + 4 * r;
                if (i < n && col < ns) Nn[i * n + col] = t[I][r];
            }
    }
}

// solution += N z;  N <- N V2  (nHQP.cpp:182-196).  n <= 32: lane = c + 32 h (the halves split the sum); 32 < n <= 64: lane = row
__global__ void __launch_bounds__(64) osot_nhqp_accumulate_kernel(const DevNhqpAcc Q) {
    const long long inst = blockIdx.x;
    const int lane = threadIdx.x;
    if (inst >= Q.B) return;
    const int n = Q.n, nf = Q.nf, ns = Q.ns;
    const bool wide = n > 32;
    const int c = wide ? lane : (lane & 31), h = wide ? 0 : (lane >> 5), hv = wide ? 1 : 2;
    int st = Q.first ? 0 : Q.status[inst];
    if (st == 0 && Q.qp_status[inst] != 0) st = Q.qp_status[inst];
    if (lane == 0) Q.status[inst] = st;
    if (st != 0) {
        if (Q.last && h == 0 && c < n) Q.dq[inst * n + c] = 0.0;      // failed instances return 0 (coman_ik.cpp:189-190)
        return;
    }
    const double* Ng = Q.N + inst * (long long)n * n;
    const double* z = Q.z + inst * nf;
    double acc = 0.0;
    if (c < n) {
        if (Q.first) acc = (h == 0) ? z[c] : 0.0;
        else {
            // (eight products per trip with their sixteen loads in flight together, clamped addresses and masked terms: element by
            //  element every multiply-add waited for its own HBM / L2 round trip -- 43 us of the level-1 accumulation at config 3)
            for (int j0 = h; j0 < nf; j0 += 8 * hv) {
                double a[8], b[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int j = j0 + hv * u, jc = (j < nf) ? j : nf - 1; a[u] = Ng[c * n + jc]; b[u] = z[jc]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = fma((j0 + hv * u < nf) ? a[u] : 0.0, b[u], acc);
            }
        }
    }
    if (!wide) acc = halfsum<32>(acc);
    const double qn = (Q.first ? 0.0 : ((c < n) ? Q.q0[inst * n + c] : 0.0)) + acc;
    if (h == 0 && c < n) { Q.q0[inst * n + c] = qn; if (Q.last) Q.dq[inst * n + c] = qn; }
    if (!Q.last && ns > 0) {
        const double* Vg = Q.V2 + inst * (long long)n * n;
        double* Nn = Q.Nnext + inst * (long long)n * n;
        // N V2 (n x ns).  Levels below the first: on the fp64 matrix core, operands straight from HBM / L2 (round 5: acc_tile_product --
        // one entry per lane through the vector unit issued 2 nf gathered loads per entry, 1400 per lane on the 35-coordinate COMAN;
        // OSOT_NHQP_ACC_VALU keeps that form for A/B).  First level (N = I): the rows of V2 are copied.
#ifndef OSOT_NHQP_ACC_VALU
        if (!Q.first) {
            const int MB = uniform_i((n + 15) >> 4);
            if (MB <= 1) acc_tile_product<1>(Ng, Vg, Nn, n, nf, ns, lane);
            else if (MB == 2) acc_tile_product<2>(Ng, Vg, Nn, n, nf, ns, lane);
            else if (MB == 3) acc_tile_product<3>(Ng, Vg, Nn, n, nf, ns, lane);
            else acc_tile_product<4>(Ng, Vg, Nn, n, nf, ns, lane);
            return;
        }
#endif
        for (int e = lane; e < n * ns; e += 64) {
            const int i = e / ns, t = e - i * ns;
            double a2 = 0.0;
            if (Q.first) a2 = (i < nf) ? Vg[i * n + t] : 0.0;
            else {
                for (int j0 = 0; j0 < nf; j0 += 8) {
                    double a[8], b[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { const int j = j0 + u, jc = (j < nf) ? j : nf - 1; a[u] = Ng[i * n + jc]; b[u] = Vg[jc * n + t]; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) a2 = fma((j0 + u < nf) ? a[u] : 0.0, b[u], a2);
                }
            }
            Nn[i * n + t] = a2;
        }
    }
}

}  // namespace osot

// osot_kernels.h -- the gfx950 kernels of the OpenSoT hot path.
//
//   osot_cascade_kernel<NP> .. Solver::solve() = iHQP::solve (src/solvers/iHQP.cpp:263-358) for B
//                              instances: per level  H = A'WA, g = -A'Wb + c  (iHQP.cpp:129-162),
//                              constraints = global rows + optimality rows of the higher levels
//                              (iHQP.cpp:282-333), one strictly convex QP (QPOasesBackEnd.cpp:248-307),
//                              x of the last active level is the answer (iHQP.cpp:349).  All levels
//                              run in ONE launch; H, its factor, J and the working set never leave the
//                              CU (registers / LDS).  NP = 32: H = A'WA and the blocked Cholesky +
//                              inverse run on the fp64 matrix core (v_mfma_f64_16x16x4_f64), levels with
//                              <= 4 stored rows take a closed-form path, the optimality rows of a
//                              Postural last level are eliminated by a register-resident Gauss-Jordan.
//   osot_qp_kernel<NP> ....... B generic QPs in BackEnd convention (BackEnd.h:125-150).
//   osot_update_kernel ....... AutoStack::update() (src/utils/AutoStack.cpp:385-393): leaf -> b, W,
//                              merged box, collision / friction-cone rows, inverse-dynamics bounds.
//                              One 64-lane block per instance, one lane per stack row.
//   order_body ............... dispatch order of the next solve (longest first): an extra workgroup of the solve launch.
// (osot_kin.h holds the batched kinematics producer.)
//
// Mapping: ONE WAVEFRONT PER INSTANCE (one wavefront per workgroup).  NP = 32 for n <= 32 (two lanes per
// column: the halves split every inner product), NP = 64 for n <= 64.  Instance-major fp64 arrays, so a
// wave's reads of its stacked Jacobian rows are contiguous 8n-byte segments.
#pragma once
#include <osot_mi355x.h>   // OSOT_MAX_* (the C-ABI's limits are the kernels' limits)
#include "osot_qp_core.h"

#ifndef OSOT_WAVES32
#define OSOT_WAVES32 2   // waves per SIMD the NP = 32 cascade is compiled for (register budget 512 / OSOT_WAVES32).  3 (168
                         // VGPRs; with the 15.9 KB LDS slice of the packed R: 10 waves per CU, 2560 instances in flight) was
                         // measured: 43 spilled registers make every wave 10 % slower, and a batch of 4096 still needs two
                         // jobs on 1536 of the slots -- launch 168 us against 166 us, bench step 0.229 ms against 0.218 ms
#endif
#ifndef OSOT_WAVES40
#define OSOT_WAVES40 2   // waves per SIMD the NP = 40 instantiations (33 .. 38 variables) are compiled for
#endif
#define OSOT_KMAX_LEVELS 8
#define OSOT_KMAX_TASKS 8
#define OSOT_KMAX_FLAT_TASKS 24
#define OSOT_KMAX_BOUNDS 4
#define OSOT_KMAX_ROWBLOCKS 8
#define OSOT_KMAX_FLAT_ROWS 256

namespace osot {

struct DevPlan {
    int n, L, nc;
    int m[OSOT_KMAX_LEVELS];        // rows per level
    int ma[OSOT_KMAX_LEVELS];       // rows stored in A_k (the rest is Postural's implicit identity)
    int optoff[OSOT_KMAX_LEVELS + 1];  // prefix sums of m[]
    int ident_rows[OSOT_KMAX_LEVELS];  // rows of the implicit [I 0] block (Postural) of each level
    int nblocks, nc_stored;            // global row blocks; rows of C that are stored
    int blk_rows[OSOT_KMAX_ROWBLOCKS], blk_off[OSOT_KMAX_ROWBLOCKS], blk_stored_off[OSOT_KMAX_ROWBLOCKS];
    int blk_implicit[OSOT_KMAX_ROWBLOCKS], blk_first_col[OSOT_KMAX_ROWBLOCKS];
    int blk_level[OSOT_KMAX_ROWBLOCKS];   // 0: global rows; k + 1: task-local rows of level k (absent at the other levels)
    int max_iter;
    unsigned active_mask;           // bit k: level k active (iHQP::setActiveStack)
    // Task::setActive(false) (Task.h:232-239, 375-400): bit j of inactive[k] = task j of level k is inactive, i.e. its
    // rows [task_off[k][j], task_off[k][j+1]) count as zero rows: nothing in H and g, void optimality rows
    int ntask[OSOT_KMAX_LEVELS];
    int task_off[OSOT_KMAX_LEVELS][OSOT_KMAX_TASKS + 1];
    unsigned inactive[OSOT_KMAX_LEVELS];
    double eps_abs;
    // user regularisation task A_r = [I_rows 0], W_r = w I (iHQP.cpp:274-278): Hr = w on the first reg_rows diagonal
    // entries of every level's H, gr = -w b_r
    int reg_rows;
    double reg_w;
    int reg_dense;                  // 1: the regularisation task has a stored Jacobian A_r (DevBatch.A_reg): H += w A_r'A_r,
                                    // g -= w A_r'b_r at every level, in the EXTRA instantiation (no folding into the diagonal)
    // LDS carve-up of the wave's slice (doubles): M1, M2, V first (sizes fixed by NP), then
    int lds_rows_off;               // the row table: rlo, rup, rptr (8 B per row), rowstate, eqlist, rsrc (4 B per row)
    int lds_rows_cap;               // capacity (rows), even
    int rows_doubles;               // size of the row table (doubles)
    int rows_in_global;             // 1: the row table is the instance's slice of DevBatch.rows_scratch, not LDS
};

struct DevBatch {
    int B;
    const double* A[OSOT_KMAX_LEVELS];
    const double* b[OSOT_KMAX_LEVELS];
    const double* w[OSOT_KMAX_LEVELS];
    const double* c[OSOT_KMAX_LEVELS];
    const double* C;
    const double* lo;
    const double* up;
    const double* l;
    const double* u;
    double* dq;
    double* x_levels;
    int* status;
    int* iterations;
    long long* prof;   // [B][PH_COUNT] shader-clock cycles per phase (profiling instantiation only)
    const int* order;  // dispatch order: workgroup g solves instance order[g] (null: g).  Longest-first, see below
    int* cost_out;     // [B] active-set iterations of this solve = the cost estimate for the next dispatch
    const int* cost_in;  // [B] the estimates as the PREVIOUS launch left them (the filter's memory, and what this launch's
                         // order workgroup sorts); cost_in and cost_out are two buffers the solver alternates
    int* order_next;   // [B] dispatch order for the NEXT launch, written by this launch's extra workgroup (order_body); null:
                       // no extra workgroup in the grid
    int slots;         // wavefronts the chip holds at once for this kernel (see order_body)
    const double* b_reg;   // [B][reg_rows] b of the regularisation task (null: none)
    const double* A_reg;   // [B][reg_rows][n] its Jacobian when the plan says reg_dense
    const double* WA[OSOT_KMAX_LEVELS];   // [B][ma_k][n] W_k A_k, [B][m_k] W_k b_k of a level with a non-diagonal weight
    const double* Wb[OSOT_KMAX_LEVELS];   // (osot_update_kernel writes them); null: W_k is diag(w[k])
    double* accepted_slack;   // [B] largest constraint violation accepted as round-off (0: none); may be null
    double* rows_scratch;     // [B][rows_doubles] row tables of the instances when the plan keeps them out of LDS
    int* hot;                 // [B][L][NP] hot-start state: the inequality working set every level of every instance ended
                              // with (constraint codes, -1 = none), read at the start of a level and rewritten at its end
                              // (gi_inequalities); null: cold start, nothing recorded
};

// Row table of level k (LDS): [ global C rows ; A_0 ; ... ; A_{k-1} ]  (iHQP.cpp:282-333).  The C entries are
// filled once per instance, the optimality entries of level j are appended when level j has been solved
// (lo = up = A_j x_j, iHQP.cpp:164-170; an inactive level contributes 0*x in [-1,1], iHQP.cpp:301-309).
// EXTRA: the instantiation that knows about non-diagonal weights (WA / Wb operands) and Task::setActive(false); the plain
// one carries none of that code (registers, scalar loads, branches) -- launches pick it whenever the plan has no dense-weight
// level and every task is active
// BOX: plans without constraint rows (the bounds are the only inequalities; see gi_inequalities)
template <int NP, bool PROF, bool EXTRA, bool BOX = false>
__device__ __forceinline__ void cascade_body(const DevPlan& P, const DevBatch& D, const long long inst, const int lane, char* osot_smem) {
    constexpr int HV = WaveCtx<NP>::HV, S = WaveCtx<NP>::S;
    constexpr int HB = (NP % 16 == 0) ? 16 : 8;   // rows of H per broadcast group of the register H build
    const int n = P.n;
    double* base = reinterpret_cast<double*>(osot_smem);
    WaveCtx<NP> w;
    w.c = WaveCtx<NP>::col_of(lane); w.h = WaveCtx<NP>::half_of(lane); w.n = n;
    w.M1 = base;
    w.M2 = base + WaveCtx<NP>::M1_DOUBLES;
    w.V = w.M2 + WaveCtx<NP>::ROWS * S;
    // NP = 32 never moves the table out of LDS (osot_host_plan.h): said at compile time, the table's pointers are LDS pointers
    // (ds_read instead of flat_load, and loads from a uniform address are uniform values: a pointer that MAY be global makes
    // every table entry a divergent value, and every branch on one an exec-mask branch with the loop state in vector registers)
    if constexpr (NP == 32) w.rlo = base + P.lds_rows_off;
    else w.rlo = P.rows_in_global ? D.rows_scratch + inst * P.rows_doubles : base + P.lds_rows_off;
    w.rup = w.rlo + P.lds_rows_cap;
    w.rptr = reinterpret_cast<unsigned long long*>(w.rup + P.lds_rows_cap);
    w.rowstate = reinterpret_cast<int*>(w.rptr + P.lds_rows_cap);
    w.eqlist = w.rowstate + P.lds_rows_cap;
    w.safe_row = reinterpret_cast<unsigned long long>(D.dq + inst * n);   // n readable doubles (value is discarded)
    w.rsrc = reinterpret_cast<signed char*>(w.eqlist + P.lds_rows_cap);
    const int c = w.c, h = w.h;
    const bool valid = c < n;
    // zero the matrices once: the padding beyond n stays zero for the whole kernel
    for (int e = lane; e < WaveCtx<NP>::LDS_DOUBLES; e += 64) base[e] = 0.0;
    wave_sync();

    const bool has_box = D.l != nullptr;
    double lb = (has_box && valid) ? D.l[inst * n + c] : -INFINITY;   // (a level may relax a bound it accepted as satisfied)
    double ub = (has_box && valid) ? D.u[inst * n + c] : INFINITY;

    // global rows: bounds and row addresses, lane = row
    for (int j = 0; j < P.nblocks; ++j) {
        for (int q = lane; q < P.blk_rows[j]; q += 64) {
            const int r = P.blk_off[j] + q;
            w.rlo[r] = clamp_inf(D.lo[inst * P.nc + r]);
            w.rup[r] = clamp_inf(D.up[inst * P.nc + r]);
            w.rptr[r] = P.blk_implicit[j]
                ? ((((unsigned long long)(P.blk_first_col[j] + q)) << 1) | 1ull)   // unit row, not stored
                : reinterpret_cast<unsigned long long>(D.C + (inst * P.nc_stored + P.blk_stored_off[j] + q) * n);
            w.rsrc[r] = -1;
        }
    }
    wave_sync();

    // regularisation task (same for every level): its diagonal joins eps, its linear term joins g
    const bool regd = EXTRA && P.reg_dense && D.b_reg != nullptr;       // stored Jacobian: added row by row in the H build
    const bool regc = D.b_reg != nullptr && !regd && c < P.reg_rows;
    const double greg = regc ? -P.reg_w * D.b_reg[inst * P.reg_rows + c] : 0.0;
    const double dreg = regc ? P.reg_w : 0.0;

    double x = 0.0;
    int status = QP_SOLVED;
    int iters_total = 0;
    bool any = false;
    double slack = 0.0;   // largest violation a level accepted as round-off (kSlackTol), 0 if none
    long long prof[PH_COUNT];
    if (PROF) for (int i = 0; i < PH_COUNT; ++i) prof[i] = 0;
    const long long t_begin = PROF ? (long long)clock64() : 0;
    for (int k = 0; k < P.L; ++k) {
        if (!((P.active_mask >> k) & 1u)) {
            // inactive level: its optimality rows are 0*x in [-1, 1] (iHQP.cpp:301-309): never binding
            if (k + 1 < P.L) {
                const int off = P.nc + P.optoff[k];
                for (int q = lane; q < P.m[k]; q += 64) {
                    w.rlo[off + q] = -kInfty; w.rup[off + q] = kInfty; w.rptr[off + q] = (0x7fffffffull << 1) | 1ull;
                    w.rsrc[off + q] = -1;
                }
                wave_sync();
            }
            continue;
        }
        // task-local row blocks (Task::getConstraints(), iHQP.cpp:190, 282-287): real bounds at their own level,
        // "absent" (infinite bounds: never scanned as violated, never an equality) at every other level
        for (int j = 0; j < P.nblocks; ++j) {
            if (P.blk_level[j] == 0) continue;
            const bool on = P.blk_level[j] - 1 == k;
            for (int q = lane; q < P.blk_rows[j]; q += 64) {
                const int r = P.blk_off[j] + q;
                w.rlo[r] = on ? clamp_inf(D.lo[inst * P.nc + r]) : -kInfty;
                w.rup[r] = on ? clamp_inf(D.up[inst * P.nc + r]) : kInfty;
                w.rsrc[r] = on ? -2 : -1;   // -2: the previous level's solution owes this row nothing (osot_qp_core.h)
            }
            wave_sync();
        }
        OSOT_PH_BEGIN();
        {   // re-derive the lane coordinates per level (see launder_i)
            const int l2 = launder_i(lane);
            w.c = WaveCtx<NP>::col_of(l2); w.h = WaveCtx<NP>::half_of(l2);
        }
        const int c = w.c, h = w.h;
        const bool valid = c < n;
        const int m = P.m[k], ma = P.ma[k];
        // hot start (osot_solver_set_hotstart, default off) lives in the EXTRA instantiation only: the plain one carries none of its
        // code (its branches at the top of every active-set trip, a vector and a handful of scalar registers: measured +4 %)
        int* hotk = (EXTRA && D.hot) ? D.hot + (inst * P.L + k) * WaveCtx<NP>::LW : nullptr;
        const int hotcode = hotk ? hotk[c] : -1;   // (requested here: the answer is not needed before the inequality loop)
        const double* Ak = D.A[k] ? D.A[k] + inst * ma * n : nullptr;
        const double* bk = D.b[k] + inst * m;
        const double* wk = D.w[k] ? D.w[k] + inst * m : nullptr;
        // non-diagonal W_k: left operand W_k A_k and W_k b_k come from the update kernel (stored rows only: an implicit
        // Postural block keeps its diagonal weights w)
        const bool dense = EXTRA && D.WA[k] != nullptr;
        const double* WAk = dense ? D.WA[k] + inst * ma * n : nullptr;
        const double* Wbk = dense ? D.Wb[k] + inst * m : nullptr;
        // Task::setActive(false): rows of an inactive task are zero rows
        const unsigned inact = EXTRA ? P.inactive[k] : 0u;
        auto row_off = [&](int r) -> bool {
            bool off = false;
            for (int j = 0; j < P.ntask[k]; ++j)
                off = off || (((inact >> j) & 1u) && r >= P.task_off[k][j] && r < P.task_off[k][j + 1]);
            return off;
        };
        auto wrow = [&](int r) -> double { return (inact && row_off(r)) ? 0.0 : (wk ? wk[r] : 1.0); };
        double g = 0.0, hdiag = 0.0;
        const bool diag_h = (ma == 0) && !regd;
        double hacc[NP / HV];
        // (defined on EVERY path through the level: left undefined on the diagonal / low-rank paths, the compiler resolves the merge by
        //  keeping the PREVIOUS level's tiles alive across the whole loop body -- 2 x NP / HV registers through the active-set loops,
        //  spilled and reloaded at the loop's back edge in the 64-lane layouts)
#ifndef OSOT_X_HACC_UNDEF
#pragma unroll
        for (int ii = 0; ii < NP / HV; ++ii) hacc[ii] = 0.0;
#endif
        // round 6, 40-lane layout: a dense level (diagonal weights, every task active) under >= 8 equality rows that the previous level's
        // solution satisfies -- the reference's COMAN stacks below their first level -- takes the null-space method AHEAD of the H build:
        // no H, no 40-column factorisation, no row-by-row reflections (nullspace_dense_wide).  -1: not taken (or handed back: rank-deficient
        // rows, more than 24 free columns), the level runs as before.
        int dn_rank = -1, dn_neq = 0;
        double dn_x = 0.0, dn_hinv = 0.0;
        if constexpr (NP == 40 && kDenseNull40) {
            if (!diag_h && ma <= kDenseNullRows && !dense && !inact && !regd) {
                const int nrows_k = P.nc + P.optoff[k];
                int ne = 0;
                bool loc = false;
                for (int r0 = 0; r0 < nrows_k; r0 += 64) {
                    const int r = r0 + lane;
                    bool is_eq = false, is_loc = false;
                    if (r < nrows_k) {
                        const double lo = w.rlo[r], up = w.rup[r];
                        is_eq = (lo == up) && (lo > -kInfty) && (lo < kInfty);
                        is_loc = is_eq && (w.rsrc[r] == -2);
                    }
                    loc = loc || (wave_ballot(is_loc) != 0ull);
                    const unsigned long long mask = wave_ballot(is_eq);
                    if (is_eq) w.eqlist[ne + lanes_below(mask)] = r;
                    ne += __builtin_popcountll(mask);
                }
                wave_sync();
                if (!loc && ne >= 8 && ne <= 32 && n - ne <= kDenseNullFree) {
                    const int npost = m - ma;
                    const bool postc = valid && c < npost;
                    const double wpost = postc ? (wk ? wk[ma + c] : 1.0) : 0.0;
                    double clin = ((D.c[k] && valid) ? D.c[k][inst * n + c] : 0.0) + greg;
                    if (postc) clin -= wpost * bk[ma + c];
                    dn_rank = uniform_i(nullspace_dense_wide<NP, PROF>(w, ne, Ak, bk, wk, ma, P.eps_abs + wpost + dreg, clin, any, x, dn_x, dn_hinv, prof));
                    dn_neq = ne;
                }
            }
        }
        const bool dense_done = dn_rank >= 0;
        bool lowrank = false;   // few stored rows: J and x in closed form, no H, no factorisation (lowrank_prepare32)
        double xprep = 0.0;
        if constexpr (NP == 32 || (NP == 40 && kLowRank40)) {
#ifndef OSOT_X_NO_LOWRANK
            // (five or six stored rows -- one Cartesian task -- only next to a Postural block over every variable, BASELINE config 2: D >= w
            //  there.  With D = eps alone the scaled rows carry 1 / sqrt(eps) and six of them lose what the Cholesky path keeps: at the
            //  default eps the closed-loop instance of default_eps_stuck_instances[tasks] ended lexicographically worse than eiQuadProg)
            if (!dense_done && !diag_h && (ma <= 4 || (ma <= kLowRankMax && m - ma >= n)) && !dense && !inact && !regd) {
#else
            if (false) {
#endif
                lowrank = true;
                const int npost = m - ma;
                const bool postc = valid && c < npost;
                const double wpost = postc ? (wk ? wk[ma + c] : 1.0) : 0.0;
                double cvec = ((D.c[k] && valid) ? D.c[k][inst * n + c] : 0.0) + greg;
                if (postc) cvec -= wpost * bk[ma + c];
                const bool has_c = D.c[k] != nullptr || npost > 0 || D.b_reg != nullptr;
                if (ma <= 3) lowrank_prepare<NP, 3>(w, Ak, bk, wk, ma, P.eps_abs + wpost + dreg, cvec, has_c, xprep);
                else if (kLowRankMax <= 4 || ma <= 4) lowrank_prepare<NP, (kLowRankMax < 4 ? kLowRankMax : 4)>(w, Ak, bk, wk, ma, P.eps_abs + wpost + dreg, cvec, has_c, xprep);
                else lowrank_prepare<NP, kLowRankMax>(w, Ak, bk, wk, ma, P.eps_abs + wpost + dreg, cvec, has_c, xprep);
            }
        }
        if (lowrank || dense_done) {
        } else if (!diag_h) {
          if constexpr (NP == 32) {
            // ---- H = A'WA + eps I on the fp64 MATRIX CORE, g = -A'Wb + c.  Four rows of A per step: lane
            // l = (a, q) = (l & 15, l >> 4) loads A[r0 + q][a] and A[r0 + q][16 + a] (two coalesced loads cover the
            // four rows); those two registers ARE the MFMA operands of every 16 x 16 tile of the outer product
            // (A-operand m = column-in-tile, k = row; B-operand likewise), so no LDS broadcast and no VALU FMA
            // is spent on the 32 x 32 x m product.  Tile (I, C) element r of lane l is H[16 I + q + 4 r][16 C + a].
            const int ta = lane & 15, tq = lane >> 4;
            v4f64 Ht[2][2];
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int C2 = 0; C2 < 2; ++C2) Ht[I][C2] = v4f64{0.0, 0.0, 0.0, 0.0};
            double gp0 = 0.0, gp1 = 0.0;
            // (a0, a1): the row of A (right operand);  (l0, l1): the row of W A (left operand) -- w_r * a for a diagonal W,
            // read from WA otherwise;  gb: b_r, or (W b)_r for a non-diagonal W: g = -A'(W b) (iHQP.cpp:153, Task::getWb)
            // b_r and w_r of up to 64 rows in ONE load each, lane = row, handed to the quarter-rows through the LDS crossbar
            // (ds_bpermute): as uniform-address loads inside fetch() they were a third of the level's vector-memory
            // instructions, and eight synchronised wavefronts per CU queue on the CU's one address unit
            double bw_b = 0.0, bw_w = 1.0;
            int bw_base = -1;
            auto fetch = [&](int r0, double& a0, double& a1, double& l0, double& l1, double& gb) {
                const int r = r0 + tq;
                const bool in = r < ma && !(inact && row_off(r));
                const int rr = (r < ma) ? r : ma - 1;
                const int c0 = (ta < n) ? ta : 0, c1 = (16 + ta < n) ? 16 + ta : 0;
                const double v0 = Ak[rr * n + c0];
                const double v1 = Ak[rr * n + c1];
                a0 = (in && ta < n) ? v0 : 0.0;
                a1 = (in && 16 + ta < n) ? v1 : 0.0;
                if (dense) {
                    gb = Wbk[rr];
                    const double u0 = WAk[rr * n + c0], u1 = WAk[rr * n + c1];
                    l0 = (in && ta < n) ? u0 : 0.0;
                    l1 = (in && 16 + ta < n) ? u1 : 0.0;
                } else {
                    const int src4 = (rr - bw_base) << 2;
                    gb = permute_f64(bw_b, src4);
                    const double wv = wk ? permute_f64(bw_w, src4) : 1.0;
                    l0 = wv * a0; l1 = wv * a1;
                }
            };
            // 32 rows at a time: all eight groups of four rows are requested before the first MFMA (32 fp64
            // registers in flight): one HBM/L2 round trip per 32 rows instead of one per group
#ifndef OSOT_HB_DEPTH
#define OSOT_HB_DEPTH 8
#endif
            constexpr int HBD = OSOT_HB_DEPTH;   // groups of four rows requested before the first MFMA
            for (int rb = 0; rb < ma; rb += 4 * HBD) {
                if (!dense && (rb & 63) == 0) {          // rows rb .. rb + 63 of b and w, lane = row
                    const int rl = rb + lane;
                    bw_b = (rl < ma) ? bk[rl] : 0.0;
                    bw_w = (wk && rl < ma) ? wk[rl] : 1.0;
                    bw_base = rb;
                }
                double ca0[HBD], ca1[HBD], cl0[HBD], cl1[HBD], cbr[HBD];
#pragma unroll
                for (int ch = 0; ch < HBD; ++ch) {
                    ca0[ch] = 0.0; ca1[ch] = 0.0; cl0[ch] = 0.0; cl1[ch] = 0.0; cbr[ch] = 0.0;
                    if (rb + 4 * ch < ma) fetch(rb + 4 * ch, ca0[ch], ca1[ch], cl0[ch], cl1[ch], cbr[ch]);
                }
#pragma unroll
                for (int ch = 0; ch < HBD; ++ch) {
                    if (rb + 4 * ch < ma) {
                        const double a0 = ca0[ch], a1 = ca1[ch], br = cbr[ch];
                        const double wa0 = cl0[ch], wa1 = cl1[ch];
                        gp0 = fma(dense ? -a0 : -wa0, br, gp0);
                        gp1 = fma(dense ? -a1 : -wa1, br, gp1);
                        Ht[0][0] = mfma_f64_16x16x4(wa0, a0, Ht[0][0]);
                        Ht[0][1] = mfma_f64_16x16x4(wa0, a1, Ht[0][1]);
                        Ht[1][1] = mfma_f64_16x16x4(wa1, a1, Ht[1][1]);
                    }
                }
            }
            if (regd) {   // the regularisation task's rows: H += w A_r'A_r, g -= w A_r'b_r (iHQP.cpp:274-278), same operand layout
                const double* Ar = D.A_reg + inst * (long long)P.reg_rows * n;
                const double* br = D.b_reg + inst * P.reg_rows;
                for (int r0 = 0; r0 < P.reg_rows; r0 += 4) {
                    const int r = r0 + tq;
                    const bool in = r < P.reg_rows;
                    const int rr = in ? r : P.reg_rows - 1;
                    const double v0 = Ar[rr * n + ((ta < n) ? ta : 0)], v1 = Ar[rr * n + ((16 + ta < n) ? 16 + ta : 0)];
                    const double a0 = (in && ta < n) ? v0 : 0.0, a1 = (in && 16 + ta < n) ? v1 : 0.0;
                    const double wa0 = P.reg_w * a0, wa1 = P.reg_w * a1, bb = br[rr];
                    gp0 = fma(-wa0, bb, gp0);
                    gp1 = fma(-wa1, bb, gp1);
                    Ht[0][0] = mfma_f64_16x16x4(wa0, a0, Ht[0][0]);
                    Ht[0][1] = mfma_f64_16x16x4(wa0, a1, Ht[0][1]);
                    Ht[1][1] = mfma_f64_16x16x4(wa1, a1, Ht[1][1]);
                }
            }
            OSOT_PH_END(PH_INV);   // (profiling slot reused: MFMA loop of the H build)
            // g: the partial sums of a column sit in the four rows of 16 lanes
            gp0 = rowgroup_sum(gp0);
            gp1 = rowgroup_sum(gp1);
            g = valid ? ((tq & 1) ? gp1 : gp0) : 0.0;
            const int npost = m - ma;   // Postural block appended to the level: A = [I 0] (Postural.cpp:37)
            if (npost > 0 && c < npost) g -= wrow(ma + c) * bk[ma + c];
            // diagonal: Postural weights, eps I, unit diagonal beyond n (see factor_rows64 for the padding)
#pragma unroll
            for (int I = 0; I < 2; ++I) {
                const int i = 16 * I + ta;    // diagonal element (i, i) lives in tile (I, I) where a == q + 4 r
                double dv = (i < n) ? P.eps_abs : 1.0;
                if (i < npost) dv += wrow(ma + i);
                if (D.b_reg && !regd && i < P.reg_rows) dv += P.reg_w;
#pragma unroll
                for (int r = 0; r < 4; ++r) Ht[I][I][r] += (ta == tq + 4 * r) ? dv : 0.0;
            }
            OSOT_PH_END(PH_SUBST);   // (profiling slot reused: g reduction + diagonal of the H build)
            // the factorisation (factor_tiles32) works on the tiles as they are: hacc[4 (2 I + C) + r]
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int C2 = 0; C2 < 2; ++C2)
#pragma unroll
                    for (int r = 0; r < 4; ++r) hacc[4 * (2 * I + C2) + r] = Ht[I][C2][r];
          } else if constexpr (kWideTiles) {
            // ---- round 5: the same H build on the matrix core for the 64-lane layouts (33 .. 64 variables): T x T tiles of
            // 16 x 16, the UPPER triangle only (H is symmetric and factor_tiles_wide reads nothing else).  Lane (a, q) of the
            // PHYSICAL lane loads A[r0 + q][16 X + a], X < T: T coalesced loads cover four rows, and those registers are the
            // MFMA operands of every tile -- where the register build below spends an LDS broadcast and NP fma per row and lane.
            constexpr int T = wide_tiles(NP), NT = (T * (T + 1)) / 2;
            const int ta = lane & 15, tq = lane >> 4;
            v4f64 Ht[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) Ht[t] = v4f64{0.0, 0.0, 0.0, 0.0};
            double gp[T];
#pragma unroll
            for (int X = 0; X < T; ++X) gp[X] = 0.0;
            double bw_b = 0.0, bw_w = 1.0;     // b_r and w_r of up to 64 rows, lane = row (see the NP = 32 branch)
            int bw_base = -1;
            auto fetch = [&](int r0, double (&a)[T], double (&l)[T], double& gb) {
                const int r = r0 + tq;
                const bool in = r < ma && !(inact && row_off(r));
                const int rr = (r < ma) ? r : ma - 1;
                double v[T];
#pragma unroll
                for (int X = 0; X < T; ++X) v[X] = Ak[rr * n + ((16 * X + ta < n) ? 16 * X + ta : 0)];
#pragma unroll
                for (int X = 0; X < T; ++X) a[X] = (in && 16 * X + ta < n) ? v[X] : 0.0;
                if (dense) {
                    gb = Wbk[rr];
                    double u[T];
#pragma unroll
                    for (int X = 0; X < T; ++X) u[X] = WAk[rr * n + ((16 * X + ta < n) ? 16 * X + ta : 0)];
#pragma unroll
                    for (int X = 0; X < T; ++X) l[X] = (in && 16 * X + ta < n) ? u[X] : 0.0;
                } else {
                    const int src4 = (rr - bw_base) << 2;
                    gb = permute_f64(bw_b, src4);
                    const double wv = wk ? permute_f64(bw_w, src4) : 1.0;
#pragma unroll
                    for (int X = 0; X < T; ++X) l[X] = wv * a[X];
                }
            };
#ifndef OSOT_HB_DEPTH_WIDE
#define OSOT_HB_DEPTH_WIDE 4
#endif
            constexpr int HBW = OSOT_HB_DEPTH_WIDE;   // groups of four rows requested before the first MFMA
            for (int rb = 0; rb < ma; rb += 4 * HBW) {
                if (!dense && (rb & 63) == 0) {          // rows rb .. rb + 63 of b and w, lane = row
                    const int rl = rb + lane;
                    bw_b = (rl < ma) ? bk[rl] : 0.0;
                    bw_w = (wk && rl < ma) ? wk[rl] : 1.0;
                    bw_base = rb;
                }
                double ca[HBW][T], cl[HBW][T], cbr[HBW];
#pragma unroll
                for (int ch = 0; ch < HBW; ++ch) {
#pragma unroll
                    for (int X = 0; X < T; ++X) { ca[ch][X] = 0.0; cl[ch][X] = 0.0; }
                    cbr[ch] = 0.0;
                    if (rb + 4 * ch < ma) fetch(rb + 4 * ch, ca[ch], cl[ch], cbr[ch]);
                }
#pragma unroll
                for (int ch = 0; ch < HBW; ++ch) {
                    if (rb + 4 * ch < ma) {
#pragma unroll
                        for (int X = 0; X < T; ++X) gp[X] = fma(dense ? -ca[ch][X] : -cl[ch][X], cbr[ch], gp[X]);
#pragma unroll
                        for (int I = 0; I < T; ++I)
#pragma unroll
                            for (int C2 = I; C2 < T; ++C2)
                                Ht[tile_u<T>(I, C2)] = mfma_f64_16x16x4(cl[ch][I], ca[ch][C2], Ht[tile_u<T>(I, C2)]);
                    }
                }
            }
            if (regd) {   // the regularisation task's rows: H += w A_r'A_r, g -= w A_r'b_r (iHQP.cpp:274-278), same operand layout
                const double* Ar = D.A_reg + inst * (long long)P.reg_rows * n;
                const double* br = D.b_reg + inst * P.reg_rows;
                for (int r0 = 0; r0 < P.reg_rows; r0 += 4) {
                    const int r = r0 + tq;
                    const bool in = r < P.reg_rows;
                    const int rr = in ? r : P.reg_rows - 1;
                    double a[T], wa[T];
#pragma unroll
                    for (int X = 0; X < T; ++X) a[X] = Ar[rr * n + ((16 * X + ta < n) ? 16 * X + ta : 0)];
                    const double bb = br[rr];
#pragma unroll
                    for (int X = 0; X < T; ++X) {
                        a[X] = (in && 16 * X + ta < n) ? a[X] : 0.0;
                        wa[X] = P.reg_w * a[X];
                        gp[X] = fma(-wa[X], bb, gp[X]);
                    }
#pragma unroll
                    for (int I = 0; I < T; ++I)
#pragma unroll
                        for (int C2 = I; C2 < T; ++C2) Ht[tile_u<T>(I, C2)] = mfma_f64_16x16x4(wa[I], a[C2], Ht[tile_u<T>(I, C2)]);
                }
            }
            OSOT_PH_END(PH_INV);   // (profiling slot reused: MFMA loop of the H build)
            // g: the partial sums of a column sit in the four rows of 16 lanes; lane = column 16 q + a takes tile column q
            double gsel = 0.0;
#pragma unroll
            for (int X = 0; X < T; ++X) { const double gs = rowgroup_sum(gp[X]); gsel = (tq == X) ? gs : gsel; }
            g = (lane < n) ? gsel : 0.0;
            const int npost = m - ma;   // Postural block appended to the level: A = [I 0] (Postural.cpp:37)
            if (npost > 0 && c < npost) g -= wrow(ma + c) * bk[ma + c];
            // diagonal: Postural weights, eps I (+ the regularisation task's diagonal), unit diagonal beyond n
#pragma unroll
            for (int I = 0; I < T; ++I) {
                const int i = 16 * I + ta;    // diagonal element (i, i) lives in tile (I, I) where a == q + 4 r
                double dv = (i < n) ? P.eps_abs : 1.0;
                if (i < npost) dv += wrow(ma + i);
                if (D.b_reg && !regd && i < P.reg_rows) dv += P.reg_w;
#pragma unroll
                for (int r = 0; r < 4; ++r) Ht[tile_u<T>(I, I)][r] += (ta == tq + 4 * r) ? dv : 0.0;
            }
            OSOT_PH_END(PH_SUBST);   // (profiling slot reused: g reduction + diagonal of the H build)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) hacc[4 * t + r] = Ht[t][r];
          } else {
            // ---- H = A'WA + eps I, g = -A'Wb + c.  Lane (c,h) accumulates H[i][c] for i = ii*HV + h in
            // registers; stored rows are staged four at a time through LDS for the broadcasts.
#pragma unroll
            for (int ii = 0; ii < NP / HV; ++ii) hacc[ii] = 0.0;
            for (int r0 = 0; r0 < ma; r0 += 4) {
                OSOT_SUB_BEGIN();
                double a[4], wa[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = r0 + u;
                    const bool in = r < ma && !(inact && row_off(r));
                    a[u] = (in && valid) ? Ak[r * n + c] : 0.0;
                    if (dense) wa[u] = (in && valid) ? WAk[r * n + c] : 0.0;
                    else wa[u] = (in ? (wk ? wk[r] : 1.0) : 0.0) * a[u];
                    g -= dense ? a[u] * (in ? Wbk[r] : 0.0) : wa[u] * (in ? bk[r] : 0.0);
                }
                wave_sync();   // the previous group's broadcasts are done
                OSOT_SUB_END(PH_INV);     // (profiling slot reused: wait for the rows)
                if (h == 0) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) w.V[u * WaveCtx<NP>::LW + c] = a[u];
                }
                wave_sync();
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double* Vu = w.V + u * WaveCtx<NP>::LW + h;
#pragma unroll
                    for (int i0 = 0; i0 < NP / HV; i0 += HB) {
                        double vv[HB];
#pragma unroll
                        for (int t = 0; t < HB; ++t) vv[t] = Vu[(i0 + t) * HV];
#pragma unroll
                        for (int t = 0; t < HB; ++t) hacc[i0 + t] = fma(wa[u], vv[t], hacc[i0 + t]);
                    }
                }
                OSOT_SUB_END(PH_SUBST);   // (profiling slot reused: LDS broadcast + outer product)
            }
            if (regd) {   // the regularisation task's stored rows (see the NP = 32 branch)
                const double* Ar = D.A_reg + inst * (long long)P.reg_rows * n;
                const double* br = D.b_reg + inst * P.reg_rows;
                for (int r0 = 0; r0 < P.reg_rows; r0 += 4) {
                    double a[4], wa[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r = r0 + u;
                        const bool in = r < P.reg_rows;
                        a[u] = (in && valid) ? Ar[r * n + c] : 0.0;
                        wa[u] = P.reg_w * a[u];
                        g -= wa[u] * (in ? br[r] : 0.0);
                    }
                    wave_sync();
                    if (h == 0) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) w.V[u * WaveCtx<NP>::LW + c] = a[u];
                    }
                    wave_sync();
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const double* Vu = w.V + u * WaveCtx<NP>::LW + h;
#pragma unroll
                        for (int i0 = 0; i0 < NP / HV; i0 += HB) {
                            double vv[HB];
#pragma unroll
                            for (int t = 0; t < HB; ++t) vv[t] = Vu[(i0 + t) * HV];
#pragma unroll
                            for (int t = 0; t < HB; ++t) hacc[i0 + t] = fma(wa[u], vv[t], hacc[i0 + t]);
                        }
                    }
                }
            }
            if (m > ma && c < m - ma) {   // Postural block appended to the level: A = [I 0] (Postural.cpp:37)
                const double wi = wrow(ma + c);
                g -= wi * bk[ma + c];
#pragma unroll
                for (int ii = 0; ii < NP / HV; ++ii) if (ii * HV + h == c) hacc[ii] += wi;
            }
            wave_sync();
#pragma unroll
            for (int ii = 0; ii < NP / HV; ++ii) {
                const int i = ii * HV + h;
                // factorised straight from these registers; unit diagonal beyond n (see factor_rows64)
                hacc[ii] += (i == c) ? (valid ? P.eps_abs + dreg : 1.0) : 0.0;
            }
            wave_sync();
          }
        } else if (valid) {   // level = one Postural block [I_m 0]: H = blockdiag(W, 0) + eps I is diagonal
            const bool inb = c < m;
            const double wi = inb ? wrow(c) : 0.0;
            hdiag = wi + P.eps_abs + dreg;
            g = inb ? -wi * bk[c] : 0.0;
        }
        if (D.c[k] && valid && !lowrank) g += D.c[k][inst * n + c];
        if (!lowrank) g += greg;

        const int nrows = P.nc + P.optoff[k];
        int iters = 0;
        OSOT_PH_END(PH_HBUILD);
        int st;
        if (NP > 32) {   // must be inlined: hacc would otherwise be passed through scratch memory
            OSOT_ALWAYS_INLINE_CALL st = gi_solve<NP, PROF, BOX>(w, nrows, g, diag_h, hdiag, hacc,
                                                                   has_box, lb, ub, P.max_iter, any, x, x, iters, prof, slack,
                                                                   lowrank, xprep, hotcode, hotk, dn_rank, dn_neq, dn_x, dn_hinv);
        } else {          // NP = 32: inlined as well (as a CALL the solver spends ~50 % more cycles: the tiles travel through scratch)
            OSOT_ALWAYS_INLINE_CALL st = gi_solve<NP, PROF, BOX>(w, nrows, g, diag_h, hdiag, hacc,
                                           has_box, lb, ub, P.max_iter, any, x, x, iters, prof, slack, lowrank, xprep,
                                           hotcode, hotk);
        }
        if (PROF) ph_t0_ = (long long)clock64();
#ifdef OSOT_TRACE_LEVELS   // developer knob (emulator builds): one line per (instance, level) with the level's trip count
        if (lane == 0) printf("LEVEL %lld %d %d %d\n", inst, k, iters, st);
#endif
        iters_total += iters;
        if (st != QP_SOLVED) { status = st; break; }
        any = true;
        if (D.x_levels && valid && h == 0) D.x_levels[(inst * P.L + k) * n + c] = x;
        // optimality rows A_k x = A_k x_k for the lower levels (iHQP.cpp:164-170) -> row table.  Only the
        // row addresses and x_k are recorded: the right-hand side is taken relative to the current iterate,
        // a'(x_k - x), when the row is added, so A_k is not re-read and no reduction pass is needed here.
        if (k + 1 < P.L) {
            const int off = P.nc + P.optoff[k];
            for (int q = lane; q < m; q += 64) {
                const bool void_row = inact && row_off(q);   // inactive task: 0 x = 0 (Task.h:383-387), i.e. no row
                w.rlo[off + q] = void_row ? -kInfty : 0.0;
                w.rup[off + q] = void_row ? kInfty : 0.0;
                w.rsrc[off + q] = void_row ? (signed char)-1 : (signed char)k;
                w.rptr[off + q] = void_row ? ((0x7fffffffull << 1) | 1ull)
                                : (q < ma) ? reinterpret_cast<unsigned long long>(Ak + q * n)
                                           : (((unsigned long long)(q - ma) << 1) | 1ull);   // Postural: e_(q-ma)
            }
            wave_sync();
        }
        OSOT_PH_END(PH_OPT);
    }
    if (PROF && D.prof && lane == 0) {
        prof[PH_TOTAL] = (long long)clock64() - t_begin;
        for (int i = 0; i < PH_COUNT; ++i) D.prof[inst * PH_COUNT + i] = prof[i];
    }
    if (status != QP_SOLVED || !any) x = 0.0;   // failed instances return dq = 0 (coman_ik.cpp:189-190)
    if (valid && h == 0) D.dq[inst * n + c] = x;
    if (lane == 0) {
        D.status[inst] = status;
        if (D.iterations) D.iterations[inst] = iters_total;
#ifndef OSOT_COST_EMA
#define OSOT_COST_EMA 1
#endif
        // cost estimate for the next dispatch, fixed point x4: this solve's iteration count blended 1:1 with the previous
        // estimate.  An instance's count jitters by a few iterations from cycle to cycle around a level that drifts slowly:
        // the filtered value predicts the next cycle better than the last sample alone (bench, drifting cycles: 0.196 ->
        // 0.175 ms per launch with 4 cycles in rotation, 0.206 -> 0.190 with 8, 0.212 -> 0.203 with 16)
        if (D.cost_out) {
            const int now = 4 * iters_total;
            D.cost_out[inst] = (OSOT_COST_EMA && D.order && D.cost_in) ? ((D.cost_in[inst] + now + 1) >> 1) : now;
        }
        if (D.accepted_slack) D.accepted_slack[inst] = slack;
    }
}

__device__ __forceinline__ long long dispatch_instance(const DevBatch& D, char* smem);   // (below, with the order workgroup)

template <int NP, bool PROF, bool EXTRA = false, bool BOX = false>
__global__ void __launch_bounds__(64, (NP == 32 ? OSOT_WAVES32 : (NP == 40 ? OSOT_WAVES40 : 1))) osot_cascade_kernel(const DevPlan P, const DevBatch D) {
    OSOT_DYNAMIC_LDS(osot_smem);
    const long long inst = dispatch_instance(D, osot_smem);
    if (inst < 0) return;
    cascade_body<NP, PROF, EXTRA, BOX>(P, D, inst, (int)threadIdx.x, osot_smem);
}

// Longest-first dispatch.  One wavefront solves one instance and an MI355X holds 2048 of them at a time, so a
// batch of 4096 is two rounds: the kernel ends when the slowest LATE starter ends, and with the active-set
// iteration count varying 2x between instances (mean 32, max 63 at BASELINE config 3) that tail was ~30 % of the
// launch.  Control loops are temporally coherent: an instance's iteration count changes slowly from one cycle to
// the next.  So the instances are dispatched in DESCENDING order of a filtered iteration count of their previous solves
// (classic longest-processing-time list scheduling).  The order is built by ONE EXTRA WORKGROUP OF THE SOLVE LAUNCH ITSELF
// (block 0, round 3; a launch of its own before: ~5 us of kernel plus two launch gaps in every step's dependent chain): while
// the other workgroups solve step t, it sorts the estimates step t-1 left behind (cost_in: complete, nobody writes it during
// this launch) into the order step t+1 will use.  The order is one step staler than a sort between the launches would be; the
// estimates are exponential averages, so that costs nothing measurable.  A counting sort of the instance ids by
// min(cost, 255), 64 lanes, histogram and bin cursors in LDS.  Results do not depend on the order (instances are independent).
//
// `slots` = wavefronts the chip holds at once for this kernel (CUs x resident waves per CU).  The sorted list (descending
// cost) is mapped to dispatch positions by the number of rounds the batch makes:
//   B <= slots or B >= 2 slots : longest first (classic LPT list scheduling)
//   slots < B < 2 slots        : k = B - slots jobs must start late, so 2k jobs share a slot and slots - k run alone.  The
//                                ones that run alone are the LONGEST (the launch cannot end before its longest job anyway);
//                                the 2k shortest share: first the k shortest (ascending: they free their slots early, in
//                                that order), then the k medium ones in DESCENDING order, so that the first slot to free
//                                takes the longest of them.  With plain LPT the last slot to free (the longest job's)
//                                would have taken a late job: measured at BASELINE config 3, B = 4096 on 2560 slots, the
//                                launch ended 20 us after its longest wave (tools/prof_cycle.py).
#ifndef OSOT_EMULATION   // (LDS atomics: outside what tests/emu models; covered by the GPU tests)
__device__ __forceinline__ void order_body(const int* cost, int* order, int B, int slots, int lane, char* smem) {
    int* hist = reinterpret_cast<int*>(smem);   // [256] instances per key
    int* start = hist + 256;                    // [256] next rank (descending order of key) of each key
    for (int i = lane; i < 256; i += 64) hist[i] = 0;
    __syncthreads();
    auto key = [&](int i) { int k = cost[i]; return k < 0 ? 0 : (k > 255 ? 255 : k); };
    for (int i = lane; i < B; i += 64) __hip_atomic_fetch_add(&hist[key(i)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    // start[k] = number of instances with a larger key: lane l owns the bins 4 l .. 4 l + 3; suffix sums across the lanes
    const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
    const int tot = h0 + h1 + h2 + h3;
    int v = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_down(v, off, 64);
        v += (lane + off < 64) ? t : 0;
    }
    const int above = v - tot;                  // instances in the bins of the lanes above this one
    start[4 * lane + 3] = above;
    start[4 * lane + 2] = above + h3;
    start[4 * lane + 1] = above + h3 + h2;
    start[4 * lane] = above + h3 + h2 + h1;
    __syncthreads();
    const int late = B - slots;                       // jobs that cannot be in the first round
    const bool paired = late > 0 && B < 2 * slots;
    for (int i = lane; i < B; i += 64) {
        const int pos = __hip_atomic_fetch_add(&start[key(i)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // rank, descending cost
        int where = pos;
        if (paired) {
            const int alone = slots - late;           // the longest `alone` jobs keep a slot to themselves
            if (pos >= B - late) where = alone + (B - 1 - pos);          // the `late` shortest: ascending, right after
            else if (pos >= alone) where = slots + (pos - alone);        // the medium ones: descending, second round
        }
        order[where] = i;
    }
}
#else
inline void order_body(const int*, int*, int, int, int, char*) {}
#endif

// workgroup -> instance of a solve launch: block 0 is the order workgroup when the launch carries one (returns -1 for it)
__device__ __forceinline__ long long dispatch_instance(const DevBatch& D, char* smem) {
    int g = (int)blockIdx.x;
    if (D.order_next) {
        if (g == 0) { order_body(D.cost_in, D.order_next, D.B, D.slots, (int)threadIdx.x, smem); return -1; }
        g -= 1;
    }
    if (D.order) wave_priority_by_rank((unsigned)g, (unsigned)D.B);
    return D.order ? D.order[g] : g;
}

// ---------------------------------------------------------------------------------------------------
// generic batched QP in BackEnd convention
// ---------------------------------------------------------------------------------------------------
struct DevQP {
    int B, n, nc, max_iter;
    double eps_abs;
    const double* H;   // [B][n][n]
    const double* g;   // [B][n]
    const double* A;   // [B][nc][n]
    const double* lA;
    const double* uA;
    const double* l;   // may be null
    const double* u;
    double* x;
    int* status;
    int* iterations;
    int lds_rows_off, lds_rows_cap;
    int* hot;          // [B][LW] hot-start state (HOT instantiation): the inequality working set of the instance's previous solve
                       // (constraint codes, -1 = none), read before the first scan and rewritten at the end; null: cold start
};

// HOT: the instantiation that carries the hot-start code (the batch-of-one BackEnd surface keeps its working set from solve() to
// solve() like QPOasesBackEnd::solve, QPOasesBackEnd.cpp:258-285); the plain one carries none of it
template <int NP, bool HOT = false>
__global__ void __launch_bounds__(64, (NP == 40 ? OSOT_WAVES40 : 1)) osot_qp_kernel(const DevQP Q) {
    OSOT_DYNAMIC_LDS(osot_smem);
    constexpr int S = WaveCtx<NP>::S;
    const int lane = threadIdx.x;
    const long long inst = blockIdx.x;
    const int n = Q.n;
    double* base = reinterpret_cast<double*>(osot_smem);
    WaveCtx<NP> w;
    w.c = WaveCtx<NP>::col_of(lane); w.h = WaveCtx<NP>::half_of(lane); w.n = n;
    w.M1 = base; w.M2 = base + WaveCtx<NP>::M1_DOUBLES; w.V = w.M2 + WaveCtx<NP>::ROWS * S;
    w.rlo = base + Q.lds_rows_off;
    w.rup = w.rlo + Q.lds_rows_cap;
    w.rptr = reinterpret_cast<unsigned long long*>(w.rup + Q.lds_rows_cap);
    w.rowstate = reinterpret_cast<int*>(w.rptr + Q.lds_rows_cap);
    w.eqlist = w.rowstate + Q.lds_rows_cap;
    w.safe_row = reinterpret_cast<unsigned long long>(Q.x + inst * n);
    w.rsrc = reinterpret_cast<signed char*>(w.eqlist + Q.lds_rows_cap);
    const int c = w.c, h = w.h;
    const bool valid = c < n;
    for (int e = lane; e < WaveCtx<NP>::LDS_DOUBLES; e += 64) base[e] = 0.0;
    wave_sync();
    constexpr int HV = WaveCtx<NP>::HV;
    double Hc[NP / HV];
    if constexpr (NP == 32) {   // accumulator-tile layout of factor_tiles32: Hc[4 (2 I + C) + r] = H[16 I + q + 4 r][16 C + a]
        const int ta = lane & 15, tq = lane >> 4;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int i = 16 * (t >> 3) + tq + 4 * (t & 3), cc = 16 * ((t >> 2) & 1) + ta;
            Hc[t] = (i < n && cc < n) ? Q.H[(inst * n + i) * n + cc] + ((i == cc) ? Q.eps_abs : 0.0)
                                      : ((i == cc) ? 1.0 : 0.0);   // unit diagonal beyond n
        }
    } else if constexpr (kWideTiles) {   // the upper triangle of 16 x 16 tiles (factor_tiles_wide): Hc[4 tile_u(I, C) + r] = H[16 I + q + 4 r][16 C + a]
        constexpr int T = wide_tiles(NP);
        const int ta = lane & 15, tq = lane >> 4;
#pragma unroll
        for (int I = 0; I < T; ++I)
#pragma unroll
            for (int C2 = I; C2 < T; ++C2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * I + tq + 4 * r, cc = 16 * C2 + ta;
                    Hc[4 * tile_u<T>(I, C2) + r] = (i < n && cc < n) ? Q.H[(inst * n + i) * n + cc] + ((i == cc) ? Q.eps_abs : 0.0)
                                                                      : ((i == cc) ? 1.0 : 0.0);   // unit diagonal beyond n
                }
    } else {
#pragma unroll
        for (int ii = 0; ii < NP / HV; ++ii) {
            const int i = HV * ii + h;
            Hc[ii] = (valid && i < n) ? Q.H[(inst * n + i) * n + c] + ((i == c) ? Q.eps_abs : 0.0)
                                      : ((i == c) ? 1.0 : 0.0);   // unit diagonal beyond n (see factor_rows64)
        }
    }
    wave_sync();
    const double g = valid ? Q.g[inst * n + c] : 0.0;
    const bool has_box = Q.l != nullptr;
    double lb = (has_box && valid) ? Q.l[inst * n + c] : -INFINITY;
    double ub = (has_box && valid) ? Q.u[inst * n + c] : INFINITY;
    for (int r = lane; r < Q.nc; r += 64) {
        w.rlo[r] = clamp_inf(Q.lA[inst * Q.nc + r]);
        w.rup[r] = clamp_inf(Q.uA[inst * Q.nc + r]);
        w.rptr[r] = reinterpret_cast<unsigned long long>(Q.A + (inst * Q.nc + r) * n);
        w.rsrc[r] = -1;
    }
    wave_sync();
    double x = 0.0, slack = 0.0;
    int iters = 0;
    int st;
    int* hotk = (HOT && Q.hot) ? Q.hot + inst * WaveCtx<NP>::LW : nullptr;
    const int hotcode = hotk ? hotk[c] : -1;
    if (NP > 32) {
        OSOT_ALWAYS_INLINE_CALL st = gi_solve<NP, false>(w, Q.nc, g, false, 0.0, Hc, has_box, lb, ub, Q.max_iter, false, 0.0, x, iters, nullptr, slack,
                                                         false, 0.0, hotcode, hotk);
    } else {
        st = gi_solve<NP, false>(w, Q.nc, g, false, 0.0, Hc, has_box, lb, ub, Q.max_iter, false, 0.0, x, iters, nullptr, slack,
                                 false, 0.0, hotcode, hotk);
    }
    if (valid && h == 0) Q.x[inst * n + c] = (st == QP_SOLVED) ? x : 0.0;
    if (lane == 0) {
        Q.status[inst] = st;
        if (Q.iterations) Q.iterations[inst] = iters;
    }
}

// ---------------------------------------------------------------------------------------------------
// AutoStack::update(): leaf inputs -> b, W diagonal, merged box, constraint rows
// ---------------------------------------------------------------------------------------------------
// The STATIC part of the update (kinds, sizes, gains: from the plan) lives in device memory, uploaded once per solver;
// the per-call part (leaf and output pointers) travels as the kernel argument (kernel arguments are limited to 4 KB).
struct DevTaskS {
    int level, kind, rows, off, prow, body, dense, gains;
    double weight, lambda, ogain, lambda2, sublam;
    unsigned long long mask;   // SubTask: kept rows of the parent (0: whole task); prow = rows of the parent
};
struct DevBoundS { int kind; double scaling, dT; };
struct DevRowBlockS {
    int kind, rows, off, stored_off, first_col, body, ncand;
    double d_threshold, detection_threshold, bound_scaling, dT, p, mu, lambda, ogain;
    double err_lb[OSOT_MAX_BAND_ROWS], err_ub[OSOT_MAX_BAND_ROWS];
};
struct DevUpdatePlan {
    int n, L, nc, nc_stored;
    int m[OSOT_KMAX_LEVELS], ma[OSOT_KMAX_LEVELS];
    int dense_level[OSOT_KMAX_LEVELS];   // level holds a block with a non-diagonal W: W_k A_k and W_k b_k are formed
    int any_dense;                       // some level does
    int total_rows;                      // task rows of all levels + the regularisation task (<= OSOT_KMAX_FLAT_ROWS)
    unsigned char row_task[OSOT_KMAX_FLAT_ROWS];   // flat row -> flat task index
    short row_in_task[OSOT_KMAX_FLAT_ROWS];        // flat row -> row inside its task
    int ntasks;                          // all levels, flat
    DevTaskS task[OSOT_KMAX_FLAT_TASKS];
    int nbounds;
    DevBoundS bound[OSOT_KMAX_BOUNDS];
    int nrowblocks;
    DevRowBlockS rowblock[OSOT_KMAX_ROWBLOCKS];
};
struct DevPtr4 { const double *p0, *p1, *p2, *W; };
struct DevPtr3 { const double *p0, *p1, *p2; };
struct DevUpdate {
    int B;
    const DevUpdatePlan* plan;           // device memory (host memory under the emulation)
    DevPtr4 task[OSOT_KMAX_FLAT_TASKS];
    DevPtr3 bound[OSOT_KMAX_BOUNDS];
    DevPtr3 rows[OSOT_KMAX_ROWBLOCKS];
    double* b[OSOT_KMAX_LEVELS];
    double* w[OSOT_KMAX_LEVELS];
    double* WA[OSOT_KMAX_LEVELS];        // [B][ma_k][n] W_k A_k, [B][m_k] W_k b_k: levels with a dense block only
    double* Wb[OSOT_KMAX_LEVELS];
    const double* A[OSOT_KMAX_LEVELS];   // the stacked Jacobians (read for W_k A_k)
    double* C;
    double* lo;
    double* up;
    double* l;
    double* u;
    double* b_reg;                       // b of the regularisation task (flat task entry with level = -1), [B][rows]
};
static_assert(sizeof(DevUpdate) <= 4096, "kernel arguments are limited to 4 KB");
static_assert(sizeof(DevUpdate) % 16 == 0 && sizeof(DevUpdatePlan) % 16 == 0, "staged into LDS in 16-byte pieces");
// LDS the update needs (bytes): copies of the per-call arguments and of the static plan, then the collision block's
// candidate-ranking scratch (256 doubles, 256 + 2 ints)
constexpr int kUpdateLdsBytes = (int)sizeof(DevUpdate) + (int)sizeof(DevUpdatePlan) + 256 * 8 + 258 * 4 + 8;

// Eigen's Quaterniond(Matrix3d) as invoked by cartesian_utils::computeCartesianError
// (src/utils/cartesian_utils.cpp:83-84); R row-major, q = (x, y, z, w)
__device__ inline void rot_to_quat(const double* R, double* q) {
    double t = R[0] + R[4] + R[8];
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t;
        q[1] = (R[2] - R[6]) * t;
        q[2] = (R[3] - R[1]) * t;
    } else {
        // Eigen picks i = argmax of the diagonal (first wins ties), j = (i+1)%3, k = (j+1)%3.  The three cases are written
        // out with CONSTANT indices: a lane-dependent index into R / q would put both arrays into scratch memory (that was
        // 208 bytes per lane and a dozen memory round trips on the critical path of every wave)
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > ((i == 1) ? R[4] : R[0])) i = 2;
        if (i == 0) {            // j = 1, k = 2
            t = sqrt(R[0] - R[4] - R[8] + 1.0);
            q[0] = 0.5 * t;
            t = 0.5 / t;
            q[3] = (R[7] - R[5]) * t;
            q[1] = (R[3] + R[1]) * t;
            q[2] = (R[6] + R[2]) * t;
        } else if (i == 1) {     // j = 2, k = 0
            t = sqrt(R[4] - R[8] - R[0] + 1.0);
            q[1] = 0.5 * t;
            t = 0.5 / t;
            q[3] = (R[2] - R[6]) * t;
            q[2] = (R[7] + R[5]) * t;
            q[0] = (R[1] + R[3]) * t;
        } else {                 // j = 0, k = 1
            t = sqrt(R[8] - R[0] - R[4] + 1.0);
            q[2] = 0.5 * t;
            t = 0.5 / t;
            q[3] = (R[3] - R[1]) * t;
            q[0] = (R[2] + R[6]) * t;
            q[1] = (R[5] + R[7]) * t;
        }
    }
}

// velocity::Cartesian::update_b (src/tasks/velocity/Cartesian.cpp:279-285) with the quaternion error
// of include/OpenSoT/utils/cartesian_utils.h:144-164; one lane computes the 6 entries
// body: the task has a BODY Jacobian (Cartesian.cpp:93-100): b is rotated by Ad(R') = blockdiag(R', R'), R = actual rotation
__device__ inline void cartesian_b(const double* Ta, const double* Td, const double* twist,
                                   double lambda, double ogain, double* b6, bool body = false) {
    double q[4], qd[4];
    rot_to_quat(Ta, q);
    rot_to_quat(Td, qd);
    const double dot = q[0] * qd[0] + q[1] * qd[1] + q[2] * qd[2] + q[3] * qd[3];
    const double sg = (dot < 0.0) ? -1.0 : 1.0;
    const double qx = sg * q[0], qy = sg * q[1], qz = sg * q[2], qw = sg * q[3];
    double eo[3];
    eo[0] = qd[3] * qx - qw * qd[0] + (-qd[2] * qy + qd[1] * qz);
    eo[1] = qd[3] * qy - qw * qd[1] + (qd[2] * qx - qd[0] * qz);
    eo[2] = qd[3] * qz - qw * qd[2] + (-qd[1] * qx + qd[0] * qy);
    for (int i = 0; i < 3; ++i) {
        const double tw_p = twist ? twist[i] : 0.0, tw_o = twist ? twist[3 + i] : 0.0;
        b6[i] = tw_p + lambda * (Td[9 + i] - Ta[9 + i]);
        b6[3 + i] = tw_o + lambda * (-ogain * eo[i]);
    }
    if (body) {
        double r6[6];
        for (int i = 0; i < 3; ++i) {   // (R'v)_i = sum_k R[k][i] v_k, R row-major in Ta[0..8]
            r6[i] = Ta[i] * b6[0] + Ta[3 + i] * b6[1] + Ta[6 + i] * b6[2];
            r6[3 + i] = Ta[i] * b6[3] + Ta[3 + i] * b6[4] + Ta[6 + i] * b6[5];
        }
        for (int i = 0; i < 6; ++i) b6[i] = r6[i];
    }
}

// AutoStack::update() of one instance by one wavefront.  `args_global` = the kernel's DevUpdate as MEMORY (the kernarg
// segment), `lds` = kUpdateLdsBytes of LDS.  First the per-call pointers and the static plan are staged into LDS by ONE
// batch of 16-byte vector loads (6 per lane): after that every lane looks up its row's task, gains and pointers with
// per-lane LDS reads -- no walk over the task table with one dependent scalar load per entry (that walk, three times over,
// was most of the 14 us a wave spent here), and no lane-indexed access to kernel arguments (which the compiler can only
// serve from a scratch copy).  Dependent chain: stage -> look up -> leaf loads -> arithmetic -> stores.
__device__ __forceinline__ void update_body(const DevUpdate* args_global, const long long inst, const int t, char* lds) {
    DevUpdate* Ul = reinterpret_cast<DevUpdate*>(lds);
    DevUpdatePlan* Pl = reinterpret_cast<DevUpdatePlan*>(lds + sizeof(DevUpdate));
    double* dcand = reinterpret_cast<double*>(lds + sizeof(DevUpdate) + sizeof(DevUpdatePlan));
    int* src_of_row = reinterpret_cast<int*>(dcand + 256);
    int* n_used_s = src_of_row + 256;
    {
        struct alignas(16) v2f64 { double a, b; };     // one 16-byte load / LDS store per piece
        constexpr int NA = (int)sizeof(DevUpdate) / 16, NP_ = (int)sizeof(DevUpdatePlan) / 16;
        const v2f64* ga = reinterpret_cast<const v2f64*>(args_global);
        const v2f64* gp = reinterpret_cast<const v2f64*>(args_global->plan);   // (a scalar kernarg load: both copies are in flight together)
        v2f64* la = reinterpret_cast<v2f64*>(Ul);
        v2f64* lp = reinterpret_cast<v2f64*>(Pl);
        v2f64 va[(NA + 63) / 64], vp[(NP_ + 63) / 64];
#pragma unroll
        for (int i = 0; i < (NA + 63) / 64; ++i) { const int e = t + 64 * i; va[i] = ga[e < NA ? e : 0]; }
#pragma unroll
        for (int i = 0; i < (NP_ + 63) / 64; ++i) { const int e = t + 64 * i; vp[i] = gp[e < NP_ ? e : 0]; }
#pragma unroll
        for (int i = 0; i < (NA + 63) / 64; ++i) { const int e = t + 64 * i; if (e < NA) la[e] = va[i]; }
#pragma unroll
        for (int i = 0; i < (NP_ + 63) / 64; ++i) { const int e = t + 64 * i; if (e < NP_) lp[e] = vp[i]; }
        wave_sync();
    }
    const DevUpdate& U = *Ul;
    const DevUpdatePlan& PL = *Pl;
    const int n = PL.n;
    // ---- tasks: b and diag(W) (tasks::Aggregated::generateAll / generateWeight, Aggregated.cpp:113-132, 265-279).
    // ONE LANE PER ROW of the whole stack (all levels, flat); every kind goes through the SAME four loads (p0[ia], p0[ib],
    // p1[ic], p2[id], each optional), so that all rows of all tasks cost one memory round trip.
    for (int fr = t; fr < PL.total_rows; fr += 64) {
        const int j = PL.row_task[fr], r = PL.row_in_task[fr];
        const DevTaskS& tk = PL.task[j];
        const int kind = tk.kind, level = tk.level, prow = tk.prow;
        const unsigned long long mask = tk.mask;
        const double *p0 = U.task[j].p0, *p1 = U.task[j].p1, *p2 = U.task[j].p2;
        double* bl;
        double* wl = nullptr;
        if (level < 0) bl = U.b_reg + inst * tk.rows + r;   // regularisation task: own output, weight stays in the plan
        else {
            bl = U.b[level] + inst * PL.m[level] + tk.off + r;
            if (U.w[level]) wl = U.w[level] + inst * PL.m[level] + tk.off + r;
        }
        if (wl) *wl = tk.weight;
        if (kind == 1) continue;   // Cartesian rows: below, one lane per task
        // acceleration kinds read (pose error, velocity error) from p0: [2 rows] per instance
        const bool acc = (kind == 4 || kind == 5 || kind == 6);
        // SubTask (SubTask.cpp:22-112): row r of this block is row pr of the parent, whose size the leaf inputs have
        int pr = r;
        if (mask != 0ull) {
            unsigned long long mm = mask;
            for (int q = 0; q < r; ++q) mm &= mm - 1ull;      // drop the r lowest set bits
            pr = __builtin_ctzll(mm);
        }
        const long long base = inst * (long long)prow + pr;
        double x0, x1;
        if (acc && tk.gains) {
            // gain matrices (acceleration/Cartesian.cpp:152-173): p0 = [pose_err; vel_err; Gp; Gd] per instance; this row
            // takes (Gp pose_err)_pr and (Gd vel_err)_pr (Gp = Kp or Mi Kp, see osot_task_desc.acc_gain_matrices)
            const double* e = p0 + inst * (2LL * prow + 2LL * prow * prow);
            const double* Gp = e + 2 * prow + pr * prow;
            const double* Gd = Gp + prow * prow;
            double ap = 0.0, ad = 0.0;
            for (int q = 0; q < prow; ++q) { ap = fma(Gp[q], e[q], ap); ad = fma(Gd[q], e[prow + q], ad); }
            x0 = ap; x1 = ad;
        } else {
            x0 = acc ? p0[inst * 2LL * prow + pr] : p0[base];
            x1 = acc ? p0[inst * 2LL * prow + prow + pr] : 0.0;
        }
        const double x2 = (p1 && kind != 0 && kind != 6) ? p1[base] : 0.0;
        const double x3 = (p2 && kind != 0) ? p2[base] : 0.0;
        const double lam = tk.lambda, lam2 = tk.lambda2;
        double v;
        if (kind == 2 || kind == 3) {            // CoM (CoM.cpp:145-149), Postural (Postural.cpp:97-100)
            v = x3 + lam * (x2 - x0);
        } else if (kind == 4 || kind == 5) {
            // acceleration::Cartesian / CoM (acceleration/Cartesian.cpp:152-160, acceleration/CoM.cpp:86-92):
            // J qddot + Jdot qdot - a_ref - lambda2 Kd vel_err - lambda Kp pose_err = 0, Kp = Kd = I
            v = x3 + lam2 * x1 + lam * x0 - x2;
        } else if (kind == 6) {                  // acceleration::Postural (acceleration/Postural.cpp:145-158)
            v = x3 + lam2 * x1 + lam * x0;
        } else {                                 // Generic: b supplied
            v = x0;
        }
        *bl = (mask != 0ull) ? v * tk.sublam : v;
    }
    // ---- Cartesian tasks, ONE LANE PER TASK (lane j = flat task j): the pose error (Cartesian.cpp:190-240: position
    // difference + quaternion orientation error) is a chain of ~200 dependent fp64 operations; both poses (and the
    // feed-forward twist) are fetched in ONE batch before any arithmetic
    if (t < PL.ntasks && PL.task[t].kind == 1) {
        const DevTaskS& tk = PL.task[t];
        const double *cp0 = U.task[t].p0, *cp1 = U.task[t].p1, *cp2 = U.task[t].p2;
        double* cb = (tk.level < 0) ? U.b_reg + inst * tk.rows      // a Cartesian regularisation task (stored Jacobian)
                                    : U.b[tk.level] + inst * PL.m[tk.level] + tk.off;
        const unsigned cmask = (tk.mask != 0ull) ? ((unsigned)tk.mask & 0x3fu) : 0x3fu;   // e.g. position only
        const double csub = (tk.mask != 0ull) ? tk.sublam : 1.0;
        double Ta[12], Td[12], tw[6], b6[6];
#pragma unroll
        for (int i = 0; i < 12; ++i) { Ta[i] = cp0[inst * 12 + i]; Td[i] = cp1[inst * 12 + i]; }
#pragma unroll
        for (int i = 0; i < 6; ++i) tw[i] = cp2 ? cp2[inst * 6 + i] : 0.0;
        cartesian_b(Ta, Td, tw, tk.lambda, tk.ogain, b6, tk.body != 0);
        int kq = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) if ((cmask >> i) & 1u) { cb[kq] = b6[i] * csub; kq++; }
    }
    // ---- levels with a non-diagonal weight: W_k A_k and W_k b_k with W_k = blockdiag(weight_i W_i) (Task::getWA / getWb,
    // Task.h:273-300; Aggregated::generateWeight, Aggregated.cpp:265-279).  Blocks without a matrix contribute
    // weight_i * rows.  b_k was written above by other lanes of this workgroup: one barrier.
    {
        if (PL.any_dense) {
            workgroup_fence();     // b_k is in global memory: make this workgroup's stores visible to its loads
            __syncthreads();
            for (int j = 0; j < PL.ntasks; ++j) {
                const DevTaskS& tk = PL.task[j];
                if (tk.level < 0 || !PL.dense_level[tk.level]) continue;
                const int k = tk.level, mk = PL.m[k], mak = PL.ma[k];
                const double* bk = U.b[k] + inst * mk + tk.off;
                double* wbk = U.Wb[k] + inst * mk + tk.off;
                const bool stored = tk.off + tk.rows <= mak;          // (an implicit Postural block has no rows in A_k)
                const double* Ak = stored ? U.A[k] + (inst * mak + tk.off) * n : nullptr;
                double* WAk = stored ? U.WA[k] + (inst * mak + tk.off) * n : nullptr;
                const double* Wm = tk.dense ? U.task[j].W + inst * tk.rows * tk.rows : nullptr;
                for (int r = 0; r < tk.rows; ++r) {
                    if (Wm) {
                        double wb = 0.0;
                        for (int q = 0; q < tk.rows; ++q) wb = fma(Wm[r * tk.rows + q], bk[q], wb);
                        if (t == 0) wbk[r] = tk.weight * wb;
                        for (int c0 = t; c0 < n; c0 += 64) {
                            double acc = 0.0;
                            for (int q = 0; q < tk.rows; ++q) acc = fma(Wm[r * tk.rows + q], Ak[q * n + c0], acc);
                            WAk[r * n + c0] = tk.weight * acc;
                        }
                    } else {
                        if (t == 0) wbk[r] = tk.weight * bk[r];
                        if (stored) for (int c0 = t; c0 < n; c0 += 64) WAk[r * n + c0] = tk.weight * Ak[r * n + c0];
                    }
                }
            }
        }
    }
    // ---- box: min/max merge (constraints::Aggregated, Aggregated.cpp:141-148)
    if (PL.nbounds > 0) {
        for (int i = t; i < n; i += 64) {
            double l = 0.0, u = 0.0;
            for (int j = 0; j < PL.nbounds; ++j) {
                const DevBoundS& bd = PL.bound[j];
                const DevPtr3& bp = U.bound[j];
                double l2, u2;
                if (bd.kind == 1) {        // JointLimits.cpp:47-52
                    const double q = bp.p0[inst * n + i];
                    u2 = fmax((bp.p2[inst * n + i] - q) * bd.scaling, 0.0);
                    l2 = fmin((bp.p1[inst * n + i] - q) * bd.scaling, 0.0);
                } else if (bd.kind == 2) { // VelocityLimits.cpp:81-89
                    const double v = fabs(bp.p0[inst * n + i]) * bd.dT;
                    l2 = -1.0 * v; u2 = 1.0 * v;
                } else { l2 = bp.p0[inst * n + i]; u2 = bp.p1[inst * n + i]; }
                if (j == 0) { l = l2; u = u2; } else { u = fmin(u, u2); l = fmax(l, l2); }
            }
            U.l[inst * n + i] = l;
            U.u[inst * n + i] = u;
        }
    }
    // ---- constraint rows
    for (int j = 0; j < PL.nrowblocks; ++j) {
        const DevRowBlockS& rb = PL.rowblock[j];
        const DevPtr3& rp = U.rows[j];
        double* Cb = U.C ? U.C + (inst * PL.nc_stored + rb.stored_off) * n : nullptr;
        double* lob = U.lo + inst * PL.nc + rb.off;
        double* upb = U.up + inst * PL.nc + rb.off;
        if (rb.kind == 1) {   // CollisionAvoidance.cpp:96-152
            const int ncand = rb.ncand > 0 ? rb.ncand : rb.rows;   // pairs supplied; the `rows` closest become rows
            const double* Jd = rp.p0 + inst * ncand * n;
            const double* dist = rp.p1 + inst * ncand;
            // the pairs within the detection threshold, CLOSEST FIRST (getOrderedCollisionPairIndices,
            // CollisionAvoidance.cpp:120-131): lane = candidate, rank = number of candidates that come before it (smaller
            // distance; ties by index), by a walk over the distances staged in LDS.  Up to 256 candidates.
            int& n_used = n_used_s[0];
            for (int i = t; i < ncand; i += 64) {
                const double d = dist[i];
                dcand[i] = (rb.detection_threshold > 0 && d > rb.detection_threshold) ? INFINITY : d;   // skipped pair
            }
            for (int i = t; i < 256; i += 64) src_of_row[i] = -1;
            __syncthreads();
            int mine_used = 0;
            for (int i = t; i < ncand; i += 64) {
                const double d = dcand[i];
                if (d < INFINITY) {
                    int rank = 0;
                    for (int q = 0; q < ncand; ++q) { const double dq = dcand[q]; rank += (dq < d || (dq == d && q < i)) ? 1 : 0; }
                    if (rank < rb.rows) src_of_row[rank] = i;
                    mine_used++;
                }
            }
            (void)mine_used;
            if (t == 0) {   // (uniform walk; the count of pairs inside the detection threshold)
                int cnt = 0;
                for (int q = 0; q < ncand; ++q) cnt += (dcand[q] < INFINITY) ? 1 : 0;
                n_used = cnt;
            }
            __syncthreads();
            const int nu = n_used < rb.rows ? n_used : rb.rows;
            for (int r = 0; r < rb.rows; ++r) {
                const bool used = r < nu;
                const int src = used ? src_of_row[r] : 0;
                for (int i = t; i < n; i += 64) Cb[r * n + i] = used ? -Jd[src * n + i] : 0.0;
                if (t == 0) {
                    if (used) {
                        const double ub = rb.bound_scaling * (dist[src] - rb.d_threshold);
                        upb[r] = ub < 0.0 ? 0.0 : ub;
                    } else upb[r] = 1.7976931348623157e308;
                    lob[r] = -1.7976931348623157e308;
                }
            }
            __syncthreads();
        } else if (rb.kind == 9) {   // generic unit rows (a box on a range of variables as rows): bounds supplied
            for (int r = t; r < rb.rows; r += 64) { lob[r] = rp.p0[inst * rb.rows + r]; upb[r] = rp.p1[inst * rb.rows + r]; }
        } else if (rb.kind == 7) {   // TaskToConstraint(velocity::Cartesian) (TaskToConstraint.cpp:59-68): rows J are in C already
            if (t == 0) {
                double Ta[12], Td[12], tw[6], b6[6];
#pragma unroll
                for (int i = 0; i < 12; ++i) { Ta[i] = rp.p0[inst * 12 + i]; Td[i] = rp.p1[inst * 12 + i]; }
#pragma unroll
                for (int i = 0; i < 6; ++i) tw[i] = rp.p2 ? rp.p2[inst * 6 + i] : 0.0;
                cartesian_b(Ta, Td, tw, rb.lambda, rb.ogain, b6, rb.body != 0);
                for (int i = 0; i < 6; ++i) { lob[i] = b6[i] + rb.err_lb[i]; upb[i] = b6[i] + rb.err_ub[i]; }
            }
        } else if (rb.kind == 8) {   // TaskToConstraint(velocity::CoM): b = v_des + lambda (p_d - p) (CoM.cpp:145-149)
            if (t < 3) {
                const double b = (rp.p2 ? rp.p2[inst * 3 + t] : 0.0) + rb.lambda * (rp.p1[inst * 3 + t] - rp.p0[inst * 3 + t]);
                lob[t] = b + rb.err_lb[t]; upb[t] = b + rb.err_ub[t];
            }
        } else if (rb.kind == 2) {   // DynamicFeasibility.cpp:22-46 as equality: rows [B_u, -J_f'] are in C already
            if (t < 6) { const double v = -rp.p0[inst * 6 + t]; lob[t] = v; upb[t] = v; }
        } else if (rb.kind == 3) {   // TorqueLimits.cpp:25-46: rows [B, -Jc'] are in C already
            for (int r = t; r < rb.rows; r += 64) {
                const double hh = rp.p0[inst * rb.rows + r], tm = rp.p1[inst * rb.rows + r];
                lob[r] = -tm - hh; upb[r] = tm - hh;
            }
        } else if (rb.kind == 4) {   // FrictionCone.cpp:35-56: Ci * wRl', mu/sqrt(2) pyramid, 5 rows per contact
            const int nct = rb.rows / 5;
            const double mu = rb.mu / sqrt(2.0);
            for (int e = t; e < rb.rows * n; e += 64) Cb[e] = 0.0;
            __syncthreads();
            for (int e = t; e < nct * 15; e += 64) {
                const int ct = e / 15, rr = (e % 15) / 3, col = e % 3;
                const double* R = rp.p0 + (inst * nct + ct) * 9;   // wRl row-major; Ci*wRl' (rr,col) = sum_k Ci[rr][k] R[col][k]
                const double ci0 = (rr == 0) ? 1.0 : (rr == 1 ? -1.0 : 0.0);
                const double ci1 = (rr == 2) ? 1.0 : (rr == 3 ? -1.0 : 0.0);
                const double ci2 = (rr == 4) ? -1.0 : -mu;
                Cb[(ct * 5 + rr) * n + rb.first_col + ct * 3 + col] = ci0 * R[col * 3 + 0] + ci1 * R[col * 3 + 1] + ci2 * R[col * 3 + 2];
            }
            for (int r = t; r < rb.rows; r += 64) { lob[r] = -1.0e20; upb[r] = 0.0; }
        } else if (rb.kind == 5) {   // acceleration::JointLimits (constraints/acceleration/JointLimits.cpp:58-176)
            const int nr = rb.rows;
            const double dt = rb.dT * rb.p;
            for (int i = t; i < nr; i += 64) {
                const double q = rp.p0[inst * 2 * nr + i], qd = rp.p0[inst * 2 * nr + nr + i];
                const double qmin = rp.p1[inst * 2 * nr + i], qmax = rp.p1[inst * 2 * nr + nr + i];
                const double am = rp.p2[inst * nr + i];
                const double a = .5 * dt * dt / am;
                const double b_sup = dt * qd / am + .5 * dt * dt;
                const double c_sup = q + dt * qd - qmax + .5 * qd * qd / am;
                double delta_sup = b_sup * b_sup - 4 * a * c_sup;
                const double b_inf = dt * qd / am - .5 * dt * dt;
                const double c_inf = -q - dt * qd + qmin + .5 * qd * qd / am;
                double delta_inf = b_inf * b_inf - 4 * a * c_inf;
                if (delta_sup < 0) delta_sup = 0;
                if (delta_inf < 0) delta_inf = 0;
                const double ub_sup = .5 / a * (-b_sup + sqrt(delta_sup)), lb_sup = .5 / a * (-b_sup - sqrt(delta_sup));
                const double ub_inf = .5 / a * (-b_inf + sqrt(delta_inf)), lb_inf = .5 / a * (-b_inf - sqrt(delta_inf));
                double ub = fmin(ub_sup, ub_inf);
                const double lb = fmax(lb_sup, lb_inf);
                if (ub < lb) ub = lb;
                lob[i] = lb; upb[i] = ub;
            }
        } else if (rb.kind == 6) {   // acceleration::VelocityLimits (constraints/acceleration/VelocityLimits.cpp:50-63)
            for (int i = t; i < rb.rows; i += 64) {
                const double qd = rp.p0[inst * rb.rows + i], lim = rp.p1[inst * rb.rows + i];
                upb[i] = (lim - qd) / (rb.dT * rb.p);
                lob[i] = (-lim - qd) / (rb.dT * rb.p);
            }
        } else {
            for (int e = t; e < rb.rows * n; e += 64) Cb[e] = rp.p0[inst * rb.rows * n + e];
            for (int r = t; r < rb.rows; r += 64) { lob[r] = rp.p1[inst * rb.rows + r]; upb[r] = rp.p2[inst * rb.rows + r]; }
        }
    }
}

__global__ void __launch_bounds__(64) osot_update_kernel(const DevUpdate U) {
    OSOT_STATIC_LDS(double, upd_lds, kUpdateLdsBytes / 8);
    if ((long long)blockIdx.x >= U.B) return;
    update_body(OSOT_KERNARG_PTR(DevUpdate, U), blockIdx.x, threadIdx.x, reinterpret_cast<char*>(upd_lds));
}

// One control cycle in ONE launch: AutoStack::update() and Solver::solve() of an instance by the same wavefront
// (coman_ik.cpp:186-192: `stack->update(); solver->solve(dq)`).  The assembled b / W / box / rows still go through their
// HBM arrays (they are outputs of the update in their own right) but come back from the CU's own L1 / L2 lines; what is
// saved is a launch, its tail and the gap between the two (18 + ~4 us of a 227 us step at BASELINE config 3).
template <int NP, bool EXTRA = false, bool BOX = false>
__global__ void __launch_bounds__(64, (NP == 32 ? OSOT_WAVES32 : (NP == 40 ? OSOT_WAVES40 : 1))) osot_cycle_kernel(const DevUpdate U, const DevPlan P, const DevBatch D) {
    OSOT_DYNAMIC_LDS(osot_smem);
    const long long inst = dispatch_instance(D, osot_smem);
    if (inst < 0) return;
    const long long tc0 = D.prof ? (long long)clock64() : 0;
    const long long tw0 = D.prof ? (long long)wall_clock64() : 0;
    update_body(OSOT_KERNARG_PTR(DevUpdate, U), inst, (int)threadIdx.x, osot_smem);   // (the cascade's slice is idle until it starts)
    workgroup_fence();      // the update's global stores are visible to the cascade's loads (same workgroup)
    __syncthreads();
    const long long tc1 = D.prof ? (long long)clock64() : 0;
    cascade_body<NP, false, EXTRA, BOX>(P, D, inst, (int)threadIdx.x, osot_smem);
    if (D.prof && threadIdx.x == 0) {   // diagnostic (osot_solver_profile_cycle): shader-clock cycles of the two halves
        D.prof[inst * 4] = tc1 - tc0;
        D.prof[inst * 4 + 1] = (long long)clock64() - tc1;
        D.prof[inst * 4 + 2] = tw0;                          // constant-rate (100 MHz) timestamps: the launch's timeline
        D.prof[inst * 4 + 3] = (long long)wall_clock64();
    }
}

}  // namespace osot

// osot_host_plan.h -- host-side translation of the C-ABI plan into kernel arguments (no HIP calls).
#pragma once
#include <cstddef>
#include <cstring>
#include "../../include/osot_mi355x.h"
#include "osot_kernels.h"

namespace osot {

// a Postural block has A = [I 0] (Postural.cpp:37): implicit, never stored -- unless it is a SubTask of one
inline bool task_is_implicit(const osot_task_desc& t) {
    return (t.kind == OSOT_TASK_POSTURAL || t.kind == OSOT_TASK_ACC_POSTURAL) && t.row_mask == 0ull && !t.dense_weight;
}
// rows of the parent of a sub-task (the kind's own size unless given)
inline int task_parent_rows(const osot_task_desc& t, int n) {
    if (t.row_mask == 0ull) return t.rows;
    if (t.parent_rows > 0) return t.parent_rows;
    switch (t.kind) {
        case OSOT_TASK_CARTESIAN: case OSOT_TASK_ACC_CARTESIAN: return 6;
        case OSOT_TASK_COM: case OSOT_TASK_ACC_COM: return 3;
        case OSOT_TASK_POSTURAL: case OSOT_TASK_ACC_POSTURAL: return n;
        default: return 0;
    }
}

inline int plan_level_rows(const osot_plan_desc* p, int k, int* m_total, int* m_stored) {
    if (!p || k < 0 || k >= p->n_levels) return OSOT_ERR_INVALID;
    int m = 0, ma = 0;
    const osot_level_desc& lv = p->level[k];
    for (int j = 0; j < lv.n_tasks; ++j) {
        m += lv.task[j].rows;
        if (!task_is_implicit(lv.task[j])) ma += lv.task[j].rows;
    }
    if (m_total) *m_total = m;
    if (m_stored) *m_stored = ma;
    return OSOT_OK;
}

inline int plan_constraint_rows(const osot_plan_desc* p, int* nc) {
    if (!p) return OSOT_ERR_INVALID;
    int s = 0;
    for (int j = 0; j < p->n_rowblocks; ++j) s += p->rowblock[j].rows;
    if (nc) *nc = s;
    return OSOT_OK;
}

inline bool rows_are_implicit(int kind) {
    return kind == OSOT_ROWS_ACC_JOINT_LIMITS || kind == OSOT_ROWS_ACC_VELOCITY_LIMITS || kind == OSOT_ROWS_UNIT_GENERIC;
}
// every constraint row of the plan is an EQUALITY by construction: TaskToConstraint blocks (`stack << l_sole`,
// TaskToConstraint.cpp:34-52) whose error band is a point -- the update writes lo = b + err_lb and up = b + err_ub, bit-equal
// then.  Such rows live in the equality phase of every level; the bounds are the only inequalities, which is what the BOX
// instantiation of the kernels assumes (a plan without rows is the trivial case).
inline bool plan_rows_all_equalities(const osot_plan_desc& p) {
    for (int j = 0; j < p.n_rowblocks; ++j) {
        const osot_rows_desc& rb = p.rowblock[j];
        if (rb.kind != OSOT_ROWS_TASK_CARTESIAN && rb.kind != OSOT_ROWS_TASK_COM) return false;
        for (int i = 0; i < rb.rows && i < OSOT_MAX_BAND_ROWS; ++i)
            if (!(rb.err_lb[i] == rb.err_ub[i])) return false;
    }
    return true;
}
inline int plan_stored_constraint_rows(const osot_plan_desc* p, int* nc_stored) {
    if (!p) return OSOT_ERR_INVALID;
    int s = 0;
    for (int j = 0; j < p->n_rowblocks; ++j) if (!rows_are_implicit(p->rowblock[j].kind)) s += p->rowblock[j].rows;
    if (nc_stored) *nc_stored = s;
    return OSOT_OK;
}

inline int plan_validate(const osot_plan_desc* p, const char** why) {
    static const char* ok = "";
    *why = ok;
    if (!p) { *why = "null plan"; return OSOT_ERR_INVALID; }
    if (p->n < 1 || p->n > OSOT_MAX_VARS) { *why = "n out of range (1..64)"; return OSOT_ERR_INVALID; }
    if (p->n_levels < 1 || p->n_levels > OSOT_MAX_LEVELS) { *why = "n_levels out of range"; return OSOT_ERR_INVALID; }
    if (p->n_bounds < 0 || p->n_bounds > OSOT_MAX_BOUNDS) { *why = "n_bounds out of range"; return OSOT_ERR_INVALID; }
    if (p->n_rowblocks < 0 || p->n_rowblocks > OSOT_MAX_ROWBLOCKS) { *why = "n_rowblocks out of range"; return OSOT_ERR_INVALID; }
    if (!(p->eps_abs >= 0.0)) { *why = "negative eps"; return OSOT_ERR_INVALID; }
    int flat = 0;
    for (int k = 0; k < p->n_levels; ++k) {
        const osot_level_desc& lv = p->level[k];
        if (lv.n_tasks < 1 || lv.n_tasks > OSOT_MAX_TASKS) { *why = "n_tasks out of range"; return OSOT_ERR_INVALID; }
        flat += lv.n_tasks;
        for (int j = 0; j < lv.n_tasks; ++j) {
            const osot_task_desc& t = lv.task[j];
            if (t.rows < 1) { *why = "task with no rows"; return OSOT_ERR_INVALID; }
            if (t.row_mask != 0ull) {   // SubTask: rows = kept rows, all inside the parent
                const int pr = task_parent_rows(t, p->n);
                if (pr < 1 || pr > 64) { *why = "sub-task: parent rows out of range (1..64)"; return OSOT_ERR_INVALID; }
                if (__builtin_popcountll(t.row_mask) != t.rows) { *why = "sub-task: rows != popcount(row_mask)"; return OSOT_ERR_INVALID; }
                if (pr < 64 && (t.row_mask >> pr) != 0ull) { *why = "sub-task: row_mask selects rows beyond the parent"; return OSOT_ERR_INVALID; }
                if ((t.kind == OSOT_TASK_CARTESIAN || t.kind == OSOT_TASK_ACC_CARTESIAN) && pr != 6) { *why = "Cartesian parent has 6 rows"; return OSOT_ERR_INVALID; }
                if ((t.kind == OSOT_TASK_COM || t.kind == OSOT_TASK_ACC_COM) && pr != 3) { *why = "CoM parent has 3 rows"; return OSOT_ERR_INVALID; }
                if (t.kind < OSOT_TASK_GENERIC || t.kind > OSOT_TASK_ACC_POSTURAL) { *why = "unknown task kind"; return OSOT_ERR_UNSUPPORTED; }
                if (!(t.weight >= 0.0)) { *why = "negative task weight"; return OSOT_ERR_INVALID; }
                continue;
            }
            switch (t.kind) {
                case OSOT_TASK_GENERIC: break;
                case OSOT_TASK_CARTESIAN: case OSOT_TASK_ACC_CARTESIAN:
                    if (t.rows != 6) { *why = "Cartesian task must have 6 rows"; return OSOT_ERR_INVALID; } break;
                case OSOT_TASK_COM: case OSOT_TASK_ACC_COM:
                    if (t.rows != 3) { *why = "CoM task must have 3 rows"; return OSOT_ERR_INVALID; } break;
                case OSOT_TASK_POSTURAL: case OSOT_TASK_ACC_POSTURAL:
                    if (t.rows > p->n) { *why = "Postural task cannot have more than n rows"; return OSOT_ERR_INVALID; }
                    if (task_is_implicit(t) && j != lv.n_tasks - 1) { *why = "an implicit Postural block must be the last block of its level"; return OSOT_ERR_UNSUPPORTED; }
                    break;
                default: *why = "unknown task kind"; return OSOT_ERR_UNSUPPORTED;
            }
            if (!(t.weight >= 0.0)) { *why = "negative task weight"; return OSOT_ERR_INVALID; }
            if (t.body_frame && t.kind != OSOT_TASK_CARTESIAN) { *why = "body_frame is an option of velocity::Cartesian"; return OSOT_ERR_INVALID; }
            if (t.dense_weight && t.rows > 64) { *why = "dense weight: at most 64 rows per block"; return OSOT_ERR_INVALID; }
            if (t.acc_gain_matrices && t.kind != OSOT_TASK_ACC_CARTESIAN && t.kind != OSOT_TASK_ACC_COM) {
                *why = "gain matrices are an option of the acceleration Cartesian / CoM tasks"; return OSOT_ERR_INVALID; }
        }
    }
    if (p->has_regularisation) {   // identity-Jacobian regularisation only: Hr is folded into the diagonal
        const osot_task_desc& t = p->regularisation;
        if (p->regularisation_dense) {   // stored Jacobian A_r (osot_qp_batch.A_reg): any kind whose b the update forms without A
            if (t.kind != OSOT_TASK_GENERIC && t.kind != OSOT_TASK_CARTESIAN && t.kind != OSOT_TASK_COM) {
                *why = "regularisation task with a stored Jacobian: kinds GENERIC, CARTESIAN, COM"; return OSOT_ERR_UNSUPPORTED; }
            if (t.rows < 1 || t.rows > 64) { *why = "regularisation task: rows out of range (1..64)"; return OSOT_ERR_INVALID; }
        } else {
            if (t.kind != OSOT_TASK_GENERIC && t.kind != OSOT_TASK_POSTURAL && t.kind != OSOT_TASK_ACC_POSTURAL) {
                *why = "regularisation task without a stored Jacobian: identity-Jacobian kinds (generic b with A = [I 0], Postural)"; return OSOT_ERR_UNSUPPORTED; }
            if (t.rows < 1 || t.rows > p->n) { *why = "regularisation task: rows out of range (1..n)"; return OSOT_ERR_INVALID; }
        }
        if (t.row_mask != 0ull) { *why = "regularisation task cannot be a sub-task"; return OSOT_ERR_UNSUPPORTED; }
        if (t.dense_weight || t.body_frame) { *why = "regularisation task: scalar weight, no frame option"; return OSOT_ERR_UNSUPPORTED; }
        if (!(t.weight >= 0.0)) { *why = "negative task weight"; return OSOT_ERR_INVALID; }
        flat += 1;
    }
    if (flat > OSOT_KMAX_FLAT_TASKS) { *why = "too many leaf tasks in total"; return OSOT_ERR_UNSUPPORTED; }
    {
        int rows_total = p->has_regularisation ? p->regularisation.rows : 0;
        for (int k = 0; k < p->n_levels; ++k) for (int j = 0; j < p->level[k].n_tasks; ++j) rows_total += p->level[k].task[j].rows;
        if (rows_total > OSOT_KMAX_FLAT_ROWS) { *why = "more than 256 task rows in all levels together"; return OSOT_ERR_UNSUPPORTED; }
    }
    for (int j = 0; j < p->n_bounds; ++j)
        if (p->bound[j].kind < 0 || p->bound[j].kind > OSOT_BOUND_VELOCITY_LIMITS) { *why = "unknown bound kind"; return OSOT_ERR_UNSUPPORTED; }
    for (int j = 0; j < p->n_rowblocks; ++j) {
        const osot_rows_desc& rb = p->rowblock[j];
        if (rb.kind < 0 || rb.kind > OSOT_ROWS_UNIT_GENERIC) { *why = "unknown row-block kind"; return OSOT_ERR_UNSUPPORTED; }
        if (rb.kind == OSOT_ROWS_TASK_CARTESIAN && rb.rows != 6) { *why = "a Cartesian task as a constraint has 6 rows"; return OSOT_ERR_INVALID; }
        if (rb.kind == OSOT_ROWS_TASK_COM && rb.rows != 3) { *why = "a CoM task as a constraint has 3 rows"; return OSOT_ERR_INVALID; }
        if (rb.kind == OSOT_ROWS_TASK_CARTESIAN || rb.kind == OSOT_ROWS_TASK_COM)
            for (int i = 0; i < rb.rows; ++i) if (!(rb.err_ub[i] >= rb.err_lb[i])) {
                *why = "Some components of err_ub are smaller than err_lb!!!"; return OSOT_ERR_INVALID; }   // TaskToConstraint.cpp:43
        if (rb.kind == OSOT_ROWS_COLLISION && (rb.n_candidates < 0 || rb.n_candidates > 256 || (rb.n_candidates > 0 && rb.n_candidates < rb.rows))) {
            *why = "collision block: n_candidates must be 0 or in rows..256"; return OSOT_ERR_INVALID; }
        if (rb.rows < 1 || rb.rows > 256) { *why = "row block size out of range (1..256)"; return OSOT_ERR_INVALID; }
        if (rb.kind == OSOT_ROWS_DYN_FEASIBILITY && rb.rows != 6) { *why = "DynamicFeasibility has 6 rows"; return OSOT_ERR_INVALID; }
        if (rb.kind == OSOT_ROWS_FRICTION_CONE && (rb.rows % 5 != 0 || rb.first_col < 0 || rb.first_col + 3 * (rb.rows / 5) > p->n)) {
            *why = "friction cone block: rows = 5*contacts and 3 force columns per contact inside x"; return OSOT_ERR_INVALID; }
        if (rows_are_implicit(rb.kind) && (rb.first_col < 0 || rb.first_col + rb.rows > p->n)) {
            *why = "unit-row block exceeds the variables"; return OSOT_ERR_INVALID; }
        if (rows_are_implicit(rb.kind) && rb.kind != OSOT_ROWS_UNIT_GENERIC && !(rb.dT * rb.p > 0.0)) { *why = "acceleration limits need dT*p > 0"; return OSOT_ERR_INVALID; }
        if (rb.only_level < 0 || rb.only_level > p->n_levels) { *why = "row block: only_level out of range (0..n_levels)"; return OSOT_ERR_INVALID; }
    }
    return OSOT_OK;
}

// LDS carve-up (doubles) of one wave's slice for padded size NP: M1, M2, V, then the row table
// (rlo, rup, rptr: 8 B per row each; rowstate, eqlist: 4 B per row each; rsrc: 1 B per row).  Returns the total in doubles.
inline int lds_layout(int NP, int n_rows, int* rows_off, int* rows_cap) {
    // (round 5: the phantom-lane layouts 40 and 56 too -- osot_qp_kernel<40> / <56> for the plugin route and nHQP's level QPs)
    int d = NP == 32 ? WaveCtx<32>::LDS_DOUBLES : (NP == 40 ? WaveCtx<40>::LDS_DOUBLES : (NP == 56 ? WaveCtx<56>::LDS_DOUBLES : WaveCtx<64>::LDS_DOUBLES));
    d = (d + 1) & ~1;
    *rows_off = d;
    const int cap = ((n_rows > 0 ? n_rows : 1) + 1) & ~1;
    *rows_cap = cap;
    d += 3 * cap + cap + (cap + 7) / 8;   // rlo, rup, rptr + (rowstate, eqlist as ints) + rsrc bytes
    d = (d + 1) & ~1;
    return d;
}

// returns OSOT_OK and fills P, the padded size NP (32/64) and the dynamic LDS bytes per workgroup (= wave)
// task_active: [OSOT_MAX_LEVELS][OSOT_MAX_TASKS] flags (Task::setActive), null = all active
inline int make_dev_plan(const osot_plan_desc& p, const unsigned char* level_active, DevPlan& P, int& NP,
                         size_t& lds_bytes, const unsigned char* task_active = nullptr) {
    std::memset(&P, 0, sizeof(P));
    P.n = p.n;
    P.L = p.n_levels;
    plan_constraint_rows(&p, &P.nc);
    P.optoff[0] = 0;
    P.active_mask = 0;
    for (int k = 0; k < p.n_levels; ++k) {
        plan_level_rows(&p, k, &P.m[k], &P.ma[k]);
        P.optoff[k + 1] = P.optoff[k] + P.m[k];
        if (!level_active || level_active[k]) P.active_mask |= (1u << k);
    }
    P.nblocks = p.n_rowblocks;
    {
        int off = 0, soff = 0;
        for (int j = 0; j < p.n_rowblocks; ++j) {
            P.blk_rows[j] = p.rowblock[j].rows;
            P.blk_off[j] = off;
            P.blk_implicit[j] = rows_are_implicit(p.rowblock[j].kind) ? 1 : 0;
            P.blk_first_col[j] = p.rowblock[j].first_col;
            P.blk_level[j] = p.rowblock[j].only_level;
            P.blk_stored_off[j] = soff;
            off += p.rowblock[j].rows;
            if (!P.blk_implicit[j]) soff += p.rowblock[j].rows;
        }
        P.nc_stored = soff;
    }
    for (int k = 0; k < p.n_levels; ++k) P.ident_rows[k] = P.m[k] - P.ma[k];
    for (int k = 0; k < p.n_levels; ++k) {
        P.ntask[k] = p.level[k].n_tasks;
        int off = 0;
        for (int j = 0; j < p.level[k].n_tasks; ++j) {
            P.task_off[k][j] = off;
            off += p.level[k].task[j].rows;
            if (task_active && !task_active[k * OSOT_MAX_TASKS + j]) P.inactive[k] |= (1u << j);
        }
        P.task_off[k][p.level[k].n_tasks] = off;
    }
    const int nrows_max = P.nc + P.optoff[p.n_levels];
    P.max_iter = p.max_iter > 0 ? p.max_iter : 20 * (p.n + nrows_max) + 100;
    P.eps_abs = p.eps_abs;
    P.reg_rows = p.has_regularisation ? p.regularisation.rows : 0;
    P.reg_w = p.has_regularisation ? p.regularisation.weight : 0.0;
    P.reg_dense = (p.has_regularisation && p.regularisation_dense) ? 1 : 0;
    // 56: the 64-lane solver with the LDS of n <= 54, four wavefronts per CU instead of three (osot_qp_core.h, WaveCtx)
    // 40 (round 5): the same for n <= 38 (the reference's 35-coordinate COMAN) -- two wavefronts per SIMD
#ifdef OSOT_X_NO_NP40   // developer knob (A/B builds)
    NP = (p.n <= 32) ? 32 : ((p.n <= WaveCtx<56>::NMAX) ? 56 : 64);
#else
    NP = (p.n <= 32) ? 32 : ((p.n <= WaveCtx<40>::NMAX) ? 40 : ((p.n <= WaveCtx<56>::NMAX) ? 56 : 64));
#endif
    // cascade layout: M1, M2, V | rlo, rup, rptr | rowstate, eqlist | rsrc bytes
    const int cap = ((nrows_max > 0 ? nrows_max : 1) + 1) & ~1;
    int total = ((NP == 32 ? WaveCtx<32>::LDS_DOUBLES : (NP == 40 ? WaveCtx<40>::LDS_DOUBLES : (NP == 56 ? WaveCtx<56>::LDS_DOUBLES : WaveCtx<64>::LDS_DOUBLES))) + 1) & ~1;
    P.lds_rows_off = total;
    P.lds_rows_cap = cap;
    const int table = 3 * cap + cap + (cap + 7) / 8;      // doubles of the row table
    P.rows_doubles = table;
    // NP = 64 (round 3): the row table lives in a per-instance slice of device memory (DevBatch.rows_scratch, served by the
    // CU's L1 / L2) whenever taking it out of LDS buys another resident wavefront per CU
    P.rows_in_global = 0;
    if (NP > 32) {
        const size_t with = (size_t)(total + table) * sizeof(double), without = (size_t)total * sizeof(double);
        if ((160 * 1024) / without > (160 * 1024) / with) P.rows_in_global = 1;
    }
    if (!P.rows_in_global) total += table;
    lds_bytes = (size_t)total * sizeof(double);
    return OSOT_OK;
}

// the static part of the update kernel's arguments (uploaded once per solver)
inline void make_update_plan(const osot_plan_desc& pl, DevUpdatePlan& U) {
    std::memset(&U, 0, sizeof(U));
    U.n = pl.n; U.L = pl.n_levels;
    plan_constraint_rows(&pl, &U.nc);
    plan_stored_constraint_rows(&pl, &U.nc_stored);
    int flat = 0;
    for (int k = 0; k < pl.n_levels; ++k) {
        plan_level_rows(&pl, k, &U.m[k], &U.ma[k]);
        int off = 0;
        for (int j = 0; j < pl.level[k].n_tasks; ++j) {
            const osot_task_desc& t = pl.level[k].task[j];
            DevTaskS& d = U.task[flat++];
            d.level = k; d.kind = t.kind; d.rows = t.rows; d.off = off;
            d.weight = t.weight; d.lambda = t.lambda; d.ogain = t.orientation_gain; d.lambda2 = t.lambda2;
            d.mask = t.row_mask; d.prow = task_parent_rows(t, pl.n); d.sublam = t.row_mask ? t.sub_lambda : 1.0;
            d.gains = t.acc_gain_matrices ? 1 : 0;
            d.body = t.body_frame; d.dense = t.dense_weight;
            if (t.dense_weight) U.dense_level[k] = 1;
            off += t.rows;
        }
    }
    if (pl.has_regularisation) {   // one more flat entry; its b goes to out->b_reg (level = -1)
        const osot_task_desc& t = pl.regularisation;
        DevTaskS& d = U.task[flat++];
        d.level = -1; d.kind = t.kind; d.rows = t.rows; d.off = 0;
        d.weight = t.weight; d.lambda = t.lambda; d.ogain = t.orientation_gain; d.lambda2 = t.lambda2;
        d.mask = 0ull; d.prow = t.rows; d.sublam = 1.0;
        d.gains = 0; d.body = 0; d.dense = 0;
    }
    U.ntasks = flat;
    {
        int fr = 0;
        for (int j = 0; j < flat; ++j)
            for (int r = 0; r < U.task[j].rows && fr < OSOT_KMAX_FLAT_ROWS; ++r, ++fr) { U.row_task[fr] = (unsigned char)j; U.row_in_task[fr] = (short)r; }
        U.total_rows = fr;
        for (int k = 0; k < pl.n_levels; ++k) if (U.dense_level[k]) U.any_dense = 1;
    }
    U.nbounds = pl.n_bounds;
    for (int j = 0; j < pl.n_bounds; ++j) {
        U.bound[j].kind = pl.bound[j].kind; U.bound[j].scaling = pl.bound[j].scaling; U.bound[j].dT = pl.bound[j].dT;
    }
    U.nrowblocks = pl.n_rowblocks;
    int roff = 0, soff = 0;
    for (int j = 0; j < pl.n_rowblocks; ++j) {
        const osot_rows_desc& r = pl.rowblock[j];
        DevRowBlockS& d = U.rowblock[j];
        d.kind = r.kind; d.rows = r.rows; d.off = roff; d.stored_off = soff; d.first_col = r.first_col;
        d.body = r.task_body_frame; d.ncand = r.n_candidates;
        d.dT = r.dT; d.p = r.p; d.mu = r.mu; d.lambda = r.task_lambda; d.ogain = r.task_orientation_gain;
        for (int i = 0; i < OSOT_MAX_BAND_ROWS; ++i) { d.err_lb[i] = r.err_lb[i]; d.err_ub[i] = r.err_ub[i]; }
        d.d_threshold = r.d_threshold; d.detection_threshold = r.detection_threshold; d.bound_scaling = r.bound_scaling;
        if (!rows_are_implicit(d.kind)) soff += d.rows;
        roff += d.rows;
    }
}

// the per-call part of the update kernel's arguments: leaf / output pointers, checked against the plan
inline int make_update_args(const osot_plan_desc& pl, const DevUpdatePlan& PL, const osot_leaf_batch* leaf,
                            const osot_assembled_out* out, const DevUpdatePlan* dev_plan, DevUpdate& U, const char** why) {
    std::memset(&U, 0, sizeof(U));
    U.B = leaf->B;
    U.plan = dev_plan;
    int flat = 0;
    for (int k = 0; k < pl.n_levels; ++k) {
        if (!out->b[k]) { *why = "out.b[k] is null"; return OSOT_ERR_INVALID; }
        U.b[k] = out->b[k]; U.w[k] = out->w[k];
        if (PL.dense_level[k]) {
            if (!out->WA[k] || !out->Wb[k] || (PL.ma[k] > 0 && !out->A[k]))
                { *why = "level has a non-diagonal weight: out.WA[k], out.Wb[k] and out.A[k] are required"; return OSOT_ERR_INVALID; }
            U.WA[k] = out->WA[k]; U.Wb[k] = out->Wb[k]; U.A[k] = out->A[k];
        }
        for (int j = 0; j < pl.level[k].n_tasks; ++j) {
            const osot_task_desc& t = pl.level[k].task[j];
            DevPtr4& d = U.task[flat++];
            d.p0 = leaf->task[k][j].p0; d.p1 = leaf->task[k][j].p1; d.p2 = leaf->task[k][j].p2; d.W = leaf->task[k][j].W;
            if (!d.p0) { *why = "leaf input p0 of a task is null"; return OSOT_ERR_INVALID; }
            if (t.kind != OSOT_TASK_GENERIC && t.kind != OSOT_TASK_ACC_POSTURAL && !d.p1)
                { *why = "leaf input p1 of a task is null"; return OSOT_ERR_INVALID; }
            if (t.dense_weight && !d.W) { *why = "task has dense_weight but its leaf W is null"; return OSOT_ERR_INVALID; }
        }
    }
    if (pl.has_regularisation) {   // one more flat entry; its b goes to out->b_reg (level = -1)
        const osot_task_desc& t = pl.regularisation;
        if (!out->b_reg) { *why = "out.b_reg is null"; return OSOT_ERR_INVALID; }
        DevPtr4& d = U.task[flat++];
        d.p0 = leaf->regularisation.p0; d.p1 = leaf->regularisation.p1; d.p2 = leaf->regularisation.p2; d.W = nullptr;
        if (!d.p0) { *why = "leaf input p0 of the regularisation task is null"; return OSOT_ERR_INVALID; }
        if (t.kind == OSOT_TASK_POSTURAL && !d.p1) { *why = "leaf input p1 of the regularisation task is null"; return OSOT_ERR_INVALID; }
        U.b_reg = out->b_reg;
    }
    for (int j = 0; j < pl.n_bounds; ++j) {
        DevPtr3& d = U.bound[j];
        const int kind = pl.bound[j].kind;
        d.p0 = leaf->bound[j].p0; d.p1 = leaf->bound[j].p1; d.p2 = leaf->bound[j].p2;
        if (!d.p0) { *why = "leaf input p0 of a bound is null"; return OSOT_ERR_INVALID; }
        if (kind == OSOT_BOUND_JOINT_LIMITS && (!d.p1 || !d.p2)) { *why = "joint limits need q, q_min, q_max"; return OSOT_ERR_INVALID; }
        if (kind == OSOT_BOUND_GENERIC && !d.p1) { *why = "generic bound needs l and u"; return OSOT_ERR_INVALID; }
    }
    if (pl.n_bounds > 0 && (!out->l || !out->u)) { *why = "out.l/out.u is null"; return OSOT_ERR_INVALID; }
    U.l = out->l; U.u = out->u;
    for (int j = 0; j < pl.n_rowblocks; ++j) {
        DevPtr3& d = U.rows[j];
        const int kind = pl.rowblock[j].kind;
        d.p0 = leaf->rows[j].p0; d.p1 = leaf->rows[j].p1; d.p2 = leaf->rows[j].p2;
        if (!d.p0) { *why = "leaf input p0 of a row block is null"; return OSOT_ERR_INVALID; }
        if ((kind == OSOT_ROWS_GENERIC || kind == OSOT_ROWS_COLLISION || kind == OSOT_ROWS_TORQUE_LIMITS ||
             kind == OSOT_ROWS_ACC_JOINT_LIMITS || kind == OSOT_ROWS_ACC_VELOCITY_LIMITS ||
             kind == OSOT_ROWS_TASK_CARTESIAN || kind == OSOT_ROWS_TASK_COM || kind == OSOT_ROWS_UNIT_GENERIC) && !d.p1)
            { *why = "leaf input p1 of a row block is null"; return OSOT_ERR_INVALID; }
        if (kind == OSOT_ROWS_ACC_JOINT_LIMITS && !d.p2) { *why = "acceleration joint limits need qddot_max"; return OSOT_ERR_INVALID; }
        if (kind == OSOT_ROWS_GENERIC && !d.p2) { *why = "generic rows need C, lo, up"; return OSOT_ERR_INVALID; }
    }
    if (PL.nc > 0 && (!out->lo || !out->up)) { *why = "out.lo/up is null"; return OSOT_ERR_INVALID; }
    if (PL.nc_stored > 0 && !out->C) { *why = "out.C is null"; return OSOT_ERR_INVALID; }
    U.C = out->C; U.lo = out->lo; U.up = out->up;
    return OSOT_OK;
}

}  // namespace osot

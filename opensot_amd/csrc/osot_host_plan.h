// osot_host_plan.h -- host-side translation of the C-ABI plan into kernel arguments (no HIP calls).
#pragma once
#include <cstddef>
#include <cstring>
#include "../../include/osot_mi355x.h"
#include "osot_kernels.h"

namespace osot {

// a Postural block has A = [I 0] (Postural.cpp:37): implicit, never stored -- unless it is a SubTask of one
inline bool task_is_implicit(const osot_task_desc& t) {
    return (t.kind == OSOT_TASK_POSTURAL || t.kind == OSOT_TASK_ACC_POSTURAL) && t.row_mask == 0ull;
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
                    if (j != lv.n_tasks - 1) { *why = "Postural must be the last block of its level"; return OSOT_ERR_UNSUPPORTED; }
                    break;
                default: *why = "unknown task kind"; return OSOT_ERR_UNSUPPORTED;
            }
            if (!(t.weight >= 0.0)) { *why = "negative task weight"; return OSOT_ERR_INVALID; }
        }
    }
    if (p->has_regularisation) {   // identity-Jacobian regularisation only: Hr is folded into the diagonal
        const osot_task_desc& t = p->regularisation;
        if (t.kind != OSOT_TASK_GENERIC && t.kind != OSOT_TASK_POSTURAL && t.kind != OSOT_TASK_ACC_POSTURAL) {
            *why = "regularisation task: only identity-Jacobian kinds (generic b with A = [I 0], Postural) are supported"; return OSOT_ERR_UNSUPPORTED; }
        if (t.row_mask != 0ull) { *why = "regularisation task cannot be a sub-task"; return OSOT_ERR_UNSUPPORTED; }
        if (t.rows < 1 || t.rows > p->n) { *why = "regularisation task: rows out of range (1..n)"; return OSOT_ERR_INVALID; }
        if (!(t.weight >= 0.0)) { *why = "negative task weight"; return OSOT_ERR_INVALID; }
        flat += 1;
    }
    if (flat > OSOT_KMAX_FLAT_TASKS) { *why = "too many leaf tasks in total"; return OSOT_ERR_UNSUPPORTED; }
    for (int j = 0; j < p->n_bounds; ++j)
        if (p->bound[j].kind < 0 || p->bound[j].kind > OSOT_BOUND_VELOCITY_LIMITS) { *why = "unknown bound kind"; return OSOT_ERR_UNSUPPORTED; }
    for (int j = 0; j < p->n_rowblocks; ++j) {
        const osot_rows_desc& rb = p->rowblock[j];
        if (rb.kind < 0 || rb.kind > OSOT_ROWS_UNIT_GENERIC) { *why = "unknown row-block kind"; return OSOT_ERR_UNSUPPORTED; }
        if (rb.kind == OSOT_ROWS_TASK_CARTESIAN && rb.rows != 6) { *why = "a Cartesian task as a constraint has 6 rows"; return OSOT_ERR_INVALID; }
        if (rb.kind == OSOT_ROWS_TASK_COM && rb.rows != 3) { *why = "a CoM task as a constraint has 3 rows"; return OSOT_ERR_INVALID; }
        if ((rb.kind == OSOT_ROWS_TASK_CARTESIAN || rb.kind == OSOT_ROWS_TASK_COM) && !(rb.err_ub >= rb.err_lb)) {
            *why = "Some components of err_ub are smaller than err_lb!!!"; return OSOT_ERR_INVALID; }   // TaskToConstraint.cpp:43
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
    const int S = NP + 1;
    int d = 2 * NP * S + 4 * NP;
    d = (d + 1) & ~1;
    *rows_off = d;
    const int cap = ((n_rows > 0 ? n_rows : 1) + 1) & ~1;
    *rows_cap = cap;
    d += 3 * cap + cap + (cap + 7) / 8;   // rlo, rup, rptr + (rowstate, eqlist as ints) + rsrc bytes
    d = (d + 1) & ~1;
    return d;
}

// returns OSOT_OK and fills P, the padded size NP (32/64) and the dynamic LDS bytes per workgroup (= wave)
inline int make_dev_plan(const osot_plan_desc& p, const unsigned char* level_active, DevPlan& P, int& NP,
                         size_t& lds_bytes) {
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
    const int nrows_max = P.nc + P.optoff[p.n_levels];
    P.max_iter = p.max_iter > 0 ? p.max_iter : 20 * (p.n + nrows_max) + 100;
    P.eps_abs = p.eps_abs;
    P.reg_rows = p.has_regularisation ? p.regularisation.rows : 0;
    P.reg_w = p.has_regularisation ? p.regularisation.weight : 0.0;
    NP = (p.n <= 32) ? 32 : 64;
    // cascade layout: M1, M2, V | rlo, rup, rptr | rowstate, eqlist | rsrc bytes
    const int S = NP + 1;
    const int cap = ((nrows_max > 0 ? nrows_max : 1) + 1) & ~1;
    int total = ((2 * NP * S + 4 * NP) + 1) & ~1;
    P.lds_rows_off = total;
    P.lds_rows_cap = cap;
    total += 3 * cap + cap;
    total += (cap + 7) / 8;
    lds_bytes = (size_t)total * sizeof(double);
    return OSOT_OK;
}

}  // namespace osot

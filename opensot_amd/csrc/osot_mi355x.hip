// osot_mi355x.hip -- implementation of the C-ABI in include/osot_mi355x.h for gfx950 (MI355X).
//
// Host side: argument checking with the reference's error behaviour (size mismatches are refused like
// BackEnd::updateTask/updateConstraints do, src/solvers/BackEnd.cpp:19-93), translation of the static
// plan into kernel arguments, stream-ordered launches, hipEvent timing, RCCL all-gather.
// Device side: osot_kernels.h / osot_qp_core.h.  There is no CPU code path in this library.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "osot_host_plan.h"
#include "osot_kin.h"

using namespace osot;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(OSOT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

template <class K>
int ensure_lds(K kernel, size_t bytes) {
    if (bytes > 160 * 1024) return fail(OSOT_ERR_UNSUPPORTED, "problem does not fit the 160 KiB LDS of a CU");
    if (bytes > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return OSOT_OK;
}

}  // namespace

struct osot_solver {
    osot_plan_desc plan;
    int max_batch;
    int device;
    bool timing;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;   // recorded, not yet accumulated
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    // longest-first dispatch (osot_order_kernel): cost of the previous solve and the order built from it
    int schedule = 1;        // 0: in order, 1: longest first
    int* d_cost = nullptr;   // [max_batch]
    int* d_order = nullptr;  // [max_batch]
    int order_B = -1;        // batch size d_order is valid for (-1: none yet)
};

extern "C" {

const char* osot_version(void) { return "osot-mi355x 0.1 (gfx950)"; }
const char* osot_last_error(void) { return g_err.c_str(); }

int osot_device_count(int* count) {
    if (!count) return fail(OSOT_ERR_INVALID, "null count");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *count = 0; return fail(OSOT_ERR_HIP, hipGetErrorString(e)); }
    *count = c;
    return OSOT_OK;
}

int osot_plan_validate(const osot_plan_desc* plan) {
    const char* why;
    int rc = plan_validate(plan, &why);
    if (rc != OSOT_OK) return fail(rc, why);
    return OSOT_OK;
}
int osot_plan_level_rows(const osot_plan_desc* plan, int level, int* m_total, int* m_stored) {
    int rc = plan_level_rows(plan, level, m_total, m_stored);
    return rc == OSOT_OK ? rc : fail(rc, "bad plan/level");
}
int osot_plan_constraint_rows(const osot_plan_desc* plan, int* nc) {
    int rc = plan_constraint_rows(plan, nc);
    return rc == OSOT_OK ? rc : fail(rc, "bad plan");
}
int osot_plan_stored_constraint_rows(const osot_plan_desc* plan, int* nc_stored) {
    int rc = plan_stored_constraint_rows(plan, nc_stored);
    return rc == OSOT_OK ? rc : fail(rc, "bad plan");
}

int osot_solver_create(const osot_plan_desc* plan, int max_batch, int device, osot_solver** out) {
    if (!out) return fail(OSOT_ERR_INVALID, "null out");
    *out = nullptr;
    int rc = osot_plan_validate(plan);
    if (rc != OSOT_OK) return rc;
    if (max_batch < 1) return fail(OSOT_ERR_INVALID, "max_batch < 1");
    HIP_TRY(hipSetDevice(device));
    DevPlan P; int T; size_t lds;
    make_dev_plan(*plan, nullptr, P, T, lds);
    rc = (T == 32) ? ensure_lds(osot_cascade_kernel<32, false>, lds) : ensure_lds(osot_cascade_kernel<64, false>, lds);
    if (rc == OSOT_OK) rc = (T == 32) ? ensure_lds(osot_cascade_kernel<32, true>, lds) : ensure_lds(osot_cascade_kernel<64, true>, lds);
    if (rc != OSOT_OK) return rc;
    osot_solver* s = new osot_solver();
    s->plan = *plan;
    s->max_batch = max_batch;
    s->device = device;
    s->timing = false;
    *out = s;
    return OSOT_OK;
}

int osot_solver_destroy(osot_solver* s) {
    if (!s) return OSOT_OK;
    for (auto& p : s->events) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    for (auto& p : s->pool) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    if (s->d_cost) hipFree(s->d_cost);
    if (s->d_order) hipFree(s->d_order);
    delete s;
    return OSOT_OK;
}

int osot_solver_set_schedule(osot_solver* s, int mode) {
    if (!s) return fail(OSOT_ERR_INVALID, "null solver");
    if (mode != OSOT_SCHEDULE_IN_ORDER && mode != OSOT_SCHEDULE_LONGEST_FIRST) return fail(OSOT_ERR_INVALID, "unknown schedule mode");
    s->schedule = mode;
    s->order_B = -1;
    return OSOT_OK;
}

int osot_solver_set_timing(osot_solver* s, int enabled) {
    if (!s) return fail(OSOT_ERR_INVALID, "null solver");
    s->timing = enabled != 0;
    return OSOT_OK;
}

int osot_solver_kernel_time_ms(osot_solver* s, int reset, double* avg_ms, int* launches) {
    if (!s) return fail(OSOT_ERR_INVALID, "null solver");
    double total = 0.0;
    int cnt = 0;
    for (auto& p : s->events) {
        HIP_TRY(hipEventSynchronize(p.second));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p.first, p.second));
        total += ms;
        cnt++;
    }
    if (avg_ms) *avg_ms = cnt ? total / cnt : 0.0;
    if (launches) *launches = cnt;
    if (reset) {
        for (auto& p : s->events) s->pool.push_back(p);
        s->events.clear();
    }
    return OSOT_OK;
}

static int ihqp_launch(osot_solver* s, const osot_qp_batch* b, void* hip_stream, long long* prof);

int osot_ihqp_solve(osot_solver* s, const osot_qp_batch* b, void* hip_stream) {
    return ihqp_launch(s, b, hip_stream, nullptr);
}

int osot_solver_profile_phases(osot_solver* s, const osot_qp_batch* b, long long* cycles, void* hip_stream) {
    if (!cycles) return fail(OSOT_ERR_INVALID, "null cycles");
    return ihqp_launch(s, b, hip_stream, cycles);
}

static int ihqp_launch(osot_solver* s, const osot_qp_batch* b, void* hip_stream, long long* prof) {
    if (!s || !b) return fail(OSOT_ERR_INVALID, "null solver/batch");
    if (b->B < 0 || b->B > s->max_batch) return fail(OSOT_ERR_INVALID, "batch size exceeds max_batch");
    if (b->B == 0) return OSOT_OK;   // empty batch: nothing to do
    const osot_plan_desc& pl = s->plan;
    DevPlan P; int T; size_t lds;
    make_dev_plan(pl, b->level_active, P, T, lds);
    DevBatch D;
    std::memset(&D, 0, sizeof(D));
    D.B = b->B;
    for (int k = 0; k < pl.n_levels; ++k) {
        if (P.ma[k] > 0 && !b->A[k]) return fail(OSOT_ERR_INVALID, "A[k] is null for a level with stored rows");
        if (!b->b[k]) return fail(OSOT_ERR_INVALID, "b[k] is null");
        D.A[k] = b->A[k]; D.b[k] = b->b[k]; D.w[k] = b->w[k]; D.c[k] = b->c[k];
    }
    if (P.nc > 0 && (!b->lo || !b->up)) return fail(OSOT_ERR_INVALID, "plan has constraint rows but lo/up is null");
    if (P.nc_stored > 0 && !b->C) return fail(OSOT_ERR_INVALID, "plan has stored constraint rows but C is null");
    if (pl.n_bounds > 0 && (!b->l || !b->u)) return fail(OSOT_ERR_INVALID, "plan has bounds but l/u is null");
    if (!b->dq || !b->status) return fail(OSOT_ERR_INVALID, "dq/status output is null");
    D.C = P.nc_stored ? b->C : nullptr; D.lo = b->lo; D.up = b->up;
    D.l = pl.n_bounds ? b->l : nullptr; D.u = pl.n_bounds ? b->u : nullptr;
    D.dq = b->dq; D.x_levels = b->x_levels; D.status = b->status; D.iterations = b->iterations;
    if (pl.has_regularisation && !b->b_reg) return fail(OSOT_ERR_INVALID, "plan has a regularisation task but b_reg is null");
    D.b_reg = pl.has_regularisation ? b->b_reg : nullptr;
    D.prof = prof;
    D.accepted_slack = b->accepted_slack;
    hipStream_t st = (hipStream_t)hip_stream;
    if (s->schedule == 1) {
        if (!s->d_cost) {
            HIP_TRY(hipMalloc(&s->d_cost, sizeof(int) * (size_t)s->max_batch));
            HIP_TRY(hipMalloc(&s->d_order, sizeof(int) * (size_t)s->max_batch));
        }
        D.order = (s->order_B == b->B) ? s->d_order : nullptr;
        D.cost_out = s->d_cost;
    }
    const unsigned grid = (unsigned)b->B;
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (s->timing) {
        if (!s->pool.empty()) { ev = s->pool.back(); s->pool.pop_back(); }
        else { HIP_TRY(hipEventCreate(&ev.first)); HIP_TRY(hipEventCreate(&ev.second)); }
        HIP_TRY(hipEventRecord(ev.first, st));
    }
    if (prof) {
        if (T == 32) hipLaunchKernelGGL((osot_cascade_kernel<32, true>), dim3(grid), dim3(64), lds, st, P, D);
        else hipLaunchKernelGGL((osot_cascade_kernel<64, true>), dim3(grid), dim3(64), lds, st, P, D);
    } else {
        if (T == 32) hipLaunchKernelGGL((osot_cascade_kernel<32, false>), dim3(grid), dim3(64), lds, st, P, D);
        else hipLaunchKernelGGL((osot_cascade_kernel<64, false>), dim3(grid), dim3(64), lds, st, P, D);
    }
    HIP_TRY(hipGetLastError());
    if (s->timing) {
        HIP_TRY(hipEventRecord(ev.second, st));
        s->events.push_back(ev);
    }
    if (s->schedule == 1) {   // the order for the next solve of a batch of this size
        hipLaunchKernelGGL(osot_order_kernel, dim3(1), dim3(1024), 0, st, (const int*)s->d_cost, s->d_order, b->B);
        HIP_TRY(hipGetLastError());
        s->order_B = b->B;
    }
    return OSOT_OK;
}

int osot_stack_update(osot_solver* s, const osot_leaf_batch* leaf, const osot_assembled_out* out, void* hip_stream) {
    if (!s || !leaf || !out) return fail(OSOT_ERR_INVALID, "null argument");
    if (leaf->B < 0 || leaf->B > s->max_batch) return fail(OSOT_ERR_INVALID, "batch size exceeds max_batch");
    if (leaf->B == 0) return OSOT_OK;
    const osot_plan_desc& pl = s->plan;
    DevUpdate U;
    std::memset(&U, 0, sizeof(U));
    U.B = leaf->B; U.n = pl.n; U.L = pl.n_levels;
    plan_constraint_rows(&pl, &U.nc);
    plan_stored_constraint_rows(&pl, &U.nc_stored);
    int flat = 0;
    for (int k = 0; k < pl.n_levels; ++k) {
        plan_level_rows(&pl, k, &U.m[k], nullptr);
        if (!out->b[k]) return fail(OSOT_ERR_INVALID, "out.b[k] is null");
        U.b[k] = out->b[k]; U.w[k] = out->w[k];
        int off = 0;
        for (int j = 0; j < pl.level[k].n_tasks; ++j) {
            const osot_task_desc& t = pl.level[k].task[j];
            DevTask& d = U.task[flat++];
            d.level = k; d.kind = t.kind; d.rows = t.rows; d.off = off;
            d.weight = t.weight; d.lambda = t.lambda; d.ogain = t.orientation_gain; d.lambda2 = t.lambda2;
            d.mask = t.row_mask; d.prow = task_parent_rows(t, pl.n); d.sublam = t.row_mask ? t.sub_lambda : 1.0;
            d.p0 = leaf->task[k][j].p0; d.p1 = leaf->task[k][j].p1; d.p2 = leaf->task[k][j].p2;
            if (!d.p0) return fail(OSOT_ERR_INVALID, "leaf input p0 of a task is null");
            if (t.kind != OSOT_TASK_GENERIC && t.kind != OSOT_TASK_ACC_POSTURAL && !d.p1)
                return fail(OSOT_ERR_INVALID, "leaf input p1 of a task is null");
            off += t.rows;
        }
    }
    if (pl.has_regularisation) {   // one more flat entry; its b goes to out->b_reg (level = -1)
        const osot_task_desc& t = pl.regularisation;
        if (!out->b_reg) return fail(OSOT_ERR_INVALID, "out.b_reg is null");
        DevTask& d = U.task[flat++];
        d.level = -1; d.kind = t.kind; d.rows = t.rows; d.off = 0;
        d.weight = t.weight; d.lambda = t.lambda; d.ogain = t.orientation_gain; d.lambda2 = t.lambda2;
        d.mask = 0ull; d.prow = t.rows; d.sublam = 1.0;
        d.p0 = leaf->regularisation.p0; d.p1 = leaf->regularisation.p1; d.p2 = leaf->regularisation.p2;
        if (!d.p0) return fail(OSOT_ERR_INVALID, "leaf input p0 of the regularisation task is null");
        if (t.kind == OSOT_TASK_POSTURAL && !d.p1) return fail(OSOT_ERR_INVALID, "leaf input p1 of the regularisation task is null");
        U.b_reg = out->b_reg;
    }
    U.ntasks = flat;
    U.nbounds = pl.n_bounds;
    for (int j = 0; j < pl.n_bounds; ++j) {
        DevBound& d = U.bound[j];
        d.kind = pl.bound[j].kind; d.scaling = pl.bound[j].scaling; d.dT = pl.bound[j].dT;
        d.p0 = leaf->bound[j].p0; d.p1 = leaf->bound[j].p1; d.p2 = leaf->bound[j].p2;
        if (!d.p0) return fail(OSOT_ERR_INVALID, "leaf input p0 of a bound is null");
        if (d.kind == OSOT_BOUND_JOINT_LIMITS && (!d.p1 || !d.p2)) return fail(OSOT_ERR_INVALID, "joint limits need q, q_min, q_max");
        if (d.kind == OSOT_BOUND_GENERIC && !d.p1) return fail(OSOT_ERR_INVALID, "generic bound needs l and u");
    }
    if (pl.n_bounds > 0 && (!out->l || !out->u)) return fail(OSOT_ERR_INVALID, "out.l/out.u is null");
    U.l = out->l; U.u = out->u;
    U.nrowblocks = pl.n_rowblocks;
    int roff = 0, soff = 0;
    for (int j = 0; j < pl.n_rowblocks; ++j) {
        DevRowBlock& d = U.rowblock[j];
        d.kind = pl.rowblock[j].kind; d.rows = pl.rowblock[j].rows; d.off = roff;
        d.stored_off = soff; d.first_col = pl.rowblock[j].first_col;
        d.dT = pl.rowblock[j].dT; d.p = pl.rowblock[j].p; d.mu = pl.rowblock[j].mu;
        d.lambda = pl.rowblock[j].task_lambda; d.ogain = pl.rowblock[j].task_orientation_gain;
        d.err_lb = pl.rowblock[j].err_lb; d.err_ub = pl.rowblock[j].err_ub;
        if (!rows_are_implicit(d.kind)) soff += d.rows;
        d.d_threshold = pl.rowblock[j].d_threshold;
        d.detection_threshold = pl.rowblock[j].detection_threshold;
        d.bound_scaling = pl.rowblock[j].bound_scaling;
        d.p0 = leaf->rows[j].p0; d.p1 = leaf->rows[j].p1; d.p2 = leaf->rows[j].p2;
        if (!d.p0) return fail(OSOT_ERR_INVALID, "leaf input p0 of a row block is null");
        if ((d.kind == OSOT_ROWS_GENERIC || d.kind == OSOT_ROWS_COLLISION || d.kind == OSOT_ROWS_TORQUE_LIMITS ||
             d.kind == OSOT_ROWS_ACC_JOINT_LIMITS || d.kind == OSOT_ROWS_ACC_VELOCITY_LIMITS ||
             d.kind == OSOT_ROWS_TASK_CARTESIAN || d.kind == OSOT_ROWS_TASK_COM || d.kind == OSOT_ROWS_UNIT_GENERIC) && !d.p1)
            return fail(OSOT_ERR_INVALID, "leaf input p1 of a row block is null");
        if (d.kind == OSOT_ROWS_ACC_JOINT_LIMITS && !d.p2) return fail(OSOT_ERR_INVALID, "acceleration joint limits need qddot_max");
        if (d.kind == OSOT_ROWS_GENERIC && !d.p2) return fail(OSOT_ERR_INVALID, "generic rows need C, lo, up");
        roff += d.rows;
    }
    if (U.nc > 0 && (!out->lo || !out->up)) return fail(OSOT_ERR_INVALID, "out.lo/up is null");
    if (U.nc_stored > 0 && !out->C) return fail(OSOT_ERR_INVALID, "out.C is null");
    U.C = out->C; U.lo = out->lo; U.up = out->up;
    hipLaunchKernelGGL(osot_update_kernel, dim3((unsigned)leaf->B), dim3(64), 0, (hipStream_t)hip_stream, U);
    HIP_TRY(hipGetLastError());
    return OSOT_OK;
}

int osot_qp_solve_batch(int B, int n, int nc, const double* H, const double* g, const double* A,
                        const double* lA, const double* uA, const double* l, const double* u,
                        double eps_abs, int max_iter, double* x, int* status, int* iterations,
                        void* hip_stream) {
    if (B < 0 || n < 1 || n > OSOT_MAX_VARS || nc < 0) return fail(OSOT_ERR_INVALID, "bad sizes");
    if (B == 0) return OSOT_OK;
    if (!H || !g || !x || !status) return fail(OSOT_ERR_INVALID, "null H/g/x/status");
    if (nc > 0 && (!A || !lA || !uA)) return fail(OSOT_ERR_INVALID, "nc > 0 but A/lA/uA is null");
    if ((l == nullptr) != (u == nullptr)) return fail(OSOT_ERR_INVALID, "l and u must both be given or both be null");
    DevQP Q;
    std::memset(&Q, 0, sizeof(Q));
    Q.B = B; Q.n = n; Q.nc = nc;
    Q.max_iter = max_iter > 0 ? max_iter : 20 * (n + nc) + 100;
    Q.eps_abs = eps_abs;
    Q.H = H; Q.g = g; Q.A = A; Q.lA = lA; Q.uA = uA; Q.l = l; Q.u = u;
    Q.x = x; Q.status = status; Q.iterations = iterations;
    const int T = n <= 32 ? 32 : 64;
    const size_t lds = (size_t)lds_layout(T, nc, &Q.lds_rows_off, &Q.lds_rows_cap) * sizeof(double);
    int rc = (T == 32) ? ensure_lds(osot_qp_kernel<32>, lds) : ensure_lds(osot_qp_kernel<64>, lds);
    if (rc != OSOT_OK) return rc;
    const unsigned grid = (unsigned)B;
    if (T == 32) hipLaunchKernelGGL(osot_qp_kernel<32>, dim3(grid), dim3(64), lds, (hipStream_t)hip_stream, Q);
    else hipLaunchKernelGGL(osot_qp_kernel<64>, dim3(grid), dim3(64), lds, (hipStream_t)hip_stream, Q);
    HIP_TRY(hipGetLastError());
    return OSOT_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// batch-of-one BackEnd surface (host pointers).  Mirrors OpenSoT::solvers::BackEnd
// (include/OpenSoT/solvers/BackEnd.h:23-171, src/solvers/BackEnd.cpp:19-104) and the qpOASES
// back-end's re-allocation on a changed row count (QPOasesBackEnd.cpp:229-244).
// ---------------------------------------------------------------------------------------------------
struct osot_backend {
    int nv, nc;
    int hessian_type;
    double eps_abs;
    bool inited, has_bounds;
    std::vector<double> H, g, A, lA, uA, l, u, x;
    double* d_buf;      // device arena
    size_t d_cap;
    int* d_status;
    int last_status, last_iters;
};

namespace {
int backend_run(osot_backend* be) {
    const int n = be->nv, nc = be->nc;
    const size_t need = (size_t)n * n + n + (size_t)nc * n + 2 * (size_t)nc + 2 * (size_t)n + n;
    if (need > be->d_cap) {
        if (be->d_buf) hipFree(be->d_buf);
        be->d_buf = nullptr;
        HIP_TRY(hipMalloc((void**)&be->d_buf, need * sizeof(double)));
        be->d_cap = need;
    }
    if (!be->d_status) HIP_TRY(hipMalloc((void**)&be->d_status, 2 * sizeof(int)));
    double* dH = be->d_buf;
    double* dg = dH + (size_t)n * n;
    double* dA = dg + n;
    double* dlA = dA + (size_t)nc * n;
    double* duA = dlA + nc;
    double* dl = duA + nc;
    double* du = dl + n;
    double* dx = du + n;
    HIP_TRY(hipMemcpy(dH, be->H.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dg, be->g.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    if (nc) {
        HIP_TRY(hipMemcpy(dA, be->A.data(), sizeof(double) * nc * n, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dlA, be->lA.data(), sizeof(double) * nc, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(duA, be->uA.data(), sizeof(double) * nc, hipMemcpyHostToDevice));
    }
    if (be->has_bounds) {
        HIP_TRY(hipMemcpy(dl, be->l.data(), sizeof(double) * n, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(du, be->u.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    }
    int rc = osot_qp_solve_batch(1, n, nc, dH, dg, nc ? dA : nullptr, nc ? dlA : nullptr, nc ? duA : nullptr,
                                 be->has_bounds ? dl : nullptr, be->has_bounds ? du : nullptr, be->eps_abs, 0,
                                 dx, be->d_status, be->d_status + 1, nullptr);
    if (rc != OSOT_OK) return rc;
    int st[2];
    HIP_TRY(hipMemcpy(st, be->d_status, 2 * sizeof(int), hipMemcpyDeviceToHost));
    be->last_status = st[0];
    be->last_iters = st[1];
    if (st[0] != OSOT_STATUS_SOLVED) return fail(OSOT_ERR_NOT_SOLVED, "QP not solved (infeasible / iteration limit / H not PD)");
    HIP_TRY(hipMemcpy(be->x.data(), dx, sizeof(double) * n, hipMemcpyDeviceToHost));
    return OSOT_OK;
}
}  // namespace

extern "C" {

int osot_backend_create(int number_of_variables, int number_of_constraints, int hessian_type,
                        double eps_regularisation, osot_backend** out) {
    if (!out) return fail(OSOT_ERR_INVALID, "null out");
    *out = nullptr;
    if (number_of_variables < 1 || number_of_variables > OSOT_MAX_VARS)
        return fail(OSOT_ERR_INVALID, "number_of_variables out of range (1..64)");
    if (number_of_constraints < 0) return fail(OSOT_ERR_INVALID, "negative number_of_constraints");
    if (eps_regularisation < 0) return fail(OSOT_ERR_INVALID, "Negative eps is not allowed!");
    osot_backend* be = new osot_backend();
    be->nv = number_of_variables; be->nc = number_of_constraints; be->hessian_type = hessian_type;
    be->eps_abs = 1.0e3 * 2.221e-16 * eps_regularisation;   // QPOasesBackEnd.cpp:57,67
    be->inited = false; be->has_bounds = false;
    be->d_buf = nullptr; be->d_cap = 0; be->d_status = nullptr;
    be->last_status = 0; be->last_iters = 0;
    be->x.assign(be->nv, 0.0);
    *out = be;
    return OSOT_OK;
}

int osot_backend_destroy(osot_backend* be) {
    if (!be) return OSOT_OK;
    if (be->d_buf) hipFree(be->d_buf);
    if (be->d_status) hipFree(be->d_status);
    delete be;
    return OSOT_OK;
}

int osot_backend_update_task(osot_backend* be, const double* H, const double* g) {
    if (!be || !H || !g) return fail(OSOT_ERR_INVALID, "null argument");
    be->H.assign(H, H + (size_t)be->nv * be->nv);
    be->g.assign(g, g + be->nv);
    return OSOT_OK;
}

int osot_backend_update_constraints(osot_backend* be, const double* A, const double* lA, const double* uA,
                                    int number_of_constraints) {
    if (!be || number_of_constraints < 0) return fail(OSOT_ERR_INVALID, "bad argument");
    if (number_of_constraints > 0 && (!A || !lA || !uA)) return fail(OSOT_ERR_INVALID, "null A/lA/uA");
    be->nc = number_of_constraints;   // a changed row count re-allocates (QPOasesBackEnd.cpp:229-244)
    be->A.assign(A, A + (size_t)be->nc * be->nv);
    be->lA.assign(lA, lA + be->nc);
    be->uA.assign(uA, uA + be->nc);
    for (int i = 0; i < be->nc; ++i)
        if (be->lA[i] > be->uA[i]) return fail(OSOT_ERR_INVALID, "lA > uA");
    return OSOT_OK;
}

int osot_backend_update_bounds(osot_backend* be, const double* l, const double* u) {
    if (!be) return fail(OSOT_ERR_INVALID, "null backend");
    if (!l && !u) { be->has_bounds = false; be->l.clear(); be->u.clear(); return OSOT_OK; }
    if (!l || !u) return fail(OSOT_ERR_INVALID, "l and u must both be given");
    for (int i = 0; i < be->nv; ++i)
        if (l[i] > u[i]) return fail(OSOT_ERR_INVALID, "l > u");   // BackEnd.cpp:76-84
    be->l.assign(l, l + be->nv);
    be->u.assign(u, u + be->nv);
    be->has_bounds = true;
    return OSOT_OK;
}

int osot_backend_init_problem(osot_backend* be, const double* H, const double* g, const double* A,
                              const double* lA, const double* uA, const double* l, const double* u) {
    if (!be) return fail(OSOT_ERR_INVALID, "null backend");
    int rc = osot_backend_update_task(be, H, g);
    if (rc != OSOT_OK) return rc;
    rc = osot_backend_update_constraints(be, A, lA, uA, be->nc);
    if (rc != OSOT_OK) return rc;
    rc = osot_backend_update_bounds(be, l, u);
    if (rc != OSOT_OK) return rc;
    rc = backend_run(be);
    be->inited = (rc == OSOT_OK);
    return rc;
}

int osot_backend_solve(osot_backend* be) {
    if (!be) return fail(OSOT_ERR_INVALID, "null backend");
    if (be->H.empty()) return fail(OSOT_ERR_INVALID, "solve() before initProblem()");
    return backend_run(be);
}

int osot_backend_get_solution(osot_backend* be, double* x) {
    if (!be || !x) return fail(OSOT_ERR_INVALID, "null argument");
    std::memcpy(x, be->x.data(), sizeof(double) * be->nv);
    return OSOT_OK;
}

int osot_backend_get_objective(osot_backend* be, double* f) {
    if (!be || !f) return fail(OSOT_ERR_INVALID, "null argument");
    if (be->H.empty()) return fail(OSOT_ERR_INVALID, "no problem");
    const int n = be->nv;
    double v = 0.0;
    for (int i = 0; i < n; ++i) {
        double hx = 0.0;
        for (int j = 0; j < n; ++j) hx += be->H[(size_t)i * n + j] * be->x[j];
        v += be->x[i] * (0.5 * (hx + be->eps_abs * be->x[i]) + be->g[i]);
    }
    *f = v;
    return OSOT_OK;
}

int osot_backend_set_eps_regularisation(osot_backend* be, double eps_abs) {
    if (!be) return fail(OSOT_ERR_INVALID, "null backend");
    if (eps_abs < 0.0) return fail(OSOT_ERR_INVALID, "Negative eps is not allowed!");   // QPOasesBackEnd.cpp:358-366
    be->eps_abs = eps_abs;
    return OSOT_OK;
}
int osot_backend_get_eps_regularisation(osot_backend* be, double* eps_abs) {
    if (!be || !eps_abs) return fail(OSOT_ERR_INVALID, "null argument");
    *eps_abs = be->eps_abs;
    return OSOT_OK;
}
int osot_backend_get_num_variables(osot_backend* be, int* nv) {
    if (!be || !nv) return fail(OSOT_ERR_INVALID, "null argument");
    *nv = be->nv;
    return OSOT_OK;
}
int osot_backend_get_num_constraints(osot_backend* be, int* nc) {
    if (!be || !nc) return fail(OSOT_ERR_INVALID, "null argument");
    *nc = be->nc;
    return OSOT_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// batched kinematics producer
// ---------------------------------------------------------------------------------------------------
struct osot_kin {
    DevKin* dev;
    int n, n_frames, n_pairs, device;
};

extern "C" {

int osot_kin_create(const osot_kin_desc* d, int device, osot_kin** out) {
    if (!d || !out) return fail(OSOT_ERR_INVALID, "null argument");
    if (d->n < 1 || d->n > OSOT_KIN_MAX_JOINTS) return fail(OSOT_ERR_INVALID, "joint count out of range");
    if (d->n_frames < 0 || d->n_frames > OSOT_KIN_MAX_FRAMES) return fail(OSOT_ERR_INVALID, "frame count out of range");
    DevKin h;
    std::memset(&h, 0, sizeof(h));
    h.d = *d;
    for (int j = 0; j < d->n; ++j) {
        if (d->parent[j] >= j || d->parent[j] < -1) return fail(OSOT_ERR_INVALID, "joints must be in tree order (parent[j] < j)");
        if (d->type[j] != OSOT_JOINT_REVOLUTE && d->type[j] != OSOT_JOINT_PRISMATIC) return fail(OSOT_ERR_INVALID, "unknown joint type");
        h.anc[j] = (1ull << j) | (d->parent[j] >= 0 ? h.anc[d->parent[j]] : 0ull);
        for (int a = 0; a <= j; ++a) if ((h.anc[j] >> a) & 1ull) h.sub[a] |= (1ull << j);
        h.total_mass += d->mass[j];
    }
    for (int f = 0; f < d->n_frames; ++f)
        if (d->frame_joint[f] < 0 || d->frame_joint[f] >= d->n) return fail(OSOT_ERR_INVALID, "frame attached to a joint out of range");
    if (d->n_pairs < 0 || d->n_pairs > OSOT_KIN_MAX_PAIRS) return fail(OSOT_ERR_INVALID, "collision pair count out of range");
    for (int p = 0; p < d->n_pairs; ++p) {
        for (int sd = 0; sd < 2; ++sd)
            if (d->pair_joint[p][sd] < 0 || d->pair_joint[p][sd] >= d->n) return fail(OSOT_ERR_INVALID, "collision shape attached to a joint out of range");
        if (!(d->pair_radius[p][0] >= 0.0) || !(d->pair_radius[p][1] >= 0.0)) return fail(OSOT_ERR_INVALID, "negative capsule radius");
    }
    if (!(h.total_mass > 0.0)) h.total_mass = 1.0;
    HIP_TRY(hipSetDevice(device));
    osot_kin* k = new osot_kin();
    k->n = d->n; k->n_frames = d->n_frames; k->n_pairs = d->n_pairs; k->device = device;
    hipError_t e = hipMalloc(&k->dev, sizeof(DevKin));
    if (e == hipSuccess) e = hipMemcpy(k->dev, &h, sizeof(DevKin), hipMemcpyHostToDevice);
    if (e != hipSuccess) { delete k; return fail(OSOT_ERR_HIP, hipGetErrorString(e)); }
    *out = k;
    return OSOT_OK;
}

int osot_kin_destroy(osot_kin* k) {
    if (!k) return OSOT_OK;
    if (k->dev) hipFree(k->dev);
    delete k;
    return OSOT_OK;
}

int osot_kinematics(osot_kin* k, const osot_kin_batch* b, void* hip_stream) {
    if (!k || !b) return fail(OSOT_ERR_INVALID, "null argument");
    if (b->B < 0) return fail(OSOT_ERR_INVALID, "negative batch");
    if (b->B == 0) return OSOT_OK;
    if (!b->q) return fail(OSOT_ERR_INVALID, "q is null");
    if (k->n_pairs > 0 && (b->pair_dist || b->pair_J))
        hipLaunchKernelGGL(osot_kin_kernel<true>, dim3((unsigned)b->B), dim3(64), 0, (hipStream_t)hip_stream, (const DevKin*)k->dev, *b);
    else
        hipLaunchKernelGGL(osot_kin_kernel<false>, dim3((unsigned)b->B), dim3(64), 0, (hipStream_t)hip_stream, (const DevKin*)k->dev, *b);
    HIP_TRY(hipGetLastError());
    return OSOT_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// multi-GPU: one RCCL all-gather of the solved dq shards (instances are independent)
// ---------------------------------------------------------------------------------------------------
struct osot_comm {
    ncclComm_t comm;
    int rank, world;
};

extern "C" {

int osot_comm_unique_id(void* id128) {
    if (!id128) return fail(OSOT_ERR_INVALID, "null id");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(OSOT_ERR_COMM, ncclGetErrorString(r));
    std::memcpy(id128, &id, sizeof(id));
    return OSOT_OK;
}

int osot_comm_create(const void* id128, int rank, int world, int device, osot_comm** out) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return fail(OSOT_ERR_INVALID, "bad argument");
    *out = nullptr;
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t c;
    ncclResult_t r = ncclCommInitRank(&c, world, id, rank);
    if (r != ncclSuccess) return fail(OSOT_ERR_COMM, ncclGetErrorString(r));
    osot_comm* oc = new osot_comm();
    oc->comm = c; oc->rank = rank; oc->world = world;
    *out = oc;
    return OSOT_OK;
}

int osot_comm_destroy(osot_comm* c) {
    if (!c) return OSOT_OK;
    ncclCommDestroy(c->comm);
    delete c;
    return OSOT_OK;
}

int osot_allgather_dq(osot_comm* c, const double* send, double* recv, long long count, void* hip_stream) {
    if (!c || !send || !recv || count < 0) return fail(OSOT_ERR_INVALID, "bad argument");
    ncclResult_t r = ncclAllGather(send, recv, (size_t)count, ncclDouble, c->comm, (hipStream_t)hip_stream);
    if (r != ncclSuccess) return fail(OSOT_ERR_COMM, ncclGetErrorString(r));
    return OSOT_OK;
}

}  // extern "C"

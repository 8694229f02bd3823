// osot_mi355x.hip -- implementation of the C-ABI in include/osot_mi355x.h for gfx950 (MI355X).
//
// Host side: argument checking with the reference's error behaviour (size mismatches are refused like
// BackEnd::updateTask/updateConstraints do, src/solvers/BackEnd.cpp:19-93), translation of the static
// plan into kernel arguments, stream-ordered launches, hipEvent timing, RCCL all-gather.
// Device side: osot_kernels.h / osot_qp_core.h.  There is no CPU code path in this library.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "osot_host_plan.h"
#include "osot_kin.h"
#include "osot_control.h"
#include "osot_ehqp.h"
#include "osot_id.h"
#include "osot_nhqp_host.h"
#include "osot_admm.h"
#include "osot_qp_big.h"

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

// Every entry point that touches a device runs under this guard: the calling thread's current device is saved, the
// object's device made current, and the caller's restored on the way out (several solvers on different GPUs may live
// in one process; the caller's current device is never changed).
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) ok = (hipSetDevice(device) == hipSuccess);
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

template <class K>
int ensure_lds(K kernel, size_t bytes) {
    if (bytes > 160 * 1024) return fail(OSOT_ERR_UNSUPPORTED, "problem does not fit the 160 KiB LDS of a CU");
    if (bytes > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return OSOT_OK;
}

}  // namespace

// calls f with the padded size of the cascade instantiation as a compile-time constant (make_dev_plan picks it: 32, 56 or 64)
template <typename F>
static inline auto by_np(int T, F&& f) {
#ifdef OSOT_ONLY_NP32      // developer builds (tools/build_variant.sh NAME -DOSOT_ONLY_NP32): the 32-lane instantiations only -- a fifth of the
    (void)T;               // compile time for A/B runs of the headline kernel; every other layout would run THIS one: never shipped
    return f(std::integral_constant<int, 32>{});
#endif
    if (T == 32) return f(std::integral_constant<int, 32>{});
    if (T == 40) return f(std::integral_constant<int, 40>{});
    if (T == 56) return f(std::integral_constant<int, 56>{});
    return f(std::integral_constant<int, 64>{});
}

struct osot_solver {
    osot_plan_desc plan;
    int max_batch;
    int device;
    bool timing;
    int timing_stride = 1, timing_count = 0;   // every timing_stride-th launch is bracketed by events
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;   // recorded, not yet accumulated
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    // longest-first dispatch (order_body in osot_kernels.h): cost estimates of the previous solves and the order built from them
    int schedule = 1;        // 0: in order, 1: longest first
    int* d_cost = nullptr;   // [2][max_batch]: the launches alternate (read one, write the other)
    int* d_order = nullptr;  // [2][max_batch]: likewise (use one, the launch's order workgroup fills the other)
    int flip = 0;            // which halves the next launch writes
    int order_B = -1;        // batch size the order the next launch reads is valid for (-1: none yet)
    // d_cost / d_order are stream-ordered state shared by consecutive solves.  A solver is meant to be driven from one
    // thread on one stream (like the reference's Solver: no locks, SURVEY 8b "threading"); a solve that arrives on a
    // DIFFERENT stream than the previous one first waits (host side, rare) for that stream, so that it never reads
    // d_order while the previous launch's order workgroup is still writing it.  No event traffic on the common path.
    hipStream_t order_stream = nullptr;
    DevUpdatePlan* d_uplan = nullptr;   // static part of the update kernel's arguments, uploaded at creation
    DevUpdatePlan h_uplan;
    unsigned char task_active[OSOT_MAX_LEVELS * OSOT_MAX_TASKS];   // Task::setActive flags
    bool any_inactive = false;
    int slots = 1;      // wavefronts of the cascade kernel the device holds at once (CUs x resident workgroups per CU)
    NhqpWorkspace nhqp; // scratch of the null-space front-end, allocated at its first use
    bool nhqp_ready = false;
    bool nhqp_lds_ready = false;
    // hot start (osot_solver_set_hotstart): the inequality working set each level of each instance ended with
    double* d_rows = nullptr;   // [max_batch][rows_doubles] row tables (plans whose 64-lane kernels keep them out of LDS)
    int hotstart = 0;
    int specialise = 1;         // osot_solver_set_specialisation: the BOX instantiation for plans without constraint rows
    int* d_hot = nullptr;    // [max_batch][n_levels][T] constraint codes, -1 = none (allocated when first switched on)
    int hot_T = 0;
};

extern "C" {

const char* osot_version(void) { return "osot-mi355x 0.1 (gfx950)"; }
const char* osot_last_error(void) { return g_err.c_str(); }

int osot_abi_layout(const char* name, unsigned long long* size, unsigned long long* offsets, int max_fields, int* n_fields) {
    if (!name || !size) return fail(OSOT_ERR_INVALID, "null name/size");
    std::vector<unsigned long long> off;
    unsigned long long sz = 0;
    const std::string nm(name);
#define OSOT_LAYOUT_BEGIN(T) if (nm == #T) { typedef T osot_layout_t; sz = sizeof(T);
#define OSOT_F(f) off.push_back((unsigned long long)offsetof(osot_layout_t, f));
#define OSOT_LAYOUT_END() }
    OSOT_LAYOUT_BEGIN(osot_task_desc) OSOT_F(kind) OSOT_F(rows) OSOT_F(weight) OSOT_F(lambda) OSOT_F(orientation_gain) OSOT_F(lambda2)
        OSOT_F(row_mask) OSOT_F(parent_rows) OSOT_F(sub_lambda) OSOT_F(body_frame) OSOT_F(dense_weight) OSOT_F(acc_gain_matrices) OSOT_LAYOUT_END()
    OSOT_LAYOUT_BEGIN(osot_level_desc) OSOT_F(n_tasks) OSOT_F(task) OSOT_LAYOUT_END()
    OSOT_LAYOUT_BEGIN(osot_bound_desc) OSOT_F(kind) OSOT_F(scaling) OSOT_F(dT) OSOT_LAYOUT_END()
    OSOT_LAYOUT_BEGIN(osot_rows_desc) OSOT_F(kind) OSOT_F(rows) OSOT_F(d_threshold) OSOT_F(detection_threshold) OSOT_F(bound_scaling)
        OSOT_F(first_col) OSOT_F(dT) OSOT_F(p) OSOT_F(mu) OSOT_F(task_lambda) OSOT_F(task_orientation_gain) OSOT_F(err_lb) OSOT_F(err_ub)
        OSOT_F(task_body_frame) OSOT_F(n_candidates) OSOT_F(only_level) OSOT_LAYOUT_END()
    OSOT_LAYOUT_BEGIN(osot_plan_desc) OSOT_F(n) OSOT_F(n_levels) OSOT_F(level) OSOT_F(n_bounds) OSOT_F(bound) OSOT_F(n_rowblocks)
        OSOT_F(rowblock) OSOT_F(eps_abs) OSOT_F(max_iter) OSOT_F(has_regularisation) OSOT_F(regularisation) OSOT_F(regularisation_dense) OSOT_LAYOUT_END()
    OSOT_LAYOUT_BEGIN(osot_qp_batch) OSOT_F(B) OSOT_F(A) OSOT_F(b) OSOT_F(w) OSOT_F(c) OSOT_F(C) OSOT_F(lo) OSOT_F(up) OSOT_F(l) OSOT_F(u)
        OSOT_F(level_active) OSOT_F(dq) OSOT_F(x_levels) OSOT_F(status) OSOT_F(iterations) OSOT_F(b_reg) OSOT_F(WA) OSOT_F(Wb)
        OSOT_F(accepted_slack) OSOT_F(A_reg) OSOT_LAYOUT_END()
    OSOT_LAYOUT_BEGIN(osot_leaf_ptrs) OSOT_F(p0) OSOT_F(p1) OSOT_F(p2) OSOT_F(W) OSOT_LAYOUT_END()
    OSOT_LAYOUT_BEGIN(osot_leaf_batch) OSOT_F(B) OSOT_F(task) OSOT_F(bound) OSOT_F(rows) OSOT_F(regularisation) OSOT_LAYOUT_END()
    OSOT_LAYOUT_BEGIN(osot_assembled_out) OSOT_F(b) OSOT_F(w) OSOT_F(C) OSOT_F(lo) OSOT_F(up) OSOT_F(l) OSOT_F(u) OSOT_F(b_reg) OSOT_F(WA)
        OSOT_F(Wb) OSOT_F(A) OSOT_LAYOUT_END()
    OSOT_LAYOUT_BEGIN(osot_backend_options) OSOT_F(max_iterations) OSOT_F(last_iterations) OSOT_F(last_status) OSOT_LAYOUT_END()
    OSOT_LAYOUT_BEGIN(osot_nhqp_options) OSOT_F(free_vars) OSOT_F(min_sv_ratio) OSOT_F(no_ab_regularization)
        OSOT_F(no_selective_ns_regularization) OSOT_F(min_sv_ratio_is_set) OSOT_F(level_no_ab_regularization)
        OSOT_F(level_no_selective_ns_regularization) OSOT_F(level_min_sv_ratio_is_set) OSOT_F(level_min_sv_ratio) OSOT_F(level_W) OSOT_LAYOUT_END()
    OSOT_LAYOUT_BEGIN(osot_admm_options) OSOT_F(eps_abs) OSOT_F(eps_rel) OSOT_F(rho) OSOT_F(sigma) OSOT_F(alpha) OSOT_F(max_iter)
        OSOT_F(scaling) OSOT_F(check_every) OSOT_LAYOUT_END()
    OSOT_LAYOUT_BEGIN(osot_id_model) OSOT_F(B) OSOT_F(nv) OSOT_F(n_contacts) OSOT_F(contact_dim) OSOT_F(Bm) OSOT_F(h) OSOT_F(Jc)
        OSOT_F(floating_base) OSOT_LAYOUT_END()
    OSOT_LAYOUT_BEGIN(osot_kin_desc) OSOT_F(n) OSOT_F(parent) OSOT_F(type) OSOT_F(axis) OSOT_F(R0) OSOT_F(p0) OSOT_F(mass) OSOT_F(com)
        OSOT_F(n_frames) OSOT_F(frame_joint) OSOT_F(frame_R) OSOT_F(frame_p) OSOT_F(n_pairs) OSOT_F(pair_joint) OSOT_F(pair_seg)
        OSOT_F(pair_radius) OSOT_F(frame_body) OSOT_F(frame_col_mask) OSOT_F(com_col_mask) OSOT_F(pair_kind) OSOT_F(pair_env)
        OSOT_F(pair_box) OSOT_F(pair_shape_R) OSOT_F(pair_shape_p) OSOT_F(n_env) OSOT_F(frame_base) OSOT_LAYOUT_END()
    OSOT_LAYOUT_BEGIN(osot_kin_batch) OSOT_F(B) OSOT_F(q) OSOT_F(frame_pose) OSOT_F(frame_J) OSOT_F(frame_J_stride) OSOT_F(com)
        OSOT_F(com_J) OSOT_F(com_J_stride) OSOT_F(pair_dist) OSOT_F(pair_J) OSOT_F(pair_J_stride) OSOT_F(env_pose) OSOT_F(env_pose_stride) OSOT_LAYOUT_END()
#undef OSOT_LAYOUT_BEGIN
#undef OSOT_F
#undef OSOT_LAYOUT_END
    if (sz == 0) return fail(OSOT_ERR_INVALID, std::string("osot_abi_layout: unknown struct ") + name);
    *size = sz;
    if (n_fields) *n_fields = (int)off.size();
    if (offsets) for (int i = 0; i < (int)off.size() && i < max_fields; ++i) offsets[i] = off[i];
    return OSOT_OK;
}

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
    DeviceGuard guard(device);
    if (!guard.ok) return fail(OSOT_ERR_HIP, "hipSetDevice failed");
    DevPlan P; int T; size_t lds;
    make_dev_plan(*plan, nullptr, P, T, lds);
    rc = by_np(T, [&](auto np) {
        constexpr int NP = decltype(np)::value;
        int r = ensure_lds(osot_cascade_kernel<NP, false>, lds);
        if (r == OSOT_OK) r = ensure_lds(osot_cascade_kernel<NP, true>, lds);
        if (r == OSOT_OK) r = ensure_lds(osot_cycle_kernel<NP, false>, lds);
        if (r == OSOT_OK) r = ensure_lds(osot_cycle_kernel<NP, true>, lds);
        if (r == OSOT_OK) r = ensure_lds(osot_cascade_kernel<NP, false, true>, lds);
        if constexpr (NP != 32) {
            if (r == OSOT_OK) r = ensure_lds(osot_cascade_kernel<NP, false, false, true>, lds);
            if (r == OSOT_OK) r = ensure_lds(osot_cycle_kernel<NP, false, true>, lds);
            if (r == OSOT_OK) r = ensure_lds(osot_control_cycle_kernel<NP, false, true>, lds > control_kin_lds_bytes(NP) ? lds : control_kin_lds_bytes(NP));
            if (r == OSOT_OK) r = ensure_lds(osot_control_cycle_kernel<NP, false, true, true>, lds > control_kin_lds_bytes(NP) ? lds : control_kin_lds_bytes(NP));
            if (r == OSOT_OK) r = ensure_lds(osot_control_cycle_kernel<NP, false, false>, lds > control_kin_lds_bytes(NP) ? lds : control_kin_lds_bytes(NP));
            if (r == OSOT_OK) r = ensure_lds(osot_control_cycle_kernel<NP, false, false, true>, lds > control_kin_lds_bytes(NP) ? lds : control_kin_lds_bytes(NP));
            if (r == OSOT_OK) r = ensure_lds(osot_control_cycle_kernel<NP, true, false>, lds > control_kin_lds_bytes(NP) ? lds : control_kin_lds_bytes(NP));
            if (r == OSOT_OK) r = ensure_lds(osot_control_cycle_kernel<NP, true, false, true>, lds > control_kin_lds_bytes(NP) ? lds : control_kin_lds_bytes(NP));
        }
        if constexpr (NP == 32) {
            if (r == OSOT_OK) r = ensure_lds(osot_cascade_kernel<32, false, false, true>, lds);
            if (r == OSOT_OK) r = ensure_lds(osot_cycle_kernel<32, false, true>, lds);
            if (r == OSOT_OK) r = ensure_lds(osot_control_cycle_kernel<32, false, true>, lds > control_kin_lds_bytes(32) ? lds : control_kin_lds_bytes(32));
            if (r == OSOT_OK) r = ensure_lds(osot_control_cycle_kernel<32, false, true, true>, lds > control_kin_lds_bytes(32) ? lds : control_kin_lds_bytes(32));
            if (r == OSOT_OK) r = ensure_lds(osot_control_cycle_kernel<32, false, false>, lds > control_kin_lds_bytes(32) ? lds : control_kin_lds_bytes(32));
            if (r == OSOT_OK) r = ensure_lds(osot_control_cycle_kernel<32, false, false, true>, lds > control_kin_lds_bytes(32) ? lds : control_kin_lds_bytes(32));
            if (r == OSOT_OK) r = ensure_lds(osot_control_cycle_kernel<32, true, false>, lds > control_kin_lds_bytes(32) ? lds : control_kin_lds_bytes(32));
            if (r == OSOT_OK) r = ensure_lds(osot_control_cycle_kernel<32, true, false, true>, lds > control_kin_lds_bytes(32) ? lds : control_kin_lds_bytes(32));
            if (r == OSOT_OK) r = ensure_lds(osot_cascade_kernel<32, true, false, true>, lds);
        }
        return r;
    });
    if (rc != OSOT_OK) return rc;
    osot_solver* s = new osot_solver();
    s->plan = *plan;
    s->max_batch = max_batch;
    s->device = device;
    s->timing = false;
    std::memset(s->task_active, 1, sizeof(s->task_active));
    {   // resident workgroups: what the longest-first dispatch order is planned for (order_body)
        int per_cu = 0, cus = 0;
        hipError_t e1 = by_np(T, [&](auto np) {
            return hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, osot_cycle_kernel<decltype(np)::value, false>, 64, lds);
        });
        hipError_t e2 = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
        s->slots = (e1 == hipSuccess && e2 == hipSuccess && per_cu > 0 && cus > 0) ? per_cu * cus : 2048;
    }
    if (P.rows_in_global && hipMalloc(&s->d_rows, sizeof(double) * (size_t)max_batch * (size_t)P.rows_doubles) != hipSuccess) {
        delete s;
        return fail(OSOT_ERR_HIP, "device allocation for the row tables failed");
    }
    make_update_plan(*plan, s->h_uplan);
    // dispatch-order state and the static update plan live with the solver from the start (no lazy allocation on
    // whatever device is current)
    if (hipMalloc(&s->d_cost, sizeof(int) * 2 * (size_t)max_batch) != hipSuccess ||
        hipMemset(s->d_cost, 0, sizeof(int) * 2 * (size_t)max_batch) != hipSuccess ||
        hipMalloc(&s->d_order, sizeof(int) * 2 * (size_t)max_batch) != hipSuccess ||
        [&] {   // both halves of the order buffer start as the identity permutation: a half that no launch has written yet
                // (a graph capture that failed half-way leaves the host-side flip ahead of the device) is still a valid order
            std::vector<int> id(2 * (size_t)max_batch);
            for (int i = 0; i < 2 * max_batch; ++i) id[i] = i % max_batch;
            return hipMemcpy(s->d_order, id.data(), sizeof(int) * id.size(), hipMemcpyHostToDevice) != hipSuccess;
        }() ||
        hipMalloc(&s->d_uplan, sizeof(DevUpdatePlan)) != hipSuccess ||
        hipMemcpy(s->d_uplan, &s->h_uplan, sizeof(DevUpdatePlan), hipMemcpyHostToDevice) != hipSuccess) {
        if (s->d_cost) hipFree(s->d_cost);
        if (s->d_order) hipFree(s->d_order);
        if (s->d_uplan) hipFree(s->d_uplan);
        delete s;
        return fail(OSOT_ERR_HIP, "device allocation for the solver failed");
    }
    *out = s;
    return OSOT_OK;
}

int osot_solver_destroy(osot_solver* s) {
    if (!s) return OSOT_OK;
    DeviceGuard guard(s->device);
    for (auto& p : s->events) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    for (auto& p : s->pool) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    if (s->d_cost) hipFree(s->d_cost);
    if (s->d_order) hipFree(s->d_order);
    if (s->d_uplan) hipFree(s->d_uplan);
    if (s->d_hot) hipFree(s->d_hot);
    if (s->d_rows) hipFree(s->d_rows);
    if (s->nhqp_ready) {
        void* ptrs[] = {s->nhqp.N[0], s->nhqp.N[1], s->nhqp.q0, s->nhqp.H, s->nhqp.g, s->nhqp.R, s->nhqp.rlo, s->nhqp.rup, s->nhqp.z,
                        s->nhqp.V2, s->nhqp.qp_status, s->nhqp.qp_iters};
        for (void* q : ptrs) if (q) hipFree(q);
    }
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

int osot_solver_set_specialisation(osot_solver* s, int enabled) {
    if (!s) return fail(OSOT_ERR_INVALID, "null solver");
    s->specialise = enabled ? 1 : 0;
    return OSOT_OK;
}

int osot_solver_set_hotstart(osot_solver* s, int enabled) {
    if (!s) return fail(OSOT_ERR_INVALID, "null solver");
    DeviceGuard guard(s->device);
    if (!guard.ok) return fail(OSOT_ERR_HIP, "hipSetDevice failed");
    if (enabled) {
        DevPlan P; int T; size_t lds;
        make_dev_plan(s->plan, nullptr, P, T, lds);
        const size_t bytes = sizeof(int) * (size_t)s->max_batch * (size_t)s->plan.n_levels * (size_t)(T <= 32 ? 32 : 64);   // (WaveCtx::LW entries per level)
        if (!s->d_hot) {
            if (hipMalloc(&s->d_hot, bytes) != hipSuccess) { s->d_hot = nullptr; return fail(OSOT_ERR_HIP, "device allocation for the hot-start state failed"); }
            s->hot_T = T;
        }
        // (re-)enabling forgets what was recorded: every instance starts cold once.  Device-wide synchronous, like the
        // allocation: this is a configuration call, not part of a control cycle
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemset(s->d_hot, 0xff, bytes));
    }
    s->hotstart = enabled ? 1 : 0;
    return OSOT_OK;
}

int osot_solver_set_task_active(osot_solver* s, int level, int task, int active) {
    if (!s) return fail(OSOT_ERR_INVALID, "null solver");
    if (level < 0 || level >= s->plan.n_levels || task < 0 || task >= s->plan.level[level].n_tasks)
        return fail(OSOT_ERR_INVALID, "no such task");
    s->task_active[level * OSOT_MAX_TASKS + task] = active ? 1 : 0;
    s->any_inactive = false;
    for (int k = 0; k < s->plan.n_levels; ++k)
        for (int j = 0; j < s->plan.level[k].n_tasks; ++j)
            if (!s->task_active[k * OSOT_MAX_TASKS + j]) s->any_inactive = true;
    return OSOT_OK;
}

int osot_solver_set_timing(osot_solver* s, int enabled) {
    if (!s) return fail(OSOT_ERR_INVALID, "null solver");
    s->timing = enabled != 0;
    s->timing_stride = enabled > 1 ? enabled : 1;
    s->timing_count = 0;
    if (s->timing) {   // the event pairs exist before the first timed launch: creating them costs more than recording them
        DeviceGuard guard(s->device);
        if (!guard.ok) return fail(OSOT_ERR_HIP, "hipSetDevice failed");
        while (s->pool.size() < 64) {
            std::pair<hipEvent_t, hipEvent_t> ev;
            HIP_TRY(hipEventCreate(&ev.first));
            HIP_TRY(hipEventCreate(&ev.second));
            s->pool.push_back(ev);
        }
    }
    return OSOT_OK;
}

int osot_ehqp_solve(osot_solver* s, const osot_qp_batch* b, double sigma_min, void* hip_stream) {
    if (!s || !b) return fail(OSOT_ERR_INVALID, "null solver/batch");
    if (b->B < 0 || b->B > s->max_batch) return fail(OSOT_ERR_INVALID, "batch size exceeds max_batch");
    if (b->B == 0) return OSOT_OK;
    if (!b->dq || !b->status) return fail(OSOT_ERR_INVALID, "dq/status output is null");
    const osot_plan_desc& pl = s->plan;
    DevEhqp Q;
    const char* why = "";
    int rc = ehqp_args(pl, b, sigma_min, s->any_inactive ? s->task_active : nullptr, Q, &why);
    if (rc != OSOT_OK) return fail(rc, why);
    DeviceGuard guard(s->device);
    if (!guard.ok) return fail(OSOT_ERR_HIP, "hipSetDevice failed");
    if (Q.use_qr) {
        const int NPq = Q.n <= 32 ? 32 : 64;
        const size_t lds = ehqp_qr_lds_bytes(NPq, Q.rows8);
        rc = NPq == 32 ? ensure_lds(osot_ehqp_qr_kernel<32>, lds) : ensure_lds(osot_ehqp_qr_kernel<64>, lds);
        if (rc != OSOT_OK) return rc;
        if (NPq == 32) hipLaunchKernelGGL(osot_ehqp_qr_kernel<32>, dim3((unsigned)b->B), dim3(64), lds, (hipStream_t)hip_stream, Q);
        else hipLaunchKernelGGL(osot_ehqp_qr_kernel<64>, dim3((unsigned)b->B), dim3(64), lds, (hipStream_t)hip_stream, Q);
    } else {
        hipLaunchKernelGGL(osot_ehqp_kernel, dim3((unsigned)b->B), dim3(64), 0, (hipStream_t)hip_stream, Q);
    }
    HIP_TRY(hipGetLastError());
    return OSOT_OK;
}

int osot_nhqp_solve(osot_solver* s, const osot_qp_batch* b, const osot_nhqp_options* opt, void* hip_stream) {
    if (!s || !b) return fail(OSOT_ERR_INVALID, "null solver/batch");
    if (b->B < 0 || b->B > s->max_batch) return fail(OSOT_ERR_INVALID, "batch size exceeds max_batch");
    if (b->B == 0) return OSOT_OK;
    if (!b->dq || !b->status) return fail(OSOT_ERR_INVALID, "dq/status output is null");
    DeviceGuard guard(s->device);
    if (!guard.ok) return fail(OSOT_ERR_HIP, "hipSetDevice failed");
    const osot_plan_desc& pl = s->plan;
    if (b->level_active)     // (iHQP::setActiveStack: the reference's nHQP has no such switch, nHQP.h)
        for (int k = 0; k < pl.n_levels; ++k)
            if (!b->level_active[k]) return fail(OSOT_ERR_UNSUPPORTED, "nHQP front-end: level_active is iHQP's setActiveStack; nHQP has no such switch");
    {
        int fv[OSOT_MAX_LEVELS]; const char* why = "";
        int rc = nhqp_validate(pl, opt, fv, &why);
        if (rc != OSOT_OK) return fail(rc, why);
        for (int k = 0; k < pl.n_levels; ++k) {
            int m, ma; plan_level_rows(&pl, k, &m, &ma);
            if (ma > 0 && !b->A[k]) return fail(OSOT_ERR_INVALID, "A[k] is null for a level with stored rows");
            if (!b->b[k]) return fail(OSOT_ERR_INVALID, "b[k] is null");
        }
        int nc = 0; plan_constraint_rows(&pl, &nc);
        if (nc > 0 && (!b->C || !b->lo || !b->up)) return fail(OSOT_ERR_INVALID, "plan has constraint rows but C/lo/up is null");
        if (pl.n_bounds > 0 && (!b->l || !b->u)) return fail(OSOT_ERR_INVALID, "plan has bounds but l/u is null");
    }
    if (!s->nhqp_ready) {
        const NhqpSizes z = nhqp_sizes(pl, s->max_batch);
        NhqpWorkspace& w = s->nhqp;
        std::memset(&w, 0, sizeof(w));
        auto dm = [](double** p, size_t cnt) { return hipMalloc((void**)p, sizeof(double) * (cnt ? cnt : 1)) == hipSuccess; };
        bool ok = dm(&w.N[0], z.N) && dm(&w.N[1], z.N) && dm(&w.q0, z.q0) && dm(&w.H, z.H) && dm(&w.g, z.g) && dm(&w.R, z.R) &&
                  dm(&w.rlo, z.rl) && dm(&w.rup, z.rl) && dm(&w.z, z.z) && dm(&w.V2, z.V2) &&
                  hipMalloc((void**)&w.qp_status, sizeof(int) * z.st) == hipSuccess && hipMalloc((void**)&w.qp_iters, sizeof(int) * z.st) == hipSuccess;
        if (!ok) return fail(OSOT_ERR_HIP, "device allocation for the nHQP workspace failed");
        s->nhqp_ready = true;
    }
    if (pl.n > 32) {    // the 64-column preparation kernels' dynamic LDS, once per solver and with its error reported (a
                        // failed attribute call used to skip the launch silently and the solve went on with a stale workspace)
        if (!s->nhqp_lds_ready) {
            int r = ensure_lds(osot_nhqp_prepare64_kernel<32>, nhqp_prepare64_lds_bytes(32, pl.n));
            if (r == OSOT_OK) r = ensure_lds(osot_nhqp_prepare64_kernel<64>, nhqp_prepare64_lds_bytes(64, pl.n));
            if (r == OSOT_OK) r = ensure_lds(osot_nhqp_prepare_wide_kernel, nhqp_prepare_wide_lds_bytes(64, pl.n));
            if (r != OSOT_OK) return r;
            s->nhqp_lds_ready = true;
        }
    }
    hipStream_t st = (hipStream_t)hip_stream;
    const unsigned grid = (unsigned)b->B;
    const char* why = "";
    int rc = nhqp_run(pl, b, opt, s->nhqp,
        [&](const DevNhqp& Q) {
            if (nhqp_level_is_wide(Q.m, Q.nf))      // min(rows, free variables) > 32: the full column-side Gram matrix (sym_eig_wide)
                hipLaunchKernelGGL(osot_nhqp_prepare_wide_kernel, dim3(grid), dim3(64), nhqp_prepare_wide_lds_bytes(Q.m, Q.n), st, Q);
            else if (Q.n > 32) {     // 33 .. 64 variables: the 64-column kernel (dynamic LDS beyond the 64 KB default: raised once, below)
                if (Q.m <= 32) hipLaunchKernelGGL(osot_nhqp_prepare64_kernel<32>, dim3(grid), dim3(64), nhqp_prepare64_lds_bytes(32, Q.n), st, Q);
                else hipLaunchKernelGGL(osot_nhqp_prepare64_kernel<64>, dim3(grid), dim3(64), nhqp_prepare64_lds_bytes(64, Q.n), st, Q);
            }
            else if (Q.m <= 32) hipLaunchKernelGGL(osot_nhqp_prepare_kernel<32>, dim3(grid), dim3(64), 0, st, Q);
            else hipLaunchKernelGGL(osot_nhqp_prepare_kernel<64>, dim3(grid), dim3(64), 0, st, Q);
        },
        [&](int B, int n, int nc, const double* H, const double* g, const double* A, const double* lA, const double* uA,
            const double* l, const double* u, double eps, double* x, int* status, int* iters) {
            return osot_qp_solve_batch(B, n, nc, H, g, A, lA, uA, l, u, eps, 0, x, status, iters, hip_stream);
        },
        [&](const DevNhqpAcc& A) { hipLaunchKernelGGL(osot_nhqp_accumulate_kernel, dim3(grid), dim3(64), 0, st, A); }, &why,
        s->any_inactive ? s->task_active : nullptr);
    if (rc != OSOT_OK) return fail(rc, why);
    HIP_TRY(hipGetLastError());
    return OSOT_OK;
}

int osot_solver_resident_waves(osot_solver* s, int* waves) {
    if (!s || !waves) return fail(OSOT_ERR_INVALID, "null argument");
    *waves = s->slots;
    return OSOT_OK;
}

// the same for the null-space front-end: its level-preparation kernels are what a solve's time goes to, and the one with the fewest
// resident wavefronts sets the round (static 25 KB of LDS at n <= 32: six per CU; the 64-column and wide kernels by the plan's sizes)
int osot_solver_resident_waves_nhqp(osot_solver* s, const osot_nhqp_options* opt, int* waves) {
    if (!s || !waves) return fail(OSOT_ERR_INVALID, "null argument");
    DeviceGuard guard(s->device);
    if (!guard.ok) return fail(OSOT_ERR_HIP, "hipSetDevice failed");
    const osot_plan_desc& pl = s->plan;
    int fv[OSOT_MAX_LEVELS]; const char* why = "";
    int rc = nhqp_validate(pl, opt, fv, &why);
    if (rc != OSOT_OK) return fail(rc, why);
    int cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s->device));
    int least = 1 << 30;
    for (int k = 0; k < pl.n_levels; ++k) {
        int m, ma; plan_level_rows(&pl, k, &m, &ma);
        int per_cu = 0;
        hipError_t e;
        if (nhqp_level_is_wide(m, fv[k])) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, osot_nhqp_prepare_wide_kernel, 64, nhqp_prepare_wide_lds_bytes(m, pl.n));
        else if (pl.n > 32) {
            if (m <= 32) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, osot_nhqp_prepare64_kernel<32>, 64, nhqp_prepare64_lds_bytes(32, pl.n));
            else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, osot_nhqp_prepare64_kernel<64>, 64, nhqp_prepare64_lds_bytes(64, pl.n));
        }
        else if (m <= 32) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, osot_nhqp_prepare_kernel<32>, 64, 0);
        else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, osot_nhqp_prepare_kernel<64>, 64, 0);
        if (e != hipSuccess || per_cu < 1) return fail(OSOT_ERR_HIP, "occupancy query of a preparation kernel failed");
        if (per_cu * cus < least) least = per_cu * cus;
    }
    *waves = least;
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

static int ihqp_launch(osot_solver* s, const osot_qp_batch* b, void* hip_stream, long long* prof, const DevUpdate* fused = nullptr,
                       const DevControl* control = nullptr);

int osot_ihqp_solve(osot_solver* s, const osot_qp_batch* b, void* hip_stream) {
    return ihqp_launch(s, b, hip_stream, nullptr);
}

int osot_cycle(osot_solver* s, const osot_leaf_batch* leaf, const osot_assembled_out* out, const osot_qp_batch* b,
               void* hip_stream) {
    if (!s || !leaf || !out || !b) return fail(OSOT_ERR_INVALID, "null argument");
    if (leaf->B != b->B) return fail(OSOT_ERR_INVALID, "leaf and batch disagree on B");
    if (leaf->B < 0 || leaf->B > s->max_batch) return fail(OSOT_ERR_INVALID, "batch size exceeds max_batch");
    if (leaf->B == 0) return OSOT_OK;
    DevUpdate U;
    const char* why = "";
    int rc = make_update_args(s->plan, s->h_uplan, leaf, out, s->d_uplan, U, &why);
    if (rc != OSOT_OK) return fail(rc, why);
    // developer knob: OSOT_DEBUG_CYCLE_PROF=<device pointer, hex> receives [B][2] int64 cycle counts (update, cascade)
    static const char* dbg = getenv("OSOT_DEBUG_CYCLE_PROF");
    return ihqp_launch(s, b, hip_stream, dbg ? (long long*)strtoull(dbg, nullptr, 16) : nullptr, &U);
}

int osot_solver_profile_phases(osot_solver* s, const osot_qp_batch* b, long long* cycles, void* hip_stream) {
    if (!cycles) return fail(OSOT_ERR_INVALID, "null cycles");
    return ihqp_launch(s, b, hip_stream, cycles);
}

static int ihqp_launch(osot_solver* s, const osot_qp_batch* b, void* hip_stream, long long* prof, const DevUpdate* fused, const DevControl* control) {
    if (!s || !b) return fail(OSOT_ERR_INVALID, "null solver/batch");
    if (b->B < 0 || b->B > s->max_batch) return fail(OSOT_ERR_INVALID, "batch size exceeds max_batch");
    if (b->B == 0) return OSOT_OK;   // empty batch: nothing to do
    DeviceGuard guard(s->device);
    if (!guard.ok) return fail(OSOT_ERR_HIP, "hipSetDevice failed");
    const osot_plan_desc& pl = s->plan;
    DevPlan P; int T; size_t lds;
    make_dev_plan(pl, b->level_active, P, T, lds, s->any_inactive ? s->task_active : nullptr);
    {   // developer knob: extra dynamic LDS per wave (lowers the occupancy; used to measure how the kernel responds to it)
        static const char* extra = getenv("OSOT_DEBUG_EXTRA_LDS");
        if (extra) lds += (size_t)atoi(extra);
    }
    if (control && lds < control_kin_lds_bytes(T)) lds = control_kin_lds_bytes(T);
    DevBatch D;
    std::memset(&D, 0, sizeof(D));
    D.B = b->B;
    for (int k = 0; k < pl.n_levels; ++k) {
        if (P.ma[k] > 0 && !b->A[k]) return fail(OSOT_ERR_INVALID, "A[k] is null for a level with stored rows");
        if (!b->b[k]) return fail(OSOT_ERR_INVALID, "b[k] is null");
        D.A[k] = b->A[k]; D.b[k] = b->b[k]; D.w[k] = b->w[k]; D.c[k] = b->c[k];
        if (s->h_uplan.dense_level[k]) {
            if (!b->WA[k] || !b->Wb[k]) return fail(OSOT_ERR_INVALID, "level has a non-diagonal weight but WA[k] / Wb[k] is null");
            D.WA[k] = b->WA[k]; D.Wb[k] = b->Wb[k];
        }
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
    if (pl.has_regularisation && pl.regularisation_dense) {
        if (!b->A_reg) return fail(OSOT_ERR_INVALID, "the regularisation task has a stored Jacobian but A_reg is null");
        D.A_reg = b->A_reg;
    }
    D.prof = prof;
    D.accepted_slack = b->accepted_slack;
    D.hot = (s->hotstart && !prof) ? s->d_hot : nullptr;
    D.rows_scratch = s->d_rows;
    hipStream_t st = (hipStream_t)hip_stream;
    if (s->schedule == 1) {
        if (s->order_B >= 0 && s->order_stream != st) HIP_TRY(hipStreamSynchronize(s->order_stream));
        const size_t mb = (size_t)s->max_batch;
        D.order = (s->order_B == b->B) ? s->d_order + (size_t)(s->flip ^ 1) * mb : nullptr;
        D.cost_in = s->d_cost + (size_t)(s->flip ^ 1) * mb;
        D.cost_out = s->d_cost + (size_t)s->flip * mb;
        D.order_next = s->d_order + (size_t)s->flip * mb;
        D.slots = s->slots;
    }
    const unsigned grid = (unsigned)b->B + (D.order_next ? 1u : 0u);   // (+ the order workgroup, block 0)
    // the instantiation with the dense-weight / inactive-task code only where the plan or the solver state asks for it
    bool extra = s->any_inactive || (pl.has_regularisation && pl.regularisation_dense);
    for (int k = 0; k < pl.n_levels; ++k) extra = extra || (s->h_uplan.dense_level[k] != 0);
    if (prof && !fused && extra)   // (the instrumented instantiation carries no dense-weight / inactive-task code: it would
        return fail(OSOT_ERR_UNSUPPORTED, "phase profiling is not available for plans with dense weights or inactive tasks");   // solve another problem)
    extra = extra || (D.hot != nullptr);
    {   // developer knob: the EXTRA instantiation for every launch (A/B of the two register allocations)
        static const char* force = getenv("OSOT_DEBUG_FORCE_EXTRA");
        if (force && force[0] == '1' && !prof) extra = true;
    }   // (hot start: the EXTRA instantiation carries its code; never together with prof, see D.hot)
    // plans without constraint rows (the bounds are the only inequalities): the BOX instantiation of the 32-column kernels
    // ... and plans whose rows are all equalities by construction (the feet as TaskToConstraint rows: the reference's COMAN stacks), at
    // every size -- but only where THIS launch's update half writes lo == up itself (fused): on the solve-only path lo / up are the
    // caller's arrays, and a row with lo < up would never be scanned by the BOX instantiation
    bool box = s->specialise && !extra && (P.nc == 0 ? T == 32 : (fused != nullptr && plan_rows_all_equalities(pl)));
    // the BOX instantiations of osot_control_cycle_kernel are compiled WITHOUT the collision-pair stage of the kinematics (PAIRS = !BOX):
    // a model with pairs whose batch asks for their outputs takes the general instantiation, whatever the plan's rows (such a
    // launch used to leave pair_dist / pair_J unwritten without an error)
    if (control && control->K_pairs > 0 && (control->Bt.pair_dist || control->Bt.pair_J)) box = false;
    if (control && (prof || !fused))
        return fail(OSOT_ERR_UNSUPPORTED, "the fused control cycle carries no profiling code (use osot_kinematics + osot_cycle)");
    std::pair<hipEvent_t, hipEvent_t> ev;
    const bool timed = s->timing && (s->timing_count++ % s->timing_stride) == 0;   // (after EVERY early return: nothing is taken from the pool for a launch that does not happen)
    if (timed) {
        if (!s->pool.empty()) { ev = s->pool.back(); s->pool.pop_back(); }
        else { HIP_TRY(hipEventCreate(&ev.first)); HIP_TRY(hipEventCreate(&ev.second)); }
        HIP_TRY(hipEventRecord(ev.first, st));
    }
    const bool roll = control && (control->steps > 1 || control->dq_steps || control->status_steps);
    by_np(T, [&](auto np) {
        constexpr int NP = decltype(np)::value;
        if constexpr (NP == 32) {
            if (control && roll) {
                if (box) hipLaunchKernelGGL((osot_control_cycle_kernel<32, false, true, true>), dim3(grid), dim3(64), lds, st, *fused, P, D, *control);
                else if (extra) hipLaunchKernelGGL((osot_control_cycle_kernel<32, true, false, true>), dim3(grid), dim3(64), lds, st, *fused, P, D, *control);
                else hipLaunchKernelGGL((osot_control_cycle_kernel<32, false, false, true>), dim3(grid), dim3(64), lds, st, *fused, P, D, *control);
                return 0;
            }
            if (control) {
                if (box) hipLaunchKernelGGL((osot_control_cycle_kernel<32, false, true>), dim3(grid), dim3(64), lds, st, *fused, P, D, *control);
                else if (extra) hipLaunchKernelGGL((osot_control_cycle_kernel<32, true, false>), dim3(grid), dim3(64), lds, st, *fused, P, D, *control);
                else hipLaunchKernelGGL((osot_control_cycle_kernel<32, false, false>), dim3(grid), dim3(64), lds, st, *fused, P, D, *control);
                return 0;
            }
            if (box) {
                if (fused) hipLaunchKernelGGL((osot_cycle_kernel<32, false, true>), dim3(grid), dim3(64), lds, st, *fused, P, D);
                else if (prof) hipLaunchKernelGGL((osot_cascade_kernel<32, true, false, true>), dim3(grid), dim3(64), lds, st, P, D);
                else hipLaunchKernelGGL((osot_cascade_kernel<32, false, false, true>), dim3(grid), dim3(64), lds, st, P, D);
                return 0;
            }
        }
        if constexpr (NP != 32) {
            if (control && roll) {
                if (box) hipLaunchKernelGGL((osot_control_cycle_kernel<NP, false, true, true>), dim3(grid), dim3(64), lds, st, *fused, P, D, *control);
                else if (extra) hipLaunchKernelGGL((osot_control_cycle_kernel<NP, true, false, true>), dim3(grid), dim3(64), lds, st, *fused, P, D, *control);
                else hipLaunchKernelGGL((osot_control_cycle_kernel<NP, false, false, true>), dim3(grid), dim3(64), lds, st, *fused, P, D, *control);
                return 0;
            }
            if (control) {
                if (box) hipLaunchKernelGGL((osot_control_cycle_kernel<NP, false, true>), dim3(grid), dim3(64), lds, st, *fused, P, D, *control);
                else if (extra) hipLaunchKernelGGL((osot_control_cycle_kernel<NP, true, false>), dim3(grid), dim3(64), lds, st, *fused, P, D, *control);
                else hipLaunchKernelGGL((osot_control_cycle_kernel<NP, false, false>), dim3(grid), dim3(64), lds, st, *fused, P, D, *control);
                return 0;
            }
            if (box && !prof) {
                if (fused) hipLaunchKernelGGL((osot_cycle_kernel<NP, false, true>), dim3(grid), dim3(64), lds, st, *fused, P, D);
                else hipLaunchKernelGGL((osot_cascade_kernel<NP, false, false, true>), dim3(grid), dim3(64), lds, st, P, D);
                return 0;
            }
        }
        if (fused) {
            if (extra) hipLaunchKernelGGL((osot_cycle_kernel<NP, true>), dim3(grid), dim3(64), lds, st, *fused, P, D);
            else hipLaunchKernelGGL((osot_cycle_kernel<NP, false>), dim3(grid), dim3(64), lds, st, *fused, P, D);
        } else if (extra && !prof) {
            hipLaunchKernelGGL((osot_cascade_kernel<NP, false, true>), dim3(grid), dim3(64), lds, st, P, D);
        } else if (prof) {
            hipLaunchKernelGGL((osot_cascade_kernel<NP, true>), dim3(grid), dim3(64), lds, st, P, D);
        } else {
            hipLaunchKernelGGL((osot_cascade_kernel<NP, false>), dim3(grid), dim3(64), lds, st, P, D);
        }
        return 0;
    });
    HIP_TRY(hipGetLastError());
    if (timed) {
        HIP_TRY(hipEventRecord(ev.second, st));
        s->events.push_back(ev);
    }
    if (s->schedule == 1) {   // this launch's order workgroup has written the order for the next solve of a batch of this size
        s->flip ^= 1;
        s->order_B = b->B;
        s->order_stream = st;
    }
    return OSOT_OK;
}

int osot_stack_update(osot_solver* s, const osot_leaf_batch* leaf, const osot_assembled_out* out, void* hip_stream) {
    if (!s || !leaf || !out) return fail(OSOT_ERR_INVALID, "null argument");
    if (leaf->B < 0 || leaf->B > s->max_batch) return fail(OSOT_ERR_INVALID, "batch size exceeds max_batch");
    if (leaf->B == 0) return OSOT_OK;
    DeviceGuard guard(s->device);
    if (!guard.ok) return fail(OSOT_ERR_HIP, "hipSetDevice failed");
    const osot_plan_desc& pl = s->plan;
    const DevUpdatePlan& PL = s->h_uplan;
    DevUpdate U;
    const char* why = "";
    int rc = make_update_args(pl, PL, leaf, out, s->d_uplan, U, &why);
    if (rc != OSOT_OK) return fail(rc, why);
    hipLaunchKernelGGL(osot_update_kernel, dim3((unsigned)leaf->B), dim3(64), 0, (hipStream_t)hip_stream, U);
    HIP_TRY(hipGetLastError());
    return OSOT_OK;
}

static int qp_solve_batch_impl(int B, int n, int nc, const double* H, const double* g, const double* A,
                               const double* lA, const double* uA, const double* l, const double* u,
                               double eps_abs, int max_iter, double* x, int* status, int* iterations,
                               void* hip_stream, int* hot);

int osot_qp_solve_batch(int B, int n, int nc, const double* H, const double* g, const double* A,
                        const double* lA, const double* uA, const double* l, const double* u,
                        double eps_abs, int max_iter, double* x, int* status, int* iterations,
                        void* hip_stream) {
    return qp_solve_batch_impl(B, n, nc, H, g, A, lA, uA, l, u, eps_abs, max_iter, x, status, iterations, hip_stream, nullptr);
}

// hot: [B][32 or 64] device ints (see DevQP.hot) or null
static int qp_solve_batch_impl(int B, int n, int nc, const double* H, const double* g, const double* A,
                               const double* lA, const double* uA, const double* l, const double* u,
                               double eps_abs, int max_iter, double* x, int* status, int* iterations,
                               void* hip_stream, int* hot) {
    if (B < 0 || n < 1 || n > OSOT_MAX_QP_VARS || nc < 0) return fail(OSOT_ERR_INVALID, "bad sizes");
    if (B == 0) return OSOT_OK;
    if (!H || !g || !x || !status) return fail(OSOT_ERR_INVALID, "null H/g/x/status");
    if (nc > 0 && (!A || !lA || !uA)) return fail(OSOT_ERR_INVALID, "nc > 0 but A/lA/uA is null");
    if ((l == nullptr) != (u == nullptr)) return fail(OSOT_ERR_INVALID, "l and u must both be given or both be null");
    if (n > OSOT_MAX_VARS) {
        // wider than a wavefront (65 .. 128 variables): one 256-thread workgroup per QP (osot_qp_big.h).  Cold start (the hot-start
        // record of the plugin route is not used), stream-ordered workspace: not for HIP graph capture.
        if (nc > big::kMaxRows) return fail(OSOT_ERR_UNSUPPORTED, "more than 2048 constraint rows with more than 64 variables");
        DevQPBig Q;
        std::memset(&Q, 0, sizeof(Q));
        Q.B = B; Q.n = n; Q.nc = nc;
        Q.max_iter = max_iter > 0 ? max_iter : 20 * (n + nc) + 100;
        Q.eps_abs = eps_abs;
        Q.H = H; Q.g = g; Q.A = A; Q.lA = lA; Q.uA = uA; Q.l = l; Q.u = u;
        Q.x = x; Q.status = status; Q.iterations = iterations;
        const size_t lds = big::shared_bytes(n, nc);
        int r = ensure_lds(osot_qp_big_kernel, lds);
        if (r != OSOT_OK) return r;
        // (workgroups in flight: 30 KB of LDS at n = 70 lets five share a CU, 82 KB at n = 128 one; the workspace is 2 n^2 doubles each)
        const unsigned cap = n <= 96 ? 1024u : 512u;
        const unsigned grid = (unsigned)B < cap ? (unsigned)B : cap;
        const size_t wbytes = (size_t)grid * 2 * (size_t)n * n * sizeof(double);
        HIP_TRY(hipMallocAsync((void**)&Q.work, wbytes, (hipStream_t)hip_stream));
        hipLaunchKernelGGL(osot_qp_big_kernel, dim3(grid), dim3(256), lds, (hipStream_t)hip_stream, Q);
        const hipError_t le = hipGetLastError();
        HIP_TRY(hipFreeAsync(Q.work, (hipStream_t)hip_stream));
        HIP_TRY(le);
        return OSOT_OK;
    }
    DevQP Q;
    std::memset(&Q, 0, sizeof(Q));
    Q.B = B; Q.n = n; Q.nc = nc;
    Q.max_iter = max_iter > 0 ? max_iter : 20 * (n + nc) + 100;
    Q.eps_abs = eps_abs;
    Q.H = H; Q.g = g; Q.A = A; Q.lA = lA; Q.uA = uA; Q.l = l; Q.u = u;
    Q.x = x; Q.status = status; Q.iterations = iterations;
    // the lane layout of the cascade kernels (make_dev_plan): 32, 40 (33..38 variables, two wavefronts per SIMD), 56, 64
#ifdef OSOT_X_NO_NP40
    const int T = n <= 32 ? 32 : 64;
#else
    const int T = n <= 32 ? 32 : (n <= WaveCtx<40>::NMAX ? 40 : (n <= WaveCtx<56>::NMAX ? 56 : 64));
#endif
    const size_t lds = (size_t)lds_layout(T, nc, &Q.lds_rows_off, &Q.lds_rows_cap) * sizeof(double);
    Q.hot = hot;
    const unsigned grid = (unsigned)B;
    const int rc = by_np(T, [&](auto np) -> int {
        constexpr int NP = decltype(np)::value;
        if (hot) {
            const int r = ensure_lds(osot_qp_kernel<NP, true>, lds);
            if (r != OSOT_OK) return r;
            hipLaunchKernelGGL((osot_qp_kernel<NP, true>), dim3(grid), dim3(64), lds, (hipStream_t)hip_stream, Q);
        } else {
            const int r = ensure_lds(osot_qp_kernel<NP, false>, lds);
            if (r != OSOT_OK) return r;
            hipLaunchKernelGGL((osot_qp_kernel<NP, false>), dim3(grid), dim3(64), lds, (hipStream_t)hip_stream, Q);
        }
        return OSOT_OK;
    });
    if (rc != OSOT_OK) return rc;
    HIP_TRY(hipGetLastError());
    return OSOT_OK;
}

int osot_qp_solve_batch_admm_warm(int B, int n, int nc, const double* H, const double* g, const double* A,
                                  const double* lA, const double* uA, const double* l, const double* u,
                                  double eps_reg, const osot_admm_options* opt, double* warm_x, double* warm_y, double* warm_rho,
                                  double* x, int* status, int* iterations, void* hip_stream) {
    if (B < 0 || n < 1 || n > OSOT_MAX_VARS || nc < 0) return fail(OSOT_ERR_INVALID, "bad sizes");
    if (B == 0) return OSOT_OK;
    if (!H || !g || !x || !status) return fail(OSOT_ERR_INVALID, "null H/g/x/status");
    if (nc > 0 && (!A || !lA || !uA)) return fail(OSOT_ERR_INVALID, "nc > 0 but A/lA/uA is null");
    if ((l == nullptr) != (u == nullptr)) return fail(OSOT_ERR_INVALID, "l and u must both be given or both be null");
    const bool has_y = (nc + (l ? n : 0)) > 0;
    if ((warm_x == nullptr) != (warm_rho == nullptr) || (warm_x && has_y && !warm_y) || (!warm_x && warm_y))
        return fail(OSOT_ERR_INVALID, "warm_x, warm_y and warm_rho must be given together (or all be null)");
    const DevAdmm Q = admm_args(B, n, nc, H, g, A, lA, uA, l, u, eps_reg, opt, warm_x, warm_y, warm_rho, x, status, iterations);
    const size_t lds = admm_lds_bytes(n, nc, l != nullptr);
    int rc = ensure_lds(osot_admm_kernel, lds);
    if (rc != OSOT_OK) return rc;
    hipLaunchKernelGGL(osot_admm_kernel, dim3((unsigned)B), dim3(64), lds, (hipStream_t)hip_stream, Q);
    HIP_TRY(hipGetLastError());
    return OSOT_OK;
}

int osot_qp_solve_batch_admm(int B, int n, int nc, const double* H, const double* g, const double* A,
                             const double* lA, const double* uA, const double* l, const double* u,
                             double eps_reg, int max_iter, double* x, int* status, int* iterations, void* hip_stream) {
    osot_admm_options opt;
    std::memset(&opt, 0, sizeof(opt));
    opt.max_iter = max_iter;
    return osot_qp_solve_batch_admm_warm(B, n, nc, H, g, A, lA, uA, l, u, eps_reg, &opt, nullptr, nullptr, nullptr, x, status, iterations, hip_stream);
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
    int device;         // the device that was current when the back-end was created
    std::vector<double> H, g, A, lA, uA, l, u, x;
    double* d_buf;      // device arena
    size_t d_cap;
    int* d_status;
    int last_status, last_iters;
    int max_iterations = 0;   // osot_backend_options: active-set iteration cap (0: the kernel's default, 20 (n + nc) + 100)
    // hot start across solve() calls (QPOasesBackEnd::solve hot-starts every call, QPOasesBackEnd.cpp:258-285): the inequality
    // working set of the previous solve, 64 constraint codes on the device; forgotten (cold start) by initProblem and whenever
    // the row count changes (the reference re-creates its SQProblem then, QPOasesBackEnd.cpp:229-244)
    int* d_hot = nullptr;
    int hot_nc = -1;          // row count the recorded set belongs to (-1: nothing recorded)
};

namespace {
int backend_run(osot_backend* be) {
    const int n = be->nv, nc = be->nc;
    DeviceGuard guard(be->device);
    if (!guard.ok) return fail(OSOT_ERR_HIP, "hipSetDevice failed");
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
    if (!be->d_hot) HIP_TRY(hipMalloc((void**)&be->d_hot, 64 * sizeof(int)));
    if (be->hot_nc != nc) {   // nothing recorded for this problem shape: cold start
        HIP_TRY(hipMemset(be->d_hot, 0xff, 64 * sizeof(int)));
        be->hot_nc = nc;
    }
    int rc = qp_solve_batch_impl(1, n, nc, dH, dg, nc ? dA : nullptr, nc ? dlA : nullptr, nc ? duA : nullptr,
                                 be->has_bounds ? dl : nullptr, be->has_bounds ? du : nullptr, be->eps_abs, be->max_iterations,
                                 dx, be->d_status, be->d_status + 1, nullptr, be->d_hot);
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
    if (number_of_variables < 1 || number_of_variables > OSOT_MAX_QP_VARS)
        return fail(OSOT_ERR_INVALID, "number_of_variables out of range (1..128)");
    if (number_of_constraints < 0) return fail(OSOT_ERR_INVALID, "negative number_of_constraints");
    if (eps_regularisation < 0) return fail(OSOT_ERR_INVALID, "Negative eps is not allowed!");
    osot_backend* be = new osot_backend();
    be->nv = number_of_variables; be->nc = number_of_constraints; be->hessian_type = hessian_type;
    be->eps_abs = 1.0e3 * 2.221e-16 * eps_regularisation;   // QPOasesBackEnd.cpp:57,67
    be->inited = false; be->has_bounds = false;
    be->d_buf = nullptr; be->d_cap = 0; be->d_status = nullptr;
    be->last_status = 0; be->last_iters = 0;
    if (hipGetDevice(&be->device) != hipSuccess) be->device = 0;
    be->x.assign(be->nv, 0.0);
    *out = be;
    return OSOT_OK;
}

int osot_backend_destroy(osot_backend* be) {
    if (!be) return OSOT_OK;
    DeviceGuard guard(be->device);
    if (be->d_buf) hipFree(be->d_buf);
    if (be->d_status) hipFree(be->d_status);
    if (be->d_hot) hipFree(be->d_hot);
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
    // validate first, commit after: a refused update leaves the previous problem intact (BackEnd::updateConstraints,
    // BackEnd.cpp:47-71)
    for (int i = 0; i < number_of_constraints; ++i)
        if (lA[i] > uA[i]) return fail(OSOT_ERR_INVALID, "lA > uA");
    be->nc = number_of_constraints;   // a changed row count re-allocates (QPOasesBackEnd.cpp:229-244)
    be->A.assign(A, A + (size_t)be->nc * be->nv);
    be->lA.assign(lA, lA + be->nc);
    be->uA.assign(uA, uA + be->nc);
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
    // every argument is checked before anything is committed: a refused initProblem leaves the object as it was
    if (!H || !g) return fail(OSOT_ERR_INVALID, "null argument");
    if (be->nc > 0 && (!A || !lA || !uA)) return fail(OSOT_ERR_INVALID, "null A/lA/uA");
    for (int i = 0; i < be->nc; ++i)
        if (lA[i] > uA[i]) return fail(OSOT_ERR_INVALID, "lA > uA");
    if ((l == nullptr) != (u == nullptr)) return fail(OSOT_ERR_INVALID, "l and u must both be given");
    if (l) for (int i = 0; i < be->nv; ++i)
        if (l[i] > u[i]) return fail(OSOT_ERR_INVALID, "l > u");
    int rc = osot_backend_update_task(be, H, g);
    if (rc != OSOT_OK) return rc;
    rc = osot_backend_update_constraints(be, A, lA, uA, be->nc);
    if (rc != OSOT_OK) return rc;
    rc = osot_backend_update_bounds(be, l, u);
    if (rc != OSOT_OK) return rc;
    be->hot_nc = -1;          // initProblem is a cold start (QPOasesBackEnd::initProblem -> SQProblem::init)
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

int osot_backend_get_options(osot_backend* be, osot_backend_options* opt) {
    if (!be || !opt) return fail(OSOT_ERR_INVALID, "null argument");
    opt->max_iterations = be->max_iterations;
    opt->last_iterations = be->last_iters;
    opt->last_status = be->last_status;
    return OSOT_OK;
}
int osot_backend_set_options(osot_backend* be, const osot_backend_options* opt) {
    if (!be || !opt) return fail(OSOT_ERR_INVALID, "null argument");
    if (opt->max_iterations < 0) return fail(OSOT_ERR_INVALID, "negative iteration cap");
    be->max_iterations = opt->max_iterations;
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
    int n, n_frames, n_pairs, n_env, device;
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
    }
    kin_build_tables(h);
    for (int f = 0; f < d->n_frames; ++f)
        if (d->frame_joint[f] < 0 || d->frame_joint[f] >= d->n) return fail(OSOT_ERR_INVALID, "frame attached to a joint out of range");
    for (int f = 0; f < d->n_frames; ++f)      // relative base link (Cartesian.cpp:40-51: a base link that is not the distal link)
        if (d->frame_base[f] < 0 || d->frame_base[f] > d->n_frames || d->frame_base[f] == f + 1)
            return fail(OSOT_ERR_INVALID, "frame_base must be 0 (world) or 1 + the index of another frame");
    if (d->n_pairs < 0 || d->n_pairs > OSOT_KIN_MAX_PAIRS) return fail(OSOT_ERR_INVALID, "collision pair count out of range");
    if (d->n_env < 0 || d->n_env > OSOT_KIN_MAX_ENV) return fail(OSOT_ERR_INVALID, "environment shape count out of range");
    for (int p = 0; p < d->n_pairs; ++p) {
        if (d->pair_joint[p][0] < 0 || d->pair_joint[p][0] >= d->n) return fail(OSOT_ERR_INVALID, "collision shape attached to a joint out of range");
        if (d->pair_joint[p][1] < -1 || d->pair_joint[p][1] >= d->n)      // (-1: side b is a world / environment shape)
            return fail(OSOT_ERR_INVALID, "collision shape attached to a joint out of range");
        if (!(d->pair_radius[p][0] >= 0.0) || !(d->pair_radius[p][1] >= 0.0)) return fail(OSOT_ERR_INVALID, "negative capsule radius");
        if (d->pair_kind[p] != OSOT_SHAPE_CAPSULE && d->pair_kind[p] != OSOT_SHAPE_BOX) return fail(OSOT_ERR_UNSUPPORTED, "unknown collision shape kind");
        if (d->pair_env[p] < 0 || d->pair_env[p] > d->n_env) return fail(OSOT_ERR_INVALID, "pair refers to an environment shape out of range");
        if (d->pair_env[p] > 0 && d->pair_joint[p][1] != -1) return fail(OSOT_ERR_INVALID, "an environment shape is carried by the world (joint -1)");
        if (d->pair_kind[p] == OSOT_SHAPE_BOX)
            for (int i = 0; i < 3; ++i) if (!(d->pair_box[p][i] > 0.0)) return fail(OSOT_ERR_INVALID, "box half extents must be positive");
    }
    DeviceGuard guard(device);
    if (!guard.ok) return fail(OSOT_ERR_HIP, "hipSetDevice failed");
    osot_kin* k = new osot_kin();
    k->n = d->n; k->n_frames = d->n_frames; k->n_pairs = d->n_pairs; k->n_env = d->n_env; k->device = device;
    hipError_t e = hipMalloc(&k->dev, sizeof(DevKin));
    if (e == hipSuccess) e = hipMemcpy(k->dev, &h, sizeof(DevKin), hipMemcpyHostToDevice);
    if (e != hipSuccess) { delete k; return fail(OSOT_ERR_HIP, hipGetErrorString(e)); }
    *out = k;
    return OSOT_OK;
}

int osot_kin_destroy(osot_kin* k) {
    if (!k) return OSOT_OK;
    DeviceGuard guard(k->device);
    if (k->dev) hipFree(k->dev);
    delete k;
    return OSOT_OK;
}

int osot_kinematics(osot_kin* k, const osot_kin_batch* b, void* hip_stream) {
    if (!k || !b) return fail(OSOT_ERR_INVALID, "null argument");
    if (b->B < 0) return fail(OSOT_ERR_INVALID, "negative batch");
    if (b->B == 0) return OSOT_OK;
    if (!b->q) return fail(OSOT_ERR_INVALID, "q is null");
    DeviceGuard guard(k->device);
    if (!guard.ok) return fail(OSOT_ERR_HIP, "hipSetDevice failed");
    const bool pairs = k->n_pairs > 0 && (b->pair_dist || b->pair_J);
    if (pairs && k->n_env > 0 && !b->env_pose) return fail(OSOT_ERR_INVALID, "the model has environment shapes but env_pose is null");
    const dim3 grid((unsigned)(k->n <= 32 ? (b->B + 1) / 2 : b->B)), block(64);   // <= 32 joints: two instances per wavefront
    hipStream_t st = (hipStream_t)hip_stream;
    const DevKin* dk = (const DevKin*)k->dev;
    if (k->n <= 32) {
        if (pairs) hipLaunchKernelGGL((osot_kin_kernel<true, 32>), grid, block, 0, st, dk, *b);
        else hipLaunchKernelGGL((osot_kin_kernel<false, 32>), grid, block, 0, st, dk, *b);
    } else {
        if (pairs) hipLaunchKernelGGL((osot_kin_kernel<true, 64>), grid, block, 0, st, dk, *b);
        else hipLaunchKernelGGL((osot_kin_kernel<false, 64>), grid, block, 0, st, dk, *b);
    }
    HIP_TRY(hipGetLastError());
    return OSOT_OK;
}

static int control_launch(osot_solver* s, osot_kin* k, const osot_kin_batch* kb, const osot_leaf_batch* leaf, const osot_assembled_out* out,
                          const osot_qp_batch* b, double* q_integrate, int steps, double* dq_steps, int* status_steps, void* hip_stream);

int osot_control_cycle(osot_solver* s, osot_kin* k, const osot_kin_batch* kb, const osot_leaf_batch* leaf, const osot_assembled_out* out,
                       const osot_qp_batch* b, double* q_integrate, void* hip_stream) {
    return control_launch(s, k, kb, leaf, out, b, q_integrate, 1, nullptr, nullptr, hip_stream);
}

int osot_control_rollout(osot_solver* s, osot_kin* k, const osot_kin_batch* kb, const osot_leaf_batch* leaf, const osot_assembled_out* out,
                         const osot_qp_batch* b, double* q_integrate, int steps, double* dq_steps, int* status_steps, void* hip_stream) {
    if (steps < 1) return fail(OSOT_ERR_INVALID, "a rollout has at least one step");
    if (steps > 1 && !q_integrate) return fail(OSOT_ERR_INVALID, "a rollout of several steps integrates q (q_integrate is null: every step would solve the same problem)");
    return control_launch(s, k, kb, leaf, out, b, q_integrate, steps, dq_steps, status_steps, hip_stream);
}

static int control_launch(osot_solver* s, osot_kin* k, const osot_kin_batch* kb, const osot_leaf_batch* leaf, const osot_assembled_out* out,
                          const osot_qp_batch* b, double* q_integrate, int steps, double* dq_steps, int* status_steps, void* hip_stream) {
    if (!s || !k || !kb || !leaf || !out || !b) return fail(OSOT_ERR_INVALID, "null argument");
    if (leaf->B != b->B || kb->B != b->B) return fail(OSOT_ERR_INVALID, "kinematics batch, leaf batch and qp batch disagree on B");
    if (leaf->B < 0 || leaf->B > s->max_batch) return fail(OSOT_ERR_INVALID, "batch size exceeds max_batch");
    if (leaf->B == 0) return OSOT_OK;
    if (!kb->q) return fail(OSOT_ERR_INVALID, "q is null");
    if (k->device != s->device) return fail(OSOT_ERR_INVALID, "the model and the solver live on different devices");
    if (k->n != s->plan.n) return fail(OSOT_ERR_INVALID, "the model's joint count is not the plan's variable count");
    if (k->n_pairs > 0 && (kb->pair_dist || kb->pair_J) && k->n_env > 0 && !kb->env_pose)
        return fail(OSOT_ERR_INVALID, "the model has environment shapes but env_pose is null");
    DevUpdate U;
    const char* why = "";
    int rc = make_update_args(s->plan, s->h_uplan, leaf, out, s->d_uplan, U, &why);
    if (rc != OSOT_OK) return fail(rc, why);
    DevControl C;
    C.K = (const DevKin*)k->dev;
    C.K_pairs = k->n_pairs;
    C.Bt = *kb;
    C.q_int = q_integrate;
    C.steps = steps; C.dq_steps = dq_steps; C.status_steps = status_steps;
    return ihqp_launch(s, b, hip_stream, nullptr, &U, &C);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// inverse-dynamics producers and computedTorque (osot_id.h)
// ---------------------------------------------------------------------------------------------------
namespace {
int id_model_check(const osot_id_model* m, int* n_out) {
    if (!m) return fail(OSOT_ERR_INVALID, "null model");
    if (m->B < 0 || m->nv < 1 || m->n_contacts < 0) return fail(OSOT_ERR_INVALID, "bad sizes");
    if (m->contact_dim != 3 && m->contact_dim != 6) return fail(OSOT_ERR_INVALID, "Unsupported  contact model");   // InverseDynamics.cpp:27
    const int nf = m->n_contacts * m->contact_dim;
    if (nf > OSOT_ID_MAX_FORCE_VARS || m->nv + nf > OSOT_MAX_VARS) return fail(OSOT_ERR_UNSUPPORTED, "nv + force variables exceed 64 (or forces exceed 24)");
    if (!m->Bm || (nf > 0 && !m->Jc)) return fail(OSOT_ERR_INVALID, "null B / Jc");
    *n_out = m->nv + nf;
    return OSOT_OK;
}
}  // namespace

extern "C" {

int osot_id_rows(const osot_id_model* m, double* C_dyn, long long dyn_stride, double* C_tau, long long tau_stride,
                 int n_tasks, const double* const* J, const int* J_rows, double* const* A_dst, const long long* A_stride,
                 void* hip_stream) {
    int n = 0;
    int rc = id_model_check(m, &n);
    if (rc != OSOT_OK) return rc;
    if (m->B == 0) return OSOT_OK;
    if (n_tasks < 0 || n_tasks > OSOT_MAX_TASKS) return fail(OSOT_ERR_INVALID, "n_tasks out of range");
    if (n_tasks > 0 && (!J || !J_rows || !A_dst || !A_stride)) return fail(OSOT_ERR_INVALID, "null task arrays");
    if (C_dyn && m->nv < 6) return fail(OSOT_ERR_INVALID, "DynamicFeasibility needs a floating base (nv >= 6)");
    DevIdRows R;
    std::memset(&R, 0, sizeof(R));
    R.B = m->B; R.nv = m->nv; R.n_contacts = m->n_contacts; R.cdim = m->contact_dim; R.n = n;
    R.Bm = m->Bm; R.Jc = m->Jc;
    R.C_dyn = C_dyn; R.dyn_stride = dyn_stride; R.C_tau = C_tau; R.tau_stride = tau_stride;
    R.n_tasks = n_tasks;
    for (int i = 0; i < n_tasks; ++i) {
        if (!J[i] || !A_dst[i] || J_rows[i] < 1) return fail(OSOT_ERR_INVALID, "bad task block");
        R.J[i] = J[i]; R.J_rows[i] = J_rows[i]; R.A_dst[i] = A_dst[i]; R.A_stride[i] = A_stride[i];
    }
    hipLaunchKernelGGL(osot_id_rows_kernel, dim3((unsigned)m->B), dim3(64), 0, (hipStream_t)hip_stream, R);
    HIP_TRY(hipGetLastError());
    return OSOT_OK;
}

int osot_id_force_gains(int B, int nv, int rows, const double* J, const double* Bi, const double* Kp, const double* Kd,
                        const double* f_virtual, double* p0_gains, long long p0_stride, double* a_ref, void* hip_stream) {
    if (B < 0 || nv < 1 || nv > 64 || rows < 1 || rows > 6) return fail(OSOT_ERR_INVALID, "force gains: sizes out of range (nv <= 64, rows <= 6)");
    if (B == 0) return OSOT_OK;
    if (!J || !Bi || !Kp || !Kd || !p0_gains) return fail(OSOT_ERR_INVALID, "force gains: null argument");
    if (f_virtual && !a_ref) return fail(OSOT_ERR_INVALID, "force gains: a virtual force needs a_ref to add Mi f to");
    DevForceGains F;
    std::memset(&F, 0, sizeof(F));
    F.B = B; F.nv = nv; F.rows = rows; F.J = J; F.Bi = Bi; F.f = f_virtual; F.G = p0_gains; F.G_stride = p0_stride; F.a_ref = a_ref;
    for (int i = 0; i < rows * rows; ++i) { F.Kp[i] = Kp[i]; F.Kd[i] = Kd[i]; }
    hipLaunchKernelGGL(osot_force_gains_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)hip_stream, F);
    HIP_TRY(hipGetLastError());
    return OSOT_OK;
}

int osot_computed_torque(const osot_id_model* m, const double* x, double* tau, int* ok, double fb_tol, void* hip_stream) {
    int n = 0;
    int rc = id_model_check(m, &n);
    if (rc != OSOT_OK) return rc;
    if (m->B == 0) return OSOT_OK;
    if (!m->h || !x || !tau) return fail(OSOT_ERR_INVALID, "null h / x / tau");
    DevTorque T;
    std::memset(&T, 0, sizeof(T));
    T.B = m->B; T.nv = m->nv; T.n_contacts = m->n_contacts; T.cdim = m->contact_dim; T.n = n; T.floating_base = m->floating_base;
    T.Bm = m->Bm; T.h = m->h; T.Jc = m->Jc; T.x = x; T.tau = tau; T.ok = ok; T.fb_tol = fb_tol;
    hipLaunchKernelGGL(osot_torque_kernel, dim3((unsigned)m->B), dim3(64), 0, (hipStream_t)hip_stream, T);
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
    DeviceGuard guard(device);
    if (!guard.ok) return fail(OSOT_ERR_HIP, "hipSetDevice failed");
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

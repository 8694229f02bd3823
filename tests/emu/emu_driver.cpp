#include <cstdlib>
// tests/emu/emu_driver.cpp -- TEST INFRASTRUCTURE ONLY.
// Runs the product's kernel bodies (opensot_amd/csrc/osot_kernels.h, osot_qp_core.h) through the host
// lock-step emulation in tests/emu/hip/hip_runtime.h, with HOST pointers in the batch structs.
// Built by tests/emu/build.sh; used by tests/test_emulated_kernels.py (no GPU needed).
#include <osot_team.h>
#include "osot_host_plan.h"
#include "osot_kin.h"
#include "osot_id.h"
#include "osot_nhqp_host.h"
#include "osot_admm.h"
#include "osot_ehqp.h"
#include <vector>

using namespace osot;

// task_active: [OSOT_MAX_LEVELS * OSOT_MAX_TASKS] Task::setActive flags, or null
// hot: [B][L][32 or 64] hot-start state (constraint codes, -1 = none; read and rewritten), or null for a cold start
extern "C" __attribute__((visibility("default"))) int emu_ihqp_solve(const osot_plan_desc* plan, const osot_qp_batch* b,
                                                                     const unsigned char* task_active, int* hot) {
    const char* why;
    int rc = plan_validate(plan, &why);
    if (rc != OSOT_OK) { fprintf(stderr, "emu: %s\n", why); return rc; }
    DevPlan P; int T; size_t lds;
    make_dev_plan(*plan, b->level_active, P, T, lds, task_active);
    DevBatch D;
    memset(&D, 0, sizeof(D));
    D.B = b->B;
    for (int k = 0; k < plan->n_levels; ++k) { D.A[k] = b->A[k]; D.b[k] = b->b[k]; D.w[k] = b->w[k]; D.c[k] = b->c[k];
                                               D.WA[k] = b->WA[k]; D.Wb[k] = b->Wb[k]; }
    D.C = b->C; D.lo = b->lo; D.up = b->up; D.l = b->l; D.u = b->u;
    D.dq = b->dq; D.x_levels = b->x_levels; D.status = b->status; D.iterations = b->iterations;
    D.b_reg = plan->has_regularisation ? b->b_reg : nullptr;
    D.A_reg = (plan->has_regularisation && plan->regularisation_dense) ? b->A_reg : nullptr;
    D.accepted_slack = b->accepted_slack;
    D.hot = hot;
    std::vector<double> rows_scratch(P.rows_in_global ? (size_t)b->B * P.rows_doubles : 1);
    D.rows_scratch = rows_scratch.data();
    const unsigned grid = (unsigned)b->B;
    // (the emulation runs the instantiation with the dense-weight / inactive-task / hot-start code: it is a superset.  With
    //  OSOT_EMU_BOX=1 in the environment a plan the product would give the BOX instantiation to -- no constraint rows, at most 32
    //  variables, none of the EXTRA features in use -- runs THAT one: tests/test_emulated_kernels.py compares the two)
    const char* want_box = getenv("OSOT_EMU_BOX");
    bool extra = hot != nullptr || task_active != nullptr || (plan->has_regularisation && plan->regularisation_dense);
    for (int k = 0; k < plan->n_levels; ++k) extra = extra || b->WA[k] != nullptr;
    if (want_box && want_box[0] == '1' && !extra && (P.nc == 0 ? T == 32 : plan_rows_all_equalities(*plan))) {
        if (T == 32) emu::launch(osot_cascade_kernel<32, false, false, true>, grid, lds, 64, P, D);
        else if (T == 40) emu::launch(osot_cascade_kernel<40, false, false, true>, grid, lds, 64, P, D);
        else if (T == 56) emu::launch(osot_cascade_kernel<56, false, false, true>, grid, lds, 64, P, D);
        else emu::launch(osot_cascade_kernel<64, false, false, true>, grid, lds, 64, P, D);
        return OSOT_OK + 100;   // (tells the test that the BOX instantiation ran)
    }
    if (T == 32) emu::launch(osot_cascade_kernel<32, false, true>, grid, lds, 64, P, D);
    else if (T == 40) emu::launch(osot_cascade_kernel<40, false, true>, grid, lds, 64, P, D);
    else if (T == 56) emu::launch(osot_cascade_kernel<56, false, true>, grid, lds, 64, P, D);
    else emu::launch(osot_cascade_kernel<64, false, true>, grid, lds, 64, P, D);
    return OSOT_OK;
}

extern "C" __attribute__((visibility("default"))) int emu_qp_solve_batch(int B, int n, int nc, const double* H, const double* g, const double* A,
                                  const double* lA, const double* uA, const double* l, const double* u,
                                  double eps_abs, int max_iter, double* x, int* status, int* iterations) {
    DevQP Q;
    memset(&Q, 0, sizeof(Q));
    Q.B = B; Q.n = n; Q.nc = nc;
    Q.max_iter = max_iter > 0 ? max_iter : 20 * (n + nc) + 100;
    Q.eps_abs = eps_abs;
    Q.H = H; Q.g = g; Q.A = A; Q.lA = lA; Q.uA = uA; Q.l = l; Q.u = u; Q.x = x; Q.status = status; Q.iterations = iterations;
    const int T = n <= 32 ? 32 : (n <= WaveCtx<40>::NMAX ? 40 : (n <= WaveCtx<56>::NMAX ? 56 : 64));   // as qp_solve_batch_impl
    const size_t lds = (size_t)lds_layout(T, nc, &Q.lds_rows_off, &Q.lds_rows_cap) * sizeof(double);
    const unsigned grid = (unsigned)B;
    if (T == 32) emu::launch(osot_qp_kernel<32>, grid, lds, 64, Q);
    else if (T == 40) emu::launch(osot_qp_kernel<40>, grid, lds, 64, Q);
    else if (T == 56) emu::launch(osot_qp_kernel<56>, grid, lds, 64, Q);
    else emu::launch(osot_qp_kernel<64>, grid, lds, 64, Q);
    return OSOT_OK;
}

// the OSQP-convention ADMM back-end (osot_admm.h) on host pointers; scaling as osot_admm_options.scaling, warm_* may be null
extern "C" __attribute__((visibility("default"))) int emu_qp_solve_batch_admm(int B, int n, int nc, const double* H, const double* g,
        const double* A, const double* lA, const double* uA, const double* l, const double* u, double eps_reg, int max_iter,
        double* x, int* status, int* iterations, int scaling, double* warm_x, double* warm_y, double* warm_rho) {
    osot_admm_options opt;
    memset(&opt, 0, sizeof(opt));
    opt.max_iter = max_iter; opt.scaling = scaling;
    const DevAdmm Q = admm_args(B, n, nc, H, g, A, lA, uA, l, u, eps_reg, &opt, warm_x, warm_y, warm_rho, x, status, iterations);
    emu::launch(osot_admm_kernel, (unsigned)B, admm_lds_bytes(n, nc, l != nullptr), 64, Q);
    return OSOT_OK;
}

// the null-space front-end (osot_nhqp_host.h: the product's own orchestration) on host pointers
extern "C" __attribute__((visibility("default"))) int emu_nhqp_solve(const osot_plan_desc* plan, const osot_qp_batch* b,
                                                                     const osot_nhqp_options* opt, const unsigned char* task_active) {
    const char* why = "";
    int rc = plan_validate(plan, &why);
    if (rc != OSOT_OK) { fprintf(stderr, "emu: %s\n", why); return rc; }
    const NhqpSizes z = nhqp_sizes(*plan, b->B);
    std::vector<double> N0(z.N), N1(z.N), q0(z.q0), H(z.H), g(z.g), R(z.R + 1), rlo(z.rl + 1), rup(z.rl + 1), zz(z.z), V2(z.V2);
    std::vector<int> st(z.st), it(z.st);
    NhqpWorkspace ws = {{N0.data(), N1.data()}, q0.data(), H.data(), g.data(), R.data(), rlo.data(), rup.data(), zz.data(), V2.data(), st.data(), it.data()};
    const unsigned grid = (unsigned)b->B;
    rc = nhqp_run(*plan, b, opt, ws,
        [&](const DevNhqp& Q) {
            if (nhqp_level_is_wide(Q.m, Q.nf)) emu::launch(osot_nhqp_prepare_wide_kernel, grid, nhqp_prepare_wide_lds_bytes(Q.m, Q.n), 64, Q);
            else if (Q.n > 32) {
                if (Q.m <= 32) emu::launch(osot_nhqp_prepare64_kernel<32>, grid, nhqp_prepare64_lds_bytes(32, Q.n), 64, Q);
                else emu::launch(osot_nhqp_prepare64_kernel<64>, grid, nhqp_prepare64_lds_bytes(64, Q.n), 64, Q);
            }
            else if (Q.m <= 32) emu::launch(osot_nhqp_prepare_kernel<32>, grid, 0, 64, Q);
            else emu::launch(osot_nhqp_prepare_kernel<64>, grid, 0, 64, Q);
        },
        [&](int B, int n, int nc, const double* Hh, const double* gg, const double* A, const double* lA, const double* uA,
            const double* l, const double* u, double eps, double* x, int* status, int* iters) {
            return emu_qp_solve_batch(B, n, nc, Hh, gg, A, lA, uA, l, u, eps, 0, x, status, iters);
        },
        [&](const DevNhqpAcc& A) { emu::launch(osot_nhqp_accumulate_kernel, grid, 0, 64, A); }, &why, task_active);
    if (rc != OSOT_OK) fprintf(stderr, "emu: %s\n", why);
    return rc;
}

// the equality-only front-end (osot_ehqp.h) on host pointers
// task_active: [OSOT_MAX_LEVELS * OSOT_MAX_TASKS] Task::setActive flags, or null
extern "C" __attribute__((visibility("default"))) int emu_ehqp_solve(const osot_plan_desc* plan, const osot_qp_batch* b, double sigma_min,
                                                                     const unsigned char* task_active) {
    const char* why = "";
    int rc = plan_validate(plan, &why);
    if (rc != OSOT_OK) { fprintf(stderr, "emu: %s\n", why); return rc; }
    DevEhqp Q;
    rc = ehqp_args(*plan, b, sigma_min, task_active, Q, &why);
    if (rc != OSOT_OK) { fprintf(stderr, "emu: %s\n", why); return rc; }
    if (Q.use_qr && Q.n <= 32) emu::launch(osot_ehqp_qr_kernel<32>, (unsigned)b->B, ehqp_qr_lds_bytes(32, Q.rows8), 64, Q);
    else if (Q.use_qr) emu::launch(osot_ehqp_qr_kernel<64>, (unsigned)b->B, ehqp_qr_lds_bytes(64, Q.rows8), 64, Q);
    else emu::launch(osot_ehqp_kernel, (unsigned)b->B, 0, 64, Q);
    return OSOT_OK;
}

// AutoStack::update (osot_update_kernel) on host pointers
extern "C" __attribute__((visibility("default"))) int emu_stack_update(const osot_plan_desc* plan, const osot_leaf_batch* leaf,
                                                                       const osot_assembled_out* out) {
    const char* why;
    int rc = plan_validate(plan, &why);
    if (rc != OSOT_OK) { fprintf(stderr, "emu: %s\n", why); return rc; }
    static DevUpdatePlan PL;
    make_update_plan(*plan, PL);
    DevUpdate U;
    rc = make_update_args(*plan, PL, leaf, out, &PL, U, &why);
    if (rc != OSOT_OK) { fprintf(stderr, "emu: %s\n", why); return rc; }
    emu::launch(osot_update_kernel, (unsigned)leaf->B, 0, 64, U);
    return OSOT_OK;
}

// the inverse-dynamics producers and computedTorque (opensot_amd/csrc/osot_id.h) on host pointers
extern "C" __attribute__((visibility("default"))) int emu_id_rows(const osot_id_model* m, double* C_dyn, long long dyn_stride,
        double* C_tau, long long tau_stride, int n_tasks, const double* const* J, const int* J_rows, double* const* A_dst,
        const long long* A_stride) {
    DevIdRows R;
    memset(&R, 0, sizeof(R));
    R.B = m->B; R.nv = m->nv; R.n_contacts = m->n_contacts; R.cdim = m->contact_dim; R.n = m->nv + m->n_contacts * m->contact_dim;
    R.Bm = m->Bm; R.Jc = m->Jc; R.C_dyn = C_dyn; R.dyn_stride = dyn_stride; R.C_tau = C_tau; R.tau_stride = tau_stride;
    R.n_tasks = n_tasks;
    for (int i = 0; i < n_tasks; ++i) { R.J[i] = J[i]; R.J_rows[i] = J_rows[i]; R.A_dst[i] = A_dst[i]; R.A_stride[i] = A_stride[i]; }
    emu::launch(osot_id_rows_kernel, (unsigned)m->B, 0, 64, R);
    return OSOT_OK;
}
extern "C" __attribute__((visibility("default"))) int emu_computed_torque(const osot_id_model* m, const double* x, double* tau,
                                                                          int* ok, double fb_tol) {
    DevTorque T;
    memset(&T, 0, sizeof(T));
    T.B = m->B; T.nv = m->nv; T.n_contacts = m->n_contacts; T.cdim = m->contact_dim; T.n = m->nv + m->n_contacts * m->contact_dim;
    T.floating_base = m->floating_base; T.Bm = m->Bm; T.h = m->h; T.Jc = m->Jc; T.x = x; T.tau = tau; T.ok = ok; T.fb_tol = fb_tol;
    emu::launch(osot_torque_kernel, (unsigned)m->B, 0, 64, T);
    return OSOT_OK;
}

// the kinematics producer (opensot_amd/csrc/osot_kin.h) on host pointers; the ancestor / subtree masks are built as in
// osot_kin_create (opensot_amd/csrc/osot_mi355x.hip)
extern "C" __attribute__((visibility("default"))) int emu_kinematics(const osot_kin_desc* d, const osot_kin_batch* b) {
    if (!d || !b || d->n < 1 || d->n > OSOT_KIN_MAX_JOINTS) return OSOT_ERR_INVALID;
    static DevKin h;
    memset(&h, 0, sizeof(h));
    h.d = *d;
    for (int j = 0; j < d->n; ++j) if (d->parent[j] >= j) return OSOT_ERR_INVALID;
    kin_build_tables(h);
    const bool pairs = d->n_pairs > 0 && (b->pair_dist || b->pair_J);
    if (d->n <= 32) {
        if (pairs) emu::launch(osot_kin_kernel<true, 32>, (unsigned)((b->B + 1) / 2), 0, 64, (const DevKin*)&h, *b);
        else emu::launch(osot_kin_kernel<false, 32>, (unsigned)((b->B + 1) / 2), 0, 64, (const DevKin*)&h, *b);
    } else {
        if (pairs) emu::launch(osot_kin_kernel<true, 64>, (unsigned)b->B, 0, 64, (const DevKin*)&h, *b);
        else emu::launch(osot_kin_kernel<false, 64>, (unsigned)b->B, 0, 64, (const DevKin*)&h, *b);
    }
    return OSOT_OK;
}

// the symmetric eigen-solver of the nHQP / eHQP front-ends on its own (tests/test_nhqp.py): K [32][33] in, eigenvalues on
// its diagonal and E [32][33] = eigenvectors (columns) out
static void emu_eig_kernel(double* Kg, double* Eg, int k) {
    OSOT_STATIC_LDS(double, K, 32 * kNS);
    OSOT_STATIC_LDS(double, E, 32 * kNS);
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    for (int e = lane; e < 32 * kNS; e += 64) { K[e] = Kg[e]; E[e] = 0.0; }
    wave_sync();
    sym_eig32(K, E, k, c, h);
    wave_sync();
    for (int e = lane; e < 32 * kNS; e += 64) { Kg[e] = K[e]; Eg[e] = E[e]; }
}
static void emu_eig_fast_kernel(double* Kg, double* Eg, int k) {
    OSOT_STATIC_LDS(double, K, 32 * kNS);
    OSOT_STATIC_LDS(double, E, 32 * kNS);
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    for (int e = lane; e < 32 * kNS; e += 64) { K[e] = Kg[e]; E[e] = 0.0; }
    wave_sync();
    sym_eig32_fast(K, E, k, c, h);
    wave_sync();
    for (int e = lane; e < 32 * kNS; e += 64) { Kg[e] = K[e]; Eg[e] = E[e]; }
}
extern "C" __attribute__((visibility("default"))) int emu_sym_eig32_fast(double* K, double* E, int k) {
    emu::launch(emu_eig_fast_kernel, 1u, 0, 64, K, E, k);
    return 0;
}
extern "C" __attribute__((visibility("default"))) int emu_sym_eig32(double* K, double* E, int k) {
    emu::launch(emu_eig_kernel, 1u, 0, 64, K, E, k);
    return 0;
}

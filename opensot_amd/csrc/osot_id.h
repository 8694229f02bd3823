// opensot_amd/csrc/osot_id.h -- device side of the inverse-dynamics formulation (BASELINE config 5), x = [qddot; F]
// (src/utils/InverseDynamics.cpp:12-28): the matrices that are pure copies of model quantities, written straight into
// their row ranges of the stacked A_k / C, and InverseDynamics::computedTorque.
//
//   osot_id_rows_kernel ..... [B_u, -J_f'] (6 rows: acceleration::DynamicFeasibility::_update, DynamicFeasibility.cpp:22-46)
//                             and [B, -Jc'] (nv rows: TorqueLimits::update, TorqueLimits.cpp:25-46) from the inertia matrix
//                             and the contact Jacobians; [J 0] task rows (acceleration::Cartesian / CoM: the task matrix is
//                             the Jacobian on the qddot columns and zero on the force columns, Cartesian.cpp:152-160)
//   osot_torque_kernel ...... tau = B qddot + h - sum_c Jc' F_c and the floating-base acceptance test
//                             (InverseDynamics.cpp:57-96)
// One wavefront per instance, LANE = COLUMN of x (n = nv + contacts * contact_dim <= 64): every output row is one coalesced
// 8 n-byte store; the contact Jacobians are staged through LDS once (their transposes are what the rows need).  HBM-bound by
// construction: reads (nv + contacts * dim) * nv doubles, writes (6 + nv + task rows) * n.
#pragma once
#include <osot_team.h>
#include <osot_mi355x.h>

namespace osot {

struct DevIdRows {
    int B, nv, n_contacts, cdim, n;
    const double* Bm;      // [B][nv][nv]
    const double* Jc;      // [B][n_contacts][cdim][nv]
    double* C_dyn; long long dyn_stride;    // first of the 6 rows in instance 0 (or null), doubles between instances
    double* C_tau; long long tau_stride;    // first of the nv rows (or null)
    int n_tasks;                            // [J 0] blocks
    const double* J[OSOT_MAX_TASKS]; int J_rows[OSOT_MAX_TASKS];
    double* A_dst[OSOT_MAX_TASKS]; long long A_stride[OSOT_MAX_TASKS];
};

__global__ void __launch_bounds__(64) osot_id_rows_kernel(const DevIdRows R) {
    OSOT_STATIC_LDS(double, JcS, OSOT_ID_MAX_FORCE_VARS * 64);   // contact Jacobian rows, nv columns each
    const long long inst = blockIdx.x;
    const int c = threadIdx.x;
    const int nv = R.nv, n = R.n, nf = R.n_contacts * R.cdim;
    if (inst >= R.B) return;
    if (R.C_dyn || R.C_tau) {
        const double* Jc = R.Jc + inst * nf * nv;
        for (int r = 0; r < nf; ++r) if (c < nv) JcS[r * 64 + c] = Jc[r * nv + c];   // coalesced row loads
        wave_sync();
        const double* Bm = R.Bm + inst * nv * nv;
        const int rows = R.C_tau ? nv : 6;
        for (int r = 0; r < rows; ++r) {
            // column c of row r:  B[r][c] on the qddot columns,  -Jc[ct][d][r] on the force column nv + cdim ct + d
            double v = 0.0;
            if (c < nv) v = Bm[r * nv + c];
            else if (c < n) v = -JcS[(c - nv) * 64 + r];
            if (c < n) {
                if (R.C_tau) R.C_tau[inst * R.tau_stride + r * n + c] = v;
                if (R.C_dyn && r < 6) R.C_dyn[inst * R.dyn_stride + r * n + c] = v;
            }
        }
    }
    for (int j = 0; j < R.n_tasks; ++j) {
        const double* J = R.J[j] + inst * R.J_rows[j] * nv;
        double* A = R.A_dst[j] + inst * R.A_stride[j];
        for (int r = 0; r < R.J_rows[j]; ++r)
            if (c < n) A[r * n + c] = (c < nv) ? J[r * nv + c] : 0.0;
    }
}

struct DevTorque {
    int B, nv, n_contacts, cdim, n, floating_base;
    const double* Bm; const double* h; const double* Jc;
    const double* x;       // [B][n] solved [qddot; F]
    double* tau;           // [B][nv]
    int* ok;               // [B] 1 = accepted (floating-base rows of tau within fb_tol), may be null
    double fb_tol;
};

// lane = joint row of tau.  B is symmetric (inertia matrix), so row r of B x is read as COLUMN r of B: lane r walks
// B[k][r], k = 0 .. nv-1 -- consecutive lanes read consecutive addresses (coalesced), x_k is an LDS broadcast.
__global__ void __launch_bounds__(64) osot_torque_kernel(const DevTorque T) {
    OSOT_STATIC_LDS(double, xs, 64);
    const long long inst = blockIdx.x;
    const int r = threadIdx.x;
    if (inst >= T.B) return;
    const int nv = T.nv, nf = T.n_contacts * T.cdim;
    xs[r] = (r < T.n) ? T.x[inst * T.n + r] : 0.0;
    wave_sync();
    const double* Bm = T.Bm + inst * nv * nv;
    const double* Jc = T.Jc + inst * nf * nv;
    double acc0 = 0.0, acc1 = 0.0;
    if (r < nv) {
        int k = 0;
        for (; k + 1 < nv; k += 2) {
            acc0 = fma(Bm[k * nv + r], xs[k], acc0);
            acc1 = fma(Bm[(k + 1) * nv + r], xs[k + 1], acc1);
        }
        if (k < nv) acc0 = fma(Bm[k * nv + r], xs[k], acc0);
        double t = (acc0 + acc1) + T.h[inst * nv + r];
        for (int f = 0; f < nf; ++f) t = fma(-Jc[f * nv + r], xs[nv + f], t);   // - sum_c Jc' F_c (InverseDynamics.cpp:73-77)
        T.tau[inst * nv + r] = t;
        acc0 = t;
    }
    if (T.ok) {
        // "Floating Base Wrench is not 0!" (InverseDynamics.cpp:83-92): |tau_i| > tolerance for one of the first six rows
        const bool bad = T.floating_base && r < 6 && r < nv && fabs(acc0) > T.fb_tol;
        const unsigned long long m = wave_ballot(bad);
        if (r == 0) T.ok[inst] = (m == 0ull) ? 1 : 0;
    }
}

}  // namespace osot

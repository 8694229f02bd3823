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

// GainType::Force of acceleration::Cartesian (src/tasks/acceleration/Cartesian.cpp:161-169): Mi = J Bi J' (:517-524), then
// Gp = Mi Kp, Gd = Mi Kd into the task's leaf array and a_ref += Mi f.  One wavefront per instance; lane = column of Bi for
// the product T = J Bi (rows x nv, through LDS), then lanes (r, s) < rows^2 each own one entry of the small products.
struct DevForceGains {
    int B, nv, rows;
    const double* J;       // [B][rows][nv]
    const double* Bi;      // [B][nv][nv] inverse inertia matrix (symmetric)
    double Kp[36], Kd[36]; // rows x rows, row-major
    const double* f;       // [B][rows] virtual force, may be null
    double* G; long long G_stride;   // Gp of instance 0 (Gd follows it); doubles between instances
    double* a_ref;         // [B][rows], += Mi f (may be null when f is)
};

__global__ void __launch_bounds__(64) osot_force_gains_kernel(const DevForceGains F) {
    OSOT_STATIC_LDS(double, Ts, 6 * 64);    // T = J Bi
    OSOT_STATIC_LDS(double, Js, 6 * 64);    // J
    OSOT_STATIC_LDS(double, Ms, 36);        // Mi
    const long long inst = blockIdx.x;
    const int c = threadIdx.x;
    if (inst >= F.B) return;
    const int nv = F.nv, R = F.rows;
    const double* J = F.J + inst * R * nv;
    const double* Bi = F.Bi + inst * (long long)nv * nv;
    for (int r = 0; r < R; ++r) Js[r * 64 + c] = (c < nv) ? J[r * nv + c] : 0.0;
    wave_sync();
    // T[r][c] = sum_k J[r][k] Bi[k][c]: lane c walks column c of Bi (coalesced rows), J[r][k] is an LDS broadcast
    double t[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (c < nv)
        for (int k = 0; k < nv; ++k) {
            const double b = Bi[k * nv + c];
#pragma unroll
            for (int r = 0; r < 6; ++r) if (r < R) t[r] = fma(Js[r * 64 + k], b, t[r]);
        }
#pragma unroll
    for (int r = 0; r < 6; ++r) if (r < R) Ts[r * 64 + c] = t[r];
    wave_sync();
    // Mi[r][s] = sum_c T[r][c] J[s][c]: lane (r, s)
    if (c < R * R) {
        const int r = c / R, sidx = c % R;
        double acc = 0.0;
        for (int k = 0; k < nv; ++k) acc = fma(Ts[r * 64 + k], Js[sidx * 64 + k], acc);
        Ms[c] = acc;
    }
    wave_sync();
    if (c < R * R) {
        const int r = c / R, sidx = c % R;
        double gp = 0.0, gd = 0.0;
        for (int k = 0; k < R; ++k) { gp = fma(Ms[r * R + k], F.Kp[k * R + sidx], gp); gd = fma(Ms[r * R + k], F.Kd[k * R + sidx], gd); }
        double* G = F.G + inst * F.G_stride;
        G[c] = gp;
        G[R * R + c] = gd;
    }
    if (F.f && F.a_ref && c < R) {
        double acc = 0.0;
        for (int k = 0; k < R; ++k) acc = fma(Ms[c * R + k], F.f[inst * R + k], acc);
        F.a_ref[inst * R + c] += acc;
    }
}

}  // namespace osot

/* oracle/osot_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-file CPU restatement of the reference's per-cycle hot path
 * (ADVRHumanoids/OpenSoT @2024-10-24): AutoStack::update() leaf/aggregate assembly and the
 * iHQP cascade behind Solver::solve(), with a Goldfarb-Idnani dual active-set QP that follows
 * the reference's vendored eiQuadProg back-end.  Every function cites the reference file:line
 * it restates.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (opensot_amd/csrc) never does.
 *
 * Parity pinning: see the header comment of osot_oracle.c.
 */
#ifndef OSOT_ORACLE_H
#define OSOT_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 8

/* "no bound" threshold: QPOasesBackEnd::checkINFTY clamps to +-1e20
 * (src/solvers/QPOasesBackEnd.cpp:339-356; qpOASES Constants.hpp:61). */
#define ORC_INFTY 1.0e20

/* ---------- assembled, batched stack data (host pointers) ----------
 * All arrays are instance-major: element [i][r][c] of a B x rows x n array lives at
 * ((i*rows)+r)*n + c. */
typedef struct {
    int n;                       /* number of variables                                   */
    int L;                       /* number of priority levels                              */
    int B;                       /* number of independent instances                        */
    int m[ORC_MAX_LEVELS];       /* task rows per level                                    */
    int ma[ORC_MAX_LEVELS];      /* rows of A_k stored explicitly; rows ma_k..m_k-1 (if any,
                                    then m_k-ma_k == n) are the identity block of a
                                    velocity::Postural task (Postural.cpp:37), never stored */
    const double* A[ORC_MAX_LEVELS]; /* [B][ma_k][n] (may be NULL when ma_k == 0)          */
    const double* b[ORC_MAX_LEVELS]; /* [B][m_k]                                           */
    const double* w[ORC_MAX_LEVELS]; /* [B][m_k] diagonal of W_k, NULL = identity          */
    const double* c[ORC_MAX_LEVELS]; /* [B][n] linear term, NULL = 0                       */
    int nc;                      /* global constraint rows (same at every level)           */
    const double* C;             /* [B][nc][n]                                             */
    const double* lo;            /* [B][nc]                                                */
    const double* up;            /* [B][nc]                                                */
    const double* l;             /* [B][n] box, NULL = none                                */
    const double* u;             /* [B][n]                                                 */
    double eps_abs;              /* absolute epsilon added to diag(H) by the back-end      */
    const unsigned char* active; /* [L] iHQP::setActiveStack flags, NULL = all active      */
    /* user regularisation task (AutoStack::setRegularisationTask, AutoStack.h:78-92): its cost is ADDED to the
     * cost of every level, H += Hr, g += gr (iHQP.cpp:265-266, 274-278); it never becomes an optimality row */
    int mr;                      /* rows of the regularisation task, 0 = none              */
    const double* Ar;            /* [B][mr][n]; NULL = [I_mr 0]                            */
    const double* br;            /* [B][mr]                                                */
    double wr;                   /* scalar weight: W_r = wr * I                            */
    /* task-local constraint rows (Task::getConstraints(), iHQP.cpp:190, 282-287): row r of C belongs to every level
     * when row_level[r] == 0 and to level k only when row_level[r] == k + 1 */
    const int* row_level;        /* [nc], NULL = all rows global                           */
    /* non-diagonal weight matrices (Task::setWeight(W), Task.h:273-300; Aggregated::generateWeight builds the level's
     * blockdiag, Aggregated.cpp:265-279): the FULL m_k x m_k matrix W_k; when given, w[k] is ignored */
    const double* Wd[ORC_MAX_LEVELS]; /* [B][m_k][m_k], NULL = diagonal (w[k])             */
} orc_batch;

/* back-end selection for the cascade */
enum {
    ORC_BE_EIQP_REFFORM = 0, /* eiQuadProgBackEnd's own form: CI=[I;-I;A;-A], no CE, psi tol 100 */
    ORC_BE_EIQP_EQ      = 1, /* same GI routine; lA==uA rows passed as CE, infinite sides skipped,
                                tight infeasibility tolerance (checker mode)                       */
    ORC_BE_QPOASES_REF  = 2  /* the reference's real qpOASES via oracle/_ref (dlopen)             */
};

/* ---------- Goldfarb-Idnani QP (restates external/eiQuadProg-ext/include/eiquadprog.hpp) ----
 * min 0.5 x'Gx + g0'x  s.t.  CE x + ce0 = 0 (p rows),  CI x + ci0 >= 0 (m rows).
 * G is n x n (symmetric, destroyed), CE/CI are row-per-constraint (the transpose of the
 * reference's column-per-constraint Eigen storage).  Returns the objective, +inf if infeasible.
 * psi_tol_scale: the reference uses 100.0 (eiquadprog.hpp:262); <0 selects the absolute
 * tolerance |psi_tol_scale| on each violated slack instead (checker mode). */
double orc_eiquadprog(int n, double* G, const double* g0, int p, const double* CE, const double* ce0,
                      int m, const double* CI, const double* ci0, double* x, double psi_tol_scale,
                      int* iterations, int* active_set /* m+p ints or NULL */, int* n_active);

/* one QP in OpenSoT BackEnd convention: min 0.5x'Hx+g'x, lA<=Ax<=uA, l<=x<=u (l,u may be NULL).
 * eps_abs is added to diag(H) (eiQuadProgBackEnd.cpp:64-65 / QPOasesBackEnd.cpp:253-255).
 * returns 1 on success, 0 if infeasible. */
int orc_backend_solve(int form, int n, const double* H, const double* g, int nc, const double* A,
                      const double* lA, const double* uA, const double* l, const double* u,
                      double eps_abs, double* x, int* iterations);

/* ---------- iHQP cascade (src/solvers/iHQP.cpp:129-170, 263-358) ---------- */
/* H = A'WA, g = -A'Wb + c for one instance of level k */
void orc_cost_function(const orc_batch* P, int inst, int k, double* H, double* g);
/* H += Hr, g += gr of the regularisation task (no-op when mr == 0) */
void orc_add_regularisation(const orc_batch* P, int inst, double* H, double* g);

/* Solve the cascade for instance `inst`. x_levels: [L][n] (may be NULL), dq: [n].
 * `be_state`: NULL, or persistent per-instance state created by orc_ref_state_create (qpOASES
 * hot-start objects).  returns 1 on success, 0 on failure at some level (dq = 0 then, like
 * the reference's callers do: examples/cpp/coman_ik.cpp:189-190). */
int orc_ihqp_solve(const orc_batch* P, int inst, int backend, void* be_state, double* x_levels,
                   double* dq, int* iterations);

/* batch driver with pthreads; cycles>1 re-solves the same data (hot-start path for qpOASES).
 * status[B] (1 ok / 0 failed).  returns seconds spent in the timed region (all cycles). */
double orc_ihqp_solve_batch(const orc_batch* P, int backend, int nthreads, int cycles,
                            double* dq /* [B][n] */, double* x_levels /* [B][L][n] or NULL */,
                            int* status, long long* total_iterations);

/* path to oracle/_ref/libqpoases_ref.so for ORC_BE_QPOASES_REF; returns 1 if it loads */
int orc_ref_load(const char* so_path);
/* eps_factor/termination tolerance used when creating qpOASES objects (defaults 2e2 / 0) */
void orc_ref_configure(double eps_factor, double termination_tolerance);

/* ---------- leaf updates (AutoStack::update -> Task::_update / Constraint::update) ---------- */
/* rotation matrix (row-major 3x3) -> quaternion (x,y,z,w), Eigen's conversion that
 * cartesian_utils.cpp:83-84 invokes through Eigen::Quaterniond(R). */
void orc_rot_to_quat(const double* R, double* q);
/* cartesian_utils::computeCartesianError, src/utils/cartesian_utils.cpp:79-96 +
 * quaternion::error include/OpenSoT/utils/cartesian_utils.h:144-164 */
void orc_cartesian_error(const double* R, const double* p, const double* Rd, const double* pd,
                         double* pos_err, double* ori_err);
/* velocity::Cartesian::update_b, src/tasks/velocity/Cartesian.cpp:279-285 */
void orc_cartesian_b(const double* R, const double* p, const double* Rd, const double* pd,
                     const double* twist_des, double lambda, double orientation_gain, double* b6);
/* the same with a body Jacobian: b rotated by Ad(R'), Cartesian.cpp:93-100 */
void orc_cartesian_b_body(const double* R, const double* p, const double* Rd, const double* pd,
                          const double* twist_des, double lambda, double orientation_gain, double* b6);
/* velocity::CoM::update_b, src/tasks/velocity/CoM.cpp:145-149 */
void orc_com_b(const double* p, const double* pd, const double* v_des, double lambda, double* b3);
/* velocity::Postural::update_b, src/tasks/velocity/Postural.cpp:97-100 (Euclidean joints) */
void orc_postural_b(int n, const double* q, const double* q_des, const double* v_des, double lambda,
                    double* b);
/* velocity::JointLimits::update, src/constraints/velocity/JointLimits.cpp:37-58
 * (dq = q - q_neutral is passed in as q, Euclidean joints) */
void orc_joint_limits(int n, const double* q, const double* qmin, const double* qmax, double scaling,
                      double* l, double* u);
/* velocity::VelocityLimits::generateBounds, src/constraints/velocity/VelocityLimits.cpp:72-89 */
void orc_velocity_limits(int n, const double* qdot_max, double dT, double* l, double* u);
/* constraints::Aggregated box merge, src/constraints/Aggregated.cpp:141-148 */
void orc_merge_box(int n, double* l, double* u, const double* l2, const double* u2);
/* velocity::CollisionAvoidance::update rows, src/constraints/velocity/CollisionAvoidance.cpp:96-152.
 * Jd: [P][n] distance Jacobians ordered by distance, d: [P]; rows beyond the detection threshold are
 * skipped; unused rows are zero with [-DBL_MAX, +DBL_MAX]. */
void orc_collision_rows(int n, int P, int max_pairs, const double* Jd, const double* d,
                        double d_threshold, double detection_threshold, double bound_scaling,
                        double* Aineq, double* lo, double* up);


/* ---------- inverse-dynamics row producers (x = [qddot; forces]) ---------- */
/* acceleration::Cartesian / CoM, src/tasks/acceleration/Cartesian.cpp:152-160, CoM.cpp:86-92 with Kp = Kd = I:
 * b = a_ref + lambda2*vel_err + lambda*pose_err - Jdot*qdot (rows = 6 or 3; a_ref may be NULL) */
void orc_acc_task_b(int rows, const double* pose_err, const double* vel_err, const double* jdotqdot,
                    const double* a_ref, double lambda, double lambda2, double* b);
/* acceleration::Postural, src/tasks/acceleration/Postural.cpp:145-158 (Acceleration gain type) */
void orc_acc_task_b_gains(int rows, const double* pose_err, const double* vel_err, const double* jdotqdot,
                          const double* a_ref, const double* Gp, const double* Gd, double lambda, double lambda2, double* b);
void orc_cartesian_inertia_inverse(int rows, int nv, const double* J, const double* Bi, double* Mi);
void orc_acc_postural_b(int rows, const double* q_err, const double* qdot_err, const double* qddot_ref,
                        double lambda, double lambda2, double* b);
/* acceleration::TorqueLimits bounds, src/constraints/acceleration/TorqueLimits.cpp:44-45 */
void orc_torque_limit_bounds(int rows, const double* h, const double* tau_max, double* lo, double* up);
/* force::FrictionCone rows for one contact, src/constraints/force/FrictionCone.cpp:35-56:
 * A53 (5x3 row-major) = Ci(mu/sqrt2) * wRl' */
void orc_friction_cone_rows(const double* wRl, double mu, double* A53);
/* acceleration::JointLimits bounds, src/constraints/acceleration/JointLimits.cpp:58-131 */
void orc_acc_joint_limits(int rows, const double* q, const double* qdot, const double* qmin, const double* qmax,
                          const double* qddot_max, double dt, double* lo, double* up);
/* acceleration::VelocityLimits bounds, src/constraints/acceleration/VelocityLimits.cpp:50-63 */
void orc_acc_velocity_limits(int rows, const double* qdot, const double* qdot_max, double dT, double p,
                             double* lo, double* up);

#ifdef __cplusplus
}
#endif
#endif

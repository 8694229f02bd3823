/* include/osot_mi355x.h -- the drop-in C-ABI of the MI355X-native OpenSoT hot path.
 *
 * Plain C: opaque handles, POD structs, raw pointers + sizes, int error codes; no C++/Eigen/torch
 * types cross this boundary and no exception does.  Implemented by
 * opensot_amd/csrc/libosot_mi355x.so (hand-written HIP for gfx950).
 *
 * What each group replaces in the reference (ADVRHumanoids/OpenSoT @2024-10-24; paths relative to
 * the reference root):
 *
 *  osot_plan_desc ................ the *result* of the stack algebra (src/utils/AutoStack.cpp:7-331,
 *                                  examples/cpp/coman_ik.cpp:425-449): levels -> leaf task blocks,
 *                                  box-bound producers, global constraint-row producers.  Static.
 *  osot_stack_update ............. AutoStack::update()            src/utils/AutoStack.cpp:385-393
 *                                  (-> Task::update Task.h:375-400, tasks::Aggregated::_update
 *                                  src/tasks/Aggregated.cpp:92-132, constraints::Aggregated::update
 *                                  src/constraints/Aggregated.cpp:60-257 and the leaf _update()s)
 *  osot_ihqp_solve ............... Solver::solve() = iHQP::solve  src/solvers/iHQP.cpp:263-358
 *                                  (computeCostFunction :129-162, optimality rows :164-170,
 *                                  one BackEnd::solve per level, QPOasesBackEnd.cpp:248-307)
 *  osot_backend_* ................ the BackEnd plugin surface     include/OpenSoT/solvers/BackEnd.h:23-171
 *                                  and its C factory `create_instance`
 *                                  (src/solvers/QPOasesBackEnd.cpp:14-24), batch-of-one, host pointers
 *  osot_kinematics ............... batched frame poses / Jacobians / CoM: what the leaf _update()s fetch from
 *                                  XBot::ModelInterface (velocity/Cartesian.cpp:73-81, velocity/CoM.cpp:59-74)
 *  osot_allgather_dq ............. new surface (the reference has no multi-instance API): collects the
 *                                  per-rank dq shards; RCCL all-gather on the solve stream
 *
 * All batched arrays are INSTANCE-MAJOR, row-major inside an instance, fp64, resident in HBM:
 *   X[B][rows][n]  ->  element (i, r, c) at ((i*rows)+r)*n + c.
 * One wavefront solves one instance, so an instance's rows are a single contiguous, coalesced read.
 */
#ifndef OSOT_MI355X_H
#define OSOT_MI355X_H

#ifdef __cplusplus
extern "C" {
#endif

#define OSOT_MAX_LEVELS 8
#define OSOT_MAX_TASKS 8      /* leaf task blocks per level   */
#define OSOT_MAX_BOUNDS 4     /* box-bound producers          */
#define OSOT_MAX_ROWBLOCKS 8  /* constraint-row blocks (global and task-local) */
#define OSOT_MAX_BAND_ROWS 6  /* rows of a task used as a constraint (TaskToConstraint error band) */
#define OSOT_MAX_VARS 64      /* one lane per variable        */
#define OSOT_MAX_QP_VARS 128  /* the explicit-QP surface (osot_qp_solve_batch, osot_backend_*): 65 .. 128 variables run one
                                 256-thread workgroup per QP instead of one wavefront (round 6; cold start, no graph capture);
                                 plans (osot_solver_create), nHQP / eHQP and the ADMM kernel stay at OSOT_MAX_VARS */

/* error codes (the reference returns bool / throws std::runtime_error) */
enum {
    OSOT_OK = 0,
    OSOT_ERR_INVALID = 1,      /* bad argument / size mismatch (BackEnd.cpp:23-27, :47-60)          */
    OSOT_ERR_UNSUPPORTED = 2,  /* plan uses something this build does not implement                 */
    OSOT_ERR_HIP = 3,          /* a HIP runtime call failed (osot_last_error() has the text)        */
    OSOT_ERR_NOT_SOLVED = 4,   /* batch-of-one back-end: QP infeasible / iteration limit            */
    OSOT_ERR_COMM = 5          /* RCCL failure                                                       */
};

/* per-instance status written by osot_ihqp_solve (the reference's `bool` per solve,
 * iHQP.cpp:279-347; failed instances return dq = 0 like examples/cpp/coman_ik.cpp:189-190) */
enum {
    OSOT_STATUS_SOLVED = 0,
    OSOT_STATUS_INFEASIBLE = 1,
    OSOT_STATUS_MAX_ITER = 2,
    OSOT_STATUS_NOT_PD = 3      /* Cholesky of H + eps I broke down */
};

/* ---- static stack plan ---------------------------------------------------------------------- */
typedef enum {
    OSOT_TASK_GENERIC = 0,   /* tasks::GenericTask (src/tasks/GenericTask.cpp:5-56): A and b supplied   */
    OSOT_TASK_CARTESIAN = 1, /* velocity::Cartesian (src/tasks/velocity/Cartesian.cpp:68-105, 279-285)  */
    OSOT_TASK_COM = 2,       /* velocity::CoM (src/tasks/velocity/CoM.cpp:59-74, 145-149)               */
    OSOT_TASK_POSTURAL = 3,  /* velocity::Postural (src/tasks/velocity/Postural.cpp:50-62, 97-100);
                                A = [I_rows 0] is implicit (never stored), rows <= n; must be the last block of
                                its level */
    /* inverse-dynamics formulation, x = [qddot; contact forces] (src/utils/InverseDynamics.cpp:12-28).  The
     * producer writes the task matrix ([J 0] ...) into A_k; the update computes b from the supplied errors. */
    OSOT_TASK_ACC_CARTESIAN = 4, /* acceleration::Cartesian (src/tasks/acceleration/Cartesian.cpp:127-180):
                                    b = a_ref + lambda2*Kd*vel_err + lambda*Kp*pose_err - Jdot*qdot (Kp = Kd = I
                                    unless osot_task_desc.acc_gain_matrices) */
    OSOT_TASK_ACC_COM = 5,       /* acceleration::CoM (src/tasks/acceleration/CoM.cpp:74-97), 3 rows */
    OSOT_TASK_ACC_POSTURAL = 6   /* acceleration::Postural (src/tasks/acceleration/Postural.cpp:135-163):
                                    A = [I_rows 0] implicit like OSOT_TASK_POSTURAL,
                                    b = qddot_ref + lambda2*(qdot_ref - qdot) + lambda*(q_ref - q) */
} osot_task_kind;

typedef struct {
    int kind;                /* osot_task_kind */
    int rows;                /* 6 / 3 / n / user-defined */
    double weight;           /* scalar on this block's W: `0.1*l_wrist` (AutoStack.cpp:16-47) */
    double lambda;           /* Task::setLambda; enters b only (Cartesian.cpp:284, CoM.cpp:148) */
    double orientation_gain; /* Cartesian::setOrientationErrorGain (Cartesian.cpp:283) */
    double lambda2;          /* velocity gain of the acceleration tasks (acceleration/Cartesian.cpp:158) */
    /* SubTask (src/tasks/SubTask.cpp:22-112, `task % {indices}`): this block keeps only some rows of a parent task.
     * row_mask bit i = row i of the parent is kept (0: not a sub-task); rows = popcount(row_mask); parent_rows = rows
     * of the parent (0: the kind's own size: 6 / 3 / n); b = sub_lambda * b_parent[kept rows] (SubTask.cpp:56-57);
     * the weight is the parent's scalar.  The producer writes the kept Jacobian rows; a Postural sub-task is NOT
     * implicit: its unit rows are stored like any other block's. */
    unsigned long long row_mask;
    int parent_rows;
    double sub_lambda;
    /* velocity::Cartesian with a BODY Jacobian (Cartesian::setIsBodyJacobian, src/tasks/velocity/Cartesian.cpp:93-100):
     * A and b are both rotated by Ad(R') with R the actual orientation.  The producer writes the rotated Jacobian
     * (osot_kin_desc.frame_body does); the update rotates b.  0 = world frame. */
    int body_frame;
    /* non-diagonal weight matrix (Task::setWeight(W) with a full W, include/OpenSoT/Task.h:273-300; in an aggregate the
     * level's W is blockdiag(weight_i * W_i), src/tasks/Aggregated.cpp:265-279).  1: the block's W_i [B][rows][rows]
     * comes with the leaf inputs (osot_leaf_ptrs.W); the update then forms W_k A_k and W_k b_k for the whole level
     * (osot_assembled_out.WA / Wb) and the cascade takes them as the left operand of H = A'(WA), g = -A'(Wb).  A
     * Postural block with dense_weight is stored like any other block (its unit rows are written by the producer). */
    int dense_weight;
    /* gain MATRICES of the acceleration tasks (acceleration::Cartesian::setGains(Kp, Kd), setKp / setKd,
     * src/tasks/acceleration/Cartesian.cpp:152-173; round 3 -- before: Kp = Kd = I).  1: the task's leaf array p0 carries
     * them per instance behind the errors, p0 = [pose_err (rows); vel_err (rows); Gp (rows x rows, row-major); Gd (rows x
     * rows)], and b = a_ref + lambda2 Gd vel_err + lambda Gp pose_err - Jdot qdot.
     *   GainType::Acceleration (:155-160): Gp = Kp, Gd = Kd, the same for every instance.
     *   GainType::Force        (:161-169): Gp = Mi Kp, Gd = Mi Kd with Mi = J B^-1 J' the inverse Cartesian inertia, and the
     *     virtual force enters as a_ref + Mi f -- osot_id_force_gains below writes all three from J, B^-1, Kp, Kd, f.
     * Only OSOT_TASK_ACC_CARTESIAN / OSOT_TASK_ACC_COM; 0 = scalar gains. */
    int acc_gain_matrices;
} osot_task_desc;

typedef struct {
    int n_tasks;
    osot_task_desc task[OSOT_MAX_TASKS]; /* tasks::Aggregated row order (Aggregated.cpp:113-132) */
} osot_level_desc;

typedef enum {
    OSOT_BOUND_GENERIC = 0,        /* l,u supplied */
    OSOT_BOUND_JOINT_LIMITS = 1,   /* velocity::JointLimits (src/constraints/velocity/JointLimits.cpp:37-58) */
    OSOT_BOUND_VELOCITY_LIMITS = 2 /* velocity::VelocityLimits (…/VelocityLimits.cpp:72-89) */
} osot_bound_kind;

typedef struct {
    int kind;       /* osot_bound_kind */
    double scaling; /* JointLimits boundScaling */
    double dT;      /* VelocityLimits dT */
} osot_bound_desc;

typedef enum {
    OSOT_ROWS_GENERIC = 0,  /* constraints::GenericConstraint / TaskToConstraint rows: C, lo, up supplied */
    OSOT_ROWS_COLLISION = 1,/* velocity::CollisionAvoidance rows (…/CollisionAvoidance.cpp:96-152):
                               -J_d dq <= max(0, s (d - d_min)) for pairs within the detection threshold */
    /* inverse-dynamics constraint rows on x = [qddot; forces] */
    OSOT_ROWS_DYN_FEASIBILITY = 2, /* acceleration::DynamicFeasibility as an equality (DynamicFeasibility.cpp:22-46):
                                      producer writes [B_u, -J_f'] (6 rows); lo = up = -h_u */
    OSOT_ROWS_TORQUE_LIMITS = 3,   /* acceleration::TorqueLimits (src/constraints/acceleration/TorqueLimits.cpp:25-46):
                                      producer writes [B, -Jc'] ; lo = -tau_max - h, up = tau_max - h */
    OSOT_ROWS_FRICTION_CONE = 4,   /* force::FrictionCone (src/constraints/force/FrictionCone.cpp:35-56): 5 rows per
                                      contact = pyramid(mu/sqrt2) * wRl' on the contact's 3 force columns,
                                      lo = -1e20, up = 0; rows = 5 * contacts, first_col = first force column */
    OSOT_ROWS_ACC_JOINT_LIMITS = 5,/* acceleration::JointLimits (src/constraints/acceleration/JointLimits.cpp:58-176):
                                      unit rows e_(first_col+i) (NOT stored), bounds from q, qdot, limits */
    OSOT_ROWS_ACC_VELOCITY_LIMITS = 6,/* acceleration::VelocityLimits (…/VelocityLimits.cpp:50-63): unit rows
                                      (NOT stored), (qdot_lim - qdot)/(dT*p) */
    /* constraints::TaskToConstraint (src/constraints/TaskToConstraint.cpp:25-68; `stack << l_sole` in
     * examples/cpp/coman_ik.cpp:430-449): the rows are the task's A (written by the producer into C, like the task's
     * Jacobian into A_k), the bounds are the task's b computed HERE from the task's leaf inputs, widened by the block's
     * error band: lo = b + err_lb, up = b + err_ub (0, 0: the task as an equality).  task_lambda /
     * task_orientation_gain are the task's gains. */
    OSOT_ROWS_TASK_CARTESIAN = 7,     /* velocity::Cartesian as a constraint: 6 rows; leaf as for OSOT_TASK_CARTESIAN */
    OSOT_ROWS_TASK_COM = 8,           /* velocity::CoM as a constraint: 3 rows; leaf as for OSOT_TASK_COM */
    OSOT_ROWS_UNIT_GENERIC = 9        /* unit rows e_(first_col+i) (NOT stored), lo / up supplied: a box on a range of
                                         variables as rows.  With only_level = k + 1 this is a TASK-LOCAL BOUND
                                         (`task << joint_limits`: iHQP merges a level's own bounds into that level's
                                         box only, iHQP.cpp:190, 336-340) */
} osot_rows_kind;

typedef struct {
    int kind;  /* osot_rows_kind */
    int rows;  /* for COLLISION: max_pairs */
    double d_threshold, detection_threshold, bound_scaling;
    int first_col;   /* unit-row / friction-cone blocks: column of the block's first variable */
    double dT, p;    /* acceleration limits: time step and horizon factor (dt = dT*p) */
    double mu;       /* friction coefficient */
    double task_lambda, task_orientation_gain;   /* OSOT_ROWS_TASK_*: gains of the underlying task */
    double err_lb[OSOT_MAX_BAND_ROWS], err_ub[OSOT_MAX_BAND_ROWS];   /* OSOT_ROWS_TASK_*: TaskToConstraint's error band,
                        one pair per row (err_lb / err_ub are vectors, src/constraints/TaskToConstraint.cpp:34-52, 61-68) */
    int task_body_frame;   /* OSOT_ROWS_TASK_CARTESIAN: the task has a body Jacobian (see osot_task_desc.body_frame) */
    int n_candidates;      /* OSOT_ROWS_COLLISION: collision pairs supplied per instance (0 = rows).  With more candidates
                              than rows the `rows` CLOSEST pairs are taken, in order of distance
                              (getOrderedCollisionPairIndices, src/constraints/velocity/CollisionAvoidance.cpp:120-131) */
    int only_level;  /* 0 = global rows: constrain every level (AutoStack `<<`, iHQP.cpp:191-193).  k + 1 = TASK-LOCAL
                        rows of level k (`task << constraint`, Task::getConstraints(), iHQP.cpp:190, 282-287): they
                        constrain the QP of level k only; at every other level they are absent */
} osot_rows_desc;

typedef struct {
    int n;         /* number of variables (<= OSOT_MAX_VARS) */
    int n_levels;
    osot_level_desc level[OSOT_MAX_LEVELS];
    int n_bounds;  /* 0 = no box (iHQP.cpp:340-344 then never calls updateBounds) */
    osot_bound_desc bound[OSOT_MAX_BOUNDS];
    int n_rowblocks;
    osot_rows_desc rowblock[OSOT_MAX_ROWBLOCKS];
    double eps_abs;  /* absolute epsilon on diag(H): 1e3 * 2.221e-16 * eps_factor
                        (QPOasesBackEnd.cpp:57,67; qpOASES Options.cpp:147) */
    int max_iter;    /* active-set iteration cap per level; 0 = default (the reference's nWSR is 13200,
                        QPOasesBackEnd.cpp:30) */
    /* user regularisation task (AutoStack::setRegularisationTask, include/OpenSoT/utils/AutoStack.h:78-92): iHQP adds
     * its cost to the cost of EVERY level, H += Hr, g += gr (iHQP.cpp:265-266, 274-278); it never becomes an
     * optimality row.  Supported: identity-Jacobian tasks A_r = [I_rows 0] with W_r = weight * I, i.e.
     * OSOT_TASK_GENERIC (b supplied: GenericTask(I, b) as in tests/solvers/TestiHQP.cpp:118-120, MinimumVelocity
     * with b = 0), OSOT_TASK_POSTURAL and OSOT_TASK_ACC_POSTURAL; row_mask must be 0.  Then Hr = weight * [I_rows 0;
     * 0 0] is folded into the diagonal and gr = -weight * b_r.  (A stored Jacobian: regularisation_dense below.) */
    int has_regularisation;
    osot_task_desc regularisation;
    /* regularisation task with a STORED Jacobian (round 3; iHQP.cpp:265-278 takes any task): 1 = A_r is the
     * [B][regularisation.rows][n] array osot_qp_batch.A_reg (written by the producer, like a level's A_k), W_r = weight * I,
     * b_r from the update like any task of its kind (TASK_GENERIC, TASK_CARTESIAN, TASK_COM); H += A_r'W_r A_r and
     * g -= A_r'W_r b_r at every level, never an optimality row.  0 = the identity-Jacobian form above. */
    int regularisation_dense;
} osot_plan_desc;

/* ---- assembled, batched QP data (device pointers) ------------------------------------------ */
typedef struct {
    int B;                              /* instances in this call (<= max_batch of the solver) */
    const double* A[OSOT_MAX_LEVELS];   /* [B][ma_k][n], ma_k = rows of the non-Postural blocks */
    const double* b[OSOT_MAX_LEVELS];   /* [B][m_k] */
    const double* w[OSOT_MAX_LEVELS];   /* [B][m_k] diagonal of W_k; NULL = identity */
    const double* c[OSOT_MAX_LEVELS];   /* [B][n] Task::getc(); NULL = 0 */
    const double* C;                    /* [B][nc_stored][n] global rows: the STORED row blocks stacked in plan
                                           order (unit-row blocks have no storage, see osot_plan_constraint_rows) */
    const double* lo;                   /* [B][nc] (all rows, stored or not) */
    const double* up;                   /* [B][nc] */
    const double* l;                    /* [B][n] merged box; NULL iff plan.n_bounds == 0 */
    const double* u;                    /* [B][n] */
    const unsigned char* level_active;  /* HOST pointer, [n_levels] iHQP::setActiveStack
                                           (iHQP.cpp:391-395); NULL = all active */
    double* dq;                         /* out [B][n]: x of the last active level (iHQP.cpp:349) */
    double* x_levels;                   /* out [B][n_levels][n], may be NULL */
    int* status;                        /* out [B] OSOT_STATUS_* */
    int* iterations;                    /* out [B] active-set iterations summed over levels; may be NULL */
    const double* b_reg;                /* [B][regularisation.rows] b of the regularisation task; NULL iff the plan
                                           has none */
    const double* WA[OSOT_MAX_LEVELS];  /* [B][ma_k][n] W_k A_k and                                              */
    const double* Wb[OSOT_MAX_LEVELS];  /* [B][m_k]     W_k b_k of a level with a non-diagonal W_k (written by
                                           osot_stack_update); NULL = diagonal W_k (w[k])                          */
    double* accepted_slack;             /* out [B], may be NULL: the largest constraint violation that a level of the
                                           instance accepted as round-off of the levels above it (no direction left and
                                           no multiplier to trade; at most min(1e-6 * max(1, |bound|), 1e-5)); 0 = none.
                                           The status stays OSOT_STATUS_SOLVED, like the reference's `true` when qpOASES
                                           stops inside its own tolerances */
    const double* A_reg;                /* [B][regularisation.rows][n] Jacobian of the regularisation task; read iff
                                           plan.regularisation_dense */
} osot_qp_batch;

/* ---- leaf inputs of AutoStack::update (device pointers) ------------------------------------ */
/* meaning of (p0, p1, p2) by kind:
 *   TASK_CARTESIAN : p0 = actual pose  [B][12] (R row-major 9, then p 3), p1 = desired pose [B][12],
 *                    p2 = desired twist [B][6] (NULL = 0)
 *   TASK_COM       : p0 = actual CoM [B][3], p1 = desired CoM [B][3], p2 = desired velocity [B][3] (NULL = 0)
 *   TASK_POSTURAL  : p0 = q [B][n], p1 = q_desired [B][n], p2 = v_desired [B][n] (NULL = 0)
 *   TASK_GENERIC   : p0 = b [B][rows] (copied);  p1, p2 unused
 *   BOUND_JOINT_LIMITS    : p0 = q - q_neutral [B][n], p1 = q_min [B][n], p2 = q_max [B][n]
 *   BOUND_VELOCITY_LIMITS : p0 = qdot_max [B][n]
 *   BOUND_GENERIC         : p0 = l [B][n], p1 = u [B][n]
 *   ROWS_COLLISION : p0 = distance Jacobians J_d [B][rows][n] (ordered by distance), p1 = distances [B][rows]
 *   ROWS_GENERIC   : p0 = C [B][rows][n], p1 = lo [B][rows], p2 = up [B][rows]
 *   TASK_ACC_CARTESIAN / TASK_ACC_COM : p0 = [pose_err ; vel_err] [B][2*rows], p1 = Jdot*qdot [B][rows],
 *                    p2 = a_ref [B][rows] (NULL = 0)
 *   TASK_ACC_POSTURAL : p0 = [q_ref - q ; qdot_ref - qdot] [B][2*rows], p2 = qddot_ref [B][rows] (NULL = 0)
 *   ROWS_DYN_FEASIBILITY : p0 = h_u [B][6]            (the 6 x n rows are written by the producer into C)
 *   ROWS_TORQUE_LIMITS   : p0 = h [B][rows], p1 = tau_max [B][rows]   (rows written by the producer into C)
 *   ROWS_FRICTION_CONE   : p0 = contact rotations wRl [B][contacts][9] (row-major)
 *   ROWS_ACC_JOINT_LIMITS    : p0 = [q ; qdot] [B][2*rows], p1 = [q_min ; q_max] [B][2*rows], p2 = qddot_max [B][rows]
 *   ROWS_ACC_VELOCITY_LIMITS : p0 = qdot [B][rows], p1 = qdot_max [B][rows]
 *   ROWS_UNIT_GENERIC : p0 = lo [B][rows], p1 = up [B][rows]
 *   ROWS_TASK_CARTESIAN / ROWS_TASK_COM : as TASK_CARTESIAN / TASK_COM (the 6 / 3 rows are written by the producer into C)
 * Task Jacobians are NOT passed here: the producer writes them straight into their row range of
 * osot_qp_batch.A[k] (zero-copy stacking; the reference copies them twice through MatrixPiler,
 * src/tasks/Aggregated.cpp:113-132).
 *   any task with dense_weight : W = the block's weight matrix [B][rows][rows] (row-major) */
typedef struct { const double *p0, *p1, *p2, *W; } osot_leaf_ptrs;

typedef struct {
    int B;
    osot_leaf_ptrs task[OSOT_MAX_LEVELS][OSOT_MAX_TASKS];
    osot_leaf_ptrs bound[OSOT_MAX_BOUNDS];
    osot_leaf_ptrs rows[OSOT_MAX_ROWBLOCKS];
    osot_leaf_ptrs regularisation;      /* leaf inputs of the regularisation task (same meaning as for its kind) */
} osot_leaf_batch;

/* writable view of the assembled arrays that osot_stack_update fills (same shapes as osot_qp_batch) */
typedef struct {
    double* b[OSOT_MAX_LEVELS];
    double* w[OSOT_MAX_LEVELS];         /* NULL = leave the weights alone: for a DIAGONAL W_k with per-row entries
                                           (Task::setWeight(W), Aggregated.cpp:265-279) fill osot_qp_batch.w[k] once */
    double* C;
    double* lo;
    double* up;
    double* l;
    double* u;
    double* b_reg;                      /* [B][regularisation.rows]; NULL iff the plan has no regularisation task */
    double* WA[OSOT_MAX_LEVELS];        /* [B][ma_k][n], [B][m_k]: required for the levels that hold a dense_weight block  */
    double* Wb[OSOT_MAX_LEVELS];
    const double* A[OSOT_MAX_LEVELS];   /* the stacked Jacobians A_k the producer wrote (read to form W_k A_k); required for
                                           the same levels */
} osot_assembled_out;

typedef struct osot_solver osot_solver;

/* library / device */
const char* osot_version(void);
const char* osot_last_error(void);           /* thread-local text of the last failure */
int osot_device_count(int* count);

/* plan-derived sizes (host only, no GPU needed) */
int osot_plan_validate(const osot_plan_desc* plan);
int osot_plan_level_rows(const osot_plan_desc* plan, int level, int* m_total, int* m_stored);
int osot_plan_constraint_rows(const osot_plan_desc* plan, int* nc);
int osot_plan_stored_constraint_rows(const osot_plan_desc* plan, int* nc_stored);

/* solver object: owns the per-plan device workspace for up to max_batch instances on `device` */
int osot_solver_create(const osot_plan_desc* plan, int max_batch, int device, osot_solver** out);
int osot_solver_destroy(osot_solver* s);

/* AutoStack::update() for B instances: leaf inputs -> b_k, w_k, merged box, constraint rows.
 * Stream-ordered on `hip_stream` (a hipStream_t passed as void*, NULL = default stream). */
int osot_stack_update(osot_solver* s, const osot_leaf_batch* leaf, const osot_assembled_out* out,
                      void* hip_stream);

/* Solver::solve() for B instances: the whole iHQP cascade in one launch. Stream-ordered. */
int osot_ihqp_solve(osot_solver* s, const osot_qp_batch* batch, void* hip_stream);

/* One control cycle in ONE launch: osot_stack_update followed by osot_ihqp_solve for the same B instances
 * (`stack->update(); solver->solve(dq)`, examples/cpp/coman_ik.cpp:186-192) with each instance's update and cascade run by
 * the same wavefront.  Results are identical to the two calls; the assembled arrays are still written. */
int osot_cycle(osot_solver* s, const osot_leaf_batch* leaf, const osot_assembled_out* out, const osot_qp_batch* batch,
               void* hip_stream);

/* average device time (ms) of the cascade kernel over the launches recorded since the last call
 * with reset != 0; measured with hipEvents on the launch stream. *launches receives the count. */
int osot_solver_kernel_time_ms(osot_solver* s, int reset, double* avg_ms, int* launches);
int osot_solver_set_timing(osot_solver* s, int enabled);   /* 0 off, 1 every launch, k > 1: every k-th launch (two event
                                                             * packets per timed launch sit in the stream's critical path) */
/* Dispatch order of the instances inside osot_ihqp_solve.  One wavefront solves one instance and the chip holds a
 * fixed number of them, so a batch runs in rounds and its tail is set by the slowest late starter.
 * OSOT_SCHEDULE_LONGEST_FIRST (default) dispatches in descending order of each instance's active-set iteration
 * count in the PREVIOUS solve of a batch of the same size on this solver (control loops are temporally coherent);
 * the first solve runs in order.  Results do not depend on the mode: the reference solves the robots of a batch
 * independently (one iHQP per robot, coman_ik.cpp:425-470). */
#define OSOT_SCHEDULE_IN_ORDER 0
#define OSOT_SCHEDULE_LONGEST_FIRST 1
int osot_solver_set_schedule(osot_solver* s, int mode);
/* Hot start of the working sets across control cycles (reference: QPOasesBackEnd.cpp:258-285 tries hotstart() first;
 * external/qpOASES-ext/src/SQProblem.cpp:149-193).  enabled != 0: every level of every instance records the inequality
 * working set it ends with, and the next osot_ihqp_solve / osot_cycle of THIS solver re-adds it before the first
 * violation scan (signed steps, then the constraints whose multipliers came out negative are taken out again); state is
 * per instance index, so instance i of consecutive batches should be the same robot.  Calling it (again) with
 * enabled != 0 forgets the recorded sets; 0 (default) = every solve is a cold start and nothing is recorded.  The answer
 * does not depend on the mode beyond round-off (each level's minimiser is unique); the iteration counts do: it pays
 * when the active sets persist from cycle to cycle and costs two iterations per constraint that has left the set. */
int osot_solver_set_hotstart(osot_solver* s, int enabled);
/* Kernel instantiation by plan structure.  enabled != 0 (default): a plan WITHOUT constraint rows (the bounds l <= x <= u are
 * its only inequalities: a velocity stack with joint / velocity limits, BASELINE configs 2 and 3) of at most 32 variables runs
 * the instantiation of the cascade that carries no constraint-row code (row classification, row scans, row normals) -- and so
 * does, at every size up to 64 variables, a plan whose constraint rows are all TaskToConstraint blocks with a point band
 * (OSOT_ROWS_TASK_* with err_lb == err_ub: `stack << l_sole`, the feet of the reference's COMAN stacks,
 * examples/cpp/coman_ik.cpp:425-449): those rows are equalities of every level, the bounds stay the only inequalities;
 * 0: every plan runs the general instantiation.  The two are the same arithmetic in the same order: results are bit-identical
 * (tests/test_gpu_cascade.py), only the speed differs.  (The reference has nothing to mirror here: it is how one BackEnd
 * covers plans of different shape without paying for the features a plan does not use.) */
int osot_solver_set_specialisation(osot_solver* s, int enabled);
/* instances the device works on at once for this solver's plan (one wavefront each: CUs x resident wavefronts per CU, from
 * the kernel's register and LDS footprint): the dispatch order is planned for it, and batches that are a multiple of it
 * waste no round */
int osot_solver_resident_waves(osot_solver* s, int* waves);
/* Task::setActive (include/OpenSoT/Task.h:232-239, 375-400): an inactive task's A is zero -- it adds nothing to H and g
 * of its level and its optimality rows are void for the levels below.  The producer's Jacobian rows stay untouched in
 * A_k; the cascade ignores them.  Takes effect at the next osot_ihqp_solve / osot_cycle / osot_ehqp_solve (the equality-only
 * front-end gives the task's rows weight zero, i.e. treats the task as absent; a task beyond row 64 of its level is refused
 * there).  osot_nhqp_solve does not take inactive tasks.  (Column masks, Task::setActiveJointsMask, are the producer's:
 * osot_kin_desc.frame_col_mask.) */
int osot_solver_set_task_active(osot_solver* s, int level, int task, int active);
/* diagnostic: run the cascade once through the instrumented instantiation of the kernel and write, per
 * instance, OSOT_N_PHASES shader-clock cycle counts (H/g build, Cholesky, L^-1, substitution, equality
 * phase, inequality loop, optimality rhs, total, then four sub-phases of the equality adds: J'a,
 * reductions, step direction, Householder update; and six of the inequality loop: violation scan, d = J'n,
 * step direction z, dual direction r + step lengths, Householder add, drop) to cycles[B][OSOT_N_PHASES]
 * (device, int64). */
#define OSOT_N_PHASES 18
int osot_solver_profile_phases(osot_solver* s, const osot_qp_batch* batch, long long* cycles, void* hip_stream);

/* ---- the null-space front-end: OpenSoT::solvers::nHQP (src/solvers/nHQP.cpp:155-204, 236-317, 357-390) ----
 * Same stack, same assembled arrays as osot_ihqp_solve; per level the task is projected into the cumulated null space N of
 * the levels above, an SVD of A N drives the reference's A/b regularisation and yields the level's null space, the QP is
 * solved in the nf free coordinates (bounds become rows N z, compute_contraints :282-317) and q += N z, N <- N V2.  Three
 * launches per level (prepare, the batched QP kernel, accumulate).  n <= 64 (a level with min(rows, free variables) <= 32 goes
 * through a 32-wide eigen-decomposition; round 5: the others -- the reference's one-level stack S1, 50 rows in 35 variables --
 * through tridiagonalisation + QL on the full column-side Gram matrix), <= 64 rows per level, inactive tasks as zero rows and non-diagonal weights through level_W
 * (round 5), global rows and the box only (the reference refuses task-local constraints, nHQP.cpp:41-44). */
typedef struct {
    int free_vars[OSOT_MAX_LEVELS];      /* free variables of each level.  The reference fixes them in its constructor from the
                                            singular values of A N at construction (>= 1e-6 counts as rank, nHQP.cpp:88-103)
                                            and never changes them; 0 = default: n, then previous minus the rows of the level
                                            above (full row rank) */
    double min_sv_ratio;                 /* setMinSingularValueRatio (nHQP.cpp:342-355 accepts 0 <= s <= 1; 0 lifts nothing).
                                            A positive value is honoured as it is; 0 means DEFAULT_MIN_SV_RATIO = 0.05
                                            (nHQP.h:66), so that a zeroed struct is "the reference's defaults", UNLESS
                                            min_sv_ratio_is_set != 0, which makes 0 itself the value */
    int no_ab_regularization;            /* setPerformAbRegularization(false) */
    int no_selective_ns_regularization;  /* setPerformSelectiveNullSpaceRegularization(false) */
    int min_sv_ratio_is_set;             /* != 0: min_sv_ratio is the caller's value, 0 included (only needed to say 0) */
    /* per level (round 5; nHQP::setPerformAbRegularization(level, .), setPerformSelectiveNullSpaceRegularization(level, .),
       setMinSingularValueRatio(std::vector<double>): nHQP.cpp:127-152, 206-221).  A level's switch is OFF if the solver-wide one
       or its own says so; a level's ratio, where level_min_sv_ratio_is_set[k] != 0, replaces the solver-wide one. */
    int level_no_ab_regularization[OSOT_MAX_LEVELS];
    int level_no_selective_ns_regularization[OSOT_MAX_LEVELS];
    int level_min_sv_ratio_is_set[OSOT_MAX_LEVELS];
    double level_min_sv_ratio[OSOT_MAX_LEVELS];
    /* a level whose task(s) carry a NON-DIAGONAL weight (osot_task_desc.dense_weight; Task::setWeight(W)): the level's full weight
       matrix [B][m_k][m_k] (device; Task::getWeight() of the level: block diagonal over its tasks, symmetric), which nHQP.cpp:381-382
       multiplies the REGULARISED A N and b0 by -- so W A / W b (osot_qp_batch.WA / Wb, what iHQP and eHQP take) do not suffice.  NULL
       for a level with diagonal weights; a dense_weight level without it is refused. */
    const double* level_W[OSOT_MAX_LEVELS];
} osot_nhqp_options;
int osot_nhqp_solve(osot_solver* s, const osot_qp_batch* batch, const osot_nhqp_options* options, void* hip_stream);
/* instances the device works on at once in the null-space front-end's level preparation (the kernels a solve's time goes to; the
 * level with the fewest resident wavefronts counts): what a sub-batch size is chosen against, like osot_solver_resident_waves for
 * the cascade (opensot_amd/parallel.py: suggest_lanes).  opt: as for osot_nhqp_solve (its free_vars decide which kernel a level takes). */
int osot_solver_resident_waves_nhqp(osot_solver* s, const osot_nhqp_options* opt, int* waves);

/* ---- the equality-only front-end: OpenSoT::solvers::eHQP (src/solvers/eHQP.cpp:64-95, 124-146) -----------
 * Same stack, same assembled arrays (A_k, b_k, w_k or WA_k / Wb_k) as osot_ihqp_solve; the constraints and bounds of the
 * stack are NOT used (the reference's eHQP does not use them either: eHQP.cpp:45-47) and there is no active set: per level
 * JP = L'A P (W = L L'), x += JP^+ (L'b - L'A x) with the reference's damped pseudo-inverse, P <- P - V V' (thin right
 * singular vectors).  One launch for all levels.  sigma_min <= 0 = the reference's default 1e-12 (eHQP.cpp:56).
 * n <= 32; no regularisation task (the reference's eHQP has none); status is OSOT_STATUS_SOLVED for every instance
 * (eHQP::solve returns true). */
int osot_ehqp_solve(osot_solver* s, const osot_qp_batch* batch, double sigma_min, void* hip_stream);

/* ---- batch-of-one BackEnd surface (host pointers; mirrors BackEnd.h) ------------------------ */
typedef struct osot_backend osot_backend;
/* create_instance(number_of_variables, number_of_constraints, hessian_type, eps_regularisation)
 * (QPOasesBackEnd.cpp:14-24). eps_regularisation is the FACTOR; absolute eps = 2.221e-13 * factor.
 * number_of_variables <= OSOT_MAX_QP_VARS (128; see osot_qp_solve_batch for the path beyond 64). */
int osot_backend_create(int number_of_variables, int number_of_constraints, int hessian_type,
                        double eps_regularisation, osot_backend** out);
int osot_backend_destroy(osot_backend* be);
/* matrices row-major; A is nc x nv; l/u may be NULL (no box) */
int osot_backend_init_problem(osot_backend* be, const double* H, const double* g, const double* A,
                              const double* lA, const double* uA, const double* l, const double* u);
int osot_backend_update_task(osot_backend* be, const double* H, const double* g);
int osot_backend_update_constraints(osot_backend* be, const double* A, const double* lA,
                                    const double* uA, int number_of_constraints);
int osot_backend_update_bounds(osot_backend* be, const double* l, const double* u);
int osot_backend_solve(osot_backend* be);
int osot_backend_get_solution(osot_backend* be, double* x);
int osot_backend_get_objective(osot_backend* be, double* f);
/* BackEnd::getOptions / setOptions (include/OpenSoT/solvers/BackEnd.h:139-145; the qpOASES back-end hands out its
 * qpOASES::Options through boost::any, QPOasesBackEnd.cpp:309-318).  What this back-end has to set is the iteration cap of
 * the active-set loop (the counterpart of nWSR, QPOasesBackEnd.cpp:30: 0 = default 20 (n + nc) + 100); get also reports
 * the iteration count and OSOT_STATUS_* of the last solve.  INTEGRATION.md: the adapter carries this struct in the
 * boost::any. */
typedef struct { int max_iterations; int last_iterations; int last_status; } osot_backend_options;
int osot_backend_get_options(osot_backend* be, osot_backend_options* opt);
int osot_backend_set_options(osot_backend* be, const osot_backend_options* opt);
int osot_backend_set_eps_regularisation(osot_backend* be, double eps_abs);
int osot_backend_get_eps_regularisation(osot_backend* be, double* eps_abs);
int osot_backend_get_num_variables(osot_backend* be, int* nv);
int osot_backend_get_num_constraints(osot_backend* be, int* nc);

/* batched generic QP: B independent problems of the SAME shape in BackEnd convention, device
 * pointers, one wavefront each.  H: [B][n][n], g: [B][n], A: [B][nc][n], ...; x out [B][n].
 * n <= OSOT_MAX_QP_VARS (128).  Problems of 65 .. 128 variables -- wider than a wavefront: include/OpenSoT/Task.h:47-565 has no limit, a
 * 45-DoF robot or a floating-base inverse-dynamics stack with five contacts is such a problem -- run one 256-thread WORKGROUP per QP
 * (opensot_amd/csrc/osot_qp_big.h: the same dual active-set method; J in a stream-ordered device workspace of 2 n^2 doubles per
 * resident QP, R and the vectors in LDS; nc <= 2048).  That path starts cold on every call (osot_backend_solve included: the plugin's
 * hot-start record is for n <= 64) and allocates with hipMallocAsync on the caller's stream: do not capture it into a HIP graph. */
int osot_qp_solve_batch(int B, int n, int nc, const double* H, const double* g, const double* A,
                        const double* lA, const double* uA, const double* l, const double* u,
                        double eps_abs, int max_iter, double* x, int* status, int* iterations,
                        void* hip_stream);

/* The same call through the SECOND back-end: an OSQP-convention ADMM solver -- the problem as
 * src/solvers/OSQPBackEnd.cpp poses it (P = H + eps I, one constraint matrix [A; I] with piled bounds, eps_abs = eps_rel =
 * 1e-5, sigma = 1e-6, alpha = 1.6, rho = 0.1 adaptive, 4000 iterations, "solved inaccurate" counts as solved: :25-49,
 * 198-226); osqp itself is not vendored in the reference, so this restates the published algorithm (parity vs osqp
 * unpinned) and is cross-checked against the active-set kernel.  eps_reg is the ABSOLUTE epsilon (the factory's factor
 * times 2.22e-13, OSQPBackEnd.cpp:8, 29).  max_iter 0 = 4000. */
int osot_qp_solve_batch_admm(int B, int n, int nc, const double* H, const double* g, const double* A,
                             const double* lA, const double* uA, const double* l, const double* u,
                             double eps_reg, int max_iter, double* x, int* status, int* iterations, void* hip_stream);

/* The same with osqp's settings exposed and WARM START: OSQPBackEnd keeps its osqp workspace between control cycles
 * (src/solvers/OSQPBackEnd.cpp:120-143 updates P, q, A and the bounds in place; :268-287), so every solve after the first
 * starts from the previous x and y with the rho the previous solve ended on (osqp's warm_start = 1).  Here the per-instance
 * state is three device arrays owned by the caller: warm_x [B][n], warm_y [B][nc + n] (the rows of A, then the box rows;
 * [B][nc] without l / u), warm_rho [B]; set warm_rho to 0 for "no state yet" (e.g. hipMemset the arrays once) -- an instance
 * whose solve fails is reset to that.  All three null = cold start.  opt null = the defaults below. */
typedef struct osot_admm_options {
    double eps_abs, eps_rel;     /* 0 = 1e-5 each (OSQPBackEnd.cpp:38-39) */
    double rho, sigma, alpha;    /* 0 = osqp's 0.1, 1e-6, 1.6 */
    int max_iter;                /* 0 = 4000 */
    int scaling;                 /* Ruiz equilibration passes: 0 = osqp's 10, negative = none */
    int check_every;             /* residual test period, 0 = 25 */
} osot_admm_options;
int osot_qp_solve_batch_admm_warm(int B, int n, int nc, const double* H, const double* g, const double* A,
                                  const double* lA, const double* uA, const double* l, const double* u,
                                  double eps_reg, const osot_admm_options* opt,
                                  double* warm_x, double* warm_y, double* warm_rho,
                                  double* x, int* status, int* iterations, void* hip_stream);

/* ---- batched kinematics producer ------------------------------------------------------
 * What the leaf tasks ask XBot::ModelInterface for every cycle: frame poses and 6 x n frame Jacobians
 * (velocity::Cartesian::_update, src/tasks/velocity/Cartesian.cpp:73-81: getJacobian / getPose), the centre
 * of mass and its 3 x n Jacobian (velocity::CoM::_update, src/tasks/velocity/CoM.cpp:59-74: getCOM /
 * getCOMJacobian).  xbot2_interface is not vendored in the reference, so this is an own implementation for a
 * kinematic TREE of revolute / prismatic joints (a floating base is the usual chain of three prismatic and three
 * revolute virtual joints); parity at this boundary is pinned by the CPU restatement in oracle/pykin.py and by
 * finite differences, not by the reference.  Convention: Jacobians are expressed in the world
 * frame (or the frame's base link frame: frame_base below), rows [linear; angular] of the frame origin, columns = joints; poses are [R row-major | p] (the Cartesian
 * leaf layout, osot_leaf_ptrs).  The Jacobians are written STRAIGHT into their row range of the stacked A_k. */
#define OSOT_KIN_MAX_JOINTS 64
#define OSOT_KIN_MAX_FRAMES 8
#define OSOT_KIN_MAX_PAIRS 32
enum { OSOT_JOINT_REVOLUTE = 0, OSOT_JOINT_PRISMATIC = 1 };
typedef struct {
    int n;                                   /* joints = generalised coordinates; tree order: parent[j] < j     */
    int parent[OSOT_KIN_MAX_JOINTS];         /* -1 = world                                                      */
    int type[OSOT_KIN_MAX_JOINTS];
    double axis[OSOT_KIN_MAX_JOINTS][3];     /* joint axis in the joint frame (unit)                            */
    double R0[OSOT_KIN_MAX_JOINTS][9];       /* fixed transform parent joint frame -> joint frame at q = 0      */
    double p0[OSOT_KIN_MAX_JOINTS][3];
    double mass[OSOT_KIN_MAX_JOINTS];        /* of the link the joint moves                                     */
    double com[OSOT_KIN_MAX_JOINTS][3];      /* its centre of mass in the joint frame                           */
    int n_frames;
    int frame_joint[OSOT_KIN_MAX_FRAMES];    /* joint whose link carries frame f                                */
    double frame_R[OSOT_KIN_MAX_FRAMES][9];  /* frame f in that joint frame                                     */
    double frame_p[OSOT_KIN_MAX_FRAMES][3];
    /* self-collision pairs: what velocity::CollisionAvoidance::update asks the (un-vendored) xbot2
     * collision module for every cycle, computeDistance / getDistanceJacobian
     * (src/constraints/velocity/CollisionAvoidance.cpp:96-118).  Each pair is two CAPSULES (segment + radius; a
     * segment of zero length is a sphere), each rigidly attached to a link.  Output per pair: the distance d between
     * the two surfaces and the row J_d with delta d = J_d dq, in the layout the OSOT_ROWS_COLLISION leaf expects. */
    int n_pairs;
    int pair_joint[OSOT_KIN_MAX_PAIRS][2];       /* joints whose links carry the two shapes                      */
    double pair_seg[OSOT_KIN_MAX_PAIRS][2][6];   /* capsule axis end points (a0, a1) in that joint's frame       */
    double pair_radius[OSOT_KIN_MAX_PAIRS][2];
    /* per-frame options of the Jacobian the producer writes (what Task::update applies after _update(), Task.h:375-400,
     * and Cartesian's own frame choice, Cartesian.cpp:73-100) */
    int frame_body[OSOT_KIN_MAX_FRAMES];         /* 1: BODY Jacobian Ad(R_f') J (Cartesian::setIsBodyJacobian,
                                                    Cartesian.cpp:93-100; pair with osot_task_desc.body_frame)       */
    unsigned long long frame_col_mask[OSOT_KIN_MAX_FRAMES]; /* Task::setActiveJointsMask (Task.h:129-139): bit j CLEAR =
                                                    column j of the frame's Jacobian is written as zero; 0 = no mask  */
    unsigned long long com_col_mask;             /* the same for the CoM Jacobian                                    */
    /* ---- environment shapes and boxes (reference: CollisionAvoidance::addCollisionShape /
     * moveCollisionShape / setLinksVsEnvironment, include/OpenSoT/constraints/velocity/CollisionAvoidance.h:115-144,
     * src/constraints/velocity/CollisionAvoidance.cpp:166-200; the shapes are XBot::Collision::Shape's sphere / capsule /
     * box).  Side a of a pair is a capsule (or sphere) on a link, as before.  Side b may be
     *   - carried by the WORLD: pair_joint[p][1] = -1 (a static world shape: its data are in world coordinates), and, with
     *     pair_env[p] = e + 1 > 0, placed by the RUNTIME pose osot_kin_batch.env_pose[e] (moveCollisionShape: the shape's
     *     data are then in the shape's own frame);
     *   - a BOX (pair_kind[p] = OSOT_SHAPE_BOX): half extents pair_box[p], box frame = carrier frame (link, world or
     *     env pose) composed with (pair_shape_R[p], pair_shape_p[p]) = link_T_shape; pair_seg[p][1] is unused and
     *     pair_radius[p][1] is a rounding radius (0 for a sharp box).
     * Distance = exact closest points of side a's axis segment and the box (piecewise-quadratic minimisation over the
     * segment parameter); a segment that touches or enters the box has no normal: distance -r_a - r_b, zero row (the
     * reference caps the bound at 0 there, CollisionAvoidance.cpp:141-146).  Box against box is not offered.
     * All zero (the value of a struct written before these fields existed) = capsule pairs on links only. */
    int pair_kind[OSOT_KIN_MAX_PAIRS];
    int pair_env[OSOT_KIN_MAX_PAIRS];
    double pair_box[OSOT_KIN_MAX_PAIRS][3];
    double pair_shape_R[OSOT_KIN_MAX_PAIRS][9];
    double pair_shape_p[OSOT_KIN_MAX_PAIRS][3];
    int n_env;                                   /* environment shapes with a runtime pose (env_pose entries)        */
    /* RELATIVE BASE LINK of a frame (round 6; velocity::Cartesian with base_link != "world", src/tasks/velocity/Cartesian.cpp:73-81:
     * getRelativeJacobian(distal, base, A) / getPose(distal, base, T); DefaultHumanoidStack's waist2LeftArm / waist2RightArm /
     * right2LeftLeg, tests/DefaultHumanoidStack.cpp:24-35, 52).  frame_base[f] = 0: the world (as before); g + 1 > 0: frame g
     * of this description is the base link frame (g != f; it may be a frame without outputs).  The producer then writes
     *   pose  = base_T_distal = [R_b'R_d | R_b'(p_d - p_b)],
     *   J     = R_b' (J_d - J_b shifted to the distal origin): column j = (anc_d(j) - anc_b(j)) [z_j x (p_d - p_j); z_j] rotated by
     *           R_b' (prismatic: [z_j; 0]) -- the twist of the distal frame relative to the base frame, in base coordinates; the
     *           joints both chains share drop out exactly, the joints of the base's own chain enter with a minus sign.
     * With frame_body[f] the rotation is R_d' instead (Ad(bR_d') after the R_b' of the relative Jacobian, Cartesian.cpp:93-100). */
    int frame_base[OSOT_KIN_MAX_FRAMES];
} osot_kin_desc;
enum { OSOT_SHAPE_CAPSULE = 0, OSOT_SHAPE_BOX = 1 };
#define OSOT_KIN_MAX_ENV 16
typedef struct {
    int B;
    const double* q;                               /* [B][n]                                                    */
    double* frame_pose[OSOT_KIN_MAX_FRAMES];       /* [B][12] or NULL                                           */
    double* frame_J[OSOT_KIN_MAX_FRAMES];          /* first of the 6 rows of frame f in instance 0, or NULL     */
    long long frame_J_stride[OSOT_KIN_MAX_FRAMES]; /* doubles from one instance to the next (ma_k * n)          */
    double* com;                                   /* [B][3] or NULL                                            */
    double* com_J;                                 /* first of the 3 rows in instance 0, or NULL                */
    long long com_J_stride;
    double* pair_dist;                             /* [B][n_pairs] surface distances (negative: penetration), or NULL */
    double* pair_J;                                /* first of the n_pairs rows J_d in instance 0, or NULL (the
                                                      OSOT_ROWS_COLLISION leaf p0: [B][rows][n])                */
    long long pair_J_stride;                       /* doubles from one instance to the next (rows * n)          */
    const double* env_pose;                        /* [n_env][12] world_T_shape = [R row-major | p] of the environment
                                                      shapes (moveCollisionShape), or NULL when n_env = 0           */
    long long env_pose_stride;                     /* 0: one world shared by all instances; 12 n_env: per instance  */
} osot_kin_batch;
typedef struct osot_kin osot_kin;
int osot_kin_create(const osot_kin_desc* desc, int device, osot_kin** out);
int osot_kin_destroy(osot_kin* k);
int osot_kinematics(osot_kin* k, const osot_kin_batch* batch, void* hip_stream);

/* The body of the reference's control loop for B robots in ONE launch (examples/cpp/coman_ik.cpp:186-219:
 * `model.update(); stack->update(); solver->solve(dq); q = model.sum(q, dq)`): per instance the same wavefront runs
 * osot_kinematics (q -> frame poses, Jacobian rows, CoM, written where `kin_batch` says: the arrays the leaf inputs and A_k
 * point into), then osot_cycle (AutoStack::update + the iHQP cascade), then, when q_integrate is not NULL,
 * q_integrate[i] += dq[i] (usually kin_batch->q itself: the next call starts from the integrated posture).
 * Results are those of the three calls; every array they write is still written.  Not offered with the collision-pair
 * stage, dense weights, inactive tasks or the hot start (OSOT_ERR_UNSUPPORTED -- use the three calls).  It pays where several
 * sub-batches share the chip (a sub-batch's kinematics launch otherwise waits for wavefront slots the other's cascade holds).
 * Stream-ordered. */
int osot_control_cycle(osot_solver* s, osot_kin* k, const osot_kin_batch* kin_batch, const osot_leaf_batch* leaf,
                       const osot_assembled_out* out, const osot_qp_batch* batch, double* q_integrate, void* hip_stream);
/* A ROLLOUT (round 5): `steps` control cycles of every robot in ONE launch -- the loop of the reference's example
 * (examples/cpp/coman_ik.cpp:174-219: for every iteration model.update(); stack->update(); solver->solve(dq); q = model.sum(q, dq))
 * run by the robot's own wavefront, cycle after cycle, with the leaf inputs (references, gains) held fixed over the rollout.  The
 * robots are independent, so the results are those of `steps` osot_control_cycle calls (tested bit-identical) -- without a launch
 * per step and without every robot waiting, at every step, for the slowest robot of the batch.  What MPC rollouts and sample-based
 * planners ask for: B candidate roll-outs advanced K control cycles each.
 * q_integrate (required for steps > 1) is advanced by every cycle's dq; dq / status of the batch hold the LAST cycle's dq and the
 * FIRST non-zero status of the rollout (a failed cycle leaves the robot where it is: dq = 0, coman_ik.cpp:189-190);
 * dq_steps [steps][B][n] and status_steps [steps][B] (either may be NULL) receive every cycle's.  Stream-ordered. */
int osot_control_rollout(osot_solver* s, osot_kin* k, const osot_kin_batch* kin_batch, const osot_leaf_batch* leaf,
                         const osot_assembled_out* out, const osot_qp_batch* batch, double* q_integrate, int steps,
                         double* dq_steps, int* status_steps, void* hip_stream);

/* ---- inverse-dynamics formulation (BASELINE config 5): x = [qddot (nv); contact forces / wrenches] -------------
 * (src/utils/InverseDynamics.cpp:12-28).  The matrices that are pure copies of model quantities are written by these
 * producers straight into their row ranges of the stacked A_k / C (zero-copy, like the kinematics producer's Jacobians);
 * the model quantities themselves (inertia matrix B, non-linear term h, contact Jacobians) come from the caller's
 * dynamics library, as they come from XBot::ModelInterface in the reference. */
#define OSOT_ID_MAX_FORCE_VARS 24
typedef struct {
    int B, nv, n_contacts, contact_dim;   /* contact_dim: 3 = point contact (force), 6 = surface contact (wrench)
                                             (InverseDynamics.cpp:16-27); nv + n_contacts * contact_dim <= OSOT_MAX_VARS */
    const double* Bm;                     /* [B][nv][nv] inertia matrix (symmetric)                                      */
    const double* h;                      /* [B][nv] non-linear term                                                     */
    const double* Jc;                     /* [B][n_contacts][contact_dim][nv] first contact_dim rows of each contact's
                                             Jacobian                                                                    */
    int floating_base;                    /* computedTorque checks the first six rows of tau (InverseDynamics.cpp:83-92)  */
} osot_id_model;
/* rows [B_u, -J_f'] of acceleration::DynamicFeasibility (DynamicFeasibility.cpp:22-46; 6 rows) and [B, -Jc'] of
 * TorqueLimits (TorqueLimits.cpp:25-46; nv rows): C_dyn / C_tau point at the block's first row in instance 0 (either may
 * be NULL), *_stride = doubles from one instance to the next (nc_stored * n).  Plus n_tasks task matrices [J_i 0]
 * (acceleration::Cartesian / CoM, Cartesian.cpp:152-160): J[i] is [B][J_rows[i]][nv], A_dst[i] the block's first row in
 * instance 0 of A_k, A_stride[i] = ma_k * n. */
int osot_id_rows(const osot_id_model* m, double* C_dyn, long long dyn_stride, double* C_tau, long long tau_stride,
                 int n_tasks, const double* const* J, const int* J_rows, double* const* A_dst, const long long* A_stride,
                 void* hip_stream);
/* GainType::Force of acceleration::Cartesian (src/tasks/acceleration/Cartesian.cpp:161-169, 517-524): per instance
 * Mi = J Bi J' (compute_cartesian_inertia_inverse: Bi = the model's inverse inertia matrix, [B][nv][nv]; J [B][rows][nv]),
 * then Gp = Mi Kp, Gd = Mi Kd written into the task's leaf array p0 behind its 2 rows errors (see
 * osot_task_desc.acc_gain_matrices: p0_gains points at instance 0's Gp, p0_stride = 2 rows + 2 rows^2 doubles), and
 * a_ref[B][rows] += Mi f for a virtual force f [B][rows] (NULL: none).  Kp, Kd: HOST pointers to rows x rows row-major
 * matrices (the task's settings).  rows <= 6. */
int osot_id_force_gains(int B, int nv, int rows, const double* J, const double* Bi, const double* Kp, const double* Kd,
                        const double* f_virtual, double* p0_gains, long long p0_stride, double* a_ref, void* hip_stream);
/* InverseDynamics::computedTorque (InverseDynamics.cpp:57-96): tau[B][nv] = B qddot + h - sum_c Jc' F_c from the solved
 * x[B][n]; ok[B] (may be NULL) = 0 where a floating-base row of tau exceeds fb_tol (the reference uses 10e-3 and
 * returns false). */
int osot_computed_torque(const osot_id_model* m, const double* x, double* tau, int* ok, double fb_tol, void* hip_stream);

/* ---- layout of the structs above as THIS library was compiled (for bindings that mirror them by hand: ctypes, cgo, JNI ...).
 * name = the typedef's name ("osot_plan_desc", "osot_qp_batch", ...).  *size = sizeof; the byte offset of every member, in
 * declaration order, goes to offsets[0 .. *n_fields) (at most max_fields are written; offsets may be NULL).  Unknown name:
 * OSOT_ERR_INVALID.  opensot_amd/abi.py is checked against it in tests/test_abi_host.py. */
int osot_abi_layout(const char* name, unsigned long long* size, unsigned long long* offsets, int max_fields, int* n_fields);

/* ---- multi-GPU: collect solved dq shards ---------------------------------------------------- */
typedef struct osot_comm osot_comm;
/* unique id exchange is the caller's job (e.g. torch.distributed broadcast of the 128-byte id) */
int osot_comm_unique_id(void* id128);
int osot_comm_create(const void* id128, int rank, int world, int device, osot_comm** out);
int osot_comm_destroy(osot_comm* c);
/* all ranks: recv[world*count] <- concat_r send_r[count] (fp64), on hip_stream */
int osot_allgather_dq(osot_comm* c, const double* send, double* recv, long long count, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif

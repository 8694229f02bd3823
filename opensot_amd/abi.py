"""ctypes mirror of include/osot_mi355x.h (structs, enums, prototypes).

Host-side plumbing only: the product's arithmetic lives in opensot_amd/csrc (HIP, gfx950).
"""
import ctypes as C
import os

MAX_LEVELS, MAX_TASKS, MAX_BOUNDS, MAX_ROWBLOCKS, MAX_VARS = 8, 8, 4, 8, 64
MAX_QP_VARS = 128      # the explicit-QP surface (osot_qp_solve_batch, osot_backend_*): OSOT_MAX_QP_VARS
MAX_BAND_ROWS, ID_MAX_FORCE_VARS = 6, 24

OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_HIP, ERR_NOT_SOLVED, ERR_COMM = range(6)
STATUS_SOLVED, STATUS_INFEASIBLE, STATUS_MAX_ITER, STATUS_NOT_PD = range(4)
TASK_GENERIC, TASK_CARTESIAN, TASK_COM, TASK_POSTURAL, TASK_ACC_CARTESIAN, TASK_ACC_COM, TASK_ACC_POSTURAL = range(7)
BOUND_GENERIC, BOUND_JOINT_LIMITS, BOUND_VELOCITY_LIMITS = range(3)
(ROWS_GENERIC, ROWS_COLLISION, ROWS_DYN_FEASIBILITY, ROWS_TORQUE_LIMITS, ROWS_FRICTION_CONE,
 ROWS_ACC_JOINT_LIMITS, ROWS_ACC_VELOCITY_LIMITS, ROWS_TASK_CARTESIAN, ROWS_TASK_COM, ROWS_UNIT_GENERIC) = range(10)
# OpenSoT::HessianType (include/OpenSoT/Task.h:33-41)
HST_UNDEFINED, HST_ZERO, HST_IDENTITY, HST_POSDEF, HST_POSDEF_NULLSPACE, HST_SEMIDEF, HST_UNKNOWN = range(7)

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


class TaskDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("rows", C.c_int), ("weight", C.c_double),
                ("lambda_", C.c_double), ("orientation_gain", C.c_double), ("lambda2", C.c_double),
                ("row_mask", C.c_ulonglong), ("parent_rows", C.c_int), ("sub_lambda", C.c_double),
                ("body_frame", C.c_int), ("dense_weight", C.c_int), ("acc_gain_matrices", C.c_int)]


class LevelDesc(C.Structure):
    _fields_ = [("n_tasks", C.c_int), ("task", TaskDesc * MAX_TASKS)]


class BoundDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("scaling", C.c_double), ("dT", C.c_double)]


class RowsDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("rows", C.c_int), ("d_threshold", C.c_double),
                ("detection_threshold", C.c_double), ("bound_scaling", C.c_double),
                ("first_col", C.c_int), ("dT", C.c_double), ("p", C.c_double), ("mu", C.c_double),
                ("task_lambda", C.c_double), ("task_orientation_gain", C.c_double),
                ("err_lb", C.c_double * MAX_BAND_ROWS), ("err_ub", C.c_double * MAX_BAND_ROWS),
                ("task_body_frame", C.c_int), ("n_candidates", C.c_int), ("only_level", C.c_int)]


class PlanDesc(C.Structure):
    _fields_ = [("n", C.c_int), ("n_levels", C.c_int), ("level", LevelDesc * MAX_LEVELS),
                ("n_bounds", C.c_int), ("bound", BoundDesc * MAX_BOUNDS),
                ("n_rowblocks", C.c_int), ("rowblock", RowsDesc * MAX_ROWBLOCKS),
                ("eps_abs", C.c_double), ("max_iter", C.c_int),
                ("has_regularisation", C.c_int), ("regularisation", TaskDesc), ("regularisation_dense", C.c_int)]


class QpBatch(C.Structure):
    _fields_ = [("B", C.c_int),
                ("A", C.c_void_p * MAX_LEVELS), ("b", C.c_void_p * MAX_LEVELS),
                ("w", C.c_void_p * MAX_LEVELS), ("c", C.c_void_p * MAX_LEVELS),
                ("C", C.c_void_p), ("lo", C.c_void_p), ("up", C.c_void_p),
                ("l", C.c_void_p), ("u", C.c_void_p),
                ("level_active", C.c_void_p),
                ("dq", C.c_void_p), ("x_levels", C.c_void_p),
                ("status", C.c_void_p), ("iterations", C.c_void_p), ("b_reg", C.c_void_p),
                ("WA", C.c_void_p * MAX_LEVELS), ("Wb", C.c_void_p * MAX_LEVELS),
                ("accepted_slack", C.c_void_p), ("A_reg", C.c_void_p)]


class LeafPtrs(C.Structure):
    _fields_ = [("p0", C.c_void_p), ("p1", C.c_void_p), ("p2", C.c_void_p), ("W", C.c_void_p)]


class LeafBatch(C.Structure):
    _fields_ = [("B", C.c_int),
                ("task", (LeafPtrs * MAX_TASKS) * MAX_LEVELS),
                ("bound", LeafPtrs * MAX_BOUNDS),
                ("rows", LeafPtrs * MAX_ROWBLOCKS), ("regularisation", LeafPtrs)]


class AssembledOut(C.Structure):
    _fields_ = [("b", C.c_void_p * MAX_LEVELS), ("w", C.c_void_p * MAX_LEVELS),
                ("C", C.c_void_p), ("lo", C.c_void_p), ("up", C.c_void_p),
                ("l", C.c_void_p), ("u", C.c_void_p), ("b_reg", C.c_void_p),
                ("WA", C.c_void_p * MAX_LEVELS), ("Wb", C.c_void_p * MAX_LEVELS), ("A", C.c_void_p * MAX_LEVELS)]


class BackendOptions(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("last_iterations", C.c_int), ("last_status", C.c_int)]


class NhqpOptions(C.Structure):
    _fields_ = [("free_vars", C.c_int * MAX_LEVELS), ("min_sv_ratio", C.c_double),
                ("no_ab_regularization", C.c_int), ("no_selective_ns_regularization", C.c_int),
                ("min_sv_ratio_is_set", C.c_int),
                ("level_no_ab_regularization", C.c_int * MAX_LEVELS), ("level_no_selective_ns_regularization", C.c_int * MAX_LEVELS),
                ("level_min_sv_ratio_is_set", C.c_int * MAX_LEVELS), ("level_min_sv_ratio", C.c_double * MAX_LEVELS),
                ("level_W", C.c_void_p * MAX_LEVELS)]


class AdmmOptions(C.Structure):
    _fields_ = [("eps_abs", C.c_double), ("eps_rel", C.c_double), ("rho", C.c_double), ("sigma", C.c_double),
                ("alpha", C.c_double), ("max_iter", C.c_int), ("scaling", C.c_int), ("check_every", C.c_int)]


class IdModel(C.Structure):
    _fields_ = [("B", C.c_int), ("nv", C.c_int), ("n_contacts", C.c_int), ("contact_dim", C.c_int),
                ("Bm", C.c_void_p), ("h", C.c_void_p), ("Jc", C.c_void_p), ("floating_base", C.c_int)]


# every symbol include/osot_mi355x.h declares (tests/test_abi_symbols.py checks the .so exports all)
KIN_MAX_JOINTS, KIN_MAX_FRAMES, KIN_MAX_PAIRS = 64, 8, 32
KIN_MAX_ENV = 16
SHAPE_CAPSULE, SHAPE_BOX = 0, 1
JOINT_REVOLUTE, JOINT_PRISMATIC = 0, 1


class KinDesc(C.Structure):
    _fields_ = [("n", C.c_int), ("parent", C.c_int * KIN_MAX_JOINTS), ("type", C.c_int * KIN_MAX_JOINTS),
                ("axis", (C.c_double * 3) * KIN_MAX_JOINTS), ("R0", (C.c_double * 9) * KIN_MAX_JOINTS),
                ("p0", (C.c_double * 3) * KIN_MAX_JOINTS), ("mass", C.c_double * KIN_MAX_JOINTS),
                ("com", (C.c_double * 3) * KIN_MAX_JOINTS), ("n_frames", C.c_int),
                ("frame_joint", C.c_int * KIN_MAX_FRAMES), ("frame_R", (C.c_double * 9) * KIN_MAX_FRAMES),
                ("frame_p", (C.c_double * 3) * KIN_MAX_FRAMES), ("n_pairs", C.c_int),
                ("pair_joint", (C.c_int * 2) * KIN_MAX_PAIRS), ("pair_seg", ((C.c_double * 6) * 2) * KIN_MAX_PAIRS),
                ("pair_radius", (C.c_double * 2) * KIN_MAX_PAIRS),
                ("frame_body", C.c_int * KIN_MAX_FRAMES), ("frame_col_mask", C.c_ulonglong * KIN_MAX_FRAMES),
                ("com_col_mask", C.c_ulonglong),
                ("pair_kind", C.c_int * KIN_MAX_PAIRS), ("pair_env", C.c_int * KIN_MAX_PAIRS),
                ("pair_box", (C.c_double * 3) * KIN_MAX_PAIRS), ("pair_shape_R", (C.c_double * 9) * KIN_MAX_PAIRS),
                ("pair_shape_p", (C.c_double * 3) * KIN_MAX_PAIRS), ("n_env", C.c_int),
                ("frame_base", C.c_int * KIN_MAX_FRAMES)]


class KinBatch(C.Structure):
    _fields_ = [("B", C.c_int), ("q", C.c_void_p), ("frame_pose", C.c_void_p * KIN_MAX_FRAMES),
                ("frame_J", C.c_void_p * KIN_MAX_FRAMES), ("frame_J_stride", C.c_longlong * KIN_MAX_FRAMES),
                ("com", C.c_void_p), ("com_J", C.c_void_p), ("com_J_stride", C.c_longlong),
                ("pair_dist", C.c_void_p), ("pair_J", C.c_void_p), ("pair_J_stride", C.c_longlong),
                ("env_pose", C.c_void_p), ("env_pose_stride", C.c_longlong)]


SYMBOLS = [
    "osot_version", "osot_last_error", "osot_device_count",
    "osot_plan_validate", "osot_plan_level_rows", "osot_plan_constraint_rows",
    "osot_plan_stored_constraint_rows",
    "osot_solver_create", "osot_solver_destroy", "osot_stack_update", "osot_ihqp_solve", "osot_cycle", "osot_nhqp_solve", "osot_ehqp_solve",
    "osot_solver_kernel_time_ms", "osot_solver_set_timing", "osot_solver_set_schedule", "osot_solver_set_hotstart", "osot_solver_set_specialisation", "osot_solver_set_task_active", "osot_solver_resident_waves", "osot_solver_resident_waves_nhqp",
    "osot_id_rows", "osot_id_force_gains", "osot_computed_torque", "osot_kin_create", "osot_kin_destroy", "osot_kinematics", "osot_control_cycle", "osot_control_rollout", "osot_solver_profile_phases",
    "osot_backend_create", "osot_backend_destroy", "osot_backend_init_problem",
    "osot_backend_update_task", "osot_backend_update_constraints", "osot_backend_update_bounds",
    "osot_backend_solve", "osot_backend_get_solution", "osot_backend_get_objective",
    "osot_backend_get_options", "osot_backend_set_options",
    "osot_backend_set_eps_regularisation", "osot_backend_get_eps_regularisation",
    "osot_backend_get_num_variables", "osot_backend_get_num_constraints",
    "osot_qp_solve_batch", "osot_qp_solve_batch_admm", "osot_qp_solve_batch_admm_warm",
    "osot_comm_unique_id", "osot_comm_create", "osot_comm_destroy", "osot_allgather_dq", "osot_abi_layout",
]

# the header's typedef name of every struct mirrored above (osot_abi_layout: tests/test_abi_host.py checks size and every offset)
STRUCTS = {"osot_task_desc": TaskDesc, "osot_level_desc": LevelDesc, "osot_bound_desc": BoundDesc, "osot_rows_desc": RowsDesc,
           "osot_plan_desc": PlanDesc, "osot_qp_batch": QpBatch, "osot_leaf_ptrs": LeafPtrs, "osot_leaf_batch": LeafBatch,
           "osot_assembled_out": AssembledOut, "osot_backend_options": BackendOptions, "osot_nhqp_options": NhqpOptions,
           "osot_admm_options": AdmmOptions, "osot_id_model": IdModel, "osot_kin_desc": KinDesc, "osot_kin_batch": KinBatch}

_HERE = os.path.dirname(os.path.abspath(__file__))
# (OSOT_MI355X_LIB: developer override, to A/B two builds of the HIP library on the GPU box)
LIB_PATH = os.environ.get("OSOT_MI355X_LIB") or os.path.join(_HERE, "csrc", "libosot_mi355x.so")
_lib = None


class NativeLibraryMissing(RuntimeError):
    pass


def lib():
    """Load the HIP library.  There is NO fallback: a missing .so is a hard error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). opensot_amd has no CPU fallback.")
    # torch bundles its own libamdhip64/librccl; import it FIRST so that our library binds to the same
    # (single) HIP runtime instance whose streams and allocations we are handed.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.osot_version.restype = C.c_char_p
    L.osot_last_error.restype = C.c_char_p
    L.osot_device_count.argtypes = [ip]
    L.osot_plan_validate.argtypes = [C.POINTER(PlanDesc)]
    L.osot_plan_level_rows.argtypes = [C.POINTER(PlanDesc), C.c_int, ip, ip]
    L.osot_plan_constraint_rows.argtypes = [C.POINTER(PlanDesc), ip]
    L.osot_plan_stored_constraint_rows.argtypes = [C.POINTER(PlanDesc), ip]
    L.osot_solver_create.argtypes = [C.POINTER(PlanDesc), C.c_int, C.c_int, C.POINTER(vp)]
    L.osot_solver_destroy.argtypes = [vp]
    L.osot_stack_update.argtypes = [vp, C.POINTER(LeafBatch), C.POINTER(AssembledOut), vp]
    L.osot_ihqp_solve.argtypes = [vp, C.POINTER(QpBatch), vp]
    L.osot_nhqp_solve.argtypes = [vp, C.POINTER(QpBatch), C.POINTER(NhqpOptions), vp]
    L.osot_ehqp_solve.argtypes = [vp, C.POINTER(QpBatch), C.c_double, vp]
    L.osot_cycle.argtypes = [vp, C.POINTER(LeafBatch), C.POINTER(AssembledOut), C.POINTER(QpBatch), vp]
    L.osot_solver_kernel_time_ms.argtypes = [vp, C.c_int, dp, ip]
    L.osot_solver_set_timing.argtypes = [vp, C.c_int]
    L.osot_solver_set_schedule.argtypes = [vp, C.c_int]
    L.osot_solver_set_hotstart.argtypes = [vp, C.c_int]
    L.osot_solver_set_specialisation.argtypes = [vp, C.c_int]
    L.osot_solver_set_task_active.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.osot_solver_resident_waves.argtypes = [vp, ip]
    L.osot_solver_resident_waves_nhqp.argtypes = [vp, vp, ip]
    L.osot_id_force_gains.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, dp, dp, vp, vp, C.c_longlong, vp, vp]
    L.osot_id_rows.argtypes = [C.POINTER(IdModel), vp, C.c_longlong, vp, C.c_longlong, C.c_int, vp, vp, vp, vp, vp]
    L.osot_computed_torque.argtypes = [C.POINTER(IdModel), vp, vp, vp, C.c_double, vp]
    L.osot_kin_create.argtypes = [C.POINTER(KinDesc), C.c_int, C.POINTER(vp)]
    L.osot_kin_destroy.argtypes = [vp]
    L.osot_kinematics.argtypes = [vp, C.POINTER(KinBatch), vp]
    L.osot_control_cycle.argtypes = [vp, vp, C.POINTER(KinBatch), C.POINTER(LeafBatch), C.POINTER(AssembledOut), C.POINTER(QpBatch), vp, vp]
    L.osot_control_rollout.argtypes = [vp, vp, C.POINTER(KinBatch), C.POINTER(LeafBatch), C.POINTER(AssembledOut), C.POINTER(QpBatch), vp, C.c_int, vp, vp, vp]
    L.osot_solver_profile_phases.argtypes = [vp, C.POINTER(QpBatch), vp, vp]
    L.osot_backend_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(vp)]
    L.osot_backend_destroy.argtypes = [vp]
    L.osot_backend_init_problem.argtypes = [vp, dp, dp, dp, dp, dp, dp, dp]
    L.osot_backend_update_task.argtypes = [vp, dp, dp]
    L.osot_backend_update_constraints.argtypes = [vp, dp, dp, dp, C.c_int]
    L.osot_backend_update_bounds.argtypes = [vp, dp, dp]
    L.osot_backend_solve.argtypes = [vp]
    L.osot_backend_get_solution.argtypes = [vp, dp]
    L.osot_backend_get_objective.argtypes = [vp, dp]
    L.osot_backend_get_options.argtypes = [vp, C.POINTER(BackendOptions)]
    L.osot_backend_set_options.argtypes = [vp, C.POINTER(BackendOptions)]
    L.osot_backend_set_eps_regularisation.argtypes = [vp, C.c_double]
    L.osot_backend_get_eps_regularisation.argtypes = [vp, dp]
    L.osot_backend_get_num_variables.argtypes = [vp, ip]
    L.osot_backend_get_num_constraints.argtypes = [vp, ip]
    L.osot_qp_solve_batch.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp,
                                      C.c_double, C.c_int, vp, vp, vp, vp]
    L.osot_qp_solve_batch_admm.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp,
                                           C.c_double, C.c_int, vp, vp, vp, vp]
    L.osot_qp_solve_batch_admm_warm.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp,
                                                C.c_double, C.POINTER(AdmmOptions), vp, vp, vp, vp, vp, vp, vp]
    L.osot_comm_unique_id.argtypes = [vp]
    L.osot_comm_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.osot_comm_destroy.argtypes = [vp]
    L.osot_allgather_dq.argtypes = [vp, vp, vp, C.c_longlong, vp]
    _lib = L
    return L


def check(rc, what=""):
    if rc != OK:
        msg = lib().osot_last_error()
        raise RuntimeError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")

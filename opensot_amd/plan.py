"""Static stack plan: the result of OpenSoT's stack algebra, described once for B instances.

Mirrors what `(com/(0.1*l_wrist + r_wrist)/postural) << jl << vl` builds in the reference
(src/utils/AutoStack.cpp:7-331, examples/cpp/coman_ik.cpp:425-449): levels of leaf task blocks,
box-bound producers and global constraint-row producers.  Pure description, no arithmetic.
"""
from dataclasses import dataclass, field
from typing import List, Optional

from . import abi

EPS_BASE = 1.0e3 * 2.221e-16  # qpOASES Options.cpp:147 x Constants.hpp:50 (QPOasesBackEnd.cpp:57)


def eps_abs_from_factor(factor: float) -> float:
    """BackEndFactory's eps_regularisation factor -> absolute epsilon (QPOasesBackEnd.cpp:57,67)."""
    return EPS_BASE * factor


@dataclass
class Task:
    kind: int
    rows: int
    weight: float = 1.0
    lam: float = 1.0
    orientation_gain: float = 1.0
    name: str = ""
    lam2: float = 1.0
    # SubTask (src/tasks/SubTask.cpp:22-112): keep only the parent's rows whose bit is set; rows = popcount(row_mask);
    # b = sub_lam * b_parent[rows]; parent_rows 0 = the kind's own size (6 / 3 / n)
    row_mask: int = 0
    parent_rows: int = 0
    sub_lam: float = 1.0
    # velocity::Cartesian with a body Jacobian (Cartesian::setIsBodyJacobian, Cartesian.cpp:93-100): b rotated by Ad(R')
    body_frame: bool = False
    # non-diagonal weight matrix W_i [rows][rows] (Task::setWeight(W), Task.h:273-300): comes with the leaf inputs
    dense_weight: bool = False
    # gain matrices of an acceleration task (acceleration::Cartesian::setGains, Cartesian.cpp:152-173): the leaf array p0
    # carries [pose_err; vel_err; Gp; Gd] per instance (osot_task_desc.acc_gain_matrices)
    acc_gain_matrices: bool = False

    @property
    def implicit(self):
        """A = [I 0], never stored: a whole Postural block with a scalar weight (a Postural sub-task, or one with a
        dense weight matrix, stores its unit rows)"""
        return self.kind in IMPLICIT_IDENTITY_TASKS and self.row_mask == 0 and not self.dense_weight

    def parent_size(self, n):
        if not self.row_mask:
            return self.rows
        if self.parent_rows:
            return self.parent_rows
        return {abi.TASK_CARTESIAN: 6, abi.TASK_ACC_CARTESIAN: 6, abi.TASK_COM: 3, abi.TASK_ACC_COM: 3,
                abi.TASK_POSTURAL: n, abi.TASK_ACC_POSTURAL: n}.get(self.kind, self.rows)

    def kept_rows(self):
        return [i for i in range(64) if (self.row_mask >> i) & 1]


def subtask(parent: "Task", indices, lam=1.0, n=None) -> "Task":
    """`parent % indices` (SubTask): same kind / weight / gains, only the listed rows of the parent"""
    mask = 0
    for i in indices:
        mask |= 1 << int(i)
    return Task(parent.kind, len(set(indices)), weight=parent.weight, lam=parent.lam, orientation_gain=parent.orientation_gain,
                name=parent.name + "%" + ",".join(str(i) for i in sorted(set(indices))), lam2=parent.lam2, row_mask=mask,
                parent_rows=parent.rows, sub_lam=lam)


IMPLICIT_IDENTITY_TASKS = (abi.TASK_POSTURAL, abi.TASK_ACC_POSTURAL)
UNIT_ROW_BLOCKS = (abi.ROWS_ACC_JOINT_LIMITS, abi.ROWS_ACC_VELOCITY_LIMITS, abi.ROWS_UNIT_GENERIC)


@dataclass
class Bound:
    kind: int
    scaling: float = 1.0
    dT: float = 0.0
    name: str = ""


@dataclass
class Rows:
    kind: int
    rows: int
    d_threshold: float = 0.0
    detection_threshold: float = 0.0
    bound_scaling: float = 1.0
    name: str = ""
    first_col: int = 0
    dT: float = 0.0
    p: float = 1.0
    mu: float = 0.0
    # None: global rows (every level).  k: task-local rows of level k (`task << constraint`, Task::getConstraints(),
    # iHQP.cpp:190, 282-287): they constrain level k's QP only
    level: Optional[int] = None
    # ROWS_TASK_*: constraints::TaskToConstraint (`stack << l_sole`): the underlying task's gains and the error band
    lam: float = 1.0
    orientation_gain: float = 1.0
    err_lb: object = 0.0      # scalar or one value per row (TaskToConstraint.cpp:34-52: err_lb / err_ub are vectors)
    err_ub: object = 0.0
    body_frame: bool = False  # ROWS_TASK_CARTESIAN: the task has a body Jacobian
    n_candidates: int = 0     # ROWS_COLLISION: pairs supplied per instance (0 = rows); the `rows` closest become rows

    def band(self):
        """(err_lb, err_ub) as per-row lists"""
        import numpy as np
        return (np.broadcast_to(np.asarray(self.err_lb, dtype=float), (self.rows,)).tolist(),
                np.broadcast_to(np.asarray(self.err_ub, dtype=float), (self.rows,)).tolist())


@dataclass
class StackPlan:
    n: int
    levels: List[List[Task]]
    bounds: List[Bound] = field(default_factory=list)
    rowblocks: List[Rows] = field(default_factory=list)
    eps_abs: float = eps_abs_from_factor(2e2)  # iHQP default eps_regularisation (iHQP.h:32)
    max_iter: int = 0
    # AutoStack::setRegularisationTask (AutoStack.h:78-92): an identity-Jacobian task (TASK_GENERIC with b supplied,
    # TASK_POSTURAL, TASK_ACC_POSTURAL; rows <= n) whose cost iHQP adds to every level (iHQP.cpp:274-278)
    regularisation: Optional[Task] = None
    regularisation_dense: bool = False   # the regularisation task has a stored Jacobian (osot_plan_desc.regularisation_dense)

    # ---- derived sizes -------------------------------------------------------------------
    @property
    def L(self):
        return len(self.levels)

    def m(self, k):
        return sum(t.rows for t in self.levels[k])

    def ma(self, k):
        """rows of level k stored explicitly (Postural's identity block is implicit)."""
        return sum(t.rows for t in self.levels[k] if not t.implicit)

    @property
    def nc(self):
        return sum(r.rows for r in self.rowblocks)

    @property
    def nc_stored(self):
        """rows of the constraint matrix C that are stored (unit-row blocks are implicit)."""
        return sum(r.rows for r in self.rowblocks if r.kind not in UNIT_ROW_BLOCKS)

    def rows_stored_offset(self, j):
        return sum(r.rows for r in self.rowblocks[:j] if r.kind not in UNIT_ROW_BLOCKS)

    def dense_level(self, k):
        """level k holds a block with a non-diagonal weight matrix: W_k A_k / W_k b_k are formed by the update"""
        return any(t.dense_weight for t in self.levels[k])

    def task_row_offset(self, k, j):
        return sum(t.rows for t in self.levels[k][:j])

    def rows_offset(self, j):
        return sum(r.rows for r in self.rowblocks[:j])

    def validate(self):
        assert 1 <= self.n <= abi.MAX_VARS
        assert 1 <= self.L <= abi.MAX_LEVELS
        for lev in self.levels:
            assert 1 <= len(lev) <= abi.MAX_TASKS
            for j, t in enumerate(lev):
                if t.row_mask:
                    assert bin(t.row_mask).count("1") == t.rows and (t.row_mask >> t.parent_size(self.n)) == 0
                    continue
                if t.implicit:
                    assert j == len(lev) - 1 and 1 <= t.rows <= self.n
                assert not t.body_frame or t.kind == abi.TASK_CARTESIAN
                if t.kind in (abi.TASK_CARTESIAN, abi.TASK_ACC_CARTESIAN):
                    assert t.rows == 6
                if t.kind in (abi.TASK_COM, abi.TASK_ACC_COM):
                    assert t.rows == 3
        assert len(self.bounds) <= abi.MAX_BOUNDS and len(self.rowblocks) <= abi.MAX_ROWBLOCKS
        if self.regularisation is not None:
            r = self.regularisation
            assert not r.row_mask
            if self.regularisation_dense:      # stored Jacobian A_r [B][rows][n] (BatchedStack.A_reg)
                assert r.kind in (abi.TASK_GENERIC, abi.TASK_CARTESIAN, abi.TASK_COM) and 1 <= r.rows <= 64
            else:
                assert r.kind in (abi.TASK_GENERIC, abi.TASK_POSTURAL, abi.TASK_ACC_POSTURAL)
                assert 1 <= r.rows <= self.n

    def to_c(self) -> abi.PlanDesc:
        self.validate()
        p = abi.PlanDesc()
        p.n = self.n
        p.n_levels = self.L
        for k, lev in enumerate(self.levels):
            p.level[k].n_tasks = len(lev)
            for j, t in enumerate(lev):
                d = p.level[k].task[j]
                d.kind, d.rows, d.weight, d.lambda_, d.orientation_gain, d.lambda2 = (
                    t.kind, t.rows, t.weight, t.lam, t.orientation_gain, t.lam2)
                d.row_mask, d.parent_rows, d.sub_lambda = t.row_mask, t.parent_rows, t.sub_lam
                d.body_frame, d.dense_weight = int(t.body_frame), int(t.dense_weight)
                d.acc_gain_matrices = int(t.acc_gain_matrices)
        p.n_bounds = len(self.bounds)
        for j, b in enumerate(self.bounds):
            p.bound[j].kind, p.bound[j].scaling, p.bound[j].dT = b.kind, b.scaling, b.dT
        p.n_rowblocks = len(self.rowblocks)
        for j, r in enumerate(self.rowblocks):
            d = p.rowblock[j]
            d.kind, d.rows, d.d_threshold, d.detection_threshold, d.bound_scaling = (
                r.kind, r.rows, r.d_threshold, r.detection_threshold, r.bound_scaling)
            d.first_col, d.dT, d.p, d.mu = r.first_col, r.dT, r.p, r.mu
            d.task_lambda, d.task_orientation_gain = r.lam, r.orientation_gain
            if r.kind in (abi.ROWS_TASK_CARTESIAN, abi.ROWS_TASK_COM):
                elb, eub = r.band()
                for i in range(r.rows):
                    d.err_lb[i], d.err_ub[i] = elb[i], eub[i]
            d.task_body_frame, d.n_candidates = int(r.body_frame), int(r.n_candidates)
            d.only_level = 0 if r.level is None else r.level + 1
        p.eps_abs = self.eps_abs
        p.max_iter = self.max_iter
        if self.regularisation is not None:
            t = self.regularisation
            p.has_regularisation = 1
            d = p.regularisation
            d.kind, d.rows, d.weight, d.lambda_, d.orientation_gain, d.lambda2 = (
                t.kind, t.rows, t.weight, t.lam, t.orientation_gain, t.lam2)
            d.row_mask, d.parent_rows, d.sub_lambda = 0, 0, 1.0
            p.regularisation_dense = 1 if self.regularisation_dense else 0
        return p

"""Host-side mirror of the reference's per-cycle interface for B stacks at once.

    stack.update()      -> BatchedStack.update(leaf)         (AutoStack::update, AutoStack.cpp:385-393)
    solver.solve(dq)    -> BatchedStack.solve()              (iHQP::solve, iHQP.cpp:263-358)
    BackEnd (one QP)    -> BackEnd class below               (BackEnd.h:23-171, same method names)

All arithmetic happens in libosot_mi355x.so (HIP, gfx950) through the C-ABI of include/osot_mi355x.h.
torch is used for device memory and streams only.  There is no CPU fallback: importing works without
a GPU (so that host logic can be tested), but every compute call needs the HIP library and a device.
"""
import ctypes as C

import numpy as np
import torch

from . import abi
from .plan import StackPlan


def _dev_ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream_ptr(device, stream=None):
    return C.c_void_p((stream if stream is not None else torch.cuda.current_stream(device)).cuda_stream)


def stored_rows(plan, C_dense):
    """[B][nc][n] dense constraint matrix -> [B][nc_stored][n] (unit-row blocks have no storage)."""
    from .plan import UNIT_ROW_BLOCKS
    keep, off = [], 0
    for r in plan.rowblocks:
        if r.kind not in UNIT_ROW_BLOCKS:
            keep.extend(range(off, off + r.rows))
        off += r.rows
    return np.ascontiguousarray(C_dense[:, keep, :])


class BatchedStack:
    """B independent instances of one static stack (same topology, different numbers)."""

    def __init__(self, plan: StackPlan, max_batch: int, device: int = 0, want_levels: bool = True):
        self.plan = plan
        self.max_batch = int(max_batch)
        self.device = torch.device("cuda", device)
        self._lib = abi.lib()   # raises NativeLibraryMissing when the HIP extension is not built
        self._cplan = plan.to_c()
        h = C.c_void_p()
        abi.check(self._lib.osot_solver_create(C.byref(self._cplan), self.max_batch, device, C.byref(h)),
                  "osot_solver_create")
        self._h = h
        n, L, B = plan.n, plan.L, self.max_batch
        f64 = dict(dtype=torch.float64, device=self.device)
        self.A = [torch.zeros((B, plan.ma(k), n), **f64) if plan.ma(k) else None for k in range(L)]
        self.b = [torch.zeros((B, plan.m(k)), **f64) for k in range(L)]
        self.w = [torch.ones((B, plan.m(k)), **f64) for k in range(L)]
        self.c = [None] * L
        # levels with a non-diagonal weight: W_k A_k and W_k b_k, written by update()
        self.WA = [torch.zeros((B, plan.ma(k), n), **f64) if plan.dense_level(k) and plan.ma(k) else None for k in range(L)]
        self.Wb = [torch.zeros((B, plan.m(k)), **f64) if plan.dense_level(k) else None for k in range(L)]
        nc = plan.nc
        self.C = torch.zeros((B, plan.nc_stored, n), **f64) if plan.nc_stored else None
        self.lo = torch.zeros((B, nc), **f64) if nc else None
        self.up = torch.zeros((B, nc), **f64) if nc else None
        self.l = torch.zeros((B, n), **f64) if plan.bounds else None
        self.u = torch.zeros((B, n), **f64) if plan.bounds else None
        self.b_reg = torch.zeros((B, plan.regularisation.rows), **f64) if plan.regularisation is not None else None
        # Jacobian of a regularisation task that has one (plan.regularisation_dense): written by the producer, like A[k]
        self.A_reg = (torch.zeros((B, plan.regularisation.rows, n), **f64)
                      if plan.regularisation is not None and plan.regularisation_dense else None)
        self.dq = torch.zeros((B, n), **f64)
        self.x_levels = torch.zeros((B, L, n), **f64) if want_levels else None
        self.status = torch.zeros((B,), dtype=torch.int32, device=self.device)
        self.iterations = torch.zeros((B,), dtype=torch.int32, device=self.device)
        self.accepted_slack = torch.zeros((B,), **f64)   # largest violation accepted as round-off, per instance
        self._leaf_keep = None
        self.stream = None         # torch.cuda.Stream every call of this stack is enqueued on (None: torch's current stream)
        self._cycle_args = {}
        self.level_active = None   # iHQP::setActiveStack (iHQP.cpp:391-395)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.osot_solver_destroy(h)
            self._h = None

    # ---- data movement helpers (plumbing) ------------------------------------------------------
    def load_leaf(self, leaf):
        """numpy leaf dict (opensot_amd.synth layout) -> device tensors; Jacobians go straight into
        the stacked A_k buffers (zero-copy aggregation: the producer owns those row ranges)."""
        B = leaf["B"]
        assert B <= self.max_batch
        to = lambda a: None if a is None else torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).to(self.device)
        for k in range(self.plan.L):
            if self.A[k] is not None:
                self.A[k][:B].copy_(to(leaf["A"][k]))
        for j, Cj in enumerate(leaf.get("C", [])):   # constraint rows the producer writes in place
            if Cj is not None:
                o = self.plan.rows_stored_offset(j)
                self.C[:B, o:o + Cj.shape[1]].copy_(to(Cj))
        dev = {"B": B,
               "task": [[tuple(to(x) for x in t) for t in lev] for lev in leaf["task"]],
               # dense weight matrices W_i [B][rows][rows] of the blocks that have one (same nesting as "task")
               "W": [[to(x) for x in lev] for lev in leaf["W"]] if leaf.get("W") is not None else None,
               "bound": [tuple(to(x) for x in t) for t in leaf["bound"]],
               "rows": [tuple(to(x) for x in t) for t in leaf["rows"]]}
        if self.plan.regularisation is not None:
            dev["reg"] = tuple(to(x) for x in leaf["reg"])
            if self.A_reg is not None:          # the regularisation task's Jacobian, written in place like the A_k
                self.A_reg[:B].copy_(to(leaf["reg_A"]))
        return dev

    def load_assembled(self, asm):
        """numpy assembled dict (oracle layout) -> the device buffers; used by tests that by-pass update()."""
        B = asm["B"]
        for k in range(self.plan.L):
            if self.A[k] is not None:
                self.A[k][:B].copy_(torch.as_tensor(asm["A"][k]))
            self.b[k][:B].copy_(torch.as_tensor(asm["b"][k]))
            if asm["w"][k] is not None:
                self.w[k][:B].copy_(torch.as_tensor(asm["w"][k]))
        if self.lo is not None:
            if self.C is not None:
                self.C[:B].copy_(torch.as_tensor(stored_rows(self.plan, asm["C"])))
            self.lo[:B].copy_(torch.as_tensor(asm["lo"]))
            self.up[:B].copy_(torch.as_tensor(asm["up"]))
        if self.l is not None:
            self.l[:B].copy_(torch.as_tensor(asm["l"])); self.u[:B].copy_(torch.as_tensor(asm["u"]))
        if self.b_reg is not None:
            self.b_reg[:B].copy_(torch.as_tensor(asm["reg"]["b"]))
        if self.A_reg is not None:
            self.A_reg[:B].copy_(torch.as_tensor(asm["reg"]["A"]))
        return B

    # ---- AutoStack::update ------------------------------------------------------------------------
    def _update_args(self, dev_leaf, write_weights=True):
        lb = abi.LeafBatch()
        lb.B = dev_leaf["B"]
        for k, lev in enumerate(dev_leaf["task"]):
            for j, (p0, p1, p2) in enumerate(lev):
                lp = lb.task[k][j]
                lp.p0, lp.p1, lp.p2 = _dev_ptr(p0), _dev_ptr(p1), _dev_ptr(p2)
                if dev_leaf.get("W") is not None:
                    lp.W = _dev_ptr(dev_leaf["W"][k][j])
        for j, (p0, p1, p2) in enumerate(dev_leaf["bound"]):
            lb.bound[j].p0, lb.bound[j].p1, lb.bound[j].p2 = _dev_ptr(p0), _dev_ptr(p1), _dev_ptr(p2)
        for j, (p0, p1, p2) in enumerate(dev_leaf["rows"]):
            lb.rows[j].p0, lb.rows[j].p1, lb.rows[j].p2 = _dev_ptr(p0), _dev_ptr(p1), _dev_ptr(p2)
        out = abi.AssembledOut()
        for k in range(self.plan.L):
            out.b[k] = _dev_ptr(self.b[k]); out.w[k] = _dev_ptr(self.w[k]) if write_weights else None
            out.WA[k], out.Wb[k] = _dev_ptr(self.WA[k]), _dev_ptr(self.Wb[k])
            out.A[k] = _dev_ptr(self.A[k]) if self.Wb[k] is not None else None
        out.C, out.lo, out.up = _dev_ptr(self.C), _dev_ptr(self.lo), _dev_ptr(self.up)
        out.l, out.u = _dev_ptr(self.l), _dev_ptr(self.u)
        if self.b_reg is not None:
            p0, p1, p2 = dev_leaf["reg"]
            lb.regularisation.p0, lb.regularisation.p1, lb.regularisation.p2 = _dev_ptr(p0), _dev_ptr(p1), _dev_ptr(p2)
            out.b_reg = _dev_ptr(self.b_reg)
        self._leaf_keep = dev_leaf
        return lb, out

    def update(self, dev_leaf, write_weights=True):
        """AutoStack::update for the B instances of dev_leaf (stream-ordered).  write_weights=False: the update leaves self.w
        alone (out.w[k] = NULL), for per-row DIAGONAL weight matrices (Task::setWeight(W) with a diagonal W,
        Aggregated.cpp:265-279) that the caller filled once"""
        lb, out = self._update_args(dev_leaf, write_weights)
        abi.check(self._lib.osot_stack_update(self._h, C.byref(lb), C.byref(out), _stream_ptr(self.device, self.stream)),
                  "osot_stack_update")
        return lb.B

    def cycle(self, dev_leaf, write_weights=True, cached=False):
        """one control cycle, `stack->update(); solver->solve(dq)` (coman_ik.cpp:186-192), in ONE launch: same results as
        update() followed by solve()"""
        # the three argument structs only hold pointers and sizes: they are rebuilt when a tensor they point at has been
        # replaced (the per-cycle Jacobian sets, the gather's send block ...), not on every call -- building them is ~30 us
        # of host time, a fifth of the launch they describe
        key = (id(dev_leaf), write_weights,
               tuple(0 if a is None else a.data_ptr() for a in self.A), self.dq.data_ptr(), self.status.data_ptr(),
               0 if self.x_levels is None else self.x_levels.data_ptr())
        # (cached=True is the caller's promise that dev_leaf's tensors and the stack's other buffers are not REPLACED
        #  between calls -- writing into them is fine; the per-cycle A sets and the output tensors are part of the key)
        cached = cached and self.level_active is None
        hit = self._cycle_args.get(key) if cached else None
        if hit is None:
            lb, out = self._update_args(dev_leaf, write_weights)
            qb = self._qp_batch(lb.B)
            hit = (lb, out, qb, dev_leaf)
            if cached:
                if len(self._cycle_args) > 64:
                    self._cycle_args.clear()
                self._cycle_args[key] = hit
        lb, out, qb, _ = hit
        abi.check(self._lib.osot_cycle(self._h, C.byref(lb), C.byref(out), C.byref(qb), _stream_ptr(self.device, self.stream)), "osot_cycle")
        return lb.B

    def control_cycle(self, kin, kin_batch, dev_leaf, q_integrate=None, write_weights=True, steps=None, dq_steps=None, status_steps=None):
        """the body of the reference's control loop in ONE launch (osot_control_cycle; coman_ik.cpp:186-219): per instance the
        kinematics producer `kin` (a kinematics.Kinematics; kin_batch = kin.batch_args(...), whose outputs are the tensors
        dev_leaf and self.A point at), AutoStack::update, the cascade, and q_integrate += dq when a tensor is given.  The three
        argument structs are cached per (kin_batch, dev_leaf): they hold pointers only."""
        key = ("control", id(kin_batch), id(dev_leaf), write_weights, 0 if q_integrate is None else q_integrate.data_ptr(),
               tuple(0 if a is None else a.data_ptr() for a in self.A), self.dq.data_ptr(), self.status.data_ptr())
        hit = self._cycle_args.get(key) if self.level_active is None else None
        if hit is None:
            lb, out = self._update_args(dev_leaf, write_weights)
            qb = self._qp_batch(lb.B)
            hit = (lb, out, qb, dev_leaf, kin_batch)
            if self.level_active is None:
                if len(self._cycle_args) > 64:
                    self._cycle_args.clear()
                self._cycle_args[key] = hit
        lb, out, qb, _, kb = hit
        if steps is None:
            abi.check(self._lib.osot_control_cycle(self._h, kin._h, C.byref(kb), C.byref(lb), C.byref(out), C.byref(qb),
                                                   _dev_ptr(q_integrate), _stream_ptr(self.device, self.stream)), "osot_control_cycle")
        else:
            # the kernel indexes dq_steps as (t * B + inst) * n + c and status_steps as t * B + inst, int32: checked here, not trusted
            n = self.plan.n
            for name, t, dt, shape in (("dq_steps", dq_steps, torch.float64, (int(steps), lb.B, n)), ("status_steps", status_steps, torch.int32, (int(steps), lb.B))):
                if t is not None and not (t.is_cuda and t.device == self.device and t.dtype == dt and t.is_contiguous() and tuple(t.shape) == shape):
                    raise ValueError(f"control_rollout: {name} must be a contiguous {dt} tensor of shape {shape} on {self.device}")
            abi.check(self._lib.osot_control_rollout(self._h, kin._h, C.byref(kb), C.byref(lb), C.byref(out), C.byref(qb),
                                                     _dev_ptr(q_integrate), int(steps), _dev_ptr(dq_steps), _dev_ptr(status_steps),
                                                     _stream_ptr(self.device, self.stream)), "osot_control_rollout")
        return lb.B

    def control_rollout(self, kin, kin_batch, dev_leaf, q_integrate, steps, dq_steps=None, status_steps=None, write_weights=True):
        """`steps` control cycles of every robot in ONE launch (osot_control_rollout): the loop of the reference's example
        (coman_ik.cpp:174-219) by the robot's own wavefront, q_integrate advanced by every cycle's dq.  dq_steps [steps][B][n] /
        status_steps [steps][B] (int32), when given, receive every cycle's dq / status; self.dq holds the last cycle's, self.status
        the first non-zero status.  Same results as `steps` control_cycle calls."""
        return self.control_cycle(kin, kin_batch, dev_leaf, q_integrate, write_weights, steps=steps, dq_steps=dq_steps, status_steps=status_steps)

    # ---- Solver::solve ------------------------------------------------------------------------------
    def _qp_batch(self, B):
        qb = abi.QpBatch()
        qb.B = B
        for k in range(self.plan.L):
            qb.A[k] = _dev_ptr(self.A[k]); qb.b[k] = _dev_ptr(self.b[k])
            qb.w[k] = _dev_ptr(self.w[k]); qb.c[k] = _dev_ptr(self.c[k])
        qb.C, qb.lo, qb.up = _dev_ptr(self.C), _dev_ptr(self.lo), _dev_ptr(self.up)
        qb.l, qb.u = _dev_ptr(self.l), _dev_ptr(self.u)
        qb.dq, qb.x_levels = _dev_ptr(self.dq), _dev_ptr(self.x_levels)
        qb.status, qb.iterations = _dev_ptr(self.status), _dev_ptr(self.iterations)
        qb.b_reg = _dev_ptr(self.b_reg)
        qb.A_reg = _dev_ptr(self.A_reg)
        for k in range(self.plan.L):
            qb.WA[k], qb.Wb[k] = _dev_ptr(self.WA[k]), _dev_ptr(self.Wb[k])
        qb.accepted_slack = _dev_ptr(self.accepted_slack)
        if self.level_active is not None:
            self._act = (C.c_ubyte * self.plan.L)(*[1 if a else 0 for a in self.level_active])
            qb.level_active = C.cast(self._act, C.c_void_p)
        return qb

    def solve(self, B):
        """stream-ordered; results in self.dq[:B], self.status[:B] (device)."""
        qb = self._qp_batch(B)
        abi.check(self._lib.osot_ihqp_solve(self._h, C.byref(qb), _stream_ptr(self.device, self.stream)), "osot_ihqp_solve")

    def solve_nhqp(self, B, free_vars=None, min_sv_ratio=None, ab_regularization=True, selective_ns_regularization=True, level_W=None):
        """Solver::solve() with the reference's NULL-SPACE front-end, OpenSoT::solvers::nHQP (nHQP.cpp:155-204), on the same
        assembled arrays; stream-ordered, results in self.dq[:B] / self.status[:B].  free_vars: free variables per level
        (the reference fixes them at construction); None = n, then minus the rows of the level above"""
        opt = abi.NhqpOptions()
        L = self.plan.L
        # per-level lists have ONE entry per level (nHQP::setMinSingularValueRatio(std::vector<double>) throws on a size that
        # differs from the number of layers, nHQP.cpp:127-152)
        for name, val in (("free_vars", free_vars), ("min_sv_ratio", min_sv_ratio), ("ab_regularization", ab_regularization),
                          ("selective_ns_regularization", selective_ns_regularization), ("level_W", level_W)):
            if isinstance(val, (list, tuple)) and len(val) != L:
                raise ValueError(f"solve_nhqp: {name} has {len(val)} entries, the stack has {L} levels")
        if free_vars is not None:
            for k, v in enumerate(free_vars):
                opt.free_vars[k] = int(v)
        # scalars: the solver-wide setters; lists (one entry per level): nHQP::setPerformAbRegularization(level, .),
        # setPerformSelectiveNullSpaceRegularization(level, .), setMinSingularValueRatio(std::vector<double>) (nHQP.cpp:127-152, 206-221)
        if isinstance(min_sv_ratio, (list, tuple)):
            for k, v in enumerate(min_sv_ratio):
                if v is not None:
                    opt.level_min_sv_ratio[k] = float(v); opt.level_min_sv_ratio_is_set[k] = 1
        elif min_sv_ratio is not None:       # None: the reference's default 0.05; 0.0 is a value (no singular value is lifted)
            opt.min_sv_ratio = float(min_sv_ratio)
            opt.min_sv_ratio_is_set = 1
        if isinstance(ab_regularization, (list, tuple)):
            for k, v in enumerate(ab_regularization):
                opt.level_no_ab_regularization[k] = 0 if v else 1
        else:
            opt.no_ab_regularization = 0 if ab_regularization else 1
        if isinstance(selective_ns_regularization, (list, tuple)):
            for k, v in enumerate(selective_ns_regularization):
                opt.level_no_selective_ns_regularization[k] = 0 if v else 1
        else:
            opt.no_selective_ns_regularization = 0 if selective_ns_regularization else 1
        # level_W: per level the FULL weight matrix [B][m_k][m_k] (device tensor) of a level with a non-diagonal weight, or None
        if level_W is not None:
            self._nhqp_W = list(level_W)            # (kept alive until the next call)
            for k, Wk in enumerate(level_W):
                if Wk is not None:
                    mk = self.plan.m(k)
                    # (the kernel indexes Wd as inst * m * m + r * m + s: anything else is silent garbage or an out-of-bounds read)
                    if not (Wk.is_cuda and Wk.device == self.device and Wk.dtype == torch.float64 and Wk.is_contiguous()
                            and Wk.dim() == 3 and Wk.shape[0] >= B and tuple(Wk.shape[1:]) == (mk, mk)):
                        raise ValueError(f"solve_nhqp: level_W[{k}] must be a contiguous float64 tensor [>= {B}][{mk}][{mk}] on {self.device}")
                    opt.level_W[k] = Wk.data_ptr()
        qb = self._qp_batch(B)
        abi.check(self._lib.osot_nhqp_solve(self._h, C.byref(qb), C.byref(opt), _stream_ptr(self.device, self.stream)), "osot_nhqp_solve")

    def solve_ehqp(self, B, sigma_min=0.0):
        """Solver::solve() with the reference's EQUALITY-ONLY front-end, OpenSoT::solvers::eHQP (eHQP.cpp:64-95): damped
        pseudo-inverses and projectors on the same assembled arrays; the stack's constraints and bounds are not used (as in
        the reference).  sigma_min 0 = the reference's 1e-12.  Stream-ordered, results in self.dq[:B] / self.status[:B]"""
        qb = self._qp_batch(B)
        abi.check(self._lib.osot_ehqp_solve(self._h, C.byref(qb), float(sigma_min), _stream_ptr(self.device, self.stream)), "osot_ehqp_solve")

    PHASES = ("hbuild", "chol", "inverse", "subst", "equalities", "inequalities", "opt_rhs", "total",
              "eq:J'a", "eq:reductions", "eq:z", "eq:householder",
              "in:scan", "in:d=J'n", "in:z", "in:r,steps", "in:householder", "in:drop")

    def profile_phases(self, B):
        """diagnostic: per-instance shader-clock cycles per phase, [B][OSOT_N_PHASES] (see PHASES)."""
        cyc = torch.zeros((B, len(self.PHASES)), dtype=torch.int64, device=self.device)
        qb = self._qp_batch(B)
        abi.check(self._lib.osot_solver_profile_phases(self._h, C.byref(qb), _dev_ptr(cyc), _stream_ptr(self.device, self.stream)),
                  "osot_solver_profile_phases")
        torch.cuda.synchronize(self.device)
        return cyc.cpu().numpy()

    def set_task_active(self, level, task, active):
        """Task::setActive (Task.h:232-239): an inactive task's rows count as zero rows from the next solve() on"""
        abi.check(self._lib.osot_solver_set_task_active(self._h, level, task, 1 if active else 0), "osot_solver_set_task_active")

    def resident_waves(self):
        """instances the device holds at once for this plan (one wavefront each)"""
        v = C.c_int(0)
        abi.check(self._lib.osot_solver_resident_waves(self._h, C.byref(v)), "osot_solver_resident_waves")
        return v.value

    def resident_waves_nhqp(self):
        """the same for the null-space front-end (its level-preparation kernels; default options)"""
        v = C.c_int(0)
        abi.check(self._lib.osot_solver_resident_waves_nhqp(self._h, None, C.byref(v)), "osot_solver_resident_waves_nhqp")
        return v.value

    def set_schedule(self, longest_first=True):
        """dispatch order inside solve(): longest-first from the previous solve's iteration counts (default)
        or plain instance order; results are identical either way."""
        abi.check(self._lib.osot_solver_set_schedule(self._h, 1 if longest_first else 0), "osot_solver_set_schedule")

    def set_hotstart(self, on=True):
        """hot start of every level's working set from the instance's previous solve (qpOASES' hotstart,
        QPOasesBackEnd.cpp:258-285); switching it on (again) forgets the recorded sets.  Default off."""
        abi.check(self._lib.osot_solver_set_hotstart(self._h, 1 if on else 0), "osot_solver_set_hotstart")

    def set_specialisation(self, on=True):
        """kernel instantiation by plan structure (default on): plans without constraint rows run the cascade instantiation
        that carries no constraint-row code; results are bit-identical either way (osot_solver_set_specialisation)."""
        abi.check(self._lib.osot_solver_set_specialisation(self._h, 1 if on else 0), "osot_solver_set_specialisation")

    def set_timing(self, on, stride=1):
        """HIP-event timing of the cascade / cycle launches (kernel_time_ms); stride k: every k-th launch only"""
        abi.check(self._lib.osot_solver_set_timing(self._h, (max(1, int(stride)) if on else 0)), "osot_solver_set_timing")

    def kernel_time_ms(self, reset=True):
        avg = C.c_double(0.0); cnt = C.c_int(0)
        abi.check(self._lib.osot_solver_kernel_time_ms(self._h, 1 if reset else 0, C.byref(avg), C.byref(cnt)),
                  "osot_solver_kernel_time_ms")
        return avg.value, cnt.value


class BackEnd:
    """OpenSoT::solvers::BackEnd for one QP (include/OpenSoT/solvers/BackEnd.h), same method names and
    bool returns; matrices are numpy row-major.  `eps_regularisation` is the factory's FACTOR
    (BackEndFactory.cpp:4-17)."""

    def __init__(self, number_of_variables, number_of_constraints, hessian_type=abi.HST_SEMIDEF,
                 eps_regularisation=2e2):
        self._lib = abi.lib()
        h = C.c_void_p()
        abi.check(self._lib.osot_backend_create(number_of_variables, number_of_constraints, hessian_type,
                                                eps_regularisation, C.byref(h)), "osot_backend_create")
        self._h = h
        self.nv = number_of_variables

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.osot_backend_destroy(h)
            self._h = None

    @staticmethod
    def _p(a):
        if a is None:
            return None, None
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.size == 0:
            return a, None
        return a, a.ctypes.data_as(abi.dp)

    def initProblem(self, H, g, A, lA, uA, l, u):
        ks = [self._p(a) for a in (H, g, A, lA, uA, l, u)]
        nc = C.c_int(0)
        self._lib.osot_backend_get_num_constraints(self._h, C.byref(nc))
        rows = 0 if ks[2][0] is None else (ks[2][0].shape[0] if ks[2][0].ndim == 2 else 0)
        if rows != nc.value:
            return False   # "A.rows() != _A.rows()" (QPOasesBackEnd.cpp:91-94)
        return self._lib.osot_backend_init_problem(self._h, *[k[1] for k in ks]) == abi.OK

    def updateTask(self, H, g):
        (H_, hp), (g_, gp) = self._p(H), self._p(g)
        if H_ is None or g_ is None or H_.shape != (self.nv, self.nv) or g_.shape[0] != self.nv:
            return False   # BackEnd.cpp:23-41
        return self._lib.osot_backend_update_task(self._h, hp, gp) == abi.OK

    def updateConstraints(self, A, lA, uA):
        (A_, ap), (l_, lp), (u_, up) = self._p(A), self._p(lA), self._p(uA)
        rows = A_.shape[0] if A_ is not None and A_.ndim == 2 else 0
        if rows and A_.shape[1] != self.nv:
            return False
        if (l_ is not None and l_.shape[0] != rows) or (u_ is not None and u_.shape[0] != rows):
            return False   # BackEnd.cpp:47-60
        return self._lib.osot_backend_update_constraints(self._h, ap, lp, up, rows) == abi.OK

    def updateBounds(self, l, u):
        (l_, lp), (u_, up) = self._p(l), self._p(u)
        if l_ is not None and (l_.shape[0] != self.nv or u_.shape[0] != self.nv):
            return False
        return self._lib.osot_backend_update_bounds(self._h, lp, up) == abi.OK

    def solve(self):
        return self._lib.osot_backend_solve(self._h) == abi.OK

    def getSolution(self):
        x = np.zeros(self.nv)
        self._lib.osot_backend_get_solution(self._h, x.ctypes.data_as(abi.dp))
        return x

    def getObjective(self):
        f = C.c_double(0.0)
        self._lib.osot_backend_get_objective(self._h, C.byref(f))
        return f.value

    def getOptions(self):
        """BackEnd::getOptions (BackEnd.h:139): dict(max_iterations, last_iterations, last_status)"""
        o = abi.BackendOptions()
        self._lib.osot_backend_get_options(self._h, C.byref(o))
        return {"max_iterations": o.max_iterations, "last_iterations": o.last_iterations, "last_status": o.last_status}

    def setOptions(self, options):
        """BackEnd::setOptions (BackEnd.h:145): the active-set iteration cap (the counterpart of qpOASES' nWSR)"""
        o = abi.BackendOptions()
        o.max_iterations = int(options.get("max_iterations", 0))
        return self._lib.osot_backend_set_options(self._h, C.byref(o)) == abi.OK

    def getEpsRegularisation(self):
        e = C.c_double(0.0)
        self._lib.osot_backend_get_eps_regularisation(self._h, C.byref(e))
        return e.value

    def setEpsRegularisation(self, eps):
        return self._lib.osot_backend_set_eps_regularisation(self._h, eps) == abi.OK

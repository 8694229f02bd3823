"""torch.Tensor in / out entry points: what `pyopensot` gives a Python planner in the reference
(bindings/python/solver.hpp:26-31: `solve(solver) -> VectorXd`; :67-91: iHQP(stack, eps_regularisation, be_solver), solve,
getNumberOfTasks, setActiveStack, activateAllStacks, getBackEndName, setEpsRegularisation; :93-...: nHQP), for B instances
at once on CUDA/HIP tensors: the device pointers of the caller's tensors go STRAIGHT to the C-ABI (include/osot_mi355x.h),
results come back as tensors on the same device and stream -- no host copies, no staging.

    qp_solve(H, g, A, lA, uA, l, u, ...)   B generic QPs in BackEnd convention through either back-end
    iHQP / nHQP                            the cascade front-ends on a BatchedStack, `solve()` returning dq as a tensor
"""
import ctypes as C
import enum

import torch

from . import abi
from .plan import StackPlan, eps_abs_from_factor
from .solver import BatchedStack


class solver_back_ends(enum.Enum):
    """OpenSoT::solvers::solver_back_ends (bindings/python/solver.hpp:34-43): the two this build implements"""
    qpOASES = 0     # -> the wavefront dual active-set kernel (qpOASES conventions: eps = 1e3 * 2.221e-16 * factor)
    OSQP = 1        # -> the OSQP-convention ADMM kernel (eps = 2.22e-13 * factor)


def _chk(t, name, shape=None, dtype=torch.float64):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError(f"{name} must be a tensor on the GPU")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous (its device pointer is handed to the kernel as it is)")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
    return C.c_void_p(t.data_ptr())


def admm_state(B, n, nc, box=True, device=0):
    """the warm-start state of the OSQP-convention back-end for B instances (what OSQPBackEnd's osqp workspace carries from
    one control cycle to the next, OSQPBackEnd.cpp:120-143): x, y, rho; rho = 0 means "no state yet" """
    dev = torch.device("cuda", device) if isinstance(device, int) else device
    return {"x": torch.zeros((B, n), dtype=torch.float64, device=dev),
            "y": torch.zeros((B, nc + (n if box else 0)), dtype=torch.float64, device=dev),
            "rho": torch.zeros((B,), dtype=torch.float64, device=dev)}


def qp_solve(H, g, A=None, lA=None, uA=None, l=None, u=None, eps_regularisation=2e2, be_solver=solver_back_ends.qpOASES,
             max_iter=0, warm=None, scaling=0):
    """B QPs  min 1/2 x'Hx + g'x  s.t.  lA <= A x <= uA,  l <= x <= u  (BackEnd.h:125-150).
    H [B][n][n], g [B][n], A [B][nc][n], lA / uA [B][nc], l / u [B][n] (A.. and l, u optional), float64, on one GPU.
    eps_regularisation is the back-end factory's FACTOR (BackEndFactory.cpp:4-17).
    OSQP back-end only: warm = admm_state(...) carried from call to call (warm start), scaling = Ruiz passes (0 = osqp's 10,
    negative = none).
    Returns (x [B][n], status [B] int32 OSOT_STATUS_*, iterations [B] int32), stream-ordered on the current stream."""
    lib = abi.lib()
    if H.dim() != 3 or H.shape[1] != H.shape[2]:
        raise ValueError("H must be [B][n][n]")
    B, n = H.shape[0], H.shape[1]
    nc = 0 if A is None else A.shape[1]
    dev = H.device
    pH, pg = _chk(H, "H"), _chk(g, "g", (B, n))
    pA, plA, puA = _chk(A, "A", (B, nc, n)), _chk(lA, "lA", (B, nc)), _chk(uA, "uA", (B, nc))
    pl, pu = _chk(l, "l", (B, n)), _chk(u, "u", (B, n))
    x = torch.empty((B, n), dtype=torch.float64, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    iters = torch.empty((B,), dtype=torch.int32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    with torch.cuda.device(dev):
        if be_solver == solver_back_ends.OSQP:
            opt = abi.AdmmOptions()
            opt.max_iter = max_iter
            opt.scaling = scaling
            m = nc + (n if l is not None else 0)
            wx = wy = wr = None
            if warm is not None:
                wx, wy, wr = _chk(warm["x"], "warm x", (B, n)), _chk(warm["y"], "warm y", (B, m)), _chk(warm["rho"], "warm rho", (B,))
            abi.check(lib.osot_qp_solve_batch_admm_warm(B, n, nc, pH, pg, pA, plA, puA, pl, pu, 2.22e-13 * eps_regularisation,
                                                        C.byref(opt), wx, wy, wr,
                                                        C.c_void_p(x.data_ptr()), C.c_void_p(status.data_ptr()),
                                                        C.c_void_p(iters.data_ptr()), stream), "osot_qp_solve_batch_admm_warm")
        else:
            if warm is not None:
                raise ValueError("warm is the OSQP back-end's state; the active-set back-end's hot start lives in BatchedStack.set_hotstart")
            abi.check(lib.osot_qp_solve_batch(B, n, nc, pH, pg, pA, plA, puA, pl, pu, eps_abs_from_factor(eps_regularisation), max_iter,
                                              C.c_void_p(x.data_ptr()), C.c_void_p(status.data_ptr()),
                                              C.c_void_p(iters.data_ptr()), stream), "osot_qp_solve_batch")
    return x, status, iters


class iHQP:
    """pyopensot.iHQP for B instances of one stack (bindings/python/solver.hpp:67-91).  The stack's per-cycle inputs are the
    BatchedStack's device tensors (`self.stack.A[k]`, leaf dictionaries of tensors: the caller's producers write into them);
    `solve()` runs update + cascade in one launch and returns dq [B][n] on the device."""

    def __init__(self, plan: StackPlan, max_batch: int, eps_regularisation: float = 2e2,
                 be_solver: solver_back_ends = solver_back_ends.qpOASES, device: int = 0):
        if be_solver != solver_back_ends.qpOASES:
            raise RuntimeError("Back-end is not available!")     # BackEndFactory.cpp:87 (the cascade kernel embeds the active-set solver)
        plan.eps_abs = eps_abs_from_factor(eps_regularisation)
        self.stack = BatchedStack(plan, max_batch, device=device)
        self._active = [True] * plan.L

    def solve(self, dev_leaf=None, B=None):
        """dev_leaf: the cycle's leaf tensors (BatchedStack.load_leaf layout): update + solve.  None: solve the assembled
        arrays as they are.  Returns dq [B][n] (a view of the solver's output tensor)."""
        self.stack.level_active = None if all(self._active) else list(self._active)
        if dev_leaf is not None:
            B = self.stack.cycle(dev_leaf)
        else:
            B = self.stack.max_batch if B is None else B
            self.stack.solve(B)
        return self.stack.dq[:B]

    def status(self, B=None):
        return self.stack.status[:(self.stack.max_batch if B is None else B)]

    def getNumberOfTasks(self):
        return self.stack.plan.L

    def setActiveStack(self, stack_index, flag):
        self._active[stack_index] = bool(flag)

    def activateAllStacks(self):
        self._active = [True] * self.stack.plan.L

    def getBackEndName(self):
        return "MI355X dual active set (qpOASES conventions)"

    def setEpsRegularisation(self, eps, stack_index=None):
        raise RuntimeError("eps is part of the static plan of a BatchedStack: construct the solver with it")


class nHQP(iHQP):
    """pyopensot.nHQP (bindings/python/solver.hpp: nHQP(stack, bounds, eps, be_solver), setMinSingularValueRatio,
    setPerformAbRegularization, setPerformSelectiveNullSpaceRegularization)"""

    def __init__(self, plan, max_batch, eps_regularisation=2e2, be_solver=solver_back_ends.qpOASES, device=0, free_vars=None):
        super().__init__(plan, max_batch, eps_regularisation, be_solver, device)
        self._opts = dict(free_vars=free_vars, min_sv_ratio=0.05, ab_regularization=True, selective_ns_regularization=True)

    def setMinSingularValueRatio(self, sv_min):
        if not 0.0 <= sv_min <= 1.0:
            raise ValueError("[nHQP::TaskData::set_min_sv_ratio] Minimum singular value threshold should respect 0 < s < 1")
        self._opts["min_sv_ratio"] = float(sv_min)

    def setPerformAbRegularization(self, flag):
        self._opts["ab_regularization"] = bool(flag)

    def setPerformSelectiveNullSpaceRegularization(self, flag):
        self._opts["selective_ns_regularization"] = bool(flag)

    def solve(self, dev_leaf=None, B=None):
        if dev_leaf is not None:
            B = self.stack.update(dev_leaf)
        else:
            B = self.stack.max_batch if B is None else B
        self.stack.solve_nhqp(B, **self._opts)
        return self.stack.dq[:B]


class eHQP(iHQP):
    """pyopensot.eHQP (bindings/python/solver.hpp:56-61: eHQP(stack), getSigmaMin, setSigmaMin): the equality-only front-end,
    damped pseudo-inverses and projectors (src/solvers/eHQP.cpp); the stack's constraints and bounds are not used"""

    def __init__(self, plan, max_batch, device=0):
        super().__init__(plan, max_batch, device=device)
        self._sigma_min = 1e-12          # eHQP.cpp:56

    def getSigmaMin(self):
        return self._sigma_min

    def setSigmaMin(self, sigma_min):
        if sigma_min > 0:                # eHQP.cpp:156-166: non-positive values are ignored
            self._sigma_min = float(sigma_min)

    def solve(self, dev_leaf=None, B=None):
        if dev_leaf is not None:
            B = self.stack.update(dev_leaf)
        else:
            B = self.stack.max_batch if B is None else B
        self.stack.solve_ehqp(B, self._sigma_min)
        return self.stack.dq[:B]


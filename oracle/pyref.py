"""oracle/pyref.py -- TEST INFRASTRUCTURE ONLY.

ctypes view of oracle/_ref/libqpoases_ref.so (the reference's vendored qpOASES 3.1 behind
our C shim, oracle/ref_qpoases_shim.cpp) plus a numpy restatement of the iHQP cascade
(src/solvers/iHQP.cpp:129-170, 263-358 of the reference) that drives it.  Used by tests/,
tests/golden/make_golden.py and bench.py's cpu_baseline leg; never by the product path.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF_SO = os.path.join(_HERE, "_ref", "libqpoases_ref.so")
_lib = None

HST_ZERO, HST_IDENTITY, HST_POSDEF, HST_POSDEF_NULLSPACE, HST_SEMIDEF, HST_UNKNOWN = range(6)


def available():
    return os.path.exists(_REF_SO)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_REF_SO)
        dp = C.POINTER(C.c_double)
        L.refqp_create.restype = C.c_void_p
        L.refqp_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
        L.refqp_destroy.argtypes = [C.c_void_p]
        L.refqp_eps_abs.restype = C.c_double
        L.refqp_eps_abs.argtypes = [C.c_void_p]
        L.refqp_last_nwsr.argtypes = [C.c_void_p]
        L.refqp_fallbacks.argtypes = [C.c_void_p, C.c_int]
        L.refqp_update_task.argtypes = [C.c_void_p, dp, dp]
        L.refqp_update_constraints.argtypes = [C.c_void_p, dp, dp, dp, C.c_int]
        L.refqp_update_bounds.argtypes = [C.c_void_p, dp, dp]
        L.refqp_init.argtypes = [C.c_void_p, dp, dp, dp, dp, dp, dp, dp]
        L.refqp_solve.argtypes = [C.c_void_p]
        L.refqp_get_solution.argtypes = [C.c_void_p, dp]
        L.refqp_get_dual.argtypes = [C.c_void_p, dp]
        L.refqp_objective.restype = C.c_double
        L.refqp_objective.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


class RefBackEnd:
    """Mirror of OpenSoT::solvers::QPOasesBackEnd (src/solvers/QPOasesBackEnd.cpp) over real qpOASES."""

    def __init__(self, nV, nC, hessian_type=HST_SEMIDEF, eps_factor=2e2, termination_tolerance=0.0):
        self.nV, self.nC = nV, nC
        self._h = lib().refqp_create(nV, nC, hessian_type, eps_factor, termination_tolerance)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().refqp_destroy(self._h)
            self._h = None

    @property
    def eps_abs(self):
        return lib().refqp_eps_abs(self._h)

    def initProblem(self, H, g, A, lA, uA, l, u):
        H, g, A, lA, uA, l, u = map(_c, (H, g, A, lA, uA, l, u))
        if l is not None and l.size == 0:
            l = u = None
        return bool(lib().refqp_init(self._h, _p(H), _p(g), _p(A), _p(lA), _p(uA), _p(l), _p(u)))

    def updateTask(self, H, g):
        H, g = _c(H), _c(g)
        lib().refqp_update_task(self._h, _p(H), _p(g))
        return True

    def updateConstraints(self, A, lA, uA):
        A, lA, uA = _c(A), _c(lA), _c(uA)
        nC = 0 if A is None else A.shape[0]
        self.nC = nC
        return bool(lib().refqp_update_constraints(self._h, _p(A), _p(lA), _p(uA), nC))

    def updateBounds(self, l, u):
        l, u = _c(l), _c(u)
        lib().refqp_update_bounds(self._h, _p(l), _p(u))
        return True

    def solve(self):
        return bool(lib().refqp_solve(self._h))

    def getSolution(self):
        x = np.empty(self.nV)
        lib().refqp_get_solution(self._h, _p(x))
        return x

    def getDual(self):
        y = np.empty(self.nV + self.nC)
        lib().refqp_get_dual(self._h, _p(y))
        return y

    def getObjective(self):
        return lib().refqp_objective(self._h)

    def nWSR(self):
        return lib().refqp_last_nwsr(self._h)


def cost_function(A, w, b, c=None):
    """iHQP::computeCostFunction (iHQP.cpp:129-162): H = A^T W A, g = -A^T W b + c."""
    if w is None:
        H = A.T @ A
        g = -(A.T @ b)
    else:
        WA = w[:, None] * A if w.ndim == 1 else w @ A
        Wb = w * b if w.ndim == 1 else w @ b
        H = A.T @ WA
        g = -(A.T @ Wb)
    if c is not None:
        g = g + c
    return H, g


def ihqp_solve_instance(levels, C_rows, lo, up, l, u, eps_factor, termination_tolerance=0.0,
                        hotstart_repeat=1):
    """Restated iHQP cascade for ONE instance on top of the real qpOASES.

    levels: list of (A, w, b) (w None = identity).  Returns list of per-level x (last = dq).
    Constructor semantics (prepareSoT, iHQP.cpp:172-261) = cold initProblem per level, then
    `hotstart_repeat` calls of solve() with the same data (iHQP.cpp:263-358).
    """
    n = levels[0][0].shape[1]
    xs = []
    bes = []
    prevA, prevAx = [], []
    for k, (A, w, b) in enumerate(levels):
        H, g = cost_function(A, w, b)
        rows = [C_rows] if C_rows is not None and C_rows.shape[0] else []
        rlo = [lo] if rows else []
        rup = [up] if rows else []
        for Aj, Ajx in zip(prevA, prevAx):
            rows.append(Aj); rlo.append(Ajx); rup.append(Ajx)
        if rows:
            Ac = np.vstack(rows); lA = np.concatenate(rlo); uA = np.concatenate(rup)
        else:
            Ac = np.zeros((0, n)); lA = np.zeros(0); uA = np.zeros(0)
        be = RefBackEnd(n, Ac.shape[0], HST_SEMIDEF, eps_factor, termination_tolerance)
        ok = be.initProblem(H, g, Ac, lA, uA, l, u)
        if not ok:
            return None
        for _ in range(hotstart_repeat):
            be.updateTask(H, g)
            be.updateConstraints(Ac, lA, uA)
            if l is not None:
                be.updateBounds(l, u)
            if not be.solve():
                return None
        x = be.getSolution()
        xs.append(x)
        bes.append(be)
        prevA.append(A); prevAx.append(A @ x)
    return xs

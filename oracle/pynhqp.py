"""oracle/pynhqp.py -- TEST INFRASTRUCTURE ONLY (numpy + the reference's qpOASES through oracle/_ref; imported by tests/ and
bench.py's checker legs, never by opensot_amd).

CPU restatement of the reference's NULL-SPACE front-end, OpenSoT::solvers::nHQP (src/solvers/nHQP.cpp):

    solve()                 :155-204   q0 = 0, N_0 = I; per level i:  cost + constraints in the coordinates z of the
                                       cumulated null space N_i, one QP, q0 += N_i z, N_{i+1} = N_i V2
    TaskData::compute_cost  :357-390   AN = A N, b0 = b - A q0; SVD(AN); regularize_A_b; H = AN' W AN, g = -AN' W b0;
                                       H += sv_max V2 V2' (selective null-space regularisation) if a null space is left
    regularize_A_b          :236-279   singular values below thr * sv_max are lifted, the matching components of b0 scaled
    compute_contraints      :282-317   level 0: rows C, box l..u;  below: rows [C N; N], bounds shifted by q0, no box
    constructor             :6-117     free variables per level: n, then the null-space dimension of each level's AN at
                                       construction (singular values >= 1e-6 count as rank) -- STATIC afterwards

PARITY UNPINNED: the reference holds no robot-free vector for nHQP (tests/solvers/TestnHQP.cpp needs a robot model), its
SVD is Eigen's BDCSVD (numpy: LAPACK gesdd) and the level QPs go through a BackEnd.  What this restatement is checked
against: the iHQP path on full-rank stacks, where the two front-ends pose the same lexicographic problem
(tests/test_nhqp.py).  Any orthonormal basis of a level's null space gives the same q (the QP is posed in its coordinates,
the regularisations are basis-invariant), so SVD sign / ordering conventions do not show in the result.
"""
import ctypes as C
import os

import numpy as np

from . import lexcheck
from . import pyoracle as po

DEFAULT_MIN_SV_RATIO = 0.05          # nHQP.h:66
SV_RANK_THRESHOLD = 1e-6             # nHQP.cpp:89


class _RefQP:
    """one qpOASES object of oracle/_ref through its C shim (create / init / get_solution), cold-started per QP"""
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            L = C.CDLL(po._REF_SO)
            dp = po.dp
            L.refqp_create.restype = C.c_void_p
            L.refqp_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
            L.refqp_destroy.argtypes = [C.c_void_p]
            L.refqp_init.argtypes = [C.c_void_p, dp, dp, dp, dp, dp, dp, dp]
            L.refqp_get_solution.argtypes = [C.c_void_p, dp]
            cls._lib = L
        return cls._lib

    @classmethod
    def solve(cls, H, g, A, lA, uA, l, u, eps_factor, term_tol=0.0):
        L = cls.lib()
        n, nc = g.shape[0], (0 if A is None else A.shape[0])
        c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
        H, g, A, lA, uA, l, u = map(c, (H, g, A, lA, uA, l, u))
        p = lambda a: None if a is None else a.ctypes.data_as(po.dp)
        h = L.refqp_create(n, nc, 4, eps_factor, term_tol)
        ok = L.refqp_init(h, p(H), p(g), p(A), p(lA), p(uA), p(l), p(u))
        x = np.zeros(n)
        if ok:
            L.refqp_get_solution(h, p(x))
        L.refqp_destroy(h)
        return bool(ok), x


def _solve_qp(H, g, A, lA, uA, l, u, eps_abs, backend, term_tol):
    if backend == "qpoases":
        return _RefQP.solve(H, g, A, lA, uA, l, u, eps_abs / (1.0e3 * 2.221e-16), term_tol)
    ok, x, _ = po.backend_solve(H, g, A, lA, uA, l, u, eps_abs)
    return ok, x


def regularize_A_b(U, s, Vt, b0, thr):
    """nHQP::TaskData::regularize_A_b (nHQP.cpp:236-279); returns (AN, b0) after the regularisation"""
    sv = s.copy()
    bt = U.T @ b0
    sv_max = sv[0]
    for i in range(bt.shape[0]):
        if i >= sv.shape[0]:
            bt[i] = 0.0
        elif sv[i] < thr * sv_max:
            bt[i] *= sv[i] / (thr * sv_max)
            sv[i] = (thr * sv_max) ** 2 / (sv[i] + thr / 100.0)
    k = sv.shape[0]
    return U[:, :k] @ np.diag(sv) @ Vt[:k, :], U @ bt


def free_variables(asm, inst=0):
    """what the constructor fixes (nHQP.cpp:6-117): free variables of every level = null-space dimension of the level above,
    from the singular values of A N at construction time (here: instance `inst` of the batch)"""
    n, L = asm["n"], asm["L"]
    nf = [n]
    N = np.eye(n)
    for k in range(L - 1):
        AN = lexcheck._level_matrix(asm, inst, k) @ N
        U, s, Vt = np.linalg.svd(AN, full_matrices=True)
        rank = int((s >= SV_RANK_THRESHOLD).sum())
        ns = AN.shape[1] - rank
        if ns <= 0:
            raise RuntimeError(f"[nHQP] No free variables left at layer #{k + 1}: decrease the number of layers!")
        nf.append(ns)
        N = N @ Vt.T[:, AN.shape[1] - ns:]
    return nf


def nhqp_solve(asm, free_vars=None, min_sv_ratio=DEFAULT_MIN_SV_RATIO, ab_regularization=True,
               selective_ns_regularization=True, backend="qpoases", termination_tolerance=0.0):
    """-> dict(dq [B][n], status [B], free_vars).  free_vars: per level (default: free_variables(asm, 0))."""
    n, L, B = asm["n"], asm["L"], asm["B"]
    if asm.get("row_level") is not None and np.any(np.asarray(asm["row_level"]) != 0):
        raise RuntimeError("[nHQP] Local constraints not supported")          # nHQP.cpp:41-44
    nf = list(free_vars) if free_vars is not None else free_variables(asm, 0)
    dq = np.zeros((B, n)); status = np.zeros(B, dtype=np.int32)
    clamp = lexcheck._clamp
    # per-level switches (nHQP::setPerformAbRegularization(level, .), setPerformSelectiveNullSpaceRegularization(level, .),
    # setMinSingularValueRatio(std::vector<double>): nHQP.cpp:127-152, 206-221): a list holds one entry per level
    per_level = lambda v, k: v[k] if isinstance(v, (list, tuple)) else v
    ab_all, sel_all, thr_all = ab_regularization, selective_ns_regularization, min_sv_ratio
    for i in range(B):
        q0 = np.zeros(n)
        N = np.eye(n)
        ok = True
        for k in range(L):
            ab_regularization, selective_ns_regularization = per_level(ab_all, k), per_level(sel_all, k)
            min_sv_ratio = per_level(thr_all, k)
            if min_sv_ratio is None:
                min_sv_ratio = DEFAULT_MIN_SV_RATIO
            A = lexcheck._level_matrix(asm, i, k)
            w = asm["w"][k][i] if asm["w"][k] is not None else np.ones(A.shape[0])
            W = np.diag(w)
            Wd = asm.get("Wdense")
            if Wd is not None and Wd[k] is not None:
                W = Wd[k][i]
            AN = A if k == 0 else A @ N
            b0 = asm["b"][k][i].copy() if k == 0 else asm["b"][k][i] - A @ q0
            U, s, Vt = np.linalg.svd(AN, full_matrices=True)
            sv_max = s[0]
            if ab_regularization:
                AN, b0 = regularize_A_b(U, s, Vt, b0, min_sv_ratio)
            H = AN.T @ W @ AN
            g = -AN.T @ W @ b0
            ns = nf[k + 1] if k + 1 < L else AN.shape[1] - A.shape[0]   # (TaskData's initial ns_dim for the last layer, :336)
            V2 = Vt.T[:, AN.shape[1] - ns:] if ns > 0 else None
            if V2 is not None and selective_ns_regularization:
                H = H + sv_max * (V2 @ V2.T)
            if k == 0:
                rows = asm["C"][i] if asm["nc"] else None
                lo = clamp(asm["lo"][i]) if asm["nc"] else None
                up = clamp(asm["up"][i]) if asm["nc"] else None
                l = asm["l"][i] if asm["l"] is not None else None
                u = asm["u"][i] if asm["u"] is not None else None
            else:
                parts, los, ups = [], [], []
                if asm["nc"]:
                    Cq = asm["C"][i] @ q0
                    parts.append(asm["C"][i] @ N); los.append(clamp(asm["lo"][i]) - Cq); ups.append(clamp(asm["up"][i]) - Cq)
                if asm["l"] is not None:
                    parts.append(N); los.append(clamp(asm["l"][i]) - q0); ups.append(clamp(asm["u"][i]) - q0)
                rows = np.concatenate(parts, axis=0) if parts else None
                lo = np.concatenate(los) if parts else None
                up = np.concatenate(ups) if parts else None
                l = u = None
            okk, z = _solve_qp(H, g, rows, lo, up, l, u, asm["eps_abs"], backend, termination_tolerance)
            if not okk:
                ok = False
                break
            q0 = q0 + (z if k == 0 else N @ z)
            if k + 1 < L:
                if V2 is None:
                    raise RuntimeError("Nullspace basis not available")       # nHQP.cpp:190-193
                N = V2 if k == 0 else N @ V2
        status[i] = 1 if ok else 0
        dq[i] = q0 if ok else 0.0
    return {"dq": dq, "status": status, "free_vars": nf}

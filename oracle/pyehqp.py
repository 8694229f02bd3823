"""oracle/pyehqp.py -- TEST INFRASTRUCTURE ONLY (numpy; imported by tests/ and bench.py's checker legs, never by opensot_amd).

CPU restatement of the reference's EQUALITY-ONLY front-end, OpenSoT::solvers::eHQP (src/solvers/eHQP.cpp: "Prioritized
Multi-Task Motion Control of Redundant Robots under Hard Joint Constraints", Flacco / De Luca / Khatib):

    solve()          :64-95    x = 0, P_0 = I; per level i:  L = chol(W_i),  JP = L' A_i P_{i-1},  SVD(JP) (thin U, V),
                               x += JP^+ (L' b_i - L' A_i x),   P_i = P_{i-1} - V V'
    getDampedPinv()  :124-146  rank = #(sigma_j >= sigma_min * sigma_max) (Eigen SVDBase::rank with setThreshold(sigma_min));
                               if min(sigma) >= sigma_min:  Sigma^+_jj = 1 / sigma_j (j < rank)
                               else (lambda = min(sigma)):  Sigma^+_jj = sigma_j / (sigma_j^2 + lambda^2) (j < rank)
    constructor      :12-62    sigma_min = 1e-12; constraints and bounds of the stack are NOT used ("# OF CONSTRAINTS: 0"),
                               neither is the linear term c of a task (eHQP.h:35)

PARITY UNPINNED: the reference holds no robot-free vector for eHQP (its tests need a robot model).  What this restatement is
checked against (tests/test_ehqp.py): the iHQP path on stacks without constraints whose last level has full column rank,
where both front-ends pose the same lexicographic least-squares problem.  On a level whose projected Jacobian is rank
deficient the thin V of Eigen's JacobiSVD holds implementation-defined completion vectors (they enter P_i); numpy's SVD
(LAPACK gesdd) completes differently, so such levels are comparable only through what they define mathematically: the
level's residual and the residuals of the levels above.
"""
import numpy as np

from . import lexcheck

DEFAULT_SIGMA_MIN = 1e-12          # eHQP.cpp:56


def damped_pinv_weights(s, sigma_min):
    """diagonal of Sigma^+ in J^+ = V Sigma^+ U' (eHQP.cpp:124-146) for the singular values s (descending)"""
    thr = max(s[0] * sigma_min, np.finfo(float).tiny) if s.shape[0] else 0.0
    rank = int((s >= thr).sum())                       # Eigen SVDBase::rank(): leading values that are not below the threshold
    inv = np.zeros_like(s)
    lam = s.min() if s.shape[0] else 0.0
    if lam >= sigma_min:
        inv[:rank] = 1.0 / s[:rank]
    else:
        inv[:rank] = s[:rank] / (s[:rank] ** 2 + lam * lam)
    return inv


def ehqp_solve(asm, sigma_min=DEFAULT_SIGMA_MIN, level_active=None):
    """-> dict(dq [B][n], status [B] (always 1: eHQP::solve returns true), x_levels [B][L][n])"""
    n, L, B = asm["n"], asm["L"], asm["B"]
    dq = np.zeros((B, n)); xl = np.zeros((B, L, n))
    for i in range(B):
        x = np.zeros(n)
        P = np.eye(n)
        for k in range(L):
            if level_active is not None and not level_active[k]:
                xl[i, k] = x
                continue
            A = lexcheck._level_matrix(asm, i, k)
            b = asm["b"][k][i]
            w = asm["w"][k][i] if asm["w"][k] is not None else np.ones(A.shape[0])
            W = np.diag(w)
            Wd = asm.get("Wdense")
            if Wd is not None and Wd[k] is not None:
                W = Wd[k][i]
            Lw = np.linalg.cholesky(W)
            JP = Lw.T @ A @ P
            U, s, Vt = np.linalg.svd(JP, full_matrices=False)
            inv = damped_pinv_weights(s, sigma_min)
            pinv = Vt.T @ np.diag(inv) @ U.T
            x = x + pinv @ (Lw.T @ b - Lw.T @ A @ x)
            P = P - Vt.T @ Vt
            xl[i, k] = x
        dq[i] = x
    return {"dq": dq, "status": np.ones(B, dtype=np.int32), "x_levels": xl}

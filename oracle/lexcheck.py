"""oracle/lexcheck.py -- TEST INFRASTRUCTURE ONLY (numpy; imported by tests/ and bench.py's parity leg, never by opensot_amd).

Which of two points is the optimum of the problem iHQP::solve poses (src/solvers/iHQP.cpp:263-358)?  For one instance and
a candidate chain x_0 .. x_{L-1} (or only its last element dq) this module evaluates, level by level, exactly the QP of
SURVEY.md 8(0):

    level k:  min 1/2 x'(A_k'W_k A_k + Hr + eps I) x + (-A_k'W_k b_k + c_k + gr)'x
              s.t.  l <= x <= u,   lo <= C x <= up  (global rows + the level's task-local rows),
                    A_j x = A_j x_j  for every ACTIVE level j < k       (iHQP.cpp:164-170, 314-332)

and reports
    viol ... the largest constraint violation of x_k in that QP (box, rows, optimality equalities w.r.t. the chain's own x_j)
    kkt .... the stationarity residual | (H_k + eps I) x_k + g_k - N lam |_inf with lam from a NON-NEGATIVE least-squares fit
             on the active set (rows within `act_tol` of a bound; multipliers of inequalities >= 0, of equalities free:
             kkt_fit), plus the most negative multiplier of an active inequality (0 by construction of the fit: a point
             that is not a KKT point shows up as a residual, not as a sign)
    cost ... the level's task cost 1/2 |W^1/2 (A_k x - b_k)|^2 (the eps / regularisation terms are reported separately,
             they break ties only), evaluated at x_k and at the FINAL point dq: the lexicographic cost vector of dq
The lexicographic comparison `lex_compare` orders two feasible points by (cost_0(dq), cost_1(dq), ...) with a relative
tolerance: the smaller one is the better answer to the reference's own problem, whichever solver produced it.
"""
import numpy as np

INFTY = 1.0e20


def _level_matrix(asm, i, k):
    """dense [m_k][n] task matrix of level k, instance i (the implicit Postural block [I 0] expanded)"""
    n, m, ma = asm["n"], asm["m"][k], asm["ma"][k]
    A = np.zeros((m, n))
    if ma:
        A[:ma] = asm["A"][k][i]
    for r in range(ma, m):
        A[r, r - ma] = 1.0
    return A


def _clamp(v):
    return np.clip(np.nan_to_num(v, posinf=INFTY, neginf=-INFTY), -INFTY, INFTY)


def level_qp(asm, i, k, chain, active=None):
    """(H, g, Arows, lo, up, l, u, n_opt) of level k for instance i; optimality rows use `chain` [L][n]"""
    n = asm["n"]
    A = _level_matrix(asm, i, k)
    w = asm["w"][k][i] if asm["w"][k] is not None else np.ones(A.shape[0])
    Wd = asm.get("Wdense")
    if Wd is not None and Wd[k] is not None:
        W = Wd[k][i]
        H = A.T @ W @ A
        g = -A.T @ W @ asm["b"][k][i]
    else:
        H = A.T @ (w[:, None] * A)
        g = -A.T @ (w * asm["b"][k][i])
    if asm["c"][k] is not None:
        g = g + asm["c"][k][i]
    reg = asm.get("reg")
    if reg is not None:
        mr = reg["b"].shape[1]
        Ar = reg["A"][i] if reg.get("A") is not None else np.eye(mr, n)
        H = H + reg.get("w", 1.0) * Ar.T @ Ar
        g = g - reg.get("w", 1.0) * Ar.T @ reg["b"][i]
    H = H + asm["eps_abs"] * np.eye(n)
    rows, lo, up = [], [], []
    if asm["nc"]:
        rl = asm.get("row_level")
        for r in range(asm["nc"]):
            if rl is not None and rl[r] != 0 and rl[r] - 1 != k:
                continue
            rows.append(asm["C"][i, r]); lo.append(asm["lo"][i, r]); up.append(asm["up"][i, r])
    n_glob = len(rows)
    for j in range(k):
        if active is not None and not active[j]:
            continue
        Aj = _level_matrix(asm, i, j)
        v = Aj @ chain[j]
        rows.extend(Aj); lo.extend(v); up.extend(v)
    R = np.array(rows).reshape(-1, n)
    lo = _clamp(np.array(lo, dtype=float)); up = _clamp(np.array(up, dtype=float))
    l = _clamp(asm["l"][i]) if asm["l"] is not None else np.full(n, -INFTY)
    u = _clamp(asm["u"][i]) if asm["u"] is not None else np.full(n, INFTY)
    return H, g, R, lo, up, l, u, R.shape[0] - n_glob


def kkt_fit(N, ineq, grad):
    """stationarity of a point whose active normals are the columns of N (pointing INTO the feasible set): the best
    grad = N lam with lam >= 0 on the inequality columns and free on the equality columns (non-negative least squares;
    VERDICT r2: a plain least-squares fit on linearly dependent normals can print a negative multiplier for a point that
    does have a non-negative set, e.g. six active box bounds next to 27 optimality rows).  -> (residual max-norm, lam,
    plain least-squares residual).  The columns are scaled to unit norm (unit rows sit next to Jacobian rows)."""
    from scipy.optimize import nnls
    sc = np.linalg.norm(N, axis=0); sc[sc == 0] = 1.0
    Ns = N / sc
    ineq = np.asarray(ineq, dtype=bool)
    M = np.concatenate([Ns[:, ineq], Ns[:, ~ineq], -Ns[:, ~ineq]], axis=1)
    gs = np.abs(grad).max()
    gs = gs if gs > 0 else 1.0
    y, _ = nnls(M, grad / gs, maxiter=50 * max(10, M.shape[1]))
    y = y * gs
    ni, ne = int(ineq.sum()), int((~ineq).sum())
    lam = np.zeros(N.shape[1])
    lam[ineq] = y[:ni]
    lam[~ineq] = y[ni:ni + ne] - y[ni + ne:]
    res = float(np.abs(Ns @ lam - grad).max())
    ls, *_ = np.linalg.lstsq(Ns, grad, rcond=None)
    return res, lam / sc, float(np.abs(Ns @ ls - grad).max())


def kkt_certificate(asm, i, dq, active=None, act_tol=1e-8):
    """Without any witness: is dq a lexicographic optimum of iHQP's problem (iHQP.cpp:263-358) on its own evidence?  Per
    active level k, with the optimality rows of the levels above posed AT dq (A_j x = A_j dq): feasibility of dq in that QP
    and the non-negative-multiplier stationarity residual (kkt_fit).  The last level is exact (dq is its minimiser); a level
    above it is stationary up to its eps term only (x_k minimises cost_k + eps/2 |x|^2 and the levels below move x inside
    {A_k x = A_k x_k}), so its residual is compared with eps |dq|.  -> list of dict(level, viol, kkt, kkt_allowed)."""
    L = asm["L"]
    levels = [k for k in range(L) if active is None or active[k]]
    chain = [dq] * L
    out = []
    for k in levels:
        r = level_report(asm, i, k, chain, active, act_tol)
        last = k == levels[-1]
        allowed = 1e-9 * max(1.0, np.abs(dq).max()) if last else 10.0 * asm["eps_abs"] * max(1.0, np.abs(dq).max()) + 1e-9
        out.append({"level": k, "viol": r["viol"], "kkt": r["kkt"], "kkt_allowed": allowed, "min_mult": r["min_mult"]})
    return out


def level_report(asm, i, k, chain, active=None, act_tol=1e-8):
    """dict(viol, kkt, min_mult, cost, n_active) of chain[k] in the QP of level k"""
    H, g, R, lo, up, l, u, _ = level_qp(asm, i, k, chain, active)
    x = chain[k]
    n = x.shape[0]
    viol = max(0.0, float((l - x).max()), float((x - u).max()))
    normals, ineq = [], []
    for c in range(n):
        sc = max(1.0, abs(l[c]) if l[c] > -INFTY else 0.0, abs(u[c]) if u[c] < INFTY else 0.0)
        if l[c] > -INFTY and x[c] - l[c] <= act_tol * sc:
            e = np.zeros(n); e[c] = 1.0; normals.append(e); ineq.append(l[c] < u[c])
        elif u[c] < INFTY and u[c] - x[c] <= act_tol * sc:
            e = np.zeros(n); e[c] = -1.0; normals.append(e); ineq.append(True)
    if R.shape[0]:
        ax = R @ x
        viol = max(viol, float(np.where(lo > -INFTY, lo - ax, 0.0).max()), float(np.where(up < INFTY, ax - up, 0.0).max()))
        for r in range(R.shape[0]):
            sc = max(1.0, abs(lo[r]) if lo[r] > -INFTY else 0.0, abs(up[r]) if up[r] < INFTY else 0.0)
            if lo[r] == up[r]:
                normals.append(R[r]); ineq.append(False)
            elif lo[r] > -INFTY and ax[r] - lo[r] <= act_tol * sc:
                normals.append(R[r]); ineq.append(True)
            elif up[r] < INFTY and up[r] - ax[r] <= act_tol * sc:
                normals.append(-R[r]); ineq.append(True)
    grad = H @ x + g
    kkt, min_mult = float(np.abs(grad).max()), 0.0
    if normals:
        kkt, lam, _ = kkt_fit(np.array(normals).T, np.array(ineq), grad)
        im = np.array(ineq)
        if im.any():
            min_mult = float(min(0.0, lam[im].min()))
    A = _level_matrix(asm, i, k)
    w = asm["w"][k][i] if asm["w"][k] is not None else np.ones(A.shape[0])
    r = A @ x - asm["b"][k][i]
    return {"viol": viol, "kkt": kkt, "min_mult": min_mult, "cost": float(0.5 * (w * r * r).sum()), "n_active": len(normals)}


def lex_costs(asm, i, x, active=None):
    """task cost of every (active) level at ONE point x: the lexicographic cost vector of a final answer"""
    out = []
    for k in range(asm["L"]):
        if active is not None and not active[k]:
            out.append(0.0); continue
        A = _level_matrix(asm, i, k)
        w = asm["w"][k][i] if asm["w"][k] is not None else np.ones(A.shape[0])
        r = A @ x - asm["b"][k][i]
        Wd = asm.get("Wdense")
        if Wd is not None and Wd[k] is not None:
            out.append(float(0.5 * r @ Wd[k][i] @ r))
        else:
            out.append(float(0.5 * (w * r * r).sum()))
    return out


def global_violation(asm, i, x):
    """largest violation of the box and the GLOBAL rows at x (the constraints every level shares)"""
    v = 0.0
    if asm["l"] is not None:
        v = max(v, float((_clamp(asm["l"][i]) - x).max()), float((x - _clamp(asm["u"][i])).max()))
    if asm["nc"]:
        rl = asm.get("row_level")
        for r in range(asm["nc"]):
            if rl is not None and rl[r] != 0:
                continue
            ax = float(asm["C"][i, r] @ x)
            lo, up = _clamp(asm["lo"][i, r]), _clamp(asm["up"][i, r])
            if lo > -INFTY: v = max(v, lo - ax)
            if up < INFTY: v = max(v, ax - up)
    return v


def lex_compare(ca, cb, rtol=1e-9, atol=1e-14, rtol_better=None):
    """-1: cost vector ca is lexicographically smaller (better), +1: cb is, 0: equal within tolerance at every level.
    rtol_better (default: rtol) is the tolerance for calling ca BETTER at a level; with rtol_better << rtol the test is
    the one-sided "ca is not worse than cb": a level where ca is lower by more than round-off decides for ca (a point
    that is really better at level k owes nothing at the levels below), a level where ca is higher only counts when the
    excess is beyond what the level's conditioning resolves (rtol)."""
    if rtol_better is None:
        rtol_better = rtol
    for a, b in zip(ca, cb):
        m = max(abs(a), abs(b))
        if a < b - (atol + rtol_better * m):
            return -1
        if b < a - (atol + rtol * m):
            return 1
    return 0


def instance_evidence(asm, i, chain_a, chain_b, active=None, names=("device", "qpOASES"), feas_tol=1e-9):
    """per-level KKT / violation / cost of two chains, the lexicographic cost vector of their final points and the
    verdict -- what bench.py prints for every instance whose two answers differ by more than the tolerance"""
    L = asm["L"]
    last = max(k for k in range(L) if active is None or active[k])
    out = {"instance": int(i)}
    for name, ch in zip(names, (chain_a, chain_b)):
        reps = [level_report(asm, i, k, ch, active) for k in range(L) if active is None or active[k]]
        out[name] = {"viol_per_level": [r["viol"] for r in reps], "kkt_per_level": [r["kkt"] for r in reps],
                     "min_multiplier_per_level": [r["min_mult"] for r in reps],
                     # the final point on its own evidence (kkt_certificate): residual of the non-negative-multiplier fit per
                     # level against what the level's eps term allows
                     "kkt_certificate_of_dq": [{"level": c["level"], "kkt": c["kkt"], "allowed": c["kkt_allowed"], "viol": c["viol"]}
                                               for c in kkt_certificate(asm, i, ch[last], active)],
                     "lex_cost_of_dq": lex_costs(asm, i, ch[last], active),
                     "global_violation_of_dq": global_violation(asm, i, ch[last])}
    cmp_ = lex_compare(out[names[0]]["lex_cost_of_dq"], out[names[1]]["lex_cost_of_dq"])
    out["lexicographically_smaller_cost"] = names[0] if cmp_ < 0 else (names[1] if cmp_ > 0 else "tie")
    # the optimum of the reference's problem is the lexicographically smallest FEASIBLE point: a point that violates the
    # shared constraints (here: by more than `feas_tol`, and by more than the other one) bought its cost with that violation
    va, vb = (max(out[nm]["global_violation_of_dq"], max(out[nm]["viol_per_level"])) for nm in names)
    fa, fb = va <= feas_tol, vb <= feas_tol
    if fa and not fb:
        verdict = f"{names[0]} (feasible to {va:.1e}; {names[1]}'s point violates its constraints by {vb:.1e})"
    elif fb and not fa:
        verdict = f"{names[1]} (feasible to {vb:.1e}; {names[0]}'s point violates its constraints by {va:.1e})"
    else:
        verdict = out["lexicographically_smaller_cost"]
    out["lexicographically_better"] = verdict.split(" ")[0]
    out["verdict"] = verdict
    out["max_abs_diff"] = float(np.abs(chain_a[last] - chain_b[last]).max())
    return out

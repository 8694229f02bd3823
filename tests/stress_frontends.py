"""TEST INFRASTRUCTURE (run on the GPU box, by hand or by tests/test_ehqp.py): randomised sweep of the eHQP and nHQP front-ends
(HIP kernels through the C-ABI) against their numpy restatements (oracle/pyehqp.py, oracle/pynhqp.py) over random small
stacks -- exercises the symmetric eigen-solver (sym_eig32) on many sizes, spectra and batch members."""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np, torch
from opensot_amd import synth
from opensot_amd.solver import BatchedStack
from oracle import pyoracle as oracle, pyehqp, pynhqp

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
WIDE = len(sys.argv) > 3 and sys.argv[3] == "wide"      # every stack with a level beyond 32 on both sides (osot_nhqp_prepare_wide_kernel)
B = 24
bad = 0
worst_e = worst_n = 0.0
n_nhqp = 0
for it in range(N):
    n = int(rng.integers(2, 33)) if rng.integers(0, 4) else int(rng.integers(33, 65))      # (a quarter of the stacks beyond 32 variables: the 64-lane / 64-column kernels)
    if WIDE:
        n = int(rng.integers(34, 65))
    L = int(rng.integers(1, 4))
    rows, left = [], n
    for k in range(L):
        if left <= 1:
            break
        m = int(rng.integers(1, max(2, min(left, 24))))
        if WIDE and k == 0:
            m = int(rng.integers(33, 65))          # 33 .. 64 rows in 34 .. 64 variables; the levels below work in what is left
        rows.append(m); left -= m
    postural = bool(rng.integers(0, 2)) or left <= 0
    seed = int(rng.integers(1 << 30))
    constrained = bool(rng.integers(0, 2))      # (eHQP ignores the constraints, nHQP solves with them)
    plan, leaf = synth.make_generic_stack(B, n, rows, n_eq=0, n_ineq=int(rng.integers(0, 4)) if constrained else 0, seed=seed,
                                          box=float(rng.choice([0.2, 0.6])) if constrained else 0.0, postural_last=postural,
                                          eps_factor=float(rng.choice([1e6, 2e2])))
    asm = oracle.assemble(plan, leaf)
    st = BatchedStack(plan, B, device=0)
    st.load_assembled(asm)
    st.solve_ehqp(B)
    torch.cuda.synchronize()
    e = pyehqp.ehqp_solve(asm)
    de = np.abs(st.dq[:B].cpu().numpy() - e["dq"]).max()
    # (the QR kernel of round 3; the Gram-side eigen-decomposition of round 2 left cond(JP)^2 eps: 2e-7 on the worst of 900 stacks)
    ok = de < 1e-9 and (st.status[:B].cpu().numpy() == 0).all()
    if sum(rows) >= n and postural:     # (wide mode: a first level of full column rank leaves the Postural level an EMPTY null space; eHQP's
        ok = True; de = 0.0             #  damped pseudo-inverse of the projected Jacobian -- pure round-off -- is not defined there: the
                                        #  restatement itself returns 1e15.  nHQP below is what the mode is for.)
    worst_e = max(worst_e, de)
    dn = 0.0
    if sum(rows) < n or postural:      # nHQP needs free variables at every layer below the first
        try:
            ref = pynhqp.nhqp_solve(asm, backend="qpoases" if oracle.ref_available() else "eiqp", termination_tolerance=10 * 2.221e-16)
            st.solve_nhqp(B)
            torch.cuda.synchronize()
            okr = ref["status"] == 1
            if okr.any():
                dn = np.abs(st.dq[:B].cpu().numpy()[okr] - ref["dq"][okr]).max()
                ok = ok and dn < 1e-6 and (st.status[:B].cpu().numpy()[okr] == 0).all()
                worst_n = max(worst_n, dn); n_nhqp += 1
        except RuntimeError:
            pass                          # (a stack the reference's constructor refuses: no free variables left; or, beyond 32
                                          #  variables, a level wider than the 32-wide eigen-solver: refused by the product)
    if not ok:
        bad += 1
        print("MISMATCH", dict(n=n, rows=rows, postural=postural, seed=seed), "eHQP %.2e nHQP %.2e" % (de, dn), flush=True)
print(f"{N} stacks x {B} instances: {bad} with a mismatch; worst |dq - restatement|: eHQP {worst_e:.2e}, nHQP {worst_n:.2e} ({n_nhqp} stacks)")

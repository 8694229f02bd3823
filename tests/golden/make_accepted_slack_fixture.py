"""TEST INFRASTRUCTURE.  Builds tests/golden/default_eps_accepted_slack_instance.npz: the ONE instance of round 4's randomised parity
sweep (profiles/r04_stress_parity.txt: generic stack, 25 variables, levels of 18 / 7 / 4 rows, 2 global inequality rows, DEFAULT eps
factor 2e2, seed 309140371, instance 18) whose device answer failed the literal acceptance rule: the kernel accepted a violation of
4.2e-7 as round-off of the levels above (kSlackTol), the rule asks for feasibility to 1e-7.  Stored: the assembled arrays of the
instance (what oracle.pyoracle.assemble returns) and the three witnesses' answers, computed here with the reference's qpOASES 3.1
(oracle/_ref) at OpenSoT's options, run to its exact optimum, and the restated eiQuadProg."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from opensot_amd import synth
from oracle import pyoracle as oracle

KW = dict(n_eq=0, n_ineq=2, seed=309140371, box=0.0, postural_last=False, eps_factor=200.0)
N, ROWS, B, INST = 25, [18, 7, 4], 192, 18


def plan_of():
    return synth.make_generic_stack(1, N, ROWS, **KW)[0]


if __name__ == "__main__":
    assert oracle.ref_available()
    plan, leaf = synth.make_generic_stack(B, N, ROWS, **KW)
    asm = oracle.assemble(plan, leaf)
    sl = slice(INST, INST + 1)
    out = {}
    for k in range(plan.L):
        if asm["A"][k] is not None:
            out[f"A{k}"] = asm["A"][k][sl]
        out[f"b{k}"] = asm["b"][k][sl]; out[f"w{k}"] = asm["w"][k][sl]
    for key in ("C", "lo", "up", "l", "u"):
        if asm[key] is not None:
            out[key] = asm[key][sl]
    sub = dict(asm, B=1, A=[None if a is None else a[sl] for a in asm["A"]], b=[a[sl] for a in asm["b"]], w=[a[sl] for a in asm["w"]],
               c=[None if a is None else a[sl] for a in asm["c"]], **{k: (None if asm[k] is None else asm[k][sl]) for k in ("C", "lo", "up", "l", "u")})
    for nm, kw in (("qpoases", {}), ("qpoases_exact", dict(termination_tolerance=10 * 2.221e-16))):
        r = oracle.ihqp_solve_batch(sub, oracle.BE_QPOASES_REF, nthreads=1, **kw)
        out["dq_" + nm], out["ok_" + nm] = r["dq"], (r["status"] == 1)
    r = oracle.ihqp_solve_batch(sub, oracle.BE_EIQP_EQ, nthreads=1)
    out["dq_eiquadprog"], out["ok_eiquadprog"] = r["dq"], (r["status"] == 1)
    assert all(a is None for a in asm["c"])
    np.savez(os.path.join(ROOT, "tests", "golden", "default_eps_accepted_slack_instance.npz"), **out)
    print({k: v.shape for k, v in out.items()})

"""developer helper (GPU box): one saved closed-loop instance (tests/stress_closed_loop.py's closed_loop_bug_*.npz) through the device
cascade and the witnesses.  usage: python tools/replay_bug.py FILE.npz [tasks|ttc] [eps_factor]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from helpers import closed_loop_plan
from opensot_amd.solver import BatchedStack
from oracle import pyoracle as oracle
f, mode, eps = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "ttc"), float(sys.argv[3]) if len(sys.argv) > 3 else 2e2
plan, _ = closed_loop_plan(mode, eps)
z = np.load(f)
L = plan.L
asm = {"n": plan.n, "B": 1, "L": L, "eps_abs": plan.eps_abs, "m": [plan.m(k) for k in range(L)], "ma": [plan.ma(k) for k in range(L)],
       "A": [z[f"A{k}"] if f"A{k}" in z.files else None for k in range(L)], "b": [z[f"b{k}"] for k in range(L)],
       "w": [z[f"w{k}"] for k in range(L)], "c": [None] * L, "nc": plan.nc, "C": z["C"], "lo": z["lo"], "up": z["up"], "l": z["l"], "u": z["u"]}
st = BatchedStack(plan, 1, device=0)
st.load_assembled(asm)
st.solve(1)
torch.cuda.synchronize()
rx = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=1)
print(os.path.basename(os.environ.get("OSOT_MI355X_LIB", "default")), "device status", int(st.status[0]), "iterations", int(st.iterations[0]),
      "witness", int(rx["status"][0]), "max|dq - dq_qpOASES|", float(np.abs(st.dq[:1].cpu().numpy() - rx["dq"]).max()))

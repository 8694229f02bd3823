#!/usr/bin/env python3
"""tests/golden/make_golden.py -- regenerates the golden vectors (run in the BUILD container only).

Needs /root/reference: the golden solutions come from the reference's own vendored qpOASES 3.1, compiled
in place into oracle/_ref (oracle/Makefile) and driven with OpenSoT's conventions (option set, eps on
diag(H), +-1e20 clamp, cold initProblem then hot-started solve -- oracle/ref_qpoases_shim.cpp) by the
restated iHQP cascade (oracle/osot_oracle.c).  Only data is written: inputs and expected outputs.

Per config (C2, C3, C4: 32 seeded instances each; C5 = 38-DoF floating-base inverse dynamics: 16) one .npz with
  leaf inputs .............. what XBot::ModelInterface would supply (synthetic, opensot_amd/synth.py)
  asm_* ..................... AutoStack::update() outputs from the oracle's leaf restatement
  x_ref [B][L][n] ........... per-level x of qpOASES with OpenSoT's option set (terminationTolerance 2.2e-7)
  x_exact [B][L][n] ......... same, terminationTolerance = 10*EPS ("exact" oracle, SURVEY.md 7 hard part 1)
  ok_ref, ok_exact .......... qpOASES success flags
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from opensot_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def flatten_leaf(leaf):
    out = {}
    for k, a in enumerate(leaf["A"]):
        if a is not None:
            out[f"leaf_A{k}"] = a
    for k, lev in enumerate(leaf["task"]):
        for j, t in enumerate(lev):
            for i, x in enumerate(t):
                if x is not None:
                    out[f"leaf_task{k}_{j}_p{i}"] = x
    for j, t in enumerate(leaf["bound"]):
        for i, x in enumerate(t):
            if x is not None:
                out[f"leaf_bound{j}_p{i}"] = x
    for j, t in enumerate(leaf["rows"]):
        for i, x in enumerate(t):
            if x is not None:
                out[f"leaf_rows{j}_p{i}"] = x
    for j, x in enumerate(leaf.get("C", [])):
        if x is not None:
            out[f"leaf_C{j}"] = x
    return out


def main():
    po.build()
    assert po.ref_available(), "oracle/_ref/libqpoases_ref.so missing: run `make -C oracle ref` where /root/reference exists"
    B = 32
    for cfg in ("C2", "C3", "C4", "C5"):
        seed = {"C2": 20260, "C3": 30260, "C4": 40260, "C5": 50260}[cfg]
        if cfg == "C5":
            B = 16
            plan, leaf = synth.make_id_stack(B, seed=seed)
        else:
            plan, leaf = synth.make_velocity_stack(cfg, B, seed=seed)
        asm = po.assemble(plan, leaf)
        ref = po.ihqp_solve_batch(asm, po.BE_QPOASES_REF, nthreads=1)
        exact = po.ihqp_solve_batch(asm, po.BE_QPOASES_REF, nthreads=1, termination_tolerance=10 * 2.221e-16)
        d = {"config": cfg, "seed": seed, "B": B, "eps_abs": plan.eps_abs,
             "x_ref": ref["x_levels"], "ok_ref": ref["status"],
             "x_exact": exact["x_levels"], "ok_exact": exact["status"]}
        d.update(flatten_leaf(leaf))
        for k in range(asm["L"]):
            d[f"asm_b{k}"] = asm["b"][k]
            d[f"asm_w{k}"] = asm["w"][k]
        for name in ("C", "lo", "up", "l", "u"):
            if asm[name] is not None:
                d[f"asm_{name}"] = asm[name]
        path = os.path.join(HERE, f"{cfg}_b{B}.npz")
        np.savez_compressed(path, **d)
        print(cfg, "->", path, os.path.getsize(path) // 1024, "KiB; ref ok", int(ref["status"].sum()),
              "exact ok", int(exact["status"].sum()),
              "max|x_ref - x_exact| last level", np.abs(ref["x_levels"][:, -1] - exact["x_levels"][:, -1]).max())


if __name__ == "__main__":
    main()

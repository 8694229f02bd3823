#!/usr/bin/env python3
"""developer A/B (GPU box): the headline batch (two lanes of 2048, stream launches) and a 32768-instance single launch with
the library named by OSOT_MI355X_LIB; with library paths as arguments it runs itself once per library.
usage: python tools/ab_headline.py [lib.so ...]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "--one":
    for lib in ["default"] + sys.argv[1:]:
        env = dict(os.environ)
        if lib != "default":
            env["OSOT_MI355X_LIB"] = os.path.abspath(lib)
        subprocess.run([sys.executable, __file__, "--one"], env=env)
    sys.exit(0)
sys.path.insert(0, ROOT)
import numpy as np
import torch
from opensot_amd import synth
from opensot_amd.solver import BatchedStack

K, steps = 4, 50


def sub(lf, lo, hi):
    cut = lambda a: None if a is None else a[lo:hi]
    return {"B": hi - lo, "A": [cut(a) for a in lf["A"]],
            "task": [[tuple(cut(x) for x in t) for t in lev] for lev in lf["task"]],
            "bound": [tuple(cut(x) for x in t) for t in lf["bound"]],
            "rows": [tuple(cut(x) for x in t) for t in lf["rows"]]}


def run(cfg, B, S, reps=3):
    plan, leaf = synth.make_id_stack(B, seed=3000) if cfg == "C5" else synth.make_velocity_stack(cfg, B, seed=3000)
    rng = np.random.default_rng(77)
    leaves = [leaf]
    for _ in range(K - 1):
        leaves.append(synth.perturb(leaves[-1], rng, 0.01))
    streams = [torch.cuda.Stream() for _ in range(S)]
    lanes = []
    for s in range(S):
        lo, hi = s * B // S, (s + 1) * B // S
        st = BatchedStack(plan, hi - lo, device=0, want_levels=False)
        devs, As = [], []
        for lf in leaves:
            st.A = [None if a is None else torch.empty_like(a) for a in st.A]
            devs.append(st.load_leaf(sub(lf, lo, hi)))
            As.append(st.A)
        lanes.append((st, devs, As))
    torch.cuda.synchronize()

    def step(i):
        k = i % K
        for s, (st, devs, As) in enumerate(lanes):
            with torch.cuda.stream(streams[s]):
                st.A = As[k]
                st.cycle(devs[k])
    for i in range(8):
        step(i)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        for i in range(steps):
            step(8 + i)
        torch.cuda.synchronize()
        best = max(best, B * steps / (time.perf_counter() - t0) / 1e6)
    ok = sum(int((st.status == 0).sum().item()) for st, _, _ in lanes)
    dq = torch.cat([st.dq for st, _, _ in lanes]).double().cpu().numpy()
    return best, ok, float(np.abs(dq).sum())


lib = os.path.basename(os.environ.get("OSOT_MI355X_LIB", "default"))
out = []
for cfg, B, S in (("C3", 4096, 2), ("C3", 32768, 1)):
    v, ok, chk = run(cfg, B, S)
    out.append(f"{cfg} B={B} S={S}: {v:.3f} M/s ok {ok}/{B} chk {chk:.9e}")
print(lib, "|", " | ".join(out), flush=True)

"""developer helper (GPU box): solve dumped closed-loop instances with several builds of the HIP library.
usage: python tools/dev_ab.py mode eps_factor file.npz lib1.so [lib2.so ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch
    from helpers import closed_loop_plan
    from opensot_amd.solver import BatchedStack
    mode, eps, fn = sys.argv[2], float(sys.argv[3]), sys.argv[4]
    plan, _ = closed_loop_plan(mode, eps)
    z = np.load(fn); L = plan.L; B = z["b0"].shape[0]
    asm = {"n": plan.n, "B": B, "L": L, "eps_abs": plan.eps_abs, "m": [plan.m(k) for k in range(L)], "ma": [plan.ma(k) for k in range(L)],
           "A": [z[f"A{k}"] if f"A{k}" in z.files else None for k in range(L)], "b": [z[f"b{k}"] for k in range(L)],
           "w": [z[f"w{k}"] for k in range(L)], "c": [None] * L, "nc": plan.nc, "C": z["C"], "lo": z["lo"], "up": z["up"], "l": z["l"], "u": z["u"]}
    st = BatchedStack(plan, B, device=0)
    st.load_assembled(asm); st.solve(B); torch.cuda.synchronize()
    print(os.environ.get("OSOT_MI355X_LIB", "default"), "status", st.status[:B].cpu().numpy(), "iters", st.iterations[:B].cpu().numpy(),
          "slack", st.accepted_slack[:B].cpu().numpy(), "|x_levels|", [float(st.x_levels[:B, k].abs().max()) for k in range(L)], flush=True)
else:
    mode, eps, fn = sys.argv[1:4]
    for lib in ["default"] + sys.argv[4:]:
        env = dict(os.environ)
        if lib != "default":
            env["OSOT_MI355X_LIB"] = os.path.abspath(lib)
        subprocess.run([sys.executable, __file__, "--one", mode, eps, fn], env=env)

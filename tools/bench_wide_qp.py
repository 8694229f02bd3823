"""developer helper (GPU box): throughput of the explicit-QP surface beyond 64 variables (osot_qp_big.h: one 256-thread workgroup per
QP) on the two levels of a floating-base inverse-dynamics stack of 70 and of 88 variables, against the reference's qpOASES on the host"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from opensot_amd import abi
from oracle import pyoracle as oracle
from helpers import id_like_levels, ref_qpoases_solve
dev = torch.device("cuda", 0)
for (nv, ncon) in ((55, 5), (61, 9)):
    B = 1024
    probs = []
    for b in range(B):
        rng = np.random.default_rng(1000 + b)
        n, level = id_like_levels(rng, nv, ncon)
        q0 = level(0, [])
        probs.append((level, q0))
    for k in range(2):
        if k == 1:
            qs = [p[0](1, [xs[i]]) for i, p in enumerate(probs)]
        else:
            qs = [p[1] for p in probs]
        nc = qs[0][2].shape[0]
        st_ = lambda j: np.stack([q[j] for q in qs])
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
        ts = [t(st_(j)) for j in range(7)]
        x = torch.zeros((B, n), dtype=torch.float64, device=dev); st = torch.zeros((B,), dtype=torch.int32, device=dev); it = torch.zeros((B,), dtype=torch.int32, device=dev)
        p = lambda a: C.c_void_p(a.data_ptr())
        eps = 2.221e-7
        call = lambda: abi.lib().osot_qp_solve_batch(B, n, nc, *[p(a) for a in ts], eps, 0, p(x), p(st), p(it), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert call() == abi.OK; torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): call()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 3
        xs = x.cpu().numpy()
        ok = int((st == 0).sum().item())
        t1 = time.perf_counter(); nref = 16
        for i in range(nref):
            r = ref_qpoases_solve(*qs[i], eps / 2.221e-13) or oracle.backend_solve(*qs[i], eps)[:2]
            assert r[0] and np.abs(r[1] - xs[i]).max() < 1e-6
        ref_ms = 1e3 * (time.perf_counter() - t1) / nref
        print(f"n = {n}, {nc} rows, level {k}: {B} QPs in {ms:.2f} ms = {B / ms:.1f} k QPs/s, solved {ok}/{B}, mean iterations {float(it.float().mean()):.1f}; "
              f"the reference's qpOASES on one host thread: {ref_ms:.2f} ms per QP = {1.0 / ref_ms:.2f} k QPs/s", flush=True)

"""TEST INFRASTRUCTURE (run on the GPU box by tests/test_gpu_backend.py, or by hand): randomised sweep of osot_qp_solve_batch (the BackEnd-convention kernel: explicit H, g,
rows, box) over shapes n = 2..64 (third argument "wide": 65..128, the workgroup-per-QP path) with full-rank and rank-deficient Hessians; every instance is checked by KKT and a
sample against the oracle's single-QP solve (eiQuadProg restatement; qpOASES where oracle/_ref exists)."""
import os, sys, time, ctypes as C
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import numpy as np, torch
from opensot_amd import abi
from oracle import pyoracle as oracle
from helpers import random_qp, kkt_check, ref_qpoases_solve

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
WIDE = len(sys.argv) > 3 and sys.argv[3] == "wide"      # 65 .. 128 variables: the workgroup-per-QP path (osot_qp_big.h, round 6)
dev = torch.device("cuda", 0)
bad = 0; t0 = time.time(); total = 0
for it_ in range(N):
    n = int(rng.integers(2, 65)); nc = int(rng.integers(0, 40)); n_eq = int(rng.integers(0, min(nc, n // 2) + 1)) if nc else 0
    B = 64
    if WIDE:
        n = int(rng.integers(65, 129)); nc = int(rng.integers(0, 140)); n_eq = int(rng.integers(0, min(nc, n // 2) + 1)) if nc else 0
        B = 16
    H, g, A, lA, uA, l, u = random_qp(rng, B, n, nc, n_eq, box=bool(rng.integers(0, 4) > 0))
    eps = float(rng.choice([1e-9, 2.221e-7, 4.442e-11]))
    deficient = bool(rng.integers(0, 3) == 0)
    if deficient:                      # H = M'M with fewer rows than variables: needs eps and the box to be well posed
        r = int(rng.integers(1, n))
        M = rng.normal(size=(B, r, n)); H = np.einsum("bki,bkj->bij", M, M)
        g = -np.einsum("bki,bk->bi", M, rng.normal(size=(B, r)))          # g in range(M'): the known-answer structure
        eps = max(eps, 2.221e-7)
        if l is None:
            l = -rng.uniform(0.05, 0.5, size=(B, n)); u = rng.uniform(0.05, 0.5, size=(B, n))
    t = lambda a: None if a is None else torch.as_tensor(a, dtype=torch.float64, device=dev).contiguous()
    tH, tg, tA, tlA, tuA, tl, tu = map(t, (H, g, A, lA, uA, l, u))
    x = torch.zeros((B, n), dtype=torch.float64, device=dev)
    st = torch.full((B,), -1, dtype=torch.int32, device=dev); itr = torch.zeros((B,), dtype=torch.int32, device=dev)
    p = lambda a: None if a is None else C.c_void_p(a.data_ptr())
    rc = abi.lib().osot_qp_solve_batch(B, n, nc, p(tH), p(tg), p(tA), p(tlA), p(tuA), p(tl), p(tu), eps, 0,
                                       p(x), p(st), p(itr), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == abi.OK, abi.lib().osot_last_error()
    torch.cuda.synchronize()
    xs = x.cpu().numpy(); sts = st.cpu().numpy()
    total += B
    msg = None
    args = lambda i: (H[i], g[i], None if A is None else A[i], None if A is None else lA[i], None if A is None else uA[i],
                      None if l is None else l[i], None if l is None else u[i], eps)
    solved = sts == 0
    for i in np.nonzero(~solved)[0][:12]:          # not solved: is the instance infeasible for the oracles too?
        ok_e, _, _ = oracle.backend_solve(*args(i))
        ok_q = oracle.backend_solve(*args(i), form=oracle.BE_QPOASES_REF)[0] if oracle.ref_available() else False
        if ok_e or ok_q:
            msg = f"status {sts[i]} on instance {i} which the oracle solves (eiqp {ok_e}, qpOASES {ok_q}); {int((~solved).sum())} unsolved"
    if msg is None and solved.any():
        worst = 0.0
        for i in np.nonzero(solved)[0]:
            worst = max(worst, kkt_check(H[i], g[i], None if A is None else A[i], None if A is None else lA[i], None if A is None else uA[i],
                                         None if l is None else l[i], None if l is None else u[i], xs[i], eps, tol=1e-6))
        if worst > 1e-5:
            msg = f"KKT residual {worst:.2e}"
        for i in np.nonzero(solved)[0][::16]:
            ok, xo, _ = oracle.backend_solve(*args(i))
            if ok and np.abs(xs[i] - xo).max() > 1e-6 * max(1.0, np.abs(xo).max()):
                msg = f"differs from the oracle by {np.abs(xs[i] - xo).max():.2e} (instance {i})"
            rq = ref_qpoases_solve(*args(i)[:7], eps / 2.221e-13)       # the reference's own qpOASES 3.1 (oracle/_ref), where present
            if rq is not None and rq[0] and np.abs(xs[i] - rq[1]).max() > 1e-6 * max(1.0, np.abs(rq[1]).max()):
                # (qpOASES stops at its MPC option set's tolerance: the oracle's exact answer decides whether this is the reference's slack)
                if not (ok and np.abs(xs[i] - xo).max() <= 1e-8 * max(1.0, np.abs(xo).max())):
                    msg = f"differs from the reference's qpOASES by {np.abs(xs[i] - rq[1]).max():.2e} (instance {i})"
    if msg:
        bad += 1
        print("MISMATCH n=%d nc=%d n_eq=%d box=%s eps=%.1e deficient=%s: %s" % (n, nc, n_eq, l is not None, eps, deficient, msg), flush=True)
print(f"{N} QP shapes x {B} instances ({total}) in {time.time() - t0:.0f} s: {bad} with a mismatch" + (" (65 .. 128 variables)" if WIDE else ""))

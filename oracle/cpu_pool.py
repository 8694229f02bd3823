"""oracle/cpu_pool.py -- TEST INFRASTRUCTURE ONLY (imported by bench.py's cpu_baseline leg, never by opensot_amd).

The CPU reference path timed with ONE PROCESS PER WORKER instead of one thread per worker: qpOASES keeps process-global state
(its message handler, the allocator), so a thread sweep inside one process can stop scaling for reasons that are the harness',
not the host's.  Each worker is a fresh interpreter that loads its slice of the assembled sample, waits for a common start
time, solves its instances `cycles` times on one thread (hot-started across cycles, solve only: coman_ik.cpp:186-192) and
prints its own seconds; the parent adds the solves up over the longest worker's time.

    python -m oracle.cpu_pool <sample.pkl> <lo> <hi> <cycles> <backend> <start_at_unix_time>
"""
import os
import pickle
import subprocess
import sys
import time


def _worker(argv):
    path, lo, hi, cycles, backend, start_at = argv[0], int(argv[1]), int(argv[2]), int(argv[3]), int(argv[4]), float(argv[5])
    from oracle import pyoracle as po
    asm = pickle.load(open(path, "rb"))
    sl = slice(lo, hi)
    po.ihqp_solve_batch(asm, backend, nthreads=1, cycles=1, sl=sl)          # library loaded, pages touched
    while time.time() < start_at:
        time.sleep(0.001)
    t0 = time.time()
    r = po.ihqp_solve_batch(asm, backend, nthreads=1, cycles=cycles, sl=sl)
    print(f"POOL {hi - lo} {cycles} {r['seconds']:.6f} {int(r['status'].sum())} {t0:.4f} {time.time():.4f}", flush=True)


def run(asm, backend, workers, per_worker, cycles, root):
    """`workers` processes x `per_worker` instances x `cycles` cycles -> dict(solves_per_s, seconds, workers, ok)"""
    import tempfile
    B = asm["B"]
    per_worker = max(1, min(per_worker, B // workers))
    with tempfile.NamedTemporaryFile(suffix=".pkl", delete=False) as f:
        pickle.dump(asm, f, protocol=pickle.HIGHEST_PROTOCOL)
        path = f.name
    try:
        start_at = time.time() + 4.0 + 0.02 * workers          # every worker is loaded and waiting by then
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_pool", path, str(w * per_worker), str((w + 1) * per_worker),
                                   str(cycles), str(backend), f"{start_at:.4f}"], cwd=root, env=env, stdout=subprocess.PIPE, text=True)
                 for w in range(workers)]
        solves, ok, t_first, t_last, slowest = 0, 0, None, None, 0.0
        for p in procs:
            out, _ = p.communicate(timeout=600)
            for ln in out.splitlines():
                if ln.startswith("POOL "):
                    _, nb, cyc, sec, oks, t0, t1 = ln.split()
                    solves += int(nb) * int(cyc); ok += int(oks); slowest = max(slowest, float(sec))
                    t_first = float(t0) if t_first is None else min(t_first, float(t0))
                    t_last = float(t1) if t_last is None else max(t_last, float(t1))
        wall = (t_last - t_first) if t_first is not None else 0.0
        return {"workers": workers, "instances_per_worker": per_worker, "cycles": cycles, "solves_per_s": solves / wall if wall > 0 else 0.0,
                "per_worker": (solves / wall / workers) if wall > 0 else 0.0, "wall_seconds": wall, "slowest_worker_seconds": slowest,
                "ok_last_cycle": ok}
    finally:
        os.unlink(path)


if __name__ == "__main__":
    _worker(sys.argv[1:])

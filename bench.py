#!/usr/bin/env python3
"""bench.py -- whole-body QP solves/s of the MI355X hot path (BASELINE.json metric).

One "step" = one pass of the hot path over one resident batch: AutoStack::update (leaf -> b, W, box,
rows) + the 3-level iHQP cascade (H/g assembly + one QP per level) for every instance of the shard.
Workload at N=1: BASELINE.json configs[2], batch 4096 x 32-DoF, 3 levels (CoM / 4 Cartesian / Postural),
joint-limit and velocity-limit box, eps factor 1e6 (the reference benchmark's value, coman_ik.cpp:453).
N>1: weak scaling, 4096 instances per GPU, instances sharded contiguously over ranks (no data-path
collective: they are independent), one RCCL all-gather of the solved dq + status shards per step
(opensot_amd/parallel.py: the same ShardGather / ShardedCycle / timed_steps the gloo tests run).

After the timed region rank 0 also measures, OUTSIDE the headline: the other BASELINE configurations at their per-GPU
sizes and the kinematics producer (`other_configs`), the CPU reference path on the host cores (`cpu_baseline`, a thread
sweep), and the parity of the headline batch against qpOASES with per-instance KKT / lexicographic evidence (`parity`).

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FP64_SPEC_TFLOPS = 78.6     # MI355X FP64 vector = matrix peak (AMD public spec; not in MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
PMC_PROFILE = os.path.join("profiles", "r06_pmc_kernels.json")
FP64_PEAK_FILE = os.path.join("profiles", "r06_fp64_peak.json")   # tools/ubench_fp64_peak.hip on the GPU box ("verify with a micro-benchmark")
SIMDS, PEAK_CLOCK_HZ = 1024, 2.4e9     # 256 CUs x 4 SIMDs; MI355X_MICROARCH.md's throughput tables are quoted at 2.4 GHz


def fp64_peak():
    """(TFLOP/s, where it comes from): the MEASURED dependent-free v_fma_f64 / v_mfma_f64_16x16x4 rate of an MI355X
    (profiles/r06_fp64_peak.json, the committed output of tools/ubench_fp64_peak.hip), the better of the two -- the kernels use
    both units; AMD's public figure when no measurement is committed"""
    try:
        d = json.load(open(os.path.join(ROOT, FP64_PEAK_FILE)))
        v = max(float(d["v_fma_f64_tflops"]), float(d["v_mfma_f64_16x16x4_tflops"]))
        if 10.0 < v < 200.0:
            return v, (FP64_PEAK_FILE + f" (tools/ubench_fp64_peak.hip, measured on an MI355X: v_fma_f64 {d['v_fma_f64_tflops']} / "
                       f"v_mfma_f64_16x16x4 {d['v_mfma_f64_16x16x4_tflops']} TFLOP/s; AMD's public figure is {FP64_SPEC_TFLOPS})")
    except Exception:
        pass
    return FP64_SPEC_TFLOPS, "AMD public FP64 vector = matrix spec (no measurement committed under " + FP64_PEAK_FILE + ")"


FP64_PEAK_TFLOPS, FP64_PEAK_SOURCE = fp64_peak()


def algo_flops_per_solve(plan):
    """nominal flops of one cascade (SURVEY.md 8d formulas): per level symmetric H build m n (n+1),
    g 2 m n, Cholesky n^3/3, two triangular solves 2 n^2, Schur on c equality rows c n^2 + c^2 n + c^3/3
    + 2 c n + 2 c^2, plus 2 m_j n for each optimality right-hand side."""
    n = plan.n
    total = 0.0
    c = plan.nc
    for k in range(plan.L):
        m = plan.m(k)
        total += m * n * (n + 1) + 2 * m * n + n ** 3 / 3.0 + 2 * n * n
        total += c * n * n + c * c * n + c ** 3 / 3.0 + 2 * c * n + 2 * c * c
        if k + 1 < plan.L:
            total += 2 * m * n
        c += m
    return total


def algo_bytes_per_solve(plan):
    """compulsory HBM bytes of one cascade: the stored Jacobian rows of every level, b and diag(W), the
    stored constraint rows with their bounds, the box, dq out.  C3: (27*32 + 2*59 + 2*32 + 32) * 8 = 8624."""
    n = plan.n
    d = sum(plan.ma(k) * n + 2 * plan.m(k) for k in range(plan.L))
    d += plan.nc_stored * n + 2 * plan.nc
    d += (2 * n if plan.bounds else 0) + n
    return 8 * d


def leaf_bytes_per_instance(dev_leaf):
    """bytes of leaf inputs one instance's AutoStack::update reads (poses, q, references, limits ...)"""
    tot = 0

    def walk(o):
        nonlocal tot
        if o is None:
            return
        if torch.is_tensor(o):
            tot += o.element_size() * o[0].numel() if o.dim() > 0 and o.shape[0] > 0 else 0
        elif isinstance(o, (list, tuple)):
            for x in o:
                walk(x)

    for key in ("task", "W", "bound", "rows", "reg"):
        walk(dev_leaf.get(key))
    return tot


def assembled_bytes_per_solve(plan):
    """bytes of the assembled arrays the update half writes: b and diag(W) of every level, the merged box"""
    return 8 * (sum(2 * plan.m(k) for k in range(plan.L)) + (2 * plan.n if plan.bounds else 0) + 2 * plan.nc)


def kernel_source_sha():
    """identifies the cascade kernel a PMC traffic figure belongs to (profiles/*.json carry the same hash)"""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "opensot_amd", "csrc")
    for f in sorted(x for x in os.listdir(csrc) if x.endswith(".h")):     # every kernel header (the bench line quotes them all)
        # the CODE of the header: // comments (whole-line and trailing) and blank lines do not change a kernel
        for line in open(os.path.join(csrc, f), "r", encoding="utf-8", errors="replace"):
            code = line.split("//", 1)[0].rstrip()
            if code.strip():
                h.update(code.encode() + b"\n")
    return h.hexdigest()[:16]


def pmc_traffic(kernels):
    """HBM bytes of one STEP from the committed rocprofv3 PMC passes (tools/profile_round.sh -> profiles/r05_pmc_kernels.json):
    `kernels` = [(substring of the kernel name, workgroups of the launch, launches per step)]; the per-launch FETCH / WRITE
    figures of every (kernel, grid) pair are summed.  Only if the passes were taken on THIS kernel source (hash) and every
    pair is in the table; otherwise (None, None): a stale figure is worse than none."""
    if not kernels:
        return None, None
    try:
        pm = json.load(open(os.path.join(ROOT, PMC_PROFILE)))
        if pm.get("kernel_source_sha") != kernel_source_sha():
            return None, None
        total = 0.0
        for name, wgs, count in kernels:
            rows = [r for r in pm["kernels"] if name in r["kernel"] and r["workgroups"] == wgs and "hbm_bytes_per_launch_corrected" in r]
            if not rows:
                return None, None
            total += count * max(rows, key=lambda r: r["launches"])["hbm_bytes_per_launch_corrected"]
        return total, PMC_PROFILE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 per MI355X_MICROARCH.md; same kernel source hash)"
    except Exception:
        return None, None


def pmc_mfma_busy(kernels, step_ms):
    """north_star's "MFMA utilisation against gfx950 peak" of one STEP: SQ_VALU_MFMA_BUSY_CYCLES of the step's launches (the committed
    PMC passes, same kernel-source hash as pmc_traffic; the counter counts CYCLES, MI355X_MICROARCH.md) over the cycles the chip's 1024
    matrix cores have in the step at the peak clock.  None without matching passes."""
    if not kernels or not step_ms:
        return None
    try:
        pm = json.load(open(os.path.join(ROOT, PMC_PROFILE)))
        if pm.get("kernel_source_sha") != kernel_source_sha():
            return None
        busy = 0.0
        for name, wgs, count in kernels:
            rows = [r for r in pm["kernels"] if name in r["kernel"] and r["workgroups"] == wgs and "SQ_VALU_MFMA_BUSY_CYCLES" in r.get("per_launch", {})]
            if not rows:
                return None
            busy += count * max(rows, key=lambda r: r["launches"])["per_launch"]["SQ_VALU_MFMA_BUSY_CYCLES"]
        return busy / (step_ms * 1e-3 * PEAK_CLOCK_HZ * SIMDS)
    except Exception:
        return None


def hbm_roofline(bytes_per_instance, B, step_ms, kernels, kernel_label):
    """roofline block of a path that is far from both roofs (nHQP, eHQP, ADMM, kinematics): bound "hbm", achieved = the
    path's ALGORITHMIC bytes per step / step time, traffic = PMC bytes of the step's kernels"""
    gbs = B * bytes_per_instance / (step_ms * 1e-3) / 1e9
    traffic, src = pmc_traffic(kernels)
    return {"bound": "hbm", "kernel": kernel_label, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": src or "no PMC passes committed for this kernel source: null rather than a stale figure",
            "algorithmic_bytes_per_instance": bytes_per_instance, "step_ms": step_ms}


def roofline_of(plan, Bl, kern_ms, launches, kernel_name, traffic=None, traffic_source=None):
    flops, nbytes = algo_flops_per_solve(plan), algo_bytes_per_solve(plan)
    ks = kern_ms * 1e-3
    tf = Bl * flops / ks / 1e12
    gbs = Bl * nbytes / ks / 1e9
    return ({"bound": "mfma", "kernel": kernel_name, "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
             "frac": tf / FP64_PEAK_TFLOPS, "traffic": traffic, "avg_launch_ms": kern_ms, "launches": launches,
             "algorithmic_flops_per_solve": flops, "peak_source": FP64_PEAK_SOURCE,
             "peak_spec": FP64_SPEC_TFLOPS, "frac_of_spec": tf / FP64_SPEC_TFLOPS,
             "note": "fp64 FMA roof (nominal algorithmic flops; AI ~ 20 flop/B > machine balance ~ 10)"},
            {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
             "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_solve": nbytes})


# ------------------------------------------------------------------------------------------------------------------
# CPU reference path (oracle/_ref = the reference's own qpOASES 3.1, restated cascade around it), on this host
# ------------------------------------------------------------------------------------------------------------------
def one_rate_hint(sweep):
    """solves/s of one thread as the thread sweep measured it (sizes the process sweep's cycle count)"""
    return max(1.0, sweep[0]["solves_per_s"]) if sweep else 1.0e4


def cpu_baseline(plan, leaf_sample, budget_s=12.0, process_sweep=False):
    """Timed like examples/cpp/coman_ik.cpp:186-192 (update excluded, solve only), hot-started across cycles as the
    reference does.  A thread sweep {1, 16, 64, all usable cores}: each point solves the same sample for ~budget/5 s.
    `value` is the best point of the sweep; `single_thread` the one-thread figure."""
    from oracle import pyoracle as po
    asm = po.assemble(plan, leaf_sample)
    B = asm["B"]
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    kind, be = ("reference", po.BE_QPOASES_REF) if po.ref_available() else ("port", po.BE_EIQP_EQ)
    try:
        po.ihqp_solve_batch(asm, be, nthreads=1, cycles=1, sl=slice(0, 8))
    except Exception:
        kind, be = "port", po.BE_EIQP_EQ
    # (rounds 3-4 settled that this host scales to ~16x one thread whatever the harness: the 64-thread and all-cores points and the
    #  one-process-per-worker sweep are behind --cpu-sweep; the default run pays for 1 and 16 threads only)
    counts = sorted({1, min(16, usable), min(64, usable), usable}) if process_sweep else sorted({1, min(16, usable)})
    per_point = budget_s / (len(counts) + 1)
    # one thread, ONE instance, cache-hot: what one robot on one core sees (the published 0.2333 ms/solve is this mode)
    r = po.ihqp_solve_batch(asm, be, nthreads=1, cycles=50, sl=slice(0, 1))
    cyc = int(max(50, min(200000, per_point / max(r["seconds"] / 50, 1e-7))))
    r = po.ihqp_solve_batch(asm, be, nthreads=1, cycles=cyc, sl=slice(0, 1))
    hot = cyc / r["seconds"]
    hot_seconds = r["seconds"]
    sweep = []
    for nt in counts:
        nb = min(B, max(nt * 4, 64))                 # a few instances per thread, the whole sample for many threads
        sl = slice(0, nb)
        r = po.ihqp_solve_batch(asm, be, nthreads=nt, cycles=2, sl=sl)
        cyc = int(max(2, min(20000, per_point / max(r["seconds"] / 2, 1e-6))))
        r = po.ihqp_solve_batch(asm, be, nthreads=nt, cycles=cyc, sl=sl)
        sweep.append({"threads": nt, "instances": nb, "cycles": cyc, "solves_per_s": nb * cyc / r["seconds"],
                      "per_thread": nb * cyc / r["seconds"] / nt, "seconds": r["seconds"], "ok": int(r["status"].sum())})
    # the same with ONE PROCESS per worker (oracle/cpu_pool.py): shows whether the thread sweep's ceiling is the harness
    # (qpOASES' process-global message handler, the allocator) or the host
    procs = []
    try:
        if not process_sweep:
            raise StopIteration
        from oracle import cpu_pool
        slim = {k: v for k, v in asm.items()}
        for nw in sorted({min(16, usable), min(64, usable), usable}):
            if nw < 2:
                continue
            per = max(1, min(4, B // nw))
            cyc = int(max(2, min(20000, (budget_s / 6.0) * one_rate_hint(sweep) / per)))
            procs.append(cpu_pool.run(slim, be, nw, per, cyc, ROOT))
    except StopIteration:
        pass
    except Exception as e:
        procs.append({"error": str(e)[:200]})
    best = max(sweep, key=lambda s: s["solves_per_s"])
    one = sweep[0]["solves_per_s"]
    pbest = max((p_ for p_ in procs if "solves_per_s" in p_), key=lambda p_: p_["solves_per_s"], default=None)
    note = ""
    if best["per_thread"] < 0.5 * one:
        note = (f"; per-thread rate at {best['threads']} threads is {best['per_thread'] / one:.2f} of the 1-thread rate: the host "
                f"exposes {usable} logical CPUs to this process but the sweep scales only to ~{best['solves_per_s'] / one:.0f}x one "
                "thread (SMT siblings / container CPU share), and every instance keeps three hot-started qpOASES objects "
                "(~0.6 MB) that fall out of the private caches when a thread cycles over many instances")
    value, cores, how = best["solves_per_s"], best["threads"], f"best point of the thread sweep ({best['threads']} threads)"
    if pbest is not None and pbest["solves_per_s"] > value:
        value, cores, how = pbest["solves_per_s"], pbest["workers"], f"best point of the process sweep ({pbest['workers']} single-thread processes)"
    return {"value": value, "unit": "solves/s", "cores": cores, "kind": kind, "value_is": how,
            "usable_logical_cpus": usable, "single_thread": one, "single_thread_one_instance_cache_hot": hot,
            "sweep": sweep, "process_sweep": procs,
            "sample": f"the same C3 stack, solve only (coman_ik.cpp:186-192 protocol), "
                      f"{'qpOASES 3.1 hot-started across cycles' if kind == 'reference' else 'C Goldfarb-Idnani port'}; thread sweep "
                      f"{counts}, {sum(s['seconds'] for s in sweep) + hot_seconds:.1f} s of wall clock = "
                      f"{sum(s['seconds'] * s['threads'] for s in sweep):.0f} thread-seconds of CPU work in all; value = best point "
                      f"({best['threads']} threads x {best['instances']} instances x {best['cycles']} cycles)" + note}


# ------------------------------------------------------------------------------------------------------------------
# parity of the headline batch against the reference's qpOASES, with evidence
# ------------------------------------------------------------------------------------------------------------------
def parity_report(plan, leaf_sample, dq_dev, xl_dev, slack_dev, tol=1e-6, max_evidence=8):
    """BASELINE's second metric: max_i |dq_i - dq_ref,i| against qpOASES 3.1 AT THE REFERENCE'S OWN OPTIONS.  Every instance
    farther than `tol` gets evidence (oracle/lexcheck.py): per-level constraint violation, KKT residual and most negative
    multiplier of BOTH chains in the QP iHQP.cpp:263-358 poses, the lexicographic cost vector of both final points and the
    verdict -- so the line itself shows which point is the optimum of the reference's problem."""
    from oracle import lexcheck as lc
    from oracle import pyoracle as po
    asm = po.assemble(plan, leaf_sample)
    nt = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if not po.ref_available():
        r = po.ihqp_solve_batch(asm, po.BE_EIQP_EQ, nthreads=nt)
        e = np.where(r["status"] == 1, np.abs(dq_dev - r["dq"]).max(axis=1), np.inf)
        fin = np.isfinite(e)
        return {"reference": "C Goldfarb-Idnani port (oracle/_ref not built)", "tolerance": tol, "instances": int(e.size),
                "instances_compared": int(fin.sum()), "max_abs_dq_diff": float(e[fin].max()) if fin.any() else None,
                "within_tolerance": int((e <= tol).sum())}
    rd = po.ihqp_solve_batch(asm, po.BE_QPOASES_REF, nthreads=nt)
    e = np.where(rd["status"] == 1, np.abs(dq_dev - rd["dq"]).max(axis=1), np.inf)
    fin = np.isfinite(e)
    out = {"reference": "qpOASES 3.1 (oracle/_ref, compiled from the reference's vendored sources) at OpenSoT's option set "
                        "(QPOasesBackEnd.cpp:51-76), restated iHQP cascade around it",
           "tolerance": tol, "instances": int(e.size), "instances_compared": int(fin.sum()),
           "max_abs_dq_diff": float(e[fin].max()) if fin.any() else None,
           "median_abs_dq_diff": float(np.median(e[fin])) if fin.any() else None,
           "within_tolerance": int((e <= tol).sum()),
           "device_accepted_slack_max": float(slack_dev.max())}
    far = np.nonzero(fin & (e > tol))[0]
    if far.size:
        # the same instances with qpOASES run to its exact optimum (terminationTolerance 10 EPS instead of OpenSoT's 2.2e-7)
        sub = {"B": int(far.size)}
        asm_f = dict(asm)
        for key in ("A", "b", "w", "c"):
            asm_f[key] = [None if a is None else a[far] for a in asm[key]]
        for key in ("C", "lo", "up", "l", "u"):
            asm_f[key] = None if asm[key] is None else asm[key][far]
        asm_f["B"] = int(far.size)
        asm_f.pop("Wdense", None)
        rx = po.ihqp_solve_batch(asm_f, po.BE_QPOASES_REF, nthreads=min(nt, int(far.size)), termination_tolerance=10 * 2.221e-16)
        re_ = po.ihqp_solve_batch(asm_f, po.BE_EIQP_EQ, nthreads=min(nt, int(far.size)))   # independent exact active-set method
        ev = []
        better = {"device": 0, "qpOASES": 0, "tie": 0}
        for j, i in enumerate(far):
            rec = lc.instance_evidence(asm, int(i), xl_dev[i], rd["x_levels"][i], names=("device", "qpOASES"))
            rec["device_vs_qpOASES_run_to_exact_optimum"] = (float(np.abs(dq_dev[i] - rx["dq"][j]).max())
                                                             if rx["status"][j] == 1 else None)
            rec["device_vs_eiQuadProg_restatement"] = (float(np.abs(dq_dev[i] - re_["dq"][j]).max())
                                                       if re_["status"][j] == 1 else None)
            better[rec["lexicographically_better"]] += 1
            if len(ev) < max_evidence:
                ev.append(rec)
        out["beyond_tolerance"] = {"count": int(far.size), "lexicographically_better": better, "evidence": ev,
                                   "reading": "per level: viol = constraint violation of x_k in level k's QP (box, rows, "
                                              "optimality equalities of the chain itself), kkt = stationarity residual with "
                                              "NON-NEGATIVE least-squares multipliers on the active set (inequalities >= 0, "
                                              "equalities free), so min_multiplier is 0 by construction and a point that is not a "
                                              "KKT point shows as a residual; kkt_certificate_of_dq = the same for the final point "
                                              "alone, every level's optimality rows posed at dq; lex_cost_of_dq = the task cost of "
                                              "every level at the final point: the lexicographically smaller FEASIBLE point is "
                                              "the optimum of iHQP.cpp:263-358's problem"}
    return out


# ------------------------------------------------------------------------------------------------------------------
# the other BASELINE configurations and the kinematics producer (after the timed region, rank 0, outside the headline)
# ------------------------------------------------------------------------------------------------------------------
def time_config(name, B, device, steps=20, warmup=8, cycles=4, drift=0.01, lanes=1, streams=None):
    """a BASELINE configuration at its per-GPU size under the HEADLINE'S PROTOCOL: the steps rotate through `cycles` temporally
    coherent control cycles (every input, Jacobians included, moved by `drift` from one to the next), one fused update + cascade
    launch per step, longest-first dispatch from the previous cycles' iteration counts (so the order is a prediction, never a
    replay of the same cycle).  lanes > 1 (with the caller's streams): the headline's SUBMISSION as well -- the batch as sub-batches
    on their own streams with no join between steps, the timed steps of a sub-batch as one HIP graph
    (opensot_amd.parallel.PipelinedCycle); plain launches if the capture fails."""
    from opensot_amd import synth
    from opensot_amd.parallel import PipelinedCycle, ShardedCycle, lane_ranges
    from opensot_amd.solver import BatchedStack
    if lanes > 1 and streams is not None and len(streams) >= lanes and steps % 4 == 0:
        return _time_config_lanes(name, B, device, steps, warmup, cycles, drift, lanes, streams)
    if name == "C5":
        plan, leaf = synth.make_id_stack(B, seed=5000)
    else:
        plan, leaf = synth.make_velocity_stack(name, B, seed={"C2": 2000, "C3": 3000, "C4": 4000}[name])
    rng = np.random.default_rng(77)
    leaves = [leaf]
    for _ in range(cycles - 1):
        leaves.append(synth.perturb(leaves[-1], rng, drift))
    st = BatchedStack(plan, B, device=device, want_levels=False)
    devs, A_sets = [], []
    for lf in leaves:
        st.A = [None if t is None else torch.empty_like(t) for t in st.A]
        devs.append(st.load_leaf(lf)); A_sets.append(st.A)
    loop = ShardedCycle(st, devs, A_sets, B, None)
    for _ in range(warmup):
        loop.step()
    torch.cuda.synchronize()
    st.set_timing(True, stride=4)     # (every fourth launch bracketed by HIP events: the brackets themselves cost ~10 % at config 2)
    t0 = time.perf_counter()
    for _ in range(steps):
        loop.step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    kern_ms, launches = st.kernel_time_ms()
    st.set_timing(False)
    ok = int((st.status[:B] == 0).sum().item())
    NPk = 32 if plan.n <= 32 else (40 if plan.n <= 38 else (56 if plan.n <= 54 else 64))   # the cascade instantiation make_dev_plan picks (osot_host_plan.h)
    box = NPk == 32 and plan.nc == 0     # plans without constraint rows run the BOX instantiation (osot_solver_set_specialisation)
    kname = f"osot_cycle_kernel<{NPk}, false{', true' if box else (', false' if NPk == 32 else '')}>"
    traffic, src = pmc_traffic([(kname, B + 1, 1)])
    rf, rh = roofline_of(plan, B, kern_ms, launches, kname + (" (BOX instantiation: bounds are the only inequalities)" if box else ""), traffic,
                         src or "no PMC passes committed for this kernel source: null rather than a stale figure")
    return {"workload": {"C2": "BASELINE configs[1]: 1-level Cartesian + Postural (soft priority), joint-limit box",
                         "C3": "BASELINE configs[2]", "C4": "BASELINE configs[3] shard: C3 + 16 self-collision rows",
                         "C5": "BASELINE configs[4] shard: 38-DoF floating-base inverse dynamics, x = [qddot; 4 x 3 forces] (n = 50), "
                               "102 constraint rows (dynamic feasibility, friction cones, torque limits, acceleration joint limits)"}[name],
            "batch": B, "n": plan.n, "rows_per_level": [plan.m(k) for k in range(plan.L)], "constraint_rows": plan.nc,
            "value": B * steps / el, "unit": "solves/s", "ms_per_step": 1e3 * el / steps, "steps": steps,
            "solved_ok": f"{ok}/{B}",
            "protocol": f"the headline's: steps rotate through {cycles} temporally coherent cycles ({100 * drift:.0f} % drift of every input per "
                        "cycle), one fused update + cascade launch per step, cold start, longest-first dispatch predicted from the previous cycles",
            "roofline": rf, "roofline_hbm": rh}


def _time_config_lanes(name, B, device, steps, warmup, cycles, drift, lanes, streams, hot=False):
    """time_config with the headline's submission (see there); hot: osot_solver_set_hotstart on every sub-batch"""
    from opensot_amd import synth
    from opensot_amd.parallel import PipelinedCycle, ShardedCycle, lane_ranges
    from opensot_amd.solver import BatchedStack
    if name == "C5":
        plan, leaf = synth.make_id_stack(B, seed=5000)
    else:
        plan, leaf = synth.make_velocity_stack(name, B, seed={"C2": 2000, "C3": 3000, "C4": 4000}[name])
    rng = np.random.default_rng(77)
    leaves = [leaf]
    for _ in range(cycles - 1):
        leaves.append(synth.perturb(leaves[-1], rng, drift))
    spans = lane_ranges(B, lanes)
    stacks, ls = [], []
    for a, b in spans:
        stj = BatchedStack(plan, b - a, device=device, want_levels=False)
        if hot:
            stj.set_hotstart(True)
        devs, A_sets = [], []
        for lf in leaves:
            stj.A = [None if t is None else torch.empty_like(t) for t in stj.A]
            devs.append(stj.load_leaf(sub_leaf(lf, a, b))); A_sets.append(stj.A)
        stacks.append(stj)
        ls.append(ShardedCycle(stj, devs, A_sets, b - a, None))
    cyc = PipelinedCycle(ls, list(streams[:lanes]))
    for _ in range(warmup):
        cyc.step()
    torch.cuda.synchronize()
    note, graphed = None, False
    try:
        cyc.capture(steps)
        cyc.replay(); torch.cuda.synchronize()
        graphed = True
    except Exception as e:
        note = f"graph capture unavailable: {e}"[:200]
    t0 = time.perf_counter()
    if graphed:
        cyc.replay()
    else:
        for _ in range(steps):
            cyc.step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ok = sum(int((stj.status[:b - a] == 0).sum().item()) for stj, (a, b) in zip(stacks, spans))
    NPk = 32 if plan.n <= 32 else (40 if plan.n <= 38 else (56 if plan.n <= 54 else 64))
    box = NPk == 32 and plan.nc == 0
    kname = f"osot_cycle_kernel<{NPk}, false{', true' if box else ', false'}>"
    traffic, src = pmc_traffic([(kname, (B // lanes) + 1, lanes)])
    rf, rh = roofline_of(plan, B, 1e3 * el / steps, steps * lanes, kname + " (whole step time as the divisor)", traffic,
                         src or "no PMC passes committed for this kernel source: null rather than a stale figure")
    return {"workload": {"C2": "BASELINE configs[1]: 1-level Cartesian + Postural (soft priority), joint-limit box",
                         "C3": "BASELINE configs[2]", "C4": "BASELINE configs[3] shard: C3 + 16 self-collision rows",
                         "C5": "BASELINE configs[4] shard: 38-DoF floating-base inverse dynamics, x = [qddot; 4 x 3 forces] (n = 50), "
                               "102 constraint rows (dynamic feasibility, friction cones, torque limits, acceleration joint limits)"}[name],
            "batch": B, "lanes": lanes, "n": plan.n, "rows_per_level": [plan.m(k) for k in range(plan.L)], "constraint_rows": plan.nc,
            "value": B * steps / el, "unit": "solves/s", "ms_per_step": 1e3 * el / steps, "steps": steps,
            "solved_ok": f"{ok}/{B}", "note_capture": note,
            "protocol": f"the headline's: steps rotate through {cycles} temporally coherent cycles ({100 * drift:.0f} % drift of every input per "
                        f"cycle), one fused update + cascade launch per step, cold start, longest-first dispatch predicted from the previous cycles; {lanes} "
                        "sub-batches on their own streams with no join between steps" + (f", the {steps} timed steps of a sub-batch as one HIP graph" if graphed else ", plain launches"),
            "roofline": rf, "roofline_hbm": rh}


# the reference's own published workloads (examples/cpp/coman_ik.cpp:425-449): COMAN, 35 coordinates, stacks S1..S4
# with the feet as TaskToConstraint equality rows, joint-limit and velocity-limit box, eps factor 1e6, dT = 0.01.  Published there: mean
# ms of ONE solver->solve(dq) on one Ryzen 9 4900HS core (update and model.update excluded).
COMAN_REFERENCE_MS = {"S1": {"iHQP_qpOASES": 0.0813, "nHQP_qpOASES": 0.2969}, "S2": {"iHQP_qpOASES": 0.1586, "nHQP_qpOASES": 0.2637},
                      "S3": {"iHQP_qpOASES": 0.2333, "nHQP_qpOASES": 0.3191}, "S4": {"iHQP_qpOASES": 0.3008, "nHQP_qpOASES": 0.3721}}


def coman_stack(which, n):
    """the four stacks of examples/cpp/coman_ik.cpp:425-449 as plans: frames 0..3 = l_wrist, r_wrist, l_sole, r_sole"""
    from opensot_amd import abi
    from opensot_amd.plan import Bound, Rows, StackPlan, Task, eps_abs_from_factor
    lw = lambda: Task(abi.TASK_CARTESIAN, 6, weight=0.1, lam=0.1, name="l_wrist")
    lw1 = lambda: Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="l_wrist")
    rw = lambda: Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="r_wrist")
    com = lambda: Task(abi.TASK_COM, 3, lam=0.1, name="com")
    post = lambda w=1.0: Task(abi.TASK_POSTURAL, n, weight=w, lam=0.01, name="postural")
    levels = {"S1": [[lw(), rw(), com(), post(1e-4)]],
              "S2": [[com(), lw(), rw()], [post()]],
              "S3": [[com()], [lw(), rw()], [post()]],
              "S4": [[com()], [lw1()], [rw()], [post()]]}[which]
    bounds = [Bound(abi.BOUND_JOINT_LIMITS, scaling=1.0, name="jl"), Bound(abi.BOUND_VELOCITY_LIMITS, dT=0.01, name="vl")]
    rows = [Rows(abi.ROWS_TASK_CARTESIAN, 6, lam=0.1, name="l_sole"), Rows(abi.ROWS_TASK_CARTESIAN, 6, lam=0.1, name="r_sole")]
    return StackPlan(n=n, levels=levels, bounds=bounds, rowblocks=rows, eps_abs=eps_abs_from_factor(1e6))


def time_coman35(which, B, device, steps=20, warmup=5, front_end="iHQP", specialise=True, fused=True, lanes=1, streams=None, graph=True, hot=False):
    """the reference example's control loop (coman_ik.cpp:174-219) for B robots, everything resident: q -> frame poses, Jacobians,
    CoM (rows written straight into A_k / C) -> AutoStack::update + Solver::solve -> q += dq.  Each robot chases its own random
    wrist goals (+-0.2 m, as the reference's harness draws them), so the inputs of consecutive steps are the closed loop's own
    drift.  iHQP: submitted like the headline -- the batch as `lanes` sub-batches on their own streams, ONE launch per step and
    lane (osot_control_cycle; fused=False: the kinematics launch, the update + cascade launch, the integration), the steps of a
    lane as one HIP graph.  nHQP: kinematics launch, update launch, osot_nhqp_solve (three launches per level), integration, in
    one stream."""
    from opensot_amd import kinematics as kin
    from opensot_amd.parallel import lane_ranges
    from opensot_amd.solver import BatchedStack
    m, lo, up = kin.from_json(os.path.join(ROOT, "tests", "golden", "coman_tree.json"))
    n = m.n
    plan = coman_stack(which, n)
    dev = torch.device("cuda", device)
    f64 = dict(dtype=torch.float64, device=dev)
    rng = np.random.default_rng(35)
    K = kin.Kinematics(m, device=device)
    if front_end != "iHQP":
        graph, fused = False, False          # (the front-end's host loop launches ten kernels per step: sub-batches, no graphs)
        if streams is None or len(streams) < lanes:
            lanes = 1
    work = []
    for a, b in lane_ranges(B, lanes):
        Bl = b - a
        q0 = np.zeros((Bl, n))
        for s_ in "RL":                                   # a slightly crouched, arms-bent posture inside the limits
            q0[:, m.names.index(s_ + "HipSag")] = -0.3; q0[:, m.names.index(s_ + "KneeSag")] = 0.6
            q0[:, m.names.index(s_ + "AnkSag")] = -0.3; q0[:, m.names.index(s_ + "Elbj")] = -0.8
            q0[:, m.names.index(s_ + "ShSag")] = 0.2
        q0[:, m.names.index("LShLat")] = 0.3; q0[:, m.names.index("RShLat")] = -0.3
        q0[:, 6:] += rng.normal(0.0, 0.02, (Bl, n - 6))
        q0 = np.clip(q0, np.maximum(lo, -10.0) + 1e-3, np.minimum(up, 10.0) - 1e-3)
        st = BatchedStack(plan, Bl, device=device, want_levels=False)
        if not specialise:
            st.set_specialisation(False)
        if hot:
            st.set_hotstart(True)       # (every level's working set of the robot's previous control cycle: QPOasesBackEnd.cpp:258-285)
        # (the caller's streams where it has them: fresh ones can land on a hardware queue another lane already uses)
        stream = streams[len(work)] if (streams is not None and len(work) < len(streams)) else torch.cuda.Stream(device=dev)
        st.stream = stream
        q = torch.as_tensor(q0, **f64).contiguous()
        pose = [torch.zeros((Bl, 12), **f64) for _ in range(4)]
        com = torch.zeros((Bl, 3), **f64)
        # where each task's Jacobian rows live: (level tensor, first row)
        where, off = {}, [0] * plan.L
        for k, lev in enumerate(plan.levels):
            for t in lev:
                if t.name in ("l_wrist", "r_wrist", "com"):
                    where[t.name] = (st.A[k], off[k])
                if not t.implicit:
                    off[k] += t.rows
        kw = dict(frame_pose={f: pose[f] for f in range(4)}, frame_J={0: where["l_wrist"], 1: where["r_wrist"], 2: (st.C, 0), 3: (st.C, 6)},
                  com=com, com_J=where["com"])
        with torch.cuda.stream(stream):
            K.forward(q, **kw)
        torch.cuda.synchronize()
        pose_d = [p.clone() for p in pose]
        for f in (0, 1):
            pose_d[f][:, 9:] += torch.as_tensor(rng.uniform(-0.2, 0.2, (Bl, 3)), **f64)
        big = 1.0e3
        qmin = torch.as_tensor(np.tile(np.maximum(lo, -big), (Bl, 1)), **f64); qmax = torch.as_tensor(np.tile(np.minimum(up, big), (Bl, 1)), **f64)
        leaf_of = {"l_wrist": (pose[0], pose_d[0], None), "r_wrist": (pose[1], pose_d[1], None), "com": (com, com.clone(), None), "postural": (q, q.clone(), None)}
        leaf = {"B": Bl, "task": [[leaf_of[t.name] for t in lev] for lev in plan.levels],
                "bound": [(q, qmin, qmax), (torch.full((Bl, n), 2.0, **f64), None, None)], "rows": [(pose[2], pose_d[2], None), (pose[3], pose_d[3], None)]}
        kb = K.batch_args(q, **kw)

        def step(st=st, leaf=leaf, q=q, Bl=Bl, stream=stream, kb=kb, kw=kw):
            with torch.cuda.stream(stream):
                if front_end == "nHQP":
                    K.forward(q, **kw); st.update(leaf); st.solve_nhqp(Bl); q.add_(st.dq[:Bl])
                elif fused:
                    st.control_cycle(K, kb, leaf, q_integrate=q)
                else:
                    K.forward(q, **kw); st.cycle(leaf, cached=True); q.add_(st.dq[:Bl])
        work.append((st, step, stream, Bl))
    for _ in range(warmup):
        for _, step, _, _ in work:
            step()
    torch.cuda.synchronize()
    graphs, note = [], None
    if graph:
        try:
            for st, step, stream, _ in work:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream):
                    for _ in range(steps):
                        step()
                graphs.append(g)
        except Exception as e:
            graphs, note = [], f"graph capture unavailable: {e}"[:200]
            for st, _, _, _ in work:
                st.set_schedule(True)
    torch.cuda.synchronize()

    def run_all():
        if graphs:
            for g, (_, _, stream, _) in zip(graphs, work):
                with torch.cuda.stream(stream):
                    g.replay()
        else:
            for _ in range(steps):
                for _, step, _, _ in work:
                    step()
    if graphs:
        run_all(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_all()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    ok = sum(int((st.status[:Bl] == 0).sum().item()) for st, _, _, Bl in work)
    NPk = 32 if plan.n <= 32 else (40 if plan.n <= 38 else (56 if plan.n <= 54 else 64))
    how = (f"kinematics launch, update launch, osot_nhqp_solve, integration; {lanes} sub-batch(es) on their own stream(s)" if front_end == "nHQP" else
           (f"{lanes} sub-batches on their own streams, " + ("ONE launch per step and sub-batch (osot_control_cycle)" if fused else
            "kinematics launch + update-and-cascade launch + integration per step") + (f", {steps} steps of a sub-batch per HIP graph" if graphs else ", plain launches")))
    out = {"workload": f"the reference's published workload {which} (examples/cpp/coman_ik.cpp:425-449): COMAN, 35 coordinates, "
                       + {"S1": "(0.1 l_wrist + r_wrist + com + 1e-4 postural)", "S2": "((com + 0.1 l_wrist + r_wrist) / postural)",
                          "S3": "(com / (0.1 l_wrist + r_wrist) / postural)", "S4": "(com / l_wrist / r_wrist / postural)"}[which]
                       + " << joint_limits << vel_limits << (l_sole + r_sole as 12 TaskToConstraint equality rows), eps factor 1e6; closed loop of "
                         f"{B} robots on the device: kinematics -> update + {front_end} solve -> q += dq ({how}), each robot chasing its own +-0.2 m wrist goals",
           "front_end": front_end, "batch": B, "lanes": lanes, "n": n, "rows_per_level": [plan.m(k) for k in range(plan.L)], "constraint_rows": plan.nc,
           "value": B / (ms * 1e-3), "unit": "solves/s", "ms_per_step": ms, "steps": steps, "solved_ok": f"{ok}/{B}", "note_capture": note,
           "ms_per_solve_per_instance_stream": ms,
           "reference_published_ms_per_solve": dict(COMAN_REFERENCE_MS[which], hardware="one core of a Ryzen 9 4900HS; solve only, update and model.update excluded (BASELINE.md section 1)"),
           "note": "ms_per_step is the time per control cycle INCLUDING the kinematics producer and AutoStack::update, for all "
                   f"{B} robots at once; the reference's figure is one robot's solve alone"}
    if front_end == "iHQP":
        kname = (f"osot_control_cycle_kernel<{NPk}, false, {'true' if specialise else 'false'}>" if fused else
                 f"osot_cycle_kernel<{NPk}, false, {'true' if specialise else 'false'}>")
        # (no PMC traffic here: S1 .. S4 run the same instantiation on the same grid, and the committed PMC table is keyed by
        #  (kernel, grid) -- it cannot tell the four stacks apart)
        rf, rh = roofline_of(plan, B, ms, steps * lanes, kname + " (whole step time as the divisor)", None,
                             "the four COMAN stacks share one kernel instantiation and grid: the (kernel, grid)-keyed PMC table cannot tell them apart")
        out["roofline"], out["roofline_hbm"] = rf, rh
    else:
        out["roofline"] = hbm_roofline(algo_bytes_per_solve(plan), B, ms, [], "nHQP front-end kernels (see nHQP_C3)")
    return out


def time_full_cycle(B, device, lanes=2, steps=40, warmup=8, streams=None, fused=True, solve_only=False, rollout=1):
    """q -> kinematics -> AutoStack::update + cascade -> q += dq for the 32-DoF humanoid under BASELINE config 3's stack (CoM / l_wrist(0.1)
    + r_wrist + l_sole + r_sole / Postural, joint-limit and velocity-limit box), everything resident, submitted like the headline: the
    batch as `lanes` sub-batches on their own streams, the steps of a lane as ONE HIP graph.  fused: ONE launch per step
    (osot_control_cycle: the instance's kinematics, update, cascade and integration by the same wavefront); otherwise three (the
    kinematics launch, the fused update + cascade launch, the integration of q).  The inputs of consecutive steps are the closed
    loop's own drift (every robot chases its own wrist goals); nothing is replayed from a recorded cycle.
    rollout = K > 1 (round 5, fused only): K control cycles of every robot per launch (osot_control_rollout: the reference's loop
    run by the robot's own wavefront); `steps` stays the number of control cycles timed, in steps / K launches per sub-batch."""
    from opensot_amd import abi
    from opensot_amd import kinematics as kin
    from opensot_amd.parallel import lane_ranges
    from opensot_amd.plan import Bound, StackPlan, Task, eps_abs_from_factor
    from opensot_amd.solver import BatchedStack
    rollout = max(1, int(rollout)) if (fused and not solve_only) else 1
    if steps % rollout:
        steps += rollout - steps % rollout
    calls = steps // rollout
    m = kin.humanoid32()
    n = m.n
    levels = [[Task(abi.TASK_COM, 3, lam=0.1, name="com")],
              [Task(abi.TASK_CARTESIAN, 6, weight=0.1, lam=0.1, name="l_wrist"), Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="r_wrist"),
               Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="l_sole"), Task(abi.TASK_CARTESIAN, 6, lam=0.1, name="r_sole")],
              [Task(abi.TASK_POSTURAL, n, lam=0.01, name="postural")]]
    bounds = [Bound(abi.BOUND_JOINT_LIMITS, scaling=1.0, name="jl"), Bound(abi.BOUND_VELOCITY_LIMITS, dT=0.01, name="vl")]
    plan = StackPlan(n=n, levels=levels, bounds=bounds, rowblocks=[], eps_abs=eps_abs_from_factor(1e6))
    dev = torch.device("cuda", device)
    f64 = dict(dtype=torch.float64, device=dev)
    rng = np.random.default_rng(32)
    K = kin.Kinematics(m, device=device)
    work = []
    frozen = [False]
    for a, b in lane_ranges(B, lanes):
        Bl = b - a
        q0 = np.zeros((Bl, n))
        q0[:, [m.names.index(s_ + "KneeSag") for s_ in "RL"]] = 0.5
        q0[:, [m.names.index(s_ + "HipSag") for s_ in "RL"]] = -0.25
        q0[:, [m.names.index(s_ + "AnkSag") for s_ in "RL"]] = -0.25
        q0[:, [m.names.index(s_ + "Elbj") for s_ in "RL"]] = -0.6
        q0 += rng.normal(0.0, 0.02, (Bl, n))
        st = BatchedStack(plan, Bl, device=device, want_levels=False)
        q = torch.as_tensor(q0, **f64).contiguous()
        pose = [torch.zeros((Bl, 12), **f64) for _ in range(4)]
        com = torch.zeros((Bl, 3), **f64)
        # (the caller's streams where it has them: streams created later can land on a hardware queue another lane already uses,
        #  and the lanes then run one after the other)
        stream = streams[len(work)] if streams is not None else torch.cuda.Stream(device=dev)
        st.stream = stream

        def fk(q=q, pose=pose, com=com, st=st):
            K.forward(q, frame_pose={f: pose[f] for f in range(4)}, frame_J={f: (st.A[1], 6 * f) for f in range(4)}, com=com, com_J=(st.A[0], 0))
        fk(); torch.cuda.synchronize()
        pose_d = [p_.clone() for p_ in pose]
        for f in (0, 1):
            pose_d[f][:, 9:] += torch.as_tensor(rng.uniform(-0.15, 0.15, (Bl, 3)), **f64)
        com_d = com.clone()
        qmin = torch.full((Bl, n), -2.5, **f64); qmax = torch.full((Bl, n), 2.5, **f64)
        qdot_max = torch.full((Bl, n), 2.0, **f64)
        q_ref = q.clone()
        leaf = {"B": Bl, "task": [[(com, com_d, None)], [(pose[f], pose_d[f], None) for f in range(4)], [(q, q_ref, None)]],
                "bound": [(q, qmin, qmax), (qdot_max, None, None)], "rows": []}

        kb = K.batch_args(q, frame_pose={f: pose[f] for f in range(4)}, frame_J={f: (st.A[1], 6 * f) for f in range(4)}, com=com, com_J=(st.A[0], 0))

        def step(fk=fk, st=st, leaf=leaf, q=q, Bl=Bl, stream=stream, kb=kb):
            with torch.cuda.stream(stream):
                if solve_only and frozen[0]:   # (diagnostic: update + cascade alone on the posture the closed loop has reached -- what the
                    st.cycle(leaf, cached=True)   #  solve costs on THIS data, without the producer and the integration)
                elif fused and rollout > 1:
                    st.control_rollout(K, kb, leaf, q, rollout)
                elif fused:
                    st.control_cycle(K, kb, leaf, q_integrate=q)
                else:
                    fk()
                    st.cycle(leaf, cached=True)
                    q.add_(st.dq[:Bl])
        work.append((st, step, stream, Bl))
    for _ in range(-(-warmup // rollout) + (2 * steps if solve_only else 0)):      # (solve_only: the closed loop runs as long as the timed passes do, then freezes)
        for _, step, _, _ in work:
            step()
    torch.cuda.synchronize()
    frozen[0] = True
    if solve_only:
        for _, step, _, _ in work:
            step(); step()
        torch.cuda.synchronize()
    graphs, note = [], None
    try:
        for st, step, stream, _ in work:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                for _ in range(calls):
                    step()
            graphs.append(g)
    except Exception as e:
        graphs, note = [], f"graph capture unavailable: {e}"[:200]
        for st, _, _, _ in work:
            st.set_schedule(True)
    torch.cuda.synchronize()

    def run_all():
        if graphs:
            for g, (_, _, stream, _) in zip(graphs, work):
                with torch.cuda.stream(stream):
                    g.replay()
        else:
            for _ in range(calls):
                for _, step, _, _ in work:
                    step()
    run_all(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_all()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ok = sum(int((st.status[:Bl] == 0).sum().item()) for st, _, _, Bl in work)
    its = torch.cat([st.iterations[:Bl].float() for st, _, _, Bl in work])
    how = ("ONE launch per step (osot_control_cycle_kernel<32,false,true>: the instance's kinematics, update, cascade and q += dq by the same wavefront)"
           if fused else "three launches per step (osot_kin_kernel, osot_cycle_kernel, the integration of q)")
    if rollout > 1:
        how = (f"ROLLOUTS of {rollout} control cycles per launch (osot_control_rollout: every robot's cycles follow each other on its own "
               "wavefront -- no launch and no wait for the batch's slowest robot between them)")
    return {"workload": "full control cycle on the device, BASELINE configs[2] stack on the 32-DoF humanoid: q -> kinematics (4 frame poses + "
                        "Jacobians, CoM + Jacobian, written into A_k) -> update + cascade -> q += dq; " + how + "; closed loop, every robot "
                        f"chasing its own wrist goals; {lanes} sub-batches on their own streams, {steps} steps of a lane per HIP graph" + ("" if graphs else " (plain launches)"),
            "batch": B, "lanes": lanes, "rollout": rollout, "value": B * steps / el, "unit": "solves/s", "ms_per_step": 1e3 * el / steps, "steps": steps,
            "solved_ok": f"{ok}/{B}", "note": note,
            "iterations_last_step": {"mean": float(its.mean().item()), "max": int(its.max().item()),
                                     "note": "active-set iterations per solve in the last step of the loop (the headline's synthetic batch: mean 32, max 63): "
                                             "this sub-line and the headline do not solve the same problems"},
            "roofline": roofline_of(plan, B, 1e3 * el / steps, steps * lanes, ("osot_control_cycle_kernel<32,false,true>" if fused else
                                    "osot_kin_kernel<false,32> + osot_cycle_kernel<32,false,true> + the integration of q") +
                                    " (whole step time as the divisor)")[0]}


def time_config5_coherent(B, device, steps=40, warmup=8, cycles=4, drift=0.01):
    """BASELINE config 5 at its shard size over TEMPORALLY COHERENT cycles (the headline's protocol: every input drifts 1 % from
    cycle to cycle), cold start against the hot start of the working sets (osot_solver_set_hotstart: what the reference's
    qpOASES back-end does from one control cycle to the next, QPOasesBackEnd.cpp:258-285).  The launch is its longest instance,
    and the longest instances are the ones with thirty-odd active torque limits: their working sets persist, so re-adding them
    without scans and without add-then-drop churn is what the hot start buys here (it does not at config 3, whose working
    sets are small: DESIGN.md section 4)."""
    import numpy as np
    from opensot_amd import synth
    from opensot_amd.solver import BatchedStack
    plan, leaf = synth.make_id_stack(B, seed=5000)
    rng = np.random.default_rng(77)
    leaves = [leaf]
    for _ in range(cycles - 1):
        leaves.append(synth.perturb(leaves[-1], rng, drift))
    res, dqs = {}, {}
    for hot in (False, True):
        st = BatchedStack(plan, B, device=device, want_levels=False)
        if hot:
            st.set_hotstart(True)
        devs = [st.load_leaf(lf) for lf in leaves]
        for i in range(warmup):
            st.cycle(devs[i % cycles])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            st.cycle(devs[i % cycles])
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        it = st.iterations[:B].float()
        res[hot] = {"value": B * steps / el, "ms_per_step": 1e3 * el / steps, "solved_ok": f"{int((st.status[:B] == 0).sum().item())}/{B}",
                    "iterations_mean": float(it.mean().item()), "iterations_max": int(it.max().item())}
        dqs[hot] = st.dq[:B].double().cpu().numpy()
    return {"workload": "BASELINE configs[4] shard (38-DoF floating-base inverse dynamics, n = 50, 102 constraint rows) over %d temporally "
                        "coherent cycles (%.0f %% drift per cycle), one fused update + cascade launch per step" % (cycles, 100 * drift),
            "batch": B, "unit": "solves/s", "steps": steps,
            "value": res[True]["value"], "ms_per_step": res[True]["ms_per_step"],
            "mode": "hot start of every level's working set from the instance's previous cycle (osot_solver_set_hotstart)",
            "hot_start": res[True], "cold_start": res[False],
            "hot_vs_cold_max_abs_dq_diff": float(np.abs(dqs[True] - dqs[False]).max())}


def time_nhqp(B, device, steps=5, warmup=2, lanes=1, streams=None, graph=True):
    """the null-space front-end (OpenSoT::solvers::nHQP) on the C3 stack: update + osot_nhqp_solve.  lanes > 1 (with the
    caller's streams): the batch as sub-batches on their own streams -- one sub-batch's level preparation runs under the other's
    QP / accumulation launches and under the tail of its preparation (tools/exp_frontend_lanes.py: 4.16 -> 4.79 M; three sub-batches of
    1365 / 1366 fit the preparation kernel's 1536 resident wavefronts in one round each: tools/exp_nhqp_lanes.py)"""
    from opensot_amd import synth
    from opensot_amd.parallel import lane_ranges
    from opensot_amd.solver import BatchedStack
    plan, leaf = synth.make_velocity_stack("C3", B, seed=3000)
    if streams is None or len(streams) < lanes:
        lanes = 1
    work = []
    for j, (a, b) in enumerate(lane_ranges(B, lanes)):
        st = BatchedStack(plan, b - a, device=device, want_levels=False)
        if lanes > 1:
            st.stream = streams[j]
        work.append((st, st.load_leaf(sub_leaf(leaf, a, b) if lanes > 1 else leaf), b - a))

    def step():
        for st, dv, Bl in work:
            st.update(dv); st.solve_nhqp(Bl)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    # the timed steps of a sub-batch as ONE HIP graph (like the headline: ten launches per step and sub-batch, and the host's share
    # of a launch -- argument structs through ctypes -- is not what is being measured); plain launches if capture is unavailable
    graphs = []
    if graph and lanes > 1:
        try:
            for st, dv, Bl in work:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st.stream):
                    for _ in range(steps):
                        st.update(dv); st.solve_nhqp(Bl)
                graphs.append(g)
        except Exception:
            graphs = []
        torch.cuda.synchronize()

    def run():
        if graphs:
            for g, (st, _, _) in zip(graphs, work):
                with torch.cuda.stream(st.stream):
                    g.replay()
        else:
            for _ in range(steps):
                step()
    if graphs:
        run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ok = sum(int((st.status[:Bl] == 0).sum().item()) for st, _, Bl in work)
    return {"workload": "BASELINE configs[2] stack through the reference's null-space front-end (nHQP.cpp:155-204; defaults: A/b "
                        "regularisation at 0.05 sv_max, selective null-space regularisation): per level an SVD of A N (Gram-side "
                        "tridiagonalisation, eigenvalues by Sturm bisection, vectors by twisted factorisation, in LDS), the QP in the nf = 32 / 29 / 5 free coordinates, q += N z, N <- N V2; three launches per level",
            "batch": B, "lanes": lanes, "value": B * steps / el, "unit": "solves/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "solved_ok": f"{ok}/{B}",
            "roofline": hbm_roofline(algo_bytes_per_solve(plan), B, 1e3 * el / steps,
                                     [("osot_update_kernel", B // lanes, lanes), ("osot_nhqp_prepare_kernel<32>", B // lanes, lanes * plan.L),
                                      ("osot_qp_kernel<32, false>", B // lanes, lanes * plan.L), ("osot_nhqp_accumulate_kernel", B // lanes, lanes * plan.L)],
                                     "osot_nhqp_prepare_kernel<32> + osot_qp_kernel<32> + osot_nhqp_accumulate_kernel per level (far from "
                                     "both roofs: the level preparation is one wavefront's dependent instruction stream at six wavefronts per CU -- LDS-limited -- and there are nine dependent launches)")}


def time_ehqp(B, device, steps=10, warmup=3, lanes=1, streams=None):
    """the equality-only front-end (OpenSoT::solvers::eHQP) on the C3 stack: update + osot_ehqp_solve; lanes > 1 (with the
    caller's streams): the batch as sub-batches on their own streams (the tail of one launch under the other's)"""
    from opensot_amd import synth
    from opensot_amd.parallel import lane_ranges
    from opensot_amd.solver import BatchedStack
    plan, leaf = synth.make_velocity_stack("C3", B, seed=3000)
    if streams is None or len(streams) < lanes:
        lanes = 1
    work = []
    for j, (a, b) in enumerate(lane_ranges(B, lanes)):
        st = BatchedStack(plan, b - a, device=device, want_levels=False)
        if lanes > 1:
            st.stream = streams[j]
        work.append((st, st.load_leaf(sub_leaf(leaf, a, b) if lanes > 1 else leaf), b - a))

    def step():
        for st, dv, Bl in work:
            st.update(dv); st.solve_ehqp(Bl)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ok = sum(int((st.status[:Bl] == 0).sum().item()) for st, _, Bl in work)
    return {"workload": "BASELINE configs[2] stack through the reference's equality-only front-end (eHQP.cpp:64-95: damped pseudo-"
                        "inverses and projectors, the box is not used): null-space recursion, per level a pivoted Householder QR of "
                        "W^1/2 A Z (no eigen-decomposition since round 3); all levels of an instance in one launch",
            "batch": B, "lanes": lanes, "value": B * steps / el, "unit": "solves/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "solved_ok": f"{ok}/{B}",
            "roofline": hbm_roofline(algo_bytes_per_solve(plan), B, 1e3 * el / steps,
                                     [("osot_update_kernel", B // lanes, lanes), ("osot_ehqp_qr_kernel<32>", B // lanes, lanes)],
                                     "osot_ehqp_qr_kernel<32> (all levels of an instance in one launch: row-oriented pivoted Householder QR "
                                     "in LDS, a chain of dependent reflector steps: far from both roofs)")}


def time_admm(B, device, steps=10, warmup=3):
    """the OSQP-convention back-end on explicit QPs of config-2 size: cold solves, then warm-started solves over
    drifting linear terms (what OSQPBackEnd's kept workspace does between control cycles)"""
    from opensot_amd import torch_api as ta
    rng = np.random.default_rng(6000)
    n, nc = 32, 6
    M = rng.normal(0.0, 0.3, size=(B, 38, n))
    H = np.einsum("bri,brj->bij", M, M) + 1.0e-3 * np.eye(n)
    g = rng.normal(0.0, 1.0, size=(B, n))
    A = rng.normal(0.0, 0.3, size=(B, nc, n))
    lA = -rng.uniform(0.05, 0.5, size=(B, nc)); uA = rng.uniform(0.05, 0.5, size=(B, nc))
    lo = -rng.uniform(0.05, 0.5, size=(B, n)); up = rng.uniform(0.05, 0.5, size=(B, n))
    dev = torch.device("cuda", device)
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev).contiguous()
    tH, tA, tlA, tuA, tl, tu = t(H), t(A), t(lA), t(uA), t(lo), t(up)
    gs = [t(g + 0.01 * k * rng.normal(0.0, 1.0, size=(B, n))) for k in range(4)]
    OSQP = ta.solver_back_ends.OSQP
    run = lambda gk, warm=None: ta.qp_solve(tH, gk, tA, tlA, tuA, tl, tu, eps_regularisation=1e4, be_solver=OSQP, warm=warm)

    def timed(warm):
        for i in range(warmup):
            run(gs[i % 4], warm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        its = 0
        for i in range(steps):
            x, st_, it = run(gs[i % 4], warm)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        return el, int((st_ == 0).sum().item()), float(it.float().mean().item())
    el_c, ok_c, it_c = timed(None)
    el_w, ok_w, it_w = timed(ta.admm_state(B, n, nc, box=True, device=device))
    nbytes = 8 * (n * n + n + nc * n + 2 * nc + 2 * n + n)        # H, g, A, lA, uA, l, u in; x out
    return {"workload": "osot_admm_kernel: OSQP-convention ADMM (Ruiz equilibration, rho adaptive, eps 1e-5) on synthetic QPs of config-2 "
                        "size (n = 32, 6 rows + box), linear term drifting 1 % per call; one wavefront per QP",
            "batch": B, "value": B * steps / el_w, "unit": "solves/s", "ms_per_step": 1e3 * el_w / steps, "steps": steps,
            "solved_ok": f"{ok_w}/{B}", "admm_iterations_per_solve_warm": it_w,
            "cold_start": {"value": B * steps / el_c, "ms_per_step": 1e3 * el_c / steps, "admm_iterations_per_solve": it_c, "solved_ok": f"{ok_c}/{B}"},
            "roofline": hbm_roofline(nbytes, B, 1e3 * el_w / steps, [("osot_admm_kernel", B, 1)],
                                     "osot_admm_kernel (an iteration is two mat-vecs against the LDS-resident inverse and rows: far from the HBM roof "
                                     "by construction, the problem is read once and iterated on ~100 times)")}


def time_wide_qp(B, device, steps=3, warmup=1, nv=55, ncon=5):
    """the explicit-QP surface BEYOND the 64 lanes of a wavefront (osot_qp_big.h, round 6: one 256-thread workgroup per QP): the second
    level of a floating-base inverse-dynamics stack of nv + 3 ncon = 70 variables -- a Postural task on the accelerations under the
    optimality rows of the level above, dynamic-feasibility equalities, friction pyramids, torque limits -- as the reference's iHQP hands
    it to its plugin (iHQP.cpp:263-358 -> BackEnd::solve), B problems per call through osot_qp_solve_batch"""
    import ctypes as C
    from opensot_amd import abi, synth
    dev = torch.device("cuda", device)
    n = nv + 3 * ncon
    eps = 1.0e3 * 2.221e-16 * 1.0e6
    gens = [synth.wide_id_levels(np.random.default_rng(7000 + b), nv, ncon)[1] for b in range(B)]
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    p = lambda a: C.c_void_p(a.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def solve(qs, nsteps):
        nc = qs[0][2].shape[0]
        ts = [t(np.stack([q[j] for q in qs])) for j in range(7)]
        x = torch.zeros((B, n), dtype=torch.float64, device=dev)
        st = torch.zeros((B,), dtype=torch.int32, device=dev); it = torch.zeros((B,), dtype=torch.int32, device=dev)
        call = lambda: abi.lib().osot_qp_solve_batch(B, n, nc, *[p(a) for a in ts], eps, 0, p(x), p(st), p(it), stream)
        for _ in range(warmup):
            abi.check(call(), "osot_qp_solve_batch")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            abi.check(call(), "osot_qp_solve_batch")
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / max(1, nsteps), x, st, it, nc
    _, x0, st0, _, _ = solve([g(0, []) for g in gens], 0)
    xs0 = x0.cpu().numpy()
    ms, x1, st1, it1, nc1 = solve([g(1, [xs0[b]]) for b, g in enumerate(gens)], steps)
    nbytes = 8 * (n * n + n + nc1 * n + 2 * nc1 + 2 * n + n)
    return {"workload": f"osot_qp_big_kernel: explicit QPs of {n} variables and {nc1} rows (level 1 of a floating-base inverse-dynamics stack, "
                        f"{nv} accelerations + {ncon} point contacts) through osot_qp_solve_batch, one 256-thread workgroup per QP, cold",
            "batch": B, "value": B / (ms * 1e-3), "unit": "solves/s", "ms_per_step": ms, "steps": steps,
            "solved_ok": f"{int((st1 == 0).sum().item())}/{B}", "level0_solved_ok": f"{int((st0 == 0).sum().item())}/{B}",
            "iterations_mean": float(it1.float().mean().item()),
            "roofline": hbm_roofline(nbytes, B, ms, [("osot_qp_big_kernel", 4 * min(B, 1024 if n <= 96 else 512), 1)],
                                     "osot_qp_big_kernel (a coverage path: ~10 barrier-separated sections per active-set iteration over an L2-resident J)")}


def time_kinematics(B, device, steps=20, warmup=5):
    from opensot_amd import kinematics as kin
    m = kin.humanoid32()
    K = kin.Kinematics(m, device=device)
    dev = torch.device("cuda", device)
    q = torch.as_tensor(np.random.default_rng(1).uniform(-1, 1, (B, m.n)), device=dev)
    A = torch.zeros((B, 27, m.n), dtype=torch.float64, device=dev)
    poses = {f: torch.zeros((B, 12), dtype=torch.float64, device=dev) for f in range(4)}
    com = torch.zeros((B, 3), dtype=torch.float64, device=dev)
    run = lambda: K.forward(q, frame_pose=poses, frame_J={f: (A, 6 * f) for f in range(4)}, com=com, com_J=(A, 24))
    for _ in range(warmup):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    nbytes = 8 * (m.n + 27 * m.n + 4 * 12 + 3)        # q in; 4 frame Jacobians + CoM Jacobian, 4 poses, CoM out
    gbs = B * nbytes / (ms * 1e-3) / 1e9
    return {"workload": "osot_kin_kernel: 32-DoF humanoid, 4 frame poses + 6xn Jacobians, CoM + 3xn Jacobian, written into the "
                        "stacked A_k (SURVEY 8f-1)", "batch": B, "value": B / (ms * 1e-3), "unit": "instances/s", "avg_launch_ms": ms,
            "roofline": hbm_roofline(nbytes, B, ms, [("osot_kin_kernel<false, 32>", (B + 1) // 2, 1)],
                                     "osot_kin_kernel<false,32> (two instances per wavefront)")}


COMPACT_LINE_LIMIT = 12288      # bytes; tests/test_distributed_cpu.py and tests/test_gpu_features_r2.py hold the line to it


def _num(x, nd=6):
    """a finite float rounded to nd significant digits, None for anything else (strict JSON: no NaN / Infinity)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, int):
        return x
    try:
        x = float(x)
    except (TypeError, ValueError):
        return None
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float(f"{x:.{nd}g}")


def _roof(r):
    if not isinstance(r, dict):
        return None
    out = {"bound": r.get("bound"), "kernel": str(r.get("kernel", ""))[:80], "achieved": _num(r.get("achieved")), "peak": _num(r.get("peak")),
           "unit": r.get("unit"), "frac": _num(r.get("frac")), "traffic": _num(r.get("traffic"))}
    if r.get("avg_launch_ms") is not None:
        out["avg_launch_ms"] = _num(r.get("avg_launch_ms"))
    for k in ("peak_spec", "frac_of_spec", "mfma_busy_frac"):
        if r.get(k) is not None:
            out[k] = _num(r.get(k))
    return out


def compact_line(out):
    """the ONE stdout line: the contract's keys, numbers only beyond them.  Prose, sweeps, evidence and the per-sub-line roofline
    blocks stay in bench_details.json (and on stderr)."""
    c = {k: out.get(k) for k in ("metric", "unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    c["value"], c["ms_per_step"] = _num(out.get("value"), 9), _num(out.get("ms_per_step"), 9)
    cfg = dict(out.get("config", {}))
    cfg["workload"] = str(cfg.get("workload", ""))[:300]
    cfg.pop("protocol", None)
    cfg["parallelism"] = str(cfg.get("parallelism", ""))[:80]
    c["config"] = cfg
    c["solved_ok_rank0"] = out.get("solved_ok_rank0")
    for k in ("solved_ok_all_ranks", "gathered_vs_own_max_abs_dq_diff_rank0", "lanes_vs_single_launch_max_abs_dq_diff"):
        if k in out:
            c[k] = out[k]
    if "roofline" in out:
        c["roofline"] = _roof(out["roofline"])
        h = out.get("roofline_hbm") or {}
        c["roofline_hbm"] = {k: _num(h.get(k)) for k in ("achieved", "peak", "frac", "traffic", "algorithmic_bytes_per_solve")}
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        c["cpu_baseline"] = {"value": _num(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                             "single_thread": _num(cb.get("single_thread")), "sample": str(cb.get("sample", ""))[:200]}
    pr = out.get("parity")
    if isinstance(pr, dict):
        c["parity"] = ({"error": str(pr["error"])[:120]} if "error" in pr else
                       {"tolerance": pr.get("tolerance"), "instances": pr.get("instances"), "within_tolerance": pr.get("within_tolerance"),
                        "max_abs_dq_diff": _num(pr.get("max_abs_dq_diff")),
                        "beyond_tolerance_device_lex_better": (pr.get("beyond_tolerance") or {}).get("lexicographically_better", {}).get("device")})
    oc = out.get("other_configs")
    if isinstance(oc, dict):
        c["other_configs"] = {}
        for name, r in oc.items():
            if not isinstance(r, dict) or "error" in r:
                c["other_configs"][name] = {"error": str((r or {}).get("error", "?"))[:80]}
                continue
            e = {"value": _num(r.get("value")), "frac": _num((r.get("roofline") or {}).get("frac"), 3), "ok": r.get("solved_ok")}
            c["other_configs"][name] = e
    c["details"] = "bench_details.json (full record: protocol, sweeps, evidence)"
    return c


def fit_line(c, limit=None):
    """the compact record as ONE strict-JSON line below the limit.  It degrades instead of failing (an assert here ran after
    every measurement, right in front of the only stdout write -- a line that outgrew the limit was a run without a result): optional
    blocks are shortened, then dropped, in the order of how little the contract needs them; the contract's keys always go out."""
    limit = COMPACT_LINE_LIMIT if limit is None else limit
    enc = lambda d: (json.dumps(d, allow_nan=False) + "\n").encode()
    line = enc(c)
    steps = (lambda d: d.__setitem__("other_configs", {k: (v.get("value") if isinstance(v, dict) else None) for k, v in d.get("other_configs", {}).items()}),
             lambda d: d.pop("other_configs", None), lambda d: d.pop("parity", None),
             lambda d: d.__setitem__("cpu_baseline", {k: (d.get("cpu_baseline") or {}).get(k) for k in ("value", "unit", "cores", "kind")}),
             lambda d: d.pop("roofline_hbm", None), lambda d: d.__setitem__("config", {"workload": str(d.get("config", {}).get("workload", ""))[:120]}))
    dropped = 0
    for st in steps:
        if len(line) < limit:
            break
        st(c); dropped += 1
        c["truncated"] = f"{dropped} optional block(s) shortened or dropped to fit {limit} bytes: see bench_details.json"
        line = enc(c)
    return line


def lanes_auto(plan, B, device, front_end="iHQP"):
    """sub-batch count of a sub-line from the LIBRARY's rule (opensot_amd.parallel.suggest_lanes) over the wavefronts the device
    holds at once for this plan's kernel (osot_solver_resident_waves / _nhqp): nothing here is tuned per sub-line"""
    from opensot_amd.parallel import suggest_lanes
    from opensot_amd.solver import BatchedStack
    probe = BatchedStack(plan, 1, device=device, want_levels=False)
    resident = probe.resident_waves_nhqp() if front_end == "nHQP" else probe.resident_waves()
    lanes_auto.last_resident = resident
    return suggest_lanes(B, resident)


_EXTRA_STREAMS = {}


def streams_for(streams, n, device):
    """the caller's streams first, more where a sub-line has more lanes -- from ONE pool per device, created once: HIP maps streams onto a
    handful of hardware queues in creation order, so a fresh stream per sub-line sooner or later lands on the queue of another lane of the
    same sub-line and the two serialise (seen: COMAN35 S3 / S4 at 6.0 / 3.4 M with a stream created per call against 8.3 / 5.3 M)"""
    if streams is None:
        return None
    pool = _EXTRA_STREAMS.setdefault(str(device), [])
    while len(streams) + len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return list(streams[:n]) + pool[:max(0, n - len(streams))]


def sub_leaf(lf, lo, hi):
    """rows [lo, hi) of a numpy leaf dict (opensot_amd.synth layout)"""
    cut = lambda a: None if a is None else a[lo:hi]
    out = {"B": hi - lo, "A": [cut(a) for a in lf["A"]],
           "task": [[tuple(cut(x) for x in t) for t in lev] for lev in lf["task"]],
           "bound": [tuple(cut(x) for x in t) for t in lf["bound"]],
           "rows": [tuple(cut(x) for x in t) for t in lf["rows"]]}
    if lf.get("C") is not None:
        out["C"] = [cut(a) for a in lf["C"]]
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a launcher around it: re-execute under torch.distributed.run with N ranks on this
    node (rendezvous on 127.0.0.1, a free port); rank 0 of the children prints the JSON line.  The N = 1 path is untouched."""
    import socket
    import subprocess
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-per-gpu", type=int, default=4096)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--cpu-sweep", action="store_true",
                    help="cpu_baseline with the 64-thread / all-cores points and the one-process-per-worker sweep (minutes of CPU)")
    ap.add_argument("--details", default=os.path.join(ROOT, "bench_details.json"),
                    help="where the full record goes (protocol prose, sweeps, parity evidence, every sub-line's roofline block); "
                         "the stdout line is the compact form of it")
    ap.add_argument("--cycles", type=int, default=4,
                    help="distinct, temporally coherent control cycles the steps rotate through (SURVEY 8d: cycle t+1 = "
                         "cycle t + 1 %% perturbation of every input, Jacobians included); 1 = repeat one cycle")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("OSOT_BENCH_LANES", "0")),
                    help="sub-batches per GPU, each on its own stream with no join between steps "
                         "(opensot_amd.parallel.PipelinedCycle); 1 = one launch per step; 0 (default) = the library's rule: "
                         "opensot_amd.parallel.suggest_lanes over the wavefronts the device holds for the plan's kernel "
                         "(osot_solver_resident_waves), for the headline and every sub-line")
    ap.add_argument("--no-graph", action="store_true",
                    help="submit every launch of the timed steps one by one instead of replaying a HIP graph per lane "
                         "(the graph form is used where a lane's step is the solver's own launch: one GPU, no gather)")
    ap.add_argument("--backend", default=os.environ.get("OSOT_BENCH_BACKEND", "hip"), choices=("hip", "stub"),
                    help="stub = CPU tensors + gloo + a stand-in for the solver: exercises the launcher, the sharding, the lanes "
                         "and the gather of this very script where there is no GPU (tests/test_distributed_cpu.py); never a result")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # stdout carries the ONE result line of rank 0 and nothing else: whatever the libraries below print there (RCCL writes
    # a version banner to stdout through C stdio, flushed at exit, i.e. AFTER the line) is sent to stderr, on every rank
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    stub = args.backend == "stub"
    if not stub:
        torch.cuda.set_device(local_rank)
    dist = None
    # (OSOT_BENCH_FORCE_DIST=1 runs the collective path with a world of one: a single-GPU check of the RCCL plumbing)
    use_dist = world > 1 or os.environ.get("OSOT_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        if stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from opensot_amd import synth
    from opensot_amd.parallel import PipelinedCycle, ShardedCycle, ShardGather, lane_ranges, shard_range, timed_steps

    Bl = args.batch_per_gpu
    Bg = Bl * world
    lo, hi = shard_range(Bg, rank, world)
    assert hi - lo == Bl
    plan_probe, _ = synth.make_velocity_stack(args.config, 1, seed=1)
    S = max(1, min(args.lanes, Bl)) if args.lanes > 0 else (2 if stub else lanes_auto(plan_probe, Bl, local_rank))
    device = torch.device("cpu") if stub else torch.device("cuda", local_rank)
    sync = (lambda: None) if stub else torch.cuda.synchronize
    # every rank generates only its own shard (seed = 1000*config + rank)
    plan, leaf = synth.make_velocity_stack(args.config, Bl, seed=3000 + rank)
    # K temporally coherent cycles (MPC-rollout-like): each is the previous one with every float input moved by
    # ~1 %.  The Jacobians change too: every cycle has its own A_k buffers (the kinematics producer's output) and the
    # stack just points at them -- nothing is copied inside a step.
    K = max(1, args.cycles)
    rng = np.random.default_rng(77 + rank)
    leaves = [leaf]
    for _ in range(K - 1):
        leaves.append(synth.perturb(leaves[-1], rng, 0.01))
    if stub:
        from opensot_amd.parallel import StubStack as BatchedStack
    else:
        from opensot_amd.solver import BatchedStack
    # S lanes: contiguous sub-batches of the shard, each with its own solver (dispatch-order state), stream and -- with
    # more than one rank -- its own communicator: the collectives of different lanes are never ordered against each other
    ov = os.environ.get("OSOT_GATHER_OVERLAP")      # (developer switch: 1 = asynchronous collective on the group's stream)
    # The collective inside the lane's HIP graph (RCCL collectives can be captured): with a world of one -- the single-GPU check
    # of this path -- by default; with more ranks on request (OSOT_BENCH_DIST_GRAPH=1: not run on hardware by the builder, who has
    # one GPU), otherwise plain launches with the asynchronous two-block gather (ShardGather's default for a world > 1)
    dg = os.environ.get("OSOT_BENCH_DIST_GRAPH")
    dist_graph = use_dist and not stub and (dg == "1" or (dg is None and world == 1))
    if dist_graph and ov is None:
        ov = "0"
    spans = lane_ranges(Bl, S)
    stacks, lanes, gathers = [], [], []
    for j, (a, b) in enumerate(spans):
        stj = BatchedStack(plan, b - a, device=local_rank, want_levels=False)
        devs, A_sets = [], []
        for lf in leaves:
            stj.A = [None if t is None else torch.empty_like(t) for t in stj.A]
            devs.append(stj.load_leaf(sub_leaf(lf, a, b)))
            A_sets.append(stj.A)
        gj = None
        if use_dist:
            grp = dist.new_group(list(range(world))) if S > 1 else None
            gj = ShardGather((b - a) * world, plan.n, device, torch.float64, group=grp,
                             overlap=None if ov is None else ov == "1", sizes=[b - a] * world)
        stacks.append(stj); gathers.append(gj)
        lanes.append(ShardedCycle(stj, devs, A_sets, b - a, gj))
    streams = None if (stub or S == 1) else [torch.cuda.Stream(device=device) for _ in range(S)]
    cyc = PipelinedCycle(lanes, streams)
    st = stacks[0]

    # warm up first, then switch the event timing on for the timed steps only
    for _ in range(args.warmup):
        cyc.step()
    sync()
    # the steps of a lane as ONE HIP graph (opensot_amd.parallel.PipelinedCycle.capture): G steps per replay, G an even divisor
    # of --steps so that EXACTLY that many steps are timed (one graph per starting cycle of the rotation a replay can meet, so
    # the replays follow the rotation like plain launches); the same steps submitted launch by launch are timed right after,
    # with the HIP events that give the kernel's own duration
    graph_steps, graph_note, graph_elapsed, graph_host_ms, graph_diff, graph_launch_ms = 0, None, None, None, None, None
    if not args.no_graph and not stub and streams is not None and (not use_dist or dist_graph):
        cand = [g for g in range(4, 51, 2) if args.steps % g == 0]
        if cand:
            try:
                cyc.capture(max(cand))
                graph_steps = cyc.graph_steps
                cyc.replay(); sync()                   # (the replays continue the rotation where the warm-up left it)
                graph_elapsed = timed_steps(lambda: cyc.replay(timed=True), args.steps // graph_steps, 0, sync, dist if use_dist else None, device)
                graph_launch_ms = cyc.replay_launch_ms()
                graph_host_ms = 1e3 * getattr(timed_steps, "host_enqueue_s", 0.0) / args.steps
                g_dq = torch.cat([stj.dq[:b - a] for stj, (a, b) in zip(stacks, spans)]).clone()
                g_ok = sum(int((stj.status[:b - a] == 0).sum().item()) for stj, (a, b) in zip(stacks, spans))
                # one more rotation through plain launches ends on the same cycle: bit-identical results expected
                for _ in range(K):
                    cyc.step()
                sync()
                graph_diff = float((torch.cat([stj.dq[:b - a] for stj, (a, b) in zip(stacks, spans)]) - g_dq).abs().max().item())
                if g_ok != Bl or graph_diff != 0.0:
                    graph_note = f"graph replay disagreed with plain launches (ok {g_ok}/{Bl}, max |ddq| {graph_diff}): not used"
                    graph_elapsed = None
            except Exception as e:      # capture is an optimisation of the submission, never a requirement
                graph_note, graph_elapsed = f"graph capture unavailable: {e}"[:300], None
        else:
            graph_note = "--steps has no even divisor in 4..20: launches submitted one by one"
    for stj in stacks:
        stj.set_timing(True, stride=4)     # every fourth launch of a lane is bracketed by HIP events on its stream
    elapsed = timed_steps(cyc.step, args.steps, 0, sync, dist if use_dist else None, device)
    host_enqueue_ms = 1e3 * getattr(timed_steps, "host_enqueue_s", 0.0) / args.steps
    stream_elapsed = elapsed
    if graph_elapsed is not None:
        elapsed, host_enqueue_ms = graph_elapsed, graph_host_ms
    kt = [stj.kernel_time_ms() for stj in stacks]
    launches = sum(c for _, c in kt)
    kern_ms = sum(ms * c for ms, c in kt) / launches if launches else 0.0
    ok = sum(int((stj.status[:b - a] == 0).sum().item()) for stj, (a, b) in zip(stacks, spans))
    all_ok, gather_diff = None, None
    if use_dist:
        all_ok = sum(int((g.status == 0).sum().item()) for g in gathers)
        # what the collective delivered for THIS rank's rows against what the rank's own solver holds (bit for bit)
        gather_diff = 0.0
        for g, stj, (a, b) in zip(gathers, stacks, spans):
            off = sum(g.sizes[:rank])
            mine = g.dq[off:off + (b - a)]
            gather_diff = max(gather_diff, float((mine - stj.dq[:b - a]).abs().max().item()))
    # the same kernel with plain in-order dispatch (reported beside the headline, never as the headline)
    for stj in stacks:
        stj.set_schedule(longest_first=False)
    for _ in range(3):
        cyc.step()
    sync()
    for stj in stacks:
        stj.kernel_time_ms()
    t1 = time.perf_counter()
    for _ in range(10):
        cyc.step()
    sync()
    inorder_elapsed = (time.perf_counter() - t1) / 10
    ikt = [stj.kernel_time_ms() for stj in stacks]
    inorder_kern_ms = sum(ms * c for ms, c in ikt) / max(1, sum(c for _, c in ikt))
    for stj in stacks:
        stj.set_schedule(longest_first=True)
        stj.set_timing(False)
    # one launch per step (no lanes), same steps: what the pipelining is worth, reported beside the headline
    single_value = None
    if S > 1 and not stub and not use_dist:
        st1 = BatchedStack(plan, Bl, device=local_rank, want_levels=False)
        devs1, A1 = [], []
        for lf in leaves:
            st1.A = [None if t is None else torch.empty_like(t) for t in st1.A]
            devs1.append(st1.load_leaf(lf)); A1.append(st1.A)
        one = ShardedCycle(st1, devs1, A1, Bl, None)
        single_value = Bl * args.steps / timed_steps(one.step, args.steps, args.warmup, sync, None, device)
        del one, st1, devs1, A1
    update_ms = None
    if not stub:
        # the AutoStack::update equivalent on its own (SURVEY 8d asks for it beside the whole-step figure)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20):
            for stj, ln in zip(stacks, lanes):
                stj.update(ln.dev_leaves[i % K])
        e1.record()
        sync()
        update_ms = e0.elapsed_time(e1) / 20

    if rank == 0:
        value = Bg * args.steps / elapsed
        out = {
            "metric": "whole-body QP solves/sec (32-DoF, 3-level stack)",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "host_enqueue_ms_per_step_rank0": host_enqueue_ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[2]: batch={Bl}/GPU x 32-DoF, 3-level iHQP (CoM / 4 Cartesian / Postural) + joint & velocity "
                                    f"limit box, eps factor 1e6; step = AutoStack::update + cascade of every instance, {K} drifting cycles, "
                                    f"{S} sub-batch stream(s)/GPU" + ("; + RCCL all-gather of dq and status" if use_dist else "")),
                       "protocol": f"BASELINE configs[2]: batch={Bl}/GPU x 32-DoF, 3-level iHQP "
                                   "(CoM / l_wrist(0.1)+r_wrist+l_sole+r_sole / Postural), joint-limit & "
                                   "velocity-limit box, eps factor 1e6; step = AutoStack::update + cascade solve of every "
                                   f"instance; steps rotate through {K} temporally coherent cycles (1 % input drift per cycle)"
                                   + (f"; the {Bl} instances of a GPU are {S} contiguous sub-batches, each one launch per step on "
                                      "its own stream, a sub-batch's step t+1 ordered behind ITS step t only (instances are "
                                      "independent: no join between steps inside the timed bracket)" if S > 1 else "")
                                   + ("; + RCCL all-gather of dq and status" + (" per sub-batch" if S > 1 else "") if use_dist else ""),
                       "global_batch": Bg, "n_dof": plan.n, "levels": plan.L,
                       "rows_per_level": [plan.m(k) for k in range(plan.L)], "lanes_per_gpu": S,
                       "parallelism": f"instances sharded over {world} GPU(s), no data-path collective"},
            "solved_ok_rank0": f"{ok}/{Bl}",
            "update_avg_ms_rank0": update_ms,
            "dispatch": {"mode": "longest-first by the previous cycle's active-set iteration counts "
                                 "(osot_solver_set_schedule default; results are order-independent)",
                         "in_order_value_rank0": Bl / inorder_elapsed, "in_order_avg_launch_ms": inorder_kern_ms},
        }
        if stub:
            out["data"] = "STUB BACK-END (CPU tensors, gloo, no solver): launcher / sharding / gather check only, not a result"
        if single_value is not None:
            out["single_launch_per_step_value_rank0"] = single_value
        out["submission"] = {
            "mode": ("hip graph: %d steps of a lane per graph, one graph launch per lane and %d steps" % (graph_steps, graph_steps))
                    if graph_elapsed is not None else "one launch per lane and step on the lane's stream",
            "stream_launch_value_rank0": Bl * args.steps / stream_elapsed,
            "stream_launch_ms_per_step": 1e3 * stream_elapsed / args.steps,
            "graph_vs_stream_launch_max_abs_dq_diff": graph_diff,
            "note": graph_note or ("the timed region is EXACTLY --steps steps in both forms; roofline.avg_launch_ms is measured inside the "
                                   "graph pass, roofline.avg_launch_ms_stream_pass over the event-bracketed launches of the stream pass")}
        if all_ok is not None:
            out["solved_ok_all_ranks"] = f"{all_ok}/{Bg}"
            out["gathered_vs_own_max_abs_dq_diff_rank0"] = gather_diff
        if launches > 0 and kern_ms > 0 and not stub:
            traffic, src = pmc_traffic([("osot_cycle_kernel<32, false, true>", Bl // S + 1, 1)])   # (+ the order workgroup)
            # S launches are in flight at a time (one per lane), each sharing the chip with the others: a launch's own duration is
            # not the time the chip needed for its instances.  The roofline figures therefore take the time of a whole STEP (all
            # lanes; it contains every launch of the step plus the order kernels and launch gaps, so it under-states the kernel).
            step_ms = 1e3 * elapsed / args.steps
            mfma_busy = pmc_mfma_busy([("osot_cycle_kernel<32, false, true>", Bl // S + 1, S)], step_ms)
            rf, rh = roofline_of(plan, Bl, step_ms if S > 1 else kern_ms, launches,
                                 "osot_cycle_kernel<32,false,true> (AutoStack::update + the whole cascade of an instance by one wavefront: "
                                 "fp64 MFMA H build + blocked Cholesky + null-space elimination, VALU/LDS active set; the BOX instantiation: "
                                 "config 3 has no constraint rows, its only inequalities are the joint / velocity limit box)",
                                 None if traffic is None else traffic * S,
                                 (src + "; per launch x launches per step") if traffic else
                                 "no PMC passes committed for this kernel source: null rather than a stale figure")
            # the launch duration IN THE PASS THAT PRODUCED `value`: inside the graph replays a lane's launches follow each other with
            # no host in between, so (events around the lane's graph) / (launches per graph) is one launch plus the ~1 us gap to the
            # next; the event-bracketed launches of the stream pass (which follows, and is slower per step) are reported beside it
            if graph_elapsed is not None and graph_launch_ms:
                rf["avg_launch_ms"] = graph_launch_ms
                rf["avg_launch_ms_source"] = ("HIP events on the lane's stream around each lane's graph in the timed graph pass, divided by the "
                                              "launches per graph (one launch + the gap to the next launch of the graph)")
                rf["avg_launch_ms_stream_pass"] = kern_ms
            else:
                rf["avg_launch_ms"] = kern_ms
                rf["avg_launch_ms_source"] = "HIP events bracketing every fourth launch on the lane's stream, in the timed (stream-launch) pass"
            rf["launch_batch"] = Bl // S
            rf["concurrent_launches"] = S
            # MFMA utilisation of the step against the chip's matrix cores at the peak clock (north_star asks for it beside the HBM GB/s)
            rf["mfma_busy_frac"] = mfma_busy
            rf["mfma_busy_source"] = (PMC_PROFILE + ": SQ_VALU_MFMA_BUSY_CYCLES of the step's launches / (step time x 2.4 GHz x 1024 SIMDs)"
                                      if mfma_busy is not None else "no PMC passes committed for this kernel source")
            if S > 1:
                rf["time_base"] = ("ms_per_step (whole-job): %d launches of %d instances overlap on the chip, so achieved = algorithmic "
                                   "flops of a step / step time; avg_launch_ms is one launch's own (overlapped) duration as the HIP "
                                   "events on its stream and rocprofv3 see it" % (S, Bl // S))
            # the fused kernel's algorithmic bytes: the cascade's compulsory bytes + the leaf inputs the update
            # half reads + the assembled arrays it writes (they are outputs in their own right; the cascade half reads them
            # back from the CU's L1 / L2, which costs no HBM traffic)
            lb, ab = leaf_bytes_per_instance(lanes[0].dev_leaves[0]), assembled_bytes_per_solve(plan)
            rh["algorithmic_bytes_per_solve_cascade_only"] = rh["algorithmic_bytes_per_solve"]
            rh["algorithmic_bytes_per_solve"] = rh["algorithmic_bytes_per_solve"] + lb + ab
            rh["leaf_input_bytes_per_solve"], rh["assembled_output_bytes_per_solve"] = lb, ab
            rh["achieved"] = rh["algorithmic_bytes_per_solve"] * Bl / ((step_ms if S > 1 else kern_ms) * 1e-3) / 1e9
            rh["frac"] = rh["achieved"] / rh["peak"]
            out["roofline"], out["roofline_hbm"] = rf, rh
        if world == 1 and not args.no_other_configs and not stub:
            oc = {}
            for key, name, B, st_ in (("C2", "C2", 1024, 20), ("C4", "C4", 4096, 20), ("C5", "C5", 1024, 20),
                                      ("C5_B4096", "C5", 4096, 12)):   # (the last: config 5 beyond the shard size, four rounds of resident wavefronts)
                try:
                    oc[key] = time_config(name, B, local_rank, steps=st_, lanes=S, streams=streams)
                except Exception as e:
                    oc[key] = {"error": str(e)}
            # sub-batches by the library's rule (suggest_lanes: the 40-lane kernel holds 1792 wavefronts at once -- 22 KB of LDS each --, so
            # 4096 robots go as three launches of one round each; rounds 4-5 found the same count by hand, tools/exp_coman_lanes3.py)
            for which in ("S1", "S2", "S3", "S4"):
                try:
                    nl = lanes_auto(coman_stack(which, 35), 4096, local_rank) if (streams is not None and S >= 2) else S
                    oc["COMAN35_" + which] = time_coman35(which, 4096, local_rank, lanes=nl, streams=streams_for(streams, nl, device))
                    oc["COMAN35_" + which]["lanes"] = nl
                    oc["COMAN35_" + which]["resident_wavefronts"] = getattr(lanes_auto, "last_resident", None)
                except Exception as e:
                    oc["COMAN35_" + which] = {"error": str(e)[:300]}
            for which in ("S1", "S2", "S3", "S4"):   # the same four through the reference's null-space front-end (published: 0.2969 / 0.2637 / 0.3191 / 0.3721 ms per solve)
                try:
                    nl = lanes_auto(coman_stack(which, 35), 4096, local_rank, front_end="nHQP") if (streams is not None and S >= 2) else S
                    oc[f"COMAN35_{which}_nHQP"] = time_coman35(which, 4096, local_rank, steps=6, warmup=2, front_end="nHQP", lanes=nl, streams=streams_for(streams, nl, device))
                    oc[f"COMAN35_{which}_nHQP"]["lanes"] = nl
                    oc[f"COMAN35_{which}_nHQP"]["resident_wavefronts"] = getattr(lanes_auto, "last_resident", None)
                except Exception as e:
                    oc[f"COMAN35_{which}_nHQP"] = {"error": str(e)[:300]}
            try:
                oc["full_cycle"] = time_full_cycle(Bl, local_rank, lanes=S, streams=streams)
                three = time_full_cycle(Bl, local_rank, lanes=S, streams=streams, fused=False)
                oc["full_cycle"]["as_three_launches_per_step"] = {"value": three["value"], "ms_per_step": three["ms_per_step"], "solved_ok": three["solved_ok"]}
                alone = time_full_cycle(Bl, local_rank, lanes=S, streams=streams, solve_only=True)
                oc["full_cycle"]["update_and_cascade_alone_on_the_posture_the_loop_reached"] = {
                    "value": alone["value"], "ms_per_step": alone["ms_per_step"], "iterations_mean": alone["iterations_last_step"]["mean"],
                    "note": "the yardstick for this sub-line: the same launches without the producer and the integration, on the same problems "
                            "(the headline's synthetic batch is easier: 32 iterations per solve against 38.5 here)"}
            except Exception as e:
                oc["full_cycle"] = {"error": str(e)[:300]}
            try:
                oc["C5_coherent"] = time_config5_coherent(1024, local_rank)
            except Exception as e:
                oc["C5_coherent"] = {"error": str(e)}
            try:
                # sub-batches by the library's rule over the preparation kernels' resident wavefronts (osot_solver_resident_waves_nhqp)
                n3 = lanes_auto(synth.make_velocity_stack("C3", 1, seed=1)[0], 4096, local_rank, front_end="nHQP") if (streams is not None and S >= 2) else S
                oc["nHQP_C3"] = time_nhqp(4096, local_rank, steps=10, warmup=3, lanes=n3, streams=streams_for(streams, n3, device))      # (ten timed steps: at five the fill and drain of three lanes is a tenth of the region)
                oc["nHQP_C3"]["lanes"] = n3
                oc["nHQP_C3"]["resident_wavefronts"] = getattr(lanes_auto, "last_resident", None)
            except Exception as e:
                oc["nHQP_C3"] = {"error": str(e)}
            try:
                oc["eHQP_C3"] = time_ehqp(4096, local_rank, lanes=S, streams=streams)
            except Exception as e:
                oc["eHQP_C3"] = {"error": str(e)}
            try:
                oc["ADMM_qp"] = time_admm(1024, local_rank)
            except Exception as e:
                oc["ADMM_qp"] = {"error": str(e)}
            try:
                oc["wide_qp_70"] = time_wide_qp(1024, local_rank)
            except Exception as e:
                oc["wide_qp_70"] = {"error": str(e)}
            for key, Bk in (("kinematics", 4096), ("kinematics_B32768", 32768)):
                try:
                    oc[key] = time_kinematics(Bk, local_rank)
                except Exception as e:
                    oc[key] = {"error": str(e)}
            out["other_configs"] = oc
        if not args.no_cpu_baseline and world == 1 and not stub:
            ns = min(Bl, 4096)
            sample = {"B": ns, "A": [a[:ns] if a is not None else None for a in leaf["A"]],
                      "task": [[tuple(None if x is None else x[:ns] for x in t) for t in lev] for lev in leaf["task"]],
                      "bound": [tuple(None if x is None else x[:ns] for x in t) for t in leaf["bound"]],
                      "rows": [tuple(None if x is None else x[:ns] for x in t) for t in leaf["rows"]]}
            try:
                # the GPU's answer (with its per-level solutions) for the same sample: cycle 0 of the rotation
                stp = BatchedStack(plan, Bl, device=local_rank, want_levels=True)
                stp.update(stp.load_leaf(leaf)); stp.solve(Bl); sync()
                # (the lanes solved the same instances in sub-batches: their answers are these, bit for bit)
                for ln in lanes:       # bring every lane back to cycle 0 of the rotation
                    ln.i = 0
                cyc.step(); sync()
                lane_dq = torch.cat([stj.dq[:b - a] for stj, (a, b) in zip(stacks, spans)])
                out["lanes_vs_single_launch_max_abs_dq_diff"] = float((lane_dq - stp.dq[:Bl]).abs().max().item())
                out["parity"] = parity_report(plan, sample, stp.dq[:ns].cpu().numpy(), stp.x_levels[:ns].cpu().numpy(),
                                              stp.accepted_slack[:ns].cpu().numpy())
            except Exception as e:  # the oracle is a checker; its absence must not kill the bench line
                out["parity"] = {"error": f"unavailable: {e}"}
            try:
                out["cpu_baseline"] = cpu_baseline(plan, sample, process_sweep=args.cpu_sweep)
            except Exception as e:
                out["cpu_baseline"] = {"value": None, "unit": "solves/s", "cores": 0, "kind": "port",
                                       "sample": f"unavailable: {e}"}
        sys.stdout.flush()
        # the full record: to a side file and to stderr; stdout carries the compact form only (the driver's capture is finite)
        full = json.dumps(out)
        try:
            with open(args.details, "w") as f:
                f.write(full + "\n")
        except OSError as e:
            sys.stderr.write(f"bench_details not written: {e}\n")
        sys.stderr.write(full + "\n"); sys.stderr.flush()
        line = fit_line(compact_line(out))
        while line:
            line = line[os.write(result_fd, line):]
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

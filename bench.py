#!/usr/bin/env python3
"""bench.py -- whole-body QP solves/s of the MI355X hot path (BASELINE.json metric).

One "step" = one pass of the hot path over one resident batch: AutoStack::update (leaf -> b, W, box,
rows) + the 3-level iHQP cascade (H/g assembly + one QP per level) for every instance of the shard.
Workload at N=1: BASELINE.json configs[2], batch 4096 x 32-DoF, 3 levels (CoM / 4 Cartesian / Postural),
joint-limit and velocity-limit box, eps factor 1e6 (the reference benchmark's value, coman_ik.cpp:453).
N>1: weak scaling, 4096 instances per GPU, instances sharded contiguously over ranks (no data-path
collective: they are independent), one RCCL all-gather of the solved dq shards per step.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# algorithmic figures per instance-solve for the C3 stack (formulas restated in DESIGN.md)
ALGO_BYTES_PER_SOLVE_C3 = (3 + 24) * 32 * 8 + 59 * 8 + 59 * 8 + 2 * 32 * 8 + 32 * 8   # A rows + b + w + box + dq
FP64_PEAK_TFLOPS = 78.6     # MI355X FP64 vector = matrix peak (AMD public spec; not in MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec


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


def cpu_baseline(plan, leaf_sample, seconds_target=6.0, dq_device=None):
    """CPU path timed on this host: the reference's own qpOASES (oracle/_ref, kind 'reference') when the
    prebuilt library is present, otherwise the plain-C port (kind 'port'); restated cascade around it.
    dq_device: the GPU's answer for the same sample; the second half of BASELINE's metric (max |dq - dq_ref|) is then
    reported against the CPU answers of this leg."""
    from oracle import pyoracle as po
    asm = po.assemble(plan, leaf_sample)
    cores = os.cpu_count() or 1
    kind, be = ("reference", po.BE_QPOASES_REF) if po.ref_available() else ("port", po.BE_EIQP_EQ)
    try:
        r = po.ihqp_solve_batch(asm, be, nthreads=cores, cycles=1)
    except Exception:
        kind, be = "port", po.BE_EIQP_EQ
        r = po.ihqp_solve_batch(asm, be, nthreads=cores, cycles=1)
    parity = None
    if dq_device is not None:
        # per instance the distance to the CLOSEST witness: qpOASES at OpenSoT's options (its early termination at
        # 2.2e-7 leaves a few instances per thousand 1e-5..3e-3 from the optimum), qpOASES run to the exact optimum
        # (which fails on some instances) and the line-by-line restatement of the reference's eiQuadProg
        e_ref = np.where(r["status"] == 1, np.abs(dq_device - r["dq"]).max(axis=1), np.inf)
        e = e_ref.copy()
        witness = "qpOASES 3.1 at OpenSoT's options" if kind == "reference" else "C Goldfarb-Idnani port"
        if kind == "reference":
            rx = po.ihqp_solve_batch(asm, be, nthreads=cores, cycles=1, termination_tolerance=10 * 2.221e-16)
            e = np.minimum(e, np.where(rx["status"] == 1, np.abs(dq_device - rx["dq"]).max(axis=1), np.inf))
            re_ = po.ihqp_solve_batch(asm, po.BE_EIQP_EQ, nthreads=cores, cycles=1)
            e = np.minimum(e, np.where(re_["status"] == 1, np.abs(dq_device - re_["dq"]).max(axis=1), np.inf))
            witness = ("closest of: qpOASES 3.1 at OpenSoT's options, qpOASES run to the exact optimum, the restated "
                       "eiQuadProg (DESIGN.md 2)")
        fin = np.isfinite(e)
        parity = {"max_abs_dq_diff": float(e[fin].max()) if fin.any() else None, "tolerance": 1e-6,
                  "instances_compared": int(fin.sum()), "instances": int(e.size), "witness": witness,
                  "within_tolerance_of_qpOASES_at_reference_options": int((e_ref <= 1e-6).sum()),
                  "max_abs_dq_diff_vs_qpOASES_at_reference_options": float(e_ref[np.isfinite(e_ref)].max())}
    per_cycle = max(r["seconds"], 1e-6)
    cycles = int(max(1, min(5000, seconds_target / per_cycle)))
    r = po.ihqp_solve_batch(asm, be, nthreads=cores, cycles=cycles)
    B = asm["B"]
    return parity, {"value": B * cycles / r["seconds"], "unit": "solves/s", "cores": cores, "kind": kind,
            "sample": f"{B} instances of the same C3 stack x {cycles} cycles, {cores} host threads "
                      f"({'qpOASES 3.1 hot-started across cycles' if kind == 'reference' else 'C Goldfarb-Idnani port'}), "
                      f"{r['seconds']:.1f} s, ok={int(r['status'].sum())}/{B}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-per-gpu", type=int, default=4096)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cycles", type=int, default=4,
                    help="distinct, temporally coherent control cycles the steps rotate through (SURVEY 8d: cycle t+1 = "
                         "cycle t + 1 %% perturbation of every input, Jacobians included); 1 = repeat one cycle")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    dist = None
    # (OSOT_BENCH_FORCE_DIST=1 runs the collective path with a world of one: a single-GPU check of the RCCL plumbing)
    use_dist = world > 1 or os.environ.get("OSOT_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    from opensot_amd import synth
    from opensot_amd.solver import BatchedStack
    from opensot_amd.parallel import shard_range

    Bl = args.batch_per_gpu
    Bg = Bl * world
    lo, hi = shard_range(Bg, rank, world)
    assert hi - lo == Bl
    # every rank generates only its own shard (seed = 1000*config + rank)
    plan, leaf = synth.make_velocity_stack(args.config, Bl, seed=3000 + rank)
    st = BatchedStack(plan, Bl, device=local_rank, want_levels=False)
    # K temporally coherent cycles (MPC-rollout-like): each is the previous one with every float input moved by
    # ~1 %.  The Jacobians change too: every cycle has its own A_k buffers (the kinematics producer's output) and the
    # stack just points at them -- nothing is copied inside a step.
    K = max(1, args.cycles)
    rng = np.random.default_rng(77 + rank)
    leaves = [leaf]
    for _ in range(K - 1):
        leaves.append(synth.perturb(leaves[-1], rng, 0.01))
    dev_leaves, A_sets = [], []
    for lf in leaves:
        st.A = [None if a is None else torch.empty_like(a) for a in st.A]
        dev_leaves.append(st.load_leaf(lf))
        A_sets.append(st.A)
    # The all-gather of the solved dq shards runs on a SIDE stream and is double-buffered: the gather of step t
    # overlaps the update + solve of step t+1 (which write the other dq buffer); a buffer goes back to the solver only
    # after the gather that read it has finished (stream-side event waits, the host never blocks).
    # That variant is OPT-IN (OSOT_BENCH_OVERLAP_GATHER=1): measured with a world of one on a MI355X its extra
    # host-side calls (events, stream switches) cost 22 us per step against 9 us for the plain gather on the solve
    # stream, which is therefore the default; at 8 GPUs the gather is ~1 MB per rank (~7-50 us).
    overlap = use_dist and os.environ.get("OSOT_BENCH_OVERLAP_GATHER") == "1"
    dq_bufs = [st.dq, torch.empty_like(st.dq)] if overlap else [st.dq]
    gathered = [torch.empty((Bg, plan.n), dtype=torch.float64, device=st.device) for _ in range(2)] if use_dist else None
    side = torch.cuda.Stream(device=st.device) if overlap else None
    main = torch.cuda.current_stream(st.device)
    solved = [torch.cuda.Event(), torch.cuda.Event()] if overlap else None
    gathered_ev = [None, None]
    state = {"i": 0}

    def drain():
        if overlap:
            main.wait_stream(side)

    def step():
        t = state["i"]
        i = t % K
        j = t & 1 if overlap else 0
        state["i"] += 1
        if overlap:
            if gathered_ev[j] is not None:
                main.wait_event(gathered_ev[j])      # the gather that read dq_bufs[j] two steps ago
            st.dq = dq_bufs[j]
        st.A = A_sets[i]
        st.update(dev_leaves[i])
        st.solve(Bl)
        if overlap:
            solved[j].record(main)
            with torch.cuda.stream(side):
                side.wait_event(solved[j])
                dist.all_gather_into_tensor(gathered[j], dq_bufs[j][:Bl])
                ev = torch.cuda.Event()
                ev.record(side)
                gathered_ev[j] = ev
        elif use_dist:
            dist.all_gather_into_tensor(gathered[0], st.dq[:Bl])

    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    st.set_timing(True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()                      # every gather of the timed steps has completed inside the timed region
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=st.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms, launches = st.kernel_time_ms()
    ok = int((st.status[:Bl] == 0).sum().item())
    # the same kernel with plain in-order dispatch (reported beside the headline, never as the headline): the
    # bench repeats ONE synthetic control cycle, so the previous cycle's iteration counts predict this cycle's
    # exactly -- a real control loop is temporally coherent, not identical
    st.set_schedule(longest_first=False)
    for _ in range(3):
        step()
    drain()
    torch.cuda.synchronize()
    st.kernel_time_ms()
    t1 = time.perf_counter()
    for _ in range(10):
        step()
    drain()
    torch.cuda.synchronize()
    inorder_elapsed = (time.perf_counter() - t1) / 10
    inorder_kern_ms, _ = st.kernel_time_ms()
    st.set_schedule(longest_first=True)
    st.set_timing(False)
    # the AutoStack::update equivalent on its own (SURVEY 8d asks for it beside the whole-step figure)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20):
        st.update(dev_leaves[i % K])
    e1.record()
    torch.cuda.synchronize()
    update_ms = e0.elapsed_time(e1) / 20

    if rank == 0:
        solves = Bg * args.steps
        value = solves / elapsed
        flops = algo_flops_per_solve(plan)
        bytes_per = ALGO_BYTES_PER_SOLVE_C3 if args.config == "C3" else None
        kern_s = kern_ms * 1e-3
        out = {
            "metric": "whole-body QP solves/sec (32-DoF, 3-level stack)",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: batch={Bl}/GPU x 32-DoF, 3-level iHQP "
                                   "(CoM / l_wrist(0.1)+r_wrist+l_sole+r_sole / Postural), joint-limit & "
                                   "velocity-limit box, eps factor 1e6; step = AutoStack::update + cascade solve; "
                                   f"steps rotate through {K} temporally coherent cycles (1 % input drift per cycle)"
                                   + ("; + RCCL all-gather of dq" + (" on a side stream (double-buffered)" if overlap else "") if use_dist else ""),
                       "global_batch": Bg, "n_dof": plan.n, "levels": plan.L,
                       "rows_per_level": [plan.m(k) for k in range(plan.L)],
                       "parallelism": f"instances sharded over {world} GPU(s), no data-path collective"},
            "solved_ok_rank0": f"{ok}/{Bl}",
            "update_avg_ms_rank0": update_ms,
            "dispatch": {"mode": "longest-first by the previous cycle's active-set iteration counts "
                                 "(osot_solver_set_schedule default; results are order-independent)",
                         "in_order_value_rank0": Bl / inorder_elapsed, "in_order_avg_launch_ms": inorder_kern_ms},
        }
        traffic = None
        try:   # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), same workload only
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_v11_pmc_cascade.json")))
            if args.config == "C3" and Bl == 4096:
                traffic = pm["hbm_bytes_per_launch_corrected"]
        except Exception:
            traffic = None
        if launches > 0 and kern_s > 0:
            tf = Bl * flops / kern_s / 1e12
            out["roofline"] = {
                "bound": "mfma", "kernel": "osot_cascade_kernel<32,false> (fp64 MFMA H build + blocked Cholesky, VALU/LDS active set)", "achieved": tf, "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TFLOPS, "traffic": traffic,
                "avg_launch_ms": kern_ms, "launches": launches,
                "algorithmic_flops_per_solve": flops,
                "note": "fp64 FMA roof (nominal algorithmic flops; AI ~ 20 flop/B > machine balance ~ 10). "
                        "peak = AMD public FP64 vector/matrix spec, not in MI355X_MICROARCH.md"}
            if bytes_per:
                gbs = Bl * bytes_per / kern_s / 1e9
                out["roofline_hbm"] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
                                       "traffic_source": "profiles/r01_v11_pmc_cascade.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 per MI355X_MICROARCH.md)" if traffic else None,
                                       "algorithmic_bytes_per_solve": bytes_per}
        if not args.no_cpu_baseline and world == 1:
            ns = min(Bl, 4096)
            sample = {"B": ns, "A": [a[:ns] if a is not None else None for a in leaf["A"]],
                      "task": [[tuple(None if x is None else x[:ns] for x in t) for t in lev] for lev in leaf["task"]],
                      "bound": [tuple(None if x is None else x[:ns] for x in t) for t in leaf["bound"]],
                      "rows": [tuple(None if x is None else x[:ns] for x in t) for t in leaf["rows"]]}
            try:
                # the GPU's answer for the same sample (cycle 0 of the rotation)
                st.A = A_sets[0]; st.update(dev_leaves[0]); st.solve(Bl); torch.cuda.synchronize()
                par, out["cpu_baseline"] = cpu_baseline(plan, sample, dq_device=st.dq[:ns].cpu().numpy())
                if par is not None:
                    out["parity"] = par
            except Exception as e:  # the oracle is a checker; its absence must not kill the bench line
                out["cpu_baseline"] = {"value": None, "unit": "solves/s", "cores": 0, "kind": "port",
                                       "sample": f"unavailable: {e}"}
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

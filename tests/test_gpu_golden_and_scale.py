"""GPU parity against the committed qpOASES golden vectors, and full-size (BASELINE.json) runs checked through
size-independent properties: feasibility, hierarchy (optimality rows hold), KKT of the last level, permutation
equivariance, idempotence of repeated solves, plus an oracle spot check."""
import numpy as np
import pytest
import torch

from helpers import load_golden
from opensot_amd import synth
from opensot_amd.solver import BatchedStack

pytestmark = pytest.mark.gpu


def _solve(plan, leaf, active=None, want_levels=True):
    B = leaf["B"]
    st = BatchedStack(plan, B, device=0, want_levels=want_levels)
    st.update(st.load_leaf(leaf))
    st.level_active = active
    st.solve(B)
    torch.cuda.synchronize()
    return st


@pytest.mark.parametrize("cfg", ["C2", "C3", "C4", "C5"])
def test_golden_qpoases(cfg, gpu_device):
    plan, leaf, z = load_golden(cfg)
    st = _solve(plan, leaf)
    B = leaf["B"]
    dq = st.dq[:B].cpu().numpy()
    assert (st.status[:B].cpu().numpy() == 0).all()
    # AutoStack::update outputs: bit-exact against the oracle's leaf restatement
    for k in range(plan.L):
        np.testing.assert_allclose(st.b[k][:B].cpu().numpy(), z[f"asm_b{k}"], rtol=0, atol=1e-15)
    if "asm_l" in z.files:
        np.testing.assert_array_equal(st.l[:B].cpu().numpy(), z["asm_l"])
        np.testing.assert_array_equal(st.u[:B].cpu().numpy(), z["asm_u"])
    if "asm_lo" in z.files:
        np.testing.assert_allclose(st.lo[:B].cpu().numpy(), z["asm_lo"], rtol=1e-14, atol=1e-13)
        np.testing.assert_allclose(st.up[:B].cpu().numpy(), z["asm_up"], rtol=1e-14, atol=1e-13)
    ok = z["ok_ref"].astype(bool); okx = z["ok_exact"].astype(bool)
    # north_star: solved joint velocities within 1e-6 of the reference qpOASES back-end (fp64)
    assert np.abs(dq[ok] - z["x_ref"][ok][:, -1]).max() < 1e-6
    assert np.abs(dq[okx] - z["x_exact"][okx][:, -1]).max() < 1e-8


@pytest.mark.parametrize("cfg,B", [("C2", 1024), ("C3", 4096), ("C4", 4096)])
def test_full_size_properties(cfg, B, oracle, gpu_device):
    plan, leaf = synth.make_velocity_stack(cfg, B)
    st = _solve(plan, leaf)
    dq = st.dq[:B].cpu().numpy(); xl = st.x_levels[:B].cpu().numpy()
    status = st.status[:B].cpu().numpy()
    assert (status == 0).all()
    l = st.l[:B].cpu().numpy(); u = st.u[:B].cpu().numpy()
    # box feasibility at every level
    for k in range(plan.L):
        assert (xl[:, k] >= l - 1e-10).all() and (xl[:, k] <= u + 1e-10).all()
    # hierarchy: A_j x_k == A_j x_j for all j < k (optimality rows, iHQP.cpp:164-170)
    A = [a if a is not None else None for a in leaf["A"]]
    for k in range(1, plan.L):
        for j in range(k):
            Aj = A[j] if A[j] is not None else np.broadcast_to(np.eye(plan.n), (B, plan.n, plan.n))
            lhs = np.einsum("brn,bn->br", Aj, xl[:, k]); rhs = np.einsum("brn,bn->br", Aj, xl[:, j])
            assert np.abs(lhs - rhs).max() < 1e-9
    # global rows
    if plan.nc:
        Cx = np.einsum("brn,bn->br", st.C[:B].cpu().numpy(), dq)
        assert (Cx <= np.minimum(st.up[:B].cpu().numpy(), 1e20) + 1e-10).all()
    np.testing.assert_array_equal(dq, xl[:, -1])
    # oracle spot check on a strided subset
    sub = slice(0, B, B // 64)
    asm = oracle.assemble(plan, {"B": len(range(*sub.indices(B))),
                                 "A": [a[sub] if a is not None else None for a in leaf["A"]],
                                 "task": [[tuple(None if x is None else x[sub] for x in t) for t in lev] for lev in leaf["task"]],
                                 "bound": [tuple(None if x is None else x[sub] for x in t) for t in leaf["bound"]],
                                 "rows": [tuple(None if x is None else x[sub] for x in t) for t in leaf["rows"]]})
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    assert np.abs(dq[sub] - ref["dq"]).max() < 1e-9


def test_permutation_equivariance_and_idempotence(gpu_device):
    """instances are independent: permuting the batch permutes dq bit-exactly; solving twice gives identical bits"""
    B = 512
    plan, leaf = synth.make_velocity_stack("C3", B, seed=99)
    st = _solve(plan, leaf)
    dq1 = st.dq[:B].cpu().numpy().copy()
    st.solve(B); torch.cuda.synchronize()
    np.testing.assert_array_equal(dq1, st.dq[:B].cpu().numpy())
    perm = np.random.default_rng(0).permutation(B)
    pleaf = {"B": B, "A": [a[perm] if a is not None else None for a in leaf["A"]],
             "task": [[tuple(None if x is None else x[perm] for x in t) for t in lev] for lev in leaf["task"]],
             "bound": [tuple(None if x is None else x[perm] for x in t) for t in leaf["bound"]],
             "rows": [tuple(None if x is None else x[perm] for x in t) for t in leaf["rows"]]}
    st2 = _solve(plan, pleaf)
    np.testing.assert_array_equal(dq1[perm], st2.dq[:B].cpu().numpy())


@pytest.mark.parametrize("active", [(1, 0, 1), (0, 1, 1), (1, 1, 0)])
def test_inactive_levels_gpu(active, oracle, gpu_device):
    plan, leaf = synth.make_velocity_stack("C3", 64, seed=5)
    asm = oracle.assemble(plan, leaf)
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1, active=active)
    st = _solve(plan, leaf, active=active)
    assert (st.status[:64].cpu().numpy() == 0).all()
    assert np.abs(st.dq[:64].cpu().numpy() - ref["dq"]).max() < 1e-9


def test_default_eps_factor(oracle, gpu_device):
    """iHQP's default eps_regularisation 2e2 (iHQP.h:32): H + 4.4e-11 I, cond ~ 1e11 on the upper levels"""
    plan, leaf = synth.make_velocity_stack("C3", 128, seed=17, eps_factor=2e2)
    asm = oracle.assemble(plan, leaf)
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=1)
    st = _solve(plan, leaf)
    assert (st.status[:128].cpu().numpy() == 0).all()
    assert np.abs(st.dq[:128].cpu().numpy() - ref["dq"]).max() < 1e-6


def test_empty_batch_and_timing_api(gpu_device):
    plan, leaf = synth.make_velocity_stack("C3", 8, seed=1)
    st = BatchedStack(plan, 8, device=0)
    st.solve(0)                       # empty input: accepted, nothing launched
    st.set_timing(True)
    st.update(st.load_leaf(leaf)); st.solve(8); st.solve(8)
    torch.cuda.synchronize()
    ms, cnt = st.kernel_time_ms()
    assert cnt == 2 and ms > 0
    with pytest.raises(RuntimeError):
        st.solve(9)                   # exceeds max_batch -> OSOT_ERR_INVALID


def test_dispatch_order_does_not_change_results(gpu_device):
    """longest-first dispatch (osot_order_kernel) only permutes which workgroup solves which instance: the
    second solve (ordered by the first solve's iteration counts) and an in-order solve are bit-identical"""
    plan, leaf = synth.make_velocity_stack("C4", 4096, seed=23)
    st = BatchedStack(plan, 4096, device=0)
    dl = st.load_leaf(leaf)
    st.update(dl)
    st.solve(4096)
    first = st.dq[:4096].clone()
    it1 = st.iterations[:4096].clone()
    st.solve(4096)                    # dispatched longest-first from the first solve's counts
    torch.cuda.synchronize()
    assert torch.equal(st.dq[:4096], first) and torch.equal(st.iterations[:4096], it1)
    st.solve(1000)                    # another batch size: falls back to plain order, same answers
    torch.cuda.synchronize()
    assert torch.equal(st.dq[:1000], first[:1000])
    st.set_schedule(longest_first=False)
    st.solve(4096)
    torch.cuda.synchronize()
    assert torch.equal(st.dq[:4096], first)
    assert (st.status[:4096] == 0).all()


def test_randomised_parity_sweep(oracle, gpu_device):
    """40 random stack shapes x 192 instances (generic multi-level stacks with equality / inequality rows and boxes,
    low-rank levels, the humanoid and inverse-dynamics configurations, eps 1e6 and the default 2e2) against the reference's
    qpOASES at its own options, qpOASES run to the exact optimum and the eiQuadProg restatement; ONE criterion, no
    exclusions: within 1e-6 of a witness, else feasible and lexicographically not worse (tests/stress_parity.py)"""
    if not oracle.ref_available():
        pytest.skip("needs oracle/_ref (qpOASES)")
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "stress_parity.py"), "11", "40"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "40 configurations x 192 instances" in out.stdout and ": 0 with a mismatch" in out.stdout, out.stdout[-2000:]
    import re
    m = re.search(r"\((\d+) of (\d+) instances have a witness", out.stdout)
    assert m and int(m.group(1)) > 0.9 * int(m.group(2)), out.stdout[-500:]      # the comparison is not vacuous


def test_closed_loop_robustness_sweep(oracle, gpu_device):
    """tests/stress_closed_loop.py, the two seeds that used to leave instances stuck with a false INFEASIBLE (instance
    173 of seed 2 from cycle 232 on, 278 unsolved of seed 3): 1024 humanoids x 300 cycles chasing wrist targets through
    the body with 16 capsule pairs, joint limits and the velocity box on; every unsolved instance would be re-solved by
    qpOASES and the eiQuadProg restatement -- there must be none"""
    if not oracle.ref_available():
        pytest.skip("needs oracle/_ref (qpOASES)")
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for seed in ("2", "3"):
        out = subprocess.run([sys.executable, os.path.join(root, "tests", "stress_closed_loop.py"), seed, "1024", "300"],
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        assert "307200 closed-loop solves" in out.stdout and " 0 not solved" in out.stdout and " 0 product-only failures" in out.stdout, out.stdout[-2000:]


def test_closed_loop_robustness_sweep_default_eps(oracle, gpu_device):
    """the same sweep at iHQP's DEFAULT eps factor 2e2 (iHQP.h:32; absolute 4.4e-11), seeds 4, 7 and 9: round 1 left five
    instances stuck there (1150 + 298 product-only failures) where qpOASES goes on; with the optimality rows posed
    relative to the previous level's solution (osot_qp_core.h, kFeasMargin) there must be no product-only failure"""
    if not oracle.ref_available():
        pytest.skip("needs oracle/_ref (qpOASES)")
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for seed in ("4", "7", "9"):
        out = subprocess.run([sys.executable, os.path.join(root, "tests", "stress_closed_loop.py"), seed, "1024", "300", "200"],
                             capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        assert "307200 closed-loop solves" in out.stdout and " 0 product-only failures" in out.stdout, out.stdout[-2000:]


def test_inverse_dynamics_full_size(oracle, gpu_device):
    """BASELINE config 5 shard (8192 over 8 GPUs = 1024 per GPU), ALL on the device: [B_u, -J_f'], [B, -Jc'] and the [J 0]
    task rows written by osot_id_rows from the model quantities, update + cascade, tau from osot_computed_torque
    (InverseDynamics.cpp:57-96): its floating-base rows vanish and the acceptance flag is set, torque limits / friction
    cones hold; 256 instances against the eiQuadProg restatement AND the reference's qpOASES"""
    import torch
    from opensot_amd.dynamics import IdModel
    from opensot_amd.solver import BatchedStack
    B = 1024
    plan, leaf = synth.make_id_stack(B, seed=50)
    nv = leaf["model"]["nv"]
    st = BatchedStack(plan, B, device=0)
    bare = dict(leaf); bare["A"] = [np.zeros_like(leaf["A"][0]), None]; bare["C"] = [None] * len(leaf["C"])
    dev = st.load_leaf(bare)
    md = IdModel(leaf["model"]["B"], leaf["model"]["h"], leaf["model"]["Jc"], device=0)
    J = [np.ascontiguousarray(leaf["A"][0][:, o:o + r, :nv]) for o, r in ((0, 3), (3, 6), (9, 6))]
    md.write_rows(st, dyn_block=0, tau_block=2, tasks=[(0, 0, J[0]), (0, 3, J[1]), (0, 9, J[2])])
    st.cycle(dev)
    tau_d, ok_d = md.computed_torque(st.dq[:B])
    torch.cuda.synchronize()
    dq = st.dq[:B].cpu().numpy()
    assert (st.status[:B].cpu().numpy() == 0).all()
    tau = tau_d.cpu().numpy()
    np.testing.assert_allclose(tau, synth.computed_torque(leaf, dq), rtol=0, atol=1e-10)
    assert (ok_d.cpu().numpy() == 1).all() and np.abs(tau[:, :6]).max() < 1e-8
    assert np.abs(tau[:, 6:]).max() <= 30.0 + 1e-8
    sub = slice(0, B, 4)      # 256 instances
    sl = {"B": len(range(*sub.indices(B))), "A": [a[sub] if a is not None else None for a in leaf["A"]],
          "task": [[tuple(None if x is None else x[sub] for x in t) for t in lev] for lev in leaf["task"]],
          "bound": [], "rows": [tuple(None if x is None else x[sub] for x in t) for t in leaf["rows"]],
          "C": [None if x is None else x[sub] for x in leaf["C"]]}
    asm = oracle.assemble(plan, sl)
    ref = oracle.ihqp_solve_batch(asm, oracle.BE_EIQP_EQ, nthreads=0)
    assert (ref["status"] == 1).all() and np.abs(dq[sub] - ref["dq"]).max() < 1e-8
    if oracle.ref_available():
        # north_star's tolerance as it is written: ABSOLUTE 1e-6 on x = [qddot; F] against qpOASES at OpenSoT's options, every
        # instance counted (helpers.parity_census prints the census); an instance beyond it has to pass the literal
        # acceptance rule (feasible to 1e-7 and lexicographically not worse than qpOASES and the eiQuadProg restatement)
        from helpers import parity_census
        rq = oracle.ihqp_solve_batch(asm, oracle.BE_QPOASES_REF, nthreads=0)
        within, rule, fails = parity_census(asm, dq[sub], [("qpOASES", rq), ("eiQuadProg", ref)], tol=1e-6, label="C5 full size")
        assert not fails and within >= 0.95 * asm["B"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["graph_captured_collective", "async_two_block_gather"])
def test_bench_rccl_path_with_a_world_of_one(gpu_device, mode):
    """bench.py's multi-GPU path on the one GPU a test box has (OSOT_BENCH_FORCE_DIST=1): init_process_group("nccl"), one process
    group and one ShardGather per lane on DEVICE tensors, the solver writing dq / status straight into the collective's send
    block, the timing bracket with its barriers.  Every instance solved on "all ranks", and what the collective delivered is
    the solver's own dq bit for bit."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(OSOT_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if mode == "async_two_block_gather":
        # what a world of MORE than one rank runs by default: plain launches, the collective asynchronous on its group's stream,
        # two alternating send / receive blocks (ShardGather overlap) -- forced here on the one GPU a test box has
        env.update(OSOT_GATHER_OVERLAP="1", OSOT_BENCH_DIST_GRAPH="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--no-other-configs", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    if r.returncode != 0:
        # ONE more attempt on another rendezvous port: this is the only test of the suite that brings up a process group, and a
        # rendezvous / RCCL bring-up hiccup of a fresh box must not read as a solver failure (seen once in ~40 runs of the suite,
        # never reproduced in 28 isolated runs; a second failure is reported with both stderr tails)
        first = r.stderr[-1500:]
        env["MASTER_PORT"] = "29573"
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
        assert r.returncode == 0, "first attempt:\n" + first + "\nsecond attempt:\n" + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert len(lines[0]) < 12288 and r.stdout.rstrip().endswith(lines[0])
    out = json.loads(lines[0], parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
    assert out["n_gpus"] == 1 and out["steps"] == 4
    assert out["solved_ok_all_ranks"] == "4096/4096" and out["solved_ok_rank0"] == "4096/4096"
    assert out["gathered_vs_own_max_abs_dq_diff_rank0"] == 0.0
    assert "RCCL all-gather" in out["config"]["workload"] and out["value"] > 1.0e6

"""developer experiment (GPU box): the COMAN35 closed loops submitted like the headline -- the batch as sub-batches on their own
streams, one osot_control_cycle launch per step and lane, the steps of a lane as one HIP graph"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from opensot_amd import kinematics as kin
from opensot_amd.solver import BatchedStack
from opensot_amd.parallel import lane_ranges

def run(which, B, lanes, steps=20, warmup=5, fused=True, graph=True):
    m, lo, up = kin.from_json(os.path.join(bench.ROOT, "tests", "golden", "coman_tree.json"))
    n = m.n
    plan = bench.coman_stack(which, n)
    dev = torch.device("cuda", 0)
    f64 = dict(dtype=torch.float64, device=dev)
    rng = np.random.default_rng(35)
    K = kin.Kinematics(m, device=0)
    work = []
    for a, b in lane_ranges(B, lanes):
        Bl = b - a
        q0 = np.zeros((Bl, n))
        for s_ in "RL":
            q0[:, m.names.index(s_ + "HipSag")] = -0.3; q0[:, m.names.index(s_ + "KneeSag")] = 0.6
            q0[:, m.names.index(s_ + "AnkSag")] = -0.3; q0[:, m.names.index(s_ + "Elbj")] = -0.8
            q0[:, m.names.index(s_ + "ShSag")] = 0.2
        q0[:, m.names.index("LShLat")] = 0.3; q0[:, m.names.index("RShLat")] = -0.3
        q0[:, 6:] += rng.normal(0.0, 0.02, (Bl, n - 6))
        q0 = np.clip(q0, np.maximum(lo, -10.0) + 1e-3, np.minimum(up, 10.0) - 1e-3)
        st = BatchedStack(plan, Bl, device=0, want_levels=False)
        stream = torch.cuda.Stream(device=dev)
        st.stream = stream
        q = torch.as_tensor(q0, **f64).contiguous()
        pose = [torch.zeros((Bl, 12), **f64) for _ in range(4)]
        com = torch.zeros((Bl, 3), **f64)
        where, off = {}, [0] * plan.L
        for k, lev in enumerate(plan.levels):
            for t in lev:
                if t.name in ("l_wrist", "r_wrist", "com"):
                    where[t.name] = (st.A[k], off[k])
                if not t.implicit:
                    off[k] += t.rows
        fj = {0: where["l_wrist"], 1: where["r_wrist"], 2: (st.C, 0), 3: (st.C, 6)}
        kw = dict(frame_pose={f: pose[f] for f in range(4)}, frame_J=fj, com=com, com_J=where["com"])
        K.forward(q, **kw); torch.cuda.synchronize()
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
                if fused:
                    st.control_cycle(K, kb, leaf, q_integrate=q)
                else:
                    K.forward(q, **kw); st.cycle(leaf, cached=True); q.add_(st.dq[:Bl])
        work.append((st, step, stream, Bl))
    for _ in range(warmup):
        for _, step, _, _ in work: step()
    torch.cuda.synchronize()
    graphs = []
    if graph:
        try:
            for st, step, stream, _ in work:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream):
                    for _ in range(steps): step()
                graphs.append(g)
        except Exception as e:
            print("graph capture failed:", str(e)[:200]); graphs = []
    torch.cuda.synchronize()
    def run_all():
        if graphs:
            for g, (_, _, stream, _) in zip(graphs, work):
                with torch.cuda.stream(stream): g.replay()
        else:
            for _ in range(steps):
                for _, step, _, _ in work: step()
    run_all(); torch.cuda.synchronize()
    t0 = time.perf_counter(); run_all(); torch.cuda.synchronize(); el = time.perf_counter() - t0
    ok = sum(int((st.status[:Bl] == 0).sum().item()) for st, _, _, Bl in work)
    return B * steps / el, 1e3 * el / steps, ok

for which in ("S1", "S3"):
    for lanes, fused, graph in ((1, False, False), (1, True, True), (2, True, True), (4, True, True), (2, False, True)):
        v, ms, ok = run(which, 4096, lanes, fused=fused, graph=graph)
        print(which, "lanes", lanes, "fused" if fused else "three launches", "graph" if graph else "plain", round(v / 1e6, 3), "M", round(ms, 4), "ms", ok)

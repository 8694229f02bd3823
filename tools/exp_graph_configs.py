#!/usr/bin/env python3
"""developer experiment (GPU box): update + solve of the other configurations / front-ends replayed from a HIP graph"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opensot_amd import synth
from opensot_amd.solver import BatchedStack

def run(name, B, fn_name="solve", G=10, steps=40):
    plan, leaf = synth.make_id_stack(B, seed=5000) if name == "C5" else synth.make_velocity_stack(name, B, seed=2000)
    st = BatchedStack(plan, B, device=0, want_levels=False)
    dev = st.load_leaf(leaf)
    fn = getattr(st, fn_name)
    step = lambda: (st.update(dev), fn(B))
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    plain = B * steps / (time.perf_counter() - t0)
    dq0 = st.dq[:B].clone()
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(G): step()
        torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps // G): g.replay()
        torch.cuda.synchronize()
        gr = B * steps / (time.perf_counter() - t0)
        print(f"{name} {fn_name} B={B}: plain {plain/1e6:.3f} M, graph {gr/1e6:.3f} M, same dq {bool(torch.equal(dq0, st.dq[:B]))}")
    except Exception as e:
        print(f"{name} {fn_name}: plain {plain/1e6:.3f} M, graph failed: {repr(e)[:200]}")

run("C2", 1024); run("C4", 4096); run("C5", 1024); run("C3", 4096, "solve_ehqp"); run("C3", 4096, "solve_nhqp", G=4, steps=8)

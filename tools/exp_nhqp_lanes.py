"""developer helper (GPU box): sub-batch count of the nHQP sub-lines (a launch of B / lanes instances against the kernels' resident
wavefront slots: 1536 for the 32-wide preparation, 1024 for the 64-column one, 768 for the wide one)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
streams = [torch.cuda.Stream(device=torch.device("cuda", 0)) for _ in range(6)]
for lanes in (2, 3):
    r = bench.time_nhqp(4096, 0, steps=10, warmup=3, lanes=lanes, streams=streams[:lanes])
    print("nHQP C3 lanes", lanes, round(r["value"] / 1e6, 3), "M", r["solved_ok"], flush=True)
for which in ("S1", "S2", "S3", "S4"):
    for lanes in (2, 3, 4, 6):
        r = bench.time_coman35(which, 4096, 0, steps=6, warmup=2, front_end="nHQP", lanes=lanes, streams=streams[:lanes])
        print("COMAN35", which, "nHQP lanes", lanes, round(r["value"] / 1e6, 3), "M", r.get("solved_ok"), flush=True)

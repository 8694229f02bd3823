"""developer helper (GPU box): sub-batch count of the full_cycle and config-5 sub-lines against the kernels' resident wavefronts"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
streams = [torch.cuda.Stream(device=torch.device("cuda", 0)) for _ in range(4)]
for lanes in (2, 3, 4):
    r = bench.time_full_cycle(4096, 0, lanes=lanes, streams=streams[:lanes])
    print("full_cycle lanes", lanes, round(r["value"] / 1e6, 3), "M", r.get("solved_ok"), flush=True)
for lanes in (2, 3, 4):
    r = bench.time_config("C5", 4096, 0, steps=12, lanes=lanes, streams=streams[:lanes])
    print("C5_B4096 lanes", lanes, round(r["value"] / 1e6, 3), "M", r.get("solved_ok"), flush=True)
for lanes in (2, 3):
    r = bench.time_config("C4", 4096, 0, steps=20, lanes=lanes, streams=streams[:lanes])
    print("C4 lanes", lanes, round(r["value"] / 1e6, 3), "M", r.get("solved_ok"), flush=True)

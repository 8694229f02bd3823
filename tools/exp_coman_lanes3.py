"""developer helper (GPU box): sub-batch count of the COMAN35 iHQP sub-lines (the 40-lane kernel holds 7 wavefronts per CU -- 22 KB of
LDS each -- 1792 at once: 2048 instances are one round and a seventh, 1365 one round)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
streams = [torch.cuda.Stream(device=torch.device("cuda", 0)) for _ in range(4)]
for which in ("S1", "S2", "S3", "S4"):
    for lanes in (2, 3, 4):
        r = bench.time_coman35(which, 4096, 0, lanes=lanes, streams=streams[:lanes])
        print("COMAN35", which, "iHQP lanes", lanes, round(r["value"] / 1e6, 3), "M", r.get("solved_ok"), flush=True)

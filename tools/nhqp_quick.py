"""developer helper (GPU box): the null-space front-end at BASELINE config 3 (one stream, two sub-batches) and on the reference's
COMAN35 stacks S1..S4 (closed loops of 4096 robots)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
streams = [torch.cuda.Stream(device=torch.device("cuda", 0)) for _ in range(2)]
for lanes in (1, 2):
    r = bench.time_nhqp(4096, 0, steps=10, warmup=3, lanes=lanes, streams=streams)
    print("nHQP C3 lanes", lanes, round(r["value"] / 1e6, 3), "M", round(r["ms_per_step"], 4), "ms", r["solved_ok"], flush=True)
for which in (sys.argv[1:] or ("S1", "S2", "S3", "S4")):
    r = bench.time_coman35(which, 4096, 0, 10, 3, front_end="nHQP")
    print("COMAN35", which, "nHQP", round(r["value"] / 1e6, 3), "M", round(r["ms_per_step"], 4), "ms", r.get("solved_ok"), flush=True)

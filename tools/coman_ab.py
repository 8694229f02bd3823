"""developer helper (GPU box): the reference's COMAN stacks S1..S4 as closed loops of 4096 robots -- sub-batches (lanes), the
one-launch control cycle against three launches per step, the instantiation by plan structure on and off"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
streams = [torch.cuda.Stream(device=torch.device("cuda", 0)) for _ in range(4)]
for which in ("S1", "S2", "S3", "S4"):
    for lanes, fused, spec, graph in ((1, False, True, False), (1, True, True, True), (2, True, True, True), (2, False, True, True), (4, True, True, True), (2, True, False, True)):
        r = bench.time_coman35(which, 4096, 0, 20, 5, specialise=spec, fused=fused, lanes=lanes, streams=streams, graph=graph)
        print(which, "lanes", lanes, "one launch" if fused else "three launches", "graph" if graph else "plain", "specialised" if spec else "general",
              round(r["value"] / 1e6, 3), "M", round(r["ms_per_step"], 4), "ms", r.get("solved_ok"))

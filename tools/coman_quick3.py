"""developer helper (GPU box): COMAN35 S1..S4 as the bench line submits them (THREE sub-batches, one launch per step, graphs), twice each;
OSOT_MI355X_LIB names the library"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
streams = [torch.cuda.Stream(device=torch.device("cuda", 0)) for _ in range(3)]
for which in (sys.argv[1:] or ("S2", "S3", "S4")):
    for rep in range(2):
        r = bench.time_coman35(which, 4096, 0, 20, 5, lanes=3, streams=streams)
        print(os.path.basename(os.environ.get("OSOT_MI355X_LIB", "default")), which, round(r["value"] / 1e6, 3), "M", round(r["ms_per_step"], 4), "ms", r.get("solved_ok"), flush=True)

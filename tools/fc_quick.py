import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
streams = [torch.cuda.Stream(device=torch.device("cuda", 0)) for _ in range(2)]
tag = os.path.basename(os.environ.get("OSOT_MI355X_LIB", "default"))
for rep in range(2):
    r = bench.time_full_cycle(4096, 0, lanes=2, steps=40, streams=streams)
    print(tag, "full_cycle", round(r["value"] / 1e6, 2), "M", flush=True)
for which in ("S1", "S3"):
    r = bench.time_coman35(which, 4096, 0, 20, 5, lanes=2, streams=streams)
    print(tag, which, round(r["value"] / 1e6, 3), "M", flush=True)

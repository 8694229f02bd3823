import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
streams = [torch.cuda.Stream(device=torch.device("cuda", 0)) for _ in range(2)]
for g in (False, True, False, True):
    r = bench.time_nhqp(4096, 0, steps=10, warmup=3, lanes=2, streams=streams, graph=g)
    print("graph", g, round(r["value"] / 1e6, 3), "M", r["solved_ok"], flush=True)

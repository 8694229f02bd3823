"""developer helper (GPU box): the full control cycle (32-DoF humanoid under config 3's stack, 4096 robots in closed loop) as one
launch per step against ROLLOUTS of K control cycles per launch (osot_control_rollout), and update + cascade alone on the loop's
own problems (the yardstick)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
streams = [torch.cuda.Stream(device=torch.device("cuda", 0)) for _ in range(2)]
r = bench.time_full_cycle(4096, 0, lanes=2, steps=40, streams=streams, solve_only=True)
print("update + cascade alone", round(r["value"] / 1e6, 2), "M", flush=True)
for lanes in (2, 1):
    for K in (1, 2, 5, 10, 20, 40):
        r = bench.time_full_cycle(4096, 0, lanes=lanes, steps=40, streams=streams, rollout=K)
        print("lanes", lanes, "rollout", K, round(r["value"] / 1e6, 2), "M", round(r["ms_per_step"], 4), "ms", r["solved_ok"], flush=True)

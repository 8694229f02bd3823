"""developer helper: time the batched kinematics producer (osot_kinematics) and report its HBM rate"""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from opensot_amd import kinematics as kin
m = kin.humanoid32()
K = kin.Kinematics(m, device=0)
for B in (4096, 32768):
    dev = torch.device("cuda", 0)
    q = torch.as_tensor(np.random.default_rng(1).uniform(-1, 1, (B, m.n)), device=dev)
    A1 = torch.zeros((B, 24, m.n), dtype=torch.float64, device=dev)
    A0 = torch.zeros((B, 3, m.n), dtype=torch.float64, device=dev)
    poses = {f: torch.zeros((B, 12), dtype=torch.float64, device=dev) for f in range(4)}
    com = torch.zeros((B, 3), dtype=torch.float64, device=dev)
    run = lambda: K.forward(q, frame_pose=poses, frame_J={f: (A1, 6 * f) for f in range(4)}, com=com, com_J=(A0, 0))
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    bytes_ = B * (m.n * 8 + 27 * m.n * 8 + 4 * 96 + 24)
    print(f"B={B}: {us:.1f} us per call, {bytes_ / us / 1e3:.0f} GB/s of {bytes_ / 1e6:.1f} MB algorithmic traffic, {B / us:.1f} M instances/s")

#!/usr/bin/env python3
"""developer micro-benchmark (GPU box): streaming WRITE, READ and COPY rates of the HBM as torch's own kernels reach them --
the yardstick for a kernel that is nearly all stores (the kinematics producer writes 30 x what it reads)."""
import torch
dev = torch.device("cuda", 0)
for mb in (32, 256, 2048):
    n = mb * 1024 * 1024 // 8
    x = torch.empty(n, dtype=torch.float64, device=dev)
    y = torch.empty(n, dtype=torch.float64, device=dev)
    def t(f, reps=20):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3
    tw = t(lambda: x.fill_(1.0))
    tr = t(lambda: x.sum())
    tc = t(lambda: y.copy_(x))
    print(f"{mb} MB: write {mb/1024/tw/1e3*1.0737:.2f} TB/s, read {mb/1024/tr/1e3*1.0737:.2f} TB/s, copy {2*mb/1024/tc/1e3*1.0737:.2f} TB/s (read + write)")

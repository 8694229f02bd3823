"""developer helper (GPU box): the nHQP / eHQP front-ends at BASELINE config 3 (run under rocprofv3 --kernel-trace --stats for
the per-kernel split)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import bench
for name, fn in (("nHQP_C3", bench.time_nhqp), ("eHQP_C3", bench.time_ehqp)):
    r = fn(4096, 0)
    print(name, json.dumps({k: v for k, v in r.items() if k != "workload"}))

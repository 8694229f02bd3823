"""developer helper (GPU box): the nHQP / eHQP front-ends at BASELINE config 3 and the ADMM back-end (run under rocprofv3
--kernel-trace --stats for the per-kernel split; OSOT_MI355X_LIB=tools/bin/NAME.so for an A/B against a variant build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import bench
for name, fn, B in (("nHQP_C3", bench.time_nhqp, 4096), ("eHQP_C3", bench.time_ehqp, 4096), ("ADMM_qp", bench.time_admm, 1024)):
    r = fn(B, 0)
    print(name, json.dumps({k: v for k, v in r.items() if k != "workload"}))

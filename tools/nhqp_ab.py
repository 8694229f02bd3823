import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, torch
r = bench.time_nhqp(4096, 0)
print(os.path.basename(os.environ.get("OSOT_MI355X_LIB","default")), "nHQP", round(r["value"]), "ms/step", r.get("ms_per_step"))
try:
    r = bench.time_ehqp(4096, 0)
    print("  eHQP", round(r["value"]))
except Exception as e:
    print("ehqp err", e)

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
r = bench.time_admm(1024, 0)
print(os.path.basename(os.environ.get("OSOT_MI355X_LIB", "default")), "ADMM", {k: (round(v) if isinstance(v, float) else v) for k, v in r.items() if k in ("value", "cold_value", "iterations_per_solve", "cold_iterations_per_solve")})

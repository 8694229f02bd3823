"""developer helper (GPU box): the reference's COMAN stacks S1..S4 as closed loops of 4096 robots, with and without the
instantiation by plan structure (osot_solver_set_specialisation)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for rep in range(2):
    for which in ("S1", "S2", "S3", "S4"):
        a = bench.time_coman35(which, 4096, 0, 20, 5, specialise=True)
        b = bench.time_coman35(which, 4096, 0, 20, 5, specialise=False)
        c = bench.time_coman35(which, 4096, 0, 20, 5, specialise=True, fused=False)
        print(which, "one launch, specialised", round(a["value"] / 1e6, 3), "M   general", round(b["value"] / 1e6, 3), "M   three launches, specialised",
              round(c["value"] / 1e6, 3), "M", a.get("solved_ok"), b.get("solved_ok"), c.get("solved_ok"))

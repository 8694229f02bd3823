"""developer A/B (GPU box): the null-space front-end at BASELINE config 3 with 2 / 3 / 4 sub-batches, and COMAN35 S3 through nHQP with 2 / 3,
for the libraries given as arguments ("default" first)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import torch
    import bench
    from opensot_amd import synth
    from opensot_amd.solver import BatchedStack
    lib = os.path.basename(os.environ.get("OSOT_MI355X_LIB", "default"))
    streams = [torch.cuda.Stream(device=torch.device("cuda", 0)) for _ in range(4)]
    probe = BatchedStack(synth.make_velocity_stack("C3", 1, seed=1)[0], 1, device=0, want_levels=False)
    try:
        print(lib, "resident: cascade", probe.resident_waves(), "nHQP preparation", probe.resident_waves_nhqp(), flush=True)
    except Exception as e:
        print(lib, "resident query:", e, flush=True)
    for lanes in (2, 3, 4):
        r = bench.time_nhqp(4096, 0, steps=10, warmup=3, lanes=lanes, streams=streams)
        print(lib, "nHQP C3 lanes", lanes, round(r["value"] / 1e6, 3), "M", round(r["ms_per_step"], 4), "ms", r["solved_ok"], flush=True)
    for lanes in (2, 3):
        r = bench.time_coman35("S3", 4096, 0, 6, 2, front_end="nHQP", lanes=lanes, streams=streams)
        print(lib, "COMAN35 S3 nHQP lanes", lanes, round(r["value"] / 1e6, 3), "M", round(r["ms_per_step"], 4), "ms", r.get("solved_ok"), flush=True)
    sys.exit(0)
for lib in ["default"] + sys.argv[1:]:
    env = dict(os.environ)
    if lib != "default":
        env["OSOT_MI355X_LIB"] = os.path.abspath(lib)
    subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env)

"""developer A/B (GPU box): BASELINE config 5 as the bench line times it (bench.time_config: two sub-batches, the headline's four drifting
cycles, graphs) at its shard size and at 4096, for the libraries given as arguments ("default" first)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import torch
    import bench
    streams = [torch.cuda.Stream(device=torch.device("cuda", 0)) for _ in range(2)]
    lib = os.path.basename(os.environ.get("OSOT_MI355X_LIB", "default"))
    for B in (1024, 4096):
        for rep in range(2):
            r = bench.time_config("C5", B, 0, steps=20, lanes=2, streams=streams)
            print(lib, "C5", B, f"{r['value'] / 1e6:.3f} M/s", f"{r['ms_per_step']:.4f} ms", r.get("solved_ok"), flush=True)
    sys.exit(0)
for lib in ["default"] + sys.argv[1:]:
    env = dict(os.environ)
    if lib != "default":
        env["OSOT_MI355X_LIB"] = os.path.abspath(lib)
    subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env)

"""developer helper (GPU box): COMAN35 S1..S4 as the bench line submits them (three sub-batches, one launch per step, graphs) for the
libraries given as arguments ("default" = the tree's); each stack twice"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:] or ["default"]
for lib in libs:
    env = dict(os.environ)
    if lib != "default":
        env["OSOT_MI355X_LIB"] = os.path.abspath(lib)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "coman_quick3.py"), "S1", "S2", "S3", "S4"], env=env)

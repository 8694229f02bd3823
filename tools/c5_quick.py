import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from opensot_amd import synth
from opensot_amd.solver import BatchedStack
for cfg, B in (("C5", 1024), ("C5", 4096)):
    plan, leaf = synth.make_id_stack(B, seed=1)
    st = BatchedStack(plan, B, device=0, want_levels=False)
    dev = st.load_leaf(leaf)
    for _ in range(3):
        st.cycle(dev)
    torch.cuda.synchronize()
    st.set_timing(True)
    for _ in range(20):
        st.cycle(dev)
    torch.cuda.synchronize()
    ms, cnt = st.kernel_time_ms()
    print(os.path.basename(os.environ.get("OSOT_MI355X_LIB", "default")), cfg, B, f"{ms*1e3:.1f} us/launch", flush=True)

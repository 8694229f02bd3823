"""developer helper (GPU box): phase census (shader clocks per wavefront, the instrumented cascade instantiation) of the reference's COMAN
stacks on the 35-coordinate robot AT THE POSTURE THE CLOSED LOOP REACHED after the bench's steps.  usage: prof_phases_coman.py [S3 ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from opensot_amd import kinematics as kin

for which in (sys.argv[1:] or ("S3", "S4")):
    B = 1024
    captured = {}
    orig = torch.cuda.synchronize
    # run the bench's own closed loop (one lane, plain launches), then take the lane's state from its step closure
    import opensot_amd.solver as solver
    stacks = []
    real_init = solver.BatchedStack.__init__
    def spy(self, *a, **k):
        real_init(self, *a, **k); stacks.append(self)
    solver.BatchedStack.__init__ = spy
    kins = []
    real_kinit = kin.Kinematics.__init__
    def kspy(self, *a, **k):
        real_kinit(self, *a, **k); kins.append(self)
    kin.Kinematics.__init__ = kspy
    leafs = []
    real_cc = solver.BatchedStack.control_cycle
    def ccspy(self, K, kb, leaf, **kw):
        leafs.append((K, kb, leaf, kw)); return real_cc(self, K, kb, leaf, **kw)
    solver.BatchedStack.control_cycle = ccspy
    r = bench.time_coman35(which, B, 0, steps=20, warmup=5, lanes=1, streams=None, graph=False)
    solver.BatchedStack.__init__ = real_init; kin.Kinematics.__init__ = real_kinit; solver.BatchedStack.control_cycle = real_cc
    st = stacks[-1]
    K, kb, leaf, kw = leafs[-1]
    # the same problem through the three calls, then the instrumented cascade on the assembled arrays
    q = kw["q_integrate"]
    import ctypes as C
    from opensot_amd import abi
    abi.check(K._lib.osot_kinematics(K._h, C.byref(kb), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "osot_kinematics")
    st.stream = None
    st.update(leaf); st.solve(B); torch.cuda.synchronize()
    cyc = st.profile_phases(B)
    m = cyc.mean(axis=0)
    print(f"== COMAN35 {which}: {r['value'] / 1e6:.2f} M solves/s (one lane of {B}, plain launches), rows per level {r['rows_per_level']}, {r['constraint_rows']} constraint rows")
    for name, v in zip(st.PHASES, m):
        print(f"{name:16s} {v:10.0f} cycles  {100 * v / m[7]:5.1f}%")
    it = st.iterations[:B].cpu().numpy()
    t = cyc[:, 7]
    print("total cycles per instance: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % (t.mean(), np.percentile(t, 50), np.percentile(t, 90), np.percentile(t, 99), t.max()))
    print("iterations: mean %.1f p50 %d p90 %d p99 %d max %d" % (it.mean(), np.percentile(it, 50), np.percentile(it, 90), np.percentile(it, 99), it.max()))

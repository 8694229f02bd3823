"""opensot_amd.parallel.suggest_lanes: the library-side choice of sub-batches (bench.py carried hand-tuned counts per
sub-line).  The rule against the cases rounds 4-5 worked out on hardware, and a PipelinedCycle built from it on the stub back-end."""
import numpy as np

from opensot_amd import synth
from opensot_amd.parallel import PipelinedCycle, ShardedCycle, StubStack, lane_ranges, suggest_lanes


def test_rule_reproduces_the_measured_choices():
    assert suggest_lanes(4096, 2048) == 2      # BASELINE config 3 on the 32-lane kernel: two launches of exactly one round
    assert suggest_lanes(4096, 1792) == 3      # 35-coordinate COMAN stacks, 40-lane kernel (22 KB LDS): 1365 = one round (exp_coman_lanes3.py)
    assert suggest_lanes(4096, 1536) == 3      # nHQP at config 3: the 32-wide preparation's 25 KB of LDS (exp_nhqp_lanes.py)
    assert suggest_lanes(4096, 1024) == 2      # nHQP on the 64-column preparation: 2048 = exactly two rounds
    assert suggest_lanes(1024, 1024) == 2      # config 5 at its shard size: never fewer than two lanes
    assert suggest_lanes(32768, 2048) == 2
    assert suggest_lanes(3, 2048) == 1         # nothing to split
    # whole rounds win: no choice wastes less than the one taken, and ties go to the fewer lanes
    rng = np.random.default_rng(0)
    for _ in range(200):
        B, R = int(rng.integers(8, 40000)), int(rng.integers(64, 4096))
        S = suggest_lanes(B, R)
        waste = lambda s: 1.0 - (-(-B // s)) / float(-(-(-(-B // s)) // R) * R)
        assert 2 <= S <= 4 and all(waste(S) <= waste(s) + 1e-9 for s in (2, 3, 4))
        assert all(waste(s) > waste(S) + 1e-9 for s in range(2, S))


def test_pipelined_cycle_from_the_rule_on_the_stub_backend():
    B = 96
    plan, leaf = synth.make_velocity_stack("C3", B, seed=5)
    S = suggest_lanes(B, 40)                   # 96 instances on 40 slots: 3 lanes of 32 (one round each) beat 2 of 48 (two rounds)
    assert S == 3
    lanes = []
    whole = StubStack(plan, B)
    whole.cycle(whole.load_leaf(leaf))
    for a, b in lane_ranges(B, S):
        st = StubStack(plan, b - a)
        cut = lambda x: None if x is None else x[a:b]
        lf = {"B": b - a, "A": [cut(x) for x in leaf["A"]]}
        lanes.append(ShardedCycle(st, [st.load_leaf(lf)], None, b - a))
    PipelinedCycle(lanes).step()
    got = np.concatenate([ln.stack.dq[:ln.B].numpy() for ln in lanes])
    assert np.array_equal(got, whole.dq[:B].numpy())

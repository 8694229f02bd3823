"""developer helper (GPU box): per-wave timeline of osot_cycle: shader-clock cycles of the update / cascade halves and
constant-rate (100 MHz) start / end stamps -> when waves start, when the slowest ends, how long the launch's tail is"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
buf = torch.zeros((B, 4), dtype=torch.int64, device="cuda")
os.environ["OSOT_DEBUG_CYCLE_PROF"] = hex(buf.data_ptr())
from opensot_amd import synth
from opensot_amd.solver import BatchedStack
plan, leaf = synth.make_velocity_stack("C3", B, seed=3000)
st = BatchedStack(plan, B, device=0, want_levels=False)
dev = st.load_leaf(leaf)
for _ in range(4):
    st.cycle(dev)
torch.cuda.synchronize()
c = buf.cpu().numpy()
print("resident waves:", st.resident_waves())
print("update half: mean %.0f max %.0f cycles; cascade half: mean %.0f max %.0f" % (c[:, 0].mean(), c[:, 0].max(), c[:, 1].mean(), c[:, 1].max()))
t0 = c[:, 2].min()
s = (c[:, 2] - t0) / 100.0; e = (c[:, 3] - t0) / 100.0     # microseconds
dur = e - s
print("launch span %.1f us; wave durations: mean %.1f max %.1f us  => shader clock ~ %.2f GHz" % (e.max(), dur.mean(), dur.max(), (c[:, 0] + c[:, 1]).mean() / dur.mean() / 1e3))
ss = np.sort(s)
print("start times (us): first 2048 waves by %.1f; wave #2049 at %.1f; median of the second half %.1f; last start %.1f" % (ss[min(2047, B - 1)], ss[min(2048, B - 1)], np.median(ss[B // 2:]), ss[-1]))
i = np.argmax(e)
print("last wave to end: started %.1f ran %.1f us (%d iterations); the longest wave: started %.1f ran %.1f us" % (s[i], dur[i], int(st.iterations[i]), s[np.argmax(dur)], dur.max()))


def list_schedule(order, d, slots):
    """finish time of list scheduling: jobs taken in `order`, each onto the slot that frees first"""
    import heapq
    free = [0.0] * slots
    heapq.heapify(free)
    end = 0.0
    for j in order:
        t = heapq.heappop(free) + d[j]
        end = max(end, t)
        heapq.heappush(free, t)
    return end


slots = st.resident_waves()
print("packing: sum(dur)/slots %.1f us, longest %.1f us; list schedule in the order the waves started %.1f us, longest-first with the TRUE durations %.1f us, by iteration count %.1f us"
      % (dur.sum() / slots, dur.max(), list_schedule(np.argsort(s, kind="stable"), dur, slots), list_schedule(np.argsort(-dur, kind="stable"), dur, slots),
         list_schedule(np.argsort(-st.iterations[:B].cpu().numpy(), kind="stable"), dur, slots)))
first = s < 5.0
print("update half by round: first %d waves mean %.0f cycles, later ones mean %.0f; cascade half: %.0f / %.0f"
      % (first.sum(), c[first, 0].mean(), c[~first, 0].mean(), c[first, 1].mean(), c[~first, 1].mean()))

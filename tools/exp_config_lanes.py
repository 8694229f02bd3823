"""developer experiment (GPU box): the BASELINE sub-lines with the headline's submission (sub-batches on their own streams, graphs)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
streams = [torch.cuda.Stream(device=torch.device('cuda', 0)) for _ in range(4)]
for name, B, st in (("C5", 1024, 20), ("C2", 1024, 20), ("C4", 4096, 20), ("C5", 4096, 12)):
    a = bench.time_config(name, B, 0, steps=st)
    out = [round(a['value'] / 1e6, 3)]
    for lanes in (2, 4):
        b = bench.time_config(name, B, 0, steps=st, lanes=lanes, streams=streams)
        out.append((lanes, round(b['value'] / 1e6, 3), b['solved_ok'], b.get('note_capture')))
    print(name, B, 'single', out)

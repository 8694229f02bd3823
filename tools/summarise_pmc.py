"""developer helper: per-launch averages of the rocprofv3 --pmc passes for the cascade kernel (see profile_round.sh)"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out = sys.argv[1]
per = {}
kname = None
for f in sorted(glob.glob(out + "/pmc*_counters.csv")):
    rows = list(csv.DictReader(open(f)))
    acc, cnt = {}, {}
    for r in rows:
        if "osot_cycle_kernel" not in r.get("Kernel_Name", ""):      # the bench's step: update + cascade in one launch
            continue
        kname = r["Kernel_Name"]
        c, v = r["Counter_Name"], float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), c)
        acc[key] = acc.get(key, 0.0) + v          # sum over XCDs / instances of one dispatch
    byc = {}
    for (d, c), v in acc.items():
        byc.setdefault(c, []).append(v)
    for c, vs in byc.items():
        per[c] = sum(vs) / len(vs)
res = {"kernel": kname, "workload": "C3 B=4096 (bench.py --steps 10 --warmup 2, one counter group per pass)", "per_launch": per,
       # bench.py reports this traffic figure only while the kernel sources still hash to this value
       "kernel_source_sha": bench.kernel_source_sha(), "config": "C3", "batch": 4096}
if "FETCH_SIZE" in per:
    res["hbm_bytes_per_launch_corrected"] = per["FETCH_SIZE"] * 1024 * 2 + per.get("WRITE_SIZE", 0.0) * 1024
    res["correction"] = ("MI355X_MICROARCH.md HBM section: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the "
                         "bytes of a coalesced stream, so it is doubled; WRITE_SIZE taken as is; separate --pmc passes")
res["algorithmic_bytes_per_launch"] = 4096 * ((3 + 24) * 32 * 8 + 59 * 8 + 59 * 8 + 2 * 32 * 8 + 32 * 8)
res["note"] = ("osot_cycle_kernel = AutoStack::update + cascade of an instance by one wavefront: its traffic also holds the leaf inputs "
               "(poses, q, limits: ~2.3 KB per instance) and the assembled b / w / box it writes and reads back (~1.5 KB)")
print(json.dumps(res, indent=1))

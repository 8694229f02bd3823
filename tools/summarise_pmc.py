"""developer helper: per-launch averages of the rocprofv3 --pmc passes, per (kernel, grid size) -- see profile_round.sh.
usage: summarise_pmc.py <dir with pmc*_counters.csv> > profiles/rNN_pmc_kernels.json"""
import csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out = sys.argv[1]
table = {}      # (kernel, grid) -> counter -> [per-dispatch sums]
for f in sorted(glob.glob(out + "/pmc*_counters.csv")):
    acc = {}
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "osot_" not in k:
            continue
        k = re.sub(r"\s*\[clone .*\]$", "", k)
        key = (k, int(r.get("Grid_Size", 0) or 0), r.get("Dispatch_Id"), r["Counter_Name"])
        acc[key] = acc.get(key, 0.0) + float(r["Counter_Value"])      # sum over XCDs / shader engines of one dispatch
    for (k, g, d, c), v in acc.items():
        table.setdefault((k, g), {}).setdefault(c, []).append(v)
rows = []
for (k, g), cs in sorted(table.items()):
    per = {c: sum(v) / len(v) for c, v in cs.items()}
    row = {"kernel": k, "grid_threads": g, "workgroups": g // 64, "launches": max(len(v) for v in cs.values()), "per_launch": per}
    if "FETCH_SIZE" in per:
        row["hbm_bytes_per_launch_corrected"] = per["FETCH_SIZE"] * 1024 * 2 + per.get("WRITE_SIZE", 0.0) * 1024
    rows.append(row)
print(json.dumps({"kernel_source_sha": bench.kernel_source_sha(),
                  "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of "
                                "the bytes of a coalesced stream, so it is doubled; WRITE_SIZE taken as is; separate --pmc passes, "
                                "--kernel-trace only; averages over the launches of a (kernel, grid) pair in `python bench.py "
                                "--no-cpu-baseline --steps 10 --warmup 2`",
                  "kernels": rows}, indent=1))

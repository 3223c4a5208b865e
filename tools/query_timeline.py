"""Timeline of ONE steady-state query from a `rocprofv3 --kernel-trace --output-format csv` directory of bench.py: every launch
with its start (us from the query's first launch), duration, queue, workgroups.   usage: python tools/query_timeline.py <dir> [step]"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
rows = [r for r in csv.DictReader(open(f)) if "apsu_he" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "k_eval_epilogue" in r["Kernel_Name"]]
a, b = ends[which - 1] + 1, ends[which] + 1
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
qs = {}
for r in step:
    q = qs.setdefault(r["Queue_Id"], len(qs))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "Grid_Size_X" in r:
        wg = 1
        for d in "XYZ": wg *= max(1, int(r["Grid_Size_" + d]) // max(1, int(r["Workgroup_Size_" + d])))
    else:
        wg = int(r.get("Grid_Size", 0)) // max(1, int(r.get("Workgroup_Size", 1)))
    print("%8.1f %8.1f  q%d %s%-40s wg %6d" % ((s - t0) / 1e3, (e - s) / 1e3, q, "    " * q, r["Kernel_Name"].split("(")[0].replace("void apsu_he::", "")[:40], wg))
print("wall %.1f us" % ((max(int(r["End_Timestamp"]) for r in step) - t0) / 1e3))

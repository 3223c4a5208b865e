"""Idle time between kernels inside ONE steady-state query step, from a `rocprofv3 --kernel-trace --output-format csv`
directory of `bench.py --no-profile` (launch start / end timestamps, both engine streams merged): what a hipGraph of the
per-query launch sequence could recover at most.   usage: python tools/launch_gaps.py <dir>"""
import collections, csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "apsu_he" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "k_eval_epilogue" in r["Kernel_Name"]]
for which in (-1, -2):
    a, b = ends[which - 1] + 1, ends[which] + 1
    step = rows[a:b]
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
    t0, t1 = iv[0][0], max(e for _, e in iv)
    busy, gaps, ue = 0, [], iv[0][0]
    for s, e in iv:
        if s > ue:
            gaps.append(s - ue)
        if e > ue:
            busy += e - max(s, ue)
            ue = e
    print("step %d: %d launches, wall %.1f us, some kernel running %.1f us, idle %.1f us (%.2f %%), mean gap %.2f us, max gap %.1f us"
          % (which, len(step), (t1 - t0) / 1e3, busy / 1e3, sum(gaps) / 1e3, 100.0 * sum(gaps) / (t1 - t0),
             sum(gaps) / max(1, len(gaps)) / 1e3, max(gaps or [0]) / 1e3))
c = collections.Counter()
for r in step:
    c[r["Kernel_Name"].split("(")[0].replace("void apsu_he::", "")[:44]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, v in c.most_common():
    print("  %-46s %8.1f us" % (k, v))

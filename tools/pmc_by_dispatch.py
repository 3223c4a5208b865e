"""Per-DISPATCH rows of a `rocprofv3 --kernel-trace --pmc ...` run for one kernel: dispatch order, duration, every counter.
usage: python tools/pmc_by_dispatch.py <dir> <kernel-substring> [min grid size]"""
import collections, csv, glob, os, sys
d, want = sys.argv[1], sys.argv[2]
min_grid = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for ccf in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    ktf = ccf.replace("counter_collection", "kernel_trace")
    dur = {}
    for r in csv.DictReader(open(ktf)) if os.path.exists(ktf) else []:
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(ccf)):
        if want not in r["Kernel_Name"] or int(r["Grid_Size"]) < min_grid:
            continue
        e = disp.setdefault(int(r["Dispatch_Id"]), collections.defaultdict(float))
        e[r["Counter_Name"]] += float(r["Counter_Value"])
    names = sorted({k for e in disp.values() for k in e})
    print("# " + ccf)
    print("%8s %10s  " % ("dispatch", "us") + "  ".join("%s" % n for n in names))
    for k in sorted(disp):
        print("%8d %10.1f  " % (k, dur.get(str(k), 0.0)) + "  ".join("%.6g" % disp[k][n] for n in names))

"""Per-kernel sums of a `rocprofv3 --kernel-trace --pmc ...` run (counter_collection.csv + kernel_trace.csv):
kernel, launches, total us, every counter summed, plus counter ratios given as A/B arguments.
usage: python tools/pmc_by_kernel.py <dir> [A/B ...]"""
import collections, csv, glob, os, sys
d = sys.argv[1]
ratios = [a.split("/") for a in sys.argv[2:]]
for ccf in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    ktf = ccf.replace("counter_collection", "kernel_trace")
    dur = {}
    for r in csv.DictReader(open(ktf)) if os.path.exists(ktf) else []:
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg = collections.OrderedDict()
    seen = set()
    for r in csv.DictReader(open(ccf)):
        name = r["Kernel_Name"].split("(")[0].replace("void apsu_he::", "")[:44]
        e = agg.setdefault(name, {"n": 0, "us": 0.0, "c": collections.defaultdict(float)})
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); e["n"] += 1; e["us"] += dur.get(r["Dispatch_Id"], 0.0)
        e["c"][r["Counter_Name"]] += float(r["Counter_Value"])
    for name, e in sorted(agg.items(), key=lambda kv: -kv[1]["us"])[:24]:
        c = e["c"]
        extra = "  ".join("%s/%s=%.4g" % (a, b, c[a] / c[b]) for a, b in ratios if c.get(b))
        print("%-44s n=%4d %9.1f us  " % (name, e["n"], e["us"]) + "  ".join("%s=%.3g" % kv for kv in sorted(c.items())) + "  | " + extra)

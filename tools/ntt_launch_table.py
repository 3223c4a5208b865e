"""Per-launch efficiency of the NTT kernels from a `rocprofv3 --kernel-trace --output-format csv` directory:
for every (kernel, limbs per launch) the launch count, mean duration and achieved algorithmic GB/s (16*n bytes per limb).
usage: python tools/ntt_launch_table.py <dir> [n]"""
import csv, glob, os, sys, collections
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append(r)
agg = collections.defaultdict(list)
other = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3      # us
    short = name.split("(")[0].replace("void apsu_he::", "")
    if "k_ntt" in name or "k_intt" in name:
        wg = int(r["Workgroup_Size_X"] if "Workgroup_Size_X" in r else r["Workgroup_Size"])
        grid = int(r["Grid_Size_X"] if "Grid_Size_X" in r else r["Grid_Size"])
        agg[(short, grid // wg)].append(dur)
    else:
        other[short].append(dur)
tot_t = tot_b = 0
print("%-34s %8s %6s %10s %9s" % ("kernel", "limbs", "count", "mean us", "GB/s"))
for (k, limbs), ds in sorted(agg.items(), key=lambda kv: (kv[0][0], kv[0][1])):
    mean = sum(ds) / len(ds)
    gbs = limbs * 16 * n / (mean * 1e-6) / 1e9
    tot_t += sum(ds); tot_b += limbs * 16 * n * len(ds)
    print("%-34s %8d %6d %10.1f %9.0f" % (k, limbs, len(ds), mean, gbs))
if tot_t:
    print("all NTT launches: %.1f GB/s over %.2f ms" % (tot_b / (tot_t * 1e-6) / 1e9, tot_t / 1e3))
print()
for k, ds in sorted(other.items(), key=lambda kv: -sum(kv[1])):
    print("%-60s count %5d  total %9.1f us  mean %8.1f us" % (k[:60], len(ds), sum(ds), sum(ds) / len(ds)))

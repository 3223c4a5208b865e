#!/usr/bin/env python3
"""Turns gpurun_out/<tag>/ (tools/collect_profiles.sh) into the committed summaries under profiles/."""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = os.path.join("gpurun_out", tag), "profiles"
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, tag + "_bench.json"))
shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, tag + "_bench_under_rocprof.json"))
def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)          # gpurun merges every call's files: take the last run
shutil.copy(newest(os.path.join(src, "stats", "*", "*kernel_stats.csv")), os.path.join(dst, tag + "_rocprofv3_kernel_stats.csv"))
shutil.copy(os.path.join(src, "ntt_stream.txt"), os.path.join(dst, tag + "_ntt_stream.txt"))

def agg(path):
    rows = list(csv.DictReader(open(newest(path))))
    d = collections.defaultdict(lambda: [0.0, 0, 0])
    for r in rows:
        name = r["Kernel_Name"]
        key = "k_ntt_inv" if ("k_ntt" in name and "true" in name) else "k_ntt_fwd" if "k_ntt" in name else name.split("(")[0].split("::")[-1]   # (k_intt_tensor stays its own row)
        d[key][0] += float(r["Counter_Value"]); d[key][1] += 1; d[key][2] += int(r["Grid_Size"]) // int(r["Workgroup_Size"])
    return d
f = agg(os.path.join(src, "pmc_fetch", "*", "*counter_collection.csv"))
w = agg(os.path.join(src, "pmc_write", "*", "*counter_collection.csv"))
limbs = f["k_ntt_fwd"][2] + f["k_ntt_inv"][2]
fetch = 2 * (f["k_ntt_fwd"][0] + f["k_ntt_inv"][0]) * 1024 / limbs
write = (w["k_ntt_fwd"][0] + w["k_ntt_inv"][0]) * 1024 / limbs
n = json.load(open(os.path.join(src, "bench.json")))["roofline"]["algorithmic_bytes_per_launch"]
out = {"kernel": "k_ntt (fwd+inv)", "n": 8192, "limb_transforms_counted": limbs,
       "fetch_bytes_per_limb_corrected": round(fetch), "write_bytes_per_limb": round(write),
       "hbm_bytes_per_limb": round(fetch + write), "algorithmic_bytes_per_limb": 131072,
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 3 --warmup 1 "
                 "--no-cpu-baseline --no-profile`; counters are in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 "
                 "reports 1/2 of wide coalesced streaming reads); WRITE_SIZE exact"}
json.dump(out, open(os.path.join(dst, tag + "_ntt_traffic.json"), "w"), indent=1)
with open(os.path.join(dst, tag + "_hbm_traffic_by_kernel.txt"), "w") as fh:
    fh.write("kernel launches FETCH_SIZE_KiB_raw(x2 for wide streams) WRITE_SIZE_KiB  [whole bench process: 4 steps + setup]\n")
    for k in sorted(f, key=lambda k: -f[k][0]):
        fh.write("%-32s %6d %16.0f %16.0f\n" % (k, f[k][1], f[k][0], w.get(k, [0])[0]))
print(json.dumps(out))

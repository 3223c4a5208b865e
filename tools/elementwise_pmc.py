"""Per element-wise kernel of the query path: launches, mean duration, VALU-busy fraction and effective clock from ONE
`rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES` pass of bench.py, plus -- when a
second directory with `--pmc FETCH_SIZE` / `WRITE_SIZE` passes is given -- the bytes that left L2 per launch.
    python tools/elementwise_pmc.py <pmc_dir> [<fetch_dir> <write_dir>]
VALU busy = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * cycles), cycles = effective clock (GRBM_GUI_ACTIVE / 8 / duration) * duration:
the same derivation as tools/ntt_pmc_summary.py.  The algorithmic bytes per class are in the bench line (`elementwise_roofline`)."""
import collections, csv, glob, os, sys

KEEP = ("k_behz", "k_ks_", "k_tensor", "k_modswitch", "k_eval_epilogue", "k_i0_finish", "k_mac", "k_copy_jobs", "k_sum_jobs", "k_add_many")


def short(name):
    return name.split("(")[0].replace("void apsu_he::", "").replace("apsu_he::", "")


def load(d, want=None):
    out = collections.defaultdict(lambda: {"n": 0, "us": 0.0, "c": collections.Counter()})
    for ccf in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        ktf = ccf.replace("counter_collection", "kernel_trace")
        dur = {}
        if os.path.exists(ktf):
            for r in csv.DictReader(open(ktf)):
                dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        seen = set()
        for r in csv.DictReader(open(ccf)):
            k = short(r["Kernel_Name"])
            if not k.startswith(KEEP):
                continue
            e = out[k]
            e["c"][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                e["n"] += 1
                e["us"] += dur.get(r["Dispatch_Id"], 0.0)
    return out


main = load(sys.argv[1])
fetch = load(sys.argv[2]) if len(sys.argv) > 2 else {}
write = load(sys.argv[3]) if len(sys.argv) > 3 else {}
print("%-34s %7s %9s %9s %7s %9s %12s %12s" % ("kernel", "count", "mean us", "VALU busy", "GHz", "inst/wave", "L2->mem MB", "mem<-L2 MB"))
for k, e in sorted(main.items(), key=lambda kv: -kv[1]["us"]):
    c, us = e["c"], e["us"]
    clk = c.get("GRBM_GUI_ACTIVE", 0) / 8 / (us * 1e-6) / 1e9 if us else 0
    cyc = (clk or 2.1) * 1e9 * us * 1e-6
    busy = 100 * c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / cyc if cyc else 0
    ipw = c.get("SQ_INSTS_VALU", 0) / c["SQ_WAVES"] if c.get("SQ_WAVES") else 0
    f = fetch.get(k, {"c": {}, "n": 0})
    w = write.get(k, {"c": {}, "n": 0})
    fmb = f["c"].get("FETCH_SIZE", 0) * 2 / 1024 / max(1, f["n"]) if f["n"] else float("nan")      # KB -> MB per launch, x2 (gfx950)
    wmb = w["c"].get("WRITE_SIZE", 0) / 1024 / max(1, w["n"]) if w["n"] else float("nan")
    print("%-34s %7d %9.1f %8.0f%% %7.2f %9.0f %12.1f %12.1f" % (k[:34], e["n"], us / max(1, e["n"]), busy, clk, ipw, fmb, wmb))

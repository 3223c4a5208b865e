"""Summarises `rocprofv3 --kernel-trace --pmc ...` passes of tools/ntt_prof_one.py: per k_ntt dispatch the counters, and the
derived figures (VALU instructions per butterfly, VALU busy fraction, effective clock).
usage: python tools/ntt_pmc_summary.py <dir> [<dir> ...]   (one directory per --pmc pass)"""
import collections, csv, glob, os, sys
n, logn = 8192, 13
bfly_per_limb = (n // 2) * logn
for d in sys.argv[1:]:
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not cc:
        continue
    for ccf in sorted(cc):                                       # one counter file per --pmc pass (passes may share a directory)
        ktf = ccf.replace("counter_collection", "kernel_trace")
        dur = {}
        for r in csv.DictReader(open(ktf)) if os.path.exists(ktf) else []:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        disp = collections.OrderedDict()
        for r in csv.DictReader(open(ccf)):
            if "k_ntt" not in r["Kernel_Name"]:
                continue
            key = r["Dispatch_Id"]
            e = disp.setdefault(key, {"name": r["Kernel_Name"].split("(")[0].replace("void apsu_he::", ""),
                                      "limbs": int(r["Grid_Size"]) // int(r["Workgroup_Size"]), "c": {}})
            e["c"][r["Counter_Name"]] = float(r["Counter_Value"])
        for key, e in list(disp.items())[-2:]:                       # the last forward + inverse launch (warm)
            us = dur.get(key, 0.0)
            c = e["c"]
            line = "%-28s %6d limbs %8.1f us  " % (e["name"], e["limbs"], us) + "  ".join("%s=%.3g" % kv for kv in sorted(c.items()))
            print(line)
            if "SQ_INSTS_VALU" in c:
                per_bfly = c["SQ_INSTS_VALU"] * 64 / (e["limbs"] * bfly_per_limb * 1.0) / 64 * 64
                # SQ_INSTS_VALU counts wave-level instructions: per lane-butterfly = insts * 64 lanes / (limbs * butterflies)
                print("    VALU instructions per butterfly (per lane): %.2f" % (c["SQ_INSTS_VALU"] * 64 / (e["limbs"] * bfly_per_limb)))
            if "SQ_ACTIVE_INST_VALU" in c and us:
                clk = c.get("GRBM_GUI_ACTIVE", 0) / 8 / (us * 1e-6) / 1e9 if "GRBM_GUI_ACTIVE" in c else 0
                cyc = (clk or 2.1) * 1e9 * us * 1e-6
                print("    VALU busy: %.0f %% of %d SIMD-cycles%s" % (100 * c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc, cyc,
                                                                     (", effective clock %.2f GHz" % clk) if clk else ""))

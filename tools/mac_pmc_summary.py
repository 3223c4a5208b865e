"""Summarises the `rocprofv3 --kernel-trace --pmc ...` passes of tools/mac_prof_one.py for ONE kernel (default k_mac): the last
(warm) dispatch of every pass -- duration, every counter, and the ratios that say what the waves wait on.
usage: python tools/mac_pmc_summary.py <dir-with-pass-subdirs> [kernel-substring]"""
import collections, csv, glob, os, sys
root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "k_mac<"
allc = {}
durs = []
for ccf in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    ktf = ccf.replace("counter_collection", "kernel_trace")
    dur = {}
    for r in csv.DictReader(open(ktf)) if os.path.exists(ktf) else []:
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(ccf)):
        if want not in r["Kernel_Name"]:
            continue
        e = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"].split("(")[0].replace("void apsu_he::", ""), "grid": int(r["Grid_Size"]),
                                               "wg": int(r["Workgroup_Size"]), "vgpr": r.get("VGPR_Count", r.get("Arch_VGPR_Count", "?")),
                                               "lds": r.get("LDS_Block_Size", "?"), "scr": r.get("Scratch_Size", "?"), "c": collections.defaultdict(float)})
        e["c"][r["Counter_Name"]] += float(r["Counter_Value"])
    if not disp:
        continue
    # the largest dispatch of the kernel (the in-path launch), last occurrence
    big = max(e["grid"] for e in disp.values())
    key = [k for k, e in disp.items() if e["grid"] == big][-1]
    e = disp[key]
    us = dur.get(key, 0.0)
    durs.append(us)
    print("%s  pass %s: grid %d x %d threads, VGPRs %s, LDS %s, scratch %s, %.1f us" % (e["name"], os.path.basename(os.path.dirname(os.path.dirname(ccf))) or ccf, e["grid"] // e["wg"], e["wg"], e["vgpr"], e["lds"], e["scr"], us))
    for k, v in sorted(e["c"].items()):
        print("    %-40s %.6g" % (k, v))
        allc[k] = v
    allc.setdefault("_wgs", e["grid"] // e["wg"]); allc.setdefault("_wg", e["wg"])
if not allc:
    sys.exit("no dispatch of %s found under %s" % (want, root))
c = allc
us = sum(durs) / len(durs)
print("\n---- derived (mean launch %.1f us over %d passes; SQ_* cycle counters are in quad-cycles, summed over all SQs) ----" % (us, len(durs)))
def show(label, val, fmt="%.3g"):
    print("  %-62s " % label + (fmt % val))
if "GRBM_GUI_ACTIVE" in c:
    clk = c["GRBM_GUI_ACTIVE"] / 8 / (us * 1e-6) / 1e9            # summed over the 8 XCCs
    show("effective shader clock (GRBM_GUI_ACTIVE / 8 XCC / duration), GHz", clk, "%.2f")
else:
    clk = 2.1
cyc = clk * 1e9 * us * 1e-6                                       # shader cycles of the launch
if "SQ_WAVE_CYCLES" in c and "SQ_WAVES" in c:
    show("waves launched", c["SQ_WAVES"], "%d")
    show("mean wave lifetime, us (SQ_WAVE_CYCLES x 4 / waves / clock)", c["SQ_WAVE_CYCLES"] * 4 / c["SQ_WAVES"] / (clk * 1e3), "%.1f")
    show("mean resident waves per SIMD (SQ_WAVE_CYCLES x 4 / (1024 SIMDs x cycles))", c["SQ_WAVE_CYCLES"] * 4 / (1024 * cyc), "%.2f")
if "SQ_BUSY_CU_CYCLES" in c:
    show("CU busy fraction (SQ_BUSY_CU_CYCLES x 4 / (256 CUs x cycles))", c["SQ_BUSY_CU_CYCLES"] * 4 / (256 * cyc), "%.3f")
for a in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
    if a in c and "SQ_WAVE_CYCLES" in c:
        show("%s / SQ_WAVE_CYCLES" % a, c[a] / c["SQ_WAVE_CYCLES"], "%.3f")
for a in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_FLAT"):
    if a in c:
        show("%s: busy fraction of the 1024 SIMD issue ports (x 4 / (1024 x cycles))" % a, c[a] * 4 / (1024 * cyc), "%.3f")
for a in ("SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_VMEM_WR", "SQ_INST_CYCLES_SMEM", "SQ_INST_CYCLES_SALU"):
    if a in c:
        show("%s x 4 / (1024 x cycles)" % a, c[a] * 4 / (1024 * cyc), "%.3f")
if "SQ_INST_LEVEL_VMEM" in c and "SQ_INSTS_VMEM_RD" in c:
    show("mean VMEM latency, cycles (SQ_INST_LEVEL_VMEM x 4? / SQ_INSTS_VMEM): see note", c["SQ_INST_LEVEL_VMEM"] / max(1.0, c.get("SQ_INSTS_VMEM", c["SQ_INSTS_VMEM_RD"])), "%.1f")
if "SQ_INST_LEVEL_VMEM" in c:
    show("mean VMEM instructions in flight per SIMD (SQ_INST_LEVEL_VMEM / (1024 x cycles / 4))", c["SQ_INST_LEVEL_VMEM"] * 4 / (1024 * cyc), "%.2f")
if "SQ_INSTS_VALU" in c:
    show("VALU wave-instructions per workgroup", c["SQ_INSTS_VALU"] / c["_wgs"], "%.0f")
if "SQ_INSTS_VMEM_RD" in c:
    show("VMEM read wave-instructions per workgroup", c["SQ_INSTS_VMEM_RD"] / c["_wgs"], "%.0f")
for a in ("TCP_PENDING_STALL_CYCLES_sum", "TCP_TCR_TCP_STALL_CYCLES_sum", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum",
          "TCP_TD_TCP_STALL_CYCLES_sum", "TCP_LFIFO_STALL_CYCLES_sum", "TCP_RFIFO_STALL_CYCLES_sum", "TA_TA_BUSY_sum", "TD_TD_BUSY_sum",
          "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum", "TD_TC_STALL_sum", "TCP_GATE_EN1_sum", "TCP_GATE_EN2_sum"):
    if a in c:
        show("%s / (256 CUs x cycles)" % a, c[a] / (256 * cyc), "%.3f")
if "TCP_TCC_READ_REQ_LATENCY_sum" in c and "TCP_TCC_READ_REQ_sum" in c:
    show("mean L1 -> L2 read latency, cycles (TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ)", c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"], "%.0f")
if "TCP_TCP_LATENCY_sum" in c and "TCP_TOTAL_ACCESSES_sum" in c:
    show("mean L1 latency, cycles (TCP_TCP_LATENCY / TCP_TOTAL_ACCESSES)", c["TCP_TCP_LATENCY_sum"] / c["TCP_TOTAL_ACCESSES_sum"], "%.0f")
if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
    show("L2 hit rate (TCC_HIT / (TCC_HIT + TCC_MISS))", c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), "%.3f")
for a in ("TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum", "TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum", "TCC_TAG_STALL_sum", "TCC_SRC_FIFO_FULL_sum", "TCC_LATENCY_FIFO_FULL_sum",
          "TCC_IB_STALL_sum", "TCC_BUSY_sum"):
    if a in c and "TCC_CYCLE_sum" in c:
        show("%s / TCC_CYCLE_sum" % a, c[a] / c["TCC_CYCLE_sum"], "%.3f")
if "TCC_EA0_RDREQ_LEVEL_sum" in c and "TCC_EA0_RDREQ_sum" in c:
    show("mean memory-side read latency, TCC cycles (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ)", c["TCC_EA0_RDREQ_LEVEL_sum"] / c["TCC_EA0_RDREQ_sum"], "%.0f")
if "TCC_EA0_RDREQ_sum" in c:
    b = c.get("TCC_EA0_RDREQ_32B_sum", 0) * 32 + c.get("TCC_EA0_RDREQ_64B_sum", 0) * 64 + c.get("TCC_EA0_RDREQ_128B_sum", 0) * 128
    if b:
        show("memory-side read bytes (32B/64B/128B requests), GB", b / 1e9, "%.3f")
        show("  ... per second, TB/s", b / (us * 1e-6) / 1e12, "%.2f")

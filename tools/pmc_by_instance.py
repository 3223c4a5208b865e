"""Per-INSTANCE (not summed) counter values of one kernel's dispatches from `rocprofv3 --kernel-trace --pmc ... --output-format csv json`
passes: for every pass and every raw counter, per dispatch the values by hardware instance (the json output keeps the dimensions of a raw
counter -- XCC, channel -- where the csv rows and the `_sum` forms hide them), then min / mean / max over the instances and the most and
least loaded instances.  Written for tools/microbench/macbench.hip's PLACEMENT mode: is a slow copy of the database a copy whose requests
pile up on some L2 channels or some XCDs?
usage: python tools/pmc_by_instance.py <dir-with-pass-subdirs> <kernel-substring> [last N dispatches, default 16]"""
import collections, csv, glob, json, os, sys

root, want = sys.argv[1], sys.argv[2]
last_n = int(sys.argv[3]) if len(sys.argv) > 3 else 16


def walk_json(path):
    """yields (dispatch_id, kernel_name, counter_name, instance_key, value, duration_us) from one rocprofv3 results json.  A dispatch's
    records carry the counter handle and a value, one per hardware instance, in the order of the counter's `instances` list (whose entries
    name the dimensions: DIMENSION_XCC, DIMENSION_INSTANCE = the TCC channel of that XCC ...)"""
    doc = json.load(open(path))
    for sdk in doc.get("rocprofiler-sdk-tool", []):
        cnames, cinst = {}, {}
        for c in sdk.get("counters", []):
            h = c["id"]["handle"]
            cnames[h] = c.get("name", str(h))
            cinst[h] = ["/".join("%s%d" % (d["dimension_name"].replace("DIMENSION_", "").lower(), d["index"]) for d in i.get("dimensions", []))
                        for i in c.get("instances", [])]
        ksyms = {k.get("kernel_id"): (k.get("formatted_kernel_name") or k.get("kernel_name") or "") for k in sdk.get("kernel_symbols", [])}
        for rec in sdk.get("callback_records", {}).get("counter_collection", []):
            dd = rec.get("dispatch_data", {})
            di = dd.get("dispatch_info", {})
            did, name = di.get("dispatch_id"), ksyms.get(di.get("kernel_id"), "")
            us = (dd.get("end_timestamp", 0) - dd.get("start_timestamp", 0)) / 1e3
            seen = collections.Counter()
            for r in rec.get("records", []):
                h = r["counter_id"]["handle"]
                k = seen[h]; seen[h] += 1
                labels = cinst.get(h, [])
                yield did, name, cnames.get(h, str(h)), (labels[k] if k < len(labels) else "row%03d" % k), float(r["value"]), us


def walk_csv(path):
    """csv rows of a raw counter: one per instance, in instance order, without a label; the row index within (dispatch, counter) stands in"""
    ktf = path.replace("counter_collection", "kernel_trace")
    dur = {}
    for r in csv.DictReader(open(ktf)) if os.path.exists(ktf) else []:
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    seen = collections.Counter()
    for r in csv.DictReader(open(path)):
        key = (r["Dispatch_Id"], r["Counter_Name"])
        yield int(r["Dispatch_Id"]), r["Kernel_Name"], r["Counter_Name"], "row%03d" % seen[key], float(r["Counter_Value"]), dur.get(r["Dispatch_Id"], 0.0)
        seen[key] += 1


for pdir in sorted(glob.glob(os.path.join(root, "p*"))):
    if not os.path.isdir(pdir):
        continue
    js = glob.glob(os.path.join(pdir, "**", "*results.json"), recursive=True)
    cs = glob.glob(os.path.join(pdir, "**", "*counter_collection.csv"), recursive=True)
    rows, src = [], None
    if js:
        try:
            rows = [x for x in walk_json(js[0]) if want in x[1]]
            src = "json"
        except Exception as ex:                                    # the layout of the json differs between rocprofiler-sdk versions
            print("# %s: json not understood (%s), falling back to csv" % (pdir, ex))
    if not rows and cs:
        rows = [x for x in walk_csv(cs[0]) if want in x[1]]
        src = "csv"
    if not rows:
        print("# %s: no rows for %s" % (pdir, want))
        continue
    by = collections.OrderedDict()
    for did, name, cn, inst, val, us in rows:
        e = by.setdefault(did, {"us": us, "c": collections.defaultdict(collections.OrderedDict)})
        e["c"][cn][inst] = e["c"][cn].get(inst, 0.0) + val
    dids = sorted(by)[-last_n:]
    print("# %s (%s): dispatches %s" % (pdir, src, dids))
    for did in dids:
        e = by[did]
        for cn, inst in e["c"].items():
            v = list(inst.values())
            if len(v) < 2:
                print("  dispatch %5d %8.1f us  %-36s single value %.6g" % (did, e["us"], cn, v[0]))
                continue
            mean = sum(v) / len(v)
            order = sorted(range(len(v)), key=lambda i: v[i])
            keys = list(inst.keys())
            print("  dispatch %5d %8.1f us  %-36s %3d instances: sum %.6g  min %.6g  mean %.6g  max %.6g  (max/mean %.3f, min/mean %.3f)  lowest %s  highest %s"
                  % (did, e["us"], cn, len(v), sum(v), min(v), mean, max(v), max(v) / mean if mean else 0, min(v) / mean if mean else 0,
                     keys[order[0]], keys[order[-1]]))
    # the full per-instance vector of the last slow and the last fast dispatch, for the record
    if len(dids) >= 2:
        slow = max(dids, key=lambda d: by[d]["us"]); fast = min(dids, key=lambda d: by[d]["us"])
        for label, did in (("slowest", slow), ("fastest", fast)):
            for cn, inst in by[did]["c"].items():
                v = list(inst.values())
                if len(v) >= 2:
                    print("  %s dispatch %d (%.1f us) %s by instance: %s" % (label, did, by[did]["us"], cn, " ".join("%.4g" % x for x in v)))

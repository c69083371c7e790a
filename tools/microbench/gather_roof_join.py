"""Join tools/microbench/gather_roof.bin's JSON lines with the rocprofv3 --pmc rows of the same launches (the k-th k_gather dispatch of a --pmc pass
is the k-th line that pass printed).  usage: gather_roof_join.py gpurun_out/gather_roof"""
import csv, glob, json, os, sys, collections

root = sys.argv[1]


def lines(path):
    return [json.loads(l) for l in open(path) if l.startswith("{")] if os.path.exists(path) else []


plain = lines(os.path.join(root, "plain.jsonl"))
key = lambda r: (r["table_MiB"], r["record_B"], r["wgs_per_cu"], r["chains"])
table = collections.OrderedDict((key(r), dict(r)) for r in plain)
for r in table.values():
    r.pop("gather_launches_so_far", None)
    r["counters"] = {}
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    tag = os.path.basename(d)[4:]
    out = lines(os.path.join(root, "pmc_%s.jsonl" % tag))
    rows = collections.defaultdict(dict)  # dispatch id -> counter -> value
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_gather" in r["Kernel_Name"]:
                c = rows[int(r["Dispatch_Id"])]
                c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    order = sorted(rows)
    if len(order) != len(out):
        print("pass %s: %d gather dispatches for %d lines, skipped" % (tag, len(order), len(out)), file=sys.stderr)
        continue
    for did, line in zip(order, out):
        t = table.get(key(line))
        if t is not None:
            t["counters"].update(rows[did])
            t["counters"].setdefault("ms_under_pmc", {})[tag] = line["ms"]
for r in table.values():
    c = r["counters"]
    recs = 256.0 * r["wgs_per_cu"] * 256 * r["chains"] * r["steps"]
    if "FETCH_SIZE" in c:  # KiB, x2 on gfx950 (MI355X_MICROARCH.md: 128-B requests tallied at 64 B)
        r["fabric_B_per_record"] = round(c["FETCH_SIZE"] * 1024 * 2 / recs, 1)
    if c.get("TCC_REQ_sum"):
        r["l2_hit_rate"] = round(c.get("TCC_HIT_sum", 0.0) / (c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0)), 4)
        r["l2_req_per_record"] = round(c["TCC_REQ_sum"] / recs, 3)
    for k, v in list(c.items()):
        if "UTCL" in k and isinstance(v, float):
            r.setdefault("translation_per_record", {})[k] = round(v / recs, 5)
    if c.get("TCC_EA0_RDREQ_sum") and "TCC_EA0_RDREQ_LEVEL_sum" in c:  # requests in flight summed per cycle / requests = cycles per fabric read
        r["fabric_read_latency_cycles"] = round(c["TCC_EA0_RDREQ_LEVEL_sum"] / c["TCC_EA0_RDREQ_sum"], 1)
    if c.get("TCP_TCC_READ_REQ_sum") and "TCP_TCC_READ_REQ_LATENCY_sum" in c:
        r["l1_to_l2_read_latency_cycles"] = round(c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"], 1)
print(json.dumps({"what": "random dependent gathers, k_path's launch shape (tools/microbench/gather_roof.hip)", "configs": list(table.values())}, indent=1))

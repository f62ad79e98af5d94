#!/usr/bin/env python3
"""rocprofv3 counter CSVs of tools/collect_counters.sh -> {kernel: {counter: value per launch}}.
   python3 tools/pmc_aggregate.py gpurun_out/pmc_counters_<tag> gpurun_out/counters_<tag>.json"""

import collections, csv, glob, json, sys
src, dst = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(lambda: collections.Counter())
n = collections.defaultdict(set)
for f in glob.glob(src + "/*_g*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if not k.startswith("k_"):
            continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])].add((f, r["Dispatch_Id"]))
out = {k: dict({c: v / max(1, len(n[(k, c)])) for c, v in cs.items()}, launches_seen=max(len(n[(k, c)]) for c in cs))
       for k, cs in tot.items()}
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
for k in sorted(out):
    c = out[k]
    if c.get("SQ_BUSY_CYCLES", 0) < 1e5 and c.get("SQ_WAVE_CYCLES", 0) < 1e7:
        continue
    print(k[:60], {x: "%.3g" % c[x] for x in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                                            "SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU") if x in c})

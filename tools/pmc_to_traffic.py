#!/usr/bin/env python3
"""rocprofv3 counter CSVs (FETCH_SIZE / WRITE_SIZE passes of tools/collect_traffic.sh) ->
{kernel: HBM bytes per launch} keyed like bench.py's kernel names.

Units and gfx950 correction (MI355X_MICROARCH.md, section HBM): both counters are in KiB;
FETCH_SIZE reports exactly half the bytes of a wide coalesced streaming read on gfx950, so it is
doubled.  Calibration in our own access pattern: k_reduce_sparse reads the (b,L,q) fp32 emission
tensor exactly once (6.144e9 B at b=1024, L=1e5, q=15) and its corrected FETCH_SIZE is 6.24e9 B
(the difference is the operator write-allocate + A); WRITE_SIZE of k_backward is 6.69e9 B for a
6.144e9 B posterior tensor plus 0.4e9 B of re-written partial lines.
"""
import collections
import csv
import glob
import json
import sys

NAMES = {"k_reduce_sparse": "reduce", "k_reduce": "reduce_dense", "k_scan": "scan",
         "k_scan_compose": "scan_compose", "k_scan_inner": "scan_inner",
         "k_forward": "forward", "k_backward": "backward"}


def per_kernel(path, counter):
    tot, n = collections.Counter(), collections.Counter()
    for f in glob.glob(path + "/%s/*/*counter_collection.csv" % counter):
        seen = set()
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].replace("void ", "").split("<")[0].split("(")[0]
            tot[k] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                n[k] += 1
    return {k: tot[k] / n[k] for k in tot}


def main():
    src, dst = sys.argv[1], sys.argv[2]
    fetch, write = per_kernel(src, "FETCH_SIZE"), per_kernel(src, "WRITE_SIZE")
    out = {}
    for k, name in NAMES.items():
        if k in fetch or k in write:
            rd = 2.0 * fetch.get(k, 0.0) * 1024.0
            wr = write.get(k, 0.0) * 1024.0
            out[name] = {"hbm_bytes": rd + wr, "read_bytes": rd, "write_bytes": wr}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

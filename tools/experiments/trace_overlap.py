import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("k_", "void k_"))]
rows = [r for r in rows if "topo" not in r["Kernel_Name"] and "copy" not in r["Kernel_Name"]]
t0 = min(int(r["Start_Timestamp"]) for r in rows[-60:])
for r in rows[-40:]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-28s q%-3s %9.3f -> %9.3f ms  (%.3f)" % (n, r["Queue_Id"], s / 1e6, e / 1e6, (e - s) / 1e6))

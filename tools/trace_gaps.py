"""Reads a rocprofv3 kernel-trace CSV: prints kernels longer than a threshold and idle gaps longer than it (with neighbours)."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
thr = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 5e6
t0 = rows[0][0]
print("kernels", len(rows))
for i, (s, e, n) in enumerate(rows):
    if e - s > thr:
        print("LONG  %9.3f ms  dur %8.3f ms  %s" % ((s - t0) / 1e6, (e - s) / 1e6, n))
    if i and s - rows[i - 1][1] > thr:
        print("GAP   %9.3f ms  gap %8.3f ms  after %s  before %s" % ((rows[i - 1][1] - t0) / 1e6, (s - rows[i - 1][1]) / 1e6, rows[i - 1][2], n))

"""Reads a rocprofv3 kernel-trace CSV of a solve and prints the average timeline of one SPG iteration of the full problem:
iterations are the kernel sequences between consecutive launches of `anchor` (default k_proj_fused); the most common
sequence is averaged (start offset from the anchor's start, duration, gap to the previous kernel's end)."""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_proj_fused"
idx = [i for i, r in enumerate(rows) if r[2].startswith(anchor)]
seqs = collections.defaultdict(list)
for a, b in zip(idx[:-1], idx[1:]):
    names = tuple(r[2] for r in rows[a:b])
    if rows[b][0] - rows[a][0] < 500000:        # skip windows interrupted by the host
        seqs[names].append((a, b))
print("iterations found:", len(idx) - 1)
for names, occ in sorted(seqs.items(), key=lambda kv: -len(kv[1]))[:3]:
    print("\npattern seen %d times:" % len(occ))
    n = len(names)
    period = sum(rows[b][0] - rows[a][0] for a, b in occ) / len(occ) / 1e3
    for j in range(n):
        st = sum(rows[a + j][0] - rows[a][0] for a, b in occ) / len(occ) / 1e3
        du = sum(rows[a + j][1] - rows[a + j][0] for a, b in occ) / len(occ) / 1e3
        gap = sum(rows[a + j][0] - rows[a + j - 1][1] for a, b in occ) / len(occ) / 1e3
        print("  %-34s start %7.2f us  dur %6.2f us  gap before %5.2f us" % (names[j][:34], st, du, gap))
    print("  period %.2f us; kernel time %.2f us" % (period, sum(sum(rows[a + j][1] - rows[a + j][0] for j in range(n)) for a, b in occ) / len(occ) / 1e3))

#!/usr/bin/env python3
"""Development aid: idle time on the GPU's compute timeline from a rocprofv3 --kernel-trace (+ --memory-copy-trace) csv directory:
   python scripts/timeline_gaps.py <dir>   -- busy / idle share of the last pass, the largest gaps and what stood on either side."""
import csv
import glob
import sys

d = sys.argv[1]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kt))]
rows.sort()
# the last third of the forward kernels = the last pass
fw = [i for i, r in enumerate(rows) if "pmt_forward_kernel" in r[2]]
first = fw[2 * len(fw) // 3]
rows = rows[first:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_end, gaps = 0, rows[0][0], []
for s, e, name in rows:
    if s > cur_end:
        gaps.append((s - cur_end, prev, name))
        busy += 0
    cur_end = max(cur_end, e)
    prev = name
union, ce = 0, None
for s, e, _ in rows:
    if ce is None or s > ce:
        union += e - s
        ce = e
    elif e > ce:
        union += e - ce
        ce = e
nb = sum("pmt_forward_kernel" in r[2] for r in rows)
print(f"{nb} batches over {(t1 - t0) / 1e6:.2f} ms = {(t1 - t0) / 1e3 / nb:.1f} us/batch; some kernel running {100 * union / (t1 - t0):.1f} % of the time")
print(f"sum of kernel durations per batch: {sum(e - s for s, e, _ in rows) / 1e3 / nb:.1f} us")
gaps.sort(reverse=True)
for g, a, b in gaps[:12]:
    print(f"  idle {g / 1e3:8.1f} us   after {a[:50]:50s} before {b[:50]}")
print(f"  idle total {sum(g for g, _, _ in gaps) / 1e3 / nb:.1f} us/batch in {len(gaps)} gaps")
mc = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
if mc:
    cp = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(mc[0])) if int(r["Start_Timestamp"]) >= t0]
    print(f"copies in the window: {len(cp)}, {sum(e - s for s, e in cp) / 1e3 / nb:.1f} us/batch")

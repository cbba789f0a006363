"""Timeline of the last training step in a rocprofv3 kernel trace (development aid):
python scripts/step_timeline.py <k_kernel_trace.csv> <out.txt>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "pmt_pack_kernel" in r["Kernel_Name"]]
seg = rows[idx[-2]:idx[-1]]
t0 = int(seg[0]["Start_Timestamp"])
prev_end = t0
with open(sys.argv[2], "w") as out:
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        out.write(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {r['Kernel_Name'][:90]}\n")
        prev_end = e
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    out.write(f"kernels {len(seg)} span {(prev_end - t0) / 1e3:.1f} us busy {busy / 1e3:.1f} us\n")

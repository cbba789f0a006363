"""Static instruction mix of one kernel of an assembly listing built with -gline-tables-only, attributed to source lines.

    hipcc ... --offload-device-only -gline-tables-only -S -o x.s pmt_backward.hip
    python scripts/isa_mix.py x.s <mangled-name-substring> [top]

Development tooling (not product): the counts are static, loop trip counts are not applied."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): return "vmem"
    if op.startswith("scratch_"): return "scratch"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    files = {}
    per_line = collections.defaultdict(collections.Counter)
    per_op = collections.Counter()
    cls_tot = collections.Counter()
    inside = False
    loc = ("?", 0)
    for ln in open(path):
        s = ln.strip()
        m = re.match(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s)
        if m:
            files[int(m.group(1))] = m.group(2).split("/")[-1]
            continue
        if not inside:
            if re.match(r"^_Z\S*:", ln) and key in ln:
                inside = True
            continue
        if s.startswith(".Lfunc_end"):
            break
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            loc = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        if not s or s.startswith(".") or s.startswith(";") or s.endswith(":"):
            continue
        op = s.split()[0]
        c = classify(op)
        per_line[loc][c] += 1
        per_op[op] += 1
        cls_tot[c] += 1
    print("totals:", dict(cls_tot))
    print("\n-- top opcodes")
    for op, n in per_op.most_common(top):
        print(f"{n:7d}  {op}")
    print("\n-- top source lines by VALU")
    rows = sorted(per_line.items(), key=lambda kv: -kv[1]["valu"])[:top]
    for (f, l), c in rows:
        print(f"{f}:{l:<5d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))


if __name__ == "__main__":
    main()

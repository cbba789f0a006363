#!/usr/bin/env python3
"""Static guard against the gfx950 MFMA accumulation-chain hazard that round 4 met as a "race" (profiles/r05_mfma_chain_hazard.txt,
scripts/microbench/mfma_chain_hazard.hip): an MFMA that reads as SrcC the register another MFMA of a DIFFERENT opcode is still writing
gets a stale accumulator -- the hardware interlocks a chain of ONE opcode only, and hipcc 7.2 issues no wait state for the pair (it
treats "SrcC = previous vDst" as a back-to-back accumulation whatever the opcodes).  Measured: wrong with 0 - 4 wait states between a
`v_mfma_f32_16x16x32_{bf16,f16}` and a dependent `v_mfma_f32_16x16x16_*`, right with 8.

    scripts/check_mfma_chains.py permutect_amd/libpermutect_amd.so [more libraries / objects ...]

Disassembles the gfx950 code objects of each file and reports every MFMA whose SrcC overlaps the vDst of an MFMA of another opcode
issued fewer than MIN_SLOTS issue slots earlier (other waves may fill the gap at run time -- or may not: that is what made it look like a
race).  Exit status 1 if any is found.  tests/test_host_cpu.py runs it over every library the build made."""
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
MIN_SLOTS = 10  # issue slots between the writer and the dependent reader of another opcode (measured need: more than 4, 8 suffice)


def code_objects(path, tmp):
    """the gfx950 code object(s) inside an object file or shared library (a .hip_fatbin section of offload bundles)"""
    fb = tmp + "/fatbin"
    r = subprocess.run([LLVM + "llvm-objcopy", f"--dump-section=.hip_fatbin={fb}", path, tmp + "/stripped"], capture_output=True, text=True)
    if r.returncode != 0:
        return []
    # a linked library's section is the concatenation of one bundle per translation unit: split at the bundles' magic
    blob, magic = open(fb, "rb").read(), b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    out = []
    for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(blob)])):
        part = f"{tmp}/bundle{n}"
        with open(part, "wb") as f:
            f.write(blob[a:b])
        listing = subprocess.run([LLVM + "clang-offload-bundler", "--list", "--type=o", f"--input={part}"], capture_output=True, text=True).stdout.split()
        for i, target in enumerate(t for t in listing if "gfx950" in t):
            co = f"{tmp}/dev{n}_{i}.co"
            r = subprocess.run([LLVM + "clang-offload-bundler", "--type=o", f"--targets={target}", f"--input={part}", f"--output={co}", "--unbundle"],
                               capture_output=True, text=True)
            if r.returncode == 0:
                out.append(co)
    return out


def regs(tok):
    """'v[4:7]' -> (4, 7); 'v9' -> (9, 9); 'a[0:3]' -> accumulator registers get their own space; anything else None"""
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        base = 0 if m.group(1) == "v" else 1000
        return base + int(m.group(2)), base + int(m.group(3))
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        base = 0 if m.group(1) == "v" else 1000
        return base + int(m.group(2)), base + int(m.group(2))
    return None


def scan(disassembly):
    findings, kernel = [], "?"
    last = None  # (opcode, dst range, slots since)
    for line in disassembly.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            kernel, last = m.group(1), None
            continue
        parts = line.strip().split("//")[0].split(None, 1)
        if not parts or not re.match(r"^[a-z]", parts[0]):
            continue
        op = parts[0]
        if op.startswith("s_branch") or op.startswith("s_cbranch") or op.startswith("s_endpgm") or op.startswith("s_setpc"):
            last = None  # (control flow: the straight-line distance means nothing beyond it)
            continue
        slots = 1
        if op == "s_nop" and len(parts) > 1:
            try:
                slots = int(parts[1].strip(), 0) + 1
            except ValueError:
                slots = 1
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            ops = [t.strip() for t in parts[1].split(",")] if len(parts) > 1 else []
            dst = regs(ops[0]) if ops else None
            src_c = regs(ops[3].split()[0]) if len(ops) > 3 else None
            if last is not None and src_c is not None and last[0] != op and last[2] < MIN_SLOTS:
                (lo, hi), (slo, shi) = last[1], src_c
                if slo <= hi and lo <= shi:
                    findings.append((kernel, last[0], op, last[2]))
            last = (op, dst, 0) if dst is not None else None
            continue
        if last is not None:
            last = (last[0], last[1], last[2] + slots)
    return findings


def check(paths):
    total = []
    with tempfile.TemporaryDirectory() as tmp:
        for path in paths:
            for co in code_objects(path, tmp):
                dis = subprocess.run([LLVM + "llvm-objdump", "-d", co], capture_output=True, text=True).stdout
                for kernel, w, r, gap in scan(dis):
                    total.append((path, kernel, w, r, gap))
    return total


def main():
    found = check(sys.argv[1:])
    for path, kernel, w, r, gap in found:
        name = subprocess.run(["c++filt", kernel], capture_output=True, text=True).stdout.strip()[:140]
        print(f"{path}: {name}: {r} accumulates onto {w} after {gap} issue slot(s)")
    print(f"{len(found)} MFMA accumulation chain(s) across opcodes closer than {MIN_SLOTS} issue slots")
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Register / LDS / spill figures of every kernel in a built object, read from the code object's metadata notes.
usage: scripts/kernel_resources.py permutect_amd/csrc/pmt_forward.o [name regex]"""
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"


def main():
    obj, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else ".")
    with tempfile.TemporaryDirectory() as tmp:
        co, fb = tmp + "/dev.co", tmp + "/fatbin"
        subprocess.run([LLVM + "llvm-objcopy", f"--dump-section=.hip_fatbin={fb}", obj], check=True)
        obj = fb
        for target in ("hipv4-amdgcn-amd-amdhsa--gfx950", "hip-amdgcn-amd-amdhsa--gfx950"):
            r = subprocess.run([LLVM + "clang-offload-bundler", "--type=o", f"--targets={target}", f"--input={obj}", f"--output={co}", "--unbundle"],
                               capture_output=True, text=True)
            if r.returncode == 0:
                break
        notes = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for blk in notes.split("- .agpr_count")[1:]:
        blk = ".agpr_count" + blk

        def g(key):
            m = re.search(r"\." + key + r":\s+(\S+)", blk)
            return m.group(1) if m else "?"
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        if re.search(filt, name):
            print(f"vgpr {g('vgpr_count'):>4} agpr {g('agpr_count'):>3} sgpr {g('sgpr_count'):>4} vspill {g('vgpr_spill_count'):>4} "
                  f"sspill {g('sgpr_spill_count'):>4} scratch {g('private_segment_fixed_size'):>5} lds {g('group_segment_fixed_size'):>6}  {name[:140]}")


if __name__ == "__main__":
    main()

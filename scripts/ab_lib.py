#!/usr/bin/env python3
"""Development aid: A/B of builds of the library on one box, alternating: python scripts/ab_lib.py <lib.so> [<lib.so> ...]
prints bench.py's kernel times (training forward, backward, filter forward) and step times per build, two rounds."""
import json, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(here)
libs = sys.argv[1:]
for rep in range(2):
    for lib in libs:
        env = dict(os.environ)
        if lib != "default":
            env["PMT_LIB"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-extras", "--no-cpu-baseline", "--steps", "100", "--warmup", "10"],
                             env=env, capture_output=True, text=True)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if not line:
            print(lib, "FAILED", out.stderr[-1500:]); continue
        d = json.loads(line[0])
        k = d["roofline"]["other_kernel_ms"]
        print(f"{os.path.basename(lib):45s} train {d['ms_per_step']:.3f} ms (fwd {k['pmt_forward']:.4f}, bwd {k['pmt_backward']:.4f})  filter {d['filter']['ms_per_step']:.4f} ms (kernel {d['filter']['roofline']['kernel_ms']:.4f})", flush=True)

"""Filter-forward kernel time of the production shape, f16 two-piece products against the round-3 bf16 three-piece form, alternating
(development aid): python scripts/quick_fwd_ab.py [B]"""
import os, subprocess, sys
B = sys.argv[1] if len(sys.argv) > 1 else "65536"
here = os.path.dirname(os.path.abspath(__file__))
for rep in range(2):
    for shape in ("", "bf16x3"):
        env = dict(os.environ, PMT_SHAPE=shape)
        out = subprocess.run([sys.executable, os.path.join(here, "quick_fwd.py"), B], env=env, capture_output=True, text=True)
        print(f"[{shape or 'f16x2'}]", out.stdout.strip().splitlines()[-1] if out.returncode == 0 else out.stderr[-2000:], flush=True)

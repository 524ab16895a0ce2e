import os, sys, subprocess
here = os.path.dirname(os.path.abspath(__file__))
for dbg in (0, 1, 2, 4, 8):
    env = dict(os.environ, CR_ATTN_DBG=str(dbg))
    out = subprocess.run([sys.executable, os.path.join(here, "kbench.py")], env=env, capture_output=True, text=True).stdout
    print("dbg=%d" % dbg, [l for l in out.splitlines() if l.startswith("attn_fwd rate=0.2")])

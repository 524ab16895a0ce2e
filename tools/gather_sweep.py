"""bench.py's gather block alone (fresh rows per launch), for sweeps: CASTREC_TL_NR=1|2|4|7 python tools/gather_sweep.py"""
import os, sys, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
sys.argv = ["x"]
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(R, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
import numpy as np, torch
import castrec_amd
g = b.gather_block()
print(os.environ.get("CASTREC_TL_NR", "2"), json.dumps({k: g[k] for k in ("embed_fwd", "read_only")}))

import os, sys
import os as _o; R=_o.path.dirname(_o.path.dirname(_o.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np, torch
import castrec_amd
from castrec_amd import engine as E
import test_model_gpu as tm
from oracle import fpmodel as fm
for model in ("cast_3", "cast_2", "cast_3"):
    try:
        tm._other_shapes(E, model, 50, 1, 200, 2, B=3, prec="bf16x3", itemnum=300, max_bins=200)
        print(model, "ok")
    except AssertionError as e:
        print(model, "FAIL", str(e)[:200])

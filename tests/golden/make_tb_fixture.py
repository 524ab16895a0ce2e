#!/usr/bin/env python
"""Cuts a small TFRecord fixture out of an event file TensorFlow wrote for the reference
(/root/reference/saved_models/ml-1m.txt/cast_1_10-17-2019-23-47-36/events.out.tfevents.*: the run behind BASELINE configs[1]):
the file-version record, the first three and the last eleven scalar records, byte for byte (the 3 MB graph-definition record is
left out; records are self-delimiting).  Output: tests/golden/ref_events_cast_1_ml1m.tfevents + the scalars it holds as JSON.
Run in the build container (the reference tree is not on the GPU box)."""
import glob
import json
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import castrec_amd  # noqa: E402,F401
from castrec_amd import tb_events as T  # noqa: E402

src = glob.glob("/root/reference/saved_models/ml-1m.txt/cast_1_10-17-2019-23-47-36/events.out.tfevents.*")[0]
data = open(src, "rb").read()
recs, pos = [], 0
while pos < len(data):
    (n,) = struct.unpack("<Q", data[pos:pos + 8])
    recs.append(data[pos:pos + 16 + n])
    pos += 16 + n
small = [r for r in recs if len(r) < 4096]               # everything but the graph definition
keep = small[:4] + small[-11:]
out = os.path.join(HERE, "ref_events_cast_1_ml1m.tfevents")
open(out, "wb").write(b"".join(keep))
ev = T.read_events(out)
json.dump({"source": os.path.relpath(src, "/root/reference"), "records_in_source": len(recs), "events": [[s, d] for s, d in ev]},
          open(os.path.join(HERE, "ref_events_cast_1_ml1m.json"), "w"), indent=1)
print(len(keep), "records,", os.path.getsize(out), "bytes;", len(ev), "scalar events; last:", ev[-1])

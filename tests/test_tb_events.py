"""TensorBoard event files (castrec_amd/tb_events.py) against records TensorFlow wrote for the reference's cast_1 ml-1m run
(tests/golden/ref_events_cast_1_ml1m.tfevents, cut by tests/golden/make_tb_fixture.py) and against that run's log.txt."""
import json
import os
import struct

import numpy as np

import castrec_amd  # noqa: F401
from castrec_amd import tb_events as T

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "ref_events_cast_1_ml1m.tfevents")


def test_crc32c_known_answer():
    assert T.crc32c(b"123456789") == 0xE3069283            # the check value of CRC-32C (Castagnoli)
    assert T.crc32c(b"") == 0


def test_reader_takes_the_reference_runs_records():
    ev = T.read_events(FIX)                                  # every length and data CRC is verified on the way
    want = json.load(open(os.path.join(HERE, "golden", "ref_events_cast_1_ml1m.json")))["events"]
    assert [(s, d) for s, d in ev] == [(s, d) for s, d in want] and len(ev) == 14
    # the evaluation of epoch 200 = the last line of that run's log.txt
    # (/root/reference/saved_models/ml-1m.txt/cast_1_10-17-2019-23-47-36/log.txt:10; BASELINE.md's cast_1 row)
    e200 = [d for s, d in ev if s == 200 and "TEST/NDCG@10" in d][0]
    log = {"VALID/NDCG@10": 0.6048072224485163, "VALID/HR@10": 0.8364238410596027,
           "TEST/NDCG@10": 0.5776360859970379, "TEST/HR@10": 0.8109271523178808}
    for k, v in log.items():
        assert abs(e200[k] - v) <= 1e-7 * v, (k, e200[k], v)
    assert ev[-1][0] == 201 and abs(ev[-1][1]["TRAIN/loss"] - 0.38796103) < 1e-7


def test_writer_reproduces_tensorflows_bytes():
    """Re-encoding every scalar record of the fixture (its own wall time, step, tags, values) gives TensorFlow's bytes."""
    data = open(FIX, "rb").read()
    pos, n_checked = 0, 0
    while pos < len(data):
        (n,) = struct.unpack("<Q", data[pos:pos + 8])
        body = data[pos + 12:pos + 12 + n]
        rec = data[pos:pos + 16 + n]
        pos += 16 + n
        assert body[0] == (1 << 3 | 1)
        (wall,) = struct.unpack("<d", body[1:9])
        tmp = os.path.join(os.environ.get("TMPDIR", "/tmp"), "castrec_tb_one.tfevents")
        open(tmp, "wb").write(rec)
        got = T.read_events(tmp)
        if not got:
            assert b"brain.Event:2" in body                  # the file-version record
            assert T._event(wall, 0, file_version="brain.Event:2") == body
            continue
        step, scalars = got[0]
        mine = T._event(wall, step, scalars=list(scalars.items()))
        assert mine == body, (step, scalars)
        head = struct.pack("<Q", len(mine))
        assert head + struct.pack("<I", T._masked(head)) + mine + struct.pack("<I", T._masked(mine)) == rec
        n_checked += 1
    assert n_checked == 14


def test_writer_round_trip(tmp_path):
    w = T.EventWriter(str(tmp_path))
    w.add_scalars(1, {"TRAIN/loss": 1.25, "TRAIN/auc": 0.75})
    w.add_scalars(20, {"VALID/NDCG@10": 0.5, "VALID/HR@10": 0.625, "TEST/NDCG@10": 0.25, "TEST/HR@10": 0.375})
    w.close()
    assert os.path.basename(w.path).startswith("events.out.tfevents.")
    ev = T.read_events(w.path)
    assert ev == [(1, {"TRAIN/loss": 1.25, "TRAIN/auc": 0.75}),
                  (20, {"VALID/NDCG@10": 0.5, "VALID/HR@10": 0.625, "TEST/NDCG@10": 0.25, "TEST/HR@10": 0.375})]
    raw = open(w.path, "rb").read()
    raw = raw[:40] + bytes([raw[40] ^ 1]) + raw[41:]         # a flipped bit is caught by the record's CRC
    bad = tmp_path / "bad"
    bad.write_bytes(raw)
    try:
        T.read_events(str(bad))
        assert False, "corrupted record accepted"
    except ValueError:
        pass

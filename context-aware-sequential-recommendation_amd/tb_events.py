"""TensorBoard scalar summaries without TensorFlow or the tensorboard package.

The reference logs `TRAIN/loss`, `TRAIN/auc` once per epoch (the merged summary of the epoch's last step:
/root/reference/models/sasrec.py:112-125, main.py:203,222-224) and `VALID/NDCG@10`, `VALID/HR@10`, `TEST/NDCG@10`, `TEST/HR@10` at
every evaluation (main.py:240-249) through `tf.summary.FileWriter(<run dir>)`.  An event file is a TFRecord stream of serialized
`Event` protos; both formats are small enough to write by hand:

    record  = uint64 length | uint32 masked_crc32c(length) | data | uint32 masked_crc32c(data)
    Event   = { 1: double wall_time, 2: int64 step, 3: string file_version | 5: Summary }
    Summary = { 1 (repeated): Value { 1: string tag, 2: float simple_value } }

`tensorboard --logdir saved_models/...` reads the result like the reference's files."""
import os
import socket
import struct
import time

_POLY = 0x82F63B78                                   # CRC-32C (Castagnoli), reflected
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ _POLY if _c & 1 else _c >> 1
    _TABLE.append(_c)


def crc32c(data):
    c = 0xFFFFFFFF
    for b in data:
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _bytes_field(num, payload):
    return _varint(num << 3 | 2) + _varint(len(payload)) + payload


def _event(wall_time, step, file_version=None, scalars=None):
    ev = _varint(1 << 3 | 1) + struct.pack("<d", wall_time)
    if int(step) != 0:                                       # (proto3: a zero step is not written)
        ev += _varint(2 << 3 | 0) + _varint(int(step) & (2 ** 64 - 1))
    if file_version is not None:
        ev += _bytes_field(3, file_version.encode())
    if scalars:
        summary = b"".join(_bytes_field(1, _bytes_field(1, tag.encode()) + _varint(2 << 3 | 5) + struct.pack("<f", float(v)))
                           for tag, v in scalars)
        ev += _bytes_field(5, summary)
    return ev


class EventWriter:
    """`add_scalars(step, {'TRAIN/loss': ..})` appends one Event; files are named as tf.summary.FileWriter names them."""

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, "events.out.tfevents.%010d.%s" % (int(time.time()), socket.gethostname()))
        self._f = open(self.path, "wb")
        self._write(_event(time.time(), 0, file_version="brain.Event:2"))

    def _write(self, data):
        head = struct.pack("<Q", len(data))
        self._f.write(head + struct.pack("<I", _masked(head)) + data + struct.pack("<I", _masked(data)))

    def add_scalars(self, step, scalars):
        self._write(_event(time.time(), step, scalars=list(scalars.items())))

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def read_events(path):
    """[(step, {tag: value})] of an event file written by EventWriter or TensorFlow (scalar summaries only); checks every CRC."""
    out = []
    data = open(path, "rb").read()
    pos = 0

    def parse(buf):
        fields, i = [], 0
        while i < len(buf):
            key, sh = 0, 0
            while True:
                b = buf[i]; i += 1
                key |= (b & 0x7F) << sh; sh += 7
                if not b & 0x80:
                    break
            num, wt = key >> 3, key & 7
            if wt == 0:
                v, sh = 0, 0
                while True:
                    b = buf[i]; i += 1
                    v |= (b & 0x7F) << sh; sh += 7
                    if not b & 0x80:
                        break
            elif wt == 1:
                v = buf[i:i + 8]; i += 8
            elif wt == 5:
                v = buf[i:i + 4]; i += 4
            elif wt == 2:
                n, sh = 0, 0
                while True:
                    b = buf[i]; i += 1
                    n |= (b & 0x7F) << sh; sh += 7
                    if not b & 0x80:
                        break
                v = buf[i:i + n]; i += n
            else:
                raise ValueError("wire type %d" % wt)
            fields.append((num, wt, v))
        return fields

    while pos < len(data):
        head = data[pos:pos + 8]
        (n,) = struct.unpack("<Q", head)
        if struct.unpack("<I", data[pos + 8:pos + 12])[0] != _masked(head):
            raise ValueError("length CRC mismatch at %d" % pos)
        body = data[pos + 12:pos + 12 + n]
        if struct.unpack("<I", data[pos + 12 + n:pos + 16 + n])[0] != _masked(body):
            raise ValueError("data CRC mismatch at %d" % pos)
        pos += 16 + n
        step, scalars = 0, {}
        for num, wt, v in parse(body):
            if num == 2 and wt == 0:
                step = v
            elif num == 5 and wt == 2:
                for n2, w2, val in parse(v):
                    if n2 == 1 and w2 == 2:
                        tag, sv = None, None
                        for n3, w3, x in parse(val):
                            if n3 == 1 and w3 == 2:
                                tag = bytes(x).decode()
                            elif n3 == 2 and w3 == 5:
                                (sv,) = struct.unpack("<f", x)
                        if tag is not None and sv is not None:
                            scalars[tag] = sv
        if scalars:
            out.append((step, scalars))
    return out

"""TensorBoard event files for the training scalars, without TensorBoard.

Reference: /root/reference/bootstrapper/training.py:125-133 hands Lightning a
`TensorBoardLogger(setup_dir, name="log")`, and the LightningModules log `train_loss` every 10 steps
(models/3d_affs/train.py:155).  That writer lives in the `tensorboard` package, which this image does not have; the file
format is small: TFRecord framing (length, masked CRC-32C of the length, payload, masked CRC-32C of the payload) around
`Event` protocol buffers, of which a scalar needs four fields.  `ScalarWriter` writes what `tensorboard --logdir
<setup_dir>/log` reads; `read_scalars` is the inverse (tests, and a resume that wants the curve so far).
"""
import os
import socket
import struct
import time

_CRC_TABLE = []


def _crc_table():
    if not _CRC_TABLE:
        for n in range(256):
            c = n
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1   # CRC-32C (Castagnoli), reflected
            _CRC_TABLE.append(c)
    return _CRC_TABLE


def crc32c(data):
    t = _crc_table()
    c = 0xFFFFFFFF
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _field(number, wire, payload):
    return _varint(number << 3 | wire) + payload


def _bytes_field(number, data):
    return _field(number, 2, _varint(len(data)) + data)


def event_bytes(wall_time, step=None, file_version=None, scalars=()):
    """Event{wall_time = 1 (double), step = 2 (int64), file_version = 3 (string), summary = 5 (Summary{value = 1 (repeated
    Value{tag = 1 (string), simple_value = 2 (float)})})}"""
    ev = _field(1, 1, struct.pack("<d", wall_time))
    if step is not None:
        ev += _field(2, 0, _varint(int(step)))
    if file_version is not None:
        ev += _bytes_field(3, file_version.encode())
    if scalars:
        summary = b""
        for tag, value in scalars:
            summary += _bytes_field(1, _bytes_field(1, tag.encode()) + _field(2, 5, struct.pack("<f", float(value))))
        ev += _bytes_field(5, summary)
    return ev


def record_bytes(payload):
    head = struct.pack("<Q", len(payload))
    return head + struct.pack("<I", _masked(head)) + payload + struct.pack("<I", _masked(payload))


class ScalarWriter:
    """`<logdir>/version_<n>/events.out.tfevents.<time>.<host>.<pid>.0`, as Lightning's TensorBoardLogger lays it out
    (a new version directory per run unless `version` is given)."""

    def __init__(self, logdir, version=None):
        if version is None:
            taken = [int(d.split("_")[1]) for d in (os.listdir(logdir) if os.path.isdir(logdir) else [])
                     if d.startswith("version_") and d.split("_")[1].isdigit()]
            version = max(taken) + 1 if taken else 0
        self.dir = os.path.join(logdir, f"version_{version}")
        os.makedirs(self.dir, exist_ok=True)
        now = time.time()
        self.path = os.path.join(self.dir, f"events.out.tfevents.{int(now)}.{socket.gethostname()}.{os.getpid()}.0")
        self._f = open(self.path, "ab")
        self._f.write(record_bytes(event_bytes(now, file_version="brain.Event:2")))
        self._f.flush()

    def add_scalar(self, tag, value, step, wall_time=None):
        self._f.write(record_bytes(event_bytes(time.time() if wall_time is None else wall_time, step=step, scalars=[(tag, value)])))
        self._f.flush()

    def close(self):
        if self._f:
            self._f.close()
            self._f = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _read_varint(buf, i):
    n = shift = 0
    while True:
        b = buf[i]
        i += 1
        n |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            return n, i


def _fields(buf):
    i = 0
    while i < len(buf):
        key, i = _read_varint(buf, i)
        number, wire = key >> 3, key & 7
        if wire == 0:
            v, i = _read_varint(buf, i)
        elif wire == 1:
            v, i = buf[i:i + 8], i + 8
        elif wire == 2:
            n, i = _read_varint(buf, i)
            v, i = buf[i:i + n], i + n
        elif wire == 5:
            v, i = buf[i:i + 4], i + 4
        else:
            raise ValueError(f"wire type {wire}")
        yield number, wire, v


def read_scalars(path):
    """-> (file_version, [(step, tag, value, wall_time)]); raises ValueError on a bad checksum or a truncated record."""
    data = open(path, "rb").read()
    i, version, out = 0, None, []
    while i < len(data):
        if i + 12 > len(data):
            raise ValueError("truncated record header")
        head = data[i:i + 8]
        (n,) = struct.unpack("<Q", head)
        if struct.unpack("<I", data[i + 8:i + 12])[0] != _masked(head):
            raise ValueError("length checksum")
        payload = data[i + 12:i + 12 + n]
        if len(payload) != n or i + 16 + n > len(data):
            raise ValueError("truncated record")
        if struct.unpack("<I", data[i + 12 + n:i + 16 + n])[0] != _masked(payload):
            raise ValueError("payload checksum")
        i += 16 + n
        wall, step, summary = 0.0, 0, None
        for number, wire, v in _fields(payload):
            if number == 1 and wire == 1:
                (wall,) = struct.unpack("<d", v)
            elif number == 2 and wire == 0:
                step = v
            elif number == 3 and wire == 2:
                version = bytes(v).decode()
            elif number == 5 and wire == 2:
                summary = v
        if summary is not None:
            for number, wire, v in _fields(summary):
                if number != 1 or wire != 2:
                    continue
                tag, value = None, None
                for n2, w2, v2 in _fields(v):
                    if n2 == 1 and w2 == 2:
                        tag = bytes(v2).decode()
                    elif n2 == 2 and w2 == 5:
                        (value,) = struct.unpack("<f", v2)
                if tag is not None and value is not None:
                    out.append((step, tag, value, wall))
    return version, out

"""Dependency-free TFRecord + tf.train.Example reader/writer for the reference's slice files, and the batch queue that
stands in for `Trainer.next_batch` (source_segmenter.py:331-355, adversarial.py:607-631).

Record layout (source_segmenter.py:38-46): int64 features dsize_dim0/1/2, lsize_dim0/1/2 and two bytes features
`data_vol`, `label_vol`, each the raw bytes of a float32 [256,256,3] array (label_vol too: the code reshapes it to raw_size and
takes channel 1, source_segmenter.py:343-348).  Framing: u64 length | masked crc32c(length) | payload | masked crc32c(payload).
"""
import struct
import threading

import numpy as np

# ---- crc32c (Castagnoli), masked as TFRecord wants ----------------------------------------------------------------------
_T = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _T.append(_c)
_T8 = np.array(_T, dtype=np.uint32)


def crc32c(data):
    crc = 0xFFFFFFFF
    t = _T
    for b in bytes(data):
        crc = t[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---- minimal protobuf wire format ------------------------------------------------------------------------------------------
def _varint(n):
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _ld(field, payload):          # length-delimited field
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def encode_example(features):
    """features: dict name -> int | bytes  ->  serialized tf.train.Example"""
    entries = b""
    for name in sorted(features):
        v = features[name]
        if isinstance(v, (bytes, bytearray)):
            feat = _ld(1, _ld(1, bytes(v)))                      # Feature.bytes_list = 1 { BytesList.value = 1 }
        else:
            feat = _ld(3, _ld(1, _varint(int(v))))               # Feature.int64_list = 3 { Int64List.value = 1 (packed) }
        entries += _ld(1, _ld(1, name.encode()) + _ld(2, feat))  # Features.feature map entry {key=1, value=2}
    return _ld(1, entries)                                       # Example.features = 1


def _fields(buf):
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _read_varint(buf, pos)
        wt = key & 7
        if wt == 2:
            ln, pos = _read_varint(buf, pos)
            yield key >> 3, buf[pos:pos + ln]
            pos += ln
        elif wt == 0:
            v, pos = _read_varint(buf, pos)
            yield key >> 3, v
        elif wt == 5:
            yield key >> 3, buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            yield key >> 3, buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)


def decode_example(buf):
    buf = memoryview(buf)
    out = {}
    for f, features in _fields(buf):
        if f != 1:
            continue
        for f2, entry in _fields(features):
            if f2 != 1:
                continue
            name, feat = None, None
            for f3, v in _fields(entry):
                if f3 == 1:
                    name = bytes(v).decode()
                elif f3 == 2:
                    feat = v
            for kind, lst in _fields(feat):
                if kind == 1:                                     # bytes_list
                    out[name] = [bytes(v) for f4, v in _fields(lst) if f4 == 1][0]
                elif kind == 3:                                   # int64_list (packed or not)
                    vals = []
                    for f4, v in _fields(lst):
                        if f4 != 1:
                            continue
                        if isinstance(v, int):
                            vals.append(v)
                        else:
                            p = 0
                            while p < len(v):
                                x, p = _read_varint(v, p)
                                vals.append(x)
                    out[name] = vals[0] if len(vals) == 1 else vals
                elif kind == 2:                                   # float_list
                    raw = b"".join(bytes(v) for f4, v in _fields(lst) if f4 == 1)
                    out[name] = np.frombuffer(raw, dtype="<f4")
    return out


# ---- files -----------------------------------------------------------------------------------------------------------------
def write_records(path, payloads):
    with open(path, "wb") as f:
        for p in payloads:
            ln = struct.pack("<Q", len(p))
            f.write(ln)
            f.write(struct.pack("<I", masked_crc(ln)))
            f.write(p)
            f.write(struct.pack("<I", masked_crc(p)))


def read_records(path, verify=False):
    out = []
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                break
            if len(head) < 12:
                raise IOError("%s: truncated record header" % path)
            (ln,) = struct.unpack("<Q", head[:8])
            if verify and struct.unpack("<I", head[8:])[0] != masked_crc(head[:8]):
                raise IOError("%s: corrupt record length" % path)
            p = f.read(ln)
            tail = f.read(4)
            if len(p) < ln or len(tail) < 4:
                raise IOError("%s: truncated record" % path)
            if verify and struct.unpack("<I", tail)[0] != masked_crc(p):
                raise IOError("%s: corrupt record payload" % path)
            out.append(p)
    return out


def write_slice(path, data_vol, label_vol):
    """one-record file like the reference's dataset: data_vol, label_vol float32 [256,256,3]"""
    data_vol = np.ascontiguousarray(data_vol, dtype="<f4")
    label_vol = np.ascontiguousarray(label_vol, dtype="<f4")
    feats = {"dsize_dim0": data_vol.shape[0], "dsize_dim1": data_vol.shape[1], "dsize_dim2": data_vol.shape[2],
             "lsize_dim0": label_vol.shape[0], "lsize_dim1": label_vol.shape[1], "lsize_dim2": label_vol.shape[2],
             "data_vol": data_vol.tobytes(), "label_vol": label_vol.tobytes()}
    write_records(path, [encode_example(feats)])


def read_slice(path, raw_size=(256, 256, 3), verify=False):
    """-> float32 [H,W,4]: image channels 0:3 + the middle label slice (tf.slice(label_vol,[0,0,1],[H,W,1]))"""
    ex = decode_example(read_records(path, verify)[0])
    data = np.frombuffer(ex["data_vol"], dtype="<f4").reshape(raw_size)
    label = np.frombuffer(ex["label_vol"], dtype="<f4").reshape(raw_size)
    return np.concatenate([data, label[:, :, 1:2]], axis=2).astype(np.float32)


class SliceQueue(object):
    """string_input_producer(shuffle=True) + TFRecordReader + shuffle_batch stand-in: an endless shuffled stream of slices.
    Background threads keep `capacity` decoded slices ready on `num_threads` reader threads (the reference: 4 threads, capacity 120)."""

    def __init__(self, files, batch_size, capacity=120, min_after_dequeue=30, seed=0, raw_size=(256, 256, 3), threaded=True,
                 num_threads=4, shard=None):
        if not files:
            raise ValueError("SliceQueue: empty file list")
        files = list(files)
        if shard is not None:                    # data parallelism: rank r of W reads files r, r+W, ...
            rank, world = shard
            if len(files) >= world:
                files = files[rank::world]
            seed = seed + rank
        self.files, self.batch_size, self.raw_size = files, batch_size, raw_size
        self.rng = np.random.default_rng(seed)
        self._pick = np.random.default_rng(seed + 7919)     # consumer-side shuffle (the file-order rng belongs to the readers)
        self.capacity = max(capacity, batch_size)
        # tf.train.shuffle_batch(min_after_dequeue): a batch is only drawn while at least this many slices stay behind in the buffer
        # (the mixing depth of the shuffle); bounded by what the capacity and the dataset can ever hold
        self.min_after_dequeue = max(0, min(int(min_after_dequeue), self.capacity - batch_size, len(files) - batch_size))
        self._buf, self._cv = [], threading.Condition()
        self._order, self._order_lock, self._stop = [], threading.Lock(), False
        self._threads = []
        if threaded:
            for _ in range(max(int(num_threads), 1)):
                t = threading.Thread(target=self._fill, daemon=True)
                t.start()
                self._threads.append(t)
        self._thread = self._threads[0] if self._threads else None

    def _next_file(self):
        with self._order_lock:
            if not self._order:
                self._order = list(self.rng.permutation(len(self.files)))
            return self.files[self._order.pop()]

    def _fill(self):
        try:
            while not self._stop:
                with self._cv:
                    while len(self._buf) >= self.capacity and not self._stop:
                        self._cv.wait(0.05)
                f = self._next_file()
                item = (read_slice(f, self.raw_size), f)
                with self._cv:
                    self._buf.append(item)
                    self._cv.notify_all()
        except BaseException as e:      # surface reader failures in the consumer instead of hanging it
            with self._cv:
                self._error = e
                self._cv.notify_all()

    def next_batch(self, batch_size=None):
        B = batch_size or self.batch_size
        items = []
        if self._thread is None:
            for _ in range(B):
                f = self._next_file()
                items.append((read_slice(f, self.raw_size), f))
        else:
            with self._cv:
                while len(self._buf) < min(B + self.min_after_dequeue, self.capacity):
                    if getattr(self, "_error", None) is not None:
                        raise IOError("SliceQueue reader thread failed: %r" % (self._error,))
                    self._cv.wait(0.05)
                idx = sorted(self._pick.choice(len(self._buf), size=B, replace=False), reverse=True)
                items = [self._buf.pop(i) for i in idx]
                self._cv.notify_all()
        return np.stack([a for a, _ in items]), [f for _, f in items]

    def close(self):
        self._stop = True

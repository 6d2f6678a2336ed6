"""not gpu: tfrecord framing + tf.train.Example codec round trips (config 1 "identical tfrecord inputs"), SliceQueue batches,
entry-script phase configuration (train_gan.py:85-126)."""
import os

import numpy as np

from conftest import pkg


def test_crc32c_and_mask_kats():
    t = pkg("tfrecord")
    assert t.crc32c(b"123456789") == 0xE3069283              # the standard CRC-32C check value
    assert t.crc32c(b"") == 0
    c = t.crc32c(b"\x00" * 8)
    assert t.masked_crc(b"\x00" * 8) == ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def test_example_roundtrip_and_slice_layout(tmp_path):
    t = pkg("tfrecord")
    rng = np.random.default_rng(0)
    img = rng.standard_normal((16, 16, 3)).astype(np.float32)
    lab = rng.integers(0, 5, size=(16, 16, 3)).astype(np.float32)
    p = str(tmp_path / "a.tfrecords")
    t.write_slice(p, img, lab)
    recs = t.read_records(p, verify=True)
    assert len(recs) == 1
    ex = t.decode_example(recs[0])
    assert sorted(ex) == ["data_vol", "dsize_dim0", "dsize_dim1", "dsize_dim2", "label_vol", "lsize_dim0", "lsize_dim1", "lsize_dim2"]
    assert ex["dsize_dim0"] == 16 and ex["lsize_dim2"] == 3
    s = t.read_slice(p, raw_size=(16, 16, 3), verify=True)
    assert s.shape == (16, 16, 4) and np.array_equal(s[:, :, :3], img) and np.array_equal(s[:, :, 3], lab[:, :, 1])   # middle label slice
    raw = bytearray(open(p, "rb").read())
    raw[40] ^= 0xFF
    open(p, "wb").write(raw)
    try:
        t.read_records(p, verify=True)
        assert False, "corruption not detected"
    except IOError:
        pass


def test_slice_queue_batches(tmp_path):
    syn, t = pkg("synthetic"), pkg("tfrecord")
    files = syn.write_dataset(str(tmp_path / "d"), 5, seed=0, size=32)
    assert os.path.exists(str(tmp_path / "d" / "slice_list"))
    q = t.SliceQueue(files, 4, capacity=8, raw_size=(32, 32, 3), threaded=True)
    seen = set()
    for _ in range(4):
        b, ids = q.next_batch(4)
        assert b.shape == (4, 32, 32, 4) and b.dtype == np.float32 and len(ids) == 4
        assert set(np.unique(b[..., 3])) <= {0.0, 1.0, 2.0, 3.0, 4.0}
        seen |= set(ids)
    q.close()
    assert seen == set(files)
    q2 = t.SliceQueue(files, 2, raw_size=(32, 32, 3), threaded=False)
    assert q2.next_batch()[0].shape == (2, 32, 32, 4)


def test_train_gan_phase_configuration():
    tg = pkg("train_gan")
    ck, nc, tc = tg.configure("pre-train")
    assert nc["ct_front_trainable"] is False and ck["lambda_mask_loss"] == 0
    assert (tc["gen_interval"], tc["dis_sub_iter"], tc["dis_sub_iter_inc"], tc["training_iters"], tc["epochs"]) == (0, 1, 0, 201, 100)
    ck, nc, tc = tg.configure("train-gan")
    assert nc["ct_front_trainable"] is True and ck["lambda_mask_loss"] == 0.3
    assert (tc["dis_sub_iter"], tc["gen_sub_iter"], tc["iter_upd_interval"], tc["dis_sub_iter_inc"]) == (20, 1, 300, 1)
    ck, nc, tc = tg.configure("fine-tune")
    assert tc["dis_sub_iter"] == 30 and tc["lr_update"] is False
    assert tg.opt_kwargs["learning_rate"] == 3e-4 and ck["miu_dis"] == 0.002


def test_slice_queue_shards_and_reader_threads(tmp_path):
    """rank r of W reads its own share of the list; several reader threads feed one queue"""
    syn, t = pkg("synthetic"), pkg("tfrecord")
    files = syn.write_dataset(str(tmp_path / "d"), 8, seed=0, size=16)
    seen = []
    for r in range(2):
        q = t.SliceQueue(files, 2, capacity=4, raw_size=(16, 16, 3), num_threads=3, shard=(r, 2))
        s = set()
        for _ in range(6):
            s |= set(q.next_batch(2)[1])
        q.close()
        seen.append(s)
    assert seen[0] == set(files[0::2]) and seen[1] == set(files[1::2])
    q = t.SliceQueue(files[:1], 1, raw_size=(16, 16, 3), shard=(1, 2), threaded=False)     # fewer files than ranks: not sharded
    assert q.files == files[:1]


def test_device_feeder_matches_host_one_hot(tmp_path):
    """feeder.DeviceFeeder: the staged batch equals the dequeued one and its on-device one-hot equals lib._label_decomp (lib.py:75-92)"""
    import torch
    F, L = pkg("feeder"), pkg("lib")

    class Src(object):
        def __init__(self):
            self.k = 0
            self.rng = np.random.default_rng(0)
            self.batches = []

        def next_batch(self, B):
            b = self.rng.standard_normal((B, 8, 8, 4)).astype(np.float32)
            b[..., 3] = self.rng.integers(0, 7, (B, 8, 8))      # labels 5, 6 >= num_cls -> all-zero rows
            self.batches.append(b.copy())
            self.k += 1
            return b, ["id%d" % self.k] * B

    src = Src()
    f = F.DeviceFeeder(src, 3, 5, "cpu", depth=2)
    for i in range(5):
        x, y, ids = f.next()
        ref = src.batches[i]
        assert ids == ["id%d" % (i + 1)] * 3
        assert torch.equal(x, torch.from_numpy(ref[..., 0:3]))
        assert np.array_equal(y.numpy(), L._label_decomp(5, ref[..., 3]))
    f.close()

    class Bad(object):
        def next_batch(self, B):
            raise RuntimeError("disk on fire")

    fb = F.DeviceFeeder(Bad(), 2, 5, "cpu")
    import pytest
    with pytest.raises(IOError):
        fb.next()


def test_example_wire_format_known_answers():
    """tf.train.Example bytes written out by hand from the protobuf wire format (Example.features=1 -> map<string,Feature> entries
    {key=1, value=2} -> Feature{bytes_list=1 | float_list=2 | int64_list=3} -> value=1, int64 packed) — not through this codec's own
    encoder — and the TFRecord framing of a record (u64 length, masked crc32c of the length, payload, masked crc32c of the payload)."""
    import struct
    t = pkg("tfrecord")
    ex_int = bytes.fromhex("0a0c" "0a0a" "0a0161" "1205" "1a03" "0a0105")               # {"a": int64 5}
    assert t.decode_example(ex_int) == {"a": 5}
    assert t.encode_example({"a": 5}) == ex_int
    ex_bytes = bytes.fromhex("0a0f" "0a0d" "0a026964" "1207" "0a05" "0a03" "78797a")    # {"id": b"xyz"}
    assert t.decode_example(ex_bytes) == {"id": b"xyz"}
    assert t.encode_example({"id": b"xyz"}) == ex_bytes
    # int64 300 = varint ac 02; UNPACKED encoding (tag 08 per value) and a two-value packed list must decode too
    assert t.decode_example(bytes.fromhex("0a0d" "0a0b" "0a0161" "1206" "1a04" "0a02ac02")) == {"a": 300}
    assert t.decode_example(bytes.fromhex("0a0c" "0a0a" "0a0161" "1205" "1a03" "08ac02")) == {"a": 300}            # unpacked varint
    assert t.decode_example(bytes.fromhex("0a0e" "0a0c" "0a0161" "1207" "1a05" "0a03" "05ac02")) == {"a": [5, 300]}
    # float_list (not produced by the reference's writer, accepted by the reader): {"f": [1.0]}
    got = t.decode_example(bytes.fromhex("0a0f" "0a0d" "0a0166" "1208" "1206" "0a04" "0000803f"))
    assert list(got["f"]) == [1.0]
    # framing: crc32c("123456789") = e3069283 is the standard check value; mask = ((crc >> 15 | crc << 17) + a282ead8) mod 2^32
    assert t.crc32c(b"123456789") == 0xE3069283
    crc = 0xE3069283
    assert t.masked_crc(b"123456789") == ((((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF)
    import tempfile, os
    d = tempfile.mkdtemp()
    f = os.path.join(d, "r.tfrecords")
    t.write_records(f, [ex_int])
    raw = open(f, "rb").read()
    assert raw[:8] == struct.pack("<Q", len(ex_int)) and raw[12:12 + len(ex_int)] == ex_int and len(raw) == 8 + 4 + len(ex_int) + 4
    assert struct.unpack("<I", raw[8:12])[0] == t.masked_crc(raw[:8]) and struct.unpack("<I", raw[-4:])[0] == t.masked_crc(ex_int)
    assert t.read_records(f, verify=True) == [ex_int]

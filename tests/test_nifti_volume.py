"""not gpu: the NIfTI-1 reader/writer behind lib.read_nii_image / write_nii (lib.py:47-72) against hand-built files that follow the
format definition, and the batching / bookkeeping of the volume-inference loop (adversarial.py:993-1084) with a stub predictor."""
import gzip
import struct

import numpy as np
import pytest

from conftest import pkg


def _header(e, shape, code, bitpix, slope=0.0, inter=0.0, sform=None, qform=None, pixdim=(1, 1, 1, 1, 1, 1, 1, 1)):
    h = bytearray(348)
    struct.pack_into(e + "i", h, 0, 348)
    struct.pack_into(e + "8h", h, 40, len(shape), *(list(shape) + [1] * (7 - len(shape))))
    struct.pack_into(e + "2h", h, 70, code, bitpix)
    struct.pack_into(e + "8f", h, 76, *pixdim)
    struct.pack_into(e + "f", h, 108, 352.0)
    struct.pack_into(e + "2f", h, 112, slope, inter)
    if sform is not None:
        struct.pack_into(e + "2h", h, 252, 0, 1)
        struct.pack_into(e + "12f", h, 280, *np.asarray(sform, dtype=np.float64)[:3].reshape(-1))
    elif qform is not None:
        struct.pack_into(e + "2h", h, 252, 1, 0)
        struct.pack_into(e + "6f", h, 256, *qform)
    h[344:348] = b"n+1\0"
    return bytes(h) + b"\0\0\0\0"


def test_reads_hand_built_files(tmp_path):
    N = pkg("nifti")
    # x-fastest voxel order, int16, big-endian, with an sform
    vox = np.arange(2 * 3 * 4, dtype=">i2")
    aff = np.array([[2.0, 0, 0, -10], [0, 3.0, 0, 5], [0, 0, 4.0, 7], [0, 0, 0, 1]])
    p = tmp_path / "be.nii"
    p.write_bytes(_header(">", (2, 3, 4), 4, 16, sform=aff) + vox.tobytes())
    img = N.load(str(p))
    d = img.get_data()
    assert d.shape == (2, 3, 4) and d.dtype == np.int16
    assert d[1, 0, 0] == 1 and d[0, 1, 0] == 2 and d[0, 0, 1] == 6 and d[1, 2, 3] == 23
    assert np.array_equal(img.get_affine(), aff)
    # scl_slope / scl_inter scaling, uint8, little-endian, gzip, no s/qform -> pixdim diagonal
    p2 = tmp_path / "sc.nii.gz"
    with gzip.open(str(p2), "wb") as f:
        f.write(_header("<", (2, 2), 2, 8, slope=0.5, inter=-1.0, pixdim=(1, 1.5, 2.5, 3.5, 1, 1, 1, 1)) + bytes([0, 2, 4, 255]))
    img2 = N.load(str(p2))
    assert np.array_equal(img2.get_data(), np.array([[-1.0, 1.0], [0.0, 126.5]]))
    assert np.array_equal(img2.affine, np.diag([1.5, 2.5, 3.5, 1.0]))
    # qform: quaternion (b,c,d) = (0,0,1): 180 degrees about z -> diag(-1,-1,1) * pixdim, qfac = -1 flips z
    p3 = tmp_path / "q.nii"
    p3.write_bytes(_header("<", (1, 1, 1), 16, 32, qform=(0.0, 0.0, 1.0, 1.0, 2.0, 3.0), pixdim=(-1, 2, 2, 2, 1, 1, 1, 1)) +
                   np.float32(7.5).tobytes())
    img3 = N.load(str(p3))
    assert img3.get_data()[0, 0, 0] == 7.5
    assert np.allclose(img3.affine, [[-2, 0, 0, 1], [0, -2, 0, 2], [0, 0, -2, 3], [0, 0, 0, 1]])
    # malformed files are rejected
    bad = tmp_path / "bad.nii"
    bad.write_bytes(b"\0" * 400)
    with pytest.raises(IOError):
        N.load(str(bad))
    trunc = tmp_path / "tr.nii"
    trunc.write_bytes(_header("<", (4, 4, 4), 16, 32) + b"\0" * 10)
    with pytest.raises(IOError):
        N.load(str(trunc))


@pytest.mark.parametrize("dtype", ["uint8", "int16", "int32", "float32", "float64", "uint16", "int64"])
@pytest.mark.parametrize("ext", [".nii", ".nii.gz"])
def test_write_read_round_trip(tmp_path, dtype, ext):
    L = pkg("lib")
    rng = np.random.default_rng(0)
    a = (rng.standard_normal((5, 4, 3)) * 50).astype(dtype)
    aff = np.array([[0.0, -1.2, 0, 3], [1.1, 0, 0, -4], [0, 0, 2.0, 9], [0, 0, 0, 1]])
    fid = L.write_nii(a, "v" + ext, str(tmp_path), affine=aff)
    back = L.read_nii_image(fid)
    assert back.dtype == a.dtype and np.array_equal(back, a)
    obj = L.read_nii_object(fid)
    assert np.allclose(obj.get_affine(), aff, atol=1e-6) and obj.shape == (5, 4, 3)
    raw = (gzip.open(fid, "rb") if ext.endswith("gz") else open(fid, "rb")).read()
    assert struct.unpack("<8f", raw[76:108])[1:4] == pytest.approx((1.1, 1.2, 2.0))      # zooms from the affine columns
    assert L.read_nii_image(L.write_nii(a, "noaff" + ext, str(tmp_path))).shape == (5, 4, 3)    # default affine path


def test_save_nii_prediction(tmp_path):
    L = pkg("lib")
    ref = L.write_nii(np.zeros((4, 4, 2), np.float32), "ref.nii", str(tmp_path), affine=np.diag([2.0, 2.0, 5.0, 1.0]))
    gth = np.array([[0, 1], [7, 4]], dtype=np.float64).reshape(2, 2, 1)
    L._save_nii_prediction(gth, np.ones((2, 2, 1)), ref, str(tmp_path / "out"), "dense_pred_ref.nii", num_cls=5)
    g = L.read_nii_object(str(tmp_path / "out" / "gth_dense_pred_ref.nii.gz"))
    assert np.array_equal(g.get_data()[..., 0], [[0, 1], [0, 4]])               # labels above num_cls-1 zeroed
    assert np.array_equal(g.get_affine(), np.diag([2.0, 2.0, 5.0, 1.0]))
    assert L.read_nii_image(str(tmp_path / "out" / "dense_pred_ref.nii.gz")).sum() == 4


def test_volume_loop_bookkeeping(tmp_path):
    """depth 7, batch 2: 3 batches over frames 1..5; the last batch is half zero-filled and still counted (reference behaviour)"""
    VE, L = pkg("volume_eval"), pkg("lib")
    H = 256
    rng = np.random.default_rng(3)
    raw = rng.standard_normal((H, H, 7)).astype(np.float32)
    raw_y = rng.integers(0, 5, (H, H, 7)).astype(np.float32)
    img = L.write_nii(raw, "img.nii.gz", str(tmp_path))
    lab = L.write_nii(raw_y, "lab.nii.gz", str(tmp_path))
    calls = []

    def predict(vol, slice_y):          # "perfect" predictor that also checks the 3-slice context of each input
        calls.append((vol.copy(), slice_y.copy()))
        pred = slice_y.astype(np.int64)
        cm = np.zeros((5, 5))
        np.add.at(cm, (slice_y.astype(int).ravel(), pred.ravel()), 1)
        return pred, cm

    for shuffle in (False, True):
        calls.clear()
        sl, all_cm = VE.test_eval(predict, [lab], [img], 2, 5, flip_correction=True, shuffle=shuffle, rng=np.random.default_rng(0))
        assert len(calls) == 3 and len(sl) == 1
        fr = np.flip(np.flip(raw, 0), 1)
        fy = np.flip(np.flip(raw_y, 0), 1)
        seen = []
        for vol, sy in calls:
            for b in range(2):
                if not vol[b].any():
                    assert not sy[b].any()
                    continue
                jj = [k for k in range(1, 6) if np.array_equal(sy[b], fy[..., k])]
                assert len(jj) == 1 and np.array_equal(vol[b], fr[..., jj[0] - 1:jj[0] + 2])
                seen.append(jj[0])
        assert sorted(seen) == [1, 2, 3, 4, 5]
        if not shuffle:
            assert seen == [1, 2, 3, 4, 5]
        assert all_cm.sum() == 3 * 2 * H * H                                    # zero-filled slice included
        assert all_cm[0, 0] >= H * H and np.all(all_cm - np.diag(np.diag(all_cm)) == 0)
        assert np.allclose(sl[0][0], 1.0) and np.allclose(sl[0][1], 1.0)        # perfect Dice / Jaccard
    tmp_y, _ = VE.eval_volume(predict, fr, fy, 2, 5, shuffle=False)
    assert np.array_equal(tmp_y[..., 1:6], fy[..., 1:6]) and not tmp_y[..., 0].any() and not tmp_y[..., 6].any()
    # depth 5, batch 2: 2 batches cover frames 1..3 of 3 usable -> one zero slot; depth 3, batch 4: no batch at all
    tmp_y, cm = VE.eval_volume(predict, fr[..., :3], fy[..., :3], 4, 5, shuffle=False)
    assert cm.sum() == 0 and not tmp_y.any()
    with pytest.raises(ValueError):
        VE.eval_volume(predict, fr[:100], fy[:100], 2, 5)
    with pytest.raises(Exception):
        VE.test_eval(predict, [lab], [str(tmp_path / "missing.nii")], 2, 5)
    d, j = VE.sample_metric_stddev([(np.arange(5) / 10.0, np.arange(5) / 20.0), (np.arange(5) / 5.0, np.arange(5) / 10.0)], 5,
                                   {"bg": 0, "a": 1, "b": 2, "c": 3, "d": 4}, quiet=True)
    assert np.allclose(d, np.arange(5) * 0.15) and j.shape == (1, 2) and np.allclose(j, [[0.0, 0.0]])   # the reference's [:1] slip

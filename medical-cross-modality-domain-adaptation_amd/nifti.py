"""Dependency-free NIfTI-1 single-file (.nii / .nii.gz) reader and writer — the part of nibabel 2.1 that the reference's lib.py uses
(`nib.load(fid).get_data()`, `.get_affine()`, `nib.Nifti1Image(data, affine).to_filename(fid)`; lib.py:47-72).  nibabel is not
installed here, so the NIfTI-1.1 header layout is restated from the format definition:

  offset  type        field                     offset  type        field
  0       int32       sizeof_hdr = 348          112     float32     scl_slope
  40      int16[8]    dim (dim[0] = rank)       116     float32     scl_inter
  70      int16       datatype                  252     int16       qform_code
  72      int16       bitpix                    254     int16       sform_code
  76      float32[8]  pixdim (pixdim[0]=qfac)   256     float32[6]  quatern_b,c,d, qoffset_x,y,z
  108     float32     vox_offset (>= 352)       280     float32[12] srow_x, srow_y, srow_z
  344     char[4]     magic "n+1\\0"             348     char[4]     extension flag

Voxels are stored x-fastest (Fortran order).  `get_data()` returns an array of shape dim[1..rank], scaled by scl_slope/scl_inter
when the slope is non-zero and (slope, inter) != (1, 0), like nibabel's.  Byte order is detected from sizeof_hdr.
"""
import gzip
import os
import struct

import numpy as np

_DTYPES = {2: "u1", 4: "i2", 8: "i4", 16: "f4", 64: "f8", 256: "i1", 512: "u2", 768: "u4", 1024: "i8", 1280: "u8"}
_CODES = {np.dtype(v).newbyteorder("=").str[1:]: k for k, v in _DTYPES.items()}


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith(".gz") else open(path, mode)


def _quaternion_affine(b, c, d, qoff, pixdim):
    a2 = 1.0 - (b * b + c * c + d * d)
    a = np.sqrt(a2) if a2 > 1e-7 else 0.0
    if a2 <= 1e-7:      # 180 degree rotation: renormalise (b, c, d)
        n = np.sqrt(b * b + c * c + d * d)
        b, c, d = b / n, c / n, d / n
    R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                  [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                  [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
    qfac = -1.0 if pixdim[0] < 0 else 1.0
    R = R * np.array([pixdim[1], pixdim[2], pixdim[3] * qfac])[None, :]
    aff = np.eye(4)
    aff[:3, :3] = R
    aff[:3, 3] = qoff
    return aff


class Nifti1Image(object):
    """the subset of nibabel.Nifti1Image the reference touches: get_data(), get_affine() / .affine, .shape, to_filename()"""

    def __init__(self, dataobj, affine=None):
        self._data = np.asarray(dataobj)
        self.affine = np.eye(4) if affine is None else np.asarray(affine, dtype=np.float64).reshape(4, 4)

    @property
    def shape(self):
        return self._data.shape

    def get_data(self):
        return self._data

    get_fdata = get_data

    def get_affine(self):
        return self.affine

    def to_filename(self, path):
        save(self, path)


def load(path):
    """nib.load: parse header + voxels of a single-file NIfTI-1 volume"""
    with _open(path, "rb") as f:
        raw = f.read()
    if len(raw) < 352:
        raise IOError("%s: shorter than a NIfTI-1 header" % path)
    if struct.unpack("<i", raw[0:4])[0] == 348:
        e = "<"
    elif struct.unpack(">i", raw[0:4])[0] == 348:
        e = ">"
    else:
        raise IOError("%s: sizeof_hdr is not 348 (not a NIfTI-1 file)" % path)
    magic = raw[344:348]
    if magic[:3] != b"n+1":
        raise IOError("%s: magic %r — only single-file NIfTI-1 ('n+1') is supported" % (path, magic))
    dim = struct.unpack(e + "8h", raw[40:56])
    rank = dim[0]
    if not 1 <= rank <= 7:
        raise IOError("%s: bad dim[0] = %d" % (path, rank))
    shape = tuple(int(d) for d in dim[1:1 + rank])
    datatype = struct.unpack(e + "h", raw[70:72])[0]
    if datatype not in _DTYPES:
        raise IOError("%s: unsupported datatype code %d" % (path, datatype))
    pixdim = struct.unpack(e + "8f", raw[76:108])
    vox_offset = int(struct.unpack(e + "f", raw[108:112])[0])
    slope, inter = struct.unpack(e + "2f", raw[112:120])
    qform_code, sform_code = struct.unpack(e + "2h", raw[252:256])
    dt = np.dtype(e + _DTYPES[datatype])
    n = int(np.prod(shape))
    start = max(vox_offset, 352)
    if len(raw) < start + n * dt.itemsize:
        raise IOError("%s: truncated voxel data" % path)
    data = np.frombuffer(raw, dtype=dt, count=n, offset=start).reshape(shape, order="F")
    data = data.astype(dt.newbyteorder("="))
    if slope != 0 and not np.isnan(slope) and (slope, inter) != (1.0, 0.0):
        data = data.astype(np.float64) * slope + inter
    if sform_code > 0:
        aff = np.eye(4)
        aff[:3, :] = np.array(struct.unpack(e + "12f", raw[280:328])).reshape(3, 4)
    elif qform_code > 0:
        q = struct.unpack(e + "6f", raw[256:280])
        aff = _quaternion_affine(q[0], q[1], q[2], q[3:6], pixdim)
    else:
        aff = np.diag([pixdim[1], pixdim[2], pixdim[3], 1.0])
    return Nifti1Image(data, aff)


def save(img, path):
    """Nifti1Image.to_filename: little-endian single file, sform = affine (code 2, 'aligned'), qform unset, no scaling"""
    data = np.asarray(img.get_data())
    if data.dtype == np.bool_:
        data = data.astype(np.uint8)
    key = data.dtype.newbyteorder("=").str[1:]
    if key not in _CODES:
        raise ValueError("cannot store dtype %s in NIfTI-1" % data.dtype)
    if not 1 <= data.ndim <= 7:
        raise ValueError("NIfTI-1 holds 1 to 7 dimensions")
    code = _CODES[key]
    aff = np.asarray(img.affine, dtype=np.float64)
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    dim = [data.ndim] + list(data.shape) + [1] * (7 - data.ndim)
    struct.pack_into("<8h", hdr, 40, *dim)
    struct.pack_into("<2h", hdr, 70, code, data.dtype.itemsize * 8)
    zooms = np.sqrt((aff[:3, :3] ** 2).sum(0))
    pixdim = [1.0] + [float(zooms[i]) if i < 3 else 1.0 for i in range(data.ndim)] + [1.0] * (7 - data.ndim)
    struct.pack_into("<8f", hdr, 76, *pixdim)
    struct.pack_into("<f", hdr, 108, 352.0)
    struct.pack_into("<2f", hdr, 112, 1.0, 0.0)
    hdr[123] = 2                                        # xyzt_units: millimetres
    struct.pack_into("<2h", hdr, 252, 0, 2)
    struct.pack_into("<12f", hdr, 280, *aff[:3, :].reshape(-1))
    hdr[344:348] = b"n+1\0"
    payload = bytes(hdr) + b"\0\0\0\0" + np.asfortranarray(data.astype(data.dtype.newbyteorder("<"))).tobytes(order="F")
    d = os.path.dirname(str(path))
    if d:
        os.makedirs(d, exist_ok=True)
    with _open(path, "wb") as f:
        f.write(payload)
    return path

"""Drop-in for the step-path helpers of the reference's lib.py (label one-hot, Dice / Jaccard bookkeeping,
list reading, NIfTI volume I/O).  The NIfTI helpers (lib.py:31-72) run on this package's own reader/writer (nifti.py) because
nibabel is not available; the TF checkpoint helper `_save` (lib.py:23-29) is replaced by Full_DRN.save / .restore (.npz keyed by the
TF variable names).
"""
import os

import numpy as np
import torch

from . import kernels as K


def _read_lists(fid):
    """lib.py:7-21: text file -> list of lines (blank-ish lines dropped)"""
    if not os.path.isfile(fid):
        return None
    with open(fid, 'r') as fd:
        lines = fd.readlines()
    out = []
    for item in lines:
        if len(item) < 3:
            continue
        out.append(item.split('\n')[0])
    return out


def write_nii(array_data, filename, path="", affine=None):
    """lib.py:47-62: write a numpy array into a nii file; returns the path"""
    from . import nifti
    if affine is None:
        print("No information about the global coordinate system")
        affine = np.diag([1, 1, 1, 1])
    save_fid = os.path.join(path, filename)
    try:
        nifti.Nifti1Image(array_data, affine).to_filename(save_fid)
        print("Nii object %s has been saved!" % save_fid)
    except Exception:
        raise Exception("file %s cannot be saved!" % save_fid)
    return save_fid


def read_nii_image(input_fid):
    """lib.py:64-67: the voxel array of a nii file"""
    from . import nifti
    return nifti.load(input_fid).get_data()


def read_nii_object(input_fid):
    """lib.py:69-72: the nii object itself (get_data(), get_affine())"""
    from . import nifti
    return nifti.load(input_fid)


def _save_nii_prediction(gth, comp_pred, ref_fid, out_folder, out_bname, debug=False, num_cls=5):
    """lib.py:31-45: save prediction and ground truth as nii.gz with the reference volume's affine.  (The reference body reads
    `self.num_cls` inside a module-level function — a NameError; num_cls is an argument here.)"""
    ref_affine = read_nii_object(ref_fid).get_affine()
    out_bname = out_bname.split(".")[0] + ".nii.gz"
    write_nii(comp_pred, out_bname, out_folder, affine=ref_affine)
    _local_gth = gth.copy()
    _local_gth[_local_gth > num_cls - 1] = 0
    write_nii(_local_gth, "gth_" + out_bname, out_folder, affine=ref_affine)


def _label_decomp(num_cls, label_vol):
    """lib.py:75-92: integer-valued label map [B,H,W] -> one-hot float32 [B,H,W,num_cls]
    (labels >= num_cls give an all-zero row).  Host-side numpy like the reference (it runs on the dequeued batch)."""
    label_vol = np.asarray(label_vol)
    out = np.zeros(label_vol.shape + (num_cls,), dtype=np.float32)
    for i in range(num_cls):
        out[..., i][label_vol == i] = 1.0
    return out


def label_decomp_device(num_cls, label_vol):
    """same as _label_decomp for a device tensor: pnp_label_decomp"""
    return K.label_decomp(label_vol, num_cls)


def _dice_eval(compact_pred, labels, n_class):
    """lib.py:96-110: hard Dice of the argmax map vs one-hot labels -> (mean dice, [per-class dice]) on device"""
    out = K.dice_eval(compact_pred.contiguous(), labels.contiguous())
    return out[0], [out[1 + i] for i in range(n_class)]


def compact_and_confusion(y, compact_pred=None):
    """compact_y = tf.argmax(y, 3) and tf.confusion_matrix(compact_y, compact_pred) (source_segmenter.py:83-85; rows = ground truth,
    cols = prediction) in one kernel (pnp_confusion_matrix).  Returns (compact_y device int64, confusion matrix as numpy or None)."""
    cy, cm = K.confusion_matrix(y.contiguous(), compact_pred.contiguous() if compact_pred is not None else None)
    return cy, (cm.cpu().numpy() if cm is not None else None)


def _inverse_lookup(my_dict, _value):
    """lib.py:113-118: first key holding _value (contour_map name of a label id)"""
    for key, dic_value in list(my_dict.items()):
        if dic_value == _value:
            return key
    return None


def _jaccard(conf_matrix):
    """lib.py:121-134: per-class intersection over union from a confusion matrix (rows = truth, cols = prediction); 0 for an empty class"""
    cm = np.asarray(conf_matrix, dtype=np.float64)
    hit = np.diag(cm)
    union = cm.sum(axis=0) + cm.sum(axis=1) - hit
    return np.divide(hit, union, out=np.zeros_like(hit), where=union != 0)


def _dice(conf_matrix):
    """lib.py:137-151: per-class Dice 2|P n G| / (|P| + |G|) from a confusion matrix; 0 for an empty class"""
    cm = np.asarray(conf_matrix, dtype=np.float64)
    total = cm.sum(axis=0) + cm.sum(axis=1)
    return np.divide(2.0 * np.diag(cm), total, out=np.zeros(cm.shape[0]), where=total != 0)


def _indicator_eval(cm, verbose=True):
    """lib.py:155-176"""
    contour_map = {"bg": 0, "la_myo": 1, "la_blood": 2, "lv_blood": 3, "aa": 4}
    dice = _dice(cm)
    jaccard = _jaccard(cm)
    if verbose:
        print(cm)
        for organ, ind in list(contour_map.items()):
            if ind < len(dice):
                print("organ: %s" % organ)
                print("dice: %s" % dice[int(ind)])
                print("jaccard: %s" % jaccard[int(ind)])
    return dice, jaccard


def atomic_savez(path, **arrays):
    """np.savez into `path` through a temporary file in the same directory + os.replace (periodic checkpoints overwrite in place)"""
    import os
    tmp = path + ".tmp.npz"
    np.savez(tmp, **arrays)
    os.replace(tmp, path)
    return path

"""Volume inference (`Trainer.test_eval` of adversarial.py:993-1052 and source_segmenter.py:572-632): a NIfTI volume is cut into
3-adjacent-slice inputs, segmented batch by batch with every BN in inference mode, the hard predictions are reassembled into a label
volume and a per-volume confusion matrix yields per-organ Dice / Jaccard.  The forward itself is the same HIP path as training.

Behaviour kept from the reference, quirks included:
  * `flip_correction` flips axes 0 and 1 of image and label volume (orientation mismatch between the tfrecords and the nii files);
  * the number of batches is `depth // batch_size`; usable frames are 1 .. depth-2 (they need both neighbours).  adversarial.py
    visits them in a np.random.shuffle'd order; batches that run past the frame list are zero-filled and STILL counted in the
    confusion matrix (an all-zero slice against an all-background label), and frames left over keep prediction 0;
  * source_segmenter.py's loop (its line 609 is a SyntaxError, and frame 0 would slice raw[..., -1:2]) is given the same 1 .. depth-2
    frame set in ascending order — the only reading under which it runs;
  * `sample_metric_stddev` returns `(mean_dice_per_class, subject_level_list[:1])` — the second value is the reference's own slip
    (first ROW of the [num_cls, 2] mean matrix instead of the Jaccard column); kept because callers see exactly that.
"""
import logging
import os

import numpy as np

from .lib import _dice, _jaccard, read_nii_image


def frames_of(depth, shuffle, rng=None):
    frames = [kk for kk in range(1, depth - 1)]
    if shuffle:
        (rng if rng is not None else np.random).shuffle(frames)
    return frames


def eval_volume(predict, raw, raw_y, batch_size, num_cls, shuffle=True, rng=None, in_size=(256, 256, 3)):
    """-> (label volume of predictions, confusion matrix).  predict(vol [B,H,W,3] float32, slice_y [B,H,W] float32) ->
    (compact_pred [B,H,W] integer array, confusion matrix [num_cls,num_cls])"""
    if raw.shape[:2] != tuple(in_size[:2]) or raw_y.shape != raw.shape:
        raise ValueError("volume %s / label %s: expected %dx%dxD pairs" % (raw.shape, raw_y.shape, in_size[0], in_size[1]))
    tmp_y = np.zeros(raw_y.shape)
    cm = np.zeros([num_cls, num_cls])
    frames = frames_of(raw.shape[2], shuffle, rng)
    for ii in range(int(raw.shape[2] // batch_size)):
        vol = np.zeros([batch_size, in_size[0], in_size[1], in_size[2]], dtype=np.float32)
        slice_y = np.zeros([batch_size, in_size[0], in_size[1]], dtype=np.float32)
        chunk = frames[ii * batch_size:(ii + 1) * batch_size]
        for idx, jj in enumerate(chunk):
            vol[idx, ...] = raw[..., jj - 1:jj + 2]
            slice_y[idx, ...] = raw_y[..., jj]
        pred, curr = predict(vol, slice_y)
        for idx, jj in enumerate(chunk):
            tmp_y[..., jj] = pred[idx, ...]
        cm += curr
    return tmp_y, cm


def test_eval(predict, label_list, nii_list, batch_size, num_cls, flip_correction=True, shuffle=True, rng=None, on_sample=None):
    """the per-sample loop shared by both trainers -> (sample_eval_list [(dice, jaccard)], summed confusion matrix)"""
    all_cm = np.zeros([num_cls, num_cls])
    sample_eval_list = []
    for idx_file, (label_fid, nii_fid) in enumerate(zip(label_list, nii_list)):
        if not os.path.isfile(nii_fid):
            raise Exception("cannot find sample %s" % str(nii_fid))
        raw = np.asarray(read_nii_image(nii_fid))
        raw_y = np.asarray(read_nii_image(label_fid))
        if flip_correction is True:
            raw = np.flip(np.flip(raw, axis=0), axis=1)
            raw_y = np.flip(np.flip(raw_y, axis=0), axis=1)
        tmp_y, sample_cm = eval_volume(predict, raw, raw_y, batch_size, num_cls, shuffle, rng)
        logging.info("sample %s (%s): %s batches processed" % (idx_file, os.path.basename(nii_fid), raw.shape[2] // batch_size))
        all_cm += sample_cm
        sample_eval_list.append((_dice(sample_cm), _jaccard(sample_cm)))
        if on_sample is not None:
            on_sample(raw_y, tmp_y, nii_fid)
    return sample_eval_list, all_cm


def sample_metric_stddev(sample_eval_list, num_cls, contour_map, quiet=False):
    """adversarial.py:1054-1084 / source_segmenter.py:634-664: per-organ mean and stddev across samples"""
    metric_mat = np.zeros([len(sample_eval_list), num_cls, 2])
    for organ, ind in list(contour_map.items()):
        for ii in range(len(sample_eval_list)):
            metric_mat[ii, int(ind), 0] = sample_eval_list[ii][0][int(ind)]
            metric_mat[ii, int(ind), 1] = sample_eval_list[ii][1][int(ind)]
    if not quiet:
        for what, fn in (("stddev", np.std), ("mean", np.mean)):
            print("------- inside the sample_metric_stddev file ---- ")
            for organ, ind in list(contour_map.items()):
                print("organ: %s" % organ)
                print("dice_%s: %s" % (what, fn(metric_mat[:, int(ind), 0])))
                print("jaccard_%s: %s" % (what, fn(metric_mat[:, int(ind), 1])))
        print("-------")
        print("all_dice_mean: %s" % (np.mean(metric_mat[:, 1:, 0])))
        print("all_jaccard_mean: %s" % (np.mean(metric_mat[:, 1:, 1])))
    subject_level_list = np.mean(metric_mat, axis=0)
    return subject_level_list[:, 0], subject_level_list[:1]

"""Host -> HBM staging of the dequeued mini-batches.

The reference dequeues a batch inside `sess.run`, one-hot encodes the label map on the host (`lib._label_decomp`,
lib.py:75-92) and feeds both arrays through feed_dict — a synchronous pageable H2D copy per step
(source_segmenter.py:478-489).  Here a producer thread keeps `depth` batches in flight:

    SliceQueue.next_batch -> pinned staging slot -> async H2D on a copy stream -> one-hot ON DEVICE -> event

and the training loop's `next()` only makes the compute stream wait on that event, so at >400 slices/s the step never
sees the input path.  On a CPU device (host-logic tests) the same code runs without streams or pinning.
"""
import queue
import threading

import numpy as np
import torch

from .lib import _label_decomp, label_decomp_device


class DeviceFeeder(object):
    """Iterator over (x [B,H,W,3] float32, y one-hot [B,H,W,num_cls] float32, fids) resident on `device`.

    `source` is anything with `next_batch(batch_size) -> (np.ndarray [B,H,W,4], fids)` (tfrecord.SliceQueue, synthetic sources):
    image channels 0:3, integer-valued label map in channel 3 — the pair_feed layout of source_segmenter.py:344-355.
    """

    def __init__(self, source, batch_size, num_cls, device, depth=2):
        self.source, self.batch_size, self.num_cls = source, int(batch_size), int(num_cls)
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        if self.cuda and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.depth = max(int(depth), 1)
        self._q = queue.Queue(maxsize=self.depth)
        self._slots = None
        self._stop = False
        self._error = None
        self._stream = torch.cuda.Stream(self.device) if self.cuda else None
        self._thread = threading.Thread(target=self._produce, daemon=True)
        self._thread.start()

    # -- producer ---------------------------------------------------------------------------------------------
    def _stage(self, i, batch):
        B, H, W, _ = batch.shape
        if self._slots is None or self._slots[0][0].shape[:3] != (B, H, W):
            n = self.depth + 1          # one more than the queue holds: the slot being filled is never one still being copied
            self._slots = [(torch.empty((B, H, W, 3), dtype=torch.float32, pin_memory=self.cuda),
                            torch.empty((B, H, W), dtype=torch.float32, pin_memory=self.cuda), [None]) for _ in range(n)]
        xs, ls, ev = self._slots[i % len(self._slots)]
        if ev[0] is not None:
            ev[0].synchronize()         # the previous H2D out of this slot has finished
        np.copyto(xs.numpy(), batch[:, :, :, 0:3])
        np.copyto(ls.numpy(), batch[:, :, :, 3])
        return xs, ls, ev

    def _produce(self):
        try:
            if self.cuda:
                torch.cuda.set_device(self.device)
            i = 0
            while not self._stop:
                batch, fids = self.source.next_batch(self.batch_size)
                xs, ls, ev = self._stage(i, np.asarray(batch))
                i += 1
                if self.cuda:
                    with torch.cuda.stream(self._stream):
                        xd = xs.to(self.device, non_blocking=True)
                        ld = ls.to(self.device, non_blocking=True)
                        yd = label_decomp_device(self.num_cls, ld)
                        done = torch.cuda.Event()
                        done.record(self._stream)
                    ev[0] = done
                else:
                    # host tensors (CPU plumbing tests only): the reference's own host-side one-hot, lib._label_decomp (lib.py:75-92)
                    xd = xs.clone()
                    yd = torch.from_numpy(_label_decomp(self.num_cls, ls.numpy()))
                    done = None
                item = (xd, yd, fids, done)
                while not self._stop:
                    try:
                        self._q.put(item, timeout=0.05)
                        break
                    except queue.Full:
                        continue
        except BaseException as e:      # surface input failures in the training loop instead of hanging it
            self._error = e

    # -- consumer ---------------------------------------------------------------------------------------------
    def next(self):
        while True:
            try:
                xd, yd, fids, done = self._q.get(timeout=0.05)
                break
            except queue.Empty:
                if self._error is not None:
                    raise IOError("DeviceFeeder: input thread failed: %r" % (self._error,))
                if not self._thread.is_alive():
                    raise IOError("DeviceFeeder: input thread exited")
        if done is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(done)
            xd.record_stream(cur)       # allocated on the copy stream, consumed on the compute stream
            yd.record_stream(cur)
        return xd, yd, fids

    __next__ = next

    def __iter__(self):
        return self

    def close(self):
        self._stop = True
        self._thread.join(timeout=2.0)
        if hasattr(self.source, "close"):
            self.source.close()

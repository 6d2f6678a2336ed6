"""Data parallelism for the training step: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference is single-GPU (train_segmenter.py:20 / train_gan.py:18 pin CUDA_VISIBLE_DEVICES); sharding the mini-batch
over the 8 GPUs of a node is new functionality required by BASELINE.json.  Design:
  * every rank holds a full replica (weights, optimiser state) and B/W slices of each batch; dropout streams are
    decorrelated by rank (seed + rank);
  * gradients live in ONE flat fp32 arena (variables.VariableStore.grad_arena), written in place by autograd;
    the arena is cut into ~32 MiB buckets in REVERSE variable order, and each bucket is all-reduced (sum) on a side HIP
    stream as soon as the last gradient inside it has been produced, overlapping the rest of the backward pass;
  * the 1/W average is folded into the loss-gradient kernel (gscale), so the reduced arena feeds the optimiser directly;
  * BatchNorm uses per-replica statistics (like every DP framework) and the batch-global loss normalisers are per replica too —
    unless `enable_sync_stats()` is on: then the BN batch statistics, the BN backward sums and the loss's class / Dice sums are
    all-reduced (tiny, latency-bound messages: ~2 per BN layer per pass), which makes W ranks x B slices compute exactly the
    single-GPU step on W*B slices.  It exists for that parity test (tests/test_gpu_dp.py); production keeps it off.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment; returns (rank, local_rank, world_size)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # PNP_DIST_BACKEND=gloo lets the overlapped reducer be exercised with several ranks sharing ONE GPU (RCCL refuses
            # duplicate devices); production is always nccl (= RCCL over xGMI)
            backend = os.environ.get("PNP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


_SYNC = None     # (process group, world size) while batch statistics and loss normalisers are synchronised across replicas


def enable_sync_stats(group=None):
    """SyncBN + batch-global loss normalisers (opt-in)"""
    global _SYNC
    w = dist.get_world_size(group) if dist.is_initialized() else 1
    _SYNC = (group, w) if w > 1 else None


def disable_sync_stats():
    global _SYNC
    _SYNC = None


def sync_world():
    return _SYNC[1] if _SYNC else 1


def all_sum_(t):
    """in-place sum over the replicas of the sync group (no-op when synchronisation is off)"""
    if _SYNC:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=_SYNC[0])
    return t


def sync_bn_stats(mean, var):
    """per-replica (mean, biased var) over equally many rows -> the statistics of the concatenated batch.  Combined in float64:
    E[x] = avg(mean_r), E[x^2] = avg(var_r + mean_r^2)."""
    w = sync_world()
    if w <= 1:
        return mean, var
    m = mean.double()
    st = torch.stack([m, var.double() + m * m])
    all_sum_(st)
    st /= w
    gm = st[0]
    gv = (st[1] - gm * gm).clamp_min_(0.0)
    return gm.float(), gv.float()


def barrier():
    if dist.is_initialized():
        dist.barrier()


def rank_seed(rank):
    """offset added to the per-step dropout seed so that replicas draw decorrelated masks ("seed + r")"""
    return (int(rank) * 0x9E3779B1) & 0x3FFFFFFF


class GradReducer(object):
    """Bucketed, backward-overlapped gradient all-reduce over a VariableStore's flat gradient arena."""

    def __init__(self, store, bucket_bytes=32 << 20, overlap=True, group=None):
        self.store = store
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.overlap = overlap and self.world > 1 and store.arena.is_cuda
        self.buckets = []       # (start, end) element ranges of the arena, in reverse variable order
        self._pending = []      # outstanding work handles
        self._remaining = []
        self._members = []      # variables of each bucket
        self._var_bucket = {}
        tr = store.trainable()
        # buckets: walk variables from the LAST created (first gradient to be ready) to the first
        cur_end = None
        cur_start = None
        members = []
        limit = max(int(bucket_bytes) // 4, 1)
        from ._lib import OPT_CHUNK
        for v in reversed(tr):
            seg_start = v.offset
            seg_end = v.offset + -(-v.numel // OPT_CHUNK) * OPT_CHUNK
            if cur_end is None:
                cur_end, cur_start, members = seg_end, seg_start, [v]
            else:
                cur_start = seg_start
                members.append(v)
            if cur_end - cur_start >= limit:
                self._close(cur_start, cur_end, members)
                cur_end = None
        if cur_end is not None:
            self._close(cur_start, cur_end, members)
        self.side = torch.cuda.Stream() if self.overlap else None
        if self.overlap:
            for v in tr:
                v.tensor.register_post_accumulate_grad_hook(self._make_hook(v))
        self.reset()

    def _close(self, start, end, members):
        b = len(self.buckets)
        self.buckets.append((start, end))
        self._members.append(list(members))
        for v in members:
            self._var_bucket[v.name] = b
        self._remaining.append(len(members))

    def reset(self):
        self._count = list(self._remaining)
        self._pending = []
        self._launched = [False] * len(self.buckets)

    def _make_hook(self, v):
        b = self._var_bucket[v.name]

        def hook(_param):
            self._count[b] -= 1
            if self._count[b] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        if self._launched[b]:
            return
        self._launched[b] = True
        s, e = self.buckets[b]
        view = self.store.grad_arena[s:e]
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.side.wait_event(ev)
        with torch.cuda.stream(self.side):
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)

    def allreduce(self, grad_arena=None):
        """call after backward: finishes every outstanding bucket; afterwards the arena holds the summed gradients."""
        if self.world <= 1:
            return
        if self.overlap:
            # buckets whose hook count did not reach zero: some member had no gradient this step.  A bucket none of whose
            # variables is being trained in this step (the GAN steps switch requires_grad per variable group: a dis step only
            # produces the 91 MB of critic gradients, a gen step the 20 MB of adapt_*) is skipped — identically on every rank.
            for b in range(len(self.buckets)):
                if not self._launched[b] and any(v.tensor.requires_grad for v in self._members[b]):
                    self._launch(b)
            torch.cuda.current_stream().wait_stream(self.side)
            self.reset()
        else:
            for b, (s, e) in enumerate(self.buckets):
                if any(v.tensor.requires_grad for v in self._members[b]):
                    dist.all_reduce(self.store.grad_arena[s:e], op=dist.ReduceOp.SUM, group=self.group)

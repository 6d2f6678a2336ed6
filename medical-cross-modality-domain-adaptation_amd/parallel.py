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
import ctypes
import os

import torch
import torch.distributed as dist

from . import _lib


class NativeComm(object):
    """One RCCL communicator per process behind the C-ABI (pnp_comm_*, csrc/comm.hip): the gradient / statistics all-reduces are
    enqueued by libpnp_hip.so directly on the HIP stream the caller names — no torch.distributed on the data path.  The 128-byte
    RCCL unique id travels from rank 0 over the torchrun rendezvous (a gloo broadcast of a CPU tensor: control plane only).

    Bring-up is in STAGES so that the ranks can agree after each one (_bring_up_native): `prepare()` is purely local (dlopen of
    librccl.so, rank 0 draws the unique id), `connect()` holds the two collectives (the id broadcast and ncclCommInitRank)."""

    def __init__(self, rank, world, group=None):
        self.rank, self.world, self.group = rank, world, group
        self.handle = ctypes.c_void_p()
        self.ident = torch.zeros(_lib.COMM_ID_BYTES, dtype=torch.uint8)
        self.version = 0
        self.bytes_reduced = 0          # payload handed to pnp_comm_allreduce since the last take_bytes() (bench.py's comm record)

    def prepare(self):
        """local part: bind librccl.so, rank 0 draws the unique id.  May raise; no other rank is involved."""
        lib = _lib.load()
        # bind the librccl.so that ships inside the PyTorch wheel: it links the HIP runtime this process already uses (see _lib.py)
        cand = os.environ.get("PNP_RCCL_LIB") or os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        _lib.check(lib.pnp_comm_load(cand.encode() if os.path.exists(cand) else None), "pnp_comm_load")
        if self.rank == 0:
            _lib.check(lib.pnp_comm_unique_id(ctypes.c_void_p(self.ident.data_ptr())), "pnp_comm_unique_id")
        v = ctypes.c_int()
        _lib.check(lib.pnp_comm_version(ctypes.byref(v)), "pnp_comm_version")
        self.version = v.value
        return self

    def connect(self):
        """collective part: EVERY rank of the group must call it (after all of them prepared successfully)"""
        if self.world > 1:
            dist.broadcast(self.ident, src=0, group=self.group)
        _lib.check(_lib.load().pnp_comm_init(self.rank, self.world, ctypes.c_void_p(self.ident.data_ptr()), ctypes.byref(self.handle)),
                   "pnp_comm_init")
        return self

    def allreduce_(self, t, stream=None):
        """in-place sum of a contiguous float32 / float64 device tensor, enqueued on `stream` (default: the current stream)"""
        if not (t.is_cuda and t.is_contiguous()):
            raise _lib.PnpError("pnp_comm_allreduce needs a contiguous device tensor")
        dt = {torch.float32: _lib.DTYPE_F32, torch.float64: _lib.DTYPE_F64}.get(t.dtype)
        if dt is None:
            raise _lib.PnpError("pnp_comm_allreduce: unsupported dtype %s" % t.dtype)
        st = stream if stream is not None else torch.cuda.current_stream()
        _lib.check(_lib.load().pnp_comm_allreduce(self.handle, ctypes.c_void_p(t.data_ptr()), t.numel(), dt, ctypes.c_void_p(st.cuda_stream)),
                   "pnp_comm_allreduce")
        self.bytes_reduced += t.numel() * t.element_size()
        return t

    def take_bytes(self):
        n, self.bytes_reduced = self.bytes_reduced, 0
        return n

    def destroy(self):
        if self.handle:
            torch.cuda.synchronize()
            _lib.check(_lib.load().pnp_comm_destroy(self.handle), "pnp_comm_destroy")
            self.handle = ctypes.c_void_p()


_COMM = None     # NativeComm while the native RCCL transport is up


def native_comm():
    return _COMM


def transport():
    """what carries the data-path collectives of this process"""
    if _COMM is not None:
        return "rccl-native %d (pnp_comm_*, control plane: torch.distributed/%s)" % (_COMM.version, dist.get_backend() if dist.is_initialized() else "-")
    if dist.is_initialized():
        return "torch.distributed/%s" % dist.get_backend()
    return None


def init_distributed(backend=None):
    """Initialise the process group from the torchrun environment; returns (rank, local_rank, world_size).

    On GPUs the DATA path is native RCCL through the C-ABI (NativeComm); torch.distributed (gloo) only carries the rendezvous, the
    unique-id broadcast and host-side barriers.  PNP_COMM=torch selects torch.distributed's own nccl(=RCCL) backend instead;
    PNP_DIST_BACKEND=gloo (tests) lets several ranks share ONE GPU, which RCCL refuses, with gloo on the data path too."""
    global _COMM
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        forced = backend or os.environ.get("PNP_DIST_BACKEND")
        want_native = (forced is None and torch.cuda.is_available() and os.environ.get("PNP_COMM", "rccl") != "torch")
        if want_native:
            backend = "gloo"
        elif forced is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        else:
            backend = forced
        if backend == "nccl" or want_native:      # one GPU per rank (gloo test modes put several ranks on one device)
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        if want_native:
            _bring_up_native(rank, world)
        control_group()          # (a collective when it has to create the gloo side group: every rank is here)
    return rank, local, world


_DATA_GROUP = None      # torch.distributed group that carries the data-path collectives when the native communicator is not up


def _all_ok(flag):
    ok = torch.tensor([1 if flag else 0], dtype=torch.int32)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    return int(ok.item()) == 1


def _bring_up_native(rank, world):
    """NativeComm on every rank or on none, agreed over the gloo control plane STAGE BY STAGE — a rank that fails must never leave
    the others inside a collective it does not join:
      1. every rank runs the local part (`prepare`: dlopen + symbols of librccl.so, rank 0 draws the unique id) under try/except;
      2. all ranks all_reduce(MIN) the outcome.  Any failure: nobody enters the id broadcast or ncclCommInitRank; the data path runs on
         torch.distributed's own nccl (= RCCL) group instead, loudly;
      3. only if all succeeded, every rank runs the collective part (`connect`) unconditionally, then the ranks agree once more.
    A failure INSIDE ncclCommInitRank on a subset of the ranks cannot be recovered from (the others are blocked inside RCCL's own
    bootstrap, which no host-side agreement can interrupt): the job dies on RCCL's / the launcher's timeout.  What stage 3's agreement
    buys is the symmetric case (every rank returns an error, e.g. an unusable id): all fall back together."""
    global _COMM, _DATA_GROUP
    import logging
    log = logging.getLogger(__name__)
    comm, err = NativeComm(rank, world), None
    try:
        comm.prepare()
    except Exception as e:          # noqa: BLE001 — whatever went wrong, the other ranks must learn of it before anyone proceeds
        err = e
    if _all_ok(err is None):
        try:
            comm.connect()
        except Exception as e:      # noqa: BLE001
            err = e
        if _all_ok(err is None):
            _COMM = comm
            return
        if err is None:
            comm.destroy()
    log.warning("native RCCL communicator (pnp_comm_*) unavailable on at least one rank (this rank: %s): gradient all-reduce falls "
                "back to torch.distributed's nccl (RCCL) group", err if err is not None else "ok")
    _COMM = None
    _DATA_GROUP = dist.new_group(backend="nccl")


def data_group():
    return _DATA_GROUP


_CTRL_GROUP = None


def control_group():
    """group for HOST-side agreement between the ranks (tiny CPU tensors): the default group when it is gloo, a gloo side group when
    the default group is torch.distributed's nccl (PNP_COMM=torch)"""
    global _CTRL_GROUP
    if dist.get_backend() != "nccl":
        return None
    if _CTRL_GROUP is None:
        _CTRL_GROUP = dist.new_group(backend="gloo")
    return _CTRL_GROUP


def shutdown():
    global _COMM, _DATA_GROUP, _CTRL_GROUP
    _DATA_GROUP = _CTRL_GROUP = None
    if _COMM is not None:
        _COMM.destroy()
        _COMM = None
    if dist.is_initialized():
        dist.destroy_process_group()


def all_max_scalar(value, device=None):
    """max over the ranks of a host scalar (bench timing); host tensors over gloo, a device tensor when the group is nccl"""
    if not dist.is_initialized():
        return float(value)
    on_dev = dist.get_backend() == "nccl"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if on_dev else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


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
        if _COMM is not None and _SYNC[0] is None and t.is_cuda:
            _COMM.allreduce_(t)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=_SYNC[0] if (_SYNC[0] is not None or not t.is_cuda) else _DATA_GROUP)
    return t


def sync_bn_stats(mean, var):
    """per-replica (mean, biased var) over equally many rows -> the statistics of the concatenated batch: raw moments in float64
    (pnp_bn_moments), summed over the replicas, converted back (pnp_bn_from_moments) — no torch arithmetic"""
    w = sync_world()
    if w <= 1:
        return mean, var
    if not mean.is_cuda:         # host tensors: only the gloo / CPU tests of the collective logic come here (no kernel can run on them)
        m = mean.double()
        mom = torch.cat([m, var.double() + m * m])
        all_sum_(mom)
        gm = mom[:m.numel()] / w
        return gm.float(), (mom[m.numel():] / w - gm * gm).clamp_min_(0.0).float()
    from . import kernels as K
    mom = K.bn_moments(mean, var)
    all_sum_(mom)
    return K.bn_from_moments(mom, w)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def rank_seed(rank):
    """offset added to the per-step dropout seed so that replicas draw decorrelated masks ("seed + r")"""
    return (int(rank) * 0x9E3779B1) & 0x3FFFFFFF


class GradReducer(object):
    """Bucketed, backward-overlapped gradient all-reduce over a VariableStore's flat gradient arena.

    Deadlock-proof by construction: every rank must enqueue the SAME collectives in the SAME order.
      * ORDER.  Buckets are launched strictly in index order (bucket 0 = the last-created variables, whose gradients are ready first);
        a bucket whose gradients are complete early waits for its predecessors.  The order in which autograd hooks fire therefore
        cannot change the collective sequence.
      * SET.  Which buckets a step reduces is a function of the `requires_grad` flags alone (the GAN steps switch them per variable
        group: a dis step reduces the 91 MB of critic gradients, a gen step the 20 MB of adapt_*), never of which hooks happened to
        fire.  Every step, BEFORE its first collective is enqueued, the ranks compare a checksum of the set over the host control plane
        (one 16-byte gloo all-reduce, ~0.1 ms against a >= 35 ms step); a rank whose set differs makes EVERY rank raise instead of
        hanging in RCCL.  (Checking a set only the first time it is seen would not be symmetric: the rank that diverges sees a NEW set
        while the others see a known one and go straight to the collective.)"""

    def __init__(self, store, bucket_bytes=32 << 20, overlap=True, group=None):
        self.store = store
        if group is None and _COMM is None and store.arena.is_cuda:
            group = _DATA_GROUP          # native communicator not up: torch.distributed's nccl group (None: the default group)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.overlap = overlap and self.world > 1 and store.arena.is_cuda
        self.buckets = []       # (start, end) element ranges of the arena, in reverse variable order
        self._remaining = []
        self._members = []      # variables of each bucket
        self._var_bucket = {}
        self.sets_seen = set()  # distinct bucket sets reduced so far (bench.py's comm record)
        self.launch_log = []    # bucket indices in launch order, this step (tests; bench.py's comm record)
        self.bytes_step = 0     # payload all-reduced by the last finished step
        self.exposed_ms = None  # with measure_exposed: time the compute stream waited for the side stream in the last step
        self.measure_exposed = False
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
        self.native = _COMM if (group is None and store.arena.is_cuda) else None      # data path: pnp_comm_allreduce on OUR stream
        self.side = torch.cuda.Stream() if self.overlap else None
        if self.overlap:
            from . import gradsink
            for v in tr:
                hook = self._make_hook(v)
                v.tensor.register_post_accumulate_grad_hook(hook)          # gradients that come through the autograd engine
                gradsink.set_ready(v.tensor, (lambda h=hook, t=v.tensor: h(t)))   # gradients the kernels add straight into the arena
        self.reset()

    def _close(self, start, end, members):
        b = len(self.buckets)
        self.buckets.append((start, end))
        self._members.append(list(members))
        for v in members:
            self._var_bucket[v.name] = b
        self._remaining.append(len(members))

    def _arm_counts(self):
        """per-bucket number of gradients this step will deliver: the members that take a gradient NOW (the GAN steps toggle
        requires_grad per variable group between steps, so a bucket that mixes critic and adapt_* variables is complete when its
        trainable members have reported — counting every member left such a bucket waiting for allreduce() and, at the head of the
        active set, held every complete bucket behind it: no overlap)"""
        self._count = [sum(1 for v in m if v.tensor.requires_grad) for m in self._members]
        self._armed = True

    def reset(self):
        self._count = list(self._remaining)
        self._armed = False     # counts are re-derived from the requires_grad flags at the first gradient of the next step
        self._launched = [False] * len(self.buckets)
        self._ready = [False] * len(self.buckets)
        self._seen = set()
        self._active = None     # this step's bucket set (tuple of indices), fixed when the first bucket becomes ready
        self._next = 0          # position in _active of the next bucket to launch
        self.launch_log = []

    # ---- which buckets this step reduces: requires_grad flags only -> identical on every rank of a correct program --------------
    def active_set(self):
        return tuple(b for b in range(len(self.buckets)) if any(v.tensor.requires_grad for v in self._members[b]))

    def _begin_step(self):
        if self._active is not None:
            return
        self._active = self.active_set()
        self._next = 0
        self.launch_log = []            # (the previous step's log stayed readable until now)
        self.sets_seen.add(self._active)
        if dist.is_initialized() and self.world > 1:
            import zlib
            h = zlib.crc32(repr((len(self.buckets), self._active)).encode())
            t = torch.tensor([h, -h], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=control_group())     # host control plane; nothing of this step is enqueued yet
            if int(t[0]) != h or int(t[1]) != -h:
                raise RuntimeError("GradReducer: the ranks disagree on which gradient buckets this step reduces (rank %d: %s) — "
                                   "requires_grad flags / graphs differ between ranks; refusing to enqueue mismatched collectives"
                                   % (dist.get_rank(), (len(self.buckets), self._active)))

    def _make_hook(self, v):
        b = self._var_bucket[v.name]
        name = v.name

        def hook(_param):
            # once per variable and step: a gradient the kernels added straight into the arena announces itself through its sink, and
            # the engine's AccumulateGrad node still runs its post-accumulate hook for the (undefined) gradient it was handed — both
            # come after the variable's last use of the step, the first one counts
            if name in self._seen:
                return
            if not self._armed:
                self._arm_counts()
            self._seen.add(name)
            self._count[b] -= 1
            if self._count[b] == 0:
                self._mark_ready(b)
        return hook

    def _mark_ready(self, b):
        """bucket b is complete: launch it and every complete successor — but never ahead of an incomplete predecessor"""
        self._begin_step()
        self._ready[b] = True
        while self._next < len(self._active) and self._ready[self._active[self._next]]:
            self._launch(self._active[self._next])
            self._next += 1

    def _launch(self, b):
        if self._launched[b]:
            return
        self._launched[b] = True
        self.launch_log.append(b)
        s, e = self.buckets[b]
        view = self.store.grad_arena[s:e]
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.side.wait_event(ev)
        # filter gradients run on functional's own side stream (wgrad_overlap): their sinks announce them as soon as the kernel is QUEUED
        # there, so the all-reduce is fenced behind that stream's tail as well
        from .functional import wgrad_side_stream
        ws = wgrad_side_stream()
        if ws is not None:
            ev2 = torch.cuda.Event()
            ev2.record(ws)
            self.side.wait_event(ev2)
        if self.native is not None:
            self.native.allreduce_(view, self.side)       # RCCL's kernel is enqueued on the side stream itself: the event above is the fence
        else:
            with torch.cuda.stream(self.side):
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)

    def allreduce(self, grad_arena=None):
        """call after backward: finishes every outstanding bucket; afterwards the arena holds the summed gradients."""
        if self.world <= 1:
            return
        if self.overlap:
            # buckets whose hook count did not reach zero (some member had no gradient this step) go out now, still in index order
            self._begin_step()
            for b in self._active[self._next:]:
                self._launch(b)
            self._next = len(self._active)
            cur = torch.cuda.current_stream()
            if self.measure_exposed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                cur.wait_stream(self.side)
                e1.record(cur)
                self._exposed_events = (e0, e1)
            else:
                cur.wait_stream(self.side)
            self.bytes_step = sum((self.buckets[b][1] - self.buckets[b][0]) * 4 for b in self.launch_log)
            log = list(self.launch_log)
            self.reset()
            self.launch_log = log              # stays readable until the next step's first bucket
        else:
            act = self.active_set()
            self.launch_log = list(act)
            for b in act:
                s, e = self.buckets[b]
                if self.native is not None:
                    self.native.allreduce_(self.store.grad_arena[s:e])
                else:
                    dist.all_reduce(self.store.grad_arena[s:e], op=dist.ReduceOp.SUM, group=self.group)
            self.bytes_step = sum((self.buckets[b][1] - self.buckets[b][0]) * 4 for b in act)

    def exposed_time_ms(self):
        """with measure_exposed: how long the compute stream sat behind the side stream at the end of the last step (synchronises)"""
        ev = getattr(self, "_exposed_events", None)
        if ev is None:
            return None
        ev[1].synchronize()
        return ev[0].elapsed_time(ev[1])

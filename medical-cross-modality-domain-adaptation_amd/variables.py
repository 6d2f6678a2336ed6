"""Variable store: the eager stand-in for the TF-1.4 default graph's variable collections.

The reference creates variables while it builds the graph (`layers.weight_variable`, layers.py:47-55;
`tf.contrib.layers.batch_norm` variables, layers.py:100) and later groups them BY NAME
(adversarial.py:478-501, 654).  Names therefore are part of the API: this store reproduces TF's
naming (`group_1/Variable_2`, `BatchNorm_7/gamma`, `cls_scope/cls_1/...`) and keeps all float32 state in
flat arenas so that the optimiser, the weight clip and the gradient all-reduce each run as ONE launch /
ONE collective over contiguous memory (PNP_OPT_CHUNK-aligned segments).
"""
import contextlib
from collections import OrderedDict

import numpy as np
import torch

from . import gradsink
from ._lib import OPT_CHUNK

_current = None


def current_store():
    if _current is None:
        raise RuntimeError("no active VariableStore: wrap graph-building code in `with store.as_default():`")
    return _current


class Var(object):
    __slots__ = ("name", "shape", "trainable", "kind", "tensor", "offset", "numel", "l2_mult")

    def __init__(self, name, shape, trainable, kind):
        self.name, self.shape, self.trainable, self.kind = name, tuple(shape), trainable, kind
        self.tensor = None
        self.offset = -1
        self.numel = int(np.prod(shape)) if len(shape) else 1
        self.l2_mult = 0.0


def truncated_normal(rng, shape, stddev):
    """tf.truncated_normal: N(0, stddev) with samples beyond 2 sigma re-drawn (layers.py:48)."""
    out = rng.standard_normal(size=shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * stddev).astype(np.float32)


class VariableStore(object):
    def __init__(self, device, seed=0):
        self.device = torch.device(device)
        self.vars = OrderedDict()
        self.rng = np.random.default_rng(seed)
        self._nscope = []         # tf.name_scope stack: prefixes tf.Variable names (weight_variable)
        self._vscope = []         # tf.variable_scope stack: prefixes get_variable / batch_norm names
        self._uniq = {}           # per-trace unique-name counters
        self.finalized = False
        self.arena = None         # flat fp32: all trainable values, chunk aligned
        self.grad_arena = None
        self.state_arena = None   # non-trainable float state (BN moving stats, frozen weights)
        # dropout stream: seed changes per step, stream id per conv call site within a step
        self.drop_seed = 0
        self._drop_stream = 0

    # ---- scoping / naming (TF semantics) -----------------------------------------------------
    @contextlib.contextmanager
    def as_default(self):
        global _current
        prev = _current
        _current = self
        try:
            yield self
        finally:
            _current = prev

    @contextlib.contextmanager
    def name_scope(self, name):
        """tf.name_scope: affects tf.Variable (weight_variable) names only (source_segmenter.py:91...)."""
        self._nscope.append(name)
        try:
            yield
        finally:
            self._nscope.pop()

    @contextlib.contextmanager
    def variable_scope(self, name):
        """tf.variable_scope: affects get_variable / batch_norm names AND opens a name scope (adversarial.py:130...)."""
        self._vscope.append(name)
        self._nscope.append(name)
        try:
            yield
        finally:
            self._nscope.pop()
            self._vscope.pop()

    def begin_trace(self, drop_seed=None):
        """call at the start of every forward pass: unique-name counters restart so the k-th anonymous
        variable of a scope resolves to the same variable as in the build pass (graph re-use)."""
        self._uniq = {}
        self._drop_stream = 0
        if drop_seed is not None:
            self.drop_seed = int(drop_seed)

    def next_drop_stream(self):
        s = self._drop_stream
        self._drop_stream += 1
        return s

    def _prefix(self, var_scope):
        return "/".join(s for s in (self._vscope if var_scope else self._nscope) if s)

    def unique(self, base, var_scope=False):
        """TF unique_name: base, base_1, base_2, ... under the current name scope (or variable scope)."""
        pre = self._prefix(var_scope)
        key = (pre, base, var_scope)
        k = self._uniq.get(key, 0)
        self._uniq[key] = k + 1
        leaf = base if k == 0 else "%s_%d" % (base, k)
        return (pre + "/" + leaf) if pre else leaf

    def scoped(self, leaf):
        """name of a get_variable-style variable in the current variable scope"""
        pre = self._prefix(True)
        return (pre + "/" + leaf) if pre else leaf

    # ---- creation / lookup ------------------------------------------------------------------------
    def get(self, name, shape=None, init=None, trainable=True, kind="weight"):
        v = self.vars.get(name)
        if v is not None:
            if shape is not None and tuple(shape) != v.shape:
                raise ValueError("variable %s exists with shape %s, requested %s" % (name, v.shape, tuple(shape)))
            return v
        if self.finalized:
            raise RuntimeError("variable %s requested after the store was finalized" % name)
        if shape is None:
            raise KeyError(name)
        v = Var(name, shape, trainable, kind)
        if init is None:
            host = np.zeros(shape, dtype=np.float32)
        elif callable(init):
            host = np.asarray(init(self.rng, tuple(shape)), dtype=np.float32)
        else:
            host = np.full(shape, float(init), dtype=np.float32)
        v.tensor = torch.from_numpy(np.ascontiguousarray(host)).to(self.device)
        if self.device.type == "cuda" and kind == "weight":
            from . import kernels as K
            K.weights_changed()      # a new filter may land on the address of a freed one whose bf16 shadow is still cached
        self.vars[name] = v
        return v

    def finalize(self):
        """Pack every variable into chunk-aligned flat arenas (trainable -> arena(+grad), others -> state arena).

        Every trainable variable registers a GRADIENT SINK (gradsink.py): the backward kernels ADD its gradient straight into its slot
        of `grad_arena` and hand None to the autograd engine.  Consequences for code outside the step functions of this package:
        `torch.autograd.grad(loss, params)`, tensor hooks on a weight and a second backward over a retained graph see None / add into
        the arena as a side effect — wrap such code in `with gradsink.disabled():`; and `zero_grad()` must run before the forward pass
        of a step (it re-arms the sinks' use counters)."""
        if self.finalized:
            return
        tr = [v for v in self.vars.values() if v.trainable]
        st = [v for v in self.vars.values() if not v.trainable]

        def pack(vs):
            off = 0
            for v in vs:
                v.offset = off
                off += -(-v.numel // OPT_CHUNK) * OPT_CHUNK
            flat = torch.zeros(max(off, OPT_CHUNK), dtype=torch.float32, device=self.device)
            for v in vs:
                flat[v.offset:v.offset + v.numel].copy_(v.tensor.reshape(-1))
                v.tensor = flat[v.offset:v.offset + v.numel].view(v.shape)
            return flat

        self.arena = pack(tr)
        self.state_arena = pack(st)
        from . import functional as F_
        F_.forget_stat_versions(self.state_arena.data_ptr(), self.state_arena.data_ptr() + self.state_arena.numel() * 4)
        if self.device.type == "cuda":
            from . import kernels as K
            K.weights_changed()      # every filter moved: bf16 shadows cached under the old addresses must not be hit by a later allocation
        self.grad_arena = torch.zeros_like(self.arena)
        for v in tr:
            t = v.tensor.detach()
            t.requires_grad_(True)
            t.grad = self.grad_arena[v.offset:v.offset + v.numel].view(v.shape)
            v.tensor = t
            gradsink.register(t)        # kernels may add this variable's gradient straight into its slot
        for v in self.vars.values():    # store-owned filters: the Winograd route may keep their transformed copies (kernels._wino_u)
            if v.tensor.dim() == 4 and v.tensor.is_cuda:
                v.tensor._pnp_var = True
        self.finalized = True

    # ---- helpers for the optimisers ---------------------------------------------------------------
    def trainable(self):
        return [v for v in self.vars.values() if v.trainable]

    def written_ranges(self, mask_host=None):
        """byte-address ranges [(lo, hi), ...] of the trainable arena an optimiser with this per-chunk mask writes (None: all of it)"""
        base = self.arena.data_ptr()
        if mask_host is None:
            return [(base, base + 4 * self.arena.numel())]
        out, start = [], None
        for i, m in enumerate(list(mask_host) + [0]):
            if m and start is None:
                start = i
            elif not m and start is not None:
                out.append((base + 4 * start * OPT_CHUNK, base + 4 * i * OPT_CHUNK))
                start = None
        return out

    def chunk_table(self, fn, dtype):
        """per-chunk table over the trainable arena: value fn(var) for every chunk the variable covers"""
        n = self.arena.numel() // OPT_CHUNK
        tab = np.zeros(n, dtype=dtype)
        for v in self.trainable():
            c0 = v.offset // OPT_CHUNK
            c1 = c0 + -(-v.numel // OPT_CHUNK)
            tab[c0:c1] = fn(v)
        return torch.from_numpy(tab).to(self.device)

    # ---- optimiser slots, keyed by variable name like tf.train.Saver keys them ('<variable>/<slot>') ---------------------------------
    def slots_to_dict(self, slot_arena, slot, select=None):
        """{'<variable name>|<slot>': values} for every trainable variable (optionally only those `select(var)` accepts).  The arena
        LAYOUT depends on which variables are trainable (it differs between the pre-train and train-gan graphs), names do not."""
        out = OrderedDict()
        for v in self.trainable():
            if select is None or select(v):
                out["%s|%s" % (v.name.replace("/", "|"), slot)] = slot_arena[v.offset:v.offset + v.numel].detach().cpu().numpy().reshape(v.shape)
        return out

    def slots_from_dict(self, slot_arena, sd, slot, select=None):
        """inverse of slots_to_dict; name-matched and relaxed like tf.train.Saver.restore over a variable subset.
        Returns (restored names, missing names)."""
        done, missing = [], []
        for v in self.trainable():
            if select is not None and not select(v):
                continue
            key = "%s|%s" % (v.name.replace("/", "|"), slot)
            if key in sd and tuple(np.shape(sd[key])) == v.shape:
                slot_arena[v.offset:v.offset + v.numel].copy_(torch.from_numpy(np.asarray(sd[key], dtype=np.float32)).reshape(-1))
                done.append(v.name)
            else:
                missing.append(v.name)
        return done, missing

    def zero_grad(self):
        if self.grad_arena.is_cuda:
            from . import kernels as K
            K.fill_(self.grad_arena, 0.0)          # pnp_fill, like every other value of a step
        else:
            self.grad_arena.zero_()                # host-side stores (graph / naming / data-parallel logic tests)
        gradsink.rearm(v.tensor for v in self.trainable())

    def unit_grad(self, n=1):
        """the root gradient handed to backward(): ones, filled once by pnp_fill (torch's implicit root gradient is a fill launch per
        backward pass; the loss Functions take the upstream gradient to be 1 and fold the data-parallel 1/W into their kernels)"""
        cache = self.__dict__.setdefault("_unit_grads", {})
        t = cache.get(n)
        if t is None:
            if self.device.type == "cuda":
                from . import kernels as K
                t = K.filled((n,), 1.0, self.device)
            else:
                t = torch.ones(n)
            cache[n] = t
        return t

    def state_dict(self):
        return OrderedDict((k, v.tensor.detach().cpu().numpy().copy()) for k, v in self.vars.items())

    def load_state_dict(self, sd, strict=True):
        if self.device.type == "cuda":
            from . import kernels as K
            K.weights_changed()                # the filters' bf16 shadows (bf16-resident convolutions) are stale after a load
        for k, arr in sd.items():
            if k not in self.vars:
                if strict:
                    raise KeyError("unexpected variable %s" % k)
                continue
            v = self.vars[k]
            with torch.no_grad():
                v.tensor.copy_(torch.from_numpy(np.asarray(arr, dtype=np.float32)).reshape(v.shape))
        if strict:
            missing = [k for k in self.vars if k not in sd]
            if missing:
                raise KeyError("missing variables: %s" % missing[:5])

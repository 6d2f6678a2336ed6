"""not gpu: host-side logic — TF padding arithmetic KATs, dropout mask stream, oracle self-checks (closed forms and float64
finite differences), optimiser restatements, tfrecord round trip, 2-rank gloo gradient all-reduce."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, pkg
from oracle import tf_ops as T


def test_same_padding_kats():
    K = pkg("kernels")
    # k3 s2 in256 -> (0,1); k5 s2 in128 -> (1,2); k5 s4 in16 -> (0,1); atrous rate 2 k3 -> (2,2)
    assert K.same_pad(256, 3, 2) == (128, 0, 1) == T.same_pad(256, 3, 2)
    assert K.same_pad(128, 5, 2) == (64, 1, 2) == T.same_pad(128, 5, 2)
    assert K.same_pad(16, 5, 4) == (4, 0, 1) == T.same_pad(16, 5, 4)
    assert K.same_pad(32, 3, 1, 2) == (32, 2, 2)
    g = K.conv_geom((2, 4, 4, 512), (3, 3, 512, 512), 2, 1, "SYMMETRIC")     # cls_6: 4x4 -> 2x2
    assert (g.OH, g.OW) == (2, 2)
    assert T.sym_index(3, 1) == [0, 0, 1, 2, 2] and T.sym_index(3, 2) == [1, 0, 0, 1, 2, 2, 1]


def test_conv_oracle_vs_definition():
    rng = np.random.default_rng(0)
    for (N, H, W, C, Kc, R, s, d, p) in [(2, 9, 7, 5, 6, 3, 1, 1, "SAME"), (1, 16, 16, 4, 3, 3, 2, 1, "SAME"), (1, 16, 16, 4, 3, 5, 2, 1, "SAME"),
                                         (1, 16, 16, 4, 3, 5, 4, 1, "SAME"), (1, 8, 8, 3, 4, 3, 1, 2, "SAME"), (1, 6, 6, 3, 4, 5, 1, 1, "SYMMETRIC"),
                                         (1, 4, 4, 3, 4, 3, 2, 1, "SYMMETRIC")]:
        x = rng.standard_normal((N, H, W, C))
        w = rng.standard_normal((R, R, C, Kc))
        a = T.conv2d(torch.from_numpy(x), torch.from_numpy(w), s, d, p).numpy()
        b = T.conv2d_direct_np(x, w, s, d, p)
        assert a.shape == b.shape and np.abs(a - b).max() < 1e-12


def test_dropout_mask_stream():
    m = T.dropout_mask((1000, 257), 0.75, 5, 2)
    assert abs(m.mean() - 0.75) < 5e-3
    assert not np.array_equal(m, T.dropout_mask((1000, 257), 0.75, 5, 3))      # stream id decorrelates call sites
    assert not np.array_equal(m, T.dropout_mask((1000, 257), 0.75, 6, 2))      # seed decorrelates steps
    assert np.array_equal(m, T.dropout_mask((1000, 257), 0.75, 5, 2))          # reproducible
    assert T.dropout_mask((10,), 1.0, 0, 0).all()
    # hash KATs shared with csrc/pnp_common.h (fmix32 is the murmur3 finaliser)
    assert int(T._fmix32(np.array([1], np.uint32))[0]) == 0x514E28B7
    assert int(T.drop_thresh(0.75)) == 4194304


def test_bn_closed_forms():
    x = torch.full((2, 4, 4, 3), 5.0)
    mm, mv = torch.zeros(3), torch.ones(3)
    y = T.batch_norm(x, torch.ones(3), torch.zeros(3), mm, mv, True)
    assert torch.allclose(y, torch.zeros_like(y))                       # constant input -> zero output
    assert torch.allclose(mm, torch.full((3,), 0.5)) and torch.allclose(mv, torch.full((3,), 0.9))   # m -= (m-batch)*(1-.9)
    x = torch.arange(32, dtype=torch.float64).reshape(2, 4, 4, 1)
    mm, mv = torch.zeros(1, dtype=torch.float64), torch.ones(1, dtype=torch.float64)
    T.batch_norm(x, torch.ones(1, dtype=torch.float64), torch.zeros(1, dtype=torch.float64), mm, mv, True)
    assert abs(float(mv) - (0.9 + 0.1 * float(x.var(unbiased=True)))) < 1e-12       # Bessel-corrected moving variance


def test_loss_closed_forms_and_finite_differences():
    rng = np.random.default_rng(0)
    lab = rng.integers(0, 5, size=(2, 6, 6))
    y = torch.from_numpy(T.label_decomp(5, lab)).double()
    big = (y * 60.0 - 30.0)                                              # confident and correct
    assert abs(float(T.dice_loss(big, y)) + 1.0) < 1e-6                  # Dice of identical masks -> -1
    assert abs(float(T.softmax_weighted_loss(big, y))) < 1e-6
    de, arr = T.dice_eval(torch.from_numpy(lab), y, 5)
    assert abs(float(de) - 1.0) < 1e-6
    z = torch.from_numpy(rng.standard_normal((2, 6, 6, 5))).double().requires_grad_(True)
    f = lambda t: T.softmax_weighted_loss(t, y) + T.dice_loss(t, y)
    f(z).backward()
    g = z.grad.clone()
    eps = 1e-6
    for idx in [(0, 0, 0, 0), (1, 3, 2, 4), (0, 5, 5, 2)]:
        zp, zm = z.detach().clone(), z.detach().clone()
        zp[idx] += eps
        zm[idx] -= eps
        fd = (float(f(zp)) - float(f(zm))) / (2 * eps)
        assert abs(fd - float(g[idx])) < 1e-7


def test_whole_network_fd_gradient_fp64():
    """float64 finite-difference check of the oracle's segmenter backward (tiny random direction on two variables)"""
    from oracle import nets
    rng = np.random.default_rng(1)
    state = {}
    for k, s in nets.segmenter_variable_shapes().items():
        if "Variable" in k:
            state[k] = rng.standard_normal(s) * np.sqrt(2.0 / (s[0] * s[1] * s[2]))
        elif k.endswith("gamma") or k.endswith("moving_variance"):
            state[k] = np.ones(s)
        else:
            state[k] = np.zeros(s)
    state["output/Variable"] *= 0.05
    V = nets.make_variables(state, dtype=torch.float64)
    x = torch.from_numpy(rng.standard_normal((2, 256, 256, 3)))
    lab = np.zeros((2, 256, 256), np.int64)
    lab[:, 60:120, 50:200] = 1
    lab[:, 130:200, 30:90] = 2
    lab[:, 150:220, 120:220] = 3
    lab[0, 10:40, 10:60] = 4
    y = torch.from_numpy(T.label_decomp(5, lab)).double()

    def loss():
        lg = nets.segmenter_forward(V, x, 1.0, True, True)
        c, r, _, _ = nets.segmenter_cost(V, lg, y)
        return c + r
    L = loss()
    L.backward()
    for name in ("group_9/Variable_1", "BatchNorm_29/gamma"):
        d = torch.from_numpy(rng.standard_normal(tuple(V[name].shape)))
        d = d / d.norm()
        ana = float((V[name].grad * d).sum())
        eps = 1e-4          # small enough that few leaky-ReLU / max-pool kinks are crossed
        with torch.no_grad():
            V[name] += eps * d
            lp = float(loss())
            V[name] -= 2 * eps * d
            lm = float(loss())
            V[name] += eps * d
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - ana) < 5e-3 * abs(ana) + 1e-9, (name, fd, ana)


def test_optimizer_restatements():
    w = torch.tensor([1.0, -2.0], dtype=torch.float64)
    g = torch.tensor([0.5, -0.25], dtype=torch.float64)
    m, v = torch.zeros(2, dtype=torch.float64), torch.zeros(2, dtype=torch.float64)
    T.adam_update(w, g, m, v, 1e-3, 1)
    # step 1 of Adam moves every weight by lr*g/(|g| + eps*sqrt(1-b2)) ~ lr*sign(g)
    assert torch.allclose(w, torch.tensor([1.0 - 1e-3, -2.0 + 1e-3], dtype=torch.float64), atol=1e-9)
    w = torch.tensor([1.0], dtype=torch.float64)
    ms = torch.ones(1, dtype=torch.float64)                               # RMSProp slot "rms" is initialised to ONE in TF
    T.rmsprop_update(w, torch.tensor([2.0], dtype=torch.float64), ms, 3e-4)
    assert abs(float(ms) - (0.9 + 0.1 * 4.0)) < 1e-12 and abs(float(w) - (1.0 - 3e-4 * 2.0 / np.sqrt(1.3 + 1e-10))) < 1e-12


def test_variable_store_arena_layout():
    ss = pkg("source_segmenter")
    net = ss.Full_DRN(channels=3, n_class=5, batch_size=2, device="cpu", cost_kwargs={})
    st = net.store
    assert st.arena.numel() % 1024 == 0 and st.grad_arena.shape == st.arena.shape
    tot = 0
    for v in st.trainable():
        assert v.offset % 1024 == 0 and v.tensor.requires_grad and v.tensor.grad is not None
        assert v.tensor.data_ptr() == st.arena.data_ptr() + 4 * v.offset        # views of ONE flat arena
        assert v.tensor.grad.data_ptr() == st.grad_arena.data_ptr() + 4 * v.offset
        tot += v.numel
    assert tot == 39302456 + 30 * 2 * 0 + sum(v.numel for v in st.trainable() if v.kind == "bn")
    l2 = st.chunk_table(lambda v: 1e-4 * v.l2_mult, np.float32).numpy()
    w43 = st.vars["group_4/Variable_3"]
    assert np.allclose(l2[w43.offset // 1024], 2e-4) and l2[st.vars["group_4/Variable_2"].offset // 1024] == 0
    assert l2[st.vars["BatchNorm/gamma"].offset // 1024] == 0


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_two_rank_gradient_allreduce(tmp_path, world):
    """N>1 path on CPU: `world` ranks (gloo), each fills its gradient arena with rank-dependent values; after GradReducer.allreduce
    both hold the sum; buckets tile the arena exactly."""
    script = tmp_path / "w.py"
    script.write_text('''
import importlib, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
par = importlib.import_module("medical-cross-modality-domain-adaptation_amd.parallel")
ss = importlib.import_module("medical-cross-modality-domain-adaptation_amd.source_segmenter")
rank, local, world = par.init_distributed("gloo")
net = ss.Full_DRN(channels=3, n_class=5, batch_size=2, device="cpu", cost_kwargs={}, world_size=world)
red = par.GradReducer(net.store, bucket_bytes=8 << 20, overlap=True)
assert not red.overlap      # CPU tensors: plain all-reduce path
cover = np.zeros(net.store.arena.numel(), np.int32)
for s, e in red.buckets:
    cover[s:e] += 1
assert (cover == 1).all(), "buckets must tile the arena exactly once"
assert len(red.buckets) >= 10
net.store.grad_arena.fill_(float(rank + 1))
net.store.grad_arena[:7] = torch.arange(7, dtype=torch.float32) * (rank + 1)
red.allreduce()
exp = sum(r + 1 for r in range(world))
assert torch.all(net.store.grad_arena[7:] == exp)
assert torch.equal(net.store.grad_arena[:7], torch.arange(7, dtype=torch.float32) * exp)
# variable groups switched off for a step (GAN dis / gen steps): buckets without a trained variable are not reduced
keep = red._members[0]
for v in net.store.trainable():
    v.tensor.requires_grad_(any(v is k for k in keep))
net.store.grad_arena.fill_(float(rank + 1))
red.allreduce()
s0, e0 = red.buckets[0]
assert torch.all(net.store.grad_arena[s0:e0] == exp)
mask = torch.ones_like(net.store.grad_arena, dtype=torch.bool)
mask[s0:e0] = False
assert torch.all(net.store.grad_arena[mask] == float(rank + 1))
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
''' % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
                        "--master-port", str(29731 + 10 * world), str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert p.stdout.count("ok") == world


@pytest.mark.parametrize("world", [2, 4, 8])
def test_gloo_two_rank_bucket_order_and_set_agreement(tmp_path, world):
    """(world = 2, 4, 8: the 8-rank rehearsal of the bucket protocol the driver's N = 8 run will execute over RCCL.)
    The deadlock guards of GradReducer on gloo ranks over the ADAPTATION graph's variable store (376 variables, the dis / gen
    steps train different groups): (1) gradients become ready in a DIFFERENT order on the two ranks, yet both enqueue the same bucket
    sequence (index order) and end with the same sums; (2) dis-step and gen-step variable groups reduce different bucket sets; (3) a rank whose requires_grad flags differ makes EVERY rank raise before anything is enqueued — nobody hangs."""
    script = tmp_path / "w.py"
    script.write_text('''
import importlib, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
par = importlib.import_module("medical-cross-modality-domain-adaptation_amd.parallel")
adv = importlib.import_module("medical-cross-modality-domain-adaptation_amd.adversarial")
rank, local, world = par.init_distributed("gloo")
net = adv.Full_DRN(channels=3, n_class=5, batch_size=2, device="cpu", world_size=world,
                   network_config={"mr_front_trainable": False, "joint_trainable": False, "ct_front_trainable": True})
red = par.GradReducer(net.store, bucket_bytes=4 << 20, overlap=True)
assert len(red.buckets) >= 8
# CPU stand-in for the device path: overlap logic on, the launch is a synchronous gloo all-reduce of the bucket
red.overlap = True
def launch(b):
    assert not red._launched[b]
    red._launched[b] = True
    red.launch_log.append(b)
    s, e = red.buckets[b]
    dist.all_reduce(red.store.grad_arena[s:e])
red._launch = launch
class _S:                       # allreduce() waits on the side stream
    pass
import torch.cuda
red.side = None
cur = type("C", (), {"wait_stream": lambda self, s: None})()
torch.cuda.current_stream = lambda *a, **k: cur
hooks = {v.name: red._make_hook(v) for v in net.store.trainable()}

def step(group, order_seed):
    net._activate(group)
    net.store.grad_arena.fill_(float(rank + 1))
    names = [v.name for v in net.store.trainable() if v.tensor.requires_grad]
    rng = np.random.default_rng(order_seed)
    rng.shuffle(names)                                  # the order in which gradients become ready: different per rank
    for n in names:
        hooks[n](None)
    # every bucket of the step went out from the HOOKS (overlap), also the buckets that mix trainable and frozen variables: their
    # count is taken over the members that take a gradient in THIS step (round-3 advisor finding: counting all members left a mixed
    # bucket — and every complete bucket behind it — waiting for allreduce())
    assert red._active is not None and red._next == len(red._active), (group, red._next, red._active)
    mixed = [b for b in red._active if not all(v.tensor.requires_grad for v in red._members[b])]
    step.mixed += len(mixed)
    red.allreduce()
    log = list(red.launch_log)
    logs = [None] * world
    dist.all_gather_object(logs, log)
    assert all(l == logs[0] for l in logs), logs       # same collective sequence on every rank
    assert log == sorted(log)                           # index order
    exp = float(sum(r + 1 for r in range(world)))
    for b, (s, e) in enumerate(red.buckets):
        want = exp if b in log else float(rank + 1)
        assert torch.all(net.store.grad_arena[s:e] == want), (group, b)
    return log

step.mixed = 0
dis = step("cls", 100 + rank)
gen = step("adapt", 200 + rank)
assert step.mixed > 0, "no bucket mixes trainable and frozen variables: the mixed-bucket path is not exercised"
assert dis and gen and dis != gen and len(red.sets_seen) == 2
assert step("cls", 300 + rank) == dis and len(red.sets_seen) == 2
assert red.bytes_step == sum((red.buckets[b][1] - red.buckets[b][0]) * 4 for b in dis)
# (3) rank 1 forgets to freeze the critics in a generator step
net._activate("adapt")
if rank == 1:
    for v in net.store.trainable():
        if "cls" in v.name:
            v.tensor.requires_grad_(True)
try:
    red.allreduce()
    raise SystemExit("rank %%d: mismatched bucket sets went through" %% rank)
except RuntimeError as e:
    assert "disagree" in str(e), e
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
''' % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
                        "--master-port", str(29735 + 10 * world), str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert p.stdout.count("ok") == world


def test_gloo_two_rank_synchronised_statistics(tmp_path):
    """parallel.sync_bn_stats / all_sum_ on CPU with 2 ranks: the combined (mean, biased variance) of two half-batches equals the
    statistics of the whole batch; with synchronisation off both are identities."""
    script = tmp_path / "w.py"
    script.write_text('''
import importlib, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
par = importlib.import_module("medical-cross-modality-domain-adaptation_amd.parallel")
rank, local, world = par.init_distributed("gloo")
rng = np.random.default_rng(0)
x = torch.from_numpy((rng.standard_normal((2, 64, 7)) * [1, 2, 3, 4, 5, 6, 7] + np.arange(7) * 10.0).astype(np.float32))   # [rank][rows][C]
mine = x[rank]
mean, var = mine.mean(0), mine.var(0, unbiased=False)
m0, v0 = par.sync_bn_stats(mean, var)
assert m0 is mean and v0 is var and par.sync_world() == 1          # off: identity
t = torch.ones(3) * (rank + 1)
assert torch.equal(par.all_sum_(t.clone()), t)
par.enable_sync_stats()
assert par.sync_world() == 2
gm, gv = par.sync_bn_stats(mean, var)
full = x.reshape(-1, 7).double()
assert torch.allclose(gm.double(), full.mean(0), rtol=1e-6, atol=1e-6), (gm, full.mean(0))
assert torch.allclose(gv.double(), full.var(0, unbiased=False), rtol=1e-5), (gv, full.var(0, unbiased=False))
assert torch.equal(par.all_sum_(t.clone()), torch.ones(3) * 3)
par.disable_sync_stats()
assert par.sync_world() == 1
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
''' % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29733", str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert p.stdout.count("ok") == 2


def test_stride_phase_data_gradient_restatement():
    """the decomposition behind the strided data gradients of the HIP path (one stride-1 correlation per residue class of pixels) equals
    autograd's gradient for every stride / filter / padding / extent combination, including the ones the model never uses"""
    rng = np.random.default_rng(0)
    cases = 0
    for stride in (2, 3, 4):
        for R, S in ((3, 3), (5, 5), (2, 3), (4, 2), (7, 5), (1, 1), (2, 2)):
            for H, W in ((8, 8), (9, 7), (15, 13), (4, 5)):
                for padding in ("SAME", "VALID", "explicit"):
                    if padding == "SAME":
                        OH, pt, _ = T.same_pad(H, R, stride)
                        OW, pl, _ = T.same_pad(W, S, stride)
                    elif padding == "VALID":
                        if H < R or W < S:
                            continue
                        OH, OW, pt, pl = (H - R) // stride + 1, (W - S) // stride + 1, 0, 0
                    else:                                   # arbitrary explicit zero padding (incl. more than SAME would use)
                        pt, pl = R - 1, S // 2
                        OH, OW = (H + 2 * pt - R) // stride + 1, (W + 2 * pl - S) // stride + 1
                        if OH < 1 or OW < 1:
                            continue
                    x = torch.from_numpy(rng.standard_normal((2, H, W, 3))).requires_grad_(True)
                    w = torch.from_numpy(rng.standard_normal((R, S, 3, 4)))
                    xp = torch.nn.functional.pad(x.permute(0, 3, 1, 2), (pl, S + stride, pt, R + stride))     # generous zeros after
                    y = torch.nn.functional.conv2d(xp, w.permute(3, 2, 0, 1), stride=stride).permute(0, 2, 3, 1)[:, :OH, :OW]
                    dy = torch.from_numpy(rng.standard_normal(tuple(y.shape)))
                    y.backward(dy)
                    got = T.conv2d_dgrad_by_phases(dy, w, (H, W), stride, pt, pl)
                    assert torch.allclose(got, x.grad, rtol=1e-10, atol=1e-10), (stride, R, S, H, W, padding)
                    cases += 1
    assert cases > 200


def test_segmenter_training_schedule_with_stub_step(tmp_path):
    """source_segmenter.py:474-523 on the CPU with the train step stubbed out: one dequeued batch per step, the train-batch and
    validation monitoring forwards every display_step (the first in BN TRAIN mode like the reference), loss fetched one step late
    but for every step, final checkpoint by rank 0."""
    ss = pkg("source_segmenter")

    class Src(object):
        def __init__(self):
            self.n = 0

        def next_batch(self, B):
            self.n += 1
            b = np.zeros((B, 4, 4, 4), np.float32)
            b[..., 0] = self.n
            b[..., 3] = self.n % 5
            return b, ["f%d" % self.n] * B

    events = []

    class Net(object):
        device = torch.device("cpu")
        cost = torch.tensor(1.0)
        dice_eval = torch.tensor(0.5)
        confusion_matrix = np.eye(5)

        def evaluate(self, x, y, keep_prob=1.0, main_bn=True, adapt_bn=True, want_confusion=False):
            events.append(("eval", main_bn, adapt_bn, want_confusion, float(x[0, 0, 0, 0])))
            assert tuple(y.shape) == (2, 4, 4, 5) and float(y.sum()) == 2 * 16          # one-hot labels made on the device side
            return self.cost

        def save(self, path):
            events.append(("save",))
            return path

    class Opt(object):
        lr = 1e-3

        def state_dict(self):
            return {"m": np.zeros(4, np.float32), "v": np.zeros(4, np.float32), "t": np.int64(7), "lr": np.float64(self.lr)}

    tr = ss.Trainer(Net(), Src(), Src(), num_cls=5, batch_size=2, optimizer="adam", opt_kwargs={"learning_rate": 1e-3})
    tr.opt = Opt()
    tr.train_step = lambda bx, by, dropout, step: (events.append(("step", step, float(bx[0, 0, 0, 0]), dropout)), torch.tensor(float(step)))[1]
    logged = []
    import logging
    h = logging.Handler()
    h.emit = lambda rec: logged.append(rec.getMessage())
    logging.getLogger().addHandler(h)
    lvl = logging.getLogger().level
    logging.getLogger().setLevel(logging.INFO)
    ss_verbose = ss.verbose
    try:
        tr.train(str(tmp_path / "o"), restore=False, training_iters=7, epochs=1, display_step=5, dropout=0.75)
    finally:
        logging.getLogger().removeHandler(h)
        logging.getLogger().setLevel(lvl)
        ss.verbose = ss_verbose
    kinds = [e[0] for e in events]
    assert kinds == ["step", "eval", "eval"] + ["step"] * 4 + ["step", "eval", "eval"] + ["step", "save"], kinds
    steps = [e for e in events if e[0] == "step"]
    assert [e[1] for e in steps] == list(range(7)) and [e[2] for e in steps] == [float(i) for i in range(1, 8)] and steps[0][3] == 0.75
    evals = [e for e in events if e[0] == "eval"]
    assert evals[0][1:4] == (True, True, False) and evals[0][4] == 1.0        # train batch of step 0, BN train mode
    assert evals[1][1:4] == (False, False, True)                             # validation batch, inference mode, confusion matrix
    assert evals[2][4] == 6.0                                                # train batch of step 5
    losses = [m for m in logged if m.startswith("Training at step")]
    assert len(losses) == 7 and "step 6 " in losses[-1] and "6.0000" in losses[-1]
    assert len(tr.step_times) == 7
    with np.load(str(tmp_path / "o" / "optimizer.npz")) as z:                 # tf.train.Saver semantics: slots, step and lr travel with the model
        assert str(z["kind"]) == "adam" and int(z["t"]) == 7 and int(z["global_step"]) == 0 and float(z["lr"]) == 1e-3


def _same_slots(store, a, b, select=None):
    """slot arenas agree on every (selected) variable's range — the chunk-alignment gaps between variables are not state"""
    return all(torch.equal(a[v.offset:v.offset + v.numel], b[v.offset:v.offset + v.numel]) for v in store.trainable()
               if select is None or select(v))


def test_optimizer_state_travels_with_checkpoints(tmp_path):
    """Adam / momentum / RMSProp slots, step counters and learning rates are TF variables in the reference, i.e. saved and restored by
    tf.train.Saver; restore honours lr_update_flag / clear_rms / lr_update (source_segmenter.py:460-462, adversarial.py:503-574, 803-805)"""
    ss, adv = pkg("source_segmenter"), pkg("adversarial")
    net = ss.Full_DRN(channels=3, n_class=5, batch_size=2, device="cpu", cost_kwargs={"regularizer": 1e-4})
    out = str(tmp_path / "seg")
    os.makedirs(out)
    tr = ss.Trainer(net, None, None, num_cls=5, batch_size=2, optimizer="adam", opt_kwargs={"learning_rate": 1e-3})
    tr.opt = tr._get_optimizer(10)
    tr.opt.m.normal_()
    tr.opt.v.uniform_()
    tr.opt.t, tr.opt.lr, tr.global_step = 41, 5e-4, 41
    tr.save_checkpoint(out)
    for flag, lr_expected in ((False, 5e-4), (True, 1e-3)):
        tr2 = ss.Trainer(net, None, None, num_cls=5, batch_size=2, optimizer="adam", opt_kwargs={"learning_rate": 1e-3}, lr_update_flag=flag)
        tr2.opt = tr2._get_optimizer(10)
        assert tr2.restore_optimizer(out)
        assert _same_slots(net.store, tr2.opt.m, tr.opt.m) and _same_slots(net.store, tr2.opt.v, tr.opt.v) and tr2.opt.t == 41 and tr2.global_step == 41
        assert tr2.opt.lr == lr_expected
    trm = ss.Trainer(net, None, None, num_cls=5, batch_size=2, optimizer="momentum", opt_kwargs={"learning_rate": 0.2})
    trm.opt = trm._get_optimizer(10)
    assert not trm.restore_optimizer(out)                        # an Adam checkpoint does not feed a momentum optimiser
    trm.opt.t = 25
    trm.opt.lr = 0.05                                            # periodic decay assigns the rate (staircase factor divided out)
    assert abs(trm.opt.lr - 0.05) < 1e-12
    # adaptation trainer: two RMSProp optimisers
    anet = adv.Full_DRN(channels=3, n_class=5, batch_size=2, device="cpu",
                        cost_kwargs={"regularizer": 1e-4, "gan_regularizer": 1e-4, "miu_gen": 0.002, "miu_dis": 0.002, "lambda_mask_loss": 0.3},
                        network_config={"mr_front_trainable": False, "joint_trainable": False, "ct_front_trainable": True,
                                        "cls_trainable": True, "m_cls_trainable": True})
    aout = str(tmp_path / "gan")
    os.makedirs(aout)
    at = adv.Trainer(anet, None, None, None, None, num_cls=5, batch_size=2, opt_kwargs={"learning_rate": 3e-4})
    at._get_optimizer()
    at.dis_optimizer.ms.uniform_(0.5, 2.0)
    at.gen_optimizer.ms.uniform_(0.5, 2.0)
    at.dis_optimizer.lr = at.gen_optimizer.lr = 1e-4
    at.global_step = 9
    at.save_checkpoint(aout)
    for clear_rms, lr_update in ((False, False), (True, False), (False, True)):
        a2 = adv.Trainer(anet, None, None, None, None, num_cls=5, batch_size=2, opt_kwargs={"learning_rate": 3e-4})
        a2._get_optimizer()
        assert a2.restore_optimizer(aout, clear_rms=clear_rms, lr_update=lr_update)
        same = (_same_slots(anet.store, a2.dis_optimizer.ms, at.dis_optimizer.ms, lambda v: "cls" in v.name) and
                _same_slots(anet.store, a2.gen_optimizer.ms, at.gen_optimizer.ms, lambda v: "adapt" in v.name))
        assert same == (not clear_rms)
        if clear_rms:
            assert torch.all(a2.dis_optimizer.ms == 1.0)                     # TF's RMSProp slot initial value
        assert a2.dis_optimizer.lr == (3e-4 if lr_update else 1e-4) and a2.global_step == 9
    assert not at.restore_optimizer(out, False, False)           # the segmenter's checkpoint folder has no RMSProp state for this graph
    # the documented hand-off --phase pre-train -> --phase train-gan: the pre-train graph freezes adapt_* (its arena holds the critics
    # only), the train-gan graph trains adapt_* as well (another arena layout).  tf.train.Saver restores slots BY NAME: the critics'
    # warmed-up RMSProp slots and the global step must arrive, the adapt_* slots start at TF's initial value 1.0
    pnet = adv.Full_DRN(channels=3, n_class=5, batch_size=2, device="cpu",
                        cost_kwargs={"regularizer": 1e-4, "gan_regularizer": 1e-4, "miu_gen": 0.002, "miu_dis": 0.002, "lambda_mask_loss": 0.0},
                        network_config={"mr_front_trainable": False, "joint_trainable": False, "ct_front_trainable": False,
                                        "cls_trainable": True, "m_cls_trainable": True})
    assert pnet.store.arena.numel() != anet.store.arena.numel()
    pt = adv.Trainer(pnet, None, None, None, None, num_cls=5, batch_size=2, opt_kwargs={"learning_rate": 3e-4})
    pt._get_optimizer()
    pt.dis_optimizer.ms.uniform_(0.5, 2.0)
    pt.global_step = 123
    pout = str(tmp_path / "pre")
    os.makedirs(pout)
    pt.save_checkpoint(pout)
    g2 = adv.Trainer(anet, None, None, None, None, num_cls=5, batch_size=2, opt_kwargs={"learning_rate": 3e-4})
    g2._get_optimizer()
    assert g2.restore_optimizer(pout, clear_rms=False, lr_update=True) and g2.global_step == 123
    pv = {v.name: v for v in pnet.store.trainable()}
    n_cls = 0
    for v in anet.store.trainable():
        got = g2.dis_optimizer.ms[v.offset:v.offset + v.numel]
        if "cls" in v.name:
            want = pt.dis_optimizer.ms[pv[v.name].offset:pv[v.name].offset + v.numel]
            assert torch.equal(got, want), v.name
            n_cls += 1
        elif "adapt" in v.name:
            assert torch.all(g2.gen_optimizer.ms[v.offset:v.offset + v.numel] == 1.0), v.name
    assert n_cls > 40
    # a restore that asks for slots (clear_rms=False) from a folder that holds none for this graph says so instead of dropping them
    np.savez(os.path.join(pout, "optimizer.npz"), kind="rmsprop", dis_lr=np.float64(1e-4), gen_lr=np.float64(1e-4), global_step=np.int64(1))
    with pytest.raises(RuntimeError):
        g2.restore_optimizer(pout, clear_rms=False, lr_update=False)
    assert g2.restore_optimizer(pout, clear_rms=True, lr_update=False)
    # no_gan restore: only the main ('group' / 'output') variables, neither adapt_* nor the critics
    before = anet.store.state_dict()
    tweaked = {k: v + 1.0 for k, v in before.items()}
    np.savez(os.path.join(aout, "tweaked.npz"), **{k.replace("/", "|"): v for k, v in tweaked.items()})
    anet.restore(None, os.path.join(aout, "tweaked.npz"), no_gan=True)
    after = anet.store.state_dict()
    for k in before:
        moved = not np.array_equal(after[k], before[k])
        assert moved == (("group" in k or "output" in k) and "adapt" not in k and "cls" not in k), k


def test_mfma_16x16x4_tile_restatement_of_the_small_convolutions():
    """the lane / tile index maps of csrc/conv_small.hip (k-permuted channel quarters, 8x32 output tiles, 4-pixel reduction groups of
    the filter gradient) restated in numpy equal the plain convolution and autograd's filter gradient, ragged extents included"""
    rng = np.random.default_rng(3)
    for (N, H, W, C, K, pad) in ((1, 9, 35, 16, 16, 1), (2, 8, 33, 32, 16, 1), (1, 12, 40, 16, 16, 0)):
        x = rng.standard_normal((N, H, W, C))
        w = rng.standard_normal((3, 3, C, K))
        xt = torch.from_numpy(x)
        wt = torch.from_numpy(w).requires_grad_(True)
        ref = T.conv2d(xt, wt, 1, 1, "SAME" if pad else "VALID")
        got = T.conv3x3_n16_by_mfma_tiles(x, w, pad)
        assert got.shape == tuple(ref.shape) and np.allclose(got, ref.detach().numpy(), rtol=1e-11, atol=1e-11), (N, H, W, C, K, pad)
        dy = rng.standard_normal(tuple(ref.shape))
        ref.backward(torch.from_numpy(dy))
        assert np.allclose(T.wgrad3x3_n16_by_mfma_tiles(x, dy, pad), wt.grad.numpy(), rtol=1e-11, atol=1e-10)
    # 32 filters = two 16-filter workgroup columns of the filter gradient
    x, dy = rng.standard_normal((1, 8, 34, 16)), rng.standard_normal((1, 8, 34, 32))
    wt = torch.zeros((3, 3, 16, 32), dtype=torch.float64, requires_grad=True)
    T.conv2d(torch.from_numpy(x), wt, 1, 1, "SAME").backward(torch.from_numpy(dy))
    assert np.allclose(T.wgrad3x3_n16_by_mfma_tiles(x, dy, 1), wt.grad.numpy(), rtol=1e-11, atol=1e-10)


def test_ring_filter_gradient_coordinate_walk():
    """conv_wgrad_ring_kernel carries (ih, iw, offset) of a loader row incrementally over the strided output walk; restated in
    oracle.tf_ops.wgrad_ring_walk and held to the closed form for every stride / dilation / padding / tap / start pixel swept here
    (the -m gpu tests can only afford a handful of geometries)"""
    checked = 0
    for stride in (1, 2, 3):
        for dil in (1, 2):
            for R in (1, 3, 5):
                for (H, W) in ((40, 70), (33, 97), (64, 64), (35, 140)):
                    for padding in ("SAME", "VALID"):
                        eff = (R - 1) * dil + 1
                        if padding == "SAME":
                            OH, pt, _ = T.same_pad(H, R, stride, dil)
                            OW, pl, _ = T.same_pad(W, R, stride, dil)
                        else:
                            if H < eff or W < eff:
                                continue
                            OH, OW, pt, pl = (H - eff) // stride + 1, (W - eff) // stride + 1, 0, 0
                        if OW < 32:
                            continue                      # the kernel's precondition (one row wrap per 32-pixel step)
                        N, C = 3, 8
                        P = N * OH * OW
                        for (r, s, c) in ((0, 0, 0), (R - 1, R // 2, 4), (R // 2, R - 1, 4)):
                            for p_first in (0, 7, 32 * 5 + 3, OH * OW - 1, OH * OW + 9):
                                nst = (P - p_first + 31) // 32 + 2          # runs past the last pixel like the kernel's ring does
                                walk = T.wgrad_ring_walk(N, H, W, C, OH, OW, stride, dil, pt, pl, r, s, c, p_first, nst)
                                for k, (ok, off) in enumerate(walk):
                                    p = p_first + 32 * k
                                    n, rem = divmod(p, OH * OW)
                                    oh, ow = divmod(rem, OW)
                                    ih, iw = oh * stride - pt + r * dil, ow * stride - pl + s * dil
                                    assert off == (((n * H + ih) * W + iw) * C + c) * 4, (stride, dil, R, H, W, padding, r, s, p_first, k)
                                    assert ok == (0 <= ih < H and 0 <= iw < W)
                                    if p >= P and ok:
                                        assert off >= N * H * W * C * 4     # rows past the last pixel lie beyond x: hardware zero
                                    checked += 1
    assert checked > 50000


@pytest.mark.parametrize("m", [2, 4])
def test_winograd_restatement_equals_the_direct_convolution(m):
    """the tile enumeration (dilation phases, ragged m x m tiles, padding 0 / dil / 2 dil), the three transforms and the flipped, transposed
    filter of the data gradient of csrc/conv_wino.hip, restated in numpy (oracle.tf_ops.conv3x3_winograd_np) for F(2x2, 3x3) and F(4x4, 3x3):
    equal to tf.nn.conv2d / atrous_conv2d and to autograd's data gradient in float64; in float32 as close to float64 as the direct sum is"""
    rng = np.random.default_rng(11)
    tol = 1e-12 if m == 2 else 1e-11            # (F(4,3): thirds and fifteenths in G, entries up to 8 in A^T)
    for (N, H, W, C, K, d, pad) in ((2, 8, 8, 4, 5, 1, 1), (1, 6, 10, 3, 4, 2, 2), (2, 5, 7, 4, 4, 1, 1), (1, 12, 4, 2, 3, 2, 2),
                                    (1, 7, 9, 3, 2, 1, 0), (1, 8, 12, 3, 2, 2, 0), (1, 2, 2, 3, 2, 1, 1), (1, 13, 9, 2, 3, 1, 1)):
        x, w = rng.standard_normal((N, H, W, C)), rng.standard_normal((3, 3, C, K))
        xt = torch.from_numpy(x).requires_grad_(True)
        ref = T.conv2d(xt, torch.from_numpy(w), 1, d, "SAME" if pad == d else "VALID")
        got = T.conv3x3_winograd_np(x, w, d, dtype=np.float64, pad=pad, m=m)
        assert got.shape == tuple(ref.shape) and np.allclose(got, ref.detach().numpy(), rtol=tol, atol=tol), (N, H, W, C, K, d, pad)
        dy = rng.standard_normal(tuple(ref.shape))
        ref.backward(torch.from_numpy(dy))
        gd = T.conv3x3_winograd_np(dy, w, d, flip_transpose=True, dtype=np.float64, pad=2 * d - pad, m=m)
        assert gd.shape == x.shape and np.allclose(gd, xt.grad.numpy(), rtol=tol, atol=tol), (N, H, W, C, K, d, pad)
        # the filter gradient: the transposed algorithm (dy tile spread to the transform points, reduction over tiles split 3 ways)
        wt = torch.from_numpy(w).requires_grad_(True)
        T.conv2d(torch.from_numpy(x), wt, 1, d, "SAME" if pad == d else "VALID").backward(torch.from_numpy(dy))
        gw = T.wgrad3x3_winograd_np(x, dy, d, dtype=np.float64, pad=pad, nsplit=3, m=m)
        assert gw.shape == w.shape and np.allclose(gw, wt.grad.numpy(), rtol=10 * tol, atol=10 * tol), (N, H, W, C, K, d, pad)
    x = rng.standard_normal((1, 8, 8, 512)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 512, 32)) * 0.02).astype(np.float32)
    ref = T.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), 1, 1, "SAME").numpy()
    e_w = np.abs(T.conv3x3_winograd_np(x, w, 1, dtype=np.float32, m=m) - ref).max() / np.abs(ref).max()
    e_d = np.abs(T.conv2d(torch.from_numpy(x), torch.from_numpy(w), 1, 1, "SAME").numpy() - ref).max() / np.abs(ref).max()
    # F(2,3): as close as the direct sum; F(4,3) on the points (0, +-1, 1/2, -2): within 2e-5 of max|y| (tools/wino_f43_study.py: 4e-6 typical)
    assert (e_w < 5e-6 and e_w < 4 * e_d + 1e-6) if m == 2 else e_w < 2e-5, (e_w, e_d)


def test_split_bf16_gemm_restatement():
    """round 6: the arithmetic of csrc/conv_wino_x3.hip restated (oracle.tf_ops.split3_bf16 / gemm_x3_np): a float32 value IS the sum of its
    three bf16 planes; the six kept plane products with fp32 accumulation in 64-channel chunks land closer to the float64 product than a
    float32 chain does; and the F(4x4, 3x3) layer on that GEMM — forward, data gradient (a 2 560-channel reduction: the case F(4x4) on the
    fp32 pipe was capped for) and filter gradient — is closer to the float64 convolution than on plain float32 GEMMs"""
    rng = np.random.default_rng(23)
    a = (rng.standard_normal(4096) * np.exp(rng.uniform(-30, 30, 4096))).astype(np.float32)
    hi, mid, lo = T.split3_bf16(a)
    for p in (hi, mid, lo):          # each plane is a bfloat16 value: the low 16 bits of its float32 pattern are zero
        assert not (p.view(np.uint32) & 0xFFFF).any()
    assert np.array_equal((hi.astype(np.float64) + mid + lo).astype(np.float32), a) and np.array_equal(hi + mid + lo, a)
    assert np.abs(mid).max() <= np.abs(hi).max() * 2.0 ** -8 and (np.abs(lo) <= np.abs(a) * 2.0 ** -16).all()
    A = rng.standard_normal((3, 64, 512)).astype(np.float32)
    B = (rng.standard_normal((3, 512, 48)) * 0.05).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    seq = np.zeros(ref.shape, np.float32)
    for c in range(512):             # one fp32 rounding per multiply-add: the fp32 matrix pipe's chain
        seq = (seq.astype(np.float64) + A[..., c:c + 1].astype(np.float64) * B[:, c:c + 1, :]).astype(np.float32)
    e_x3 = np.abs(T.gemm_x3_np(A, B) - ref).max() / np.abs(ref).max()
    e_seq = np.abs(seq - ref).max() / np.abs(ref).max()
    print("GEMM over 512 channels vs float64: fp32 chain %.2e, split-bf16 six products in 64-channel chunks %.2e" % (e_seq, e_x3))
    assert e_x3 < 2e-7 and e_x3 < 0.5 * e_seq
    # the whole layer
    x = rng.standard_normal((1, 8, 8, 512)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 512, 32)) * 0.02).astype(np.float32)
    ref = T.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), 1, 1, "SAME").numpy()
    rel = lambda got, r: float(np.abs(got - r).max() / np.abs(r).max())
    e4, e4x = rel(T.conv3x3_winograd_np(x, w, 1, dtype=np.float32, m=4), ref), rel(T.conv3x3_winograd_np(x, w, 1, dtype=np.float32, m=4, x3=True), ref)
    print("F(4x4) 512->32 forward vs float64: float32 GEMMs %.2e, split-bf16 GEMMs %.2e" % (e4, e4x))
    assert e4x < 3e-6 and e4x < e4
    dy = rng.standard_normal((1, 8, 8, 2560)).astype(np.float32)
    wd = (rng.standard_normal((3, 3, 32, 2560)) * 0.01).astype(np.float32)
    dyt = torch.from_numpy(dy).double()
    xg = torch.zeros((1, 8, 8, 32), dtype=torch.float64, requires_grad=True)
    T.conv2d(xg, torch.from_numpy(wd).double(), 1, 1, "SAME").backward(dyt)
    ed, edx = (rel(T.conv3x3_winograd_np(dy, wd, 1, flip_transpose=True, dtype=np.float32, m=4, x3=f), xg.grad.numpy()) for f in (False, True))
    print("F(4x4) data gradient over 2 560 channels vs float64: float32 GEMMs %.2e, split-bf16 GEMMs %.2e" % (ed, edx))
    assert edx < 3e-6 and edx < ed
    xs = rng.standard_normal((2, 16, 16, 32)).astype(np.float32)
    dys = rng.standard_normal((2, 16, 16, 48)).astype(np.float32)
    wt = torch.zeros((3, 3, 32, 48), dtype=torch.float64, requires_grad=True)
    T.conv2d(torch.from_numpy(xs).double(), wt, 1, 1, "SAME").backward(torch.from_numpy(dys).double())
    ew, ewx = (rel(T.wgrad3x3_winograd_np(xs, dys, 1, dtype=np.float32, nsplit=1, m=4, x3=f), wt.grad.numpy()) for f in (False, True))
    print("F(4x4) filter gradient over 32 tiles (padded to 64) vs float64: float32 GEMMs %.2e, split-bf16 GEMMs %.2e" % (ew, ewx))
    assert ewx < 3e-6


def test_written_ranges_of_an_optimiser_mask():
    """VariableStore.written_ranges: the byte ranges kernels.weights_changed hands to pnp_weights_changed — maximal runs of selected
    chunks of the trainable arena (an optimiser over a var_list writes exactly those)"""
    V = pkg("variables")
    L = pkg("_lib")
    st = V.VariableStore(device="cpu", seed=0)
    for i, n in enumerate((10, 3 * L.OPT_CHUNK, 5, L.OPT_CHUNK + 1)):
        st.get("v%d" % i, (n,), init=0.0)
    st.finalize()
    base, ch = st.arena.data_ptr(), 4 * L.OPT_CHUNK
    assert st.written_ranges(None) == [(base, base + 4 * st.arena.numel())]
    nch = st.arena.numel() // L.OPT_CHUNK
    assert nch == 1 + 3 + 1 + 2
    mask = st.chunk_table(lambda v: 1 if v.name in ("v1", "v3") else 0, np.uint8).numpy()
    assert list(mask) == [0, 1, 1, 1, 0, 1, 1]
    assert st.written_ranges(mask) == [(base + ch, base + 4 * ch), (base + 5 * ch, base + 7 * ch)]
    assert st.written_ranges(np.zeros(nch, np.uint8)) == []


def test_transformed_filter_cache_rules_on_the_host(monkeypatch):
    """kernels._wino_u (the Python side of pnp_conv2d_wino_filter_bind) against a recording stand-in for the library: a store-owned filter
    gets ONE binding per pass; a torch in-place write (version counter) reports the filter's byte range; the binding is withdrawn when the
    tensor object dies and when an ad-hoc filter turns up on a cached address; ad-hoc filters are never bound; weights_changed hands
    ranges through"""
    import ctypes
    K = pkg("kernels")
    calls = []

    class FakeLib(object):
        def pnp_conv2d_wino_filter_bytes(self, C, Kf):
            return 36 * C * Kf * 4

        def pnp_conv2d_wino_filter_bind(self, w, kind, U, n):
            calls.append(("bind" if U is not None else "unbind", getattr(w, "value", w), kind))
            return 0

        def pnp_weights_changed(self, lo, hi):
            calls.append(("changed", getattr(lo, "value", lo), getattr(hi, "value", hi)))

        def pnp_conv2d_wino_chosen(self, g, kind):
            return 4

    monkeypatch.setattr(K._lib, "load", lambda: FakeLib())
    monkeypatch.setattr(ctypes, "byref", lambda g: g)
    monkeypatch.setattr(K, "_u_cache", {})
    monkeypatch.setattr(K, "U_CACHE", True)

    class G(object):
        R = S = 3

    w = torch.zeros(3, 3, 32, 32)
    K._wino_u(w, G(), 0)                                  # ad-hoc: nothing
    assert calls == [] and not K._u_cache
    w._pnp_var = True
    K._wino_u(w, G(), 0)
    K._wino_u(w, G(), 0)
    K._wino_u(w, G(), 1)
    assert calls == [("bind", w.data_ptr(), 0), ("bind", w.data_ptr(), 1)] and len(K._u_cache) == 2
    w.add_(1.0)                                           # a torch op on the filter
    K._wino_u(w, G(), 0)
    assert calls[-1] == ("changed", w.data_ptr(), w.data_ptr() + 4 * w.numel())
    K.weights_changed([(10, 20), (30, 40)])
    assert calls[-2:] == [("changed", 10, 20), ("changed", 30, 40)]
    K.weights_changed()
    assert calls[-1] == ("changed", None, None)
    ptr = w.data_ptr()
    del w                                                 # the tensor goes: both bindings with it
    assert not K._u_cache and sorted(calls[-2:]) == [("unbind", ptr, 0), ("unbind", ptr, 1)]
    w2 = torch.zeros(3, 3, 32, 32)
    w2._pnp_var = True
    K._wino_u(w2, G(), 1)
    v = torch.zeros(3, 3, 32, 32)                         # an ad-hoc filter "on the same address": move the entry under its key
    K._u_cache[(v.data_ptr(), 1)] = K._u_cache.pop((w2.data_ptr(), 1))
    K._wino_u(v, G(), 1)
    assert calls[-1] == ("unbind", v.data_ptr(), 1) and not K._u_cache


def test_captured_gan_steps_fall_back_to_eager_when_the_learning_rate_moved():
    """adversarial.Trainer._captured: a recording froze the RMSProp learning rates by value — after a decay
    (Trainer.train, restore_optimizer) the captured steps must not be replayed"""
    adv = pkg("adversarial")

    class Opt(object):
        def __init__(self, lr):
            self.lr = lr

    t = adv.Trainer.__new__(adv.Trainer)
    t.dis_optimizer, t.gen_optimizer = Opt(3e-4), Opt(3e-4)
    t._cap = {"dropout": 0.75, "mr": (2, 256, 256, 3), "ct": (2, 256, 256, 3), "lr": (3e-4, 3e-4), "dis": "D", "gen": "G"}
    assert t._captured("dis", 0.75, {"mr": (2, 256, 256, 3), "ct": (2, 256, 256, 3)}) == "D"
    assert t._captured("gen", 0.75, {"ct": (2, 256, 256, 3)}) == "G"
    assert t._captured("gen", 0.5, {"ct": (2, 256, 256, 3)}) is None and t._captured("gen", 0.75, {"ct": (4, 256, 256, 3)}) is None
    t.gen_optimizer.lr *= 0.95
    assert t._captured("dis", 0.75, {"mr": (2, 256, 256, 3), "ct": (2, 256, 256, 3)}) is None
    assert t._captured("gen", 0.75, {"ct": (2, 256, 256, 3)}) is None and t._cap.get("warned")


def test_memory_report_counts_what_the_caches_hold():
    """kernels.memory_report: every cache the host side keeps between steps is in the sum"""
    import torch
    K = pkg("kernels")
    before = K.memory_report()
    key = ("cpu", "report-test", 0)
    K._ws_cache[key] = torch.empty(1 << 12, dtype=torch.uint8)
    K._ws_retired.append(torch.empty(1 << 10, dtype=torch.uint8))
    try:
        rep = K.memory_report()
        assert rep["workspaces"] == before["workspaces"] + (1 << 12) and rep["workspace_buffers"] == before["workspace_buffers"] + 1
        assert rep["workspaces_retired"] == before["workspaces_retired"] + (1 << 10)
        assert rep["total"] == before["total"] + (1 << 12) + (1 << 10)
        assert set(rep) >= {"winograd_filters", "bf16_filter_shadows", "recorded_steps_alive"}
    finally:
        del K._ws_cache[key]
        K._ws_retired.pop()

import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
P = "medical-cross-modality-domain-adaptation_amd."
K, L, F, layers, variables = (importlib.import_module(P + m) for m in ("kernels", "_lib", "functional", "layers", "variables"))
dev = torch.device("cuda:0")
rng = np.random.default_rng(2)
x0 = torch.from_numpy(rng.standard_normal((4, 32, 32, 128)).astype(np.float32)).to(dev)
rel = lambda a, b: 1.0 - float((a.double().reshape(-1) * b.double().reshape(-1)).sum() / (a.double().norm() * b.double().norm()))

def run(resident, nblocks, keep, dtype="bf16"):
    F.set_conv_dtype(dtype)
    old = K.bf16r
    if not resident:
        K.bf16r = lambda g, kind: False
    try:
        store = variables.VariableStore(dev, seed=4)
        with store.as_default():
            def graph(xin):
                store.begin_trace(drop_seed=11)
                with store.name_scope("group_a"):
                    w = [layers.weight_variable(s_, stddev=0.03) for s_ in ([3, 3, 128, 128], [3, 3, 128, 128], [3, 3, 128, 256], [3, 3, 256, 256])]
                    if nblocks == 0:
                        return layers.conv_bn_relu2d(xin, w[0], keep, is_train=True)
                    h = layers.residual_block(xin, w[0], w[1], keep, is_train=True)
                    if nblocks == 1:
                        return h
                    return layers.residual_block(h, w[2], w[3], keep, inc_dim=True, is_train=True)
            graph(x0)
            store.finalize()
            xin = x0.clone().requires_grad_(True)
            store.zero_grad()
            out = graph(xin)
            out.backward(torch.from_numpy(np.random.default_rng(5).standard_normal(tuple(out.shape)).astype(np.float32)).to(dev))
            torch.cuda.synchronize()
            return out.detach().clone(), xin.grad.clone(), store.grad_arena.detach().clone()
    finally:
        K.bf16r = old
        F.set_conv_dtype("f32")

for link in (True,):
    F.RES_LINK = link
    for nb in (1, 2):
        for keep in (0.75,):
            a = run(True, nb, keep); b = run(False, nb, keep); c = run(False, nb, keep, "f32")
            print("link %s blocks %d keep %.2f: resident vs staged out %.2e dx %.2e grads %.2e | resident vs f32 %.2e %.2e %.2e | staged vs f32 %.2e %.2e %.2e" % (
                link, nb, keep, rel(a[0], b[0]), rel(a[1], b[1]), rel(a[2], b[2]), rel(a[0], c[0]), rel(a[1], c[1]), rel(a[2], c[2]),
                rel(b[0], c[0]), rel(b[1], c[1]), rel(b[2], c[2])), flush=True)

#!/bin/bash
# round 5, call 9: tail split of the persistent GEMM — parity, per-layer GEMM times (PNP_WINO_TAILSPLIT=0/1), joint-step A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5i; mkdir -p $O
(timeout 300 python -m pytest tests/test_gpu_wino.py -q -x -p no:cacheprovider 2>&1 | tail -8) | tee $O/pytest_wino.log
# the B=16 layers whose tile counts trigger the split, against float64 (the parity cases above are mostly below 512 tiles): forward + data gradient
timeout 200 python - <<'PY' 2>&1 | tee $O/parity_big.log
import importlib, sys, numpy as np, torch
sys.path.insert(0, ".")
K = importlib.import_module("medical-cross-modality-domain-adaptation_amd.kernels")
from oracle import tf_ops as T
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
K.wino_mode(2); K.wino_tile(4)
for (N, H, C, Kf, dil, pad) in ((16, 32, 512, 512, 1, "SAME"), (16, 32, 256, 256, 1, "SAME"), (8, 32, 512, 512, 2, "SAME"), (4, 64, 256, 256, 1, "SAME"), (6, 34, 512, 2560, 1, "VALID"), (16, 32, 256, 512, 1, "SAME"), (3, 30, 512, 544, 1, "SAME")):
    x = rng.standard_normal((N, H, H, C)).astype(np.float32); w = (rng.standard_normal((3, 3, C, Kf)) * np.sqrt(2.0 / (9 * C))).astype(np.float32)
    g = K.conv_geom(x.shape, w.shape, 1, dil, pad)
    dy = rng.standard_normal((N, g.OH, g.OW, Kf)).astype(np.float32)
    xd, wd, dyd = (torch.from_numpy(a).to(dev) for a in (x, w, dy))
    y = K.conv2d_fwd(xd, wd, g); dx = K.conv2d_dgrad(dyd, wd, g)
    K.wino_mode(0); y0 = K.conv2d_fwd(xd, wd, g); dx0 = K.conv2d_dgrad(dyd, wd, g); K.wino_mode(2)
    ey = float((y - y0).abs().max() / y0.abs().max()); ed = float((dx - dx0).abs().max() / dx0.abs().max())
    print("big", (N, H, C, Kf, dil, pad), "tiles fwd/dgrad", K.wino_chosen(g, 0), K.wino_chosen(g, 1), "y vs direct %.2e dx vs direct %.2e" % (ey, ed), "OK" if max(ey, ed) < 2e-5 else "FAIL")
PY
L="g5/6,g7,g8,g10,cls3 256,cls5"
for p in 0 1; do echo "== PNP_WINO_TAILSPLIT=$p"; PNP_WINO_TAILSPLIT=$p PROF=1 SKIP_WGRAD=1 ONLY="$L" WINO=2 WINO_WGRAD=2 TILE=4 timeout 150 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids | grep "wino_gemm\|wino_out\|->" | tee $O/prof_split$p.txt; done
run() { local tag=$1; shift
  env "$@" timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sub > $O/bench_$tag.json 2> $O/bench_$tag.err
  python -c "
import json; r=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print('$tag joint', r['value'], r['ms_per_step'], r['roofline']['kernel'], r['roofline']['frac'], r['roofline']['avg_launch_ms'])" || tail -5 $O/bench_$tag.err
}
run nosplit PNP_WINO_TAILSPLIT=0
run split X=1
run nosplit2 PNP_WINO_TAILSPLIT=0
run split2 X=1

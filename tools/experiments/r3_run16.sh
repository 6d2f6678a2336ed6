# round 3, GPU call 16: host side of the library under UBSan + libstdc++ assertions on the GPU (ASAN cannot run there: run 15)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p; mkdir -p $O
P=$GRAFT_REPO_ROOT/medical-cross-modality-domain-adaptation_amd
PNP_LIB=$P/libpnp_hip_ubsan.so UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 timeout 900 python -m pytest tests/test_abi.py tests/test_api_errors.py \
   tests/test_gpu_conv.py tests/test_gpu_elementwise.py tests/test_gpu_loss_optim.py tests/test_gpu_adversarial.py tests/test_gpu_bf16.py -q -s > $O/ubsan_gpu.log 2>&1
echo "rc=$?" >> $O/ubsan_gpu.log; grep -c "runtime error" $O/ubsan_gpu.log; tail -4 $O/ubsan_gpu.log
python - <<'PY' >> $O/ubsan_gpu.log 2>&1
import os, ctypes
os.environ["PNP_LIB"] = os.environ.get("GRAFT_REPO_ROOT", ".") + "/medical-cross-modality-domain-adaptation_amd/libpnp_hip_ubsan.so"
import importlib, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
L = importlib.import_module("medical-cross-modality-domain-adaptation_amd._lib")
L.load()
print("library under test:", sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "libpnp" in l or "ubsan" in l)))
PY
tail -2 $O/ubsan_gpu.log | cut -c1-300

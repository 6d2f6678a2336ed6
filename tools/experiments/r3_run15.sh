# round 3, GPU call 15: the ASAN GPU run dies silently inside the first GPU test (rc 1, pytest's fd capture swallows the report): -s + log_path
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3o; mkdir -p $O
P=$GRAFT_REPO_ROOT/medical-cross-modality-domain-adaptation_amd
ASAN_LIB=$(/opt/rocm/bin/hipcc -print-file-name=libclang_rt.asan-x86_64.so)
LD_PRELOAD=$ASAN_LIB ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:log_path=$GRAFT_REPO_ROOT/$O/asan_report PNP_LIB=$P/libpnp_hip_asan.so \
  timeout 300 python -X faulthandler -m pytest tests/test_gpu_conv.py -q -x -s -k "3 and 16" > $O/asan_gpu_s.log 2>&1; echo "rc=$?" >> $O/asan_gpu_s.log
tail -40 $O/asan_gpu_s.log | cut -c1-300; ls $O; head -60 $O/asan_report* 2>/dev/null | cut -c1-300
# the same without LD_PRELOAD-ing the runtime into python: only torch + HIP under ASAN's allocator is the suspect — a plain C driver
LD_PRELOAD=$ASAN_LIB ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 timeout 120 python -c "
import torch; x=torch.ones(4,device='cuda'); print('torch cuda under asan preload ok', float(x.sum()))" > $O/asan_torch_only.log 2>&1; echo "rc=$?" >> $O/asan_torch_only.log; tail -5 $O/asan_torch_only.log | cut -c1-300

# round 3, GPU call 9: per-stage shader-clock trace of the narrow (128x64, three-stage) and wide (128x128) forward kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i; mkdir -p $O
PNP_LIB=$GRAFT_REPO_ROOT/medical-cross-modality-domain-adaptation_amd/libpnp_hip_trace.so timeout 200 python tools/experiments/stage_trace.py > $O/stage_trace.txt 2>&1
grep -v amdgpu $O/stage_trace.txt

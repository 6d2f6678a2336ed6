cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=$GRAFT_REPO_ROOT/medical-cross-modality-domain-adaptation_amd
for v in d2 wab2 wab1 d2 wab2; do
L=$P/libpnp_hip.so; D=2
case $v in d1) D=1;; d3) D=3;; wab1) L=$P/libpnp_hip_wab1.so;; wab2) L=$P/libpnp_hip_wab2.so;; esac
ONLY=${ONLY:-512} PNP_WGRAD_DEPTH=$D PNP_LIB=$L python tools/bench_conv.py 2>/dev/null | grep -v "^layer\|totals" | awk "{print \$1,\$2,\$3,\"wgrad\",\$(NF-1),\$NF}" | sed "s/^/$v /"
done

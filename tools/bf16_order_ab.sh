run() { echo "== $1: $2"; shift; "$@" 2>&1 | grep -v amdgpu | tail -${TAILN:-3}; }
TAILN=1 run "K8 bf16x3 product" env NFA_K8_ENGINE=bf16x3 python tools/k8h_time.py 65536
TAILN=1 run "K8 bf16x3 reordered" env NFA_K8_ENGINE=bf16x3 NFLOWS_AMD_LIB=build_variants/bf16order_rqs_resnet.so python tools/k8h_time.py 65536
TAILN=12 run "K14 product" python tools/k14_micro.py 65536
TAILN=12 run "K14 reordered" env NFLOWS_AMD_LIB=build_variants/bf16order_resnet_train.so python tools/k14_micro.py 65536
TAILN=3 run "K13 product" python tools/cfg5_forward_probe.py
TAILN=3 run "K13 reordered" env NFLOWS_AMD_LIB=build_variants/bf16order_made_output.so python tools/cfg5_forward_probe.py
TAILN=2 run "K7b product" python tools/k7_micro.py
TAILN=2 run "K7b reordered" env NFLOWS_AMD_LIB=build_variants/bf16order_rqs_fused_linear.so python tools/k7_micro.py

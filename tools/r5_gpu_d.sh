#!/bin/bash
set -u
ROOTDIR=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOTDIR
for i in 1 2 3 4; do
PROBE_POISON=nan timeout 200 python tests/probes/k8h_fresh_flow_stress.py 8 act_tanh_k10 16384 0 2>&1 | grep -v amdgpu.ids | tail -5
done
echo "== no redo pass"
for i in 1 2 3; do
PROBE_POISON=nan NFA_K8H_NOREDO=1 timeout 200 python tests/probes/k8h_fresh_flow_stress.py 8 act_tanh_k10 16384 0 2>&1 | grep -v amdgpu.ids | tail -5
done
echo "== relu k8"
for i in 1 2 3; do
PROBE_POISON=nan timeout 200 python tests/probes/k8h_fresh_flow_stress.py 8 steep_nsf_k8 16384 0 2>&1 | grep -v amdgpu.ids | tail -5
done

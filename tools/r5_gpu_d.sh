#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}
timeout 300 python -m pytest tests/test_gpu_realnvp.py -q -m gpu -k "d80" 2>&1 | grep -v "^    \|^$" | tail -15 | cut -c1-300
grep "realnvp_d80" gpurun_out/parity_report.jsonl | tail -1 | cut -c1-500

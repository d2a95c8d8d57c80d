#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}
timeout 300 python -m pytest tests/test_gpu_context.py -q -m gpu -k "every_served" 2>&1 | grep -v "^    \|^$" | tail -15 | cut -c1-400

#!/bin/bash
# round 5, second GPU call: bin-index tests, the steep / bins / activations / trained files under the 65 536-row rules,
# the integration stub, the weight-stream probe, a bench line with the redo pass's early exit
set -u
ROOTDIR=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOTDIR/gpurun_out/r5b
mkdir -p $OUT
cd $ROOTDIR
export NFA_PARITY_LOG=$OUT/parity.jsonl
timeout 600 python -m pytest tests/test_gpu_bin_index.py tests/test_gpu_integration_stub.py -q --durations=15 > $OUT/bins.log 2>&1
tail -25 $OUT/bins.log
timeout 900 python -m pytest tests/test_gpu_steep.py tests/test_gpu_bins.py tests/test_gpu_activations.py tests/test_gpu_trained.py -q --durations=25 > $OUT/rules.log 2>&1
tail -60 $OUT/rules.log
timeout 120 python -m pytest tests/test_gpu_headline_parity.py -q -k "config5 or small_batch" --durations=5 > $OUT/cfg5.log 2>&1
tail -12 $OUT/cfg5.log
unset NFA_PARITY_LOG
timeout 120 tools/bin/stream_probe > $OUT/stream_probe.txt 2>&1
cat $OUT/stream_probe.txt
timeout 200 python bench.py --no-cpu-baseline --skip-mfma-ceiling > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
b=json.load(open("$OUT/bench.json"))
print(b["value"], b["ms_per_step"], b["small_shards_extra"], b["rows_65536_per_gpu_extra"])
PY

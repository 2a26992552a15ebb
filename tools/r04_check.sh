#!/bin/bash
# the tests that touch the many-small-alignments batch, then config 4's timings (tools/run_config4_share.py) and host phases
mkdir -p gpurun_out/r04/check
( while true; do date >> gpurun_out/r04/check/heartbeat; sleep 60; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 1000 python -u -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_fullsize.py tests/test_gpu_stress.py tests/test_gpu_parity.py -q -x -k "small or uniprot or config4 or switch or float16 or batch or view or packed" > gpurun_out/r04/check/tests.log 2>&1; tail -5 gpurun_out/r04/check/tests.log
grep -q "failed\|error" gpurun_out/r04/check/tests.log && exit 1
timeout -k 10 300 python tools/run_config4_share.py > gpurun_out/r04/check/c4.json 2> gpurun_out/r04/check/c4.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04/check/c4.json'))
for w in ('1','8'):
    r=d['rank_share']['worlds'][w]; print(w, {k:round(v,3) for k,v in r.items() if k.endswith('_ms') or k.startswith('pred')})
print(d['parity_check'])
PY
python tools/c4_share_trace.py 1 1 2>&1 | grep "mi355_sw\|world"; python tools/c4_share_trace.py 1 2>&1 | grep "mi355_sw\|world"

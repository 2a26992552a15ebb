#!/bin/bash
mkdir -p gpurun_out/r04/pool
( while true; do date >> gpurun_out/r04/pool/heartbeat; sleep 60; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 1000 python -u -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_parity.py -q -x -k "long or saved or config5 or split or single3000 or single5000 or switch or rank or shard" > gpurun_out/r04/pool/tests.log 2>&1; tail -3 gpurun_out/r04/pool/tests.log
grep -q "failed\|error" gpurun_out/r04/pool/tests.log && exit 1
python tools/run_rank_share.py > gpurun_out/r04/pool/c5_share.json 2> gpurun_out/r04/pool/c5_share.err
python - <<'PY'
import json
t=open('gpurun_out/r04/pool/c5_share.json').read()
d=json.loads(t[t.index('{'):])
w=d["rank_share"]["worlds"]
print({k:(round(v['share_ms'],2), round(v['score_kernel_ms'],2), round(v['finish_ms'],2), round(v['predicted_speedup'],2)) for k,v in w.items()})
PY

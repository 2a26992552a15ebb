#!/bin/bash
# the round's closing records on one MI355X: the whole GPU suite, the bench line at N = 1 (default flags), the two rank-path rehearsals
mkdir -p gpurun_out/r04/final
( while true; do date >> gpurun_out/r04/final/heartbeat; sleep 60; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 1000 python -u -m pytest tests -q -x -m gpu > gpurun_out/r04/final/gpu_tests.log 2>&1; tail -4 gpurun_out/r04/final/gpu_tests.log
grep -q "failed\|error" gpurun_out/r04/final/gpu_tests.log && exit 1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04/final/smoke.log 2>&1; tail -2 gpurun_out/r04/final/smoke.log
timeout -k 10 900 python bench.py > gpurun_out/r04/final/bench_n1.json 2> gpurun_out/r04/final/bench_n1.err || { tail -5 gpurun_out/r04/final/bench_n1.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/final/bench_n1.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('metric','value','unit','ms_per_step','roofline')})
PY
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-traffic > gpurun_out/r04/final/bench_1rank_nccl.json 2> gpurun_out/r04/final/bench_1rank_nccl.err; tail -c 300 gpurun_out/r04/final/bench_1rank_nccl.json
timeout -k 10 400 python bench.py --gpus 2 --dist-backend gloo --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-traffic > gpurun_out/r04/final/bench_2ranks_gloo.json 2> gpurun_out/r04/final/bench_2ranks_gloo.err; tail -c 300 gpurun_out/r04/final/bench_2ranks_gloo.json

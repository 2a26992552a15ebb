#!/bin/bash
# the round's closing records on one MI355X: the whole GPU suite, the bench line at N = 1 (default flags), the two rank-path rehearsals,
# and, with the argument `profiles`, kernel stats + PMC of config 4's passes
mkdir -p gpurun_out/r04/final
( while true; do date >> gpurun_out/r04/final/heartbeat; sleep 60; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 1000 python -u -m pytest tests -q -x -m gpu > gpurun_out/r04/final/gpu_tests.log 2>&1; tail -4 gpurun_out/r04/final/gpu_tests.log
grep -q "failed\|error" gpurun_out/r04/final/gpu_tests.log && exit 1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04/final/smoke.log 2>&1; tail -1 gpurun_out/r04/final/smoke.log
timeout -k 10 900 python bench.py > gpurun_out/r04/final/bench_n1.json 2> gpurun_out/r04/final/bench_n1.err || { tail -5 gpurun_out/r04/final/bench_n1.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/final/bench_n1.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'])
c4=d['extras']['config4_uniprot_shape']['rank_share']['worlds']
print({w:(round(r['score_argmax_ms'],3), round(r['with_traceback_ms'],3), round(r['predicted_speedup_score_argmax'],2), round(r['predicted_speedup'],2)) for w,r in c4.items()})
PY
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-traffic > gpurun_out/r04/final/bench_1rank_nccl.json 2> gpurun_out/r04/final/bench_1rank_nccl.err; tail -c 200 gpurun_out/r04/final/bench_1rank_nccl.json
timeout -k 10 400 python bench.py --gpus 2 --dist-backend gloo --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-traffic > gpurun_out/r04/final/bench_2ranks_gloo.json 2> gpurun_out/r04/final/bench_2ranks_gloo.err; tail -c 200 gpurun_out/r04/final/bench_2ranks_gloo.json
if [ "$1" = profiles ]; then
R=$(pwd); OUT=$R/gpurun_out/r04/final/prof; rm -rf $OUT; mkdir -p $OUT; PY=$(python3 -c "import os,sys;print(os.path.realpath(sys.executable))")
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt_config4 --output-format csv -- $PY $R/tools/c4_share_trace.py 1 > $OUT/config4_world1.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/kt_config4_w8 --output-format csv -- $PY $R/tools/c4_share_trace.py 8 > $OUT/config4_world8.log 2>&1
cd $R
tools/pmc_run.sh config4 'sw_wave_prof16_kernel|sw_wave_prof_kernel<9, false, true>' python3 $GRAFT_REPO_ROOT/tools/c4_share_trace.py 1 > gpurun_out/r04/final/pmc_config4.log 2>&1
tail -6 gpurun_out/r04/final/pmc_config4.log
fi

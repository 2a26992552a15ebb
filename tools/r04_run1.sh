set -x
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py -x -q -k "long_kernel or scored_range or best_range or optimistic" > gpurun_out/r04/t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/t1.log
tail -5 gpurun_out/r04/t1.log
timeout -k 10 400 python tools/run_rank_share.py > gpurun_out/r04/c5_first.json 2> gpurun_out/r04/c5_first.err; echo "c5 rc=$?"
tail -3 gpurun_out/r04/c5_first.err
timeout -k 10 300 python tools/run_config4_share.py > gpurun_out/r04/c4_first.json 2> gpurun_out/r04/c4_first.err; echo "c4 rc=$?"
tail -3 gpurun_out/r04/c4_first.err

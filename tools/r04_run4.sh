set -x
mkdir -p gpurun_out/r04
timeout -k 10 300 python tools/c5_ab.py 250000000 10000 ";no_long_save=1;long_save_what=1;long_save_what=2" > gpurun_out/r04/c5_ab3.json 2> gpurun_out/r04/c5_ab3.err; echo "c5ab rc=$?"
cat gpurun_out/r04/c5_ab3.json
timeout -k 10 300 python tools/run_config4_share.py > gpurun_out/r04/c4_fourth.json 2> gpurun_out/r04/c4_fourth.err; echo "c4 rc=$?"
timeout -k 10 200 python tools/c4_share_trace.py 8 > gpurun_out/r04/c4_trace8b.log 2>&1; echo "c4trace rc=$?"
tail -9 gpurun_out/r04/c4_trace8b.log
timeout -k 10 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -x -q -k "not every_switch" > gpurun_out/r04/t4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/t4.log
tail -15 gpurun_out/r04/t4.log
timeout -k 10 300 python - > gpurun_out/r04/readlens2.json 2> gpurun_out/r04/readlens2.err <<'PYEOF'
import json, sys, os
sys.path.insert(0, os.getcwd())
import bench
pgs = bench.load_package()
print(json.dumps(bench.extra_read_lengths(pgs, 0, 50_000_000), indent=1))
print(json.dumps(bench.extra_latency(pgs, 0), indent=1))
PYEOF
echo "readlens rc=$?"

set -x
mkdir -p gpurun_out/r04
PY=$(python3 -c "import os,sys;print(os.path.realpath(sys.executable))")
timeout -k 10 300 python tools/c5_ab.py > gpurun_out/r04/c5_ab2.json 2> gpurun_out/r04/c5_ab2.err; echo "c5ab rc=$?"
cat gpurun_out/r04/c5_ab2.json
timeout -k 10 300 python tools/run_config4_share.py > gpurun_out/r04/c4_third.json 2> gpurun_out/r04/c4_third.err; echo "c4 rc=$?"
timeout -k 10 200 python tools/c4_share_trace.py 8 > gpurun_out/r04/c4_trace8.log 2>&1; echo "c4trace rc=$?"
timeout -k 10 200 python tools/c4_share_trace.py 8 1 > gpurun_out/r04/c4_trace8_score.log 2>&1; echo "c4trace rc=$?"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04/kt_c4_w8 --output-format csv -- $PY $GRAFT_REPO_ROOT/tools/c4_share_trace.py 8 > $GRAFT_REPO_ROOT/gpurun_out/r04/kt_c4_w8.log 2>&1); echo "kt rc=$?"
timeout -k 10 900 python -m pytest tests/test_gpu_round4.py -x -q > gpurun_out/r04/t3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/t3.log
tail -15 gpurun_out/r04/t3.log
timeout -k 10 300 python - > gpurun_out/r04/readlens.json 2> gpurun_out/r04/readlens.err <<'PYEOF'
import json, sys, os
sys.path.insert(0, os.getcwd())
import bench
pgs = bench.load_package()
print(json.dumps(bench.extra_read_lengths(pgs, 0, 50_000_000), indent=1))
PYEOF
echo "readlens rc=$?"

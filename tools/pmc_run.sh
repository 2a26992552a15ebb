#!/usr/bin/env bash
# PMC counters of one command, summarised per kernel: tools/pmc_run.sh TAG 'KSUB1|KSUB2|...' PROGRAM ARGS...
# (PROGRAM = python3 or a binary: rocprofv3 gets it after `--`, never a shell).  Separate passes per counter set, no trace flags;
# one summary per kernel-name substring -> gpurun_out/pmc/TAG.<n>.json (n = position of the substring)
TAG=$1; KSUBS=$2; shift 2
R=$GRAFT_REPO_ROOT
# a bare `python3` may resolve to a shim or a `#!/usr/bin/env` launcher, i.e. an exec hop AFTER the profiler's preloaded library
# has initialised the GPU (forbidden on this pool): hand rocprofv3 the real interpreter binary
if [ "$1" = python3 ] || [ "$1" = python ]; then
  PY=$("$1" -c 'import os,sys;print(os.path.realpath(sys.executable))')
  shift
  set -- "$PY" "$@"
fi
OUT=$R/gpurun_out/pmc/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
n=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES"; do
  n=$((n+1))
  rocprofv3 --pmc $set -d $OUT/p$n --output-format csv -- "$@" > $OUT/p$n.log 2>&1 || echo "pass $n failed"
done
cd $R
k=0
IFS='|' read -ra SUBS <<< "$KSUBS"
for sub in "${SUBS[@]}"; do
  k=$((k+1))
  python tools/pmc_summary.py --kernel "$sub" $OUT/p1 $OUT/p2 > gpurun_out/pmc/$TAG.$k.json
  echo "== $TAG.$k: $sub"; python - "$R/gpurun_out/pmc/$TAG.$k.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(d["kernel"]); print(json.dumps(d["derived"]))
PY
done
rm -rf $OUT/p1 $OUT/p2

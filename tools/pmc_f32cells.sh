#!/usr/bin/env bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_f32cells
mkdir -p $OUT
n=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
  n=$((n+1))
  rocprofv3 --pmc $set -d $OUT/p$n --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --match 3.5 --mismatch -3.25 --no-cpu-baseline --no-extras --no-strong > /dev/null 2> $OUT/p$n.err
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT/p1 $OUT/p2

#!/usr/bin/env bash
# PMC counters (separate passes, kernel-trace only) of the 64-lane-tile score instances at 600 bp: packed int16 vs packed float16
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_wide
mkdir -p $OUT
for variant in i16 f16; do
  if [ $variant = f16 ]; then export MI355_SW_F16_WIDE=1; else unset MI355_SW_F16_WIDE; fi
  n=0
  for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES"; do
    n=$((n+1))
    rocprofv3 --pmc $set -d $OUT/${variant}_$n --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --reads 512 --read-len 600 --no-extras --no-cpu-baseline --no-strong > $OUT/${variant}_$n.log 2>&1 || echo "pass failed $variant $n"
  done
done
cd $GRAFT_REPO_ROOT
for variant in i16 f16; do python tools/pmc_summary.py gpurun_out/pmc_wide/${variant}_1 gpurun_out/pmc_wide/${variant}_2 > gpurun_out/pmc_wide/${variant}.json; done

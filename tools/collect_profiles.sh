#!/usr/bin/env bash
# Collects the round's measurement records on the GPU box into gpurun_out/profiles_r02/ (copied to profiles/ afterwards):
# bench line, rocprofv3 kernel stats of the same command, PMC passes (separate runs, kernel-trace only), configs 4 / 5,
# one-by-one call kernels.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-strong"
rocprofv3 --kernel-trace --stats -d $OUT/kt_bench --output-format csv -- $B > $OUT/kt_bench.json 2> $OUT/kt_bench.err
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set -d $OUT/pmc_bench_$tag --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-strong > /dev/null 2> $OUT/pmc_bench_$tag.err
done
# the 64-lane-tile float16 instance (600 bp reads) after the ds_read_b128 fix
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set -d $OUT/pmc_wide600_$tag --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --reads 512 --read-len 600 --no-cpu-baseline --no-extras --no-strong > /dev/null 2> $OUT/pmc_wide600_$tag.err
  rocprofv3 --pmc $set -d $OUT/pmc_u8_2048_$tag --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --reads 256 --read-len 2048 --semantics u8 --no-cpu-baseline --no-extras --no-strong > /dev/null 2> $OUT/pmc_u8_2048_$tag.err
  rocprofv3 --pmc $set -d $OUT/pmc_f32_2048_$tag --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --reads 256 --read-len 2048 --semantics f32 --no-cpu-baseline --no-extras --no-strong > /dev/null 2> $OUT/pmc_f32_2048_$tag.err
done
rocprofv3 --kernel-trace --stats -d $OUT/kt_config4 --output-format csv -- python3 $R/tools/run_config4.py > $OUT/config4.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/kt_config5 --output-format csv -- python3 $R/tools/run_config5.py > $OUT/config5.log 2>&1
for cfg in "1000000 0" "1000000 1" "50000000 0" "50000000 1"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats -d $OUT/kt_lat_$1_$2 --output-format csv -- python3 $R/tools/lat_probe.py $1 $2 > $OUT/lat_$1_$2.log 2>&1
done
python3 $R/tools/lone_probe.py > $OUT/lone_reads_by_length.log 2>&1
python3 $R/tools/lone_short.py > $OUT/lone_short_reads.log 2>&1
cd $R
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
for t in bench_FETCH_SIZE:bench_WRITE_SIZE:bench_SQ_INSTS_VALU:bench_SQ_INSTS_LDS wide600_SQ_INSTS_VALU:wide600_SQ_INSTS_LDS u8_2048_SQ_INSTS_VALU:u8_2048_SQ_INSTS_LDS f32_2048_SQ_INSTS_VALU:f32_2048_SQ_INSTS_LDS; do
  dirs=$(echo $t | tr ':' '\n' | sed "s#^#$OUT/pmc_#" | tr '\n' ' ')
  name=$(echo $t | cut -d: -f1 | sed 's/_SQ_INSTS_VALU//; s/_FETCH_SIZE//')
  python tools/pmc_summary.py $dirs > $OUT/pmc_$name.json
done
ls $OUT

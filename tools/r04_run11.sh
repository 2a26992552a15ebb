#!/bin/bash
# the float16 pass under randomised load: the collected stress tests, then a longer run of the small-batch generator with new seeds;
# PMC of config 4's two passes
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_stress.py tests/test_gpu_round3.py -q -x -k "stress or small_alignment" 2>&1 | tail -8 > gpurun_out/r04/t11.log; cat gpurun_out/r04/t11.log
grep -q "failed" gpurun_out/r04/t11.log && exit 1
for seed in 1001 2002 3003; do
  timeout -k 10 200 python tests/stress_small_batches.py 100 $seed 2>&1 | tail -4 > gpurun_out/r04/stress_small_$seed.log; cat gpurun_out/r04/stress_small_$seed.log
done
tools/pmc_run.sh config4 'sw_wave_prof16_kernel|sw_wave_prof_kernel<9, false, true>' python3 $GRAFT_REPO_ROOT/tools/c4_share_trace.py 1 > gpurun_out/r04/pmc_config4.log 2>&1
tail -12 gpurun_out/r04/pmc_config4.log

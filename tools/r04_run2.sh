set -x
mkdir -p gpurun_out/r04
timeout -k 10 300 python tools/c5_ab.py > gpurun_out/r04/c5_ab.json 2> gpurun_out/r04/c5_ab.err; echo "c5ab rc=$?"
cat gpurun_out/r04/c5_ab.json
timeout -k 10 300 python tools/run_config4_share.py > gpurun_out/r04/c4_second.json 2> gpurun_out/r04/c4_second.err; echo "c4 rc=$?"
timeout -k 10 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -x -q -k "small_alignment or config4 or uniprot or packed or split or reference_gtest or dropin or multi or long_kernel or scored_range or best_range or optimistic or config5" > gpurun_out/r04/t2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/t2.log
tail -5 gpurun_out/r04/t2.log
timeout -k 10 420 python tools/path_probe.py > gpurun_out/r04/paths.jsonl 2> gpurun_out/r04/paths.err; echo "paths rc=$?"
tail -3 gpurun_out/r04/paths.err

set -x
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_stress.py tests/test_gpu_round4.py -q -k "stress or no_u8_early or assumed" > gpurun_out/r04/t7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/t7.log
tail -12 gpurun_out/r04/t7.log
timeout -k 10 560 python bench.py > gpurun_out/r04/bench_n1.json 2> gpurun_out/r04/bench_n1.err; echo "bench rc=$?"
tail -3 gpurun_out/r04/bench_n1.err

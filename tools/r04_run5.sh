set -x
mkdir -p gpurun_out/r04
timeout -k 10 300 python tools/c5_ab.py 250000000 10000 ";no_long_save=1" > gpurun_out/r04/c5_ab4.json 2> gpurun_out/r04/c5_ab4.err; echo "c5ab rc=$?"
cat gpurun_out/r04/c5_ab4.json
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_stress.py > gpurun_out/r04/t5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/t5.log
tail -15 gpurun_out/r04/t5.log

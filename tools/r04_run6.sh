set -x
mkdir -p gpurun_out/r04
timeout -k 10 1100 python -m pytest tests -m gpu -q --deselect tests/test_gpu_stress.py > gpurun_out/r04/t6.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/t6.log
tail -25 gpurun_out/r04/t6.log

#!/bin/bash
mkdir -p gpurun_out/r04
for i in 1 2 3; do python tools/lat_probe.py 50000000 0 2>&1 | tail -1; done > gpurun_out/r04/lat_f32.log
for i in 1 2; do python tools/lat_probe.py 50000000 1 2>&1 | tail -1; done > gpurun_out/r04/lat_u8.log
MI355_SW_TRACE=1 python tools/lat_probe.py 50000000 0 2>&1 | tail -30 > gpurun_out/r04/lat_f32_trace.log
cat gpurun_out/r04/lat_f32.log gpurun_out/r04/lat_u8.log; tail -22 gpurun_out/r04/lat_f32_trace.log

#!/usr/bin/env bash
# whole-job rate of bench.py's workload at other read lengths (the 64-lane-tile instances), both engines;
# A/B: per-step maximum, no saturating sweep, packed integer cells
for len in 300 600 1000 2048; do
  for sem in f32 u8; do
    for env in "" "MI355_SW_NO_SAMPLE=1" "MI355_SW_NO_SAMPLE=1 MI355_SW_NO_SATFLAG=1" "MI355_SW_NO_F16=1"; do
      echo -n "len=$len sem=$sem $env: "
      env $env python bench.py --steps 2 --warmup 1 --reads 512 --read-len $len --semantics $sem --no-extras --no-cpu-baseline --no-strong 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['roofline']['kernel'], round(d['valu']['kernel_gcups']))"
    done
  done
done

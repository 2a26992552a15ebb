#!/usr/bin/env bash
for cl in 0 64 128 192 256 384 512 1024; do
  if [ "$cl" = 0 ]; then unset MI355_SW_CHUNK; else export MI355_SW_CHUNK=$cl; fi
  echo -n "chunk=$cl 1Mbp f32: "; python tools/lat_probe.py 1000000 0 2>&1 | tail -1
done
unset MI355_SW_CHUNK
echo -n "no twin: "; MI355_SW_NO_TWIN=1 python tools/lat_probe.py 1000000 0 2>&1 | tail -1

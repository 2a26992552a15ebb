#!/usr/bin/env bash
# one-by-one call latency (tools/lat_probe.py) under a few tile lengths of the score pass (MI355_SW_CHUNK)
for cl in 0 1024 1536 2048 3072 4096 8192; do
  for cfg in "50000000 0" "1000000 0"; do
    if [ "$cl" = 0 ]; then unset MI355_SW_CHUNK; else export MI355_SW_CHUNK=$cl; fi
    echo -n "chunk=$cl $cfg: "; python tools/lat_probe.py $cfg 2>&1 | tail -1
  done
done
unset MI355_SW_CHUNK
echo "no twin:"; MI355_SW_NO_TWIN=1 python tools/lat_probe.py 50000000 0 2>&1 | tail -1

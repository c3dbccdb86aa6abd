#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
for g in "4 1" "4 0" "1 1" "8 1"; do echo "== group $g"; timeout 300 python tools/r06_region_timeline.py $g; done > gpurun_out/r06_region_timeline.log 2>&1
tail -40 gpurun_out/r06_region_timeline.log

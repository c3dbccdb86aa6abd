#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
(timeout 900 python tools/r06_soak.py 50000 60000 32; timeout 900 python tools/r06_soak.py 1000000 6000 32; timeout 600 python tools/r06_soak.py 12000 60000 1) > gpurun_out/r06y_soak.log 2>&1
grep -v amdgpu gpurun_out/r06y_soak.log

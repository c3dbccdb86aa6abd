#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
(timeout 600 python tools/r06_group_sweep.py 50000; timeout 300 python tools/r06_group_sweep.py 12000 1 8 32; timeout 200 python -m pytest tests/test_pipelined.py -q -m gpu -k "group" 2>&1 | tail -3) > gpurun_out/r06t_group_sweep.log 2>&1
tail -30 gpurun_out/r06t_group_sweep.log

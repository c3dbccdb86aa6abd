#!/bin/bash
# round 3, session 5: smoke() as the driver runs it + the new read_many test on the device
set -u
O=gpurun_out/r03zi
mkdir -p $O
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | grep -v amdgpu.ids | tail -6
timeout 300 python -m pytest tests/test_parity_vfo.py -m gpu -x -q -k "read_many or four_wavefront" 2>&1 | tail -1

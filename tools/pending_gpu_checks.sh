#!/bin/bash
# Everything round 1 left unmeasured on the device, in ONE gpurun call (about 90 s of box time):
#   /usr/local/graft/bin/gpurun --timeout 300 -- 'bash tools/pending_gpu_checks.sh'
# 1. device leg of the opt-in variant parity test (remove the skip in tests/test_parity_vfo.py::test_opt_in_kernel_variants_bit_identical
#    once it is green); 2. A/B of the variants on the headline bench; 3. per-family times of the VFO bank alone.
set -u
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["kernel_ms_per_step"])'
echo "== variant parity on the device (the test itself still skips its gpu leg: run the scenario directly)"
for sw in "SDRPP_GPU_FUSE=3" "SDRPP_GPU_WIDE_STORE=1"; do
    env ${sw%%=*}=0 python tests/variant_scenario.py /tmp/v0.npz 18 && env $sw python tests/variant_scenario.py /tmp/v1.npz 18 && python - <<PY
import numpy as np
a, b = np.load("/tmp/v0.npz"), np.load("/tmp/v1.npz")
print("$sw", "bit-identical" if all(np.array_equal(a[k], b[k]) for k in a.files) else "DIFFERENT")
PY
done
echo "== headline bench, default / wide stores / fused launches"
for sw in "SDRPP_GPU_FUSE=0" "SDRPP_GPU_WIDE_STORE=1" "SDRPP_GPU_FUSE=3" "SDRPP_GPU_FUSE=0"; do
    echo "-- $sw"
    env $sw python bench.py --no-cpu-baseline --steps 30 2>&1 | tail -1 | python -c "$P"
done
echo "== VFO bank alone"
for sw in "SDRPP_GPU_FUSE=0" "SDRPP_GPU_WIDE_STORE=1"; do
    echo "-- $sw"
    env $sw python tools/vfo_only_time.py 16777216 32 10 2>&1 | tail -1
done
echo "== knock-out builds of the Toeplitz kernel (1: no stores, 2: no loads, 4: no matrix loop; DESIGN.md section 4)"
make -C sdrplusplus_amd/csrc -s knock
for m in 0 1 2 3 4 7; do
    echo "-- mask $m"
    SDRPP_TOEP_KNOCK=$m SDRPP_GPU_LIB=$PWD/sdrplusplus_amd/csrc/libsdrpp_gpu_knock.so python tools/vfo_only_time.py 16777216 32 10 2>&1 | tail -1
done

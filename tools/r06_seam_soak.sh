#!/bin/bash
# round 6: soak at the C++ seam — N blocks through IQFrontEnd::run, every byte of every audio stream hashed by the sinks; one block per launch against
# host-side launch groups of 2 / 8 / 32 (group sizes follow the threads' timing): same frames, same digest
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out
out=gpurun_out/r06y_seam_soak.log
A="sdrplusplus_amd/data/decim_plans.bin 10000000 50000 65536 32 2 0 1"
{
for g in 1 2 8 32 8 1; do
  SDRPP_BENCH_BLOCKS=${1:-40000} timeout 300 oracle/_ref/bench_blocks_ref $A $g 2>&1 | grep -v amdgpu | sed 's/"buffered.*"msps"/"msps"/; s/"audio_frames_per_s.*"soak_blocks"/"soak_blocks"/' | sed "s/^/k=$g  /"
done
} > $out 2>&1
cat $out
python - <<'PY'
import re
d = set(re.findall(r'"soak_audio_frames": (\d+), "soak_digest": "([0-9a-f]+)"', open("gpurun_out/r06y_seam_soak.log").read()))
print("distinct (frames, digest) pairs:", d)
raise SystemExit(0 if len(d) == 1 else 1)
PY

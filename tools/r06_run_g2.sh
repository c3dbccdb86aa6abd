#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
B="python bench.py --no-others --no-by-push --no-cpu-baseline --no-self-check --regions 3"
pick() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[1], 'value', d['value'], 'steady', (d.get('steady_state') or {}).get('value'), 'blocks/launch', d.get('blocks_per_launch'))
" "$1"; }
(
for g in 4 8 16; do timeout 300 $B --cfg 2 --group $g --steps 200 --warmup 10 2>/dev/null | pick "cfg2 group $g steps 200"; done
for g in 4 8; do timeout 300 $B --cfg 4 --group $g --steps 100 --warmup 10 2>/dev/null | pick "cfg4 group $g steps 100"; done
for g in 4 6 8; do timeout 300 $B --group $g --steps 20 --warmup 5 --regions 5 2>/dev/null | pick "cfg3 group $g steps 20"; done
) > gpurun_out/r06t_group_cfgs.log 2>&1
cat gpurun_out/r06t_group_cfgs.log

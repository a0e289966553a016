#!/bin/bash
# Round 4, final sanity of HEAD on the GPU: the batch / lean / ring / device-resident tests (api.hip changed after the last full suite) and smoke.
OUT=gpurun_out/c10
mkdir -p $OUT
timeout 240 python -m pytest tests/test_gpu_parity.py -x -q -k "lean or batch_api or rings or device_resident or calibrated or block_parity" 2>&1 | tail -3 | tee $OUT/pytest_final.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.log
timeout 120 python bench.py --blocks 768 --block-mib 4 --steps 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('768 x 4 MiB:', d['value'], 'MiB/s, steps', d['step_s'], 'given up', d['config']['cm_blocks_given_up'], 'ring', d['stages']['front_end_ring'])" | tee $OUT/bench_small.txt

#!/bin/bash
# Round 6, call 10: the library's DEFAULT memory mode at full size (no kept workspace: what a host program gets that does not call bz3_hip_set_keep_workspace) with the round's
# tail ring (windows of 30: 120 swap buffers out of the pool, if the device has the room beside the headroom), one step, the rings' accounting.
OUT=gpurun_out/c10
mkdir -p $OUT
BZ3_BENCH_KEEP_WS=0 BZ3_HIP_TRACE_RINGS=1 timeout 800 python3 bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline > $OUT/full_nokeep.json 2> $OUT/full_nokeep.log
echo "rc=$?"
grep "^\[bench\|bz3 rings\|bzip3_amd" $OUT/full_nokeep.log | grep -v " 1 blocks" | tee $OUT/full_nokeep.progress.txt
python3 -c "import json;d=json.loads(open('$OUT/full_nokeep.json').read().strip().splitlines()[-1]);print('value',d.get('value'),'error',d.get('error'),'keep',d['config'].get('keep_workspace'),json.dumps(d.get('stages',{}).get('front_end_ring')))"

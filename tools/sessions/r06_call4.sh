#!/bin/bash
# Round 6, call 4: (1) the whole -m gpu suite on the current tree; (2) the decoder's tail at FULL size with 64 reserved CUs and windows of 20 (the 768 x 64 MiB probe of call 3
# cannot show the waiting that 256 MiB blocks have) against the default on the same box, the rings' own accounting; (3) SURVEY 8d's timing boundary at the metric's block
# size: 160 x 256 MiB through bz3_encode_blocks / bz3_decode_blocks on host buffers.
OUT=gpurun_out/c4
mkdir -p $OUT
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.log
run_full() {  # tag, env...
  tag=$1; shift
  env "$@" BZ3_HIP_TRACE_RINGS=1 timeout 700 python3 bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline > $OUT/full_$tag.json 2> $OUT/full_$tag.log
  grep "^\[bench\|bz3 rings" $OUT/full_$tag.log | grep -v " 1 blocks" | tee $OUT/full_$tag.progress.txt
}
run_full r64_w20 BZ3_HIP_CU_RESERVE=64 BZ3_HIP_TAIL_PIPE=20,4
run_full default BZ3_BENCH_DUMMY=1
timeout 1200 python3 bench.py --blocks 160 --block-mib 256 --steps 1 --warmup 0 --no-cpu-baseline --legs host_api --host-api-block-mib 256 > $OUT/host_api_256.json 2> $OUT/host_api_256.log
grep "^\[bench" $OUT/host_api_256.log | tee $OUT/host_api_256.progress.txt
python3 -c "
import json
d=json.loads(open('gpurun_out/c4/host_api_256.json').read().strip().splitlines()[-1]); print(json.dumps(d['configs'].get('host_api')))"

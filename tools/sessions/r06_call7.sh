#!/bin/bash
# Round 6, call 7: the tail ring further out -- call 6 measured 80 reserved CUs / windows of 25 at 16.9 s (64 / 20: 18.3 s) with the inverse BWTs no slower on fewer CUs
# (they are bound by HBM line fetches, not by CUs).  96 / 30 and 128 / 32, one full-size step each, the rings' own accounting.
OUT=gpurun_out/c7
mkdir -p $OUT
for cfg in "96 30,4" "128 32,4"; do
  set -- $cfg
  BZ3_HIP_CU_RESERVE=$1 BZ3_HIP_TAIL_PIPE=$2 BZ3_HIP_TRACE_RINGS=1 timeout 700 python3 bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline > $OUT/full_r$1.json 2> $OUT/full_r$1.log
  echo "== reserve $1 window,slots $2" | tee -a $OUT/summary.txt
  grep "^\[bench\|bz3 rings" $OUT/full_r$1.log | grep -v " 1 blocks" | tee -a $OUT/summary.txt
done

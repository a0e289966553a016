#!/bin/bash
# Round 6, call 11 (measurement only, no code change): the front end under a busy host (tools/host_contention.py, VERDICT r05 item 3's third criterion), 96 x 64 MiB blocks,
# idle against 64 page-faulting threads of another process -- with the default front end and with the two-thread form (BZ3_HIP_FRONT_DUO=1).
OUT=gpurun_out/c11
mkdir -p $OUT
for duo in 0 1; do
  echo "== BZ3_HIP_FRONT_DUO=$duo" | tee -a $OUT/host_contention.txt
  BZ3_HIP_FRONT_DUO=$duo timeout 600 python3 tools/host_contention.py 64 96 --only=idle,mmap_subproc 2>&1 | tail -30 | tee -a $OUT/host_contention.txt
done

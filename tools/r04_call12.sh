#!/bin/bash
# Round 4, call 12 (the round's last GPU seconds): the trio CM encoder -- three blocks per workgroup sharing one coder wave -- against the
# three-workgroups-per-CU encoder it would replace: one launch over 768 copies of a 2 MiB / 8 MiB block (every copy's bytes compared), then
# different blocks in one batch against the CPU checker.
mkdir -p gpurun_out/c12
{
echo "== 768 x 2 MiB: rows3 as three workgroups per CU (trio 0) and as the trio kernel (trio 1); all copies compared"
timeout 60 python tools/cm_encode_split.py 2 768 --only=rows3 --trio=0,1 --check
echo "== 768 x 8 MiB"
timeout 90 python tools/cm_encode_split.py 8 768 --only=rows3 --trio=0,1 --check
echo "== different blocks in one batch against the CPU checker"
timeout 120 python tools/trio_parity_gpu.py 2
} > gpurun_out/c12/trio.txt 2>&1
tail -20 gpurun_out/c12/trio.txt

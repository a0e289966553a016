#!/bin/bash
# Round 4, last GPU minutes: the GPU tests added after call 7, and the front end of LARGE blocks under a busy host (96 x 64 MiB: idle, then
# 64 page-faulting threads of another process).
OUT=gpurun_out/c9
mkdir -p $OUT
timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -k "calibrated or suffix_sorter" 2>&1 | tail -3 | tee $OUT/pytest_new.log
timeout 330 python tools/host_contention.py 64 96 --only=idle,mmap_subproc 2>/dev/null | tee $OUT/host_contention_64MiB.json

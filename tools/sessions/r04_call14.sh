#!/bin/bash
# Round 4, call 14: the trio encoder once more with warm kernels (call 12's figures were each kernel's first launch): library built from commit 4a83c54.
mkdir -p gpurun_out/c14
timeout 50 python tools/cm_encode_split.py 2 768 768 --only=rows3 --trio=0,1 --check --lib=bzip3_amd/lib/ab/libbzip3_trio.so 2>&1 | grep -v amdgpu.ids > gpurun_out/c14/trio_warm.txt
cat gpurun_out/c14/trio_warm.txt

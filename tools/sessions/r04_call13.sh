#!/bin/bash
# Round 4, call 13: shapes of the decoder's tail pipeline (windows x slots of decode_group) on a batch of 768 x 8 MiB blocks.
mkdir -p gpurun_out/c13
timeout 110 python tools/tail_pipe_probe.py 8 768 --settings=default,8x8,4x8 --trials=2 > gpurun_out/c13/tail_pipe.txt 2>&1
grep -v amdgpu.ids gpurun_out/c13/tail_pipe.txt | tail -12

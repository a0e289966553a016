#!/bin/bash
# Round 4, GPU call 7: the whole GPU suite + smoke on HEAD, the stream driver beside the reference CLI (N1's timing evidence), the front end under host load again.
OUT=gpurun_out/c7
mkdir -p $OUT
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
echo "== stream driver vs reference CLI (2 GiB file, -b 32, 768 blocks per batch, reference -j 64)"
timeout 600 python tools/stream_time.py 2048 32 768 64 2>&1 | tail -1 | tee $OUT/stream_driver_vs_cli.json
echo "== front end under host load (after round 4's changes)"
timeout 300 python tools/host_contention.py 8 768 2>/dev/null | tee $OUT/host_contention.json

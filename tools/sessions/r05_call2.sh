#!/bin/bash
# Round 5, GPU call 2:
#  (1) parity of the round's sorter / LZP changes on the GPU (tests/test_sorter.py, the stage-parity and sorter-path tests);
#  (2) one 256 MiB block of the calibrated text through the stage hooks under rocprofv3 --kernel-trace, BZ3_LZP_LINKS = 0 (round 4's link build),
#      1 (two 9-bit passes) and 2 (+ links through position bins): per-kernel times;
#  (3) the decoder's tail (tools/tail_pipe_probe.py 64 256: 256 x 64 MiB, settings default = 16 x 4 and 8 x 8) with the HIP runtime's default
#      number of hardware queues and with GPU_MAX_HW_QUEUES=16: do the side streams of the rings share a hardware queue with the group's stream?
OUT=$(realpath -m "$1"); REPO=$(cd "$(dirname "$0")/../.." && pwd); mkdir -p "$OUT"; cd "$REPO"
echo "== (1) parity"
timeout 600 python -m pytest tests/test_sorter.py tests/test_gpu_parity.py -m gpu -x -q -k "sorter or stage_parity or block_parity or random_mixtures" 2>&1 | tail -4
echo "== (2) LZP link build forms"
cd /tmp && export TMPDIR=/tmp
for f in 0 1 2; do
  rm -rf "$OUT/p$f"
  BZ3_LZP_LINKS=$f timeout 300 rocprofv3 --kernel-trace -d "$OUT/p$f" -o pass -- python "$REPO/tools/stage_probe.py" 256 --noise=0.035 > "$OUT/probe_links$f.log" 2>&1
  grep "MiB rep" "$OUT/probe_links$f.log"
  db=$(find "$OUT/p$f" -name "*.db" | head -1)
  [ -n "$db" ] && python "$REPO/tools/rocpd_summary.py" "$db" "BZ3_LZP_LINKS=$f rocprofv3 --kernel-trace -- python tools/stage_probe.py 256 --noise=0.035   (MI355X, ROCm 7.2; both repetitions)" > "$OUT/stage_kernels_links$f.txt"
  rm -rf "$OUT/p$f"
  grep "k_lzp\|k_rs_\|k_scan" "$OUT/stage_kernels_links$f.txt" | cut -c1-60,100-170
done
cd "$REPO"
echo "== (3) tail, hardware queues"
timeout 400 python tools/tail_pipe_probe.py 64 256 --settings=default,8x8 --trials=1 2>&1 | grep "^{" | tee "$OUT/tail_probe_default_queues.txt"
GPU_MAX_HW_QUEUES=16 timeout 400 python tools/tail_pipe_probe.py 64 256 --settings=default,8x8 --trials=1 2>&1 | grep "^{" | tee "$OUT/tail_probe_16_queues.txt"

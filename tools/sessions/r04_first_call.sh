#!/bin/bash
# Suggested first GPU call of round 4 (left by round 3, whose budget ended before these could be measured):
#  (1) the LDS-staged encoder sink (tools/patches/cm_sink_lds_staging.patch): GPU parity of the CM tests, then the PMC passes
#      (WRITE_SIZE per coded byte of k_cm_encode_rows3 was 3.2 with byte-granular stores);
#  (2) the encoder at one / two / three blocks per CU (why does three per CU cost 1.3x with only six waves on the CU?);
#  (3) the tail ring at full block size with eight slots of 8 instead of four of 16 (128 blocks: BZ3_HIP_TAIL_PIPE=8,8).
# Usage: tools/r04_first_call.sh <outdir>       (~6 GPU-minutes)
OUT=$(realpath -m "$1")
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd "$REPO"
echo "== (2) encoder co-residency, HEAD"
timeout 200 python tools/cm_encode_split.py 8 256 512 768 2>&1 | grep encoder | tee "$OUT/enc_split_head.txt"
echo "== (1) LDS-staged sink"
git apply tools/patches/cm_sink_lds_staging.patch 2>/dev/null || patch -p1 < tools/patches/cm_sink_lds_staging.patch
python bzip3_amd/build.py | tail -1
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "stage_parity or cm_row_cache or three_blocks or lean_states or golden or block_parity" 2>&1 | tail -2 | tee "$OUT/parity_sink.txt"
timeout 200 python tools/cm_encode_split.py 8 768 2>&1 | grep encoder | tee "$OUT/enc_split_sink.txt"
timeout 300 bash tools/pmc_pass.sh "$OUT/pmc_sink" 2>&1 | tail -3
echo "== (3) tail ring 8 x 8 at 128 x 256 MiB"
BZ3_HIP_TAIL_PIPE=8,8 timeout 400 python bench.py --blocks 128 --block-mib 256 --lean 1 --steps 1 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tail 8x8:', d['stages']['t_dec_s'] - d['stages']['dec']['cm'] / 1e3, 's of', d['stages']['t_dec_s'])"

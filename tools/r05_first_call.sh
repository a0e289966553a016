#!/bin/bash
# Suggested first GPU call of round 5 (left by round 4, whose budget ended before these could be measured):
#  (1) `bench.py --leg cfg5 --blocks 256`: a batch of 511 MiB blocks of the 16-symbol source, round trip (BASELINE.json configs[4]; ~8 GPU-minutes:
#      a 511 MiB block's CM launches last ~2 + ~4.5 minutes one block per CU);
#  (2) the front end of FULL-size blocks beside a busy host (64 x 256 MiB: idle, then 64 page-faulting threads of another process) -- round 4 measured
#      8 MiB and 64 MiB blocks only (profiles/r04_host_contention.json: still 2-3x);
#  (3) LDS counters of the shipped decoder at three blocks per CU after the aligned halves (bank conflicts were a third of the LDS time in the
#      experimental build, profiles/r04_cm_decoder_experiments.txt), and FETCH_SIZE of k_bwt_tail / k_lzp_links at 256 MiB.
# Usage: tools/r05_first_call.sh <outdir under gpurun_out>
OUT=$(realpath -m "$1")
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd "$REPO"
echo "== (1) cfg5 leg"
timeout 900 python bench.py --leg cfg5 --blocks 256 --steps 1 > "$OUT/bench_cfg5.json" 2> "$OUT/bench_cfg5.log"; grep "^\[bench" "$OUT/bench_cfg5.log" | tail -5
echo "== (2) front end of 256 MiB blocks under host load"
timeout 700 python tools/host_contention.py 256 64 --only=idle,mmap_subproc 2>/dev/null | tee "$OUT/host_contention_256MiB.json"
echo "== (3) counters"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcx; timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT -d /tmp/pmcx -o p -- python "$REPO/tools/cm_coresidency.py" 2 768 --only=sync3 > /dev/null 2>&1
DB=$(find /tmp/pmcx -name "*.db" | head -1); [ -n "$DB" ] && python "$REPO/tools/rocpd_summary.py" --pmc "$DB" "decoder LDS counters, 768 x 2 MiB" | grep "k_cm_decode\|counter" | tee "$OUT/pmc_decoder_lds.txt"
rm -rf /tmp/pmcx; timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmcx -o p -- python "$REPO/tools/stage_probe.py" 256 --noise=0.035 > /dev/null 2>&1
DB=$(find /tmp/pmcx -name "*.db" | head -1); [ -n "$DB" ] && python "$REPO/tools/rocpd_summary.py" --pmc "$DB" "FETCH_SIZE (KiB, to be doubled on gfx950) over the stages of one 256 MiB block" | grep "k_bwt_tail\|k_lzp_links\|k_ub_walk\|counter" | tee "$OUT/pmc_fetch_stages.txt"

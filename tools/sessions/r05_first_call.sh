#!/bin/bash
# Suggested first GPU call of round 5 (left by round 4, whose budget ended before these could be measured):
#  (1) `bench.py --leg cfg5 --blocks 256`: a batch of 511 MiB blocks of the 16-symbol source, round trip (BASELINE.json configs[4]; ~8 GPU-minutes:
#      a 511 MiB block's CM launches last ~2 + ~4.5 minutes one block per CU);
#  (2) the front end of FULL-size blocks beside a busy host (64 x 256 MiB: idle, then 64 page-faulting threads of another process) -- round 4 measured
#      8 MiB and 64 MiB blocks only (profiles/r04_host_contention.json: still 2-3x);
#  (3) FETCH_SIZE of k_bwt_tail / k_lzp_links / k_ub_walk at 256 MiB (the decoder's LDS counters were taken in round 4's call 11);
#  (0) FIRST: one full-size step with the decoder's tail as eight slots of 8 blocks instead of four of 16 (BZ3_HIP_TAIL_PIPE=8,8; same 64 blocks in
#      flight): round 4's tail took 24.8 s for 12 s of whole-GPU kernels because a window's LZP decoders (0.7-1.2 s) outlast the three windows of
#      inverse BWTs they hide behind (0.84 s); seven windows of 8 are 0.98 s.  Compare "tail" in the step's split with profiles/r04_bench_768x256MiB.json
#      (at 8 MiB blocks the smaller windows lose: profiles/r04_tail_pipe_768x8MiB.txt).  ~7 GPU-minutes.
#  (0b) two full-size steps with BZ3_HIP_KEEP_WS=1 (api.hip keep_workspace: the encode call's workspace survives, the decode call reuses it and carves
#      its tail's swap buffers from it): the second step should lose the ~4-6 s of hipMalloc a step pays today.  ~11 GPU-minutes.  If both (0) and (0b)
#      win, make them the defaults.
# Usage: tools/r05_first_call.sh <outdir under gpurun_out>
OUT=$(realpath -m "$1")
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd "$REPO"
echo "== (0) full-size step, tail pipeline 8 x 8"
BZ3_HIP_TAIL_PIPE=8,8 timeout 900 python bench.py --gpus 1 --steps 1 --warmup 0 --no-extras --no-cpu-baseline > "$OUT/bench_tail8x8.json" 2> "$OUT/bench_tail8x8.log"; grep "^\[bench" "$OUT/bench_tail8x8.log" | tail -8
echo "== (0b) two full-size steps with the workspace kept across the calls (BZ3_HIP_KEEP_WS=1: no multi-GB hipMalloc after the first step; profiles/r04_first_touch.txt)"
BZ3_HIP_KEEP_WS=1 timeout 1100 python bench.py --gpus 1 --steps 2 --warmup 0 --no-extras --no-cpu-baseline > "$OUT/bench_keep_ws.json" 2> "$OUT/bench_keep_ws.log"; grep "^\[bench" "$OUT/bench_keep_ws.log" | tail -8
echo "== (1) cfg5 leg"
timeout 900 python bench.py --leg cfg5 --blocks 256 --steps 1 > "$OUT/bench_cfg5.json" 2> "$OUT/bench_cfg5.log"; grep "^\[bench" "$OUT/bench_cfg5.log" | tail -5
echo "== (2) front end of 256 MiB blocks under host load"
timeout 700 python tools/host_contention.py 256 64 --only=idle,mmap_subproc 2>/dev/null | tee "$OUT/host_contention_256MiB.json"
echo "== (3) counters"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcx; timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmcx -o p -- python "$REPO/tools/stage_probe.py" 256 --noise=0.035 > /dev/null 2>&1
DB=$(find /tmp/pmcx -name "*.db" | head -1); [ -n "$DB" ] && python "$REPO/tools/rocpd_summary.py" --pmc "$DB" "FETCH_SIZE (KiB, to be doubled on gfx950) over the stages of one 256 MiB block" | grep "k_bwt_tail\|k_lzp_links\|k_ub_walk\|counter" | tee "$OUT/pmc_fetch_stages.txt"

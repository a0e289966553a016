#!/bin/bash
# First GPU call of round 3: time everything that went into the source AFTER round 2's GPU budget was spent, each against the
# configuration the round measured.  About 15 GPU-minutes; every leg writes a small text file under <out>.
#   bash tools/r03_first_call.sh gpurun_out/r03_first
# Legs:
#   parity     the stage / block parity tests of the GPU suite (the changed kernels are in every one of them)
#   probes     tools/stage_probe.py on one 256 MiB text block under rocprofv3 --kernel-trace, kernel time by group:
#              default | inverse BWT with round 2's splitters (BZ3_UB_LOG_STRIDE=10) | two-pass BWT regrouping (BZ3_BWT_FUSED=1) |
#              LDS-staged radix scatter (BZ3_RS_STAGED=1) | radix tiles without the XCD mapping (BZ3_RS_NO_XCD=1)
#   pipe_ab    front-end / tail rings: auto (4 slots) | two slots of 6 / 32 blocks (round 2)          (BZ3_HIP_LZP_PIPE, BZ3_HIP_TAIL_PIPE)
#              on 768 x 32 MiB blocks (one step is ~1 min instead of ~8): t_enc - cm and t_dec - cm are the front end and the tail
set -e
OUT=$(realpath -m "$1")
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd "$REPO"

echo "== parity" | tee "$OUT/summary.txt"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "stage_parity or block_parity or batch_api or device_resident or front_end_and_tail_rings" > "$OUT/parity.log" 2>&1 || true
tail -2 "$OUT/parity.log" | tee -a "$OUT/summary.txt"

echo "== stage probes (256 MiB text block, rocprofv3 --kernel-trace; ms over both repetitions of tools/stage_probe.py)" | tee -a "$OUT/summary.txt"
probe() {  # name, env assignments
    local name=$1; shift
    ( cd /tmp && export TMPDIR=/tmp && rm -rf "$OUT/$name" &&
      env "$@" rocprofv3 --kernel-trace -d "$OUT/$name" -o pass -- python "$REPO/tools/stage_probe.py" 256 > "$OUT/$name.log" 2>&1 ) || { tail -5 "$OUT/$name.log"; return; }
    local db; db=$(find "$OUT/$name" -name "*.db" | head -1)
    python tools/rocpd_summary.py "$db" "rocprofv3 --kernel-trace -- $* python tools/stage_probe.py 256   (MI355X, ROCm 7.2)" > "$OUT/probe_$name.txt"
    rm -rf "$OUT/$name"
    python - "$OUT/probe_$name.txt" "$name" <<'PY'
import sys
groups = {"sorter (k_rs_*, k_scan_*)": ("k_rs_", "k_scan_"), "BWT regrouping (k_bwt_*, k_bg_*)": ("k_bwt_", "k_bg_"), "unBWT walks": ("k_ub_walk",),
          "LZP links": ("k_lzp_links",), "LZP driver": ("k_lzp_driver",)}
tot = dict.fromkeys(groups, 0.0)
for line in open(sys.argv[1]):
    f = line.split()
    if line.startswith("#") or len(f) < 6 or not f[-5].replace(".", "").isdigit():
        continue
    for g, pats in groups.items():
        if any(p in line[:110] for p in pats):
            tot[g] += float(f[-5])
print(f"{sys.argv[2]:<12}" + "   ".join(f"{g} {v:8.1f}" for g, v in tot.items()))
PY
}
probe default BZ3_PROBE_DUMMY=0 | tee -a "$OUT/summary.txt"          # what the library does now
probe stride1024 BZ3_UB_LOG_STRIDE=10 | tee -a "$OUT/summary.txt"    # inverse BWT: round 2's one splitter per 1024 rows
probe fused BZ3_BWT_FUSED=1 | tee -a "$OUT/summary.txt"              # BWT: two-pass regrouping (opt-in)
probe staged BZ3_RS_STAGED=1 | tee -a "$OUT/summary.txt"             # sorter: LDS-staged scatter (opt-in)
probe noxcd BZ3_RS_NO_XCD=1 | tee -a "$OUT/summary.txt"              # sorter: tile = blockIdx

echo "== pipe_ab" | tee -a "$OUT/summary.txt"
pipe() {  # name, env assignments
    local name=$1; shift
    env "$@" python bench.py --gpus 1 --steps 1 --warmup 0 --blocks 768 --block-mib 32 --no-cpu-baseline --no-extras > "$OUT/pipe_$name.json" 2> "$OUT/pipe_$name.progress.txt" || true
    python - "$OUT/pipe_$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
st = d["stages"]
print(f"{sys.argv[2]:<10} value {d['value']:8.1f} MiB/s   front end {st['t_enc_s'] - st['enc']['cm'] / 1e3:6.1f} s   tail {st['t_dec_s'] - st['dec']['cm'] / 1e3:6.1f} s   "
      f"(cm enc {st['enc']['cm'] / 1e3:.1f} s, dec {st['dec']['cm'] / 1e3:.1f} s; lzp driver window {st['enc']['lzp']:.0f} ms)")
PY
}
pipe ring BZ3_PIPE_DUMMY=0 | tee -a "$OUT/summary.txt"
pipe round2 BZ3_HIP_LZP_PIPE=6,2 BZ3_HIP_TAIL_PIPE=32,2 | tee -a "$OUT/summary.txt"
pipe ring_prio BZ3_HIP_AUX_PRIO=1 | tee -a "$OUT/summary.txt"   # side streams at the highest stream priority (opt-in)

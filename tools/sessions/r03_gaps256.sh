#!/bin/bash
# The front end and the tail at the real block size: 128 x 256 MiB (one block per CU: the CM launches take as long as with 768) under
# rocprofv3 --kernel-trace, then tools/rocpd_summary.py --gaps.   Usage: tools/r03_gaps256.sh <outdir> [blocks=128]
OUT=$(realpath -m "$1"); N=${2:-128}
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/p"
timeout 1200 rocprofv3 --kernel-trace -d "$OUT/p" -o pass -- python "$REPO/bench.py" --blocks $N --block-mib 256 --lean 1 --steps 1 --no-extras --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.log"
grep "^\[bench" "$OUT/bench.log" | tail -5
db=$(find "$OUT/p" -name "*.db" | head -1)
python "$REPO/tools/rocpd_summary.py" --gaps "$db" | tee "$OUT/gaps.txt"
python "$REPO/tools/rocpd_summary.py" "$db" > "$OUT/kernels.txt"
python - "$db" <<'EOP' | tee "$OUT/tail_timeline.txt"
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = db.execute(f'select {name}, start, "end" from kernels order by start').fetchall()
cm = [(s, e) for n, s, e in rows if n.startswith("bz3::k_cm_decode_sync") and e - s > 1e9]
t0 = cm[-1][1]
print("tail: kernels after the CM decode launch (ms since its end, duration ms), first 3.5 s; only kernels >= 0.3 ms and every k_lzp_decode")
for n, s, e in rows:
    if s < t0 - 2e9 or s > t0 + 3.5e9:
        continue
    d = (e - s) / 1e6
    if d >= 0.3 or "lzp_decode" in n:
        print(f"{(s - t0) / 1e6:10.2f} {d:9.2f}  {n[5:45]}")
EOP
rm -rf "$OUT/p"
python - "$OUT/bench.json" <<'EOP'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "stages", json.dumps(d["stages"]))
EOP

"""Where does the hardware put the five waves of the sync decoder's workgroups?  (GPU box.)
    python tools/cm_wave_placement.py [copies=768] [mode=sync3]
Runs the cycle-counter build over `copies` copies of a 2 MiB block and prints, per compute unit, the SIMD of every wave of the blocks it
hosted (HW_ID / XCC_ID registers), plus a summary: how many CUs have two or three walkers on one SIMD."""
import ctypes as C
import collections
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bzip3_amd  # noqa: E402
import datagen  # noqa: E402

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 768
mode = {"sync": 0, "sync2": 1, "sync3": 2}[sys.argv[2] if len(sys.argv) > 2 else "sync3"]
n = 2 << 20
lib = bzip3_amd.load()
g = bzip3_amd.StageApi(lib)
assert lib.bz3_hip_set_cm_mode(0) == 0
plain = g.bwt(datagen.text(n, seed=5, chains=2048))[1]
coded = g.cm_encode(plain)
assert lib.bz3_hip_set_cm_mode(mode) == 0
out = (C.c_uint8 * n)()
cnt = (C.c_uint64 * (16 * copies))()
os.environ["BZ3_CM_DEBUG"] = "3"
lib.bz3_hip_stage_cm_decode_many(bzip3_amd._cbuf(coded, len(coded)), len(coded), out, n, copies, cnt)
w = np.frombuffer(cnt, dtype=np.uint32).reshape(copies, 32)
a = np.frombuffer(cnt, dtype=np.uint64).reshape(copies, 16).astype(np.float64)
cus = collections.defaultdict(list)
for k in range(copies):
    ids = [(int(w[k, 22 + 2 * r]), int(w[k, 23 + 2 * r])) for r in range(5)]
    hw, xcc = ids[0]
    cu = (xcc & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)  # XCC, SE, SH, CU
    simds = [(h >> 4) & 3 for h, _ in ids]
    cus[cu].append((k, simds, a[k, 0] / n, a[k, 1] / n))
hist = collections.Counter()
for cu, blocks in sorted(cus.items()):
    walkers = collections.Counter(s[0] for _, s, _, _ in blocks)
    hist[(len(blocks), max(walkers.values()))] += 1
print("CUs used:", len(cus), " (blocks on the CU, most walkers on one SIMD) -> CUs:", dict(hist))
for cu, blocks in list(sorted(cus.items()))[:12]:
    print(cu, [(k, "".join(map(str, s)), round(wt), round(wk)) for k, s, wt, wk in blocks])
# does sharing a SIMD with another walker cost time?
by = collections.defaultdict(list)
for cu, blocks in cus.items():
    walkers = collections.Counter(s[0] for _, s, _, _ in blocks)
    for k, s, wt, wk in blocks:
        by[walkers[s[0]]].append(wt + wk)
print({k: (len(v), round(float(np.mean(v)), 1)) for k, v in sorted(by.items())}, " = walkers on the walker's SIMD -> (blocks, walker cycles per byte)")

"""Profiling aid: times k_cm_encode with only the coder wave / only the model waves running (BZ3_CM_DEBUG; output invalid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bzip3_amd, datagen
from oracle_lib import Oracle
n = int(float(sys.argv[1]) * (1 << 20)) if len(sys.argv) > 1 else 2 << 20
d = datagen.text(n, seed=5, chains=4096)
idx, u = Oracle().bwt(d)
g = bzip3_amd.StageApi(bzip3_amd.load())
for mode in ("0", "1", "2", "0"):
    os.environ["BZ3_CM_DEBUG"] = mode
    t = time.time(); out = g.cm_encode(u); dt = time.time() - t
    print(f"BZ3_CM_DEBUG={mode}: {dt*1e3:.0f} ms for {n} bytes = {dt/n*1e9:.0f} ns/B ({dt/n*2.4e9:.0f} cycles/B)", flush=True)

os.environ["BZ3_CM_DEBUG"] = "0"
enc = g.cm_encode(u)
if "--decode-phases" in sys.argv:
    import struct
    os.environ["BZ3_CM_DEBUG"] = "3"
    t = time.time(); out = g.cm_decode(enc, n); dt = time.time() - t
    v = struct.unpack("<5Q", out[:40])
    names = ("model", "barrier", "walk", "update")
    print(f"decode phases (wave 0, cycle counter ticks per byte; instrumented run {dt/n*1e9:.0f} ns/B): "
          + "  ".join(f"{k} {x / n:.1f}" for k, x in zip(names, v)) + f"  slow-path bytes {v[4] / n * 100:.1f}%", flush=True)
    os.environ["BZ3_CM_DEBUG"] = "0"
for mode in ("0", "1", "2", "0"):
    os.environ["BZ3_CM_DEBUG"] = mode
    t = time.time(); out = g.cm_decode(enc, n); dt = time.time() - t
    print(f"decode BZ3_CM_DEBUG={mode}: {dt*1e3:.0f} ms for {n} bytes = {dt/n*1e9:.0f} ns/B ({dt/n*2.4e9:.0f} cycles/B)", flush=True)

"""Profiling aid: times k_cm_encode with only the coder wave / only the model waves running (BZ3_CM_DEBUG; output invalid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bzip3_amd, datagen
n = int(float(sys.argv[1]) * (1 << 20)) if len(sys.argv) > 1 else 2 << 20
d = datagen.text(n, seed=5, chains=4096)
g = bzip3_amd.StageApi(bzip3_amd.load())
idx, u = g.bwt(d)  # what the CM stage sees: BWT output
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
    w = struct.unpack("<4Q", out[:32])
    m = struct.unpack("<4Q", out[64:96])
    print(f"decode phases, cycle counter ticks per byte (instrumented run {dt/n*1e9:.0f} ns/B):\n"
          f"   walker: wait for table {w[0]/n:.1f}  walk+publish {w[1]/n:.1f}  slow-path bytes {w[2]/n*100:.1f}%  wrong guesses {w[3]/n*100:.1f}%\n"
          f"   model wave 1: speculate {m[0]/n:.1f}  wait for byte {m[1]/n:.1f}  undo+redo {m[2]/n:.1f}  wrong guesses {m[3]/n*100:.1f}%", flush=True)
    os.environ["BZ3_CM_DEBUG"] = "0"
for mode in ("0", "0"):
    os.environ["BZ3_CM_DEBUG"] = mode
    t = time.time(); out = g.cm_decode(enc, n); dt = time.time() - t
    print(f"decode BZ3_CM_DEBUG={mode}: {dt*1e3:.0f} ms for {n} bytes = {dt/n*1e9:.0f} ns/B ({dt/n*2.4e9:.0f} cycles/B)", flush=True)

"""How do CM decoder variants behave when several blocks share a CU?  (GPU box, no torch import.)
    python tools/cm_coresidency.py [MiB=2] [copies ...=256 512 768] [--cycles] [--only=a,b] [--lib=path] [--exp=0,1,..]
For every decoder (sync = whole model, sync2 / sync3 = row caches for two / three blocks per CU) and every number of identical blocks: ONE launch of the CM decoder over
`copies` copies of the same coded block (bz3_hip_stage_cm_decode_many), launch time by HIP events, ns per byte and block,
aggregate MiB/s.  --cycles additionally runs the guess-ahead variants with BZ3_CM_DEBUG=3 and prints the decoder's phase
counters (cycles per byte: walker walk / wait, model wave speculate / wait / redo; shares of slow-path bytes and wrong
guesses), averaged over the copies -- the numbers that say whether neighbours on the CU cost issue slots or latency."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bzip3_amd  # noqa: E402
import datagen  # noqa: E402

MODES = {"auto": -1, "full": 0, "rows": 1, "rows3": 2, "sync": 0, "sync2": 1, "sync3": 2}  # (sync* = the decoders' kernel names: the same variants)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    mib = float(args[0]) if args else 2.0
    copies = [int(a) for a in args[1:]] or [256, 512, 768]
    cycles = "--cycles" in sys.argv
    n = int(mib * (1 << 20))
    libs = [a[len("--lib="):] for a in sys.argv if a.startswith("--lib=")]  # --lib=<path>: another build of the library (same-box A/B)
    lib = bzip3_amd.load(libs[0]) if libs else bzip3_amd.load()
    assert lib.bz3_hip_device_count() > 0
    g = bzip3_amd.StageApi(lib)
    assert lib.bz3_hip_set_cm_mode(0) == 0
    plain = g.bwt(datagen.text(n, seed=5, chains=2048))[1]  # what the CM stage sees: BWT output of text
    coded = g.cm_encode(plain)                              # coded by the full-model kernel; every variant must decode it back
    inb = bzip3_amd._cbuf(coded, len(coded))
    out = (C.c_uint8 * n)()
    only = [a[len("--only="):].split(",") for a in sys.argv if a.startswith("--only=")]
    exps = [a[len("--exp="):].split(",") for a in sys.argv if a.startswith("--exp=")]  # decoder experiments (bz3_hip_debug_cm_experiment), each run in turn
    exps = [int(x) for x in exps[0]] if exps else [0]
    for name, mode in MODES.items():
        if (only and name not in only[0]) or (not only and not name.startswith("sync")):
            continue
        assert lib.bz3_hip_set_cm_mode(mode) == 0
        for k, exp in [(k, e) for k in copies for e in exps]:
            if hasattr(lib, "bz3_hip_debug_cm_experiment"):
                lib.bz3_hip_debug_cm_experiment(exp)
            os.environ.pop("BZ3_CM_DEBUG", None)
            ms = lib.bz3_hip_stage_cm_decode_many(inb, len(coded), out, n, k, None)
            ok = bytes(out) == plain and ms >= 0  # (BZ3_CM_MANY_CHECK=1: ms == -2 when the copies disagree)
            rec = {"variant": name, "exp": exp, "copies": k, "block_mib": mib, "ms": round(ms, 1), "ns_per_byte_per_block": round(ms * 1e6 / n, 1),
                   "MiBps": round(k * mib / (ms * 1e-3), 1), "exact": ok}
            if cycles and not name.startswith("lock"):
                os.environ["BZ3_CM_DEBUG"] = "3"
                cnt = (C.c_uint64 * (16 * k))()
                lib.bz3_hip_stage_cm_decode_many(inb, len(coded), out, n, k, cnt)
                a = np.frombuffer(cnt, dtype=np.uint64).reshape(k, 16).astype(np.float64).mean(axis=0)
                if name.startswith("solo"):
                    rec["cyc_per_byte"] = {"e1": round(a[0] / n, 1), "walk6": round(a[1] / n, 1), "e2": round(a[3] / n, 1), "tail": round(a[4] / n, 1)}
                    rec["slow_path_share"] = round(a[2] / n, 3)
                    print(json.dumps(rec), flush=True)
                    continue
                rec["walker_cyc_per_byte"] = {"wait": round(a[0] / n, 1), "walk": round(a[1] / n, 1)}
                if name.startswith("sync") and a[3] > 0 and n > a[3]:  # wait per right / per wrong guess
                    rec["walker_wait_cyc"] = {"per_right_guess": round((a[0] - a[4]) / (n - a[3]), 1), "per_wrong_guess": round(a[4] / a[3], 1)}
                rec["walker_share"] = {"slow_path": round(a[2] / n, 3), "wrong_guess": round(a[3] / n, 3)}
                if True:
                    rec["model_wave_cyc_per_byte"] = {"speculate": round(a[8] / n, 1), "wait": round(a[9] / n, 1), "redo": round(a[10] / n, 1)}
            print(json.dumps(rec), flush=True)
    os.environ.pop("BZ3_CM_DEBUG", None)
    lib.bz3_hip_set_cm_mode(-1)


if __name__ == "__main__":
    main()

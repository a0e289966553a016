"""Which side of the CM encoder's LDS ring limits it when several blocks share a CU?  (GPU box, no torch import.)
    python tools/cm_encode_split.py [MiB=2] [copies ...=256 512 768] [--only=rows3] [--trio=0,1] [--check] [--lib=path]
For every encoder (full / rows / rows3) and every number of identical blocks: ONE launch over `copies` copies of the same BWT
output (bz3_hip_stage_cm_encode_many), normally, with the coder wave alone (BZ3_CM_DEBUG=1: a ring full of p = 1/2 events) and with
the model waves alone (BZ3_CM_DEBUG=2: nobody drains the ring, the waves never wait).
--trio=0,1: the three-per-CU encoder as three workgroups of two waves (0) and as the trio kernel (1: one workgroup of three blocks sharing one
coder wave, bz3_hip_debug_cm_encode_trio); --check: every copy's coded bytes are compared with copy 0's (ms = -2 when they differ)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bzip3_amd  # noqa: E402
import datagen  # noqa: E402

MODES = {"full": 0, "rows": 1, "rows3": 2}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    mib = float(args[0]) if args else 2.0
    copies = [int(a) for a in args[1:]] or [256, 512, 768]
    n = int(mib * (1 << 20))
    libs = [a[len("--lib="):] for a in sys.argv if a.startswith("--lib=")]  # --lib=<path>: another build of the library (same-box A/B)
    lib = bzip3_amd.load(libs[0]) if libs else bzip3_amd.load()
    assert lib.bz3_hip_device_count() > 0
    g = bzip3_amd.StageApi(lib)
    assert lib.bz3_hip_set_cm_mode(0) == 0
    plain = g.bwt(datagen.text(n, seed=5, chains=2048))[1]  # what the CM stage sees: BWT output of text
    want = g.cm_encode(plain)
    inb = bzip3_amd._cbuf(plain, n)
    out = (C.c_uint8 * (lib.bz3_bound(n) + 64))()
    only = [a[len("--only="):].split(",") for a in sys.argv if a.startswith("--only=")]
    trios = [a[len("--trio="):].split(",") for a in sys.argv if a.startswith("--trio=")]
    trios = [int(x) for x in trios[0]] if trios else [0]
    if "--check" in sys.argv:
        os.environ["BZ3_CM_MANY_CHECK"] = "1"
    for name, mode in MODES.items():
        if only and name not in only[0]:
            continue
        assert lib.bz3_hip_set_cm_mode(mode) == 0
        for k, trio in [(k, t) for k in copies for t in trios]:
            if hasattr(lib, "bz3_hip_debug_cm_encode_trio"):
                lib.bz3_hip_debug_cm_encode_trio(trio)
            rec = {"encoder": name, "trio": trio, "copies": k, "block_mib": mib}
            for what, dbg in (("all", None), ("coder_alone", "1"), ("model_alone", "2")):
                if trio and dbg == "1":
                    continue
                if dbg is None:
                    os.environ.pop("BZ3_CM_DEBUG", None)
                else:
                    os.environ["BZ3_CM_DEBUG"] = dbg
                coded = C.c_int32(0)
                ms = lib.bz3_hip_stage_cm_encode_many(inb, n, out, C.byref(coded), k)
                rec[what + "_ns_per_byte"] = round(ms * 1e6 / n, 1)
                if dbg is None:
                    rec["exact"] = ms >= 0 and bytes(out[: coded.value]) == want
                    rec["MiBps"] = round(k * mib / (ms * 1e-3), 1)
            print(json.dumps(rec), flush=True)
    os.environ.pop("BZ3_CM_DEBUG", None)
    lib.bz3_hip_set_cm_mode(-1)
    if hasattr(lib, "bz3_hip_debug_cm_encode_trio"):
        lib.bz3_hip_debug_cm_encode_trio(0)


if __name__ == "__main__":
    main()

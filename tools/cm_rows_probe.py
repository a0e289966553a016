"""CM kernel variants side by side on the GPU box (no torch import):
    python tools/cm_rows_probe.py [block MiB=16] [cfg ...]      cfg = <mode>:<blocks>, mode = full | rows | rows3 | auto
For every configuration: bz3_encode_blocks + bz3_decode_blocks on host buffers (text blocks, 64 KiB pieces of one
Markov text in a block-specific order), round trip verified, CM launch times from the library's HIP events.
Prints one JSON line per configuration."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bzip3_amd  # noqa: E402
import datagen  # noqa: E402

MODES = {"auto": -1, "full": 0, "rows": 1, "rows3": 2, "sync": 0, "sync2": 1, "sync3": 2}  # (sync* = the decoders' kernel names: the same variants)


T0 = time.time()


def note(msg):
    sys.stderr.write(f"[{time.time() - T0:7.1f}s] {msg}\n")
    sys.stderr.flush()


def main():
    mib = float(sys.argv[1]) if len(sys.argv) > 1 else 16
    cfgs = sys.argv[2:] or ["full:256", "rows:512"]
    n = int(mib * (1 << 20))
    lib = bzip3_amd.load()
    assert lib.bz3_hip_device_count() > 0
    base = np.frombuffer(datagen.text(n, seed=5, chains=8192), dtype=np.uint8)
    piece = 1 << 16
    npieces = n // piece
    cap = lib.bz3_bound(n) + 64
    for cfg in cfgs:
        mode, nblk = cfg.split(":")[:2]
        nblk = int(nblk)
        assert lib.bz3_hip_set_cm_mode(MODES[mode]) == 0
        rng = np.random.default_rng(17)
        bufs, plain = [], []
        variants = []  # 8 distinct blocks, reused round robin (the CM time of a block does not depend on its neighbours)
        for k in range(min(8, nblk)):
            d = base if k == 0 else np.concatenate([base[: npieces * piece].reshape(npieces, piece)[rng.permutation(npieces)].reshape(-1), base[npieces * piece:]])
            variants.append((np.ascontiguousarray(d), int(d.sum(dtype=np.uint64))))
        for k in range(nblk):
            d, sm = variants[k % len(variants)]
            b = (C.c_uint8 * cap)()
            C.memmove(b, d.ctypes.data, n)
            bufs.append(b)
            plain.append(sm)
        note(f"{cfg}: buffers ready")
        states = (C.c_void_p * nblk)(*[lib.bz3_new(n) for _ in range(nblk)])
        assert all(states)
        note(f"{cfg}: states created")
        ptrs = (C.c_void_p * nblk)(*[C.addressof(b) for b in bufs])
        sizes = (C.c_int32 * nblk)(*[n] * nblk)
        g0 = lib.bz3_hip_cm_blocks_given_up()
        t0 = time.time()
        lib.bz3_encode_blocks(states, ptrs, sizes, nblk)
        t1 = time.time()
        note(f"{cfg}: encode_blocks returned after {t1 - t0:.1f}s")
        assert all(lib.bz3_last_error(states[i]) == 0 and sizes[i] > 0 for i in range(nblk))
        tm = (C.c_float * 8)()
        lib.bz3_hip_last_timings(states[0], tm)
        cm_enc = tm[4]
        comp = sum(sizes)
        bsz = (C.c_size_t * nblk)(*[cap] * nblk)
        orig = (C.c_int32 * nblk)(*[n] * nblk)
        t2 = time.time()
        lib.bz3_decode_blocks(states, ptrs, bsz, sizes, orig, nblk)
        t3 = time.time()
        note(f"{cfg}: decode_blocks returned after {t3 - t2:.1f}s")
        lib.bz3_hip_last_timings(states[0], tm)
        cm_dec = tm[4]
        for i in range(nblk):
            assert lib.bz3_last_error(states[i]) == 0
            assert int(np.frombuffer(bufs[i], dtype=np.uint8, count=n).sum(dtype=np.uint64)) == plain[i], f"block {i}: round trip changed the data"
        for s in states:
            lib.bz3_free(s)
        tot = nblk * n / 2 ** 20
        print(json.dumps({"cfg": cfg, "block_mib": mib, "blocks": nblk, "cm_enc_ms": round(cm_enc, 1), "cm_dec_ms": round(cm_dec, 1),
                          "cm_enc_MiBps": round(tot / (cm_enc * 1e-3), 1), "cm_dec_MiBps": round(tot / (cm_dec * 1e-3), 1),
                          "t_enc_s": round(t1 - t0, 2), "t_dec_s": round(t3 - t2, 2), "round_trip_MiBps": round(tot / (t1 - t0 + t3 - t2), 1),
                          "ratio": round(nblk * n / comp, 3), "lean": os.environ.get("BZ3_HIP_LEAN", "0"), "given_up": lib.bz3_hip_cm_blocks_given_up() - g0}), flush=True)
        del bufs


if __name__ == "__main__":
    main()

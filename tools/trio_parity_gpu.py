"""GPU check of the trio CM encoder (cm.hip cm_encode_trio, bz3_hip_debug_cm_encode_trio) on DIFFERENT blocks in one batch: text blocks of several
sizes (odd ones too), a tiny block, a binary block (given up by the row cache: coded again by the whole-model kernel) through bz3_encode_blocks with
the three-per-CU variant forced, classic and lean (in-place) states, every coded block against the CPU checker (tests/oracle_lib.py).  No torch import.
    python tools/trio_parity_gpu.py [MiB=2]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bzip3_amd  # noqa: E402
import datagen  # noqa: E402
from oracle_lib import Oracle  # noqa: E402


def main():
    mib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    lib = bzip3_amd.load()
    assert lib.bz3_hip_device_count() > 0
    o = Oracle()
    bs = int(mib * (1 << 20))
    blocks = [datagen.text(bs, seed=60, chains=4096, noise=datagen.ENWIK_NOISE), datagen.text(bs // 2 + 1, seed=61, chains=2048), datagen.random_bytes(300001, seed=9),
              datagen.text(bs - 7, seed=62, chains=4096, noise=datagen.ENWIK_NOISE), b"tiny block " * 9, datagen.text(bs // 3, seed=63, chains=1024),
              datagen.low_entropy(bs // 2 + 3, seed=3), datagen.text(bs, seed=64, chains=4096)]
    want = [o.encode_block(d, bs)[2] for d in blocks]
    n = len(blocks)
    cap = lib.bz3_bound(bs) + 64
    out = {"block_mib": mib, "blocks": n}
    assert lib.bz3_hip_set_cm_mode(2) == 0
    try:
        for lean in (0, 1):
            lib.bz3_hip_set_lean_states(lean)
            for trio in (0, 1):
                l0 = lib.bz3_hip_debug_cm_encode_trio(trio)
                states = (C.c_void_p * n)(*[lib.bz3_new(bs) for _ in range(n)])
                assert all(states)
                bufs = [(C.c_uint8 * cap)() for _ in range(n)]
                for b, d in zip(bufs, blocks):
                    C.memmove(b, d, len(d))
                ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
                sizes = (C.c_int32 * n)(*[len(d) for d in blocks])
                g0 = lib.bz3_hip_cm_blocks_given_up()
                t0 = time.perf_counter()
                lib.bz3_encode_blocks(states, ptrs, sizes, n)
                t = time.perf_counter() - t0
                ok = all(lib.bz3_last_error(states[i]) == 0 and bytes(bufs[i][: sizes[i]]) == want[i] for i in range(n))
                out[f"lean{lean}_trio{trio}"] = {"identical": ok, "given_up": lib.bz3_hip_cm_blocks_given_up() - g0, "trio_launches": lib.bz3_hip_debug_cm_encode_trio(-1) - l0,
                                                 "t_s": round(t, 3)}
                for s in states:
                    lib.bz3_free(s)
    finally:
        lib.bz3_hip_set_lean_states(0)
        lib.bz3_hip_set_cm_mode(-1)
        lib.bz3_hip_debug_cm_encode_trio(0)
    print(json.dumps(out))
    assert all(v["identical"] for k, v in out.items() if k.startswith("lean")), out


if __name__ == "__main__":
    main()

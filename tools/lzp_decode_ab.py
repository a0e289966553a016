"""The LZP decoder (lzp.hip k_lzp_decode) of two builds of the library on the same box: one <MiB> block of the calibrated text, LZP-coded by the CPU checker,
decoded `reps` times through bz3_hip_stage_lzp_decode (host buffers: the call includes ~0.25 ms per MiB of PCIe copies on both sides, the same for both
builds); best wall time per build, output compared with the input.  No torch import.
    python tools/lzp_decode_ab.py [MiB=8] [reps=8] [--lib=<another build>]     (one build per process: run it twice)"""
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
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    mib = float(args[0]) if args else 8.0
    reps = int(args[1]) if len(args) > 1 else 8
    libs = [a[len("--lib="):] for a in sys.argv if a.startswith("--lib=")]
    d = datagen.text(int(mib * (1 << 20)), seed=5, noise=datagen.ENWIK_NOISE)
    m, lz = Oracle().lzp_encode(d)
    assert m > 0
    out = {"block_mib": mib, "lzp_bytes": m, "sequencing_points": lz.count(b"\xf2")}
    for name, path in ([("other", libs[0])] if libs else [("head", None)]):  # ONE build per process (run it twice for an A/B)
        lib = bzip3_amd.load(path) if path else bzip3_amd.load()
        assert lib.bz3_hip_device_count() > 0
        g = bzip3_amd.StageApi(lib)
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            n, got = g.lzp_decode(lz, len(d) + 64)
            best = min(best, time.perf_counter() - t0)
            assert n == len(d) and got == d, name
        out[name] = {"best_ms": round(best * 1e3, 2), "exact": True}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

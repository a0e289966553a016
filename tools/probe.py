"""Per-stage timing probe on the GPU box: python tools/probe.py <MiB> [kind] [--once]."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bzip3_amd  # noqa: E402
import datagen  # noqa: E402


def main():
    mib = float(sys.argv[1]) if len(sys.argv) > 1 else 8
    kind = sys.argv[2] if len(sys.argv) > 2 else "text"
    n = int(mib * (1 << 20))
    d = {"text": lambda: datagen.text(n, seed=5, chains=8192), "random": lambda: datagen.random_bytes(n),
         "lowent": lambda: datagen.low_entropy(n), "repeats": lambda: datagen.repeats(n),
         "bigtext": lambda: (datagen.text(32 << 20, seed=5, chains=65536) * (n // (32 << 20) + 1))[:n]}[kind]()
    lib = bzip3_amd.load()
    bs = max(n, 65 * 1024)
    with bzip3_amd.State(bs, lib) as st:
        reps = 1 if '--once' in sys.argv else 2
        for rep in range(reps):
            t0 = time.time()
            m, err, blk = st.encode_block(d)
            t1 = time.time()
            te = st.timings()
            bw = st.bwt_stats()
            k, err2, back = st.decode_block(blk, n)
            t2 = time.time()
            td = st.timings()
            assert err == 0 and err2 == 0 and back == d
            print(f"[{kind} {mib} MiB rep{rep}] enc {t1 - t0:.3f}s ({n / (t1 - t0) / 2**20:.1f} MiB/s) -> {m} B model={blk[8]}  dec {t2 - t1:.3f}s ({n / (t2 - t1) / 2**20:.1f} MiB/s)")
            print("   enc ms:", {a: round(b, 2) for a, b in te.items()}, bw)
            print("   dec ms:", {a: round(b, 2) for a, b in td.items()})
            sys.stdout.flush()


if __name__ == "__main__":
    main()

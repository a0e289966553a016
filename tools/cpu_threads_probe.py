"""How many reference threads does this host reward?  (Round 4: the GPU box's container has a 16-CPU quota -- /sys/fs/cgroup/cpu.max -- while
os.cpu_count() says 256; bench.py's cpu_baseline runs one thread per block on 64 blocks as the reference's CLI would with -j 64.)
Times the reference's bz3_encode_blocks + bz3_decode_blocks over 64 text blocks of <MiB> with 64, 32 and 16 threads (blocks in rounds).
Uses bench.py's cpu_baseline machinery (the only place outside tests/ that touches oracle/).
    python tools/cpu_threads_probe.py [MiB=32]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

import bench  # noqa: E402
import datagen  # noqa: E402


def main():
    mib = float(sys.argv[1]) if len(sys.argv) > 1 else 32.0
    n = int(mib * (1 << 20))
    base = np.frombuffer(datagen.text(n, seed=11, chains=4096, noise=datagen.ENWIK_NOISE), dtype=np.uint8)
    blocks = [np.roll(base, 4099 * k) for k in range(64)]
    label, path, probe = bench.fastest_reference(blocks[0][: 8 << 20].tobytes())
    out = {"block_mib": mib, "ref": label, "cpu_max": open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None}
    for threads in (64, 32, 16):
        t_enc = t_dec = 0.0
        for r0 in range(0, 64, threads):
            rec, _ = bench.reference_round_trip(blocks[r0 : r0 + threads], n, lib_path=path, label=label)
            t_enc += rec["t_enc_s"]
            t_dec += rec["t_dec_s"]
        out[f"threads_{threads}"] = {"MiBps": round(64 * mib / (t_enc + t_dec), 2), "t_enc_s": round(t_enc, 2), "t_dec_s": round(t_dec, 2)}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

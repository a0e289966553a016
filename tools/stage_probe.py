"""Times the whole-GPU stages (everything except the CM coder) at full block size through the stage hooks:
python tools/stage_probe.py <MiB> [--lib=<another build>] [--noise=<fraction of noise tokens, default 0>].  Host wall-clock per call, includes the H2D/D2H of the hook."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402  (first: shared HIP runtime)

import bzip3_amd  # noqa: E402
from bench import gen_text_device  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    mib = float(args[0]) if args else 256
    n = int(mib * (1 << 20))
    libs = [a[len("--lib="):] for a in sys.argv if a.startswith("--lib=")]
    noise = [float(a[len("--noise="):]) for a in sys.argv if a.startswith("--noise=")]
    d = bytes(gen_text_device(torch, n, 7, torch.device("cuda", 0), noise=noise[0] if noise else 0.0).cpu().numpy())
    g = bzip3_amd.StageApi(bzip3_amd.load(libs[0]) if libs else bzip3_amd.load())
    for rep in range(2):
        t = time.time(); crc = g.crc32c(d); t_crc = time.time() - t
        t = time.time(); rle = g.mrle_encode(d); t_rle = time.time() - t
        t = time.time(); nl, lz = g.lzp_encode(d); t_lzp = time.time() - t
        src = lz if nl > 0 else d
        t = time.time(); idx, u = g.bwt(src); t_bwt = time.time() - t; ms_bwt = g.lib.bz3_hip_stage_last_ms()
        t = time.time(); rc, back = g.unbwt(u, idx); t_unbwt = time.time() - t; ms_unbwt = g.lib.bz3_hip_stage_last_ms()
        assert rc == 0 and back == src
        t_unlzp = 0.0
        if nl > 0:
            t = time.time(); k, back2 = g.lzp_decode(lz, n + 100); t_unlzp = time.time() - t
            assert k == n and back2 == d
        print(f"[{mib:g} MiB rep{rep}] crc {t_crc*1e3:.0f} ms  rle {t_rle*1e3:.0f} ms (-> {len(rle)})  lzp {t_lzp*1e3:.0f} ms (-> {nl})  "
              f"bwt {t_bwt*1e3:.0f} ms (transform alone {ms_bwt:.1f})  unbwt {t_unbwt*1e3:.0f} ms (transform alone {ms_unbwt:.1f})  unlzp {t_unlzp*1e3:.0f} ms   [hook times include ~{n/25e9*2e3:.0f} ms of PCIe copies]")
        sys.stdout.flush()


if __name__ == "__main__":
    main()

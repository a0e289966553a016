"""How often does the sync decoder's fast walk hand a byte to the checked walk?  CPU only: runs the kernel's cycle-counter build under
the fiber emulator (tests/emu) on the BWT output of `KiB` of the bench's text and prints the share of slow-path bytes beside the
share a perfect test would have (bytes during which the coder renormalises ~ coded bytes / bytes).
    python tools/cm_slowpath_emu.py [KiB=192]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emu")]
import bzip3_amd  # noqa: E402
import datagen  # noqa: E402
from build_emu import build  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

kib = int(sys.argv[1]) if len(sys.argv) > 1 else 192
n = kib << 10
lib = bzip3_amd._declare(C.CDLL(build()))
o = Oracle()
plain = o.bwt(datagen.text(n, seed=5, chains=64))[1]
coded = o.cm_encode(plain)
assert lib.bz3_hip_set_cm_mode(5) == 0  # sync, whole model
out = (C.c_uint8 * n)()
cnt = (C.c_uint64 * 16)()
os.environ["BZ3_CM_DEBUG"] = "3"
ms = lib.bz3_hip_stage_cm_decode_many(bzip3_amd._cbuf(coded, len(coded)), len(coded), out, n, 1, cnt)
os.environ.pop("BZ3_CM_DEBUG")
rep = float((np.frombuffer(plain, dtype=np.uint8)[1:] == np.frombuffer(plain, dtype=np.uint8)[:-1]).mean())
print(f"n {n}  coded {len(coded)} ({len(coded) / n:.4f} per byte)  slow-path share {cnt[2] / n:.4f}  wrong guesses {cnt[3] / n:.4f}  (repeat rate {rep:.4f})")

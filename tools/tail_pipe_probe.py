"""Shape of the decoder's tail pipeline (api.hip decode_group: windows of blocks whose serial LZP decoders run on side streams while the
inverse BWTs of the following windows and the mRLE / CRC stages of the preceding ones run on the group's stream).
One batch of `blocks` text blocks of `MiB` each is encoded once on the GPU; the coded blocks are stashed in device memory and decoded
again under every "window,slots" setting given (BZ3_HIP_TAIL_PIPE; "default" = what decode_group picks itself), `trials` times each.
Prints t_dec, the CM launch and what is left (the tail) per trial.  No torch import: device memory through the HIP runtime directly.
    python tools/tail_pipe_probe.py [MiB=8] [blocks=768] [--settings=default,8x8,4x8] [--trials=2] [--emu]
(--emu: the CPU emulator build and host memory -- checks the script, not the GPU.)"""
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


class HipMem:
    """hipMalloc / hipMemcpy through ctypes (the runtime libbzip3.so itself uses)."""

    def __init__(self):
        import importlib.util

        cand = "libamdhip64.so"
        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.origin and os.environ.get("BZ3_HIP_SYSTEM_RUNTIME") != "1":
            p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
            if os.path.exists(p):
                cand = p
        self.rt = rt = C.CDLL(cand, mode=C.RTLD_GLOBAL)
        rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        rt.hipDeviceSynchronize.argtypes = []

    def alloc(self, n):
        p = C.c_void_p()
        rc = self.rt.hipMalloc(C.byref(p), n)
        assert rc == 0 and p.value, f"hipMalloc({n}) = {rc}"
        return p.value

    def h2d(self, dst, src_np):
        assert self.rt.hipMemcpy(dst, src_np.ctypes.data, src_np.nbytes, 1) == 0

    def d2h(self, dst_np, src):
        assert self.rt.hipMemcpy(dst_np.ctypes.data, src, dst_np.nbytes, 2) == 0

    def d2d(self, dst, src, n):
        assert self.rt.hipMemcpy(dst, src, n, 3) == 0

    def sync(self):
        assert self.rt.hipDeviceSynchronize() == 0


class HostMem:
    """--emu: the emulator's device memory is host memory."""

    def __init__(self):
        self.keep = []

    def alloc(self, n):
        b = (C.c_uint8 * n)()
        self.keep.append(b)
        return C.addressof(b)

    def h2d(self, dst, src_np):
        C.memmove(dst, src_np.ctypes.data, src_np.nbytes)

    def d2h(self, dst_np, src):
        C.memmove(dst_np.ctypes.data, src, dst_np.nbytes)

    def d2d(self, dst, src, n):
        C.memmove(dst, src, n)

    def sync(self):
        pass


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opt = {a[2:].split("=")[0]: (a.split("=", 1)[1] if "=" in a else "1") for a in sys.argv[1:] if a.startswith("--")}
    emu = "emu" in opt
    mib = float(args[0]) if args else 8.0
    nblk = int(args[1]) if len(args) > 1 else 768
    settings = opt.get("settings", "default,8x8,4x8").split(",")
    trials = int(opt.get("trials", "2"))
    if emu:
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        from build_emu import build as build_emu

        lib = bzip3_amd.load(build_emu())
        mem = HostMem()
    else:
        lib = bzip3_amd.load()
        assert lib.bz3_hip_device_count() > 0
        mem = HipMem()
    lib.bz3_hip_bind_device(0)
    lib.bz3_hip_set_lean_states(1)
    bs = int(mib * (1 << 20))
    cap = lib.bz3_bound(bs) + 4096
    base = np.frombuffer(datagen.text(bs, seed=1, noise=0.0 if emu else datagen.ENWIK_NOISE), dtype=np.uint8)
    bufs = []
    for k in range(nblk):
        p = mem.alloc(cap)
        mem.h2d(p, np.roll(base, k * 4099))
        bufs.append(p)
    states = (C.c_void_p * nblk)(*[lib.bz3_new(bs) for _ in range(nblk)])
    assert all(states)
    ptrs = (C.c_void_p * nblk)(*bufs)
    bsz = (C.c_size_t * nblk)(*[cap] * nblk)
    orig = (C.c_int32 * nblk)(*[bs] * nblk)
    sizes = (C.c_int32 * nblk)(*[bs] * nblk)
    t0 = time.perf_counter()
    lib.bz3_hip_encode_blocks_device(states, ptrs, sizes, nblk)
    t_enc = time.perf_counter() - t0
    coded = list(sizes)
    assert min(coded) > 0, min(coded)
    offs = np.concatenate([[0], np.cumsum([(c + 255) & ~255 for c in coded])]).astype(np.int64)
    stash = mem.alloc(int(offs[-1]) + 256)
    for k in range(nblk):
        mem.d2d(stash + int(offs[k]), bufs[k], coded[k])
    mem.sync()
    tm = (C.c_float * 8)()
    out = {"blocks": nblk, "block_mib": mib, "t_enc_s": round(t_enc, 3), "coded_ratio": round(bs * nblk / float(sum(coded)), 3), "runs": []}

    def decode(setting):
        for k in range(nblk):
            mem.d2d(bufs[k], stash + int(offs[k]), coded[k])
        mem.sync()
        if setting == "default":
            os.environ.pop("BZ3_HIP_TAIL_PIPE", None)
        else:
            os.environ["BZ3_HIP_TAIL_PIPE"] = setting.replace("x", ",")
        sz = (C.c_int32 * nblk)(*coded)
        t0 = time.perf_counter()
        lib.bz3_hip_decode_blocks_device(states, ptrs, bsz, sz, orig, nblk)
        t = time.perf_counter() - t0
        errs = [lib.bz3_last_error(st) for st in states]
        assert not any(errs), [e for e in errs if e][:4]
        lib.bz3_hip_last_timings(states[0], tm)
        cm = tm[bzip3_amd.T_NAMES.index("cm")] * 1e-3
        return {"setting": setting, "t_dec_s": round(t, 3), "cm_s": round(cm, 3), "tail_s": round(t - cm, 3)}

    decode("default")  # warm-up: allocations
    for _ in range(trials):
        for s in settings:
            out["runs"].append(decode(s))
            print(json.dumps(out["runs"][-1]), flush=True)
    os.environ.pop("BZ3_HIP_TAIL_PIPE", None)
    got = np.empty(bs, dtype=np.uint8)
    for k in (0, nblk - 1):
        mem.d2h(got, bufs[k])
        assert np.array_equal(got, np.roll(base, k * 4099)), f"block {k} differs"
    best = {}
    for r in out["runs"]:
        best[r["setting"]] = min(best.get(r["setting"], 1e9), r["tail_s"])
    print(json.dumps({"blocks": nblk, "block_mib": mib, "t_enc_s": out["t_enc_s"], "coded_ratio": out["coded_ratio"], "best_tail_s": best, "exact": True}))


if __name__ == "__main__":
    main()

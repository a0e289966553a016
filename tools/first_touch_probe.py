"""What does FRESH device memory cost?  hipMalloc of <GiB> GiB, then the same hipMemsetAsync twice (HIP events around each), then hipFree, then once more
(does the driver hand the same pages back warm?).  Round 4 saw the first launch of a process take ~170 ms longer than the second over 1.7 GB of new
buffers (profiles/r04_cm_encoder_trio.txt); a bench step frees and allocates ~130 GB of workspace between its two calls.  No torch import.
    python tools/first_touch_probe.py [GiB ...=1 8 32]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import bzip3_amd  # noqa: E402  (loads the HIP runtime the library uses)


def main():
    sizes = [float(a) for a in sys.argv[1:]] or [1, 8, 32]
    lib = bzip3_amd.load()
    assert lib.bz3_hip_device_count() > 0
    import importlib.util

    cand = "libamdhip64.so"
    spec = importlib.util.find_spec("torch")
    if spec is not None and spec.origin and os.environ.get("BZ3_HIP_SYSTEM_RUNTIME") != "1":
        p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(p):
            cand = p
    rt = C.CDLL(cand, mode=C.RTLD_GLOBAL)
    rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    rt.hipFree.argtypes = [C.c_void_p]
    rt.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    rt.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
    rt.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
    rt.hipEventSynchronize.argtypes = [C.c_void_p]
    rt.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
    e0, e1 = C.c_void_p(), C.c_void_p()
    assert rt.hipEventCreate(C.byref(e0)) == 0 and rt.hipEventCreate(C.byref(e1)) == 0

    def memset_ms(p, n):
        rt.hipEventRecord(e0, None)
        assert rt.hipMemsetAsync(p, 1, n, None) == 0
        rt.hipEventRecord(e1, None)
        rt.hipEventSynchronize(e1)
        ms = C.c_float(0)
        rt.hipEventElapsedTime(C.byref(ms), e0, e1)
        return round(ms.value, 2)

    warm = C.c_void_p()
    assert rt.hipMalloc(C.byref(warm), 1 << 20) == 0  # (context creation out of the way)
    memset_ms(warm, 1 << 20)
    for g in sizes:
        n = int(g * (1 << 30))
        rec = {"GiB": g}
        for rnd in ("first", "again"):
            p = C.c_void_p()
            t0 = time.perf_counter()
            rc = rt.hipMalloc(C.byref(p), n)
            t1 = time.perf_counter()
            if rc != 0:
                rec[rnd] = {"hipMalloc": rc}
                break
            a = memset_ms(p, n)
            b = memset_ms(p, n)
            t2 = time.perf_counter()
            rt.hipFree(p)
            t3 = time.perf_counter()
            rec[rnd] = {"hipMalloc_ms": round((t1 - t0) * 1e3, 2), "memset_first_ms": a, "memset_second_ms": b, "hipFree_ms": round((t3 - t2) * 1e3, 2)}
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()

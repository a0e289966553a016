"""Why did the encoder's front end run 2.6x slower while 64 reference threads ran in the SAME process?
Times bz3_hip_encode_blocks_device + bz3_hip_decode_blocks_device over 768 x <MiB> text blocks (GPU box) under four host conditions:
    idle          nothing else runs
    spin_inproc   64 threads of this process burn CPU without touching memory (zlib.crc32 over a cached 256 KiB buffer: no page faults)
    mmap_inproc   64 threads of this process map, touch and unmap 512 MiB of anonymous memory in a loop (what 64 reference states of
                  256 MiB blocks do at bz3_new / first touch: page faults and the process's mmap lock)
    mmap_subproc  the same 64 threads in ANOTHER process, pinned to the upper half of the cores
The front end (t_enc - CM launch) is the part of a step in which the host sits between kernels (stream syncs), so it is the part a busy
host can slow down.
    python tools/host_contention.py [MiB=8] [blocks=768] [--only=idle,mmap_subproc]
"""
import ctypes as C
import json
import mmap
import os
import subprocess
import sys
import threading
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
STOP = threading.Event()


def spin():
    buf = bytes(256 << 10)
    while not STOP.is_set():
        for _ in range(200):
            zlib.crc32(buf)


def churn(mb=512):
    libc = C.CDLL(None)
    libc.memset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    while not STOP.is_set():
        m = mmap.mmap(-1, mb << 20)
        a = (C.c_char * (mb << 20)).from_buffer(m)
        libc.memset(C.addressof(a), 1, mb << 20)  # (ctypes releases the GIL)
        del a
        m.close()


def load_threads(fn, n=64):
    ts = [threading.Thread(target=fn, daemon=True) for _ in range(n)]
    for t in ts:
        t.start()
    return ts


def worker_main():  # mmap_subproc's child
    ncpu = os.cpu_count() or 2
    try:
        os.sched_setaffinity(0, set(range(ncpu // 2, ncpu)))
    except OSError:
        pass
    load_threads(churn)
    time.sleep(3600)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        return worker_main()
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only = [a[len("--only="):].split(",") for a in sys.argv if a.startswith("--only=")]
    only = only[0] if only else ["idle", "spin_inproc", "mmap_inproc", "mmap_subproc", "idle_again"]
    mib = float(args[0]) if len(args) > 0 else 8.0
    nblk = int(args[1]) if len(args) > 1 else 768
    import torch

    import bench
    import bzip3_amd

    lib = bzip3_amd.load()
    lib.bz3_hip_bind_device(0)
    lib.bz3_hip_set_lean_states(1)
    dev = torch.device("cuda", 0)
    bench.seed_text_source(lib)
    bs = int(mib * (1 << 20))
    cap = lib.bz3_bound(bs) + 4096
    base = bench.gen_text_device(torch, bs, seed=1, device=dev)
    bufs = []
    for k in range(nblk):
        b = torch.empty(cap, dtype=torch.uint8, device=dev)
        b[:bs] = base.roll(k * 4099)
        bufs.append(b)
    torch.cuda.synchronize()
    states = (C.c_void_p * nblk)(*[lib.bz3_new(bs) for _ in range(nblk)])
    ptrs = (C.c_void_p * nblk)(*[b.data_ptr() for b in bufs])
    bsz = (C.c_size_t * nblk)(*[cap] * nblk)
    orig = (C.c_int32 * nblk)(*[bs] * nblk)

    def step():
        sizes = (C.c_int32 * nblk)(*[bs] * nblk)
        t0 = time.perf_counter()
        lib.bz3_hip_encode_blocks_device(states, ptrs, sizes, nblk)
        t1 = time.perf_counter()
        tm = (C.c_float * 8)()
        lib.bz3_hip_last_timings(states[0], tm)
        cm_e = tm[bzip3_amd.T_NAMES.index("cm")]
        lib.bz3_hip_decode_blocks_device(states, ptrs, bsz, sizes, orig, nblk)
        t2 = time.perf_counter()
        lib.bz3_hip_last_timings(states[0], tm)
        cm_d = tm[bzip3_amd.T_NAMES.index("cm")]
        return {"t_enc_s": round(t1 - t0, 3), "front_end_s": round(t1 - t0 - cm_e * 1e-3, 3), "t_dec_s": round(t2 - t1, 3), "tail_s": round(t2 - t1 - cm_d * 1e-3, 3)}

    step()  # warm-up: allocations
    out = {"blocks": nblk, "block_mib": mib, "host_cores": os.cpu_count()}
    if "idle" in only:
        out["idle"] = step()
    for name, fn in (("spin_inproc", spin), ("mmap_inproc", churn)):
        if name not in only:
            continue
        STOP.clear()
        ts = load_threads(fn)
        time.sleep(1.0)
        out[name] = step()
        STOP.set()
        for t in ts:
            t.join()
    if "mmap_subproc" in only:
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker"])
        time.sleep(2.0)
        out["mmap_subproc"] = step()
        p.kill()
        p.wait()
    if "idle_again" in only:
        out["idle_again"] = step()
    print(json.dumps(out))


if __name__ == "__main__":
    main()

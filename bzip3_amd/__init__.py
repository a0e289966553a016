"""bzip3_amd -- MI355X-native bzip3 block codec behind the libbz3.h C ABI.

The product is `bzip3_amd/lib/libbzip3.so` (hand-written HIP kernels for gfx950, built by
`bzip3_amd/build.py`).  This module is only the Python-side loader / thin mirror of the C API used by
the tests and bench.py; it contains no compute and NO fallback: if the shared object is missing or no
HIP device is usable, it raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libbzip3.so")

BZ3_OK = 0
BZ3_ERR_OUT_OF_BOUNDS = -1
BZ3_ERR_BWT = -2
BZ3_ERR_CRC = -3
BZ3_ERR_MALFORMED_HEADER = -4
BZ3_ERR_TRUNCATED_DATA = -5
BZ3_ERR_DATA_TOO_BIG = -6
BZ3_ERR_INIT = -7
BZ3_ERR_DATA_SIZE_TOO_SMALL = -8

T_NAMES = ["crc", "rle", "lzp", "bwt", "cm", "copy", "_6", "_7"]

_lib = None


def _declare(L, strict=True):
    vp, i32, u32, sz = C.c_void_p, C.c_int32, C.c_uint32, C.c_size_t
    sig = {
        "bz3_version": (C.c_char_p, []),
        "bz3_last_error": (C.c_int8, [vp]),
        "bz3_strerror": (C.c_char_p, [vp]),
        "bz3_new": (vp, [i32]),
        "bz3_free": (None, [vp]),
        "bz3_bound": (sz, [sz]),
        "bz3_compress": (C.c_int, [u32, vp, vp, sz, C.POINTER(sz)]),
        "bz3_decompress": (C.c_int, [vp, vp, sz, C.POINTER(sz)]),
        "bz3_min_memory_needed": (sz, [i32]),
        "bz3_encode_block": (i32, [vp, vp, i32]),
        "bz3_decode_block": (i32, [vp, vp, sz, i32, i32]),
        "bz3_encode_blocks": (None, [vp, vp, vp, i32]),
        "bz3_decode_blocks": (None, [vp, vp, vp, vp, vp, i32]),
        "bz3_orig_size_sufficient_for_decode": (C.c_int, [vp, sz, i32]),
        # bz3_hip.h
        "bz3_hip_device_count": (C.c_int, []),
        "bz3_hip_bind_device": (C.c_int, [C.c_int]),
        "bz3_hip_state_device": (C.c_int, [vp]),
        "bz3_hip_set_cm_mode": (C.c_int, [C.c_int]),
        "bz3_hip_cm_blocks_given_up": (C.c_uint, []),
        "bz3_hip_cm_blocks_routed_full": (C.c_uint, []),
        "bz3_hip_debug_bwt_big_rounds": (None, [C.c_int]),
        "bz3_hip_debug_peak_concurrent_groups": (C.c_int, [C.c_int]),
        "bz3_hip_debug_front_end_ring": (C.c_int, []),
        "bz3_hip_debug_arena_swap_buffers": (C.c_int, [C.c_int]),
        "bz3_hip_cm_variant_for": (C.c_int, [C.c_int, C.c_int, C.c_int]),
        "bz3_hip_set_lean_states": (C.c_int, [C.c_int]),
        "bz3_hip_release_cached_memory": (None, []),
        "bz3_hip_set_keep_workspace": (C.c_int, [C.c_int]),
        "bz3_hip_set_front_end_duo": (C.c_int, [C.c_int]),
        "bz3_hip_set_workspace_headroom": (None, [C.c_longlong]),
        "bz3_hip_workspace_headroom": (sz, []),
        "bz3_hip_debug_headroom_events": (C.c_uint, [C.c_int, C.POINTER(C.c_uint)]),
        "bz3_hip_debug_ring_contexts": (sz, [sz, sz, sz, sz, sz, sz, C.c_int, sz]),
        "bz3_hip_debug_arena_slack": (sz, [sz]),
        "bz3_hip_debug_cm_launches": (C.c_uint, [C.c_int]),
        "bz3_hip_debug_workspace_bytes": (sz, [sz, C.c_int]),
        "bz3_hip_debug_cached_bytes": (sz, [C.c_int]),
        "bz3_hip_encode_block_device": (i32, [vp, vp, i32]),
        "bz3_hip_decode_block_device": (i32, [vp, vp, sz, i32, i32]),
        "bz3_hip_encode_blocks_device": (None, [vp, vp, vp, i32]),
        "bz3_hip_decode_blocks_device": (None, [vp, vp, vp, vp, vp, i32]),
        "bz3_hip_last_timings": (None, [vp, C.POINTER(C.c_float)]),
        "bz3_hip_last_bwt_stats": (None, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_uint64)]),
        "bz3_hip_stage_crc32c": (u32, [vp, sz, u32]),
        "bz3_hip_stage_mrle_encode": (i32, [vp, i32, vp]),
        "bz3_hip_stage_mrle_decode": (C.c_int, [vp, vp, i32, i32]),
        "bz3_hip_stage_lzp_encode": (i32, [vp, i32, vp]),
        "bz3_hip_stage_lzp_decode": (i32, [vp, i32, vp, i32]),
        "bz3_hip_stage_bwt": (i32, [vp, vp, i32]),
        "bz3_hip_stage_unbwt": (i32, [vp, vp, i32, i32]),
        "bz3_hip_stage_last_ms": (C.c_float, []),
        "bz3_hip_debug_sort_u32": (i32, [vp, u32, C.c_int, C.c_int, vp, vp]),
        "bz3_hip_debug_scan_u32": (i32, [vp, u32, vp]),
        "bz3_hip_debug_cu_masks": (i32, [C.c_int, C.c_int, vp, vp]),
        "bz3_hip_set_collect_window_us": (None, [C.c_int]),
        "bz3_hip_debug_collected_batches": (C.c_uint, [C.c_int, C.POINTER(C.c_uint)]),
        "bz3_hip_stage_cm_encode": (i32, [vp, i32, vp]),
        "bz3_hip_stage_cm_decode": (None, [vp, i32, vp, i32]),
        "bz3_hip_stage_cm_decode_many": (C.c_float, [vp, i32, vp, i32, i32, vp]),
        "bz3_hip_stage_cm_encode_many": (C.c_float, [vp, i32, vp, C.POINTER(i32), i32]),
        "bz3_hip_encode_stream": (C.c_int, [C.c_int, C.c_int, i32, i32]),
        "bz3_hip_decode_stream": (C.c_int, [C.c_int, C.c_int, i32]),
    }
    for name, (res, args) in sig.items():
        if not strict and not hasattr(L, name):  # an OLDER build loaded for a same-box A/B (load(path)): symbols added since are absent
            continue
        fn = getattr(L, name)  # AttributeError here = the library does not export what include/*.h declares
        fn.restype = res
        fn.argtypes = args
    return L


EXPORTED_SYMBOLS = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7; two HIP runtimes in one process do not both see
    the GPU.  If torch is installed, load ITS runtime first (without importing torch) so that libbzip3.so --
    which only asks for the soname libamdhip64.so.7 -- and a later `import torch` share one runtime.
    Set BZ3_HIP_SYSTEM_RUNTIME=1 to keep the system ROCm runtime instead."""
    if os.environ.get("BZ3_HIP_SYSTEM_RUNTIME") == "1":
        return
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def load(path=None):
    """Load libbzip3.so (building nothing, falling back to nothing)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    if path is None:
        _share_hip_runtime_with_torch()
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} is missing: the HIP extension has not been built (python -m bzip3_amd.build). "
            "bzip3_amd has no CPU or PyTorch fallback by design."
        )
    L = _declare(C.CDLL(p), strict=path is None)
    if path is None:
        _lib = L
    return L


def _cbuf(data, cap):
    cap = max(1, cap)
    if len(data) == cap:
        return (C.c_uint8 * cap).from_buffer_copy(data)
    buf = (C.c_uint8 * cap)()
    if len(data):
        C.memmove(buf, data if isinstance(data, bytes) else bytes(data), len(data))
    return buf


class StageApi:
    """Per-stage hooks on host buffers (bz3_hip_stage_*), same call shapes as the CPU checker used by the tests."""

    def __init__(self, lib=None):
        self.lib = lib or load()

    def crc32c(self, data, init=1):
        return self.lib.bz3_hip_stage_crc32c(_cbuf(data, len(data)), len(data), init)

    def mrle_encode(self, data):
        out = (C.c_uint8 * (len(data) + 64))()
        n = self.lib.bz3_hip_stage_mrle_encode(_cbuf(data, len(data)), len(data), out)
        return C.string_at(out, n)

    def mrle_decode(self, data, outlen, maxin=None):
        maxin = len(data) if maxin is None else maxin
        out = (C.c_uint8 * max(1, outlen))()
        rc = self.lib.bz3_hip_stage_mrle_decode(_cbuf(data, len(data)), out, outlen, maxin)
        return rc, C.string_at(out, outlen)

    def lzp_encode(self, data):
        out = (C.c_uint8 * (len(data) + 64))()
        n = self.lib.bz3_hip_stage_lzp_encode(_cbuf(data, len(data)), len(data), out)
        return n, (C.string_at(out, n) if n > 0 else b"")

    def lzp_decode(self, data, maxout):
        out = (C.c_uint8 * max(8, maxout))()
        n = self.lib.bz3_hip_stage_lzp_decode(_cbuf(data, len(data)), len(data), out, maxout)
        return n, (C.string_at(out, n) if n > 0 else b"")

    def bwt(self, data):
        out = (C.c_uint8 * max(1, len(data)))()
        idx = self.lib.bz3_hip_stage_bwt(_cbuf(data, len(data)), out, len(data))
        return idx, C.string_at(out, len(data))

    def unbwt(self, data, idx):
        out = (C.c_uint8 * max(1, len(data)))()
        rc = self.lib.bz3_hip_stage_unbwt(_cbuf(data, len(data)), out, len(data), idx)
        return rc, C.string_at(out, len(data))

    def cm_encode(self, data):
        out = (C.c_uint8 * (len(data) + len(data) // 50 + 64))()
        n = self.lib.bz3_hip_stage_cm_encode(_cbuf(data, len(data)), len(data), out)
        return C.string_at(out, n)

    def cm_decode(self, data, n):
        out = (C.c_uint8 * max(1, n))()
        self.lib.bz3_hip_stage_cm_decode(_cbuf(data, len(data)), len(data), out, n)
        return C.string_at(out, n)


class State:
    """RAII wrapper of `struct bz3_state` (bz3_new / bz3_free)."""

    def __init__(self, block_size, lib=None):
        self.lib = lib or load()
        self.block_size = block_size
        self.ptr = self.lib.bz3_new(block_size)
        if not self.ptr:
            raise RuntimeError(f"bz3_new({block_size}) returned NULL (invalid size, no HIP device, or out of device memory)")

    def close(self):
        if self.ptr:
            self.lib.bz3_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def last_error(self):
        return self.lib.bz3_last_error(self.ptr)

    def strerror(self):
        return self.lib.bz3_strerror(self.ptr).decode()

    def timings(self):
        t = (C.c_float * 8)()
        self.lib.bz3_hip_last_timings(self.ptr, t)
        return {T_NAMES[i]: t[i] for i in range(6)}

    def bwt_stats(self):
        r, p, e = C.c_int32(), C.c_int32(), C.c_uint64()
        self.lib.bz3_hip_last_bwt_stats(self.ptr, C.byref(r), C.byref(p), C.byref(e))
        return {"rounds": r.value, "radix_passes": p.value, "sorted_elements": e.value}

    # host-buffer API ------------------------------------------------------------------------
    def encode_block(self, data):
        cap = self.lib.bz3_bound(max(len(data), self.block_size)) + 64
        buf = _cbuf(data, cap)
        n = self.lib.bz3_encode_block(self.ptr, buf, len(data))
        return n, self.last_error, (C.string_at(buf, n) if n > 0 else b"")

    def decode_block(self, data, orig_size, buffer_size=None, comp_size=None):
        cap = self.lib.bz3_bound(self.block_size) + 64
        buffer_size = cap if buffer_size is None else buffer_size
        comp_size = len(data) if comp_size is None else comp_size
        buf = _cbuf(data, max(cap, buffer_size, len(data) + 1))
        n = self.lib.bz3_decode_block(self.ptr, buf, buffer_size, comp_size, orig_size)
        return n, self.last_error, (C.string_at(buf, n) if n > 0 else b"")


def encode_block(data, block_size, lib=None):
    with State(block_size, lib) as st:
        return st.encode_block(data)


def decode_block(data, orig_size, block_size, lib=None, **kw):
    with State(block_size, lib) as st:
        return st.decode_block(data, orig_size, **kw)


def shard_blocks(n_blocks, world_size, rank):
    """Block -> GPU partition of SURVEY.md section 8e: block k belongs to rank k mod world_size."""
    return [k for k in range(n_blocks) if k % world_size == rank]

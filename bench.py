#!/usr/bin/env python3
"""bench.py -- the BASELINE.json metric on MI355X:
    "MiB/s encode+decode round-trip, 256 MiB blocks, 1/2/4/8 MI355X vs CPU -j N".

One STEP = one pass of the hot path over one batch of synthetic enwik-style text blocks that are already
resident in HBM: bz3_hip_encode_blocks_device (CRC-32C -> mRLE -> LZP -> BWT -> CM coder) followed by
bz3_hip_decode_blocks_device (the inverse chain incl. the CRC check), i.e. the reference's
bz3_encode_blocks / bz3_decode_blocks (src/libbz3.c:845-870) with device pointers.  value =
input bytes of all ranks / 2^20 / (t_encode + t_decode).

Multi-GPU: blocks are independent, so each rank owns `--blocks` blocks on its own GPU (weak
scaling), there is NO data-path collective; torch.distributed (RCCL) is used only for the barrier and the
max-over-ranks timing the contract asks for.

  python bench.py                       # N=1, default workload
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
         bench.py --gpus 8 --steps 1 --warmup 0
Extra keys on the JSON line: "roofline" (dominant kernel), "cpu_baseline" (reference CPU path on this host),
"stages" (per-stage ms of one block) and "bwt_roofline" (the HBM-bound radix-sort suffix array).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # written from a rocprofv3 --pmc pass (tools/rocpd_summary.py --pmc)
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
ALG_BYTES_ROUND_TRIP = 32.4  # 17.2 B/B encode + 15.2 B/B decode
ALG_BYTES_BWT = 11.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--blocks", type=int, default=int(os.environ.get("BZ3_BENCH_BLOCKS", "256")), help="256 MiB blocks per GPU")
    ap.add_argument("--block-mib", type=float, default=float(os.environ.get("BZ3_BENCH_BLOCK_MIB", "256")))
    ap.add_argument("--kind", default="text", choices=["text", "random"])
    ap.add_argument("--cm-mode", default=os.environ.get("BZ3_BENCH_CM_MODE", "auto"), choices=["auto", "full", "rows", "rows3", "lock3", "lock2", "measured"],
                    help="CM kernel variant (bz3_hip_set_cm_mode): auto = full-model kernels; rows / rows3 = row-cache kernels, two / three blocks per CU; measured = by batch size")
    ap.add_argument("--lean", action="store_true", default=os.environ.get("BZ3_BENCH_LEAN", "0") == "1",
                    help="lean states (bz3_hip_set_lean_states): no per-state swap buffer, in-place CM encode -- room for ~3x256 blocks of 256 MiB")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mib", type=float, default=32.0)
    ap.add_argument("--cpu-threads", type=int, default=64)
    return ap.parse_args()


T_START = time.perf_counter()


def progress(msg):
    """Milestones on stderr (stdout carries only the JSON line)."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.perf_counter() - T_START:8.1f}s] {msg}", file=sys.stderr, flush=True)


def gen_text_device(torch, nbytes, seed, device, piece=32 << 20):
    """`nbytes` of synthetic text, generated in pieces of at most 32 MiB (bounded temporaries), each piece its own seed."""
    parts = []
    done = 0
    while done < nbytes:
        m = min(piece, nbytes - done)
        parts.append(gen_text_piece(torch, m, seed * 1000 + len(parts), device))
        done += m
    return parts[0] if len(parts) == 1 else torch.cat(parts)


def gen_text_piece(torch, nbytes, seed, device):
    """Word-bigram Markov text over shakespeare.txt tokens (tests/datagen.py), generated on the GPU:
    `chains` independent chains advance in lockstep; their words are laid out chain after chain."""
    import datagen

    t = datagen.bigram_tables()
    dev = {k: torch.as_tensor(v, device=device) for k, v in t.items() if hasattr(v, "shape")}
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    chains = 1 << 16
    avg = float(t["lens"][t["succ"]].mean())  # frequency-weighted word length (+1 space)
    steps = int(nbytes / (avg * chains) * 1.15) + 16
    state = torch.randint(0, t["nvocab"], (chains,), generator=g, device=device)
    toks = torch.empty((steps, chains), dtype=torch.int64, device=device)
    nsucc = dev["succ"].numel()
    for s in range(steps):
        r = torch.randint(0, 1 << 30, (chains,), generator=g, device=device)
        state = dev["succ"][(dev["start"][state] + r % dev["counts"][state]) % nsucc]
        toks[s] = state
    toks = toks.t().reshape(-1)
    lens = dev["lens"][toks]
    ends = torch.cumsum(lens, 0)
    keep = int(torch.searchsorted(ends, torch.tensor([nbytes], device=device)).item()) + 1
    toks, lens, ends = toks[:keep], lens[:keep], ends[:keep]
    total = int(ends[-1].item())
    assert total >= nbytes, "generator came up short"
    tok_of_byte = torch.repeat_interleave(torch.arange(keep, device=device), lens)
    within = torch.arange(total, device=device) - (ends - lens)[tok_of_byte]
    out = dev["blob"][dev["off"][toks[tok_of_byte]] + within][:nbytes].contiguous()
    return out


def cpu_baseline(sample_blocks, block_size):
    """The REAL reference (oracle/_ref/libbz3ref.so, kind 'reference') -- or, if it did not travel, the plain-C
    oracle (kind 'port', 1 core) -- timed on this host over a bounded sample of the same workload, through the
    reference's own batch API (one pthread per block, src/libbz3.c:845-870)."""
    from oracle_lib import Oracle, RefLib

    ref = RefLib()
    n = len(sample_blocks)
    total = sum(len(b) for b in sample_blocks)
    if ref.available:
        L = ref.lib
        cap = L.bz3_bound(block_size) + 64
        states = (C.c_void_p * n)(*[L.bz3_new(block_size) for _ in range(n)])
        bufs = [(C.c_uint8 * cap)() for _ in range(n)]
        for b, d in zip(bufs, sample_blocks):
            C.memmove(b, d, len(d))
        ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
        sizes = (C.c_int32 * n)(*[len(d) for d in sample_blocks])
        t0 = time.perf_counter()
        L.bz3_encode_blocks(states, ptrs, sizes, n)
        t1 = time.perf_counter()
        bsz = (C.c_size_t * n)(*[cap] * n)
        orig = (C.c_int32 * n)(*[len(d) for d in sample_blocks])
        L.bz3_decode_blocks(states, ptrs, bsz, sizes, orig, n)
        t2 = time.perf_counter()
        ok = all(L.bz3_last_error(states[i]) == 0 for i in range(n)) and all(bytes(bufs[i][: len(d)]) == d for i, d in enumerate(sample_blocks))
        for s in states:
            L.bz3_free(s)
        assert ok, "reference round trip failed"
        return {"value": round(total / 2 ** 20 / (t2 - t0), 3), "unit": "MiB/s", "cores": n, "kind": "reference",
                "sample": f"{n} x {len(sample_blocks[0]) / 2 ** 20:.0f} MiB text blocks, bz3_encode_blocks+bz3_decode_blocks of gcc -O2 reference, "
                          f"enc {t1 - t0:.2f}s dec {t2 - t1:.2f}s"}
    o = Oracle()
    d = sample_blocks[0][: 4 << 20]
    t0 = time.perf_counter()
    n_enc, err, blk = o.encode_block(d, max(len(d), 65 * 1024))
    k, err2, back = o.decode_block(blk, len(d), max(len(d), 65 * 1024))
    t1 = time.perf_counter()
    assert err == 0 and err2 == 0 and back == d
    return {"value": round(len(d) / 2 ** 20 / (t1 - t0), 3), "unit": "MiB/s", "cores": 1, "kind": "port",
            "sample": "one 4 MiB text block through oracle/bz3_oracle.c (oracle/_ref absent)"}


def main():
    a = parse()
    import torch  # first: the HIP runtime of the process must be torch's (see bzip3_amd._share_hip_runtime_with_torch)
    import torch.distributed as dist

    import bzip3_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)

    lib = bzip3_amd.load()
    assert lib.bz3_hip_device_count() > 0, "no HIP device"
    assert lib.bz3_hip_bind_device(local_rank) == 0
    assert lib.bz3_hip_set_cm_mode({"auto": -1, "full": 0, "rows": 1, "rows3": 2, "lock3": 3, "lock2": 4, "measured": 100}[a.cm_mode]) == 0
    assert lib.bz3_hip_set_lean_states(1 if a.lean else 0) == 0

    block_size = int(a.block_mib * (1 << 20))
    nblk = a.blocks
    cap = lib.bz3_bound(block_size) + 4096

    # ---- synthetic input, resident in HBM ----------------------------------------------------------------
    t_gen = time.perf_counter()
    if a.kind == "text":
        base = gen_text_device(torch, block_size, seed=1 + rank, device=device)
    else:
        g = torch.Generator(device=device)
        g.manual_seed(2 + rank)
        base = torch.randint(0, 256, (block_size,), dtype=torch.uint8, generator=g, device=device)
    bufs, sums = [], []
    chunk = 1 << 16
    nchunks = block_size // chunk
    for k in range(nblk):
        buf = torch.empty(cap, dtype=torch.uint8, device=device)
        if k == 0 or nchunks < 2:
            buf[:block_size] = base
        else:  # every further block: the same text with its 64 KiB pieces in a block-specific order
            g = torch.Generator(device=device)
            g.manual_seed(1000 * (rank + 1) + k)
            perm = torch.randperm(nchunks, generator=g, device=device)
            buf[: nchunks * chunk] = base[: nchunks * chunk].view(nchunks, chunk)[perm].reshape(-1)
            buf[nchunks * chunk : block_size] = base[nchunks * chunk :]
        bufs.append(buf)
        sums.append(int(buf[:block_size].to(torch.int64).sum().item()))
    probe = bufs[0][: 1 << 16].clone()
    del base
    torch.cuda.synchronize()
    torch.cuda.empty_cache()  # hand the generator's temporaries back: the codec workspace is hipMalloc'ed outside torch
    t_gen = time.perf_counter() - t_gen
    progress(f"{nblk} x {block_size} B input blocks resident in HBM ({t_gen:.1f}s)")

    states = (C.c_void_p * nblk)(*[lib.bz3_new(block_size) for _ in range(nblk)])
    assert all(states), "bz3_new failed"
    progress("states created")
    ptrs = (C.c_void_p * nblk)(*[b.data_ptr() for b in bufs])
    bsz = (C.c_size_t * nblk)(*[cap] * nblk)
    orig = (C.c_int32 * nblk)(*[block_size] * nblk)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    comp_total = [0]
    stage = {}

    def one_step(record=False):
        sizes = (C.c_int32 * nblk)(*[block_size] * nblk)
        t0 = time.perf_counter()
        lib.bz3_hip_encode_blocks_device(states, ptrs, sizes, nblk)
        t1 = time.perf_counter()
        progress(f"encode_blocks done in {t1 - t0:.1f}s")
        for i in range(nblk):
            assert sizes[i] > 0 and lib.bz3_last_error(states[i]) == 0, f"encode failed on block {i}"
        if record:
            comp_total[0] = sum(sizes)
            tm = (C.c_float * 8)()
            lib.bz3_hip_last_timings(states[0], tm)
            stage["enc"] = {bzip3_amd.T_NAMES[j]: round(tm[j], 3) for j in range(6)}
            r, p, e = C.c_int32(), C.c_int32(), C.c_uint64()
            lib.bz3_hip_last_bwt_stats(states[0], C.byref(r), C.byref(p), C.byref(e))
            stage["bwt"] = {"rounds": r.value, "radix_passes": p.value, "sorted_elements": e.value}
        t2 = time.perf_counter()
        lib.bz3_hip_decode_blocks_device(states, ptrs, bsz, sizes, orig, nblk)
        t3 = time.perf_counter()
        progress(f"decode_blocks done in {t3 - t2:.1f}s")
        for i in range(nblk):
            assert lib.bz3_last_error(states[i]) == 0, f"decode failed on block {i}"
        if record:
            tm = (C.c_float * 8)()
            lib.bz3_hip_last_timings(states[0], tm)
            stage["dec"] = {bzip3_amd.T_NAMES[j]: round(tm[j], 3) for j in range(6)}
            stage["t_enc_s"] = round(t1 - t0, 3)
            stage["t_dec_s"] = round(t3 - t2, 3)

    for _ in range(a.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for k in range(a.steps):
        one_step(record=(k == a.steps - 1))
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- the round trip must be the identity (decode is in place: the buffers hold the plaintext again) ----
    for k in range(nblk):
        assert int(bufs[k][:block_size].to(torch.int64).sum().item()) == sums[k], f"block {k}: round trip changed the data"
    assert torch.equal(bufs[0][: 1 << 16], probe)

    total_bytes = world * nblk * block_size
    value = total_bytes * a.steps / 2 ** 20 / elapsed
    out = None
    if rank == 0:
        n_dec = block_size  # n' ~ n for text
        cm_dec_ms = stage["dec"]["cm"]
        cm_bytes = n_dec + comp_total[0] / nblk  # CM decode kernel: reads the coded bytes, writes n' bytes
        bwt_ms = stage["enc"]["bwt"]
        # HBM traffic of the dominant kernel from the PMC pass (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc runs,
        # corrected as MI355X_MICROARCH.md prescribes), recorded per decoded byte and scaled to this launch.
        traffic = None
        try:
            with open(PMC_TRAFFIC_FILE) as fh:
                pm = json.load(fh)["k_cm_decode"]
            traffic = int(pm["fetch_bytes_per_coded_byte"] * comp_total[0] + pm["write_bytes_per_decoded_byte"] * n_dec * nblk)
        except Exception:
            pass
        out = {
            "metric": "MiB/s encode+decode round-trip, 256 MiB blocks",
            "value": round(value, 3),
            "unit": "MiB/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 1),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": f"{nblk} x {a.block_mib:g} MiB synthetic enwik-style text blocks per GPU (word-bigram Markov over shakespeare.txt tokens; "
                            f"blocks 2.. are 64 KiB-piece permutations of block 1), resident in HBM, "
                            f"bz3_hip_encode_blocks_device + bz3_hip_decode_blocks_device",
                "block_bytes": block_size,
                "blocks_per_gpu": nblk,
                "parallelism": f"blocks sharded over {world} GPU(s), no collective",
                "compressed_ratio": round(world * 0 + (nblk * block_size) / max(1, comp_total[0]), 3),
                "cm_mode": a.cm_mode,
                "lean_states": bool(a.lean),
                "cm_blocks_given_up": int(lib.bz3_hip_cm_blocks_given_up()),
            },
            # dominant kernel by time: the CM decoder (one workgroup per block; a serial integer recurrence,
            # latency-bound by construction -- SURVEY.md 7/H1), priced against the HBM roofline as the contract asks
            "roofline": {
                "kernel": "k_cm_decode",
                "bound": "hbm",
                "achieved": round(cm_bytes * nblk / (cm_dec_ms * 1e-3) / 1e9, 6),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(cm_bytes * nblk / (cm_dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 9),
                "traffic": traffic,
                "launch_ms": cm_dec_ms,
                "algorithmic_bytes_per_launch": int(cm_bytes * nblk),
            },
            "path_roofline": {
                "achieved": round(ALG_BYTES_ROUND_TRIP * total_bytes * a.steps / elapsed / 1e9, 4),
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "bytes_per_input_byte": ALG_BYTES_ROUND_TRIP,
            },
            "bwt_roofline": {
                "stage": "bwt_forward (radix-sort prefix doubling), one block",
                "achieved": round(ALG_BYTES_BWT * block_size / (bwt_ms * 1e-3) / 1e9, 3),
                "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(ALG_BYTES_BWT * block_size / (bwt_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6),
                "ms": bwt_ms, **stage.get("bwt", {}),
            },
            "stages": stage,
            "gen_s": round(t_gen, 1),
        }
        if not a.no_cpu_baseline:
            try:
                ncores = max(1, min(os.cpu_count() or 1, a.cpu_threads))
                smp = int(a.cpu_sample_mib * (1 << 20))
                smp = min(smp, block_size)
                host = bufs[0][:block_size].cpu().numpy().tobytes()
                sample = [host[(i * smp) % max(1, block_size - smp + 1):][:smp] for i in range(ncores)]
                out["cpu_baseline"] = cpu_baseline(sample, max(smp, 65 * 1024))
            except Exception as e:  # the baseline is reported, never required
                out["cpu_baseline"] = {"value": None, "unit": "MiB/s", "cores": 0, "kind": "reference", "sample": f"failed: {e}"}
    for s in states:
        lib.bz3_free(s)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()

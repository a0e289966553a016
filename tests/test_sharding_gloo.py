"""CPU test of the N>1 path: blocks are sharded over ranks (block k -> rank k mod N), every rank
codes its own blocks with no data-path collective, the host gathers the chunks in order.  world_size 2, gloo.
The codec inside each rank is the kernel sources under the test-only HIP emulation (tests/emu) -- on the real machine
it is the GPU the rank is bound to; the partition / gather / timing-reduction logic under test is identical."""
import os
import socket
import struct
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, block_size, blocks, out_q):
    sys.path[:0] = [ROOT, HERE, os.path.join(HERE, "emu")]
    import ctypes as C

    import bzip3_amd
    from build_emu import build

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = bzip3_amd._declare(C.CDLL(build()))
    mine = bzip3_amd.shard_blocks(len(blocks), world, rank)
    dist.barrier()
    t0 = torch.tensor([0.0], dtype=torch.float64)
    import time

    t = time.perf_counter()
    coded = {}
    with bzip3_amd.State(block_size, lib) as st:
        for k in mine:
            n, err, blk = st.encode_block(blocks[k])
            assert err == 0
            coded[k] = (blk, len(blocks[k]))
            m, err, back = st.decode_block(blk, len(blocks[k]))
            assert (m, err) == (len(blocks[k]), 0) and back == blocks[k]
    t0[0] = time.perf_counter() - t
    dist.barrier()
    dist.all_reduce(t0, op=dist.ReduceOp.MAX)  # the bench's max-over-ranks time
    gathered = [None] * world
    dist.gather_object(coded, gathered if rank == 0 else None, dst=0)
    if rank == 0:
        merged = {}
        for g in gathered:
            merged.update(g)
        out = b"BZ3v1" + struct.pack("<I", block_size)
        for k in range(len(blocks)):
            blk, orig = merged[k]
            out += struct.pack("<II", len(blk), orig) + blk
        out_q.put((out, float(t0[0]), sorted(merged)))
    dist.destroy_process_group()


def test_two_rank_block_sharding_matches_single_process_output(oracle):
    import bzip3_amd
    import datagen

    t = datagen.shakespeare()
    block_size = 65 * 1024
    blocks = [t[i * 2500 : i * 2500 + 2400 + 37 * i] for i in range(5)] + [b"tail"]
    for w in (1, 2, 3):
        parts = [bzip3_amd.shard_blocks(len(blocks), w, r) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(len(blocks))) and all(len(set(p)) == len(p) for p in parts)
    expect = b"BZ3v1" + struct.pack("<I", block_size)
    for b in blocks:
        n, err, blk = oracle.encode_block(b, block_size)
        expect += struct.pack("<II", len(blk), len(b)) + blk
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, block_size, blocks, q)) for r in range(2)]
    for p in procs:
        p.start()
    out, tmax, keys = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert keys == list(range(len(blocks)))
    assert out == expect  # byte-identical to the single-process (oracle) stream, chunk order preserved
    assert tmax > 0

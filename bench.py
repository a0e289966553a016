#!/usr/bin/env python3
"""bench.py -- the BASELINE.json metric on MI355X:
    "MiB/s encode+decode round-trip, 256 MiB blocks, 1/2/4/8 MI355X vs CPU -j N".

One STEP = one pass of the hot path over one batch of synthetic enwik-style text blocks that are already
resident in HBM: bz3_hip_encode_blocks_device (CRC-32C -> mRLE -> LZP -> BWT -> CM coder) followed by
bz3_hip_decode_blocks_device (the inverse chain incl. the CRC check), i.e. the reference's
bz3_encode_blocks / bz3_decode_blocks (src/libbz3.c:845-870) with device pointers.  value =
input bytes of all ranks / 2^20 / (t_encode + t_decode).

Wall budget.  One step over a GPU-filling batch of 256 MiB blocks takes minutes (the CM coder is a serial
recurrence per block), so the requested --steps / --warmup are CLAMPED to what fits the wall budget
(--budget-s, default 1500 s of the driver's 1800 s): at least one timed step always runs; a warmup step is
only spent when a second step still fits.  The JSON line carries the steps / warmup actually run and the
requested ones.  A watchdog thread prints the line with whatever has been measured so far if the hard
deadline (--deadline-s) is reached while an optional extra leg is still running.

Two timed steps when the budget allows (both reported, with their spread; the first includes the first-call effects:
workspace allocation).  NOTHING else runs on the host or the GPU during the timed steps (round 3 ran the CPU baseline as
threads of this process during the second step, which slowed that step's front end 2.6x: profiles/r04_host_contention.json).

"cpu_baseline": the REAL reference (oracle/_ref) through its own bz3_encode_blocks / bz3_decode_blocks on min(cores, 64)
host threads x 256 MiB blocks -- the same blocks the GPU coded, and the reference's coded bytes are compared with the GPU's
(bit-exact parity at the metric's block size).  It runs AFTER the timed steps in a PROCESS OF ITS OWN (this file with
--cpu-worker; no torch, no HIP runtime), pinned to cores this process does not use, the blocks handed over in anonymous
shared memory (memfd); meanwhile this process only runs GPU legs that need no host cores.  64 threads because the
reference's own CLI caps -j at 64 (src/main.c:213) and its batch API is documented for 2-16 blocks (libbz3.h:202).

Workload: "enwik-style" text = the word-bigram Markov chain over shakespeare.txt tokens of SURVEY.md 8d with 3.5 % of its
tokens replaced by random [a-z0-9] strings (tests/datagen.py ENWIK_NOISE: the fraction at which 100,000,000 B compress to
enwik8's 4.41 : 1 at the reference's default -b 16, etc/BENCHMARKS.md:45-47).  Eight independent 256 MiB texts are
generated; block k is text k mod 8 with its 64 KiB pieces in a block-specific order.

Extra legs after the timed region (rank 0, N=1 only, each only while the budget lasts), all in "configs":
  random      768 incompressible blocks (LZP and RLE decline, the coder emits ~1.004 bytes per byte), reference beside it;
  mixed       text, binary and incompressible blocks in ONE batch (the single CM launch lasts as long as its slowest
              block, and blocks the row-cache kernels give up are coded again: cm_blocks_given_up is reported);
  cfg5_unbwt  BASELINE.json configs[4]'s stage: the inverse BWT of one 511 MiB block (the maximum block size) of a
              skewed order-1 Markov source over 16 symbols, GB/s against the stage's 11 B per byte;
  cfg3        BASELINE.json configs[2]: 1,000,000,000 B of text at -b 256 = 4 blocks on one GPU, with the
              reference's -j 4 path timed on the host beside it;
  cfg2        BASELINE.json configs[1]: 100,000,000 B at -b 32 = 3 blocks, with -j 3 beside it;
  host_api    SURVEY.md 8d's timing boundary: a GPU-filling batch through bz3_encode_blocks / bz3_decode_blocks on malloc'ed
              HOST buffers (H2D / D2H included), and the same batch device-resident: the PCIe-inclusive ratio.
`--leg cfg5` (not part of the default run: a 511 MiB block's CM launches take twice as long as a 256 MiB block's) times a
GPU-filling batch of 511 MiB blocks of the 16-symbol source.

Multi-GPU: blocks are independent, so each rank owns `--blocks` blocks on its own GPU (weak
scaling), there is NO data-path collective; torch.distributed (RCCL) is used only for the barrier and the
max-over-ranks timing the contract asks for.  `python bench.py --gpus N` with no WORLD_SIZE in the environment starts its
own N ranks (torch.distributed.run on 127.0.0.1); under torch.distributed.run it is one of the ranks:

  python bench.py                       # N=1, default workload
  python bench.py --gpus 8 --steps 1 --warmup 0          # spawns 8 ranks itself
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
         bench.py --gpus 8 --steps 1 --warmup 0          # the driver's form
"""
import argparse
import ctypes as C
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # written from a rocprofv3 --pmc pass (tools/rocpd_summary.py --pmc)
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
ALG_BYTES_ROUND_TRIP = 32.4  # 17.2 B/B encode + 15.2 B/B decode
ALG_BYTES_BWT = 11.0
CFG3_BYTES = 1_000_000_000  # BASELINE.json configs[2]: "enwik9 (1 GB), -b 256, single MI355X"
CM_MODES = {"auto": -1, "full": 0, "rows": 1, "rows3": 2}

# More hardware queues than the HIP runtime's default 4: the codec's rings use the group's stream + four side streams (api.hip device_count()
# explains; it sets the same default when it initialises the runtime itself -- here torch does, so it has to be in the environment before that).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

# runs of their own whose last record rides along under config.recorded (never under config.legs)
RECORDED_RUNS = {"cfg5_256x511MiB_round_trip": "r06_bench_cfg5_256x511MiB.json", "random_256x256MiB_round_trip": "r06_bench_random_256x256MiB.json"}
PHASE = {"name": "start-up", "fatal": True}  # where an exception would come from; fatal: the headline is not valid without this phase

T_START = time.perf_counter()
RANK = int(os.environ.get("RANK", "0"))
WORLD = int(os.environ.get("WORLD_SIZE", "1"))
RESULT = {"line": None, "printed": False}  # the JSON line as far as it has been measured (rank 0)
_PRINT_LOCK = threading.Lock()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--blocks", type=int, default=int(os.environ.get("BZ3_BENCH_BLOCKS", "0")),
                    help="blocks per GPU (0 = default for the CM mode: one block per CU, three with the row-cache kernels + lean states)")
    ap.add_argument("--block-mib", type=float, default=float(os.environ.get("BZ3_BENCH_BLOCK_MIB", "256")))
    ap.add_argument("--kind", default="text", choices=["text", "random"])
    ap.add_argument("--cm-mode", default=os.environ.get("BZ3_BENCH_CM_MODE", "auto"), choices=sorted(CM_MODES),
                    help="CM kernel variant (bz3_hip_set_cm_mode): auto = the library's policy; full = whole model in LDS, one block per CU; "
                         "rows / rows3 = row-cache kernels, two / three blocks per CU")
    ap.add_argument("--lean", type=int, default=int(os.environ.get("BZ3_BENCH_LEAN", "-1")),
                    help="lean states (bz3_hip_set_lean_states): 1 = no per-state swap buffer, in-place CM encode (room for 3x256 blocks of 256 MiB); -1 = by block count")
    ap.add_argument("--budget-s", type=float, default=float(os.environ.get("BZ3_BENCH_BUDGET_S", "1500")),
                    help="wall budget from process start: steps/warmup are clamped and the extra legs skipped so that the run ends before it")
    ap.add_argument("--deadline-s", type=float, default=float(os.environ.get("BZ3_BENCH_DEADLINE_S", "1690")),
                    help="hard deadline: the watchdog prints the JSON line as measured so far and exits")
    ap.add_argument("--lib", default=None, help="another build of libbzip3.so (same-box A/B of two code states; never used by the driver)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the cfg3 / random legs")
    ap.add_argument("--cpu-threads", type=int, default=64)
    ap.add_argument("--cpu-block-mib", type=float, default=0.0, help="CPU baseline block size (0 = the bench's block size)")
    ap.add_argument("--cfg3-bytes", type=int, default=CFG3_BYTES, help="size of the cfg3 leg (only BASELINE's 1,000,000,000 B at 256 MiB blocks is reported as cfg3)")
    ap.add_argument("--random-block-mib", type=float, default=8.0)
    ap.add_argument("--random-blocks", type=int, default=0, help="blocks of the random leg (0 = as many as the timed batch)")
    ap.add_argument("--host-api-block-mib", type=float, default=32.0, help="block size of the host_api leg (0 = skip the leg)")
    ap.add_argument("--noise", type=float, default=-1.0, help="fraction of noise tokens in the text (-1 = tests/datagen.py ENWIK_NOISE, the enwik8 calibration)")
    ap.add_argument("--text-bases", type=int, default=8, help="independent texts the blocks are drawn from")
    ap.add_argument("--legs", default="", help="comma-separated subset of the extra legs to run (random, mixed, cfg5_unbwt, cfg3, cfg2, host_api; default: all)")
    ap.add_argument("--leg", default="", choices=["", "cfg5"], help="run ONE optional leg instead of the default workload (cfg5: a GPU-filling batch of 511 MiB blocks)")
    ap.add_argument("--leg-block-mib", type=float, default=511.0, help="block size of --leg cfg5 (511 = BASELINE's; smaller values are for tests)")
    ap.add_argument("--emu", action="store_true", help="TESTS ONLY: the CPU emulator build of the kernels (tests/emu) and gloo instead of a GPU and RCCL; checks the control flow, measures nothing")
    ap.add_argument("--cpu-worker", default="", help="internal: run the reference on the blocks described by this JSON file (the cpu_baseline process)")
    return ap.parse_args()


def spawn_ranks(a):
    """`--gpus N` without torch.distributed.run around it: start the N ranks ourselves (one process per GPU, 127.0.0.1 rendezvous)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def elapsed():
    return time.perf_counter() - T_START


def plan_steps(left_s, t_first_s, req_steps, req_warmup):
    """How to spend what is left of the wall budget after the first step has run (it took t_first_s).  Returns (warmup, more):
    `warmup` = untimed steps in total, the first step included when it is one (0: the first step is the first timed step);
    `more` = timed steps still to run.  At least one timed step always exists; a warmup is only spent when a second step still
    fits; further warmups never displace requested timed steps."""
    afford = int(max(0.0, left_s) // (t_first_s * 1.03)) if t_first_s > 0 else 0  # further steps that still fit
    if afford < 1 or (req_warmup <= 0 and req_steps <= 1):
        return 0, 0
    if req_warmup > 0:
        warmup = 1 + max(0, min(req_warmup - 1, afford - req_steps))
        more = max(1, min(req_steps, afford - (warmup - 1)))
        return warmup, more
    return 0, max(0, min(req_steps - 1, afford))


def inject(point):
    """TESTS ONLY (tests/test_bench_ranks.py): BZ3_BENCH_INJECT_FAIL=<point>[:assert] raises at the named point, to exercise the exit paths."""
    want = os.environ.get("BZ3_BENCH_INJECT_FAIL", "")
    if want.split(":")[0] == point:
        if want.endswith(":assert"):
            raise AssertionError(f"injected assertion at {point}")
        raise RuntimeError(f"injected failure at {point}")


def progress(msg):
    """Milestones on stderr (stdout carries only the JSON line)."""
    if RANK == 0:
        print(f"[bench {elapsed():8.1f}s] {msg}", file=sys.stderr, flush=True)


def emit_line(final):
    """Print the JSON line once (rank 0)."""
    with _PRINT_LOCK:
        if RESULT["printed"] or RESULT["line"] is None:
            return False
        RESULT["line"]["complete"] = bool(final)
        RESULT["line"]["wall_s"] = round(elapsed(), 1)
        sys.stdout.write(json.dumps(RESULT["line"]) + "\n")
        sys.stdout.flush()
        RESULT["printed"] = True
        return True


def start_watchdog(deadline_s):
    """The C calls of a step cannot be interrupted from Python (signal handlers only run between bytecodes), but ctypes
    releases the GIL, so a thread can: at the hard deadline it prints what has been measured and ends the process."""

    def run():
        while elapsed() < deadline_s:
            time.sleep(min(5.0, max(0.1, deadline_s - elapsed())))
        if RANK == 0:
            if emit_line(final=False):
                progress("hard deadline: JSON line printed with the legs measured so far")
            else:
                progress("hard deadline: nothing measured yet" if RESULT["line"] is None else "hard deadline")
        os._exit(0 if RESULT["printed"] or RANK != 0 else 3)

    t = threading.Thread(target=run, daemon=True)
    t.start()

    def on_term(signum, frame):
        if RANK == 0:
            emit_line(final=False)
        os._exit(0 if RESULT["printed"] or RANK != 0 else 3)

    signal.signal(signal.SIGTERM, on_term)


# ---- synthetic data ------------------------------------------------------------------------------------------------
def gen_text_device(torch, nbytes, seed, device, piece=32 << 20, noise=0.0):
    """`nbytes` of synthetic text, generated in pieces of at most 32 MiB (bounded temporaries), each piece its own seed."""
    parts = []
    done = 0
    while done < nbytes:
        m = min(piece, nbytes - done)
        parts.append(gen_text_piece(torch, m, seed * 1000 + len(parts), device, noise))
        done += m
    return parts[0] if len(parts) == 1 else torch.cat(parts)


def gen_text_piece(torch, nbytes, seed, device, noise=0.0):
    """Word-bigram Markov text over shakespeare.txt tokens (tests/datagen.py), generated on the GPU:
    `chains` independent chains advance in lockstep; their words are laid out chain after chain.  `noise`: fraction of the tokens
    whose letters are replaced by random [a-z0-9] (datagen.text's noise: the enwik8 calibration)."""
    import datagen

    t = datagen.bigram_tables()
    dev = {k: torch.as_tensor(v, device=device) for k, v in t.items() if hasattr(v, "shape")}
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    chains = 1 << 16
    while chains > 16 and nbytes < chains * 64:  # (tiny pieces: fewer chains, so that every chain still emits a few words)
        chains >>= 1
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
    out = dev["blob"][dev["off"][toks[tok_of_byte]] + within]
    if noise > 0:
        noisy = torch.rand((keep,), generator=g, device=device) < noise
        mask = noisy[tok_of_byte] & (within < lens[tok_of_byte] - 1)  # the token's letters, not its space
        alphabet = torch.as_tensor(datagen.NOISE_ALPHABET.copy(), device=device)
        out = torch.where(mask, alphabet[torch.randint(0, alphabet.numel(), (total,), generator=g, device=device)], out)
    return out[:nbytes].contiguous()


def seed_text_source(lib):
    """The Markov generator's token table comes from the plaintext of the reference's fixture (tests/golden/shakespeare.txt.bz3).
    tests/datagen.py recovers it with the CPU checker; here the PRODUCT decodes it (two blocks through bz3_decode_block), so that
    nothing of bench.py's workload depends on oracle/ -- only the cpu_baseline leg uses the checker, as a baseline."""
    import hashlib

    import bzip3_amd
    import datagen

    if "txt" in datagen._cache:
        return
    try:
        raw = open(os.path.join(datagen.GOLDEN, "shakespeare.txt.bz3"), "rb").read()
        bs, chunks = datagen.parse_chunks(raw)
        parts = []
        for comp, orig, blk in chunks:
            n, err, dec = bzip3_amd.decode_block(blk, orig, bs, lib)
            assert n == orig and err == 0
            parts.append(dec)
        data = b"".join(parts)
        assert hashlib.md5(data).hexdigest() == datagen.SHAKESPEARE_MD5
        datagen._cache["txt"] = data
    except Exception as e:  # fall back to datagen's own path
        progress(f"fixture decode through the product failed ({e}); tests/datagen.py recovers the text instead")


def fingerprint(torch, t):
    """Position-sensitive 64-bit fingerprint of a uint8 tensor (a permuted or shifted block does not pass):
    sum over 32-bit words of word * (1 + index mod 1000003), in chunks of 64 MiB."""
    n4 = t.numel() // 4
    acc = 0
    step = 16 << 20  # words per chunk
    w = t[: n4 * 4].view(torch.int32)
    for o in range(0, n4, step):
        m = min(step, n4 - o)
        idx = torch.arange(o, o + m, device=t.device, dtype=torch.int64) % 1000003 + 1
        acc = (acc + int((w[o : o + m].to(torch.int64) * idx).sum().item())) & 0xFFFFFFFFFFFFFFFF
    tail = t[n4 * 4 :]
    if tail.numel():
        acc = (acc + int(tail.to(torch.int64).sum().item()) * 7919) & 0xFFFFFFFFFFFFFFFF
    return acc


# ---- the reference on the host (cpu_baseline) -----------------------------------------------------------------------
def cpu_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return model, os.cpu_count() or 1


def cpu_quota():
    """CPUs the container may use at once (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited: os.cpu_count() does not know it."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, p = fh.read().split()
            return None if q == "max" else round(int(q) / int(p), 2)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / p, 2)
    except (OSError, ValueError):
        return None


def effective_cpus(threads):
    """CPUs that `threads` busy threads really get: the thread count, capped by the cgroup quota and the logical CPUs."""
    q = cpu_quota()
    cap = min(float(os.cpu_count() or 1), q if q else float("inf"))
    e = min(float(threads), cap)
    return int(e) if e == int(e) else round(e, 2)


def sharding_model(world):
    """Block sharding quantises (block k -> GPU k mod G, a GPU's time = its block count x one block's latency when the
    blocks are fewer than fill it): predicted efficiency of BASELINE's multi-GPU configs = blocks / (G x ceil(blocks / G))."""
    def eff(blocks, g):
        return round(blocks / (g * -(-blocks // g)), 4)
    return {"this_run": f"weak scaling: every rank codes its own GPU-filling batch, so the predicted efficiency at {world} GPU(s) is 1.0 (no collective, no shared resource but the host)",
            "cfg4_linux_tarball_5_blocks_8_gpus": eff(5, 8), "cfg5_8GiB_17_blocks_8_gpus": eff(17, 8),
            "note": "few-block inputs are bound by ONE block's CM latency on one CU whatever the GPU count: 5 blocks on 8 GPUs = 62.5 %, 17 on 8 = 70.8 % (17 / 24)"}


def host_mem_available():
    try:
        with open("/proc/meminfo") as fh:
            for ln in fh:
                if ln.startswith("MemAvailable"):
                    return int(ln.split()[1]) * 1024
    except OSError:
        pass
    return 8 << 30


REF_VARIANTS = (("gcc -O2", "libbz3ref.so"), ("gcc -O2 -march=x86-64-v3", "libbz3ref_v3.so"), ("clang -O3 -march=x86-64-v3", "libbz3ref_clang.so"),
                ("gcc -O2 -march=znver3", "libbz3ref_znver3.so"), ("clang -O3 -march=znver5", "libbz3ref_clang_znver5.so"))

_PROBE_CODE = r"""
import sys, time
sys.path[:0] = [%r, %r]
from oracle_lib import Bz3, RefLib
r = RefLib(sys.argv[1])
assert r.available
b = Bz3(r.lib)
sample = open(sys.argv[2], "rb").read()
bs = max(len(sample), 65 * 1024)
t0 = time.perf_counter()
n_enc, err, blk = b.encode_block(sample, bs)
k, err2, back = b.decode_block(blk, len(sample), bs)
dt = time.perf_counter() - t0
assert not err and not err2 and back == sample
print(dt)
"""


def fastest_reference(sample):
    """(label, path, probe record) of the reference build that round-trips an 8 MiB sample fastest on ONE thread of this host.
    SURVEY.md 8d asks for gcc -O2 and clang -O3 with -march=native; the reference tree is not on the GPU box, so the variants are
    compiled in the build container (oracle/Makefile): plain, x86-64-v3, and -- round 6 -- what -march=native means on the GPU boxes'
    host (gcc -march=znver3, the newest this gcc knows; AMD clang -march=znver5).  Every variant is probed in a PROCESS OF ITS OWN: a build
    that uses an instruction this host lacks dies there (SIGILL) and is simply not a candidate."""
    import tempfile

    from oracle_lib import ORACLE_DIR

    probe, best = {}, None
    with tempfile.NamedTemporaryFile(suffix=".bin", prefix="bz3_probe_") as f:
        f.write(sample)
        f.flush()
        for label, name in REF_VARIANTS:
            path = os.path.join(ORACLE_DIR, "_ref", name)
            if not os.path.exists(path):
                continue
            try:
                r = subprocess.run([sys.executable, "-c", _PROBE_CODE % (ROOT, os.path.join(ROOT, "tests")), path, f.name], capture_output=True, text=True, timeout=120)
                if r.returncode != 0:
                    probe[label] = f"does not run here (exit {r.returncode})"
                    continue
                dt = float(r.stdout.strip().splitlines()[-1])
            except Exception as e:
                probe[label] = f"probe failed: {type(e).__name__}"
                continue
            probe[label] = round(len(sample) / 2 ** 20 / dt, 3)
            if best is None or dt < best[2]:
                best = (label, path, dt)
    return (best[0], best[1], probe) if best else (None, None, probe)


def reference_round_trip(sample_blocks, block_size, keep_encoded=False, lib_path=None, label="gcc -O2"):
    """The REAL reference (oracle/_ref/libbz3ref.so, kind 'reference') through its own batch API, one pthread per block
    (src/libbz3.c:845-870), on this host.  Returns (record, encoded blocks or None).  Falls back to the plain-C oracle
    (kind 'port', 1 core, 4 MiB) only if oracle/_ref did not travel."""
    from oracle_lib import Oracle, RefLib

    ref = RefLib(lib_path)
    n = len(sample_blocks)
    total = sum(len(b) for b in sample_blocks)
    model, ncpu = cpu_info()
    if ref.available:
        L = ref.lib
        cap = L.bz3_bound(block_size) + 64
        states = (C.c_void_p * n)(*[L.bz3_new(block_size) for _ in range(n)])
        assert all(states), "reference bz3_new failed (host memory?)"
        bufs = [(C.c_uint8 * cap)() for _ in range(n)]
        for b, d in zip(bufs, sample_blocks):
            C.memmove(b, d.ctypes.data if hasattr(d, "ctypes") else d, len(d))
        ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
        sizes = (C.c_int32 * n)(*[len(d) for d in sample_blocks])
        t0 = time.perf_counter()
        L.bz3_encode_blocks(states, ptrs, sizes, n)
        t1 = time.perf_counter()
        enc = [C.string_at(bufs[i], sizes[i]) for i in range(n)] if keep_encoded else None
        bsz = (C.c_size_t * n)(*[cap] * n)
        orig = (C.c_int32 * n)(*[len(d) for d in sample_blocks])
        t1b = time.perf_counter()
        L.bz3_decode_blocks(states, ptrs, bsz, sizes, orig, n)
        t2 = time.perf_counter()
        import numpy as np

        ok = all(L.bz3_last_error(states[i]) == 0 for i in range(n)) and all(
            np.array_equal(np.frombuffer(bufs[i], dtype=np.uint8, count=len(d)), np.frombuffer(d, dtype=np.uint8)) for i, d in enumerate(sample_blocks))
        for s in states:
            L.bz3_free(s)
        assert ok, "reference round trip failed"
        t_enc, t_dec = t1 - t0, t2 - t1b
        v = total / 2 ** 20 / (t_enc + t_dec)
        eff = effective_cpus(n)
        # "cores" = the CPUs that did the work: n threads, but never more than the container's cgroup quota lets run at once
        # (64 threads on a 16-CPU quota are 16 CPUs' worth of the reference, not 64)
        rec = {"value": round(v, 3), "unit": "MiB/s", "cores": eff, "threads": n, "effective_cpus": eff, "MiBps_per_cpu": round(v / eff, 3), "kind": "reference",
               "sample": f"{n} x {len(sample_blocks[0]) / 2 ** 20:.0f} MiB blocks (the GPU's blocks 0..{n - 1}), one pass of bz3_encode_blocks + "
                         f"bz3_decode_blocks of the {label} reference, one thread per block",
               "t_enc_s": round(t_enc, 2), "t_dec_s": round(t_dec, 2), "host_cpu": model, "host_cores": ncpu, "cpu_quota_cpus": cpu_quota()}
        return rec, enc
    o = Oracle()
    d = bytes(sample_blocks[0][: 4 << 20])
    t0 = time.perf_counter()
    n_enc, err, blk = o.encode_block(d, max(len(d), 65 * 1024))
    k, err2, back = o.decode_block(blk, len(d), max(len(d), 65 * 1024))
    t1 = time.perf_counter()
    assert err == 0 and err2 == 0 and back == d
    return {"value": round(len(d) / 2 ** 20 / (t1 - t0), 3), "unit": "MiB/s", "cores": 1, "threads": 1, "effective_cpus": 1, "MiBps_per_cpu": round(len(d) / 2 ** 20 / (t1 - t0), 3), "kind": "port",
            "sample": "one 4 MiB text block through oracle/bz3_oracle.c (oracle/_ref absent)", "host_cpu": model, "host_cores": ncpu}, None


# ---- the reference in a process of its own ------------------------------------------------------------------------------
class SharedBlocks:
    """Blocks in anonymous shared memory (memfd: no /dev/shm size limit, no file), filled here, mapped read-only by the worker."""

    def __init__(self, name, sizes):
        import mmap

        self.sizes = list(sizes)
        self.offsets, off = [], 0
        for n in self.sizes:
            self.offsets.append(off)
            off += (n + 4095) & ~4095
        self.total = max(off, 4096)
        self.fd = os.memfd_create(name)
        os.ftruncate(self.fd, self.total)
        self.mm = mmap.mmap(self.fd, self.total)

    def put_tensor(self, torch, k, t):  # device -> shared memory, no intermediate host copy
        torch.frombuffer(self.mm, dtype=torch.uint8, count=self.sizes[k], offset=self.offsets[k]).copy_(t[: self.sizes[k]])

    def meta(self, sel=None, sizes=None):
        sel = range(len(self.sizes)) if sel is None else sel
        return {"fd": self.fd, "total": self.total, "blocks": [[self.offsets[k], (sizes[j] if sizes else self.sizes[k])] for j, k in enumerate(sel)]}

    def close(self):
        try:
            self.mm.close()
        except BufferError:
            pass
        os.close(self.fd)


def worker_cpus():
    """Cores for the reference process: all but the first ones this process may run on (the thread that drives the GPU stays there)."""
    try:
        mine = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return []
    keep = 16 if len(mine) >= 96 else (2 if len(mine) >= 8 else 0)
    return mine[keep:]


class RefWorker:
    """`python bench.py --cpu-worker <json>`: the reference's batch API on blocks in shared memory, in a process without torch or HIP."""

    def __init__(self, plain, block_size, coded=None, ref_path=None, ref_label=None, probe=False):
        import tempfile

        fds = [plain["fd"]] + ([coded["fd"]] if coded else [])
        job = {"plain": plain, "coded": coded, "block_size": block_size, "cpus": worker_cpus(), "ref_path": ref_path, "ref_label": ref_label, "probe": probe}
        f = tempfile.NamedTemporaryFile("w", suffix=".json", prefix="bz3_ref_job_", delete=False)
        json.dump(job, f)
        f.close()
        self.job_file = f.name
        self.t0 = time.perf_counter()
        self.proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", f.name], pass_fds=fds, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)

    def result(self):
        out, err = self.proc.communicate()
        try:
            os.unlink(self.job_file)
        except OSError:
            pass
        if self.proc.returncode != 0:
            return {"err": f"reference process failed ({self.proc.returncode}): {err.strip()[-300:]}"}
        rec = json.loads(out.strip().splitlines()[-1])
        rec["process_wall_s"] = round(time.perf_counter() - self.t0, 1)
        return rec


def cpu_worker(job_file):
    """The cpu_baseline process.  Nothing of the product is loaded here."""
    import mmap

    import numpy as np

    with open(job_file) as fh:
        job = json.load(fh)
    if job.get("cpus"):
        try:
            os.sched_setaffinity(0, set(job["cpus"]))
        except OSError:
            pass

    def views(m):
        mm = mmap.mmap(m["fd"], m["total"], prot=mmap.PROT_READ)
        return [np.frombuffer(mm, dtype=np.uint8, count=n, offset=off) for off, n in m["blocks"]]

    plain = views(job["plain"])
    coded = views(job["coded"]) if job.get("coded") else None
    label, path, probe = job.get("ref_label"), job.get("ref_path"), {}
    if job.get("probe") or not path:
        label, path, probe = fastest_reference(plain[0][: 8 << 20].tobytes())
        label = label or "gcc -O2"
    rec, enc = reference_round_trip(plain, max(job["block_size"], 65 * 1024), keep_encoded=coded is not None, lib_path=path, label=label)
    out = {"rec": rec, "ref_label": label, "ref_path": path, "probe": probe, "pinned_to_cpus": len(job.get("cpus") or [])}
    if enc is not None and coded is not None:
        same = sum(1 for x, y in zip(enc, coded) if len(x) == len(y) and x == y.tobytes())
        out["parity_same"], out["parity_of"] = same, len(enc)
    sys.stdout.write(json.dumps(out) + "\n")


def main():
    a = parse()
    if a.cpu_worker:
        return cpu_worker(a.cpu_worker)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))
    start_watchdog(a.deadline_s)
    try:
        run(a)
    except SystemExit:
        raise
    except BaseException as e:  # every exit path prints the JSON line (round 5's run died with a traceback and no line)
        import traceback

        traceback.print_exc()
        sys.stderr.flush()
        msg = f"{type(e).__name__}: {str(e)[:400]}"
        if RANK == 0:
            if RESULT["line"] is None:  # nothing timed yet: a line that says so
                RESULT["line"] = {"metric": "MiB/s encode+decode round-trip, 256 MiB blocks", "value": None, "unit": "MiB/s", "n_gpus": WORLD, "steps": 0, "warmup": 0,
                                  "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic", "config": {"workload": "failed before the first timed step completed"}}
            RESULT["line"]["error"] = msg
            RESULT["line"]["failed_phase"] = PHASE["name"]
            emit_line(final=False)
        # a failure in or before the timed steps or their verification invalidates the headline: non-zero exit, with the line printed
        # (an AssertionError is a correctness check that did not hold -- a block that failed to code, a round trip that changed the data: always fatal; anything
        # else after the first step has been timed AND verified leaves a valid headline: exit 0 with the error in the line)
        fatal = PHASE["fatal"] or isinstance(e, AssertionError) or RESULT["line"] is None or RESULT["line"].get("value") is None
        os._exit(1 if fatal else 0)
    if RANK == 0 and RESULT["line"] is not None and RESULT["line"].get("parity_failed"):
        os._exit(1)


def run(a):
    import torch  # first: the HIP runtime of the process must be torch's (see bzip3_amd._share_hip_runtime_with_torch)
    import torch.distributed as dist

    import bzip3_amd

    world, rank = WORLD, RANK
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}, or without it (bench.py starts the ranks itself)"
    cuda = not a.emu
    if cuda:
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    else:
        device = torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if cuda:
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend="gloo")

    def dev_sync():
        if cuda:
            torch.cuda.synchronize()

    if a.emu:  # tests only: the kernels compiled for the CPU emulator; every rank "owns" emulated device 0
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        from build_emu import build as build_emu

        lib = bzip3_amd.load(build_emu())
        local_dev = 0
    else:
        lib = bzip3_amd.load()
        if a.lib:
            lib = bzip3_amd.load(a.lib)
        local_dev = local_rank
    assert lib.bz3_hip_device_count() > 0, "no HIP device"
    assert lib.bz3_hip_bind_device(local_dev) == 0
    assert lib.bz3_hip_set_cm_mode(CM_MODES[a.cm_mode]) == 0
    per_cu = {"rows": 2, "rows3": 3, "auto": 3, "full": 1}[a.cm_mode]  # blocks per CU the mode is made for
    cus = torch.cuda.get_device_properties(device).multi_processor_count if cuda else 4
    nblk = a.blocks if a.blocks > 0 else cus * per_cu
    block_size = int(a.block_mib * (1 << 20))
    if a.leg == "cfg5":
        block_size = int(a.leg_block_mib * (1 << 20))
        if a.blocks <= 0:
            nblk = cus  # 256 x 511 MiB + workspace is what fits beside the swap-buffer pool
    lean = a.lean == 1 or (a.lean < 0 and (nblk > cus or a.leg == "cfg5"))  # (cfg5: 256 x 511 MiB blocks with a swap buffer each would not fit)
    assert lib.bz3_hip_set_lean_states(1 if lean else 0) == 0
    cap = lib.bz3_bound(block_size) + 4096
    noise = a.noise
    if (a.kind == "text" or a.leg) and not a.emu:
        seed_text_source(lib)
        if noise < 0:
            import datagen

            noise = datagen.ENWIK_NOISE

    def markov16(n, seed):
        """cfg5's source: a skewed order-1 Markov chain over 16 symbols, repeat units < 40 B (LZP / RLE decline)."""
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        chains = 1 << 16
        steps5 = (n + chains - 1) // chains
        probs = 1.0 / torch.arange(1, 17, device=device, dtype=torch.float64) ** 1.6
        rows = torch.stack([probs[torch.randperm(16, generator=g, device=device)] for _ in range(16)])
        cdf = torch.cumsum(rows / rows.sum(1, keepdim=True), 1).to(torch.float32)
        st5 = torch.randint(0, 16, (chains,), generator=g, device=device)
        out5 = torch.empty((steps5, chains), dtype=torch.uint8, device=device)
        for t_ in range(steps5):
            r = torch.rand((chains,), generator=g, device=device)
            st5 = (cdf[st5] < r[:, None]).sum(1).clamp_(max=15)
            out5[t_] = (st5 + 97).to(torch.uint8)
        return out5.t().reshape(-1)[:n].contiguous()

    # ---- synthetic input, resident in HBM ----------------------------------------------------------------
    t_gen = time.perf_counter()
    nbase = max(1, min(a.text_bases, nblk))
    if a.leg == "cfg5" or a.emu:  # (--emu: no corpus to decode with the emulator; the control flow does not care what the bytes are)
        bases = [markov16(block_size, 50 + rank + 10 * j) for j in range(nbase if a.emu else 1)]
    elif a.kind == "text":
        bases = [gen_text_device(torch, block_size, seed=1 + rank + 100 * j, device=device, noise=noise) for j in range(nbase)]
    else:
        g = torch.Generator(device=device)
        g.manual_seed(2 + rank)
        bases = [torch.randint(0, 256, (block_size,), dtype=torch.uint8, generator=g, device=device)]
    bufs, prints = [], []
    chunk = 1 << 16
    nchunks = block_size // chunk
    for k in range(nblk):
        buf = torch.empty(cap, dtype=torch.uint8, device=device)
        base = bases[k % len(bases)]
        if k < len(bases) or nchunks < 2:
            buf[:block_size] = base
        else:  # every further block: one of the texts with its 64 KiB pieces in a block-specific order
            g = torch.Generator(device=device)
            g.manual_seed(1000 * (rank + 1) + k)
            perm = torch.randperm(nchunks, generator=g, device=device)
            buf[: nchunks * chunk] = base[: nchunks * chunk].view(nchunks, chunk)[perm].reshape(-1)
            buf[nchunks * chunk : block_size] = base[nchunks * chunk :]
        bufs.append(buf)
        prints.append(fingerprint(torch, buf[:block_size]))
    # full copies of a few blocks, kept out of the codec's reach, for a byte-for-byte comparison after the round trip
    n_keep = min(4, nblk) if not lean else min(2, nblk)
    kept = [bufs[k][:block_size].clone() for k in range(n_keep)]
    del bases, base
    dev_sync()
    if cuda:
        torch.cuda.empty_cache()  # hand the generator's temporaries back: the codec workspace is hipMalloc'ed outside torch
    t_gen = time.perf_counter() - t_gen
    progress(f"{nblk} x {block_size} B input blocks resident in HBM ({t_gen:.1f}s)")

    states = (C.c_void_p * nblk)(*[lib.bz3_new(max(block_size, 65 * 1024)) for _ in range(nblk)])  # (blocks below bz3_new's minimum: --emu tests)
    assert all(states), "bz3_new failed"
    progress("states created")
    ptrs = (C.c_void_p * nblk)(*[b.data_ptr() for b in bufs])
    bsz = (C.c_size_t * nblk)(*[cap] * nblk)
    orig = (C.c_int32 * nblk)(*[block_size] * nblk)

    def barrier():
        dev_sync()
        if world > 1:
            dist.barrier()
        dev_sync()

    def agree(x):
        """Rank 0's value of a number, on every rank (all ranks must take the same budget decisions)."""
        if world == 1:
            return x
        t = torch.tensor([float(x)], dtype=torch.float64, device=device)
        dist.broadcast(t, src=0)
        return float(t.item())

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([float(x)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    comp_sizes = [0] * nblk
    stage = {}
    # CPU parity / baseline sample: coded bytes of the first blocks, copied aside (device to device) before the in-place decode
    want_cpu = rank == 0 and world == 1 and not a.no_cpu_baseline and not a.emu  # (also for `--leg cfg5`, round 6: the reference beside the 511 MiB blocks)
    cpu_block = int(a.cpu_block_mib * (1 << 20)) if a.cpu_block_mib else block_size
    cpu_n = 0
    if want_cpu:
        per_thread = 7.5 * cpu_block + (64 << 20)  # reference state (~5.1 x block) + buffer + the shared copies
        cpu_n = int(max(1, min(os.cpu_count() or 1, a.cpu_threads, nblk, host_mem_available() * 0.8 // per_thread)))
    coded_kept = []
    parity_sample = {}  # "coded": SharedBlocks with the GPU's coded bytes of blocks 0..cpu_n-1, "how": where they were copied

    def keep_coded_sample(sizes):
        """The GPU's coded bytes of the parity sample, set aside before the in-place decode overwrites them.  Device-to-device copies into the
        headroom the library leaves (bz3_hip_set_workspace_headroom); if the device has no room for them after all (round 5: the kept workspace had
        taken every byte and the FIRST of these 48 MiB clones killed the run), straight to host shared memory instead -- slower, inside the timed
        step, and reported; if even that fails the sample is skipped and the run goes on without the byte comparison."""
        t_ = time.perf_counter()
        try:
            for i in range(cpu_n):
                coded_kept.append(bufs[i][: sizes[i]].clone())
            dev_sync()
            parity_sample["how"] = f"device-to-device copies inside the first timed step ({(time.perf_counter() - t_) * 1e3:.0f} ms)"
            return
        except RuntimeError as e:  # torch.OutOfMemoryError is one
            coded_kept.clear()
            if cuda:
                torch.cuda.empty_cache()
            progress(f"parity sample: no room on the device for the copies ({str(e)[:80]}...); copying to host memory instead")
        try:
            sh = SharedBlocks("bz3_bench_coded", [int(sizes[i]) for i in range(cpu_n)])
            for i in range(cpu_n):
                sh.put_tensor(torch, i, bufs[i])
            parity_sample["coded"] = sh
            parity_sample["how"] = f"device-to-host copies inside the first timed step ({time.perf_counter() - t_:.2f} s: the device had no room for device-side copies)"
        except Exception as e:
            parity_sample["coded"] = None
            parity_sample["how"] = f"skipped: {type(e).__name__}: {str(e)[:120]}"
            progress("parity sample " + parity_sample["how"])

    def one_step(record=False):
        sizes = (C.c_int32 * nblk)(*[block_size] * nblk)
        t0 = time.perf_counter()
        lib.bz3_hip_encode_blocks_device(states, ptrs, sizes, nblk)
        t1 = time.perf_counter()
        progress(f"encode_blocks done in {t1 - t0:.1f}s")
        for i in range(nblk):
            assert sizes[i] > 0 and lib.bz3_last_error(states[i]) == 0, f"encode failed on block {i}"
        if record:
            comp_sizes[:] = list(sizes)
            tm = (C.c_float * 8)()
            lib.bz3_hip_last_timings(states[0], tm)
            stage["enc"] = {bzip3_amd.T_NAMES[j]: round(tm[j], 3) for j in range(6)}
            r, p, e = C.c_int32(), C.c_int32(), C.c_uint64()
            lib.bz3_hip_last_bwt_stats(states[0], C.byref(r), C.byref(p), C.byref(e))
            stage["bwt"] = {"rounds": r.value, "radix_passes": p.value, "sorted_elements": e.value}
            ring = lib.bz3_hip_debug_front_end_ring()  # the encoder's front-end pipeline: context slots x blocks per window
            stage["front_end_ring"] = {"slots": (ring >> 16) & 0xFF, "window": ring & 0xFFFF, "workspace_handed_back": bool((ring >> 30) & 1), "two_threads": bool((ring >> 29) & 1)}
            if cpu_block == block_size and not coded_kept and not step_s and want_cpu and "coded" not in parity_sample:
                keep_coded_sample(sizes)  # a few ms of device-to-device copies inside the timed region, first step only (the in-place decode destroys the coded bytes)
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

    # ---- steps under the wall budget: nothing else runs on the host or the GPU meanwhile ---------------------------------
    extras_wanted = rank == 0 and world == 1 and not a.no_extras and not a.emu and not a.leg
    cpu_need = 2.6 * cpu_block / (4.5 * (1 << 20)) if want_cpu else 0.0  # ~4.5 MiB/s per thread and direction at 256 MiB blocks, with margin
    reserve = (cpu_need + 40.0 if want_cpu else 0.0) + (330.0 if extras_wanted else 0.0) + 30.0  # reference process, legs that follow it, verification
    step_s = []
    # the workspace survives between the calls of the timed steps (a caller that runs GPU-filling lean batches back to back would ask for the same:
    # include/bz3_hip.h); it is handed back before the extra legs
    keep_ws = not a.emu and os.environ.get("BZ3_BENCH_KEEP_WS", "1") != "0"
    # what the library must leave free on the device when a call returns (include/bz3_hip.h: the headroom rule): room for the parity sample's
    # copies of the coded blocks (~a fifth of a block each) and the fingerprints' temporaries
    headroom = (3 << 29) + (int(cpu_n * block_size / 4.5) if want_cpu and cpu_block == block_size and a.kind == "text" and not a.leg else (3 << 29) if not want_cpu else cpu_n * (block_size + (8 << 20)))
    headroom = max(3 << 30, min(headroom, 6 << 30 if a.kind == "text" and not a.leg else 40 << 30))
    if not a.emu and hasattr(lib, "bz3_hip_set_workspace_headroom"):
        lib.bz3_hip_set_workspace_headroom(headroom)

    def build_line():
        """The JSON line from what has been timed so far (rank 0 keeps it in RESULT; every exit path prints it)."""
        steps_run, timed, warmup_run = len(step_s), sum(step_s), 0
        total_bytes = world * nblk * block_size
        value = total_bytes * steps_run / 2 ** 20 / timed
        if rank == 0:
            comp_total = sum(comp_sizes)
            n_dec = block_size  # n' ~ n for text
            cm_dec_ms, cm_enc_ms = stage["dec"]["cm"], stage["enc"]["cm"]
            bwt_ms = stage["enc"]["bwt"]
            # Dominant kernel = the CM launch that takes longer.  Algorithmic bytes of a CM launch: every block's n' bytes on one
            # side and its coded bytes on the other (CM 1R + cW / cR + 1W); launch time from HIP events on the
            # launching stream (api.hip run_cm_jobs).
            dec_dominant = cm_dec_ms >= cm_enc_ms
            dom_ms = max(cm_dec_ms if dec_dominant else cm_enc_ms, 1e-6)
            enc_names = {0: "k_cm_encode", 1: "k_cm_encode_rows", 2: "k_cm_encode_rows3"}
            dec_names = {0: "k_cm_decode_sync", 1: "k_cm_decode_sync2", 2: "k_cm_decode_sync3"}
            kern = (dec_names if dec_dominant else enc_names).get(lib.bz3_hip_cm_variant_for(local_dev, nblk, 0 if dec_dominant else 1), "k_cm_decode_sync")
            cm_bytes = n_dec * nblk + comp_total
            # HBM traffic of the dominant kernel from the PMC pass (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc runs,
            # corrected as MI355X_MICROARCH.md prescribes), recorded per byte and scaled to this launch.
            traffic = None
            try:
                with open(PMC_TRAFFIC_FILE) as fh:
                    pm = json.load(fh).get(kern)
                if pm and dec_dominant:
                    traffic = int(pm["fetch_bytes_per_coded_byte"] * comp_total + pm["write_bytes_per_decoded_byte"] * n_dec * nblk)
                elif pm:
                    traffic = int(pm["fetch_bytes_per_input_byte"] * n_dec * nblk + pm["write_bytes_per_coded_byte"] * comp_total)
            except Exception:
                pass
            what = {"": "synthetic enwik-style text" if a.kind == "text" else "uniformly random", "cfg5": "16-symbol order-1 Markov (BASELINE.json configs[4] stand-in)"}[a.leg]
            if a.emu:
                what = "16-symbol order-1 Markov (--emu: CPU emulation of the kernels, control-flow test only)"
            out = {
                "metric": "MiB/s encode+decode round-trip, 256 MiB blocks" if not a.leg else f"MiB/s encode+decode round-trip, {block_size >> 20} MiB blocks (leg {a.leg})",
                "value": round(value, 3),
                "unit": "MiB/s",
                "n_gpus": world,
                "steps": steps_run,
                "warmup": warmup_run,
                "ms_per_step": round(timed / steps_run * 1e3, 1),
                "step_s": [round(x, 2) for x in step_s],
                "step_spread": round((max(step_s) - min(step_s)) / (timed / steps_run), 4),
                "higher_is_better": True,
                "scaling": "weak",
                "vs_baseline": None,
                "dtype": "u8",
                "data": "synthetic",
                "requested": {"steps": a.steps, "warmup": a.warmup, "budget_s": a.budget_s,
                              "note": "steps/warmup clamped to the wall budget: one step codes and decodes a GPU-filling batch of 256 MiB blocks; both steps are timed, "
                                      "the first one includes the first-call effects (workspace allocation); nothing else runs on the host or the GPU during the timed steps"},
                "config": {
                    "workload": f"{nblk} x {block_size / 2 ** 20:g} MiB {what} blocks per GPU"
                                + (f" (word-bigram Markov over shakespeare.txt tokens, {noise * 100:g} % noise tokens = enwik8's 4.41 : 1 at -b 16; {nbase} independent texts, "
                                   f"block k = text k mod {nbase} with its 64 KiB pieces in a block-specific order)" if a.kind == "text" and not a.leg and not a.emu else "")
                                + ", resident in HBM, bz3_hip_encode_blocks_device + bz3_hip_decode_blocks_device",
                    "block_bytes": block_size,
                    "blocks_per_gpu": nblk,
                    "parallelism": f"blocks sharded over {world} GPU(s), no collective",
                    "sharding_model": sharding_model(world),
                    "legs": {},  # the scalars of the extra legs (filled as they finish; the driver's record keeps `config`, not `configs`)
                    "compressed_ratio": round((nblk * block_size) / max(1, comp_total), 3),
                    "cm_mode": a.cm_mode,
                    "lean_states": bool(lean),
                    "keep_workspace": bool(keep_ws),
                    "workspace_headroom_bytes": int(headroom),
                    "parity_sample": parity_sample.get("how"),
                    "cm_blocks_given_up": int(lib.bz3_hip_cm_blocks_given_up()),
                    "bwt_output_repeat_rate_16MiB_sample": None,
                    "round_trip_check": f"position-weighted 64-bit fingerprint of every block + byte-for-byte comparison of {n_keep} blocks",
                },
                # dominant kernel by time: a CM launch (one workgroup per block; a serial integer recurrence,
                # latency-bound by construction -- SURVEY.md 7/H1), priced against the HBM roofline as the contract asks
                "roofline": {
                    "kernel": kern,
                    "bound": "hbm",
                    "achieved": round(cm_bytes / (dom_ms * 1e-3) / 1e9, 6),
                    "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s",
                    "frac": round(cm_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 9),
                    "traffic": traffic,
                    "launch_ms": dom_ms,
                    "algorithmic_bytes_per_launch": int(cm_bytes),
                },
                "path_roofline": {
                    "achieved": round(ALG_BYTES_ROUND_TRIP * total_bytes * steps_run / timed / 1e9, 4),
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "bytes_per_input_byte": ALG_BYTES_ROUND_TRIP,
                },
                "bwt_roofline": {
                    "stage": "bwt_forward (one 56-bit code-window radix sort + in-LDS group resolution), one block",
                    "achieved": round(ALG_BYTES_BWT * block_size / (max(bwt_ms, 1e-6) * 1e-3) / 1e9, 3),
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(ALG_BYTES_BWT * block_size / (max(bwt_ms, 1e-6) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6),
                    "ms": bwt_ms, **stage.get("bwt", {}),
                },
                "stages": json.loads(json.dumps(stage)),
                "gen_s": round(t_gen, 1),
                "configs": {},
                "cpu_baseline": {"value": None, "unit": "MiB/s", "cores": 0, "kind": "reference", "sample": "not run (budget, --no-cpu-baseline, --leg or N > 1)"},
            }
            prev = RESULT["line"]
            if prev is not None:  # a line built after an earlier step: what has been measured beside the steps since then stays
                out["configs"] = prev["configs"]
                out["cpu_baseline"] = prev["cpu_baseline"]
                out["config"]["legs"] = prev["config"]["legs"]
                for k_ in ("bwt_output_repeat_rate_16MiB_sample",):
                    if prev["config"].get(k_) is not None:
                        out["config"][k_] = prev["config"][k_]
            RESULT["line"] = out
            if a.leg == "cfg5":
                out["config"]["legs"]["cfg5_round_trip_MiBps"] = out["value"]
            elif not a.emu:
                # the cfg5 leg is a run of its own (a 511 MiB block's CM launches last ~2.5 + ~5.3 minutes): its last RECORDED value rides along, marked as such
                # (under a key of its own: config.legs holds only what THIS run measured -- ADVICE r05)
                rec_ = {}
                for key_, file_ in RECORDED_RUNS.items():
                    try:
                        with open(os.path.join(ROOT, "profiles", file_)) as fh:
                            j_ = json.load(fh)
                        rec_[key_] = {"value": j_["value"], "unit": j_.get("unit", "MiB/s"), "vs_cpu": (j_.get("cpu_baseline") or {}).get("gpu_over_cpu"), "file": "profiles/" + file_}
                    except Exception:
                        pass
                if rec_:
                    out["config"]["recorded"] = {"note": "runs of their own (`bench.py --leg cfg5`, `bench.py --kind random`): NOT measured in this run, copied from the files named", **rec_}


    if keep_ws:
        lib.bz3_hip_set_keep_workspace(1)
    front_duo = os.environ.get("BZ3_BENCH_FRONT_DUO", "")  # (experiments: "1" / "0" force the encoder's two-thread front end on / off; unset = the library's default)
    if front_duo in ("0", "1") and hasattr(lib, "bz3_hip_set_front_end_duo"):
        lib.bz3_hip_set_front_end_duo(int(front_duo))
    PHASE["name"] = "first timed step"
    inject("step1")
    barrier()
    t0 = time.perf_counter()
    one_step(record=True)
    barrier()
    step_s.append(max_over_ranks(time.perf_counter() - t0))

    def verify_round_trip():
        # the round trip must be the identity (decode is in place: the buffers hold the plaintext again)
        for k in range(nblk):
            assert fingerprint(torch, bufs[k][:block_size]) == prints[k], f"block {k}: round trip changed the data"
        for k in range(n_keep):
            assert torch.equal(bufs[k][:block_size], kept[k]), f"block {k}: round trip changed the data"

    build_line()  # from here on every exit path prints a line (the watchdog, SIGTERM, an exception)
    PHASE["name"] = "verification of the first step's round trip"
    inject("verify1")
    verify_round_trip()
    # the GPU's coded bytes of the parity sample go to shared memory NOW (outside the timed steps): 64 x ~50 MB of HBM that the second
    # step's ring of LZP contexts can use
    shared_coded = parity_sample.get("coded")
    if want_cpu and coded_kept:
        try:
            shared_coded = SharedBlocks("bz3_bench_coded", [int(c.numel()) for c in coded_kept])
            for i, c in enumerate(coded_kept):
                shared_coded.put_tensor(torch, i, c)
        except Exception as e:
            shared_coded = None
            parity_sample["how"] = f"skipped: {type(e).__name__}: {str(e)[:120]}"
        coded_kept.clear()
        if cuda:
            torch.cuda.empty_cache()
    PHASE["fatal"] = False  # one step timed and verified: the headline stands whatever happens next (but see main(): assertions stay fatal)
    left = agree(a.budget_s - elapsed()) - reserve
    second = (a.steps >= 2 or a.warmup > 0) and left > step_s[0] * 1.03
    if second:
        PHASE["name"] = "second timed step"
        inject("step2")
        barrier()
        t0 = time.perf_counter()
        one_step(record=True)
        barrier()
        step_s.append(max_over_ranks(time.perf_counter() - t0))
        build_line()
        PHASE["name"] = "verification of the second step's round trip"
        verify_round_trip()
    if keep_ws:
        lib.bz3_hip_set_keep_workspace(-1)
        lib.bz3_hip_release_cached_memory()
    build_line()
    steps_run = len(step_s)
    progress(f"timed {steps_run} step(s): {[round(x, 1) for x in step_s]} s (requested {a.steps} / {a.warmup}; budget {a.budget_s:.0f}s)")
    PHASE["name"] = "extra legs"
    inject("legs")
    if rank == 0 and not a.emu:
        # what the CM stage sees, on a 16 MiB sample of block 0 through the stage hooks (outside the timed region): how often a byte of
        # the BWT output repeats its predecessor decides how often the decoder's guess-ahead is right
        try:
            import numpy as np

            g_ = bzip3_amd.StageApi(lib)
            smp = bufs[0][: min(block_size, 16 << 20)].cpu().numpy().tobytes()
            nl_, lz_ = g_.lzp_encode(smp)
            u_ = np.frombuffer(g_.bwt(lz_ if nl_ > 0 else smp)[1], dtype=np.uint8)
            RESULT["line"]["config"]["bwt_output_repeat_rate_16MiB_sample"] = round(float((u_[1:] == u_[:-1]).mean()), 4)
        except Exception as e:
            progress(f"repeat-rate sample failed: {e}")
    # ---- extra legs (rank 0 at N=1; the watchdog prints the line without them if they overrun) ----------------------------
    def left_s():
        return a.budget_s - elapsed()

    legs_sel = {x for x in a.legs.split(",") if x}

    def leg_wanted(name):
        """text batches only; `--legs a,b` picks some of the extra legs (default: all of them)"""
        return extras_wanted and a.kind == "text" and (not legs_sel or name in legs_sel)

    class guard:
        """An extra leg may fail (out of memory beside the batch, a reference process that dies ...) without taking the run's line with it: the
        failure is recorded under configs[name] and the next leg runs."""

        def __init__(self, name):
            self.name = name

        def __enter__(self):
            PHASE.update(name=f"leg {self.name}", fatal=False)

        def __exit__(self, et, ev, tb):
            if et is None or not issubclass(et, Exception):
                return False
            import traceback

            progress(f"leg {self.name} FAILED: {et.__name__}: {str(ev)[:300]}")
            traceback.print_tb(tb, file=sys.stderr)
            if RESULT["line"] is not None:
                slot = RESULT["line"]["configs"].setdefault(self.name, {})
                if isinstance(slot, dict):
                    slot["failed"] = f"{et.__name__}: {str(ev)[:300]}"
                RESULT["line"].setdefault("failed_legs", []).append(self.name)
            try:
                if cuda:
                    torch.cuda.empty_cache()
            except Exception:
                pass
            return True

    def leg(name, v):
        """a leg's scalar where the driver's record keeps it (config.legs)"""
        if RESULT["line"] is not None and v is not None:
            RESULT["line"]["config"]["legs"][name] = v

    def round_trip(sel, sizes_in, host=None):
        """encode + decode of blocks `sel` of the batch (device pointers; host: a list of host buffers instead, through
        bz3_encode_blocks / bz3_decode_blocks); returns (t_enc, t_dec, coded sizes)."""
        n = len(sel)
        S = (C.c_void_p * n)(*[states[k] for k in sel])
        P = (C.c_void_p * n)(*([C.addressof(h) for h in host] if host else [bufs[k].data_ptr() for k in sel]))
        sz = (C.c_int32 * n)(*sizes_in)
        enc_fn, dec_fn = (lib.bz3_encode_blocks, lib.bz3_decode_blocks) if host else (lib.bz3_hip_encode_blocks_device, lib.bz3_hip_decode_blocks_device)
        dev_sync()
        t0 = time.perf_counter()
        enc_fn(S, P, sz, n)
        t1 = time.perf_counter()
        assert all(sz[i] > 0 and lib.bz3_last_error(S[i]) == 0 for i in range(n)), "encode failed"
        coded = list(sz)
        B = (C.c_size_t * n)(*[cap] * n)
        O = (C.c_int32 * n)(*sizes_in)
        dec_fn(S, P, B, sz, O, n)
        t2 = time.perf_counter()
        assert all(lib.bz3_last_error(S[i]) == 0 for i in range(n)), "decode failed"
        return t1 - t0, t2 - t1, coded

    ref_choice = {"label": "gcc -O2", "path": None}
    shared_plain = None
    cpu_worker_proc = None
    with guard("cpu_baseline_start"):
        if want_cpu and a.deadline_s - 30.0 - elapsed() > cpu_need:
            # plaintext of blocks 0..cpu_n-1 (verified above) and the GPU's coded bytes of the same blocks, in shared memory for the reference process
            bs_cpu = min(cpu_block, block_size)
            shared_plain = SharedBlocks("bz3_bench_plain", [bs_cpu] * cpu_n)
            for i in range(cpu_n):
                shared_plain.put_tensor(torch, i, bufs[i if cpu_block == block_size else 0])
            progress(f"cpu_baseline: {cpu_n} threads x {bs_cpu >> 20} MiB blocks in a process of its own (estimated {cpu_need:.0f}s); GPU-only legs meanwhile")
            cpu_worker_proc = RefWorker(shared_plain.meta(), max(bs_cpu, 65 * 1024), coded=shared_coded.meta() if shared_coded else None, probe=True)
        elif want_cpu:
            RESULT["line"]["cpu_baseline"]["sample"] = f"not run: needs ~{cpu_need:.0f}s, {a.deadline_s - elapsed():.0f}s to the deadline"

    def free_most_of_the_batch():
        """room for the legs below: most of the batch's buffers and the big workspace are not needed any more"""
        keep_n = min(nblk, 160)
        for s_ in states[keep_n:]:
            lib.bz3_free(s_)
        del bufs[keep_n:]
        lib.bz3_hip_release_cached_memory()
        if cuda:
            torch.cuda.empty_cache()
        return keep_n

    live_states = nblk
    random_host = None
    mixed_host = None
    with guard("random"):
        if leg_wanted("random") and left_s() > 200.0:
            # ---- random: incompressible blocks, as many as the timed batch had (LZP and RLE decline, model 0, ~1.004 bytes per byte) ----
            rb = int(a.random_block_mib * (1 << 20))
            nr = a.random_blocks if a.random_blocks > 0 else nblk
            nr = max(1, min(nr, nblk))
            if rb <= block_size:
                progress(f"random: {nr} x {a.random_block_mib:g} MiB")
                g = torch.Generator(device=device)
                g.manual_seed(2)
                fpr = []
                rsel = list(range(nblk - nr, nblk))  # (their text is not needed any more, except blocks 0..cpu_n-1's: those are in shared memory already)
                for k in rsel:
                    bufs[k][:rb] = torch.randint(0, 256, (rb,), dtype=torch.uint8, generator=g, device=device)
                    fpr.append(fingerprint(torch, bufs[k][:rb]))
                n_ref = min(nr, a.cpu_threads, os.cpu_count() or 1)
                if want_cpu:
                    random_host = SharedBlocks("bz3_bench_random", [rb] * n_ref)
                    for j in range(n_ref):
                        random_host.put_tensor(torch, j, bufs[rsel[j]])
                gu0, rf0 = int(lib.bz3_hip_cm_blocks_given_up()), int(lib.bz3_hip_cm_blocks_routed_full())
                te, td, coded = round_trip(rsel, [rb] * nr)
                gu1, rf1 = int(lib.bz3_hip_cm_blocks_given_up()), int(lib.bz3_hip_cm_blocks_routed_full())
                assert all(fingerprint(torch, bufs[k][:rb]) == f for k, f in zip(rsel, fpr)), "random: round trip changed the data"
                RESULT["line"]["configs"]["random"] = {
                    "workload": f"{nr} x {a.random_block_mib:g} MiB uniformly random blocks on one GPU (states of {a.block_mib:g} MiB)",
                    "value": round(nr * rb / 2 ** 20 / (te + td), 3), "unit": "MiB/s", "t_enc_s": round(te, 2), "t_dec_s": round(td, 2),
                    "compressed_ratio": round(nr * rb / sum(coded), 4),
                    # 256 live order-1 rows fit no row cache: since round 5 such blocks go STRAIGHT to the whole-model kernels (encode: by the BWT's histogram,
                    # decode: by a payload that did not shrink), one per CU at a time, instead of being given up by the row-cache kernels first
                    "cm_blocks_given_up": gu1 - gu0, "cm_blocks_routed_to_whole_model": rf1 - rf0}
                leg("random_MiBps", RESULT["line"]["configs"]["random"]["value"])
                leg("random_cm_blocks_given_up", gu1 - gu0)
                leg("random_block_MiB", a.random_block_mib)
                progress(f"random: {RESULT['line']['configs']['random']['value']} MiB/s")

    with guard("free_most_of_the_batch"):
        if extras_wanted and a.kind == "text" and nblk > 160 and left_s() > 60.0:
            live_states = free_most_of_the_batch()

    with guard("mixed"):
        if leg_wanted("mixed") and live_states >= 96 and block_size >= (32 << 20) and left_s() > 200.0:
            # mixed batch: text, binary and incompressible blocks through ONE pair of batch calls
            mb = 32 << 20
            per = 32
            g = torch.Generator(device=device)
            g.manual_seed(7)
            sel = list(range(64, 64 + 3 * per))  # (blocks 0..63 keep their text for the legs below)
            for j in range(per):  # the first 32 keep their text; then binary (little-endian words of a random walk); then random
                k = sel[per + j]
                walk = torch.cumsum(torch.randint(-3, 4, (mb // 4,), generator=g, device=device, dtype=torch.int32), 0).to(torch.int32)
                bufs[k][:mb] = walk.view(torch.uint8)
                bufs[sel[2 * per + j]][:mb] = torch.randint(0, 256, (mb,), dtype=torch.uint8, generator=g, device=device)
            fpm = [fingerprint(torch, bufs[k][:mb]) for k in sel]
            if want_cpu and host_mem_available() > 3 * len(sel) * mb:  # the same 96 blocks for the reference (run later, when the host is free again)
                mixed_host = SharedBlocks("bz3_bench_mixed", [mb] * len(sel))
                for j, k in enumerate(sel):
                    mixed_host.put_tensor(torch, j, bufs[k])
            before = int(lib.bz3_hip_cm_blocks_given_up())
            te, td, coded = round_trip(sel, [mb] * len(sel))
            assert all(fingerprint(torch, bufs[k][:mb]) == f for k, f in zip(sel, fpm)), "mixed: round trip changed the data"
            RESULT["line"]["configs"]["mixed"] = {
                "workload": f"{3 * per} x 32 MiB blocks in one batch on one GPU: {per} text, {per} binary (32-bit words of a random walk), {per} uniformly random",
                "value": round(len(sel) * mb / 2 ** 20 / (te + td), 3), "unit": "MiB/s", "t_enc_s": round(te, 2), "t_dec_s": round(td, 2),
                "compressed_ratio": {"text": round(per * mb / sum(coded[:per]), 3), "binary": round(per * mb / sum(coded[per : 2 * per]), 3),
                                     "random": round(per * mb / sum(coded[2 * per :]), 4)},
                "cm_blocks_given_up": int(lib.bz3_hip_cm_blocks_given_up()) - before}
            leg("mixed_MiBps", RESULT["line"]["configs"]["mixed"]["value"])
            progress(f"mixed: {RESULT['line']['configs']['mixed']['value']} MiB/s, {RESULT['line']['configs']['mixed']['cm_blocks_given_up']} blocks given up by the row-cache kernels")

    # ---- cpu_baseline: started after the timed steps, collected here (the legs above used the GPU only) ----------------
    with guard("cpu_baseline"):
        if cpu_worker_proc is not None:
            res = cpu_worker_proc.result()
            if "rec" in res:
                rec = res["rec"]
                ref_choice.update(label=res.get("ref_label") or "gcc -O2", path=res.get("ref_path"))
                rec["build_probe_1_thread_8MiB_MiBps"] = res.get("probe", {})
                rec["threads_note"] = ("threads = 64: the reference's CLI caps -j at 64 (src/main.c:213); one thread per block as in bz3_encode_blocks (src/libbz3.c:845-856).  "
                                       "cores = effective_cpus = min(threads, cpu_quota_cpus): what the container may run at once (cgroup) -- on the round-4 GPU boxes 16 of the 256 "
                                       "logical CPUs, where 64, 32 and 16 threads give the same rate within 8 % (profiles/r04_cpu_threads_probe.json).  gpu_over_cpu compares with those "
                                       "effective CPUs, not with an unthrottled -j 64; gpu_equivalent_cpus = GPU MiB/s / (reference MiB/s per effective CPU)")
                rec["process"] = f"a process of its own (no torch, no HIP runtime), pinned to {res.get('pinned_to_cpus', 0)} cores; started after the timed steps, while the GPU ran legs that use no host cores"
                rec["gpu_over_cpu"] = round(RESULT["line"]["value"] / rec["value"], 3) if rec.get("value") else None
                # one MI355X with a GPU-filling batch in flight = this many of the host's CPUs running the reference
                rec["gpu_equivalent_cpus"] = round(RESULT["line"]["value"] / rec["MiBps_per_cpu"], 1) if rec.get("MiBps_per_cpu") else None
                leg("gpu_over_cpu_threads", rec["gpu_over_cpu"])
                leg("gpu_equivalent_cpus", rec["gpu_equivalent_cpus"])
                if "parity_same" in res:
                    rec["parity"] = f"{res['parity_same']} of {res['parity_of']} blocks of {cpu_block >> 20} MiB: the GPU's coded bytes are identical to the reference's"
                RESULT["line"]["cpu_baseline"] = rec
                progress(f"cpu_baseline: {rec['value']} MiB/s on {rec['threads']} threads = {rec['effective_cpus']} effective CPUs ({ref_choice['label']}); {rec.get('parity', '')}")
                if res.get("parity_same") != res.get("parity_of"):  # NOT swallowed by the guard: the run fails, with the line printed
                    RESULT["line"]["parity_failed"] = "GPU output differs from the reference: " + rec.get("parity", "")
                    progress(RESULT["line"]["parity_failed"])
            else:
                RESULT["line"]["cpu_baseline"] = {"value": None, "unit": "MiB/s", "cores": 0, "kind": "reference", "sample": "not run: " + res.get("err", "?")}

    def ref_beside(meta, bs):
        """the reference's -j N on the same blocks, in a process of its own while the GPU codes them"""
        return RefWorker(meta, max(bs, 65 * 1024), ref_path=ref_choice["path"], ref_label=ref_choice["label"])

    with guard("random_cpu"):
        if random_host is not None and "random" in RESULT["line"]["configs"] and left_s() > 60.0:
            res = ref_beside(random_host.meta(), int(a.random_block_mib * (1 << 20))).result()
            rec = RESULT["line"]["configs"]["random"]
            if "rec" in res:
                rec["cpu"] = res["rec"]
                rec["vs_cpu"] = round(rec["value"] / res["rec"]["value"], 3)
                leg("random_vs_cpu", rec["vs_cpu"])
                progress(f"random: reference on {res['rec']['threads']} threads: {res['rec']['value']} MiB/s")
            else:
                rec["cpu"] = {"value": None, "sample": "failed: " + res.get("err", "?")}
            random_host.close()

    with guard("mixed_cpu"):
        if mixed_host is not None and "mixed" in RESULT["line"]["configs"] and left_s() > 150.0:
            res = ref_beside(mixed_host.meta(), 32 << 20).result()
            rec = RESULT["line"]["configs"]["mixed"]
            if "rec" in res:
                rec["cpu"] = res["rec"]
                rec["vs_cpu"] = round(rec["value"] / res["rec"]["value"], 3)
                leg("mixed_vs_cpu", rec["vs_cpu"])
                progress(f"mixed: reference on {res['rec']['threads']} threads: {res['rec']['value']} MiB/s")
            else:
                rec["cpu"] = {"value": None, "sample": "failed: " + res.get("err", "?")}
        if mixed_host is not None:
            mixed_host.close()
            mixed_host = None

    if leg_wanted("cfg5_unbwt") and left_s() > 200.0:
        # (after the reference process has finished: the stage hooks wait for the stream between launches, and a host whose CPU quota 64
        # reference threads saturate stretches exactly those waits -- round 4's rehearsal measured this leg 3.4x slower beside them)
        # cfg5's stage (BASELINE.json configs[4]: "-b 511 max block ... decode-path unBWT throughput"): the inverse BWT of one block of the
        # maximum size.  Verbatim long repeats would be collapsed by LZP, so the source is a skewed order-1 Markov chain
        # over 16 symbols (repeat units < 40 B: LZP / RLE decline, the BWT stage sees all n bytes).
        try:
            n5 = 511 << 20
            src5 = markov16(n5, 5).cpu().numpy().tobytes()
            if cuda:
                torch.cuda.empty_cache()
            gs = bzip3_amd.StageApi(lib)
            idx5, u5 = gs.bwt(src5)
            ms_fwd = float(lib.bz3_hip_stage_last_ms())
            rc5, back5 = gs.unbwt(u5, idx5)
            ms_inv = float(lib.bz3_hip_stage_last_ms())
            assert rc5 == 0 and back5 == src5, "cfg5_unbwt: the inverse BWT did not return the block"
            RESULT["line"]["configs"]["cfg5_unbwt"] = {
                "workload": "BASELINE.json configs[4]'s stage on one GPU: inverse BWT of ONE 511 MiB block (535,822,336 B, the maximum block size) of a skewed "
                            "order-1 Markov source over 16 symbols (bz3_hip_stage_unbwt; transform alone, PCIe copies of the hook excluded); "
                            "the whole-pipeline round trip of a batch of such blocks is `bench.py --leg cfg5` (profiles/)",
                "value": round(ALG_BYTES_BWT * n5 / (ms_inv * 1e-3) / 1e9, 3), "unit": "GB/s (11 algorithmic bytes per byte, SURVEY.md 8d)",
                "ms": round(ms_inv, 2), "MiBps": round(511.0 / (ms_inv * 1e-3), 1), "frac_of_hbm_peak": round(ALG_BYTES_BWT * n5 / (ms_inv * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                "forward_bwt_ms": round(ms_fwd, 2), "forward_bwt_GBps": round(ALG_BYTES_BWT * n5 / (ms_fwd * 1e-3) / 1e9, 3)}
            leg("cfg5_unbwt_GBps", RESULT["line"]["configs"]["cfg5_unbwt"]["value"])
            progress(f"cfg5_unbwt: inverse BWT of a 511 MiB block in {ms_inv:.1f} ms (forward {ms_fwd:.1f} ms)")
            del src5, u5, back5
            lib.bz3_hip_release_cached_memory()
        except Exception as e:
            RESULT["line"]["configs"]["cfg5_unbwt"] = {"skipped": f"failed: {e}"}

    def small_config(name, total_bytes_, bs, what):
        """BASELINE configs of a few blocks: `total_bytes_` of the batch's text at block size bs on one GPU, the reference's -j N beside it."""
        nb = (total_bytes_ + bs - 1) // bs
        sizes_ = [bs] * (nb - 1) + [total_bytes_ - (nb - 1) * bs]
        sel = list(range(nb))
        fp = [fingerprint(torch, bufs[k][: sizes_[k]]) for k in sel]
        w = None
        if shared_plain is not None and cpu_n >= nb and cpu_block == block_size:
            w = ref_beside(shared_plain.meta(sel, sizes_), bs)
        te, td, coded = round_trip(sel, sizes_)
        assert all(fingerprint(torch, bufs[k][: sizes_[k]]) == fp[k] for k in sel), f"{name}: round trip changed the data"
        rec = {"workload": f"{what}: {total_bytes_} B of synthetic text, -b {bs / 2 ** 20:g} -> {nb} blocks ({nb - 1} x {bs} + {sizes_[-1]}) on one GPU",
               "value": round(total_bytes_ / 2 ** 20 / (te + td), 3), "unit": "MiB/s", "t_enc_s": round(te, 2), "t_dec_s": round(td, 2),
               "compressed_ratio": round(total_bytes_ / sum(coded), 3)}
        RESULT["line"]["configs"][name] = rec
        leg(f"{name}_MiBps", rec["value"])
        progress(f"{name}: {rec['value']} MiB/s (enc {te:.1f}s dec {td:.1f}s)")
        if w is not None:
            res = w.result()
            if "rec" in res:
                r = res["rec"]
                r["sample"] = f"reference -j {nb} ({ref_choice['label']}): bz3_encode_blocks + bz3_decode_blocks on the same {nb} blocks, {nb} threads of a process of its own, timed while the GPU coded them"
                rec[f"cpu_j{nb}"] = r
                rec[f"vs_cpu_j{nb}"] = round(rec["value"] / r["value"], 3)
                leg(f"{name}_vs_cpu_j{nb}", rec[f"vs_cpu_j{nb}"])
                progress(f"{name}: reference -j {nb} on the host: {r['value']} MiB/s")
            else:
                rec[f"cpu_j{nb}"] = {"value": None, "sample": "failed: " + res.get("err", "?")}

    n3 = (a.cfg3_bytes + block_size - 1) // block_size  # blocks of the cfg3 leg: full blocks + one partial
    with guard("cfg3"):
        if leg_wanted("cfg3") and nblk >= n3:
            # cfg3 (BASELINE.json configs[2]): 1,000,000,000 B at -b 256 = 3 full blocks + 194,693,632 B, on one GPU
            cm_s = (stage["enc"]["cm"] + stage["dec"]["cm"]) * 1e-3
            est = cm_s * (1.0 if per_cu == 1 else 0.7) + 15.0
            if left_s() > est:
                progress(f"cfg3: {n3} blocks (estimated {est:.0f}s)")
                is_cfg3 = a.cfg3_bytes == CFG3_BYTES and block_size == 256 << 20
                small_config("cfg3", a.cfg3_bytes, block_size, "BASELINE.json configs[2] stand-in" if is_cfg3 else "(not BASELINE's size)")
            else:
                RESULT["line"]["configs"]["cfg3"] = {"skipped": f"needs ~{est:.0f}s, {left_s():.0f}s of the budget left"}

    with guard("cfg2"):
        if leg_wanted("cfg2") and block_size >= (32 << 20) and nblk >= 3:
            # cfg2 (BASELINE.json configs[1]): 100,000,000 B at -b 32 = 2 full blocks + 32,891,136 B
            if left_s() > 60.0:
                small_config("cfg2", 100_000_000, 32 << 20, "BASELINE.json configs[1] stand-in")
            else:
                RESULT["line"]["configs"]["cfg2"] = {"skipped": f"{left_s():.0f}s of the budget left"}

    hb = int(a.host_api_block_mib * (1 << 20))
    with guard("host_api"):
        if leg_wanted("host_api") and hb > 0 and hb <= block_size and live_states >= 8:
            # ---- host_api: SURVEY.md 8d's timing boundary -- bz3_encode_blocks / bz3_decode_blocks on malloc'ed HOST buffers (H2D / D2H inside the
            # timed calls) against the same batch device-resident.  Blocks of hb bytes, one per state still alive.
            nh = live_states
            est = 2.2 * (stage["enc"]["cm"] + stage["dec"]["cm"]) * 1e-3 * hb / block_size * (1.0 if nh > 2 * cus else 0.8) + 30.0
            if left_s() > est and host_mem_available() > nh * (hb + hb // 40 + 8192) * 1.3:
                progress(f"host_api: {nh} x {a.host_api_block_mib:g} MiB through host buffers (estimated {est:.0f}s)")
                sel = list(range(nh))
                hcap = lib.bz3_bound(hb) + 64
                fph = [fingerprint(torch, bufs[k][:hb]) for k in sel]
                te_d, td_d, coded_d = round_trip(sel, [hb] * nh)
                host = [(C.c_uint8 * hcap)() for _ in sel]
                for k in sel:
                    torch.frombuffer(host[k], dtype=torch.uint8, count=hb).copy_(bufs[k][:hb])
                cap_saved = cap
                cap = hcap  # (round_trip passes `cap` as the buffer size of the decode call)
                te_h, td_h, coded_h = round_trip(sel, [hb] * nh, host=host)
                cap = cap_saved
                ok = coded_h == coded_d and all(fingerprint(torch, torch.frombuffer(host[k], dtype=torch.uint8, count=hb).to(device)) == fph[k] for k in sel[:8])
                assert ok, "host_api: the host-buffer round trip differs from the device-resident one"
                RESULT["line"]["configs"]["host_api"] = {
                    "workload": f"{nh} x {a.host_api_block_mib:g} MiB text blocks through bz3_encode_blocks + bz3_decode_blocks on malloc'ed host buffers (H2D / D2H inside the timed calls: "
                                f"SURVEY.md 8d's boundary, include/libbz3.h:206-213), and the same batch device-resident",
                    "value": round(nh * hb / 2 ** 20 / (te_h + td_h), 3), "unit": "MiB/s", "t_enc_s": round(te_h, 2), "t_dec_s": round(td_h, 2),
                    "device_resident": {"value": round(nh * hb / 2 ** 20 / (te_d + td_d), 3), "t_enc_s": round(te_d, 2), "t_dec_s": round(td_d, 2)},
                    "pcie_inclusive_over_device_resident": round((te_d + td_d) / (te_h + td_h), 4)}
                leg("host_api_ratio", RESULT["line"]["configs"]["host_api"]["pcie_inclusive_over_device_resident"])
                progress(f"host_api: {RESULT['line']['configs']['host_api']['value']} MiB/s through host buffers, x{RESULT['line']['configs']['host_api']['pcie_inclusive_over_device_resident']} of device-resident")
                del host
            else:
                RESULT["line"]["configs"]["host_api"] = {"skipped": f"needs ~{est:.0f}s, {left_s():.0f}s of the budget left"}

    if rank == 0:
        emit_line(final=True)
    for s in states[:live_states]:
        lib.bz3_free(s)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""bench.py's multi-rank control flow on CPU: `python bench.py --gpus 2 --emu` starts its own two
ranks (torch.distributed.run on 127.0.0.1), every rank codes and decodes its own blocks -- the kernel sources under the test-only HIP
emulation instead of a GPU, gloo instead of RCCL -- and rank 0 prints ONE JSON line whose value is the whole job's bytes over the
max-over-ranks time.  The same file is what the driver launches on the 8-GPU node:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W
"""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--emu", "--blocks", "2", "--block-mib", "0.003", "--steps", "2", "--warmup", "0", "--text-bases", "2"]


def _run(cmd):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE line, from rank 0
    return json.loads(lines[0]), r.stderr


def _check(d, world):
    assert d["n_gpus"] == world and d["scaling"] == "weak" and d["higher_is_better"] is True and d["unit"] == "MiB/s"
    assert d["steps"] == 2 and len(d["step_s"]) == 2 and d["complete"] is True
    nbytes = world * 2 * int(0.003 * (1 << 20)) * d["steps"]  # every rank's blocks count: the whole job's aggregate
    assert abs(d["value"] - nbytes / 2 ** 20 / sum(d["step_s"])) <= 0.02 * d["value"] + 1e-3
    assert d["config"]["blocks_per_gpu"] == 2 and d["cpu_baseline"]["value"] is None
    # what the driver's record keeps of the legs and of the sharding model lives under `config`
    sm = d["config"]["sharding_model"]
    assert sm["cfg4_linux_tarball_5_blocks_8_gpus"] == 0.625 and sm["cfg5_8GiB_17_blocks_8_gpus"] == round(17 / 24, 4)
    assert d["config"]["legs"] == {} and str(world) in sm["this_run"]  # (recorded runs ride under config.recorded, never under legs)


def test_bench_spawns_its_own_ranks_and_rank_0_reports_the_aggregate():
    d, err = _run([sys.executable, "bench.py", "--gpus", "2"] + ARGS)
    _check(d, 2)
    assert err.count("encode_blocks done") == 2  # progress lines come from rank 0 only: two steps


def test_bench_under_torch_distributed_run_as_the_driver_launches_it():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    d, _ = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                 "bench.py", "--gpus", "2"] + ARGS)
    _check(d, 2)


def test_bench_eight_ranks_as_on_the_8_gpu_node():
    """World 8 (the node the driver's SCALE run uses): eight ranks under torch.distributed.run, gloo, the emulated kernels -- the control
    flow of the N = 8 line (barriers, max-over-ranks, rank 0's aggregate), nothing measured."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    d, err = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(port),
                   "bench.py", "--gpus", "8"] + ARGS)
    _check(d, 8)
    assert "8 GPU(s)" in d["config"]["parallelism"] and err.count("encode_blocks done") == 2


def test_bench_single_rank_on_the_emulator():
    d, _ = _run([sys.executable, "bench.py", "--gpus", "1"] + ARGS)
    _check(d, 1)


def test_bench_cfg5_leg_control_flow_on_the_emulator():
    """`bench.py --leg cfg5` (a batch of maximum-size blocks of the 16-symbol source; run on its own, not by the driver) at a test size."""
    env = dict(os.environ)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--emu", "--leg", "cfg5", "--leg-block-mib", "0.003", "--blocks", "2", "--steps", "1"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert "leg cfg5" in d["metric"] and d["steps"] == 1 and d["cpu_baseline"]["value"] is None and d["configs"] == {}
    assert d["config"]["legs"] == {"cfg5_round_trip_MiBps": d["value"]}


def _run_injected(point):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["BZ3_BENCH_INJECT_FAIL"] = point
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + ARGS, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-1500:])  # EVERY exit path prints exactly one line (round 5's run died without one)
    return r.returncode, json.loads(lines[0])


def test_bench_prints_a_line_on_every_failure_path():
    """an exception anywhere leaves a JSON line with what is complete.  Before the first step: value null, exit 1.  In the
    verification of the first step: the value with the error beside it, exit 1 (the headline is not valid).  In the second step, by something
    that is not a correctness check (out of memory ...): the first step's verified value, exit 0.  A correctness assertion there: exit 1.
    After the steps: the steps' line, exit 0."""
    rc, d = _run_injected("step1")
    assert rc == 1 and d["value"] is None and "injected failure at step1" in d["error"] and d["failed_phase"] == "first timed step" and d["complete"] is False
    rc, d = _run_injected("verify1")
    assert rc == 1 and d["value"] > 0 and d["steps"] == 1 and "verify1" in d["error"] and "verification" in d["failed_phase"]
    rc, d = _run_injected("step2")
    assert rc == 0 and d["value"] > 0 and d["steps"] == 1 and len(d["step_s"]) == 1 and "step2" in d["error"] and d["complete"] is False
    assert d["roofline"]["launch_ms"] > 0 and d["stages"]["t_enc_s"] > 0
    rc, d = _run_injected("step2:assert")
    assert rc == 1 and d["steps"] == 1 and "injected assertion" in d["error"]
    rc, d = _run_injected("legs")
    assert rc == 0 and d["steps"] == 2 and "legs" in d["error"]

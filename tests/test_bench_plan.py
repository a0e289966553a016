"""bench.py's wall-budget planner (CPU): the driver runs `bench.py --steps 20 --warmup 5` under an 1800 s limit while one step over
a GPU-filling batch of 256 MiB blocks takes minutes, so steps / warmup are clamped to what fits -- round 1's driver bench timed out."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def total_time(t_first, warmup, more):
    return t_first * (1 + max(0, warmup - 1) + more)


def test_planner_always_has_a_timed_step_and_never_overruns():
    for t_first in (0.5, 8.0, 120.0, 300.0, 512.0, 900.0, 2000.0):
        for left in (-100.0, 0.0, 50.0, 153.0, 600.0, 750.0, 1250.0, 5000.0):
            for steps, warm in ((1, 0), (20, 5), (3, 1), (1, 5), (20, 0), (2, 0)):
                w, more = bench.plan_steps(left, t_first, steps, warm)
                timed = more if w else 1 + more
                assert timed >= 1 and timed <= max(1, steps), (t_first, left, steps, warm, w, more)
                assert w <= warm, (t_first, left, steps, warm, w, more)
                extra = total_time(t_first, w, more) - t_first  # what is spent after the first step
                assert extra <= max(0.0, left) + 1e-9, (t_first, left, steps, warm, w, more)


def test_planner_on_the_driver_command():
    # three blocks per CU: one step = 512 s, 153 s left after the reserves -> the first step is the timed one
    assert bench.plan_steps(153.0, 512.0, 20, 5) == (0, 0)
    # multi-GPU ranks have no extra legs: 750 s left -> the first step becomes the warmup, one timed step follows
    assert bench.plan_steps(750.0, 512.0, 20, 5) == (1, 1)
    # small blocks: everything that was asked for
    assert bench.plan_steps(1200.0, 2.0, 20, 5) == (5, 20)
    # no warmup requested: the first step counts
    assert bench.plan_steps(1200.0, 2.0, 3, 0) == (0, 2)
    assert bench.plan_steps(1200.0, 2.0, 1, 0) == (0, 0)

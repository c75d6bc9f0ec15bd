"""The chunk controller of launch_render (actinon_amd/csrc/acn_chunkplan.h) on the CPU: a call is cut into chunks of positions
sized from learned queue demand; a chunk whose records overflow a queue is redone smaller.  Rendering is deterministic, so a
retry that is not strictly smaller than the chunk that overflowed repeats for ever -- round 3's controller could do that once
fill_target had decayed below 0.425 (the "take all that is left" rule undid the halving).  The loop below is
launch_render's, with the device replaced by a demand function."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Ctl(C.Structure):
    _fields_ = [("fill_target", C.c_double), ("fits_in_a_row", C.c_uint32), ("retry_bound", C.c_uint32)]


@pytest.fixture(scope="module")
def plan():
    src = os.path.join(ROOT, "tests", "csrc", "chunkplan_cpu.c")
    hdr = os.path.join(ROOT, "actinon_amd", "csrc", "acn_chunkplan.h")
    out = os.path.join(ROOT, "build", "libchunkplan_cpu.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-Wall", "-I", os.path.join(ROOT, "actinon_amd", "csrc"), src, "-o", out])
    lib = C.CDLL(out)
    lib.plan_init.argtypes = [C.POINTER(Ctl), C.c_double]
    lib.plan_next.argtypes = [C.POINTER(Ctl), C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    lib.plan_next.restype = C.c_uint32
    lib.plan_overflow.argtypes = [C.POINTER(Ctl), C.c_uint32]
    lib.plan_overflow.restype = C.c_size_t
    lib.plan_fit.argtypes = [C.POINTER(Ctl)]
    return lib


def run_call(lib, n_slots, demand, cap, fill_target, rate0, fixed=0, max_iters=10000):
    """launch_render's loop.  demand( base, cnt ) -> records the chunk puts into each of the 5 queues (deterministic).
    Returns (chunks, retries, list of (base, cnt, overflowed))."""
    ctl = Ctl()
    lib.plan_init(C.byref(ctl), fill_target)
    rate = (C.c_double * 5)(*rate0)
    caps = (C.c_uint32 * 5)(*cap)
    known = any(r > 0 for r in rate0)

    def chunk_for_caps():
        return max(64, int(min(ctl.fill_target * caps[q] / max(rate[q], 1e-3) for q in range(5))))

    chunk = chunk_for_caps() if known else 1024
    base, log, chunks, retries = 0, [], 0, 0
    for _ in range(max_iters):
        if base >= n_slots:
            return chunks, retries, log
        cnt = lib.plan_next(C.byref(ctl), n_slots - base, chunk, fixed, 1 if known else 0, rate, caps)
        assert 1 <= cnt <= n_slots - base
        fill = demand(base, cnt)
        over = any(fill[q] > caps[q] for q in range(5))
        log.append((base, cnt, over))
        if over:
            assert cnt > 1, "a single position overflows: the call fails (ACN_ERR_DEVICE), it does not loop"
            retries += 1
            chunk = lib.plan_overflow(C.byref(ctl), cnt)
            for q in range(5):
                rate[q] = max(rate[q], min(fill[q], caps[q]) / cnt)   # the marks of a lost chunk are lower bounds
            known = True
            continue
        chunks += 1
        base += cnt
        lib.plan_fit(C.byref(ctl))
        for q in range(5):
            rate[q] = max(fill[q] / cnt, 0.85 * rate[q], 1e-3)
        known = True
        if not fixed:
            chunk = chunk_for_caps()
    raise AssertionError(f"the controller did not finish in {max_iters} iterations: last {log[-5:]}")


def test_retry_is_always_smaller_than_the_chunk_that_overflowed(plan):
    """ADVICE r03's case: fill_target at its floor, the last (take-all) chunk planned 18 % under its real demand."""
    cap = [100000] * 5
    # 10 000 positions; the last 3 000 are dense (4x the demand the rates were learned on)
    def demand(base, cnt):
        pos = np.arange(base, base + cnt)
        per = np.where(pos >= 7000, 40.0, 10.0)
        return [int(per.sum())] * 5
    chunks, retries, log = run_call(plan, 10000, demand, cap, fill_target=0.3, rate0=[10.0] * 5)
    assert retries >= 1
    for (b0, c0, o0), (b1, c1, o1) in zip(log, log[1:]):
        if o0:
            assert b1 == b0 and c1 <= c0 // 2, (b0, c0, b1, c1)
    assert sum(c for b, c, o in log if not o) == 10000


def test_take_all_needs_a_prediction_that_fits(plan):
    """The rest of a call is taken in one chunk only while rate * remaining stays within 85 % of every queue."""
    ctl = Ctl()
    plan.plan_init(C.byref(ctl), 0.3)
    rate = (C.c_double * 5)(10, 1, 1, 1, 1)
    caps = (C.c_uint32 * 5)(100000, 100000, 100000, 100000, 100000)
    # planned chunk 3 000 (0.3 * cap / rate); 8 000 left would fill queue 0 to 80 %: taken whole; 9 000 (90 %) is not
    assert plan.plan_next(C.byref(ctl), 8000, 3000, 0, 1, rate, caps) == 8000
    assert plan.plan_next(C.byref(ctl), 9000, 3000, 0, 1, rate, caps) == 3000
    # a pinned chunk size (ACN_CHUNK) is never exceeded; nothing learned yet: at most 1.2 x the first guess
    assert plan.plan_next(C.byref(ctl), 8000, 3000, 1, 1, rate, caps) == 3000
    assert plan.plan_next(C.byref(ctl), 3500, 3000, 0, 0, rate, caps) == 3500
    assert plan.plan_next(C.byref(ctl), 3700, 3000, 0, 0, rate, caps) == 3000
    # while a retry is pending nothing is larger than half of what overflowed, whatever the prediction says
    assert plan.plan_overflow(C.byref(ctl), 8000) == 4000
    assert plan.plan_next(C.byref(ctl), 8000, 4000, 0, 1, rate, caps) == 4000
    assert plan.plan_next(C.byref(ctl), 8000, 100000, 0, 1, rate, caps) == 4000
    plan.plan_fit(C.byref(ctl))
    assert plan.plan_next(C.byref(ctl), 8000, 3000, 0, 1, rate, caps) == 8000


@pytest.mark.parametrize("seed", range(8))
def test_random_heavy_tailed_demand_always_terminates(plan, seed):
    """Heavy-tailed demand per tile (hanging_lamp: 15 - 35 % of the chunks are redone): every position is rendered exactly once,
    and the number of lost chunks stays bounded by the halvings."""
    rng = np.random.default_rng(seed)
    n = 50000
    per_tile = rng.pareto(1.5, size=(n + 255) // 256) * 20 + 5
    per = np.repeat(per_tile, 256)[:n]
    csum = np.concatenate([[0.0], np.cumsum(per)])
    cap = [200000, 150000, 300000, 100000, 250000]
    scale = [1.0, 0.5, 2.0, 0.2, 1.5]
    def demand(base, cnt):
        t = csum[base + cnt] - csum[base]
        return [int(t * s) for s in scale]
    chunks, retries, log = run_call(plan, n, demand, cap, fill_target=0.7, rate0=[0.0] * 5)
    assert sum(c for b, c, o in log if not o) == n
    assert retries <= 20 * chunks + 40
    # fixed chunk size: same guarantees
    chunks, retries, log = run_call(plan, n, demand, cap, fill_target=0.7, rate0=[0.0] * 5, fixed=1)
    assert sum(c for b, c, o in log if not o) == n

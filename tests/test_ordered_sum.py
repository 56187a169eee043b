"""The mean kernel's reproduction of a one-at-a-time f64 sum (SSIM.h:66-71: `sum += v` over the blocks in raster order,
then one division), on the kernel simulator: whatever the values, the result has the bits of the plain loop."""
import math
import struct

import numpy as np
import pytest

import simlib as S


def plain_mean(x):
    s = 0.0
    for v in np.asarray(x, np.float64).tolist():
        s += v
    with np.errstate(all="ignore"):
        return float(np.float64(s) / np.float64(len(x)))


def same_bits(a, b):
    if math.isnan(a) or math.isnan(b):
        return math.isnan(a) and math.isnan(b)
    return struct.pack("<d", a) == struct.pack("<d", b)


def sim_mean(x):
    return S.ordered_mean(x)


MEAN = sim_mean  # the GPU test runs the same cases through khg_ordered_mean_dev


def check(x, min_integer_windows=None):
    x = np.ascontiguousarray(x, np.float64)
    with np.errstate(all="ignore"):
        got, stat = MEAN(x)
    want = plain_mean(x)
    assert same_bits(got, want), (got, want, stat)
    if stat is None:  # no counters on the GPU
        return [1 << 30, 0, 0]
    if min_integer_windows is not None:
        assert stat[0] >= min_integer_windows, stat
    return stat


@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 255, 511, 513, 4095, 4096, 4097, 8193, 40000])
def test_block_values_like_an_image_gives(n):
    rng = np.random.default_rng(n)
    check(np.clip(0.9 + 0.1 * rng.standard_normal(n), -1.0, 1.0))


def test_a_frame_worth_of_blocks_is_summed_as_integers():
    """518400 blocks (7680x4320), 127 windows of 4096: all but the windows around the ~20 binade changes and the few
    early ties take the integer path."""
    rng = np.random.default_rng(1)
    x = np.clip(0.85 + 0.2 * rng.random(518400), -1.0, 1.0)
    stat = check(x)
    assert stat[0] >= 100 and stat[1] <= 40 and stat[2] <= 40 * 64, stat


def test_identical_pictures():
    check(np.ones(400000), min_integer_windows=25)  # every block 1.0: exact all the way


def test_uniform_values_tie_often_at_the_start():
    """Values in [0.5, 1) have their last bit at 2^-53: while the sum is in [1, 2) half of them tie."""
    for seed in range(20):
        rng = np.random.default_rng(seed)
        check(0.5 + 0.5 * rng.random(300))


def test_every_addition_a_tie():
    """x[i] = an odd multiple of half the sum's unit in the last place: every step rounds by the parity of the sum,
    and the windows go through the composed parity maps instead of one addition per tie."""
    rng = np.random.default_rng(3)
    s = 1000.0
    x = [s]
    for i in range(20000):
        half_ulp = math.ulp(s) / 2
        v = (2 * int(rng.integers(0, 1 << 20)) + 1) * half_ulp
        if rng.random() < 0.3:
            v = -v
        if not 512.0 < s + v < 1e6:
            v = -v
        x.append(v)
        s += v
    stat = check(x)
    assert stat[2] < 200, stat  # ties did not become single additions


def test_mixed_signs_and_cancellation():
    rng = np.random.default_rng(4)
    check(rng.standard_normal(30000))                     # the sum wanders through zero
    x = rng.standard_normal(20000) * 10.0 ** rng.integers(-12, 12, 20000)
    check(x)                                              # magnitudes from 1e-12 to 1e12
    y = np.empty(20000)
    y[0::2] = 1e10 + rng.random(10000)
    y[1::2] = -1e10 + rng.random(10000)
    check(y)                                              # every other step cancels nearly everything
    check(np.concatenate([[1.0], np.full(5000, -2.0 ** -53), [2.0 ** -30] * 100]))  # creeping down out of a binade
    check(np.concatenate([[1.0, -1.0], rng.random(3000), [-3.0] * 2000]))           # exact zero, later negative sums


def test_tiny_and_special_values():
    rng = np.random.default_rng(5)
    tiny = np.array([5e-324, 2.2e-308, -5e-324, 1e-300, 0.0, -0.0] * 50)
    check(tiny)                                           # the sum itself stays subnormal / tiny
    check(np.concatenate([rng.random(3000), tiny, rng.random(3000)]))
    check(np.concatenate([[0.0] * 3000, [1.0], [0.0] * 3000]))
    check(np.concatenate([[-0.0] * 10]))
    check(np.concatenate([rng.random(5000), [1e308, 1e308], rng.random(100)]))     # overflow to infinity
    check(np.concatenate([rng.random(5000), [np.inf], rng.random(100), [-np.inf], rng.random(50)]))  # NaN from there on
    check(np.concatenate([rng.random(100), [np.nan], rng.random(5000)]))
    check(np.concatenate([[1.7e308], rng.random(3000) * 1e292]))                    # near the top of the range
    check(np.concatenate([[2.0 ** 60], rng.random(5000), [-(2.0 ** 60)], rng.random(5000)]))  # values below u / 4


def test_random_bit_patterns():
    """Doubles drawn as raw bit patterns over a band of exponents, in runs of random length."""
    rng = np.random.default_rng(6)
    for rep in range(30):
        n = int(rng.integers(1, 30000))
        e0 = int(rng.integers(900, 1100))
        span = int(rng.integers(1, 70))
        exp = rng.integers(e0, e0 + span, n).astype(np.uint64)
        mant = rng.integers(0, 1 << 52, n, dtype=np.uint64)
        if rep % 3 == 0:
            mant &= ~np.uint64((1 << int(rng.integers(1, 50))) - 1)  # short mantissas: exact sums and ties
        sign = rng.integers(0, 2, n).astype(np.uint64) if rep % 2 else np.zeros(n, np.uint64)
        x = ((sign << np.uint64(63)) | (exp << np.uint64(52)) | mant).view(np.float64)
        check(x)


ALL_CASES = [test_a_frame_worth_of_blocks_is_summed_as_integers, test_identical_pictures,
             test_uniform_values_tie_often_at_the_start, test_every_addition_a_tie, test_mixed_signs_and_cancellation,
             test_tiny_and_special_values, test_random_bit_patterns]


@pytest.mark.gpu
def test_the_same_cases_on_the_gpu(monkeypatch):
    """Every case above through khg_ordered_mean_dev (the kernel khg_ssim_dev ends with)."""
    import ctypes as C
    import sys

    import torch
    from khugle_amd.lib import load
    L = load()

    def gpu_mean(x):
        d = torch.from_numpy(np.ascontiguousarray(x, np.float64)).cuda()
        out = torch.zeros(1, dtype=torch.float64, device="cuda")
        rc = L.khg_ordered_mean_dev(C.c_void_p(d.data_ptr()), d.numel(), C.c_void_p(out.data_ptr()), None)
        assert rc == 0
        torch.cuda.synchronize()
        return float(out.cpu()[0]), None

    monkeypatch.setattr(sys.modules[__name__], "MEAN", gpu_mean)
    for n in (1, 2, 7, 8, 9, 255, 511, 513, 4095, 4096, 4097, 8193, 40000):
        test_block_values_like_an_image_gives(n)
    for case in ALL_CASES:
        case()

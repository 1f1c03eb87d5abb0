"""The hand-written device radix sort (strling_amd/csrc/sort.hip) against numpy's stable sort."""
import numpy as np
import pytest

from strling_amd import api

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def _check(ctx, keys, bit_lo, bits, n_max=None):
    keys = np.ascontiguousarray(keys, np.uint64)
    vals = np.arange(keys.size, dtype=np.uint32)
    k, v = ctx.sort_pairs(keys, vals, bit_lo, bits, n_max)
    field = (keys >> np.uint64(bit_lo)) & np.uint64((1 << bits) - 1) if bits < 64 else keys >> np.uint64(bit_lo)
    order = np.argsort(field, kind="stable")
    assert np.array_equal(v, order.astype(np.uint32))
    assert np.array_equal(k, keys[order])


@pytest.mark.parametrize("n", [1, 63, 64, 2047, 2048, 2049, 100_000, 524_288, 524_289, 1_500_000])
def test_sort_random_keys(ctx, n):
    rng = np.random.Generator(np.random.Philox(n))
    _check(ctx, rng.integers(0, 2 ** 63, size=n, dtype=np.uint64) * 2 + rng.integers(0, 2, size=n, dtype=np.uint64), 0, 64)


def test_sort_is_stable_and_honours_the_bit_window(ctx):
    rng = np.random.Generator(np.random.Philox(5))
    n = 300_000
    keys = rng.integers(0, 50, size=n, dtype=np.uint64) << np.uint64(20) | rng.integers(0, 2 ** 20, size=n, dtype=np.uint64)
    _check(ctx, keys, 20, 6)          # many ties: value order must be the input order
    _check(ctx, keys, 0, 20)
    _check(ctx, keys, 3, 13)          # a window that is not a multiple of 8 bits
    _check(ctx, keys, 0, 44, n_max=4 * n)   # launch sized by an upper bound of the count


def test_sort_skewed_digits_and_large_two_level(ctx):
    rng = np.random.Generator(np.random.Philox(6))
    n = 3_000_000                       # > 2^21: histogram kernel per pass, several chunks
    keys = np.where(rng.random(n) < 0.7, np.uint64(7) << np.uint64(32), rng.integers(0, 2 ** 40, size=n, dtype=np.uint64))
    keys |= rng.integers(0, 2 ** 12, size=n, dtype=np.uint64)
    _check(ctx, keys, 0, 40)


def test_sort_above_the_atomic_histogram_limit(ctx):
    rng = np.random.Generator(np.random.Philox(8))
    n = 9_000_000                       # > 2^23: one histogram kernel per pass
    _check(ctx, rng.integers(0, 2 ** 24, size=n, dtype=np.uint64), 0, 24)

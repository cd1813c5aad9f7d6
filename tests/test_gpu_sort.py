"""The library's own radix sort (csrc/gpis_sort.hip) against numpy's stable argsort: sizes around the tile boundaries, full-range
and few-bit keys (long runs of equal keys exercise the stability the wavefront march relies on: equal lattice cells keep their ray
order, so results do not depend on the sort)."""
import numpy as np
import pytest

import _gpis_pkg

pytestmark = pytest.mark.gpu
pkg = _gpis_pkg.load_package()


@pytest.mark.parametrize("n", [0, 1, 63, 64, 2047, 2048, 2049, 4096 * 3 + 17, 1_000_003, 9_000_001])
@pytest.mark.parametrize("bits", [32, 11, 3])
def test_sort_pairs_is_a_stable_sort(n, bits):
    rng = np.random.default_rng(n * 7 + bits)
    keys = rng.integers(0, 2**bits, size=n, dtype=np.uint64).astype(np.uint32)
    if bits == 32 and n > 10:
        keys[: n // 3] &= np.uint32(0xFF00FF00)          # digits that are empty in some passes
    vals = np.arange(n, dtype=np.uint32)
    k, v = pkg.sort_pairs_u32(keys, vals)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(k, keys[order]) and np.array_equal(v, vals[order])

"""CPU: pin the Philox restatement with Random123's published known-answer vectors; ring semantics."""
import numpy as np

from oracle import replay_ref as R


def _kat(ctr, key):
    out = R.philox4x32_10(*[np.array([c], dtype=np.uint64) for c in ctr], key[0], key[1])
    return [int(x[0]) for x in out]


def test_philox4x32_10_known_answers():
    # Random123 kat_vectors, "philox4x32 10" rows
    assert _kat((0, 0, 0, 0), (0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _kat((0xffffffff,) * 4, (0xffffffff, 0xffffffff)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _kat((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_index_draw_bounds_and_uniformity():
    for length in (1, 7, 1000, 100_000):
        idx = R.sample_indices(seed=5, sample_ctr=3, batch=4096, length=length)
        assert idx.min() >= 0 and idx.max() < length
    idx = np.concatenate([R.sample_indices(11, c, 4096, 64) for c in range(32)])
    counts = np.bincount(idx, minlength=64)
    chi2 = ((counts - counts.mean()) ** 2 / counts.mean()).sum()
    assert chi2 < 120  # 63 dof: P(chi2 > 120) ~ 2e-5
    assert not np.array_equal(R.sample_indices(11, 0, 256, 1000), R.sample_indices(11, 1, 256, 1000))
    assert np.array_equal(R.sample_indices(11, 4, 256, 1000), R.sample_indices(11, 4, 256, 1000))


def test_normals_moments():
    z = R.normals(seed=1, ctr=0, site_code=16, n_elems=200_000)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    assert abs((z ** 3).mean()) < 0.03 and abs((z ** 4).mean() - 3) < 0.1


def test_ring_wraparound():
    ring = R.RingRef(10, 3, 2)
    rows = np.arange(14 * 3, dtype=np.float32).reshape(14, 3)
    ring.extend(rows, np.zeros((14, 2)), np.arange(14), rows + 100, np.zeros(14))
    assert ring.len == 10 and ring.cursor == 4
    g = ring.gather([0, 3, 4, 9])
    assert list(g["rewards"]) == [10, 13, 4, 9]

"""CPU checks of the batched-RANSAC test infrastructure (no GPU): the Philox reference against the
Random123 known-answer vectors, and the distinct-index mapping."""
import numpy as np

from philox_ref import philox4x32_10, sample4


def test_philox4x32_10_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds: counter(4) key(2) -> output(4)
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = philox4x32_10(np.array(ctr, dtype=np.uint64), key)
        assert tuple(int(v) for v in got) == want


def test_sample4_distinct_in_range_and_uniformish():
    for m in (4, 5, 33, 185, 1000):
        t = sample4(1234567, 3, 20000, m)
        assert t.min() >= 0 and t.max() < m
        s = np.sort(t, axis=1)
        assert (np.diff(s, axis=1) > 0).all()            # four distinct indices
        counts = np.bincount(t.ravel(), minlength=m)
        expect = t.size / m
        assert abs(counts - expect).max() < 6 * np.sqrt(expect) + 1
    assert (sample4(1, 0, 10, 3) == 0).all()
    # a problem's table depends on (seed, problem) only
    assert np.array_equal(sample4(99, 2, 50, 40), sample4(99, 2, 100, 40)[:50])
    assert not np.array_equal(sample4(99, 2, 50, 40), sample4(99, 3, 50, 40))
    assert not np.array_equal(sample4(99, 2, 50, 40), sample4(100, 2, 50, 40))

"""Pin the Philox4x32-10 restatement against the Random123 known-answer vectors (kat_vectors:
philox4x32 10 ...) and check the uniform mappings.  CPU only."""
import numpy as np

from oracle import philox as P


def _kat(ctr, key):
    r = P.philox4x32_10(*[np.array([c], dtype=np.uint64) for c in ctr], key[0], key[1])
    return [int(w[0]) for w in r]


def test_random123_known_answers():
    assert _kat((0, 0, 0, 0), (0, 0)) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert _kat((0xFFFFFFFF,) * 4, (0xFFFFFFFF, 0xFFFFFFFF)) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert _kat((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0)) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_uniform_maps_are_exact_and_in_range():
    x = np.array([0, 255, 256, 0xFFFFFFFF, 0x80000000], dtype=np.uint32)
    u = P.u01(x)
    assert u.dtype == np.float32 and u[0] == 0 and u[1] == 0 and u[2] == np.float32(2.0 ** -24)
    assert u[3] < 1.0 and u[4] == 0.5
    s = P.usym(x)
    assert s.min() >= -1.0 and s.max() < 1.0 and s[4] == 0.0


def test_streams_are_distinct_and_shard_invariant():
    a = P.reset_draws(7, np.arange(8), 60, 0)
    b = P.reset_draws(7, np.arange(4, 8), 60, 0)
    np.testing.assert_array_equal(a[4:], b)            # keyed by GLOBAL env id
    assert not np.array_equal(a[0], a[1])
    n0 = P.step_noise(7, np.arange(4), 60, 0, 0.2)
    n1 = P.step_noise(7, np.arange(4), 60, 1, 0.2)
    assert not np.array_equal(n0, n1) and np.abs(n0).max() <= 0.1
    big = P.step_noise(3, np.arange(64), 64, 5, 0.2)
    assert abs(float(big.mean())) < 0.01                # uniform on [-0.1, 0.1)
    act = P.random_action(7, np.arange(1000), 3)
    assert act.shape == (1000, 2) and abs(float(act.mean())) < 0.05

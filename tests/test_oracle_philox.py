"""The noise generator of the synthetic workloads: the NumPy restatement of
Philox4x32-10 against the known-answer vectors published with the Random123 library
(kat_vectors, philox4x32 with 10 rounds), and the properties the workloads rely on."""
import numpy as np

from oracle import philox

KAT = [  # counter, key, output
    ((0x00000000,) * 4, (0x00000000,) * 2, (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_philox4x32_10_known_answers():
    for ctr, key, out in KAT:
        got = tuple(int(x) for x in philox.philox4x32_10(ctr, key))
        assert got == out, [hex(v) for v in got]
    # vectorised evaluation gives the scalar answers
    ctrs = [np.array([k[0][j] for k in KAT[:1] * 3]) for j in range(4)]
    got = philox.philox4x32_10(ctrs, KAT[0][1])
    assert all(int(got[j][2]) == KAT[0][2][j] for j in range(4))


def test_counter_layout_and_sharding_invariance():
    seed = (4 << 32) + 17
    full = philox.bits(seed, 0, 12, 0, 5, 13)
    assert full.shape == (5, 12, 13, 4) and full.dtype == np.uint32
    # (instance, step, lane) -> counter words (instance lo, instance hi, step, lane); key = (seed lo, seed hi)
    w = philox.philox4x32_10((7, 0, 3, 11), (17, 4))
    assert tuple(full[3, 7, 11]) == tuple(int(x) for x in w)
    # any block of instances / steps generated alone is the slice of the whole
    part = philox.bits(seed, 5, 4, 2, 3, 13)
    assert np.array_equal(part, full[2:5, 5:9])
    assert np.array_equal(philox.randn(seed, 5, 4, 2, 3, 13), philox.randn(seed, 0, 12, 0, 5, 13)[2:5, 5:9])
    # instances beyond 2^32 use the second counter word
    hi = philox.bits(seed, (1 << 32) + 7, 1, 3, 1, 12)[0, 0, 11]
    assert tuple(hi) == tuple(int(x) for x in philox.philox4x32_10((7, 1, 3, 11), (17, 4)))
    assert not np.array_equal(hi, full[3, 7, 11])


def test_normals_have_unit_moments():
    z = philox.randn(4 << 32, 0, 4096, 0, 8, 13).ravel()
    n = z.size
    assert np.isfinite(z).all() and abs(z.mean()) < 4 / np.sqrt(n) and abs(z.var() - 1) < 4 * np.sqrt(2 / n)
    assert abs((z ** 3).mean()) < 4 * np.sqrt(15 / n) and abs((z ** 4).mean() - 3) < 4 * np.sqrt(96 / n)
    # lanes and steps are uncorrelated streams
    a = philox.randn(1, 0, 20000, 0, 2, 2)
    assert abs(np.corrcoef(a[0, :, 0], a[0, :, 1])[0, 1]) < 0.03 and abs(np.corrcoef(a[0, :, 0], a[1, :, 0])[0, 1]) < 0.03

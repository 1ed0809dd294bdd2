"""TEST ORACLE ONLY -- NumPy restatement of the noise generator of the synthetic workloads
("RNG = Philox4x32-10, key = seed, counter = (instance, step, lane); normals via
Box-Muller").  Never imported by the product (dvi_ekf_amd/); only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may use anything under oracle/.

Philox4x32-10: Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11
(the Random123 library).  Pinned by the three known-answer vectors that library publishes for
philox4x32 with 10 rounds (tests/test_oracle_philox.py).
"""
import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57          # round multipliers
W0, W1 = 0x9E3779B9, 0xBB67AE85          # key increments (golden ratio, sqrt(3) - 1)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr = 4 array-likes of 32-bit words, key = 2 words.  Returns 4 uint32 arrays."""
    c = [np.asarray(x, dtype=np.uint64) & _MASK for x in ctr]
    k0, k1 = np.uint64(int(key[0]) & 0xFFFFFFFF), np.uint64(int(key[1]) & 0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(M0) * c[0]
        p1 = np.uint64(M1) * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0 = (k0 + np.uint64(W0)) & _MASK
        k1 = (k1 + np.uint64(W1)) & _MASK
    return [x.astype(np.uint32) for x in c]


def bits(seed, first_instance, n_instances, first_step, n_steps, lanes):
    """uint32 [n_steps, n_instances, lanes, 4]: counter = (instance lo, instance hi, step, lane), key = seed."""
    s, i, l = np.meshgrid(np.arange(first_step, first_step + n_steps, dtype=np.uint64),
                          np.arange(first_instance, first_instance + n_instances, dtype=np.uint64),
                          np.arange(lanes, dtype=np.uint64), indexing="ij")
    r = philox4x32_10((i & _MASK, i >> np.uint64(32), s, l), (int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF))
    return np.stack(r, axis=-1)


def randn(seed, first_instance, n_instances, first_step, n_steps, lanes):
    """float64 [n_steps, n_instances, lanes]: Box-Muller (cosine branch) on two 53-bit uniforms,
    u1 = (k1 + 1) / 2^53 in (0,1], u2 = k2 / 2^53 in [0,1), k = (word_a << 21) | (word_b >> 11)."""
    r = bits(seed, first_instance, n_instances, first_step, n_steps, lanes).astype(np.uint64)
    k1 = (r[..., 0] << np.uint64(21)) | (r[..., 1] >> np.uint64(11))
    k2 = (r[..., 2] << np.uint64(21)) | (r[..., 3] >> np.uint64(11))
    u1 = (k1 + np.uint64(1)).astype(np.float64) * 2.0 ** -53
    u2 = k2.astype(np.float64) * 2.0 ** -53
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(6.283185307179586476925286766559 * u2)

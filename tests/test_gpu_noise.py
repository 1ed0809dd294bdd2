"""dvi_eskf_randn on the GPU against oracle/philox.py (Philox4x32-10, key = seed,
counter = (instance, step, lane), Box-Muller) and the property the sharded workloads rest on: what
an instance draws depends on its global index only."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import philox


def test_philox_words_are_bit_exact():
    from dvi_ekf_amd import noise
    seed = (4 << 32) + 0x9E3779B9
    for first_i, n_i, first_s, n_s, lanes in ((0, 37, 0, 5, 13), ((1 << 32) - 3, 8, 0xFFFFFFFF, 1, 6), (123456, 1, 7, 1, 1)):
        got = noise.philox_bits(seed, first_i, n_i, first_s, n_s, lanes).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, philox.bits(seed, first_i, n_i, first_s, n_s, lanes))
    # the published known-answer vector for the all-zero counter and key, through the device kernel
    z = noise.philox_bits(0, 0, 1, 0, 1, 1).cpu().numpy().view(np.uint32).ravel()
    assert [int(v) for v in z] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]


def test_normals_match_the_oracle_and_do_not_depend_on_the_shard():
    from dvi_ekf_amd import noise
    seed = 4 << 32
    full = noise.randn(seed, 0, 4096, 0, 6, 13)
    ref = philox.randn(seed, 0, 4096, 0, 6, 13)
    err = np.abs(full.cpu().numpy() - ref).max()
    print(f"device normals vs NumPy restatement: max abs diff {err:.1e} over {ref.size} draws, |z| up to {np.abs(ref).max():.2f}")
    assert err <= 4e-15                               # integer part is bit-exact; log / sqrt / cos differ in the last ulp
    # any block generated alone is bitwise the slice of the whole: ranks, batches, launch shape do not matter
    part = noise.randn(seed, 1000, 333, 2, 3, 13)
    assert torch.equal(part, full[2:5, 1000:1333])
    one = noise.randn(seed, 4095, 1, 5, 1, 13)
    assert torch.equal(one[0, 0], full[5, 4095])
    # fp32 output is the rounded fp64 draw
    f32 = noise.randn(seed, 0, 4096, 0, 6, 13, dtype=torch.float32)
    assert torch.equal(f32, full.float())
    # moments over 2.4e6 draws
    big = noise.randn(seed + 1, 0, 1 << 16, 0, 3, 13).double().flatten()
    n = big.numel()
    assert abs(float(big.mean())) < 4 / np.sqrt(n) and abs(float(big.var()) - 1) < 4 * np.sqrt(2 / n)
    assert bool(torch.isfinite(big).all()) and float(big.abs().max()) < 7.0


def test_argument_validation():
    from dvi_ekf_amd import _capi
    L = _capi.lib()
    buf = torch.empty(16, dtype=torch.float64, device="cuda")
    p = C.c_void_p(buf.data_ptr())
    assert L.dvi_eskf_randn(0, 0, 1, 0, 0, 0, 4, 4, p, None) == 0            # nothing to draw
    assert L.dvi_eskf_randn(0, 0, 1, 0, 2, 0, 2, 4, None, None) == -1        # no output buffer
    assert L.dvi_eskf_randn(3, 0, 1, 0, 2, 0, 2, 4, p, None) == -1           # unknown kind
    assert L.dvi_eskf_randn(0, 0, 1, -1, 2, 0, 2, 4, p, None) == -1          # negative instance
    assert L.dvi_eskf_randn(0, 0, 1, 0, 2, 0xFFFFFFFF, 2, 4, p, None) == -1  # step beyond its counter word
    assert L.dvi_eskf_randn(0, 4096, 1, 0, 2, 0, 2, 4, p, None) == -2        # no such device


def test_monte_carlo_workload_is_invariant_under_sharding():
    """Instances 5000..5127 of the Monte-Carlo batch generated as their own shard (what rank r of a
    multi-GPU run does) are bitwise the rows of the unsharded batch, and so is the filter's output."""
    from dvi_ekf_amd.workloads import monte_carlo_pairs
    whole = monte_carlo_pairs(8192, horizon=4, seed=4, device="cuda")
    shard = monte_carlo_pairs(128, horizon=4, seed=4, device="cuda", first_instance=5000)
    sl = slice(5000, 5128)
    assert torch.equal(shard.x0, whole.x0[sl])
    for name in ("om", "acc", "pos", "quat", "notch"):
        assert torch.equal(getattr(shard, name), getattr(whole, name)[:, sl]), name
    # global instance 0 is the noise-free reference run; every other instance is perturbed
    assert torch.equal(whole.x0[0].cpu(), torch.as_tensor(whole.ri.x0))
    assert bool((whole.x0[1:, 10:16] != whole.x0[0, 10:16]).any(dim=1).all())
    fw, fs = whole.make_filter(), shard.make_filter()
    for k in range(4):
        fw.step(whole.dt[k], whole.om[k], whole.acc[k], whole.pos[k], whole.quat[k], whole.notch[k])
        fs.step(shard.dt[k], shard.om[k], shard.acc[k], shard.pos[k], shard.quat[k], shard.notch[k])
    assert torch.equal(fs.x, fw.x[sl]) and torch.equal(fs.calibration(), fw.calibration()[sl])
    # noise levels are the configured ones
    sd = (whole.om - whole.om[:, :1]).std(dim=1).mean(dim=0).cpu().numpy()
    assert np.allclose(sd, whole.sigma_om, rtol=0.05)

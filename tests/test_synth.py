"""CPU suite: the package's numpy coefficient generator == the oracle's C generator."""
import numpy as np
import pytest

import helpers  # noqa: F401
from oracle import pyoracle
from jpeg_decoder_amd import synth


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_numpy_generator_matches_c_generator(kind):
    for seed, first, n in ((0x5EED0001, 0, 5000), (7, 123456789, 3000), (2**40 + 3, 2**33, 1000)):
        assert np.array_equal(synth.synth_blocks(n, seed, first, kind, chunk=1024), pyoracle.synth_blocks(n, seed, first, kind))


def test_distribution_is_jpeg_like():
    b = synth.synth_blocks(20000, 1).astype(np.int64)
    assert b[:, 0].min() >= -64 and b[:, 0].max() <= 64
    zero_frac = (b[:, 1:] == 0).mean(axis=0)
    assert zero_frac[0] < 0.1 and zero_frac[-1] > 0.9
    assert np.abs(b[:, 1]).mean() > np.abs(b[:, 20]).mean()


def test_bench_qtables_are_shauns():
    import os
    z = np.load(os.path.join(helpers.GOLDEN, "shaun_the_sheep_planes.npz"))
    q = synth.shaun_qtables()
    assert np.array_equal(q[0], z["qtables"][0]) and np.array_equal(q[1], z["qtables"][1]) and q[2] is None


def test_torch_generator_gives_the_same_bytes():
    """bench.py fills its device buffers with synth.synth_blocks_torch: same bytes as the numpy generator (and hence as the
    oracle's C generator), for every kind, across chunk boundaries and with large block offsets."""
    import numpy as np
    from jpeg_decoder_amd import synth
    for kind in (0, 1, 2):
        for seed, fb in ((0x5EED0000, 0), (0x5EED0002, 123456789012), (7, 5)):
            a = synth.synth_blocks(3000, seed, fb, kind)
            b = synth.synth_blocks_torch(3000, seed, fb, kind, device="cpu", chunk=1777).numpy()
            assert np.array_equal(a, b), (kind, seed, fb)

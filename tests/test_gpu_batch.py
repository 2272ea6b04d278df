"""GPU suite for jpeggpu_decode_batch: many same-geometry files, host threads for the entropy decoding,
one batched launch per chunk.  Every raster must equal what the single-file path produces (which the
other suites pin to the reference binary) and the CPU oracle."""
import io
import os

import numpy as np
import pytest

import helpers
from test_frontend import JDIR, oracle_backend

pytestmark = pytest.mark.gpu


def encode_set(n, size, sub, prog=False, gray=False, seed=0):
    Image = pytest.importorskip("PIL.Image")
    w, h = size
    files = []
    for i in range(n):
        rng = np.random.default_rng(seed * 1000 + i)
        y, x = np.mgrid[0:h, 0:w]
        a = np.stack([(x * (i % 5 + 1) + y) % 256, (x + y * (i % 3 + 2)) % 256, (x // 2 + y * 3 + 7 * i) % 256], -1).astype(np.float64)
        a += rng.normal(0, 14, a.shape)
        im = Image.fromarray(np.clip(a, 0, 255).astype(np.uint8))
        if gray:
            im = im.convert("L")
        b = io.BytesIO()
        kw = dict(quality=40 + (i * 7) % 55, progressive=prog, optimize=bool(i % 2))   # tables differ from file to file
        if not gray:
            kw["subsampling"] = sub
        im.save(b, "JPEG", **kw)
        files.append(b.getvalue())
    return files


def single_file(F, data):
    j = F.Jpeg(data)
    while not j.eoi:
        j.decode_scan()
        if j.eoi:
            break
        j.next_header()
    return j, j.render()


@pytest.mark.parametrize("n,size,sub,prog,gray,threads", [
    (37, (200, 120), 2, False, False, 4),      # 4:2:0 baseline, one chunk
    (19, (131, 77), 1, False, False, 3),       # 4:2:2, ragged size
    (12, (64, 64), 0, False, False, 1),        # 4:4:4, one thread
    (9, (90, 50), 0, False, True, 2),          # gray
    (10, (160, 96), 2, True, False, 4),        # progressive: image after the last scan
])
def test_batch_equals_single_file_path(gpu, n, size, sub, prog, gray, threads):
    from jpeg_decoder_amd import frontend as F
    files = encode_set(n, size, sub, prog, gray, seed=n)
    out, st, d, stats = F.decode_batch(files, threads)
    assert st == [0] * n and stats.images_ok == n and stats.threads == min(threads, n)
    for i in (0, 1, n // 2, n - 1):
        j, want = single_file(F, files[i])
        assert np.array_equal(out[i], want), i
        # and the CPU oracle over the same (quantised) planes
        assert np.array_equal(out[i], oracle_backend(j, j.desc, j.desc.nbmcuV)), i


def test_many_chunks_double_buffering(gpu):
    """More images than one chunk holds (chunk = 256 MiB of coefficients): 1080p x 150 -> several launches."""
    from jpeg_decoder_amd import frontend as F
    base = encode_set(6, (1920, 1080), 2, seed=3)
    files = [base[i % 6] for i in range(150)]
    out, st, d, stats = F.decode_batch(files, 8)
    assert st == [0] * 150 and stats.chunks >= 3 and out.shape == (150, 1088, 1920 * 3)
    refs = [single_file(F, b)[1] for b in base]
    for i in range(150):
        assert np.array_equal(out[i], refs[i % 6]), i


def test_bad_files_get_their_own_status(gpu):
    from jpeg_decoder_amd import frontend as F
    good = encode_set(5, (61, 47), 2, seed=9)
    other = encode_set(1, (64, 47), 2, seed=10)[0]                       # different geometry
    dri = open(os.path.join(JDIR, "bad_dri.jpg"), "rb").read()           # ERR_MARKER_UNKNOWN in the reference
    trunc = good[2][:len(good[2]) - 40] + b"\xff\xd9"                     # scan cut short: ERR_0XFF00 or ERR_HUFF_CODE_1
    files = [good[0], dri, good[1], other, trunc, good[3], b"", good[4]]
    out, st, d, stats = F.decode_batch(files, 3)
    assert [s == 0 for s in st] == [True, False, True, False, False, True, False, True]
    assert st[1] == 34 and st[3] == gpu.ERR_INVALID_ARG and st[4] in (1, 5) and stats.images_ok == 4
    for i, g in ((0, 0), (2, 1), (5, 3), (7, 4)):
        assert np.array_equal(out[i], single_file(F, good[g])[1])


def test_no_decodable_file_is_an_error(gpu):
    from jpeg_decoder_amd import frontend as F
    dri = open(os.path.join(JDIR, "bad_dri.jpg"), "rb").read()
    with pytest.raises(gpu.JpegGpuError) as e:
        F.decode_batch([dri, dri], 2)
    assert e.value.code == 34

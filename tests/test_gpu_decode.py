"""GPU suite for the front end + back end + output stage: real JPEG files
through the product path (host VLD -> pinned planes -> fused HIP kernel -> PPM/PGM), compared with what
the reference binary wrote for the same files (tests/golden/jpeg/expect.json) and with the CPU oracle."""
import hashlib
import io
import os
import subprocess

import numpy as np
import pytest

import helpers
from test_frontend import EXPECT, JDIR, decode_with, expected_images, oracle_backend

pytestmark = pytest.mark.gpu


def gpu_backend(j, d, mcu_rows):
    return j.render()


@pytest.mark.parametrize("name", sorted(EXPECT))
def test_scan_by_scan_images_equal_reference(gpu, name):
    e = EXPECT[name]
    rc, imgs = decode_with(open(os.path.join(JDIR, name), "rb").read(), gpu_backend)
    assert rc == e["rc"], (rc, e["err"], gpu.last_error())
    assert [hashlib.sha256(b).hexdigest() for b in imgs] == expected_images(e)


def _check_outputs(outdir, e):
    got = {}
    for f in sorted(os.listdir(outdir)):
        p = os.path.join(outdir, f)
        got[f] = ("link", os.readlink(p)) if os.path.islink(p) else ("file", hashlib.sha256(open(p, "rb").read()).hexdigest())
    exp = {o["name"]: (("link", o["link"]) if "link" in o else ("file", o["sha256"])) for o in e["outputs"]}
    assert got == exp


@pytest.mark.parametrize("name", sorted(EXPECT))
def test_jpeg2ppm_writes_the_same_files(gpu, tmp_path, name):
    """jpeggpu_jpeg2ppm == main() of the reference: status, file names, bytes, final symbolic link."""
    from jpeg_decoder_amd import frontend as F
    e = EXPECT[name]
    out = tmp_path / "o"
    out.mkdir()
    rc = 0
    try:
        F.jpeg2ppm(os.path.join(JDIR, name), out / "x.pnm")
    except gpu.JpegGpuError as err:
        rc = err.code
    assert rc == e["rc"], (rc, e["err"], gpu.last_error())
    _check_outputs(out, e)


@pytest.mark.parametrize("name", ["c1_gray_512x512_q90.jpg", "c4_prog422_160x120.jpg", "bad_dri.jpg", "bad_truncated_scan.jpg",
                                  "bad_prog_truncated.jpg"])
def test_command_line_driver(gpu, tmp_path, name):
    exe = os.path.join(helpers.ROOT, "jpeg-decoder_amd", "jpeg2ppm_gpu")
    assert os.path.exists(exe), "build it with make -C jpeg-decoder_amd/csrc"
    e = EXPECT[name]
    out = tmp_path / "o"
    r = subprocess.run([exe, "-o", str(out / "x.pnm"), os.path.join(JDIR, name)], capture_output=True, timeout=120)
    assert r.returncode == e["rc"], (r.returncode, r.stderr[-300:])
    _check_outputs(out, e)


def test_timer_option_and_timed_entry(gpu, tmp_path):
    """-t prints one line per stage in print_timer's format (src/timer.c:43) and the files are unchanged."""
    from jpeg_decoder_amd import frontend as F
    exe = os.path.join(helpers.ROOT, "jpeg-decoder_amd", "jpeg2ppm_gpu")
    name = "c4_prog422_160x120.jpg"
    out = tmp_path / "o"
    r = subprocess.run([exe, "-t", "-o", str(out / "x.pnm"), os.path.join(JDIR, name)], capture_output=True, timeout=120)
    assert r.returncode == EXPECT[name]["rc"]
    lines = r.stdout.decode().strip().split("\n")
    assert len(lines) == 6 and lines[1].startswith("Décodage entête : ") and lines[-1].startswith("Temps total : ")
    assert all(l.endswith(" s") and float(l.rsplit(" : ", 1)[1][:-2]) >= 0 for l in lines)
    _check_outputs(out, EXPECT[name])
    n, t = F.jpeg2ppm_timed(os.path.join(JDIR, name), tmp_path / "p" / "y.pnm")
    assert n == sum(1 for o in EXPECT[name]["outputs"] if "link" not in o)
    assert t.total_s > 0 and t.total_s >= t.vld_s and t.init_s + t.header_s + t.vld_s + t.gpu_wait_s + t.write_s <= t.total_s * 1.001


def test_default_output_name_next_to_input(gpu, tmp_path):
    from jpeg_decoder_amd import frontend as F
    src = tmp_path / "img.jpeg"
    src.write_bytes(open(os.path.join(helpers.GOLDEN, "shaun_the_sheep.jpeg"), "rb").read())
    assert F.jpeg2ppm(src) == 1
    assert (tmp_path / "img.ppm").read_bytes() == open(os.path.join(helpers.GOLDEN, "shaun_the_sheep.ppm"), "rb").read()


@pytest.mark.parametrize("size,sub,prog", [((1024, 768), 2, True), ((1500, 1001), 1, True), ((2048, 2048), 0, False),
                                            ((1920, 1080), 2, False),
                                            ((4096, 4096), 1, True)],   # BASELINE.json configs[3]: progressive 4:2:2, 10 scans, end to end
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else str(v))
def test_larger_files_against_cpu_oracle(gpu, size, sub, prog):
    """Files encoded on the spot (Pillow): every scan's image from the GPU == the CPU oracle's."""
    Image = pytest.importorskip("PIL.Image")
    from jpeg_decoder_amd import frontend as F
    rng = np.random.default_rng(size[0] * 7 + sub)
    w, h = size
    y, x = np.mgrid[0:h, 0:w]
    a = np.stack([(x + y) % 256, (x * 3 + y // 2) % 256, (x // 3 + y * 2) % 256], -1).astype(np.float64)
    a += rng.normal(0, 12, a.shape)
    b = io.BytesIO()
    Image.fromarray(np.clip(a, 0, 255).astype(np.uint8)).save(b, "JPEG", quality=88, subsampling=sub, progressive=prog)
    j = F.Jpeg(b.getvalue())
    d = j.desc
    nscan = 0
    while not j.eoi:
        j.decode_scan()
        nscan += 1
        got = j.render()
        if nscan in (1, 2, 6) or j.eoi or not prog:   # the oracle costs ~20 ms/MPix: check a few scans and the last
            assert np.array_equal(got, oracle_backend(j, d, d.nbmcuV)), nscan
        last = got
        if j.eoi:
            break
        j.next_header()
    assert nscan == (10 if prog else 1)
    assert np.array_equal(last, oracle_backend(j, d, d.nbmcuV))


@pytest.mark.parametrize("name", ["ref_shaun_the_sheep.jpeg", "ref_invader_normal.jpeg", "prog_420_61x47_q80.jpg", "base_gray_35x70_q30_opt.jpg"])
def test_reference_drivers_linked_against_the_compat_library(gpu, tmp_path, name):
    """SURVEY 8b, link-compatibility face: the reference's own, unmodified baseline.c / progressive.c /
    decoder_utils.c objects, linked against libjpeggpu_compat.so (iquant, izz, idct_opt, ycc2rgb_pixel
    -> one GPU round trip each) instead of iqzz.c / idct_opt.c / ycc2rgb.c, write the files the all-CPU
    reference wrote.  oracle/_ref/jpeg2ppm_compat is built in the build container (make -C oracle compat)."""
    exe = os.path.join(helpers.ROOT, "oracle", "_ref", "jpeg2ppm_compat")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/jpeg2ppm_compat was not built")
    e = EXPECT[name]
    out = tmp_path / "o"
    r = subprocess.run([exe, "-o", str(out / "x.pnm"), os.path.join(JDIR, name)], capture_output=True, timeout=110)
    assert r.returncode == e["rc"], (r.returncode, r.stderr[-300:])
    _check_outputs(out, e)


INTEGRATION_FILES = ["ref_shaun_the_sheep.jpeg", "ref_invader_normal.jpeg", "c1_gray_512x512_q90.jpg", "c4_prog422_160x120.jpg",
                     "prog_420_61x47_q80.jpg", "base_444_17x9_q95_opt.jpg", "base_422_35x70_q30_opt.jpg", "prog_gray_2x2_32x32.jpg",
                     "bad_truncated_scan.jpg", "bad_prog_truncated.jpg", "bad_dri.jpg", "flip_base_03.jpg"]


@pytest.mark.parametrize("name", INTEGRATION_FILES)
def test_integration_binding_compiled_as_c(gpu, tmp_path, name):
    """INTEGRATION.md section 2 as a program (oracle/_ref/jpeg2ppm_integration, built by `make -C oracle integration`):
    the reference's own main, header parser, VLD and bit reader, unmodified, with its two mode drivers replaced by
    integration/{progressive,baseline}_jpeggpu.c -- contiguous pinned planes, ONE jpeggpu_backend_run per scan / image,
    statuses through the erreur_t shim -- writes the files the all-CPU reference wrote, with the same exit status."""
    exe = os.path.join(helpers.ROOT, "oracle", "_ref", "jpeg2ppm_integration")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/jpeg2ppm_integration was not built")
    if name not in EXPECT:
        pytest.skip(f"{name} is not among the fixtures")
    e = EXPECT[name]
    out = tmp_path / "o"
    out.mkdir()
    r = subprocess.run([exe, "-o", str(out / "x.pnm"), os.path.join(JDIR, name)], capture_output=True, timeout=110)
    assert r.returncode == e["rc"], (r.returncode, r.stderr[-300:])
    _check_outputs(out, e)


def test_integration_binding_slow_idct_option(gpu, tmp_path):
    """The same program under the reference's -f option (all_option.idct_fast = 0 -> jpeggpu_set_idct_fast(0))."""
    exe = os.path.join(helpers.ROOT, "oracle", "_ref", "jpeg2ppm_integration")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/jpeg2ppm_integration was not built")
    name = "c4_prog422_160x120.jpg"
    e = dict(EXPECT[name], outputs=EXPECT[name]["outputs_slow"])
    out = tmp_path / "o"
    out.mkdir()
    r = subprocess.run([exe, "-f", "-o", str(out / "x.pnm"), os.path.join(JDIR, name)], capture_output=True, timeout=110)
    assert r.returncode == 0, r.stderr[-300:]
    _check_outputs(out, e)


def test_per_scan_hand_off_uploads_the_dirty_band(gpu):
    """Progressive per-scan hand-off: a DC scan uploads coefficient 0 of every block (2 of each 128 bytes),
    an AC scan the plane of its one component; every scan's image equals the reference's (expect.json)."""
    from jpeg_decoder_amd import frontend as F
    name = "c4_prog422_160x120.jpg"
    j = F.Jpeg(open(os.path.join(JDIR, name), "rb").read())
    d = j.desc
    plane_bytes = [gpu.plane_blocks(d, c) * 128 for c in range(3)]
    imgs, ups, scans = [], [], []
    while not j.eoi:
        j.decode_scan()
        scans.append((j.scan.ss, j.scan.se, j.scan.nb_scan_comp))
        before = j.uploaded_bytes
        imgs.append(F.pnm_header(d) + j.render().tobytes())
        ups.append(j.uploaded_bytes - before)
        if j.eoi:
            break
        j.next_header()
    assert [hashlib.sha256(b).hexdigest() for b in imgs] == expected_images(EXPECT[name])
    for (ss, se, nbc), up in zip(scans, ups):
        if se == 0:      # DC scan over all three components: first one and its refinement
            assert nbc == 3 and up == sum(plane_bytes) // 64, (ss, se, up)
        else:            # AC scan: one component's whole plane
            assert up in plane_bytes, (ss, se, up)
    assert sum(1 for s in scans if s[1] == 0) >= 2      # Pillow's script has a DC refinement scan


def test_plane_pointer_is_refreshed_after_a_dc_refinement_scan(gpu):
    """jpeggpu_jpeg_plane on a GPU host merges the packed coefficient-0 array of the DC scans into the host plane when it is
    CALLED (include/jpeggpu_decode.h): planes fetched after every scan -- before and after the DC refinement scan in
    particular -- rendered by the oracle give that scan's reference image, and coefficient 0 really changes in the
    refinement scan (so a pointer kept from before it would be stale)."""
    from jpeg_decoder_amd import frontend as F
    from oracle import pyoracle
    name = "c4_prog422_160x120.jpg"
    j = F.Jpeg(open(os.path.join(JDIR, name), "rb").read())
    d = j.desc
    od = pyoracle.FrameDesc.from_buffer_copy(bytes(d))
    want = expected_images(EXPECT[name])
    k, dc_changed = 0, []
    prev = None
    while not j.eoi:
        j.decode_scan()
        s = j.scan
        planes = j.planes()                        # fetched again after every scan
        img = F.pnm_header(d) + pyoracle.oracle.backend_run(od, planes, j.qtables()).tobytes()
        assert hashlib.sha256(img).hexdigest() == want[k], (k, s.ss, s.se, s.ah)
        if s.se == 0 and s.ah > 0:                 # the DC refinement scan: one more bit of coefficient 0
            dc_changed.append(any(not np.array_equal(prev[c][:, 0], planes[c][:, 0]) for c in range(3)))
            assert all(np.array_equal(prev[c][:, 1:], planes[c][:, 1:]) for c in range(3))
        prev = planes
        k += 1
        if j.eoi:
            break
        j.next_header()
    assert dc_changed and all(dc_changed)


def test_handles_on_several_threads_share_the_buffer_cache(gpu):
    """Four threads open, decode, render and close files of a few geometries over and over (each handle on its own
    stream; closed handles hand their pinned and device buffers to the next one of the same size, decode_api.cpp).
    Every image of every pass equals the reference's."""
    import threading
    from jpeg_decoder_amd import frontend as F
    names = ["c4_prog422_160x120.jpg", "prog_420_128x96_q60.jpg", "base_444_61x47_q80.jpg", "prog_444_61x47_q80.jpg",
             "base_420_61x47_q80.jpg", "ref_shaun_the_sheep.jpeg", "base_gray_61x47_q80.jpg"]
    blobs = {n: open(os.path.join(JDIR, n), "rb").read() for n in names}
    want = {n: expected_images(EXPECT[n]) for n in names}
    errors = []

    def worker(t):
        try:
            for rep in range(6):
                for n in names[t % 2::1 + t % 2]:       # overlapping subsets, different orders of sizes per thread
                    j = F.Jpeg(blobs[n])
                    d = j.desc
                    imgs = []
                    while not j.eoi:
                        j.decode_scan()
                        imgs.append(F.pnm_header(d) + j.render().tobytes())
                        if j.eoi:
                            break
                        j.next_header()
                    j.close()
                    got = [hashlib.sha256(b).hexdigest() for b in imgs]
                    assert got == want[n], (t, rep, n)
        except BaseException as e:   # noqa: BLE001  (reported by the main thread)
            errors.append((t, repr(e)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors


def test_closed_files_hand_their_buffers_to_the_next_one(gpu):
    """decode_api.cpp keeps a closed file's pinned planes for the next file that needs planes of exactly that size (a fresh
    pinned buffer costs 0.1-0.3 ms per MB at the first CPU touch): the second handle's planes lie where the first one's
    did, start out zeroed all the same, and decode to the reference's image."""
    from jpeg_decoder_amd import frontend as F
    name = "c1_gray_512x512_q90.jpg"
    data = open(os.path.join(JDIR, name), "rb").read()
    addr = []
    for rep in range(3):
        j = F.Jpeg(data)
        addr.append(int(F.lib.jpeggpu_jpeg_plane(j._h, 0) or 0))
        assert not j.plane(0).any()                      # zeroed although the buffer held the previous file's coefficients
        j.decode_scan()
        img = F.pnm_header(j.desc) + j.render().tobytes()
        assert [hashlib.sha256(img).hexdigest()] == expected_images(EXPECT[name])
        j.close()
    assert addr[0] != 0 and addr[1] == addr[0] and addr[2] == addr[0]


def test_render_into_pinned_memory_in_place(gpu):
    """jpeggpu_jpeg_render with a raster from jpeggpu_host_alloc: the download lands there directly (packed and padded
    rows), scan by scan for a progressive file, interleaved with ordinary renders; every image equals the reference's."""
    from jpeg_decoder_amd import frontend as F
    for name in ("c4_prog422_160x120.jpg", "c1_gray_512x512_q90.jpg", "base_420_61x47_q80.jpg"):
        j = F.Jpeg(open(os.path.join(JDIR, name), "rb").read())
        d = j.desc
        rows, rb = gpu.raster_rows(d), gpu.raster_row_bytes(d)
        held = [gpu.PinnedArray((rows, rb), np.uint8), gpu.PinnedArray((rows + 1, rb + 13), np.uint8)]
        try:
            imgs, k = [], 0
            while not j.eoi:
                j.decode_scan()
                if k % 3 == 2:
                    img = j.render()
                else:
                    dst = held[k % 3].array
                    dst[...] = 0xCD
                    img = j.render(out=dst)
                    if k % 3 == 1:
                        assert np.all(dst[:, rb:] == 0xCD) and np.all(dst[rows:] == 0xCD)
                imgs.append(F.pnm_header(d) + np.ascontiguousarray(img).tobytes())
                k += 1
                if j.eoi:
                    break
                j.next_header()
            assert [hashlib.sha256(b).hexdigest() for b in imgs] == expected_images(EXPECT[name]), name
        finally:
            j.close()
            [h.free() for h in held]


@pytest.mark.parametrize("step", [1, 3000, 40000], ids=["every_mcu_row", "3kB_steps", "40kB_steps"])
def test_jpeg2ppm_streams_baseline_files_in_small_steps(gpu, tmp_path, step):
    """jpeggpu_jpeg2ppm hands a baseline file's finished MCU rows to the GPU and to the output file while the rest is still
    being decoded (decode_api.cpp BaselineStream).  With steps of one MCU row / a few rows every baseline fixture -- well-formed,
    damaged in the middle of the scan, truncated -- gives the reference's status and files, byte for byte."""
    from jpeg_decoder_amd import frontend as F
    old = os.environ.get("JPEGGPU_STRIPE_BYTES")
    os.environ["JPEGGPU_STRIPE_BYTES"] = str(step)
    gpu.reload_tuning()
    try:
        n = 0
        for name in sorted(EXPECT):
            e = EXPECT[name]
            if name.startswith(("prog_", "c4_prog", "flip_prog", "flipmid_prog", "bad_prog")):
                continue
            out = tmp_path / name.replace(".", "_")
            out.mkdir()
            rc = 0
            try:
                F.jpeg2ppm(os.path.join(JDIR, name), out / "x.pnm")
            except gpu.JpegGpuError as err:
                rc = err.code
            assert rc == e["rc"], (name, rc, e["err"], gpu.last_error())
            _check_outputs(out, e)
            n += 1
        assert n > 60
    finally:
        if old is None:
            os.environ.pop("JPEGGPU_STRIPE_BYTES", None)
        else:
            os.environ["JPEGGPU_STRIPE_BYTES"] = old
        gpu.reload_tuning()


def test_jpeg2ppm_from_several_threads(gpu, tmp_path):
    """jpeggpu_jpeg2ppm called by four threads at once (every call has its own handle, stream and writer threads; buffers
    come from and go back to the shared cache): each call's status and files are the reference's."""
    import threading
    from jpeg_decoder_amd import frontend as F
    names = ["c4_prog422_160x120.jpg", "c1_gray_512x512_q90.jpg", "prog_420_128x96_q60.jpg", "ref_shaun_the_sheep.jpeg",
             "bad_truncated_scan.jpg", "flipmid_prog_00.jpg"]
    errors = []

    def worker(t):
        try:
            for rep in range(3):
                for i, name in enumerate(names[t % 3:] + names[:t % 3]):
                    e = EXPECT[name]
                    out = tmp_path / f"t{t}_{rep}_{i}"
                    out.mkdir()
                    rc = 0
                    try:
                        F.jpeg2ppm(os.path.join(JDIR, name), out / "x.pnm")
                    except gpu.JpegGpuError as err:
                        rc = err.code
                    assert rc == e["rc"], (name, rc, e["err"])
                    _check_outputs(out, e)
        except BaseException as e:   # noqa: BLE001  (reported by the main thread)
            errors.append((t, repr(e)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors


@pytest.mark.parametrize("name", ["c1_gray_512x512_q90.jpg", "c4_prog422_160x120.jpg"])
def test_jpeg2ppm_reports_a_file_that_cannot_be_written(gpu, tmp_path, name):
    """The writer threads of jpeggpu_jpeg2ppm (a band writer for baseline files, image writers for progressive ones) report
    a file that cannot be created as JPEGGPU_ERR_IO through the call's status -- no hang, no image counted -- and the
    library goes on working afterwards."""
    from jpeg_decoder_amd import frontend as F
    ro = tmp_path / "ro"
    ro.mkdir()
    os.chmod(ro, 0o555)
    try:
        if os.access(ro, os.W_OK):
            pytest.skip("running with privileges that ignore directory permissions")
        with pytest.raises(gpu.JpegGpuError) as e:
            F.jpeg2ppm(os.path.join(JDIR, name), ro / "x.pnm")
        assert e.value.code == F.ERR_IO, gpu.last_error()
        assert os.listdir(ro) == []
    finally:
        os.chmod(ro, 0o755)
    out = tmp_path / "ok"
    out.mkdir()
    F.jpeg2ppm(os.path.join(JDIR, name), out / "x.pnm")
    _check_outputs(out, EXPECT[name])


@pytest.mark.timeout(650)
def test_host_code_under_address_sanitizer(gpu):
    """tools/jpeg2ppm_gpu_asan (built by `make -C jpeg-decoder_amd/csrc asan`: the command-line driver and every library source
    in one program, host code instrumented with AddressSanitizer) decodes every fixture on the GPU, in default steps and
    in steps of one MCU row -- pools, writer threads, streamed baseline files, damaged inputs -- and tools/host_pipe_asan
    (tools/host_pipe_check.cpp, same build) drives the host-pointer entries with their helper threads: no finding, no hang."""
    exe = os.path.join(helpers.ROOT, "tools", "jpeg2ppm_gpu_asan")
    if not os.path.exists(exe):
        pytest.skip("tools/jpeg2ppm_gpu_asan has not been built")
    r = subprocess.run(["bash", os.path.join(helpers.ROOT, "tools", "asan_cli.sh")], cwd=helpers.ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " 0 with findings" in r.stdout, (r.stdout[-1500:], r.stderr[-500:])


@pytest.mark.timeout(300)
@pytest.mark.parametrize("kind", ["c2_baseline_444", "c4_progressive_422", "c3_baseline_420_8192"])
def test_full_size_files_against_the_reference_binary(gpu, tmp_path, kind):
    """BASELINE.json configs[1], configs[3] (4096x4096) and configs[2] (8192x8192 4:2:0) as real files (encoded on the spot with Pillow): the files
    jpeggpu_jpeg2ppm writes -- one image for the baseline file, one per scan plus the symbolic link for the progressive
    one -- are byte-identical to those of the reference compiled from its own sources (oracle/_ref/jpeg2ppm_sans_opt,
    strict IEEE build), run here on the same file."""
    Image = pytest.importorskip("PIL.Image")
    from jpeg_decoder_amd import frontend as F
    ref = os.path.join(helpers.ROOT, "oracle", "_ref", "jpeg2ppm_sans_opt")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/jpeg2ppm_sans_opt has not been built")
    w = h = 8192 if kind.startswith("c3") else 4096
    rng = np.random.default_rng(41)
    y, x = np.mgrid[0:h, 0:w]
    a = np.stack([(x + y) % 256, (x * 3 + y // 2) % 256, (x // 3 + y * 2) % 256], -1).astype(np.float32)
    a += 50 * np.sin(x / 23.0)[..., None] * np.cos(y / 17.0)[..., None] + rng.normal(0, 10, a.shape).astype(np.float32)
    b = io.BytesIO()
    kw = {"c2": dict(quality=90, subsampling=0), "c4": dict(quality=90, subsampling=1, progressive=True), "c3": dict(quality=85, subsampling=2)}[kind[:2]]
    Image.fromarray(np.clip(a, 0, 255).astype(np.uint8)).save(b, "JPEG", **kw)
    src = tmp_path / "in.jpg"
    src.write_bytes(b.getvalue())
    dirs = {}
    for who in ("ref", "gpu"):
        dirs[who] = tmp_path / who
        dirs[who].mkdir()
    r = subprocess.run([ref, "-o", str(dirs["ref"] / "x.ppm"), str(src)], capture_output=True, timeout=240)
    assert r.returncode == 0, r.stdout[-300:]
    F.jpeg2ppm(src, dirs["gpu"] / "x.ppm")

    def listing(d):
        out = {}
        for f in sorted(os.listdir(d)):
            p = os.path.join(d, f)
            out[f] = ("link", os.readlink(p)) if os.path.islink(p) else ("file", os.path.getsize(p), hashlib.sha256(open(p, "rb").read()).hexdigest())
        return out
    got, want = listing(dirs["gpu"]), listing(dirs["ref"])
    assert len(want) == (11 if kind.startswith("c4") else 1)
    assert got == want

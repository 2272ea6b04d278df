"""MI355X-native JPEG pixel back end (iquant+izz -> idct_opt -> upsample+ycc2rgb).

Thin ctypes binding over the C ABI in include/jpeggpu.h (libjpeggpu.so).  The directory is named
``jpeg-decoder_amd``; import it through the ``jpeg_decoder_amd`` symlink at the repository root.
"""
from .backend import (  # noqa: F401
    JpegGpuError, FrameDesc, Component, make_desc, lib, lib_path, last_error,
    init, shutdown, synchronize, plane_blocks, raster_rows, raster_row_bytes,
    backend_run, backend_run_batch, backend_run_device, backend_run_batch_device,
    iquant_blocks, izz_blocks, idct_opt_blocks, ycc2rgb_pixels,
    idct_blocks, idct_coef_table, set_idct_fast, get_idct_fast,
    Context, multi_backend_run, multi_backend_run_batch, reload_tuning, PinnedArray,
    SUCCESS, ERR_NB_COMP, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_SAMPLING, ERR_NO_QT,
)

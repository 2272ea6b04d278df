// kernels.h -- launch interface between the C-ABI layer (capi.cpp) and the HIP kernels.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stddef.h>

namespace jpeggpu_k {

#ifndef JM_MAX_THREADS
#define JM_MAX_THREADS 384
#endif
constexpr uint32_t kMaxThreads = JM_MAX_THREADS;   // __launch_bounds__ of the fused kernel = most blocks per tile

// One launch of the fused back end: a batch of same-geometry frames, cut into tiles of
// T MCUs x 1 MCU row.  Filled by plan_backend() in capi.cpp.
struct BackendParams {
   const void *coef[3];            // device plane pointers of frame 0 (blocl16_t arrays)
   void *out;                      // device raster of frame 0
   void *diag;                     // JM_DIAG_STAMP builds only: 5 x uint64 phase-time accumulators (else NULL)
   const float *slow_coef;         // "-f" mode: calc_coef's table [x][y][lambda][mu] (src/idct.c:35-48); NULL = idct_opt
   uint64_t coef_frame_stride[3];  // bytes between consecutive frames' planes
   uint64_t out_stride;            // bytes between raster rows
   uint64_t out_frame_stride;      // bytes between consecutive frames' rasters
   uint32_t width;                 // pixels per raster row (crop)
   uint32_t nbcomp;                // 1 or 3
   uint32_t apply_iquant;
   uint32_t T;                     // MCUs per tile
   uint32_t B;                     // blocks per tile = T * sum(hs*vs)
   uint32_t tile_w;                // pixels per tile row  = T * max_h * 8
   uint32_t tile_rows;             // raster rows per tile = max_v * 8
   uint32_t wave_uniform;          // 1: every boff[] is a multiple of 64 -> each wave holds blocks of one component
   uint32_t wq_shift;              // log2(tile_w / 4) when tile_w / 4 is a power of two, else 0xFFFFFFFF (generic kernel only)
   uint32_t nbH[3];                // blocks per plane row (nbmcuH*hs)
   uint32_t vs[3];                 // block rows per MCU row
   uint32_t bw[3];                 // blocks per tile row of component c = T*hs
   uint32_t boff[4];               // first tile-block index of component c; boff[nbcomp] == B
   uint32_t hf[3], vf[3];          // upsampling factors max_h/hs, max_v/vs (src/decoder_utils.c:102-107)
   uint32_t plane_off[3];          // LDS byte offset of component c's u8 plane
   uint32_t idq[3];
   uint32_t hs_shift[3];           // log2(hs) when hs is a power of two, else 0 (linear tiling needs powers of two)
   uint32_t mcu_q_shift;           // log2(quads of 4 pixels per MCU width) = log2(max_h * 2) when that is a power of two
   uint32_t linear;                // 1: a tile is T consecutive MCUs of the FRAME in raster order (grid.x = tiles per frame, grid.y = 1)
   uint32_t nbmcuH;                // MCUs per MCU row
   uint32_t total_mcus;            // nbmcuH * nbmcuV
   uint32_t inv_nbmcuH;            // floor(2^32 / max(nbmcuH, 2)): linear tiling divides by nbmcuH on the scalar unit (kernels.hip, mcu_row_of)
   uint32_t restage;               // kernels with restaged 16-byte raster stores (16-byte aligned raster): 1 = backend_kernel_rs444 (4:4:4, whole
                                   // 64-MCU tiles), 3 = backend_kernel_rs444_lin (4:4:4, linear tiling, even MCUs per row), 2 = backend_kernel_rs420 / _rs420_lin (4:2:0, uncropped, 32-MCU tiles, linear when rows are not whole tiles); 0 = general kernel
   uint32_t qt[4][32];             // quantisation tables, zigzag order, two uint16 per word
};

struct LaunchDims { uint32_t tiles_per_row, mcu_rows, frames, threads; size_t lds_bytes; };

hipError_t launch_backend(const BackendParams &p, const LaunchDims &d, hipStream_t s);
bool fast_colour_layout(const BackendParams &p);   // true: a sub-sampling-specific pixel phase handles this layout

// Device staging of jpeggpu_backend_run_noalloc: a file-scope __device__ array in kernels.hip (one copy per GPU).
// static_scratch() returns ITS DEVICE ADDRESS on <device> (hipGetSymbolAddress, once per device, cached).
constexpr size_t kStaticScratchBytes = (size_t)32 << 20;
constexpr int kMaxDevices = 64;
hipError_t static_scratch(int device, void **ptr);

// stage-level kernels (device pointers)
hipError_t launch_iquant(void *d_blocks, size_t n, int s_start, int s_end, const uint32_t qt_packed[32], hipStream_t s);
hipError_t launch_izz(const void *d_in, void *d_out, size_t n, hipStream_t s);
hipError_t launch_scatter_dc(void *d_blocks, const void *d_pack, size_t n, hipStream_t s);   // blocks[b].data[0] = pack[b]
hipError_t launch_idct(const void *d_in, void *d_out, size_t n, hipStream_t s);
hipError_t launch_idct_slow(const void *d_in, void *d_out, size_t n, const float *d_coef, hipStream_t s);
hipError_t launch_ycc2rgb(const uint8_t *d_y, const uint8_t *d_cb, const uint8_t *d_cr, uint8_t *d_rgb, size_t n, hipStream_t s);

}  // namespace jpeggpu_k

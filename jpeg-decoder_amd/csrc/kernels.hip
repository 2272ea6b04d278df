// kernels.hip -- HIP kernels of the JPEG pixel back end for gfx950 (MI355X).
//
// The fused path  iquant + izz + idct_opt + upsample + ycc2rgb + raster store.
//   One workgroup = one tile of T MCUs; one thread = one 8x8 block.
//   phase 1  each lane loads its own 128-byte blocl16_t record straight into registers (8 x 16 bytes),
//            dequantises (table words as scalar operands when the wave holds one component), and runs both
//            1-D passes of the Loeffler IDCT in registers (the inverse zigzag and the transpose between the
//            passes are register renamings, not data movement), then writes its 8 rows of 8 samples into
//            the component's u8 raster plane in LDS
//   phase 2  after the one barrier: replicate-upsampled chroma fetch (px = x/hf, py = y/vf,
//            src/decoder_utils.c:72-80), packed 16-bit integer colour conversion, non-temporal raster stores
//            (width cropped, rows not).
// Three families:
//   backend_kernel_rs444   4:4:4 in whole 64-MCU tiles: row-pair units, RGB restaged inside the wave, 16-byte stores
//   backend_kernel_rs444_lin   the same for other uncropped widths with an even number of MCUs per row (linear tiling)
//   backend_kernel_rs420 / _rs420_lin   4:2:0 in 32-MCU tiles of 192 lanes (Cb and Cr share a wave), same stores; whole tiles
//                          per MCU row / linear tiling
//   backend_kernel<H,V,T>  everything else (4:2:2, 4:4:0, gray, crops, unaligned rasters, generic factors): runs of 64
//                          quads per wave with 12-byte stores; backend_kernel_slow for "-f"
//
// Compiled with -ffp-contract=off (see jpeg_math.hpp).
#include <hip/hip_runtime.h>
#include <mutex>
#include "jpeg_math.hpp"
#include "kernels.h"

#pragma clang fp contract(off)

#ifndef JM_NT_STORE
#define JM_NT_STORE 1      // raster bytes are written once and never re-read by the kernel: +2% measured
#endif
#ifndef JM_WAVES_PER_EU
#define JM_WAVES_PER_EU 6
#endif
#ifndef JM_RS_NT
#define JM_RS_NT 1         // the 16-byte raster stores of the restage kernels carry the non-temporal hint as well
#endif
#ifndef JM_RS_WAVES_PER_EU
#define JM_RS_WAVES_PER_EU 7   // the 4:4:4 restage kernel fits 72 VGPRs without spills: 7 waves per SIMD = 9 workgroups per CU
#endif

// Raster rows start wherever out_stride puts them (a packed RGB row of odd width is not 4-byte aligned).  gfx950
// under HSA runs with unaligned dword access enabled, so the 12-byte store is issued at any address.
#if defined(JM_STORE_ALIGNED_ONLY)
#define JM_ALIGN_TEST(dst) && ((uintptr_t)(dst) & 3) == 0
#define JM_STORE_ALIGN 4
#else
#define JM_ALIGN_TEST(dst)
#define JM_STORE_ALIGN 1
#endif

namespace jpeggpu_k {

using namespace jpegmath;

typedef uint32_t rs_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void rs_store16(uint8_t *dst, rs_u32x4 v) {
#if JM_RS_NT
   __builtin_nontemporal_store(v, reinterpret_cast<rs_u32x4 *>(dst));
#else
   *reinterpret_cast<rs_u32x4 *>(dst) = v;
#endif
}
// The same store as scalar base + 32-bit lane offset (the compiler adds the two in 64 bits on the vector unit instead: three
// v_lshl_add_u64 and three zeroed registers per unit).  base must be wave-uniform.  The s_nop covers the wait states a
// following overwrite of the data registers needs (the compiler's hazard recogniser does not look into asm).
__device__ __forceinline__ void rs_store16_s(const uint8_t *base, uint32_t off, rs_u32x4 v) {
#if JM_RS_NT && !defined(JM_NO_SADDR_STORE)
   asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" : : "v"(off), "v"(v), "s"(base) : "memory");
#else
   rs_store16(const_cast<uint8_t *>(base) + off, v);
#endif
}

struct __attribute__((packed, aligned(4))) rgb4_t { uint32_t a, b, c; };

// Per-thread view of the component a tile-block belongs to.  All fields are selected from scalar
// kernel arguments with v_cndmask (no per-lane memory loads).
struct BlockCtx {
   uint32_t c, by, bxl, bw, nbH, vs, idq, plane_off;
   const uint8_t *coef; uint64_t frame_stride;
};
template <typename T> __device__ __forceinline__ T sel3(uint32_t c, T a0, T a1, T a2) { return c == 0 ? a0 : (c == 1 ? a1 : a2); }

__device__ __forceinline__ BlockCtx decode_block(const BackendParams &p, uint32_t b) {
   BlockCtx k;
   k.c = (p.nbcomp == 3) ? (uint32_t)(b >= p.boff[1]) + (uint32_t)(b >= p.boff[2]) : 0u;
   k.bw = sel3(k.c, p.bw[0], p.bw[1], p.bw[2]);
   k.nbH = sel3(k.c, p.nbH[0], p.nbH[1], p.nbH[2]);
   k.vs = sel3(k.c, p.vs[0], p.vs[1], p.vs[2]);
   k.idq = sel3(k.c, p.idq[0], p.idq[1], p.idq[2]);
   k.plane_off = sel3(k.c, p.plane_off[0], p.plane_off[1], p.plane_off[2]);
   k.coef = static_cast<const uint8_t *>(sel3(k.c, p.coef[0], p.coef[1], p.coef[2]));
   k.frame_stride = sel3(k.c, p.coef_frame_stride[0], p.coef_frame_stride[1], p.coef_frame_stride[2]);
   const uint32_t r = b - sel3(k.c, p.boff[0], p.boff[1], p.boff[2]);
   k.by = (uint32_t)(r >= k.bw) + (uint32_t)(r >= 2 * k.bw) + (uint32_t)(r >= 3 * k.bw);
   k.bxl = r - k.by * k.bw;
   return k;
}

// store up to 4 pixels (12 bytes) whose first byte is dst; n = number of valid pixels (1..4)
__device__ __forceinline__ void store_rgb_quad(uint8_t *dst, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t n) {
#if defined(JM_DIAG_NOSTORE)
   if ((w0 ^ w1 ^ w2) != 0x12345678u) return;
#endif
   if (n == 4 JM_ALIGN_TEST(dst)) {
#if JM_NT_STORE
      typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
      __builtin_nontemporal_store(u32x3{w0, w1, w2}, reinterpret_cast<u32x3 __attribute__((aligned(JM_STORE_ALIGN))) *>(dst));
#else
      rgb4_t v = { w0, w1, w2 };
      *reinterpret_cast<rgb4_t *>(dst) = v;
#endif
   } else {
      uint32_t w[3] = { w0, w1, w2 };
#pragma unroll
      for (uint32_t i = 0; i < 12; i++)
         if (i < 3 * n) dst[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
   }
}

// Colour conversion of one item: 4 pixels wide x VFC rows.  cbw / crw hold the item's chroma samples (HFC = 1: four
// bytes, one per pixel; HFC = 2: two bytes, each under two pixels, upper half made unlike the special pair by the
// caller), yy[v] the four luma bytes of row v.  w[v] = the 12 output bytes of row v.
template <int HFC, int VFC>
__device__ __forceinline__ void convert_item(const uint32_t (&yy)[VFC], uint32_t cbw, uint32_t crw, uint32_t (&w)[VFC][3]) {
#if defined(JM_DIAG_NOCOLOR)
   if (true) {
#pragma unroll
      for (int v = 0; v < VFC; v++) { w[v][0] = yy[v]; w[v][1] = cbw; w[v][2] = crw; }
   } else
#endif
   if (__builtin_expect(__any(has_special_pair(cbw, crw)), 0)) {
      // exact per-pixel form (the y-dependent green offset of (cb,cr) = (28,228))
#pragma unroll
      for (int v = 0; v < VFC; v++) {
         uint32_t px[4];
#pragma unroll
         for (int i = 0; i < 4; i++) {
            const int ci = (HFC == 1) ? i : (i >> 1);
            px[i] = ycc_to_rgb((yy[v] >> (8 * i)) & 0xFF, chroma_offsets((cbw >> (8 * ci)) & 0xFF, (crw >> (8 * ci)) & 0xFF));
         }
         w[v][0] = px[0] | (px[1] << 24); w[v][1] = (px[1] >> 8) | (px[2] << 16); w[v][2] = (px[2] >> 16) | (px[3] << 8);
      }
   } else {
      const ChromaPairs o = chroma_pairs<HFC>(cbw ^ 0x80808080u, crw ^ 0x80808080u);   // bytes become cb-128, cr-128 (int8)
#pragma unroll
      for (int v = 0; v < VFC; v++) rgb_quad_pk(yy[v], o, w[v]);
   }
}

// Where a tile sits in its frame.  A tile is T consecutive MCUs: of one MCU row (column c0 .. c0+T-1 of row m0), or --
// "linear" tiling, for frames whose MCU rows are not a whole number of tiles -- T consecutive MCUs in raster order,
// running on into the next MCU row after `wrapj` of them.  nvalid < T only for the last tile of a linear frame.
struct TileGeom { uint32_t m0, c0, wrapj, nvalid; };

// Phase 2 for the sub-sampling-specific layouts when a tile row is at least 64 quads wide: every WAVE takes whole
// runs of 64 consecutive quads (256 pixels of one raster row x VFC rows), so the row, the LDS offsets and the raster
// address of a run are wave-uniform (SALU) and a lane adds only its own 4 / 12 byte offsets.  A run that lies inside
// one MCU row and inside the frame stores 768 contiguous bytes per row from a scalar base; a run that straddles the
// end of an MCU row (linear tiling), the right edge of the frame or the end of the frame takes the per-lane form.
template <int HFC, int VFC>
__device__ __forceinline__ void pixel_phase_runs(const BackendParams &p, const uint8_t *lds, uint32_t tid, uint32_t nt, const TileGeom &g,
                                                 uint32_t frame) {
   const uint32_t ws = p.wq_shift;                           // log2(quads per tile row), >= 6 here
   const uint32_t rs = ws - 6;                               // log2(runs per tile row)
   const uint32_t nruns = (p.tile_rows / VFC) << rs;
   const uint32_t nwaves = nt >> 6;
   const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const uint32_t l = tid & 63;
   const uint32_t qs = p.mcu_q_shift;                        // log2(quads per MCU): 1, 2 or 3
   const uint32_t P = 64u >> qs;                             // MCUs per run
   const uint32_t mcu_w = 4u << qs;                          // pixels per MCU
   const uint32_t stride = (uint32_t)p.out_stride;           // <= 2^26 (run_device)
   const uint32_t band = p.tile_rows * stride;               // bytes from one MCU row's band to the next (<= 2^31)
   uint8_t *base0 = static_cast<uint8_t *>(p.out) + (size_t)frame * p.out_frame_stride + (size_t)g.m0 * band;
   const uint32_t pitch0 = p.tile_w;
   const uint32_t lc = l * (4 / HFC), ly = l * 4, l12 = l * 12;
   const uint32_t first_row = min(g.wrapj, g.nvalid);        // tile MCUs [0, first_row) lie in MCU row m0
   for (uint32_t run = wave; run < nruns; run += nwaves) {
      const uint32_t yi = run >> rs, seg = run & ((1u << rs) - 1);
      const uint32_t y = yi * VFC;
      const uint32_t ja = seg * P;                           // first tile MCU of the run
      if (ja >= g.nvalid) continue;
      // where the run's 256 pixels go, if that is the same raster row for all of them
      uint32_t xs = 0, rowoff = 0;
      bool whole = false;
      if (ja + P <= first_row) { xs = (g.c0 + ja) * mcu_w; whole = xs + 256 <= p.width; if (xs >= p.width) continue; }
      else if (ja >= g.wrapj && ja + P <= g.nvalid) { xs = (ja - g.wrapj) * mcu_w; rowoff = band; whole = xs + 256 <= p.width; if (xs >= p.width) continue; }
      const uint32_t it0 = (yi << ws) + (seg << 6);          // item index of lane 0 = word offset of its chroma samples
      uint32_t cbw, crw;
      if (HFC == 1) {
         cbw = *reinterpret_cast<const uint32_t *>(lds + (p.plane_off[1] + it0 * 4 + lc));
         crw = *reinterpret_cast<const uint32_t *>(lds + (p.plane_off[2] + it0 * 4 + lc));
      } else {
         cbw = *reinterpret_cast<const uint16_t *>(lds + (p.plane_off[1] + it0 * 2 + lc));
         crw = *reinterpret_cast<const uint16_t *>(lds + (p.plane_off[2] + it0 * 2 + lc));
         cbw |= 0x01010000u;   // unused byte lanes can never look like the special pair
      }
      const uint32_t y0 = p.plane_off[0] + (((y << ws) + (seg << 6)) << 2);
      uint32_t yy[VFC];
#pragma unroll
      for (int v = 0; v < VFC; v++) yy[v] = *reinterpret_cast<const uint32_t *>(lds + (y0 + v * pitch0 + ly));
      uint32_t w[VFC][3];
      convert_item<HFC, VFC>(yy, cbw, crw, w);
      if (whole) {
         uint8_t *rowbase = base0 + ((size_t)rowoff + (size_t)y * stride + (size_t)xs * 3);   // wave-uniform
         typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
#pragma unroll
         for (int v = 0; v < VFC; v++) {
            __builtin_nontemporal_store(u32x3{w[v][0], w[v][1], w[v][2]},
                                        reinterpret_cast<u32x3 __attribute__((aligned(JM_STORE_ALIGN))) *>(rowbase + (size_t)v * stride + l12));
         }
      } else {
         const uint32_t j = ja + (l >> qs);
         const bool r2 = j >= g.wrapj;
         const uint32_t col = r2 ? j - g.wrapj : g.c0 + j;
         const uint32_t x = col * mcu_w + (l & ((1u << qs) - 1)) * 4;
         if (j < g.nvalid && x < p.width) {
            const uint32_t n = min(4u, p.width - x);
            uint8_t *dst = base0 + ((r2 ? band : 0u) + y * stride + x * 3);
#pragma unroll
            for (int v = 0; v < VFC; v++) store_rgb_quad(dst + v * stride, w[v][0], w[v][1], w[v][2], n);
         }
      }
   }
}

// ---- 4:4:4 with 16-byte raster stores (RS = 1) ------------------------------------------------------------------
// What a store INSTRUCTION carries decides the write bandwidth (profiles/r02_ubench_store.txt): 64 lanes x 12 bytes
// of RGB reach 71.9 % of the HBM peak in c2's shape, 64 x 16 bytes 74 %.  An RGB quad is 12 bytes, so the 16-byte
// pieces are cut after the conversion: the tile's planes lie in LDS interleaved by ROW PAIR -- unit u = [Y rows 2u,
// 2u+1 | Cb rows | Cr rows], 3 x 1024 = 3072 contiguous bytes, exactly the size of the unit's RGB output (2 rows x
// 512 pixels x 3) -- and a whole unit belongs to ONE wave: every lane reads 16 pixels of each plane (3 x
// ds_read_b128), converts its four quads, writes its 48 bytes back OVER the unit at 48 * lane (the unit's samples
// are in registers by then, and no other wave ever touches it), reads the pieces [1024 s + 16 lane, + 16), s = 0..2,
// and stores them: 3 x global_store_dwordx4 of 1024 bytes (the middle one ends row 2u and starts row 2u+1).  No
// workgroup barrier is involved: LDS serves a wave's instructions in order.  4 units on 3 waves: wave 0 takes two.
// Host side (plan_backend): uncropped width; 64-MCU tiles that all lie inside the frame (width a multiple of 512), or
// (LIN, linear tiling) an even number of MCUs per row above 64, where a lane's pieces may fall either side of the
// MCU-row wrap or past the frame's last MCU; raster base and strides multiples of 16.
constexpr uint32_t kRsRowBytes = 512, kRsPlaneBytes = 1024, kRsUnitBytes = 3072;
__device__ __forceinline__ uint32_t special_pair_bits(uint32_t cbw, uint32_t crw) {   // has_special_pair before the compare
   const uint32_t t = (cbw ^ 0x1C1C1C1Cu) | (crw ^ 0xE4E4E4E4u);
   return (t - 0x01010101u) & ~t;
}
template <bool LIN>
__device__ __forceinline__ void pixel_phase_units444(const BackendParams &p, uint8_t *lds, uint32_t wave, uint32_t l, const TileGeom &g, uint32_t frame) {
   typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
   const uint32_t stride = (uint32_t)p.out_stride;
   const uint32_t band = 8 * stride;
   uint8_t *tile_out = static_cast<uint8_t *>(p.out) + (size_t)frame * p.out_frame_stride + (size_t)g.m0 * band;
   // where a lane's three 16-byte pieces go, relative to the first byte of the unit's first row in MCU row m0: byte b of
   // the tile's 1536-byte row belongs to MCU b / 24; before the wrap (linear tiling) that MCU lies c0 MCUs into MCU row m0,
   // after it at the start of the next MCU row.  wrapj and nvalid are even (host check), so no piece straddles either.
   const uint32_t b0 = l * 16, b1 = l < 32 ? 1024 + l * 16 : (l - 32) * 16, b2 = 512 + l * 16;
   const uint32_t before = g.c0 * 24, after = band - g.wrapj * 24;      // (after is only used under LIN)
   uint32_t o0 = b0 + before, o1 = b1 + before + (l < 32 ? 0u : stride), o2 = b2 + before + stride;
   bool ok0 = true, ok1 = true, ok2 = true;
   if constexpr (LIN) {
      const uint32_t j0 = b0 / 24, j1 = b1 / 24, j2 = b2 / 24;
      if (j0 >= g.wrapj) o0 += after - before;
      if (j1 >= g.wrapj) o1 += after - before;
      if (j2 >= g.wrapj) o2 += after - before;
      ok0 = j0 < g.nvalid; ok1 = j1 < g.nvalid; ok2 = j2 < g.nvalid;
   }
   for (uint32_t u = wave; u < 4; u += 3) {
      JM_MARK("unit_lds_reads");
      uint8_t *ul = lds + u * kRsUnitBytes;
      const uint4 yv = *reinterpret_cast<const uint4 *>(ul + l * 16);
      const uint4 cb = *reinterpret_cast<const uint4 *>(ul + kRsPlaneBytes + l * 16);
      const uint4 cr = *reinterpret_cast<const uint4 *>(ul + 2 * kRsPlaneBytes + l * 16);
      const uint32_t yy[4] = { yv.x, yv.y, yv.z, yv.w }, cbw[4] = { cb.x, cb.y, cb.z, cb.w }, crw[4] = { cr.x, cr.y, cr.z, cr.w };
      uint32_t w[4][3];
#if defined(JM_DIAG_NOCOLOR)
#pragma unroll
      for (int k = 0; k < 4; k++) { w[k][0] = yy[k]; w[k][1] = cbw[k]; w[k][2] = crw[k]; }
#else
      JM_MARK("unit_special_pair_test");
      const uint32_t sp = (special_pair_bits(cbw[0], crw[0]) | special_pair_bits(cbw[1], crw[1]) | special_pair_bits(cbw[2], crw[2]) |
                           special_pair_bits(cbw[3], crw[3])) & 0x80808080u;
      if (__builtin_expect(__any(sp != 0), 0)) {
         JM_MARK("unit_exact_colour_cold");
         // exact per-pixel form (the y-dependent green offset of (cb,cr) = (28,228))
#pragma unroll
         for (int k = 0; k < 4; k++) {
            uint32_t px[4];
#pragma unroll
            for (int i = 0; i < 4; i++)
               px[i] = ycc_to_rgb((yy[k] >> (8 * i)) & 0xFF, chroma_offsets((cbw[k] >> (8 * i)) & 0xFF, (crw[k] >> (8 * i)) & 0xFF));
            w[k][0] = px[0] | (px[1] << 24); w[k][1] = (px[1] >> 8) | (px[2] << 16); w[k][2] = (px[2] >> 16) | (px[3] << 8);
         }
      } else {
         JM_MARK("unit_colour");
#pragma unroll
         for (int k = 0; k < 4; k++) rgb_quad_pk(yy[k], chroma_pairs<1>(cbw[k] ^ 0x80808080u, crw[k] ^ 0x80808080u), w[k]);
      }
#endif
      JM_MARK("unit_restage");
      // every lane's samples are in registers (the values above depend on all three reads) before the first write is issued
      *reinterpret_cast<u32x4 *>(ul + l * 48) = u32x4{w[0][0], w[0][1], w[0][2], w[1][0]};
      *reinterpret_cast<u32x4 *>(ul + l * 48 + 16) = u32x4{w[1][1], w[1][2], w[2][0], w[2][1]};
      *reinterpret_cast<u32x4 *>(ul + l * 48 + 32) = u32x4{w[2][2], w[3][0], w[3][1], w[3][2]};
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const u32x4 s0 = *reinterpret_cast<const u32x4 *>(ul + l * 16);
      const u32x4 s1 = *reinterpret_cast<const u32x4 *>(ul + 1024 + l * 16);
      const u32x4 s2 = *reinterpret_cast<const u32x4 *>(ul + 2048 + l * 16);
      JM_MARK("unit_stores");
      uint8_t *ub = tile_out + (size_t)(2 * u) * stride;      // wave-uniform
      if (ok0) rs_store16_s(ub, o0, s0);
      if (ok1) rs_store16_s(ub, o1, s1);
      if (ok2) rs_store16_s(ub, o2, s2);
      JM_MARK("unit_loop");
   }
}

// Phase 2 of one tile: upsample + colour convert + store, from the u8 planes in LDS.
template <int HFC, int VFC>
__device__ __forceinline__ void pixel_phase(const BackendParams &p, const uint8_t *lds, uint32_t tid, uint32_t nt, uint32_t tile,
                                            uint32_t m, uint32_t frame) {
   // ---- phase 2: upsample + colour convert + store ------------------------------------------
   const uint32_t x0 = tile * p.tile_w;
   if (x0 >= p.width) return;
   const uint32_t wq = p.tile_w >> 2;                       // quads per tile row
   const uint32_t nitems = wq * p.tile_rows;
   // item -> (row, quad) without a division per item: one division up front, then a running pair
   uint32_t iy = tid / wq, ix = tid - iy * wq;
   const uint32_t dy = nt / wq, dx = nt - dy * wq;
   uint8_t *out = static_cast<uint8_t *>(p.out) + (size_t)frame * p.out_frame_stride +
                  (size_t)m * p.tile_rows * p.out_stride;
   const uint32_t pitch0 = p.bw[0] * 8;
   if (p.nbcomp == 1) {
      // save_mcu_ligne_bw (src/decoder_utils.c:48-63): straight copy of the luma plane.  The reference reads block
      // x/8 of the MCU row's FIRST block row with py = y % 8 (:55-58): when the single component has vsampling > 1
      // the rows 8.. of the band repeat rows 0..7 -- hence the "& 7" on the source row (a no-op for 1x1 sampling)
#if !defined(JM_GRAY_QUADS)
      if (p.wq_shift != 0xFFFFFFFFu && wq >= 4) {
         // power-of-two tile: 16 pixels per item, the item index is the plane's 16-byte word index
         const uint32_t ws = p.wq_shift - 2;
         const uint32_t nit = p.tile_rows << ws;
         const uint32_t rem = p.width - x0;
         uint8_t *tile_out = out + x0;
         const uint32_t stride = (uint32_t)p.out_stride;
         for (uint32_t it = tid; it < nit; it += nt) {
            const uint32_t yi = it >> ws, x = (it & ((wq >> 2) - 1)) * 16;
            if (x >= rem) continue;
            const uint32_t left = rem - x;
            const uint4 v = *reinterpret_cast<const uint4 *>(lds + p.plane_off[0] + (((yi & 7) << ws) | (it & ((wq >> 2) - 1))) * 16);
            uint8_t *dst = tile_out + (yi * stride + x);
            if (left >= 16) {
               typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
               __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4 __attribute__((aligned(JM_STORE_ALIGN))) *>(dst));
            } else {
               const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
               for (uint32_t i = 0; i < 15; i++)
                  if (i < left) dst[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
            }
         }
         return;
      }
#endif
      for (uint32_t it = tid; it < nitems; it += nt) {
         const uint32_t y = iy, x = ix * 4;
         iy += dy; ix += dx;
         if (ix >= wq) { ix -= wq; iy++; }
         if (x0 + x >= p.width) continue;
         uint32_t n = min(4u, p.width - (x0 + x));
         uint32_t v = *reinterpret_cast<const uint32_t *>(lds + p.plane_off[0] + (y & 7) * pitch0 + x);
         uint8_t *dst = out + (size_t)y * p.out_stride + x0 + x;
         if (n == 4 && ((uintptr_t)dst & 3) == 0) *reinterpret_cast<uint32_t *>(dst) = v;
         else for (uint32_t i = 0; i < n; i++) dst[i] = (uint8_t)(v >> (8 * i));
      }
      return;
   }
   const uint32_t pitch1 = p.bw[1] * 8, pitch2 = p.bw[2] * 8;
   const uint8_t *pl0 = lds + p.plane_off[0], *pl1 = lds + p.plane_off[1], *pl2 = lds + p.plane_off[2];
   if constexpr (HFC != 0) {
      // Y at full resolution, Cb and Cr share (HFC, VFC).  One item = 4 pixels wide x VFC rows, so
      // every chroma sample's offsets are computed once for all the pixels that replicate it.
      // tile_w / 4 is a power of two here (pick_kernel), and the planes are dense in LDS (pitch0 = tile_w,
      // pitch1 = pitch2 = tile_w / HFC): item -> (row, quad) is a shift and a mask, and the item index itself
      // is the word offset of the item's chroma samples
      const uint32_t ws = p.wq_shift;
      const uint32_t nit = (p.tile_rows / VFC) << ws;
      const uint32_t rem = p.width - x0;                       // pixels of this tile inside the frame
      uint8_t *tile_out = out + (size_t)x0 * 3;                // uniform; rows are reached with 32-bit offsets
      const uint32_t stride = (uint32_t)p.out_stride;          // <= 2^26 (run_device)
      for (uint32_t it = tid; it < nit; it += nt) {
         const uint32_t yi = it >> ws, x = (it & (wq - 1)) * 4;
         if (x >= rem) continue;
         const uint32_t n = min(4u, rem - x);
         const uint32_t y = yi * VFC;
         uint32_t cbw, crw;
         if (HFC == 1) {
            cbw = *reinterpret_cast<const uint32_t *>(pl1 + it * 4);
            crw = *reinterpret_cast<const uint32_t *>(pl2 + it * 4);
         } else {
            cbw = *reinterpret_cast<const uint16_t *>(pl1 + it * 2);
            crw = *reinterpret_cast<const uint16_t *>(pl2 + it * 2);
            cbw |= 0x01010000u;   // unused byte lanes can never look like the special pair
         }
         const uint8_t *yrow = pl0 + (it + yi * ((VFC - 1) << ws)) * 4;
         uint32_t yy[VFC];
#pragma unroll
         for (int v = 0; v < VFC; v++) yy[v] = *reinterpret_cast<const uint32_t *>(yrow + v * pitch0);
         uint32_t w[VFC][3];
         convert_item<HFC, VFC>(yy, cbw, crw, w);
#pragma unroll
         for (int v = 0; v < VFC; v++)
            store_rgb_quad(tile_out + ((y + v) * stride + x * 3), w[v][0], w[v][1], w[v][2], n);
      }
   } else {
      // generic replicate upsampling, any integer factors (src/decoder_utils.c:72-80)
      for (uint32_t it = tid; it < nitems; it += nt) {
         const uint32_t y = iy, x = ix * 4;
         iy += dy; ix += dx;
         if (ix >= wq) { ix -= wq; iy++; }
         if (x0 + x >= p.width) continue;
         uint32_t n = min(4u, p.width - (x0 + x));
         uint32_t px[4];
#pragma unroll
         for (int i = 0; i < 4; i++) {
            uint32_t xx = x + i;
            int s0 = pl0[(y / p.vf[0]) * pitch0 + xx / p.hf[0]];
            int s1 = pl1[(y / p.vf[1]) * pitch1 + xx / p.hf[1]];
            int s2 = pl2[(y / p.vf[2]) * pitch2 + xx / p.hf[2]];
            px[i] = ycc_to_rgb(s0, chroma_offsets(s1, s2));
         }
         uint32_t w0 = px[0] | (px[1] << 24);
         uint32_t w1 = (px[1] >> 8) | (px[2] << 16);
         uint32_t w2 = (px[2] >> 16) | (px[3] << 8);
         store_rgb_quad(out + (size_t)y * p.out_stride + (size_t)(x0 + x) * 3, w0, w1, w2, n);
      }
   }
}

// g / nbmcuH for a scalar g < total_mcus: one s_mul_hi_u32 with the host's floor(2^32 / nbmcuH) and one correction step
// (the estimate is short by at most 1), so that the quotient never touches the vector unit
__device__ __forceinline__ uint32_t mcu_row_of(const BackendParams &p, uint32_t g) {
   uint32_t q = __umulhi(g, p.inv_nbmcuH);
   if (g - q * p.nbmcuH >= p.nbmcuH) q++;
   return q;
}
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Phase 1 of one block: both IDCT passes in registers, 8 rows of 8 samples to the component's u8 plane in
// LDS.  zw = the block's 64 coefficients after iquant, zigzag order, two int16 per word.
template <int RS>
__device__ __forceinline__ void transform_block(const uint32_t (&zw)[32], uint8_t *plane, uint32_t pitch) {
   uint2 rows[8];
   {
      f32x2 mtx[4][8];
      JM_MARK("izz_int16_to_float");
      izz_rp(zw, mtx);   // pass-1 pairing (0,4)(2,6)(1,7)(3,5): pass 2 stays packed within a row
#if defined(JM_DIAG_NOIDCT)
#pragma unroll
      for (int r = 0; r < 8; r++) { float u[8];
#pragma unroll
         for (int c2 = 0; c2 < 8; c2++) u[c2] = (c2 & 1) ? mtx[c2 >> 1][r].y : mtx[c2 >> 1][r].x;
         rows[r] = pack_row(u); }
#else
      idct_block_rp(mtx, rows);
#endif
   }
   JM_MARK("lds_plane_writes");
   // RS kernels pass the wave's plane; the lane's column is added here, from a fresh read of the lane id, so that no
   // per-lane value but the block itself is live across the transform (72 VGPRs)
   if (RS == 1 || RS == 3) plane += lane_id() * 8;
#pragma unroll
   for (int r = 0; r < 8; r++) {
      if (RS == 1 || RS == 3) *reinterpret_cast<uint2 *>(plane + (r >> 1) * kRsUnitBytes + (r & 1) * kRsRowBytes) = rows[r];   // immediate offsets
      else *reinterpret_cast<uint2 *>(plane + r * pitch) = rows[r];
   }
}

// "-f": the direct transform idct (src/idct.c:50-71) instead of idct_opt.  Per sample (x, y) a 64-term
// sum in (lambda, mu) order, product and accumulation rounded separately in fp32, then *0.25, +128,
// clamp, truncate.  The table index is wave-uniform.  Not a fast path: 8192 fp32 operations per block.
__device__ __forceinline__ uint32_t slow_sample(const float (&m)[8][8], const float *__restrict__ c) {
   float sum = 0.0f;
#pragma unroll
   for (int l = 0; l < 8; l++)
#pragma unroll
      for (int mu = 0; mu < 8; mu++) {
         float val = c[l * 8 + mu];
         val *= m[l][mu];
         sum += val;
      }
   sum *= 0.25f;
   return to_u8(sum);
}

__device__ __forceinline__ void transform_block_slow(const uint32_t (&zw)[32], uint8_t *plane, uint32_t pitch, const float *__restrict__ coef) {
   float m[8][8];
   izz_plain(zw, m);   // m[lambda][mu] = bloct16_t.data[lambda][mu] (src/iqzz.c:23)
#pragma unroll 1
   for (uint32_t xy = 0; xy < 64; xy++)   // res->data[x][y]: column x, row y
      plane[(xy & 7) * pitch + (xy >> 3)] = (uint8_t)slow_sample(m, coef + xy * 64);
}

__device__ __forceinline__ const uint8_t *record_ptr(const BlockCtx &k, uint32_t tile, uint32_t m, uint32_t frame, bool &in_plane) {
   uint32_t col = tile * k.bw + k.bxl;
   in_plane = col < k.nbH;
#if !defined(JM_LOAD_ZEROFILL)
   col = min(col, k.nbH - 1);   // lanes right of the plane read its last column: their samples are never stored
#endif
   return k.coef + (size_t)frame * k.frame_stride + (((size_t)m * k.vs + k.by) * k.nbH + col) * 128;
}

// ---- the fused back end: one workgroup per tile, records loaded straight into registers ----------
// One workgroup = one tile of T MCUs of one MCU row, one thread per 8x8 block.  Each lane pulls its
// own 128-byte record with 8 x 16-byte loads (one full cache line per lane; tools/ubench_read.hip:
// 6.4-6.7 TB/s, no slower than lane-coalesced loads and no LDS staging or extra barrier needed).
//
// Two prologues.  When every wave of the tile holds blocks of ONE component (p.wave_uniform: all component
// boundaries are multiples of 64: 4:4:4, 4:2:2 and -- with its 384-block tiles, capi.cpp plan_backend -- 4:2:0 at
// 64 MCUs per tile) the component, its plane geometry and its quantisation table are wave-uniform: they live in
// SGPRs, the table is 32 scalar loads from the kernel arguments, and nothing is staged in LDS.  Otherwise (narrow
// frames whose tile shrinks, sampling layouts such as 4:1:1 + 2x1 chroma, the JPEGGPU_NO_UNIFORM / _TILE_MCUS knobs)
// every lane selects its component's fields with v_cndmask and the tables go through LDS.
template <int HFC, int VFC, bool SLOW, int RS = 0>   // chroma factors; 0 = generic path (any factors, any component); RS: restaged 16-byte stores
__device__ __forceinline__ void backend_body(const BackendParams &p) {
   extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
   const uint32_t tid = threadIdx.x, nt = blockDim.x;
   const uint32_t frame = blockIdx.z;
   // the tile's place in the frame (all scalar): T MCUs of MCU row blockIdx.y, or (linear tiling) MCUs
   // [blockIdx.x * T, + T) of the frame in raster order
   uint32_t tile = blockIdx.x, m = blockIdx.y;
   TileGeom geo;
   if (RS == 1 ? false : (RS == 3 ? true : p.linear != 0)) {
      const uint32_t g0 = tile * p.T;
      geo.m0 = mcu_row_of(p, g0);
      geo.c0 = g0 - geo.m0 * p.nbmcuH;
      geo.wrapj = p.nbmcuH - geo.c0;
      geo.nvalid = min(p.T, p.total_mcus - g0);
      m = geo.m0;
   } else {
      geo.m0 = m; geo.c0 = tile * p.T; geo.wrapj = 0x7FFFFFFFu; geo.nvalid = p.T;
   }
#if defined(JM_DIAG_STAMP)
   const long long ts0 = __builtin_amdgcn_s_memtime();
   const long long tr0 = __builtin_amdgcn_s_memrealtime();
#endif
   uint32_t rs_wave = 0;   // RS kernels: the wave's index in the workgroup (scalar)
   uint32_t zw[32];        // the block's record, dequantised in place
   uint8_t *plane;         // first sample of the block in its component's LDS plane
   uint32_t pitch;         // bytes per row of that plane
   bool active;
#if defined(JM_DIAG_STAMP)
   long long ts1;
#endif
   if constexpr (RS == 1 || RS == 3) {
      // 4:4:4 in 64-MCU tiles of 192 lanes: wave = component, lane = block = MCU of the tile; the only per-lane part of a
      // record's address is its column (and, under linear tiling, the MCU row it falls in)
      const uint32_t c = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
      rs_wave = c;
      const uint32_t nbH = p.nbH[c];
      uint32_t rec;                                            // record index from the start of the plane's MCU row m
      if constexpr (RS == 1) rec = geo.c0 + l;
      else {
         const bool r2 = l >= geo.wrapj && l < geo.nvalid;
         rec = r2 ? nbH + l - geo.wrapj : min(geo.c0 + l, nbH - 1);   // lanes past the frame's last MCU read a valid block: never stored
      }
      const uint8_t *base = static_cast<const uint8_t *>(p.coef[c]) + (size_t)frame * p.coef_frame_stride[c] + (size_t)m * nbH * 128;
      const uint4 *src = reinterpret_cast<const uint4 *>(base + rec * 128u);
      uint4 v[8];
      JM_MARK("record_loads");
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = src[i];
      JM_MARK("iquant");
      if (p.apply_iquant) {
         const uint32_t *q = p.qt[p.idq[c]];
#pragma unroll
         for (int i = 0; i < 8; i++) {
            zw[4 * i] = iquant_word(v[i].x, q[4 * i]); zw[4 * i + 1] = iquant_word(v[i].y, q[4 * i + 1]);
            zw[4 * i + 2] = iquant_word(v[i].z, q[4 * i + 2]); zw[4 * i + 3] = iquant_word(v[i].w, q[4 * i + 3]);
         }
      } else {
#pragma unroll
         for (int i = 0; i < 8; i++) { zw[4 * i] = v[i].x; zw[4 * i + 1] = v[i].y; zw[4 * i + 2] = v[i].z; zw[4 * i + 3] = v[i].w; }
      }
      JM_MARK("lds_address");
      pitch = 512;
      plane = lds + c * kRsPlaneBytes;       // + lane * 8 in transform_block
      active = true;
   } else
#if !defined(JM_NO_UNIFORM_PATH)
   if (RS != 0 || p.wave_uniform) {
      const uint32_t wb = __builtin_amdgcn_readfirstlane(tid);   // first block of this wave
      const uint32_t c = (p.nbcomp == 3) ? (uint32_t)(wb >= p.boff[1]) + (uint32_t)(wb >= p.boff[2]) : 0u;
      const uint32_t bw = p.bw[c], nbH = p.nbH[c], vs = p.vs[c];
      const uint32_t r = tid - p.boff[c];
      uint32_t by = 0;
      if (vs > 1) by = (uint32_t)(r >= bw) + (uint32_t)(r >= 2 * bw) + (uint32_t)(r >= 3 * bw);
      const uint32_t bxl = r - __umul24(by, bw);            // 24-bit multiplies (full rate): by <= 3, bw <= 1024
      // the block's MCU row: the tile's first one, or (linear tiling) the next one for the MCUs after the wrap point
      const uint32_t hsh = p.hs_shift[c];
      const uint32_t j = bxl >> hsh;                        // MCU of the tile this block belongs to
      const bool r2 = j >= geo.wrapj && j < geo.nvalid;
      const uint32_t col0 = p.linear ? (geo.c0 << hsh) : tile * bw;   // first block column of the tile (hs is a power of two under linear tiling only)
      uint32_t col = r2 ? bxl - (geo.wrapj << hsh) : col0 + bxl;
      col = min(col, nbH - 1);                              // lanes right of the plane / past the frame's last MCU read a valid block: never stored
      const uint8_t *base = static_cast<const uint8_t *>(p.coef[c]) + (size_t)frame * p.coef_frame_stride[c] +
                            (size_t)m * vs * nbH * 128;      // uniform; two MCU rows of a plane are < 2^32 bytes
      const uint4 *src = reinterpret_cast<const uint4 *>(base + (__umul24((r2 ? vs : 0u) + by, nbH) + col) * 128u);   // nbH < 2^18
      uint4 v[8];
      JM_MARK("record_loads");
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = src[i];   // NOT non-temporal: a lane's 8 loads share one line (nt: 1.6x slower)
#if defined(JM_DIAG_STAMP)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ts1 = __builtin_amdgcn_s_memtime();
#endif
      JM_MARK("iquant");
      if (p.apply_iquant) {
         const uint32_t *q = p.qt[p.idq[c]];
#pragma unroll
         for (int i = 0; i < 8; i++) {
            zw[4 * i] = iquant_word(v[i].x, q[4 * i]); zw[4 * i + 1] = iquant_word(v[i].y, q[4 * i + 1]);
            zw[4 * i + 2] = iquant_word(v[i].z, q[4 * i + 2]); zw[4 * i + 3] = iquant_word(v[i].w, q[4 * i + 3]);
         }
      } else {
#pragma unroll
         for (int i = 0; i < 8; i++) { zw[4 * i] = v[i].x; zw[4 * i + 1] = v[i].y; zw[4 * i + 2] = v[i].z; zw[4 * i + 3] = v[i].w; }
      }
      JM_MARK("lds_address");
      pitch = bw * 8;
      plane = lds + p.plane_off[c] + (by * 8) * pitch + bxl * 8;
      active = true;        // B is a multiple of 64 here
   } else
#endif
   {
      uint32_t *qt_lds = reinterpret_cast<uint32_t *>(lds + (size_t)nt * 64);
      active = tid < p.B;
      const BlockCtx k = decode_block(p, active ? tid : 0);
      uint4 v[8];
      {
         bool in_plane;
         const uint4 *src = reinterpret_cast<const uint4 *>(record_ptr(k, tile, m, frame, in_plane));
#if defined(JM_DIAG_NOLOAD)
         in_plane = false;
#endif
#if !defined(JM_LOAD_ZEROFILL) && !defined(JM_DIAG_NOLOAD)
         (void)in_plane;
         if (true) {   // idle lanes decode as block 0 and lanes right of the plane are clamped: every address is valid
#else
         if (active && in_plane) {
#endif
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = src[i];
         } else {
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = make_uint4(0, 0, 0, 0);
         }
      }
      if (p.apply_iquant && tid < 128) qt_lds[tid] = p.qt[tid >> 5][tid & 31];
      if (p.apply_iquant && nt < 128) qt_lds[tid + 64] = p.qt[(tid >> 5) + 2][tid & 31];   // nt == 64
      __syncthreads();
#if defined(JM_DIAG_STAMP)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ts1 = __builtin_amdgcn_s_memtime();
#endif
      if (p.apply_iquant) {
         const uint4 *q4 = reinterpret_cast<const uint4 *>(qt_lds + k.idq * 32);
#pragma unroll
         for (int i = 0; i < 8; i++) {
            const uint4 q = q4[i];
            zw[4 * i] = iquant_word(v[i].x, q.x); zw[4 * i + 1] = iquant_word(v[i].y, q.y);
            zw[4 * i + 2] = iquant_word(v[i].z, q.z); zw[4 * i + 3] = iquant_word(v[i].w, q.w);
         }
      } else {
#pragma unroll
         for (int i = 0; i < 8; i++) { zw[4 * i] = v[i].x; zw[4 * i + 1] = v[i].y; zw[4 * i + 2] = v[i].z; zw[4 * i + 3] = v[i].w; }
      }
      pitch = k.bw * 8;
      plane = lds + k.plane_off + (k.by * 8) * pitch + k.bxl * 8;
   }
   if (active) {
      if constexpr (SLOW) transform_block_slow(zw, plane, pitch, p.slow_coef);
      else transform_block<RS>(zw, plane, pitch);
   }
   JM_MARK("barrier");

#if defined(JM_DIAG_STAMP)
   asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
   const long long ts2 = __builtin_amdgcn_s_memtime();
#endif
   __syncthreads();
#if defined(JM_DIAG_STAMP)
   const long long ts3 = __builtin_amdgcn_s_memtime();
#endif
#if !defined(JM_DIAG_NOPHASE2)
   JM_MARK("pixel_setup");
   if constexpr (RS == 1 || RS == 3) {
      // the lane id is read again (2 instructions) instead of being kept in a register across phase 1: at 72 VGPRs it would be spilled
      const uint32_t lane = lane_id();
      pixel_phase_units444<RS == 3>(p, lds, rs_wave, lane, geo, frame);
      return;
   }
#if !defined(JM_NO_RUNS)
   if (HFC != 0 && p.wq_shift >= 6) pixel_phase_runs<(HFC ? HFC : 1), (VFC ? VFC : 1)>(p, lds, tid, nt, geo, frame);
   else
#endif
      pixel_phase<HFC, VFC>(p, lds, tid, nt, tile, m, frame);
#endif
#if defined(JM_DIAG_STAMP)
   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
   const long long ts4 = __builtin_amdgcn_s_memtime();
   if ((tid & 63) == 0 && p.diag) {
      const size_t wg = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      unsigned long long *slot = (unsigned long long *)p.diag + (wg * (nt >> 6) + (tid >> 6)) * 8;
      slot[0] = (unsigned long long)(ts1 - ts0); slot[1] = (unsigned long long)(ts2 - ts1); slot[2] = (unsigned long long)(ts3 - ts2);
      slot[3] = (unsigned long long)(ts4 - ts3); slot[4] = (unsigned long long)(__builtin_amdgcn_s_memrealtime() - tr0);
      slot[5] = (unsigned long long)tr0;
   }
#endif
}

// Two launch bounds per sub-sampling: most tiles are <= 256 blocks; 6-block MCUs use 384 (capi.cpp, plan_backend).
// Same body; the bound only changes the compiler's scheduling (the 256 build is ~1% faster where it applies).
template <int HFC, int VFC, int MAXT>
__global__ void __launch_bounds__(MAXT, JM_WAVES_PER_EU) backend_kernel(const BackendParams p) { backend_body<HFC, VFC, false>(p); }

// 4:4:4 in 64-MCU tiles that all lie inside the frame: pixel phase by row-pair units with 16-byte stores (pixel_phase_units444)
__global__ void __launch_bounds__(192, JM_RS_WAVES_PER_EU) backend_kernel_rs444(const BackendParams p) { backend_body<1, 1, false, 1>(p); }
// the same under linear tiling (RS = 3: any even number of MCUs per row above 64): per-lane wrap offsets and predicates
__global__ void __launch_bounds__(192, JM_RS_WAVES_PER_EU) backend_kernel_rs444_lin(const BackendParams p) { backend_body<1, 1, false, 3>(p); }

// ---- 4:2:0 in 32-MCU tiles with 16-byte stores (RS = 2) -----------------------------------------------------------
// The general kernel needs 64 MCUs = 384 lanes per tile to keep every wave on one component, which leaves room for
// four workgroups per CU.  Here a tile is 32 MCUs = 192 lanes: waves 0 and 1 hold the two block rows of Y (64 blocks
// each), wave 2 holds Cb in lanes 0..31 and Cr in lanes 32..63 -- both are 1x1 components of equal geometry and (host
// check) share a quantisation table, so the two differ in one 64-bit base address per lane.  The kernel needs
// 76-80 VGPRs (6 waves per SIMD = 8 workgroups per CU; forced to 72 it spills, and spills cost more than the ninth workgroup).
// LDS: unit u = raster rows 2u, 2u+1 of the tile = [Y row 2u (512 B) | Y row 2u+1 (512 B) | Cb row u (256 B) | Cr row u
// (256 B)] = 1536 bytes, 8 units.  Phase 2: a unit belongs to one wave; lane l reads 8 pixels x 2 rows (two ds_read_b64
// of Y, one dword of each chroma plane), converts, and the unit's 1536 bytes become the staging of ONE raster row at a
// time (row 2u: 24 bytes per lane in, 16-byte pieces out; then row 2u+1), so that the stores are the three of the
// 4:4:4 kernel: 1024 B of row 2u | 512 B of row 2u + 512 B of row 2u+1 | 1024 B of row 2u+1.  8 units on 3 waves.
// Linear tiling (frames whose MCU rows are not a whole number of tiles): a 16-byte piece never straddles an MCU (48
// bytes), so the wrap into the next MCU row is a per-lane choice between two offsets, made once per tile.
constexpr uint32_t kR2Unit = 1536, kR2Cb = 1024, kR2Cr = 1280;
#ifndef JM_R2_WAVES_PER_EU
#define JM_R2_WAVES_PER_EU 6
#endif
// Phase 1 of one wave of the 4:2:0 kernel.  Y (CHROMA = false): block row `wave` of the MCU row, 64 blocks, a scalar base
// plus a 32-bit lane offset.  CHROMA: lanes 0-31 Cb, 32-63 Cr, one 64-bit base per half.  LDS: Y wave w, block row r ->
// raster row R = 8 w + r of the tile: unit R >> 1, its row R & 1; chroma row r -> unit r: pl + (r >> 1) * U + (r & 1) * V
// with U, V = 1536, 512 (Y) / 3072, 1536 (chroma; lanes 32..63: kR2Cb + 256 + ... = kR2Cr + ...).
template <bool LIN, bool CHROMA>
__device__ __forceinline__ void rs420_phase1(const BackendParams &p, uint8_t *lds, uint32_t wave, uint32_t frame, uint32_t m0, uint32_t c0,
                                             uint32_t wrapj, uint32_t nvalid) {
   uint32_t zw[32];
   {
      const uint32_t l = threadIdx.x & 63;
      uint4 v[8];
      uint32_t idq;
      if constexpr (!CHROMA) {
         const uint32_t nbH = p.nbH[0];
         uint32_t rec;                                      // record index from the start of block row (2 m0 + wave) of the plane
         if constexpr (LIN) {
            const uint32_t j = l >> 1;
            const bool r2 = j >= wrapj && j < nvalid;
            rec = r2 ? 2 * nbH + l - 2 * wrapj : min(2 * c0 + l, nbH - 1);   // lanes past the frame's last MCU read a valid block: never stored
         } else rec = 2 * c0 + l;
         const uint8_t *base = static_cast<const uint8_t *>(p.coef[0]) + (size_t)frame * p.coef_frame_stride[0] + ((size_t)m0 * 2 + wave) * nbH * 128;
         const uint4 *src = reinterpret_cast<const uint4 *>(base + rec * 128u);
         JM_MARK("record_loads");
#pragma unroll
         for (int i = 0; i < 8; i++) v[i] = src[i];
         idq = p.idq[0];
      } else {
         const uint32_t nbH = p.nbH[1], j = l & 31;
         uint32_t rec;
         if constexpr (LIN) {
            const bool r2 = j >= wrapj && j < nvalid;
            rec = r2 ? nbH + j - wrapj : min(c0 + j, nbH - 1);
         } else rec = c0 + j;
         const uint8_t *b1 = static_cast<const uint8_t *>(p.coef[1]) + (size_t)frame * p.coef_frame_stride[1] + (size_t)m0 * nbH * 128;
         const uint8_t *b2 = static_cast<const uint8_t *>(p.coef[2]) + (size_t)frame * p.coef_frame_stride[2] + (size_t)m0 * nbH * 128;
         const uint4 *src = reinterpret_cast<const uint4 *>((l < 32 ? b1 : b2) + rec * 128u);
         JM_MARK("record_loads");
#pragma unroll
         for (int i = 0; i < 8; i++) v[i] = src[i];
         idq = p.idq[1];
      }
      JM_MARK("iquant");
      if (p.apply_iquant) {
         const uint32_t *q = p.qt[idq];
#pragma unroll
         for (int i = 0; i < 8; i++) {
            zw[4 * i] = iquant_word(v[i].x, q[4 * i]); zw[4 * i + 1] = iquant_word(v[i].y, q[4 * i + 1]);
            zw[4 * i + 2] = iquant_word(v[i].z, q[4 * i + 2]); zw[4 * i + 3] = iquant_word(v[i].w, q[4 * i + 3]);
         }
      } else {
#pragma unroll
         for (int i = 0; i < 8; i++) { zw[4 * i] = v[i].x; zw[4 * i + 1] = v[i].y; zw[4 * i + 2] = v[i].z; zw[4 * i + 3] = v[i].w; }
      }
   }
   uint2 rows[8];
   {
      f32x2 mtx[4][8];
      JM_MARK("izz_int16_to_float");
      izz_rp(zw, mtx);
#if defined(JM_DIAG_NOIDCT)
#pragma unroll
      for (int r = 0; r < 8; r++) { float u[8];
#pragma unroll
         for (int c2 = 0; c2 < 8; c2++) u[c2] = (c2 & 1) ? mtx[c2 >> 1][r].y : mtx[c2 >> 1][r].x;
         rows[r] = pack_row(u); }
#else
      idct_block_rp(mtx, rows);
#endif
   }
   JM_MARK("lds_plane_writes");
   // the lane id is read again here: nothing per-lane but the block itself lives across the transform
   uint8_t *pl = lds + (CHROMA ? kR2Cb : wave * (4 * kR2Unit)) + lane_id() * 8;
#pragma unroll
   for (int r = 0; r < 8; r++)
      *reinterpret_cast<uint2 *>(pl + (CHROMA ? (r >> 1) * (2 * kR2Unit) + (r & 1) * kR2Unit : (r >> 1) * kR2Unit + (r & 1) * 512u)) = rows[r];
}

template <bool LIN>
__device__ __forceinline__ void rs420_body(const BackendParams &p) {
   extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
   typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
   const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   const uint32_t frame = blockIdx.z;
   uint32_t m0, c0, wrapj = 0x7FFFFFFFu, nvalid = 32;
   if constexpr (LIN) {
      const uint32_t g0 = blockIdx.x * 32;
      m0 = mcu_row_of(p, g0); c0 = g0 - m0 * p.nbmcuH; wrapj = p.nbmcuH - c0; nvalid = min(32u, p.total_mcus - g0);
   } else {
      m0 = blockIdx.y; c0 = blockIdx.x * 32;
#if !defined(JM_NO_PAIR_XCD)
      // Workgroups go to the 8 XCDs round-robin by linear id.  Dealt so that the two workgroups an XCD receives one after
      // the other (ids i and i + 8) are horizontally adjacent tiles, each XCD's L2 writes 3072 contiguous bytes per raster
      // row instead of 1536: c3 +0.7 % (profiles/r03_kernel_experiments.txt; the same mapping costs 4:4:4 0.5 %)
      const uint32_t X = gridDim.x;
      if (((X * gridDim.y) & 15) == 0) {
         const uint32_t i = blockIdx.x + X * blockIdx.y, k = i & 7, n = i >> 3;
         const uint32_t j = ((n >> 1) << 4) + (k << 1) + (n & 1);
         if ((X & (X - 1)) == 0) { const uint32_t sh = (uint32_t)__builtin_ctz(X); m0 = j >> sh; c0 = (j & (X - 1)) * 32; }   // scalar shifts
         else { m0 = j / X; c0 = (j - m0 * X) * 32; }
      }
#endif
   }
   // ---- phase 1: the Y waves and the chroma wave each run a copy of their own (scalar branch) ----
   if (wave < 2) rs420_phase1<LIN, false>(p, lds, wave, frame, m0, c0, wrapj, nvalid);
   else rs420_phase1<LIN, true>(p, lds, wave, frame, m0, c0, wrapj, nvalid);
   JM_MARK("barrier");
   __syncthreads();
#if !defined(JM_DIAG_NOPHASE2)
   // ---- phase 2 ----
   JM_MARK("pixel_setup");
   const uint32_t l = lane_id();
   const uint32_t stride = (uint32_t)p.out_stride;
   const uint32_t band = 16 * stride;
   // the three pieces of a lane: (row of the unit, byte offset in the tile's 1536-byte row)
   const uint32_t b0 = l * 16, b1 = (l * 16 + 1024) % kR2Unit, b2 = 512 + l * 16;
   const uint32_t r1 = l < 32 ? 0u : stride;
   uint32_t o0, o1, o2;
   bool ok0 = true, ok1 = true, ok2 = true;
   if constexpr (LIN) {
      // MCU of a piece = byte / 48; before the wrap it lies c0 MCUs into MCU row m0, after it at the start of the next one
      const uint32_t j0 = b0 / 48, j1 = b1 / 48, j2 = b2 / 48;
      const uint32_t before = c0 * 48, after = band - wrapj * 48;
      o0 = b0 + (j0 < wrapj ? before : after); ok0 = j0 < nvalid;
      o1 = b1 + (j1 < wrapj ? before : after) + r1; ok1 = j1 < nvalid;
      o2 = b2 + (j2 < wrapj ? before : after) + stride; ok2 = j2 < nvalid;
   } else {
      o0 = c0 * 48 + b0; o1 = c0 * 48 + b1 + r1; o2 = c0 * 48 + b2 + stride;
   }
   uint8_t *tile_out = static_cast<uint8_t *>(p.out) + (size_t)frame * p.out_frame_stride + (size_t)m0 * band;
   for (uint32_t u = wave; u < 8; u += 3) {
      JM_MARK("unit_lds_reads");
      uint8_t *ul = lds + u * kR2Unit;
      const uint2 ya = *reinterpret_cast<const uint2 *>(ul + l * 8);
      const uint2 yb = *reinterpret_cast<const uint2 *>(ul + 512 + l * 8);
      const uint32_t cbw = *reinterpret_cast<const uint32_t *>(ul + kR2Cb + l * 4);
      const uint32_t crw = *reinterpret_cast<const uint32_t *>(ul + kR2Cr + l * 4);
      uint32_t wa[2][3], wb[2][3];      // rows 2u and 2u+1, two quads each
#if defined(JM_DIAG_NOCOLOR)
      wa[0][0] = ya.x; wa[0][1] = cbw; wa[0][2] = crw; wa[1][0] = ya.y; wa[1][1] = cbw; wa[1][2] = crw;
      wb[0][0] = yb.x; wb[0][1] = cbw; wb[0][2] = crw; wb[1][0] = yb.y; wb[1][1] = cbw; wb[1][2] = crw;
#else
      JM_MARK("unit_special_pair_test");
      if (__builtin_expect(__any(has_special_pair(cbw, crw)), 0)) {
         JM_MARK("unit_exact_colour_cold");
#pragma unroll
         for (int q = 0; q < 2; q++) {
            uint32_t pa[4], pb[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
               const int ci = 2 * q + (i >> 1);
               const ChromaOff co = chroma_offsets((cbw >> (8 * ci)) & 0xFF, (crw >> (8 * ci)) & 0xFF);
               pa[i] = ycc_to_rgb(((q ? ya.y : ya.x) >> (8 * i)) & 0xFF, co);
               pb[i] = ycc_to_rgb(((q ? yb.y : yb.x) >> (8 * i)) & 0xFF, co);
            }
            wa[q][0] = pa[0] | (pa[1] << 24); wa[q][1] = (pa[1] >> 8) | (pa[2] << 16); wa[q][2] = (pa[2] >> 16) | (pa[3] << 8);
            wb[q][0] = pb[0] | (pb[1] << 24); wb[q][1] = (pb[1] >> 8) | (pb[2] << 16); wb[q][2] = (pb[2] >> 16) | (pb[3] << 8);
         }
      } else {
         JM_MARK("unit_colour");
         const uint32_t cbs = cbw ^ 0x80808080u, crs = crw ^ 0x80808080u;
         const ChromaPairs oa = chroma_pairs<2>(cbs, crs), ob = chroma_pairs<2>(cbs >> 16, crs >> 16);
         rgb_quad_pk(ya.x, oa, wa[0]); rgb_quad_pk(ya.y, ob, wa[1]);
         rgb_quad_pk(yb.x, oa, wb[0]); rgb_quad_pk(yb.y, ob, wb[1]);
      }
#endif
      JM_MARK("unit_restage");
      uint8_t *ub = tile_out + (size_t)(2 * u) * stride;      // wave-uniform
      // row 2u through the unit's own 1536 bytes (every lane's samples are in registers: wa / wb depend on all four reads)
      *reinterpret_cast<uint2 *>(ul + l * 24) = make_uint2(wa[0][0], wa[0][1]);
      *reinterpret_cast<uint2 *>(ul + l * 24 + 8) = make_uint2(wa[0][2], wa[1][0]);
      *reinterpret_cast<uint2 *>(ul + l * 24 + 16) = make_uint2(wa[1][1], wa[1][2]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const u32x4 s0 = *reinterpret_cast<const u32x4 *>(ul + b0);
      u32x4 s1 = u32x4{0, 0, 0, 0};
      if (l < 32) s1 = *reinterpret_cast<const u32x4 *>(ul + b1);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // row 2u+1
      *reinterpret_cast<uint2 *>(ul + l * 24) = make_uint2(wb[0][0], wb[0][1]);
      *reinterpret_cast<uint2 *>(ul + l * 24 + 8) = make_uint2(wb[0][2], wb[1][0]);
      *reinterpret_cast<uint2 *>(ul + l * 24 + 16) = make_uint2(wb[1][1], wb[1][2]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (l >= 32) s1 = *reinterpret_cast<const u32x4 *>(ul + b1);
      const u32x4 s2 = *reinterpret_cast<const u32x4 *>(ul + b2);
      JM_MARK("unit_stores");
      if (ok0) rs_store16_s(ub, o0, s0);
      if (ok1) rs_store16_s(ub, o1, s1);
      if (ok2) rs_store16_s(ub, o2, s2);
      JM_MARK("unit_loop");
   }
#endif
}
__global__ void __launch_bounds__(192, JM_R2_WAVES_PER_EU) backend_kernel_rs420(const BackendParams p) { rs420_body<false>(p); }
__global__ void __launch_bounds__(192, JM_R2_WAVES_PER_EU) backend_kernel_rs420_lin(const BackendParams p) { rs420_body<true>(p); }

// the "-f" variant: generic pixel phase, no occupancy target (the direct transform keeps the block in 64 VGPRs)
__global__ void __launch_bounds__(JM_MAX_THREADS) backend_kernel_slow(const BackendParams p) { backend_body<0, 0, true>(p); }

// Layouts with a sub-sampling-specific pixel phase: three components, Y at full resolution, Cb and Cr sharing factors of
// 1 or 2, and a tile row that is a power of two wide.
bool fast_colour_layout(const BackendParams &p) {
   return !p.slow_coef && p.nbcomp == 3 && p.wq_shift != 0xFFFFFFFFu && p.hf[0] == 1 && p.vf[0] == 1 && p.hf[1] == p.hf[2] &&
          p.vf[1] == p.vf[2] && (p.hf[1] == 1 || p.hf[1] == 2) && (p.vf[1] == 1 || p.vf[1] == 2);
}

typedef void (*BackendFn)(const BackendParams);
static BackendFn pick_kernel(const BackendParams &p, uint32_t threads) {
   if (p.slow_coef) return (BackendFn)backend_kernel_slow;
   if (p.restage == 1) return (BackendFn)backend_kernel_rs444;
   if (p.restage == 3) return (BackendFn)backend_kernel_rs444_lin;
   if (p.restage == 2) return p.linear ? (BackendFn)backend_kernel_rs420_lin : (BackendFn)backend_kernel_rs420;
#define JM_PICK(H, V) (threads <= 256 ? (BackendFn)backend_kernel<H, V, 256> : (BackendFn)backend_kernel<H, V, JM_MAX_THREADS>)
   if (fast_colour_layout(p)) {
      const uint32_t h = p.hf[1], v = p.vf[1];
      if (h == 1 && v == 1) return JM_PICK(1, 1);
      if (h == 2 && v == 1) return JM_PICK(2, 1);
      if (h == 2 && v == 2) return JM_PICK(2, 2);
      if (h == 1 && v == 2) return JM_PICK(1, 2);
   }
   return JM_PICK(0, 0);
#undef JM_PICK
}

hipError_t launch_backend(const BackendParams &p, const LaunchDims &d, hipStream_t s) {
   dim3 grid(d.tiles_per_row, d.mcu_rows, d.frames), block(d.threads);   // linear tiling: tiles of the frame x 1
   if (d.threads > JM_MAX_THREADS) return hipErrorInvalidConfiguration;
   if (p.restage != 2 && p.linear && (!fast_colour_layout(p) || !p.wave_uniform || p.wq_shift < 6 || d.mcu_rows != 1)) return hipErrorInvalidConfiguration;
   if (p.restage == 2 && (p.slow_coef || p.nbcomp != 3 || p.T != 32 || d.threads != 192 || p.hf[0] != 1 || p.vf[0] != 1 || p.hf[1] != 2 || p.vf[1] != 2 ||
                          p.hf[2] != 2 || p.vf[2] != 2 || p.nbH[0] != 2 * p.nbmcuH || p.nbH[1] != p.nbmcuH || p.nbH[2] != p.nbmcuH ||
                          p.width != p.nbmcuH * 16 || p.nbmcuH < 32 || (p.apply_iquant && p.idq[1] != p.idq[2]) ||
                          (p.linear ? (d.mcu_rows != 1 || d.tiles_per_row != (p.total_mcus + 31) / 32) : (p.nbmcuH % 32 != 0 || d.tiles_per_row != p.nbmcuH / 32)) ||
                          (((uintptr_t)p.out | p.out_stride | p.out_frame_stride) & 15) || d.lds_bytes < 8 * kR2Unit))
      return hipErrorInvalidConfiguration;
   if ((p.restage == 1 || p.restage == 3) && ((p.restage == 3) != (p.linear != 0) || p.slow_coef || !p.wave_uniform || !fast_colour_layout(p) || p.hf[1] != 1 || p.vf[1] != 1 || p.T != 64 || d.threads != 192 ||
                          p.width != p.nbmcuH * 8 || (p.linear ? (p.nbmcuH % 2 != 0 || p.nbmcuH <= 64) : p.nbmcuH != d.tiles_per_row * 64) ||
                          p.plane_off[1] != kRsPlaneBytes || p.plane_off[2] != 2 * kRsPlaneBytes ||
                          (((uintptr_t)p.out | p.out_stride | p.out_frame_stride) & 15) || d.lds_bytes < 4 * kRsUnitBytes))
      return hipErrorInvalidConfiguration;
   BackendFn k = pick_kernel(p, d.threads);
   if (d.lds_bytes > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)d.lds_bytes);
      if (e != hipSuccess) return e;
   }
   (void)hipGetLastError();   // a stale error of this thread (any earlier HIP call, ours or the caller's) is not this launch's
   hipLaunchKernelGGL(k, grid, block, d.lds_bytes, s, p);
   return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Statically allocated device staging for the allocation-free entry point (jpeggpu_backend_run_noalloc):
// coefficient planes in, raster out, for callers that may not call hipMalloc.  Every GPU the library runs on
// holds its own copy of this array, at its own address, until the process ends.  In host code the NAME `scratch`
// is only a handle for hipGetSymbolAddress: the device address is looked up once per device (with that device
// current) and everything -- copies, kernel arguments -- uses that pointer.
// ------------------------------------------------------------------------------------------
static __device__ __align__(256) uint8_t scratch[kStaticScratchBytes];

hipError_t static_scratch(int device, void **ptr) {
   static std::mutex mu;
   static void *addr[kMaxDevices] = {};
   if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
   std::lock_guard<std::mutex> lk(mu);
   if (!addr[device]) {
      hipError_t e = hipSetDevice(device);
      if (e != hipSuccess) return e;
      void *p = nullptr;
      e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
      if (e != hipSuccess) return e;
      if (!p) return hipErrorInvalidSymbol;
      addr[device] = p;
   }
   *ptr = addr[device];
   return hipSuccess;
}

// ------------------------------------------------------------------------------------------
// Stage-level kernels: the four function contracts over arrays (parity isolation, not the hot path)
// ------------------------------------------------------------------------------------------
struct QtArg { uint32_t w[32]; };

__global__ void iquant_kernel(int16_t *blocks, size_t n, int s0, int s1, QtArg q) {
   size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one coefficient per thread
   if (i >= n * 64) return;
   int k = (int)(i & 63);
   if (k < s0 || k > s1) return;
   uint32_t qq = (q.w[k >> 1] >> (16 * (k & 1))) & 0xFFFF;
   blocks[i] = (int16_t)(short)((int)blocks[i] * (int)qq);
}

// Progressive per-scan hand-off: a DC scan changes only coefficient 0 of every block.  pack[b] -> blocks[b].data[0].
__global__ void scatter_dc_kernel(int16_t *blocks, const int16_t *pack, size_t n) {
   const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (b < n) blocks[b * 64] = pack[b];
}

__global__ void izz_kernel(const int16_t *in, int16_t *out, size_t n) {
   size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one coefficient per thread
   if (i >= n * 64) return;
   constexpr uint8_t ZR[64] = JM_ZZ_ROWS;
   constexpr uint8_t ZC[64] = JM_ZZ_COLS;
   int k = (int)(i & 63);
   out[(i & ~(size_t)63) + ZC[k] * 8 + ZR[k]] = in[i];        // data[col][row], src/iqzz.c:23
}

__global__ void idct_kernel(const int16_t *in, uint8_t *out, size_t n) {
   size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one block per thread
   if (b >= n) return;
   float mtx[8][8];
   const int16_t *src = in + b * 64;
#pragma unroll
   for (int c = 0; c < 8; c++)
#pragma unroll
      for (int r = 0; r < 8; r++) mtx[c][r] = (float)src[c * 8 + r];   // bloct16_t data[col][row]
   uint2 rows[8];
   idct_block(mtx, rows);
   uint8_t *dst = out + b * 64;                                         // bloctu8_t data[col][row]
#pragma unroll
   for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int c = 0; c < 8; c++) dst[c * 8 + r] = (uint8_t)(((c < 4 ? rows[r].x : rows[r].y) >> (8 * (c & 3))) & 0xFF);
   }
}

__global__ void __launch_bounds__(64) idct_slow_kernel(const int16_t *in, uint8_t *out, size_t n, const float *__restrict__ coef) {
   size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one block per thread
   if (b >= n) return;
   float m[8][8];
   const int16_t *src = in + b * 64;
#pragma unroll
   for (int l = 0; l < 8; l++)
#pragma unroll
      for (int mu = 0; mu < 8; mu++) m[l][mu] = (float)src[l * 8 + mu];
   uint8_t *dst = out + b * 64;
#pragma unroll 1
   for (uint32_t xy = 0; xy < 64; xy++) dst[xy] = (uint8_t)slow_sample(m, coef + xy * 64);   // data[x][y]
}

// Four pixels per thread through the SAME arithmetic as the fused kernel's pixel phase (packed fast path,
// exact per-pixel form for a wave that holds the special chroma pair), so that the exhaustive 2^24 test of
// this stage also covers the fused path's colour code.
__global__ void ycc2rgb_kernel(const uint8_t *y, const uint8_t *cb, const uint8_t *cr, uint8_t *rgb, size_t n) {
   const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
   if (i0 >= n) return;
   const uint32_t cnt = (uint32_t)(n - i0 < 4 ? n - i0 : 4);
   uint32_t yw = 0, cbw = 0, crw = 0;
   for (uint32_t j = 0; j < cnt; j++) {
      yw |= (uint32_t)y[i0 + j] << (8 * j); cbw |= (uint32_t)cb[i0 + j] << (8 * j); crw |= (uint32_t)cr[i0 + j] << (8 * j);
   }
   uint32_t w[3];
   if (__any(has_special_pair(cbw, crw))) {
      uint32_t px[4];
#pragma unroll
      for (int i = 0; i < 4; i++) px[i] = ycc_to_rgb((yw >> (8 * i)) & 0xFF, chroma_offsets((cbw >> (8 * i)) & 0xFF, (crw >> (8 * i)) & 0xFF));
      w[0] = px[0] | (px[1] << 24); w[1] = (px[1] >> 8) | (px[2] << 16); w[2] = (px[2] >> 16) | (px[3] << 8);
   } else {
      rgb_quad_pk(yw, chroma_pairs<1>(cbw ^ 0x80808080u, crw ^ 0x80808080u), w);
   }
   for (uint32_t j = 0; j < 3 * cnt; j++) rgb[3 * i0 + j] = (uint8_t)(w[j >> 2] >> (8 * (j & 3)));
}

static inline unsigned nblocks(size_t work, unsigned per) { return (unsigned)((work + per - 1) / per); }

hipError_t launch_iquant(void *d_blocks, size_t n, int s_start, int s_end, const uint32_t qt_packed[32], hipStream_t s) {
   if (n == 0) return hipSuccess;
   QtArg q; for (int i = 0; i < 32; i++) q.w[i] = qt_packed[i];
   (void)hipGetLastError();
   hipLaunchKernelGGL(iquant_kernel, dim3(nblocks(n * 64, 256)), dim3(256), 0, s, static_cast<int16_t *>(d_blocks), n, s_start, s_end, q);
   return hipGetLastError();
}
hipError_t launch_scatter_dc(void *d_blocks, const void *d_pack, size_t n, hipStream_t s) {
   if (n == 0) return hipSuccess;
   (void)hipGetLastError();
   hipLaunchKernelGGL(scatter_dc_kernel, dim3(nblocks(n, 256)), dim3(256), 0, s, static_cast<int16_t *>(d_blocks), static_cast<const int16_t *>(d_pack), n);
   return hipGetLastError();
}
hipError_t launch_izz(const void *d_in, void *d_out, size_t n, hipStream_t s) {
   if (n == 0) return hipSuccess;
   (void)hipGetLastError();
   hipLaunchKernelGGL(izz_kernel, dim3(nblocks(n * 64, 256)), dim3(256), 0, s, static_cast<const int16_t *>(d_in), static_cast<int16_t *>(d_out), n);
   return hipGetLastError();
}
hipError_t launch_idct(const void *d_in, void *d_out, size_t n, hipStream_t s) {
   if (n == 0) return hipSuccess;
   (void)hipGetLastError();
   hipLaunchKernelGGL(idct_kernel, dim3(nblocks(n, 64)), dim3(64), 0, s, static_cast<const int16_t *>(d_in), static_cast<uint8_t *>(d_out), n);
   return hipGetLastError();
}
hipError_t launch_idct_slow(const void *d_in, void *d_out, size_t n, const float *d_coef, hipStream_t s) {
   if (n == 0) return hipSuccess;
   (void)hipGetLastError();
   hipLaunchKernelGGL(idct_slow_kernel, dim3(nblocks(n, 64)), dim3(64), 0, s, static_cast<const int16_t *>(d_in), static_cast<uint8_t *>(d_out), n, d_coef);
   return hipGetLastError();
}
hipError_t launch_ycc2rgb(const uint8_t *d_y, const uint8_t *d_cb, const uint8_t *d_cr, uint8_t *d_rgb, size_t n, hipStream_t s) {
   if (n == 0) return hipSuccess;
   (void)hipGetLastError();
   hipLaunchKernelGGL(ycc2rgb_kernel, dim3(nblocks((n + 3) / 4, 256)), dim3(256), 0, s, d_y, d_cb, d_cr, d_rgb, n);
   return hipGetLastError();
}

}  // namespace jpeggpu_k

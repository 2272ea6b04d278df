// jpeg_math.hpp -- device arithmetic of the pixel back end (gfx950).
//
// Bit-exact restatement of the reference's per-sample arithmetic, re-derived for the VALU:
//   * iquant   src/iqzz.c:9-14       int16*uint16 wrapped to int16
//   * idct_opt src/idct_opt.c:32-117  Loeffler 8-point, fp32 with three fp64 sites
//   * ycc2rgb  src/ycc2rgb.c:12-22    all-fp64 colour conversion
//
// The fp64 sites and the IEEE fp32 divide are replaced by 2-instruction fp32 sequences
// fma(x, Ch, x*Cl) whose result equals the reference rounding for EVERY float x; the colour
// conversion is replaced by integer multiply-shift formulas that equal the reference on all 2^24
// inputs.  Both claims are proven exhaustively by tools/explore_exact.c and tools/explore_color.c
// (run by tests/test_exact_identities.py).  The file MUST be compiled with -ffp-contract=off:
// every a*b+c below that is not an explicit fmaf() has two roundings in the reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

#define JM_HD __host__ __device__ __forceinline__

namespace jpegmath {

// ---- Loeffler constants -------------------------------------------------------------------
// fast_idct_coef (src/idct_opt.c:27-30): double literals narrowed to float.
#define JM_C1 ((float)0.9807852804)
#define JM_S1 ((float)0.195090322)
#define JM_C3 ((float)0.8314696123)
#define JM_S3 ((float)0.555570233)
#define JM_C6 ((float)0.3826834324)
#define JM_S6 ((float)0.9238795325)
// hi/lo float splits of the three irrational factors (tools/explore_exact.c, "bump 0"):
//   1/SQRT_2 (double literal, src/idct_opt.c:8), 1/(float)SQRT_2 (parameter k, :64),
//   SQRT_8 (:9).  Butterfly halvings (src/idct_opt.c:34-35) are exact power-of-two scalings, so
//   they are deferred and folded into these constants: each 1-D transform carries a pending
//   factor 2^-3 that the final normalisation constant absorbs (SQRT_8/8).
#define JM_RH2 0x1.6a09e6p+0f    // 2 * hi(1/SQRT_2)
#define JM_RL2 0x1.9fcef4p-26f   // 2 * lo(1/SQRT_2)
#define JM_KH2 0x1.6a09e6p+0f    // 2 * hi(1/(float)SQRT_2)
#define JM_KL2 0x1.9fcef4p-25f   // 2 * lo(1/(float)SQRT_2)
#define JM_SH8 0x1.6a09e6p-2f    // hi(SQRT_8) / 8
#define JM_SL8 0x1.9fcef4p-28f   // lo(SQRT_8) / 8

// Arithmetic is written once for T = float (one transform) and T = f32x2 (two independent
// transforms side by side, which the compiler lowers to v_pk_add_f32 / v_pk_mul_f32 /
// v_pk_fma_f32: the same IEEE operations per element, two per instruction).
typedef float f32x2 __attribute__((ext_vector_type(2)));
JM_HD float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
JM_HD f32x2 fma_t(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// (float)((double)x * C) or (float)((double)x / C') as one multiply and one fma.
template <typename T> JM_HD T mul_hl(T x, float hi, float lo) { return fma_t(x, (T)hi, x * (T)lo); }

// One 8-point inverse transform, in place: idct_opt_1D (src/idct_opt.c:56-81).
// v[] in natural coefficient order on entry, sample order on exit.
template <typename T> JM_HD void idct8(T (&v)[8]) {
   // reorder (:49-54)
   T t0 = v[0], t1 = v[4], t2 = v[2], t3 = v[6], t4 = v[7], t5 = v[3], t6 = v[5], t7 = v[1];
   T a, b;
   // stage 4 inverse (:58-61); pending scale of t4..t7 becomes 2^-1
   a = t7 + t4; b = t7 - t4; t7 = a; t4 = b;
   t5 = mul_hl(t5, JM_RH2, JM_RL2);
   t6 = mul_hl(t6, JM_RH2, JM_RL2);
   // stage 3 inverse (:62-66); t0..t3 -> 2^-1, t4..t7 -> 2^-2
   a = t0 + t1; b = t0 - t1; t0 = a; t1 = b;
   {
      T x = mul_hl(t2, JM_KH2, JM_KL2), y = mul_hl(t3, JM_KH2, JM_KL2);   // 2*(t/k)
      t2 = x * (T)JM_C6 - y * (T)JM_S6;
      t3 = y * (T)JM_C6 + x * (T)JM_S6;
   }
   a = t4 + t6; b = t4 - t6; t4 = a; t6 = b;
   a = t7 + t5; b = t7 - t5; t7 = a; t5 = b;
   // stage 2 inverse (:67-71); t0..t3 -> 2^-2
   a = t0 + t3; b = t0 - t3; t0 = a; t3 = b;
   a = t1 + t2; b = t1 - t2; t1 = a; t2 = b;
   {
      T x = t4, y = t7;
      t4 = x * (T)JM_C3 - y * (T)JM_S3;
      t7 = y * (T)JM_C3 + x * (T)JM_S3;
   }
   {
      T x = t5, y = t6;
      t5 = x * (T)JM_C1 - y * (T)JM_S1;
      t6 = y * (T)JM_C1 + x * (T)JM_S1;
   }
   // stage 1 inverse (:72-76); everything -> 2^-3
   a = t0 + t7; b = t0 - t7; t0 = a; t7 = b;
   a = t1 + t6; b = t1 - t6; t1 = a; t6 = b;
   a = t2 + t5; b = t2 - t5; t2 = a; t5 = b;
   a = t3 + t4; b = t3 - t4; t3 = a; t4 = b;
   // normalisation (:77-80) with the pending 2^-3 folded in
   v[0] = mul_hl(t0, JM_SH8, JM_SL8); v[1] = mul_hl(t1, JM_SH8, JM_SL8);
   v[2] = mul_hl(t2, JM_SH8, JM_SL8); v[3] = mul_hl(t3, JM_SH8, JM_SL8);
   v[4] = mul_hl(t4, JM_SH8, JM_SL8); v[5] = mul_hl(t5, JM_SH8, JM_SL8);
   v[6] = mul_hl(t6, JM_SH8, JM_SL8); v[7] = mul_hl(t7, JM_SH8, JM_SL8);
}

// Level shift, clamp, truncate: src/idct_opt.c:110-113.
JM_HD uint32_t to_u8(float s) {
   float x = s + 128.0f;
   x = __builtin_fminf(__builtin_fmaxf(x, 0.0f), 255.0f);   // inputs are never NaN
   return (uint32_t)(int)x;
}

// Eight samples of one raster row -> two packed words.  The +128 is an ordinary (round-to-nearest)
// fp32 add as in the reference; clamp + truncation are one v_cvt_pk_u8_f32 per sample: that
// instruction saturates to [0,255] and follows MODE.fp_round, so it is issued with the fp32
// rounding mode switched to round-toward-zero (probed on gfx950 by tools/probe_round.hip).  The
// whole mode window is a single asm statement, so no other fp32 instruction can be scheduled into it.
JM_HD uint2 pack_row(const float (&u)[8]) {
   uint2 w;
#if defined(__HIP_DEVICE_COMPILE__)
   float x0 = u[0] + 128.0f, x1 = u[1] + 128.0f, x2 = u[2] + 128.0f, x3 = u[3] + 128.0f;
   float x4 = u[4] + 128.0f, x5 = u[5] + 128.0f, x6 = u[6] + 128.0f, x7 = u[7] + 128.0f;
   asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                "s_nop 1\n\t"
                "v_cvt_pk_u8_f32 %0, %2, 0, 0\n\t"
                "v_cvt_pk_u8_f32 %1, %6, 0, 0\n\t"
                "v_cvt_pk_u8_f32 %0, %3, 1, %0\n\t"
                "v_cvt_pk_u8_f32 %1, %7, 1, %1\n\t"
                "v_cvt_pk_u8_f32 %0, %4, 2, %0\n\t"
                "v_cvt_pk_u8_f32 %1, %8, 2, %1\n\t"
                "v_cvt_pk_u8_f32 %0, %5, 3, %0\n\t"
                "v_cvt_pk_u8_f32 %1, %9, 3, %1\n\t"
                "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\t"
                "s_nop 1"
                : "=&v"(w.x), "=&v"(w.y)
                : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7));
#else
   w.x = to_u8(u[0]) | (to_u8(u[1]) << 8) | (to_u8(u[2]) << 16) | (to_u8(u[3]) << 24);
   w.y = to_u8(u[4]) | (to_u8(u[5]) << 8) | (to_u8(u[6]) << 16) | (to_u8(u[7]) << 24);
#endif
   return w;
}

// 2-D transform of one block held as m[c][r] (column c, row r), idct_opt (src/idct_opt.c:93-117):
// first along r for every column, then along c for every row.  Result packed as 8 raster rows of
// 8 bytes: row[r] = {lo: columns 0..3, hi: columns 4..7}, byte c = sample (c, r).
JM_HD void idct_block(float (&m)[8][8], uint2 (&row)[8]) {
#pragma unroll
   for (int c = 0; c < 8; c++) idct8(m[c]);
#pragma unroll
   for (int r = 0; r < 8; r++) {
      float u[8];
#pragma unroll
      for (int c = 0; c < 8; c++) u[c] = m[c][r];
      idct8(u);
      row[r] = pack_row(u);
   }
}

// Same transform with two columns (pass 1) / two rows (pass 2) per packed instruction.
// p1[cp][r] = { sample(col 2cp, row r), sample(col 2cp+1, row r) }.
JM_HD void idct_block_pk(f32x2 (&p1)[4][8], uint2 (&row)[8]) {
#pragma unroll
   for (int cp = 0; cp < 4; cp++) idct8(p1[cp]);
#pragma unroll
   for (int rp = 0; rp < 4; rp++) {
      f32x2 u[8];
#pragma unroll
      for (int c = 0; c < 8; c++) {
         const f32x2 lo = p1[c >> 1][2 * rp], hi = p1[c >> 1][2 * rp + 1];
         u[c] = (c & 1) ? f32x2{lo.y, hi.y} : f32x2{lo.x, hi.x};
      }
      idct8(u);
      float a[8], b[8];
#pragma unroll
      for (int c = 0; c < 8; c++) { a[c] = u[c].x; b[c] = u[c].y; }
      row[2 * rp] = pack_row(a);
      row[2 * rp + 1] = pack_row(b);
   }
}
// Hybrid: pass 1 packed (the column pairs come straight out of the dequantiser, no shuffles), pass 2
// one row at a time on the halves of the pair registers (no 2x2 transposes).
JM_HD void idct_block_hy(f32x2 (&p1)[4][8], uint2 (&row)[8]) {
#pragma unroll
   for (int cp = 0; cp < 4; cp++) idct8(p1[cp]);
#pragma unroll
   for (int r = 0; r < 8; r++) {
      float u[8];
#pragma unroll
      for (int c = 0; c < 8; c++) u[c] = (c & 1) ? p1[c >> 1][r].y : p1[c >> 1][r].x;
      idct8(u);
      row[r] = pack_row(u);
   }
}
JM_HD void idct_block_pk(float (&m)[8][8], uint2 (&row)[8]) {
   f32x2 p1[4][8];
#pragma unroll
   for (int cp = 0; cp < 4; cp++)
#pragma unroll
      for (int r = 0; r < 8; r++) p1[cp][r] = f32x2{m[2 * cp][r], m[2 * cp + 1][r]};
   idct_block_pk(p1, row);
}

// iquant (src/iqzz.c:12): product narrowed to int16 (wraps), then the int16 -> float conversion of
// src/idct_opt.c:97.
JM_HD float dequant_to_float(int zz, uint32_t q) {
   return (float)(int)(short)(zz * (int)q);
}

// zigzag index -> (row, col): the scan pinned by test/iqzz_test.c:21-29.
#define JM_ZZ_ROWS { 0,0,1,2,1,0,0,1, 2,3,4,3,2,1,0,0, 1,2,3,4,5,6,5,4, 3,2,1,0,0,1,2,3, \
                     4,5,6,7,7,6,5,4, 3,2,1,2,3,4,5,6, 7,7,6,5,4,3,4,5, 6,7,7,6,5,6,7,7 }
#define JM_ZZ_COLS { 0,1,0,0,1,2,3,2, 1,0,0,1,2,3,4,5, 4,3,2,1,0,0,1,2, 3,4,5,6,7,6,5,4, \
                     3,2,1,0,1,2,3,4, 5,6,7,7,6,5,4,3, 2,3,4,5,6,7,7,6, 5,4,5,6,7,7,6,7 }

// iquant + izz + int16->float for one block: zw = the 64 zigzag int16 as 32 little-endian words,
// qw = the table packed the same way.  m[col][row] as izz stores it (src/iqzz.c:23).  The product
// keeps its low 16 bits (src/iqzz.c:12 narrows to int16), which is exactly v_pk_mul_lo_u16.
typedef unsigned short jm_us2 __attribute__((ext_vector_type(2)));
JM_HD void dequant_izz(const uint32_t (&zw)[32], const uint32_t (&qw)[32], float (&m)[8][8]) {
   constexpr uint8_t ZR[64] = JM_ZZ_ROWS;
   constexpr uint8_t ZC[64] = JM_ZZ_COLS;
#pragma unroll
   for (int i = 0; i < 32; i++) {
      jm_us2 a = __builtin_bit_cast(jm_us2, zw[i]), b = __builtin_bit_cast(jm_us2, qw[i]);
      jm_us2 pr = a * b;
      m[ZC[2 * i]][ZR[2 * i]] = (float)(int)(short)pr.x;
      m[ZC[2 * i + 1]][ZR[2 * i + 1]] = (float)(int)(short)pr.y;
   }
}

// dequant_izz straight into the column-pair layout of idct_block_pk.
JM_HD void dequant_izz_pk(const uint32_t (&zw)[32], const uint32_t (&qw)[32], f32x2 (&p1)[4][8]) {
   constexpr uint8_t ZR[64] = JM_ZZ_ROWS;
   constexpr uint8_t ZC[64] = JM_ZZ_COLS;
#pragma unroll
   for (int i = 0; i < 32; i++) {
      jm_us2 a = __builtin_bit_cast(jm_us2, zw[i]), b = __builtin_bit_cast(jm_us2, qw[i]);
      jm_us2 pr = a * b;
      const float f0 = (float)(int)(short)pr.x, f1 = (float)(int)(short)pr.y;
      if (ZC[2 * i] & 1) p1[ZC[2 * i] >> 1][ZR[2 * i]].y = f0; else p1[ZC[2 * i] >> 1][ZR[2 * i]].x = f0;
      if (ZC[2 * i + 1] & 1) p1[ZC[2 * i + 1] >> 1][ZR[2 * i + 1]].y = f1; else p1[ZC[2 * i + 1] >> 1][ZR[2 * i + 1]].x = f1;
   }
}

// ---- colour conversion, src/ycc2rgb.c:18-22 -------------------------------------------------
// r = clamp8(y + ((91881*(cr-128) + 100) >> 16)), b = clamp8(y + ((116130*(cb-128) + 128) >> 16)),
// g = clamp8(y + ((120 - 2886855*(cb-128) - 5990640*(cr-128)) >> 23) - fix) where fix = 1 only
// for (cb, cr) = (28, 228) and 94 <= y <= 165: there the real-number offset is exactly -37 and the
// reference's double evaluation lands just below the integer (tools/explore_color.c).
struct ChromaOff { int r, g, b, fix; };
JM_HD ChromaOff chroma_offsets(int cb, int cr) {
   ChromaOff o;
   o.r = (91881 * cr + (100 - 91881 * 128)) >> 16;
   o.b = (116130 * cb + (128 - 116130 * 128)) >> 16;
   o.g = ((120 + 2886855 * 128 + 5990640 * 128) - 2886855 * cb - 5990640 * cr) >> 23;
   o.fix = (cb == 28) & (cr == 228);
   return o;
}
JM_HD int clamp8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
// returns r | g<<8 | b<<16
JM_HD uint32_t ycc_to_rgb(int y, const ChromaOff &o) {
   int g = y + o.g - (o.fix & ((unsigned)(y - 94) < 72u));
   return (uint32_t)clamp8(y + o.r) | ((uint32_t)clamp8(g) << 8) | ((uint32_t)clamp8(y + o.b) << 16);
}

// Fast path used by the fused kernel: offsets without the per-sample pair test; the caller tests
// whole words of chroma samples for the special pair with has_special_pair() and only then takes
// the exact per-pixel form above.
struct Off3 { int r, g, b; };
// du = cb - 128, dv = cr - 128 (callers get them for free as sign-extended bytes of word ^ 0x80808080).
// With the centred form the R and B biases may be 0 (both lie inside the admissible intervals
// printed by tools/explore_color.c), so each is one 24-bit multiply whose upper half is the offset.
JM_HD Off3 chroma_off3(int du, int dv) {
   Off3 o;
#if defined(__HIP_DEVICE_COMPILE__)
   o.r = __mul24(dv, 91881) >> 16;
   o.b = __mul24(du, 116130) >> 16;
   o.g = (__mul24(dv, -5990640) + __mul24(du, -2886855) + 120) >> 23;
#else
   o.r = (91881 * dv) >> 16;
   o.b = (116130 * du) >> 16;
   o.g = (120 - 2886855 * du - 5990640 * dv) >> 23;
#endif
   return o;
}
// true iff some byte position holds cb == 28 and cr == 228 (zero-byte test on the xor words)
JM_HD bool has_special_pair(uint32_t cbw, uint32_t crw) {
   uint32_t t = (cbw ^ 0x1C1C1C1Cu) | (crw ^ 0xE4E4E4E4u);
   return ((t - 0x01010101u) & ~t & 0x80808080u) != 0;
}

}  // namespace jpegmath

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

// Phase markers for the instruction table of tools/isa_phases.py (diagnostic builds only, -DJM_PHASE_MARKS): a comment in
// the ISA between two scheduling barriers, so that every instruction can be attributed to the phase it was written in.
#if defined(JM_PHASE_MARKS) && defined(__HIP_DEVICE_COMPILE__)
#define JM_MARK(name) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; JM_PHASE " name ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define JM_MARK(name) do { } while (0)
#endif

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
// x[] already carries the +128
JM_HD uint32_t to_u8_biased(float x) { return (uint32_t)(int)__builtin_fminf(__builtin_fmaxf(x, 0.0f), 255.0f); }
JM_HD uint2 pack_row_biased(const float (&x)[8]) {
   uint2 w;
#if defined(__HIP_DEVICE_COMPILE__)
   const float x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3], x4 = x[4], x5 = x[5], x6 = x[6], x7 = x[7];
#if defined(JM_DIAG_NOSETREG)   // timing experiment only (results are rounded, not truncated)
   asm volatile("v_cvt_pk_u8_f32 %0, %2, 0, 0\n\t"
                "v_cvt_pk_u8_f32 %1, %6, 0, 0\n\t"
                "v_cvt_pk_u8_f32 %0, %3, 1, %0\n\t"
                "v_cvt_pk_u8_f32 %1, %7, 1, %1\n\t"
                "v_cvt_pk_u8_f32 %0, %4, 2, %0\n\t"
                "v_cvt_pk_u8_f32 %1, %8, 2, %1\n\t"
                "v_cvt_pk_u8_f32 %0, %5, 3, %0\n\t"
                "v_cvt_pk_u8_f32 %1, %9, 3, %1"
                : "=&v"(w.x), "=&v"(w.y)
                : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7));
#else
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
#endif
#else
   w.x = to_u8_biased(x[0]) | (to_u8_biased(x[1]) << 8) | (to_u8_biased(x[2]) << 16) | (to_u8_biased(x[3]) << 24);
   w.y = to_u8_biased(x[4]) | (to_u8_biased(x[5]) << 8) | (to_u8_biased(x[6]) << 16) | (to_u8_biased(x[7]) << 24);
#endif
   return w;
}
JM_HD uint2 pack_row(const float (&u)[8]) {
   float x[8];
#pragma unroll
   for (int i = 0; i < 8; i++) x[i] = u[i] + 128.0f;
   return pack_row_biased(x);
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

// ---- pass 2 packed WITHIN a row ------------------------------------------------------------
// Pass 1 may pair any two columns in one packed register, so it pairs (0,4) (2,6) (1,7) (3,5): for a
// row those pairs are exactly the operand pairs of the first butterflies/rotations of idct_opt_1D
// ({t0,t1} {t2,t3} {t7,t4} {t5,t6} after its reorder), and every later stage can be kept in pairs too
// with the half-swaps expressed as VOP3P op_sel modifiers -- no transposes, no moves.  Each element
// sees the same IEEE operations in the same order as in idct8<float>.
//   jm_sumdiff(p)      = { p.x + p.y, p.x - p.y }
//   jm_cross_a(q2, q3) = { q2.y + q3.y, q2.x - q3.x }
//   jm_add_sub(a, b)   = { a.x + b.x, a.y - b.y }
//   jm_swap_mul(a, k)  = { a.y * k,  a.x * k }
//   jm_sub_add(a, b)   = { a.x - b.x, a.y + b.y }
//   jm_add_swap(a, b)  = { a.x + b.y, a.y + b.x }      jm_sub_swap(a, b) = { a.x - b.y, a.y - b.x }
#if defined(__HIP_DEVICE_COMPILE__)
#define JM_PK2(name, mnem, mods)                                                                   \
   __device__ __forceinline__ f32x2 name(f32x2 a, f32x2 b) {                                        \
      f32x2 d;                                                                                     \
      asm(mnem " %0, %1, %2 " mods : "=v"(d) : "v"(a), "v"(b));                                    \
      return d;                                                                                    \
   }
JM_PK2(jm_sumdiff2, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]")
JM_PK2(jm_cross_a, "v_pk_add_f32", "op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[0,1]")
JM_PK2(jm_add_sub, "v_pk_add_f32", "neg_hi:[0,1]")
JM_PK2(jm_sub_add, "v_pk_add_f32", "neg_lo:[0,1]")
JM_PK2(jm_add_swap, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0]")
JM_PK2(jm_sub_swap, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]")
JM_PK2(jm_swap_mul2, "v_pk_mul_f32", "op_sel:[1,0] op_sel_hi:[0,1]")
#undef JM_PK2
__device__ __forceinline__ f32x2 jm_sumdiff(f32x2 p) { return jm_sumdiff2(p, p); }
__device__ __forceinline__ f32x2 jm_swap_mul(f32x2 a, float k) { return jm_swap_mul2(a, f32x2{k, k}); }
#else
JM_HD f32x2 jm_sumdiff(f32x2 p) { return f32x2{p.x + p.y, p.x - p.y}; }
JM_HD f32x2 jm_cross_a(f32x2 a, f32x2 b) { return f32x2{a.y + b.y, a.x - b.x}; }
JM_HD f32x2 jm_add_sub(f32x2 a, f32x2 b) { return f32x2{a.x + b.x, a.y - b.y}; }
JM_HD f32x2 jm_sub_add(f32x2 a, f32x2 b) { return f32x2{a.x - b.x, a.y + b.y}; }
JM_HD f32x2 jm_add_swap(f32x2 a, f32x2 b) { return f32x2{a.x + b.y, a.y + b.x}; }
JM_HD f32x2 jm_sub_swap(f32x2 a, f32x2 b) { return f32x2{a.x - b.y, a.y - b.x}; }
JM_HD f32x2 jm_swap_mul(f32x2 a, float k) { return f32x2{a.y * k, a.x * k}; }
#endif

// One row of pass 2.  In: P0 = {v0,v4}, P1 = {v2,v6}, P2 = {v1,v7}, P3 = {v3,v5} (natural coefficient
// indices).  Out: o[0..7] in sample order.
JM_HD void idct8_row_pk(f32x2 P0, f32x2 P1, f32x2 P2, f32x2 P3, float (&o)[8]) {
   // stage 4 inverse: X(7,4) on {t7,t4}; iO on {t5,t6}
   const f32x2 Q2 = jm_sumdiff(P2);                               // {t7', t4'}
   const f32x2 Q3 = mul_hl(P3, JM_RH2, JM_RL2);                   // {t5', t6'}
   // stage 3 inverse: X(0,1); C(2,3,k); X(4,6); X(7,5)
   const f32x2 S01 = jm_sumdiff(P0);                              // {t0', t1'}
   const f32x2 XY = mul_hl(P1, JM_KH2, JM_KL2);                   // {x, y}
   const f32x2 S23 = jm_sub_add(XY * (f32x2)JM_C6, jm_swap_mul(XY, JM_S6));   // {x*C6 - y*S6, y*C6 + x*S6}
   const f32x2 A = jm_cross_a(Q2, Q3);                            // {t4'+t6', t7'-t5'} = {t4'', t5''}
   const f32x2 B = jm_add_sub(Q2, Q3);                            // {t7'+t5', t4'-t6'} = {t7'', t6''}
   // stage 2 inverse: X(0,3) X(1,2); C(4,7,C3,S3) C(5,6,C1,S1)
   const f32x2 U = jm_add_swap(S01, S23);                         // {t0+t3, t1+t2}
   const f32x2 V = jm_sub_swap(S01, S23);                         // {t0-t3, t1-t2} = {t3'', t2''}
   const f32x2 Cc = f32x2{JM_C3, JM_C1}, Sc = f32x2{JM_S3, JM_S1};
   const f32x2 W45 = A * Cc - B * Sc;                             // {t4*C3 - t7*S3, t5*C1 - t6*S1}
   const f32x2 W76 = B * Cc + A * Sc;                             // {t7*C3 + t4*S3, t6*C1 + t5*S1}
   // stage 1 inverse + normalisation
   const f32x2 O01 = mul_hl(U + W76, JM_SH8, JM_SL8);             // {o0, o1}
   const f32x2 O76 = mul_hl(U - W76, JM_SH8, JM_SL8);             // {o7, o6}
   const f32x2 O32 = mul_hl(V + W45, JM_SH8, JM_SL8);             // {o3, o2}
   const f32x2 O45 = mul_hl(V - W45, JM_SH8, JM_SL8);             // {o4, o5}
   o[0] = O01.x; o[1] = O01.y; o[2] = O32.y; o[3] = O32.x; o[4] = O45.x; o[5] = O45.y; o[6] = O76.y; o[7] = O76.x;
}

// Same row with the level shift (+128, src/idct_opt.c:110) applied to the pairs: four packed adds instead of
// eight scalar ones, the same IEEE addition per element.
JM_HD void idct8_row_pk_biased(f32x2 P0, f32x2 P1, f32x2 P2, f32x2 P3, float (&o)[8]) {
   const f32x2 Q2 = jm_sumdiff(P2);
   const f32x2 Q3 = mul_hl(P3, JM_RH2, JM_RL2);
   const f32x2 S01 = jm_sumdiff(P0);
   const f32x2 XY = mul_hl(P1, JM_KH2, JM_KL2);
   const f32x2 S23 = jm_sub_add(XY * (f32x2)JM_C6, jm_swap_mul(XY, JM_S6));
   const f32x2 A = jm_cross_a(Q2, Q3);
   const f32x2 B = jm_add_sub(Q2, Q3);
   const f32x2 U = jm_add_swap(S01, S23);
   const f32x2 V = jm_sub_swap(S01, S23);
   const f32x2 Cc = f32x2{JM_C3, JM_C1}, Sc = f32x2{JM_S3, JM_S1};
   const f32x2 W45 = A * Cc - B * Sc;
   const f32x2 W76 = B * Cc + A * Sc;
   const f32x2 O01 = mul_hl(U + W76, JM_SH8, JM_SL8) + (f32x2)128.0f;
   const f32x2 O76 = mul_hl(U - W76, JM_SH8, JM_SL8) + (f32x2)128.0f;
   const f32x2 O32 = mul_hl(V + W45, JM_SH8, JM_SL8) + (f32x2)128.0f;
   const f32x2 O45 = mul_hl(V - W45, JM_SH8, JM_SL8) + (f32x2)128.0f;
   o[0] = O01.x; o[1] = O01.y; o[2] = O32.y; o[3] = O32.x; o[4] = O45.x; o[5] = O45.y; o[6] = O76.y; o[7] = O76.x;
}

// Column -> (pair, half) of the pass-1 pairing (0,4) (2,6) (1,7) (3,5).
#define JM_RP_PAIR { 0, 2, 1, 3, 0, 3, 1, 2 }
#define JM_RP_HALF { 0, 0, 0, 0, 1, 1, 1, 1 }
JM_HD void idct_block_rp(f32x2 (&p1)[4][8], uint2 (&row)[8]) {
   JM_MARK("pass1");
#pragma unroll
   for (int cp = 0; cp < 4; cp++) idct8(p1[cp]);
   JM_MARK("pass2_bias_pack");
#pragma unroll
   for (int r = 0; r < 8; r++) {
      float u[8];
      idct8_row_pk_biased(p1[0][r], p1[1][r], p1[2][r], p1[3][r], u);
      row[r] = pack_row_biased(u);
   }
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

// iquant alone on the packed words (v_pk_mul_lo_u16), and the inverse zigzag + int16 -> float of already
// dequantised words into the pairing of idct_block_rp / into m[col][row].
JM_HD uint32_t iquant_word(uint32_t zw, uint32_t qw) {
   return __builtin_bit_cast(uint32_t, (jm_us2)(__builtin_bit_cast(jm_us2, zw) * __builtin_bit_cast(jm_us2, qw)));
}
// The converted values are made opaque to the optimiser: it would otherwise rewrite the first butterflies of pass 1,
// (float)a +- (float)b, as (float)(a +- b) (exact for int16, so legal) -- one integer add and one conversion per OUTPUT where
// the packed form needs one conversion per input and one v_pk_add_f32 per two outputs: 16 instructions more per block.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(JM_NO_OPAQUE_CVT)
#define JM_OPAQUE(f) asm("" : "+v"(f))
#else
#define JM_OPAQUE(f) do { } while (0)
#endif
JM_HD void izz_rp(const uint32_t (&zw)[32], f32x2 (&p1)[4][8]) {
   constexpr uint8_t ZR[64] = JM_ZZ_ROWS;
   constexpr uint8_t ZC[64] = JM_ZZ_COLS;
   constexpr uint8_t PP[8] = JM_RP_PAIR;
   constexpr uint8_t PH[8] = JM_RP_HALF;
#pragma unroll
   for (int i = 0; i < 32; i++) {
      float f0 = (float)(int)(short)(zw[i] & 0xFFFF), f1 = (float)(int)(short)(zw[i] >> 16);
      JM_OPAQUE(f0); JM_OPAQUE(f1);
      if (PH[ZC[2 * i]]) p1[PP[ZC[2 * i]]][ZR[2 * i]].y = f0; else p1[PP[ZC[2 * i]]][ZR[2 * i]].x = f0;
      if (PH[ZC[2 * i + 1]]) p1[PP[ZC[2 * i + 1]]][ZR[2 * i + 1]].y = f1; else p1[PP[ZC[2 * i + 1]]][ZR[2 * i + 1]].x = f1;
   }
}
JM_HD void izz_plain(const uint32_t (&zw)[32], float (&m)[8][8]) {
   constexpr uint8_t ZR[64] = JM_ZZ_ROWS;
   constexpr uint8_t ZC[64] = JM_ZZ_COLS;
#pragma unroll
   for (int i = 0; i < 32; i++) {
      m[ZC[2 * i]][ZR[2 * i]] = (float)(int)(short)(zw[i] & 0xFFFF);
      m[ZC[2 * i + 1]][ZR[2 * i + 1]] = (float)(int)(short)(zw[i] >> 16);
   }
}

// dequant_izz straight into the pairing of idct_block_rp.
JM_HD void dequant_izz_rp(const uint32_t (&zw)[32], const uint32_t (&qw)[32], f32x2 (&p1)[4][8]) {
   constexpr uint8_t ZR[64] = JM_ZZ_ROWS;
   constexpr uint8_t ZC[64] = JM_ZZ_COLS;
   constexpr uint8_t PP[8] = JM_RP_PAIR;
   constexpr uint8_t PH[8] = JM_RP_HALF;
#pragma unroll
   for (int i = 0; i < 32; i++) {
      jm_us2 a = __builtin_bit_cast(jm_us2, zw[i]), b = __builtin_bit_cast(jm_us2, qw[i]);
      jm_us2 pr = a * b;
      const float f0 = (float)(int)(short)pr.x, f1 = (float)(int)(short)pr.y;
      if (PH[ZC[2 * i]]) p1[PP[ZC[2 * i]]][ZR[2 * i]].y = f0; else p1[PP[ZC[2 * i]]][ZR[2 * i]].x = f0;
      if (PH[ZC[2 * i + 1]]) p1[PP[ZC[2 * i + 1]]][ZR[2 * i + 1]].y = f1; else p1[PP[ZC[2 * i + 1]]][ZR[2 * i + 1]].x = f1;
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

// ---- packed 16-bit form of the fast colour path (device only) --------------------------------
// Four pixels of one raster row at a time.  The chroma offsets are the upper bits of the same 24-bit
// products as chroma_off3, gathered two to a register as int16 pairs (pixel 0 with pixel 2, pixel 1
// with pixel 3) by v_perm_b32; luma is widened to the same pairing; one v_pk_add_u16 adds a pair, one
// v_sat_pk_u8_i16 clamps it to [0,255] and narrows it to two bytes (the same clamp as clamp8, since
// |y + offset| < 2^15), and three v_perm_b32 interleave the result into r g b r | g b r g | b r g b.
typedef short jm_s2 __attribute__((ext_vector_type(2)));
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t jm_perm(uint32_t s0, uint32_t s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
__device__ __forceinline__ uint32_t jm_pk_add16(uint32_t a, uint32_t b) {
   return __builtin_bit_cast(uint32_t, (jm_us2)(__builtin_bit_cast(jm_us2, a) + __builtin_bit_cast(jm_us2, b)));
}
__device__ __forceinline__ uint32_t jm_pk_ashr16_7(uint32_t a) {
   return __builtin_bit_cast(uint32_t, (jm_s2)(__builtin_bit_cast(jm_s2, a) >> (jm_s2)7));
}
// {sat8(a.lo16), sat8(a.hi16)} in bytes 0,1 and {sat8(b.lo16), sat8(b.hi16)} in bytes 2,3
__device__ __forceinline__ uint32_t jm_sat2x2(uint32_t a, uint32_t b) {
   uint32_t d;
   asm("v_sat_pk_u8_i16_e32 %0, %1\n\t"
       "v_sat_pk_u8_i16_sdwa %0, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD"
       : "=&v"(d) : "v"(a), "v"(b));
   return d;
}
__device__ __forceinline__ int jm_mul24(int a, int b) { return __mul24(a, b); }
#else
// host mirrors of the instructions above (v_perm_b32: selector bytes 0-3 pick from s1, 4-7 from s0, 0x0c = zero)
JM_HD uint32_t jm_perm(uint32_t s0, uint32_t s1, uint32_t sel) {
   const uint64_t src = ((uint64_t)s0 << 32) | s1;
   uint32_t d = 0;
   for (int i = 0; i < 4; i++) {
      const uint32_t k = (sel >> (8 * i)) & 0xFF;
      const uint32_t byte = k < 8 ? (uint32_t)((src >> (8 * k)) & 0xFF) : 0u;
      d |= byte << (8 * i);
   }
   return d;
}
JM_HD uint32_t jm_pk_add16(uint32_t a, uint32_t b) { return ((a + b) & 0xFFFFu) | (((a >> 16) + (b >> 16)) << 16); }
JM_HD uint32_t jm_pk_ashr16_7(uint32_t a) {
   return ((uint32_t)((int)(short)(a & 0xFFFF) >> 7) & 0xFFFFu) | ((uint32_t)((int)(short)(a >> 16) >> 7) << 16);
}
JM_HD uint32_t jm_sat8(uint32_t h) { const int v = (short)h; return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
JM_HD uint32_t jm_sat2x2(uint32_t a, uint32_t b) {
   return jm_sat8(a & 0xFFFF) | (jm_sat8(a >> 16) << 8) | (jm_sat8(b & 0xFFFF) << 16) | (jm_sat8(b >> 16) << 24);
}
JM_HD int jm_mul24(int a, int b) { return a * b; }
#endif
// Offsets of the chroma samples under four horizontally adjacent pixels, as int16 pairs.
// HFC = 1: cbs/crs hold four samples (one per pixel); HFC = 2: two samples, each under two pixels.
struct ChromaPairs { uint32_t r02, r13, g02, g13, b02, b13; };
template <int HFC> JM_HD ChromaPairs chroma_pairs(uint32_t cbs, uint32_t crs) {
   constexpr int N = 4 / HFC;
   int pr[N], pb[N], pg[N];
#pragma unroll
   for (int ci = 0; ci < N; ci++) {
      const int du = (int)(int8_t)(cbs >> (8 * ci)), dv = (int)(int8_t)(crs >> (8 * ci));
      pr[ci] = jm_mul24(dv, 91881);                                        // offset = bits 31..16
      pb[ci] = jm_mul24(du, 116130);
      pg[ci] = jm_mul24(dv, -5990640) + jm_mul24(du, -2886855) + 120;      // offset = bits 31..23
   }
   ChromaPairs o;
   constexpr uint32_t HI = 0x07060302u;   // { S1.hi16, S0.hi16 }
   if constexpr (HFC == 1) {
      o.r02 = jm_perm((uint32_t)pr[2], (uint32_t)pr[0], HI); o.r13 = jm_perm((uint32_t)pr[3], (uint32_t)pr[1], HI);
      o.b02 = jm_perm((uint32_t)pb[2], (uint32_t)pb[0], HI); o.b13 = jm_perm((uint32_t)pb[3], (uint32_t)pb[1], HI);
      o.g02 = jm_pk_ashr16_7(jm_perm((uint32_t)pg[2], (uint32_t)pg[0], HI));
      o.g13 = jm_pk_ashr16_7(jm_perm((uint32_t)pg[3], (uint32_t)pg[1], HI));
   } else {
      o.r02 = o.r13 = jm_perm((uint32_t)pr[1], (uint32_t)pr[0], HI);
      o.b02 = o.b13 = jm_perm((uint32_t)pb[1], (uint32_t)pb[0], HI);
      o.g02 = o.g13 = jm_pk_ashr16_7(jm_perm((uint32_t)pg[1], (uint32_t)pg[0], HI));
   }
   return o;
}
// yw = four luma bytes -> the 12 output bytes
JM_HD void rgb_quad_pk(uint32_t yw, const ChromaPairs &o, uint32_t (&w)[3]) {
   const uint32_t y02 = yw & 0x00FF00FFu;
   const uint32_t y13 = jm_perm(0u, yw, 0x0c030c01u);
   const uint32_t rg = jm_sat2x2(jm_pk_add16(y02, o.r02), jm_pk_add16(y02, o.g02));   // r0 r2 g0 g2
   const uint32_t br = jm_sat2x2(jm_pk_add16(y02, o.b02), jm_pk_add16(y13, o.r13));   // b0 b2 r1 r3
   const uint32_t gb = jm_sat2x2(jm_pk_add16(y13, o.g13), jm_pk_add16(y13, o.b13));   // g1 g3 b1 b3
   w[0] = jm_perm(br, rg, 0x06040200u);   // r0 g0 b0 r1
   w[1] = jm_perm(rg, gb, 0x07050200u);   // g1 b1 r2 g2
   w[2] = jm_perm(gb, br, 0x07050301u);   // b2 r3 g3 b3
}

}  // namespace jpegmath

/*
 * explore_exact.c -- exhaustive search/verification of cheap fp32 sequences that reproduce the
 * reference's mixed fp32/fp64 roundings bit for bit (see DESIGN.md "exact arithmetic").
 *
 * Every identity the HIP kernel relies on is checked here over ALL 2^24 float mantissa/sign
 * patterns (the identities are exact under power-of-two scaling, so one binade suffices; a few
 * other binades are swept as a sanity check).
 *
 *   gcc -O2 -ffp-contract=off -o explore_exact explore_exact.c -lquadmath -lm
 */
#include <math.h>
#include <quadmath.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static const double SQRT_2 = 1.4142135623730950488;
static const double SQRT_8 = 2.8284271247461900976;

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

typedef float (*fn1)(float);

/* ---- reference semantics ---- */
static float ref_scale(float c)  { return (float)((double)c * SQRT_8); }          /* idct_opt.c:78-80 */
static float ref_shrink(float x) { return (float)((double)x / SQRT_2); }          /* idct_opt.c:45-47 */
static float ref_divk(float x)   { float k = (float)SQRT_2; return x / k; }       /* idct_opt.c:41-42, k float */

/* ---- candidates ---- */
static float Sh, Sl, Rh, Rl, Kh, Kl, Kinv;
static float cand_scale2(float c)  { return fmaf(c, Sh, c * Sl); }
static float cand_shrink2(float x) { return fmaf(x, Rh, x * Rl); }
static float cand_divk2(float x)   { return fmaf(x, Kh, x * Kl); }
static float cand_divk3(float x)   { float k = (float)SQRT_2; float q = x * Kinv; float r = fmaf(-q, k, x); return fmaf(r, Kinv, q); }
/* 4-op error-free variants */
static float cand_scale4(float c)  { float p = c * Sh; float e = fmaf(c, Sh, -p); float t = fmaf(c, Sl, e); return p + t; }
static float cand_shrink4(float x) { float p = x * Rh; float e = fmaf(x, Rh, -p); float t = fmaf(x, Rl, e); return p + t; }

static uint64_t sweep(const char *name, fn1 ref, fn1 cand, int expo, int verbose) {
   uint64_t bad = 0;
   for (uint32_t s = 0; s < 2; s++)
      for (uint32_t m = 0; m < (1u << 23); m++) {
         float x = u2f((s << 31) | ((uint32_t)expo << 23) | m);
         float a = ref(x), b = cand(x);
         if (f2u(a) != f2u(b)) {
            if (verbose && bad < 8) printf("   %s mismatch: x=%a (mant 0x%06x) ref=%a cand=%a\n", name, x, m, a, b);
            bad++;
         }
      }
   printf("%-14s expo %3d: %llu mismatches of 2^24\n", name, expo, (unsigned long long)bad);
   return bad;
}

static void split(__float128 v, float *hi, float *lo, int bump) {
   float h = (float)v;
   for (int i = 0; i < (bump < 0 ? -bump : bump); i++) h = nextafterf(h, bump > 0 ? INFINITY : -INFINITY);
   *hi = h;
   *lo = (float)(v - (__float128)h);
}

int main(void) {
   __float128 S = (__float128)SQRT_8, R = (__float128)1 / (__float128)SQRT_2, K = (__float128)1 / (__float128)(float)SQRT_2;
   Kinv = (float)K;
   uint64_t total = 0;
   /* try the canonical split and a few bumped ones, report which are exact */
   for (int bump = -3; bump <= 3; bump++) {
      split(S, &Sh, &Sl, bump); split(R, &Rh, &Rl, bump); split(K, &Kh, &Kl, bump);
      printf("== bump %d: Sh=%a Sl=%a | Rh=%a Rl=%a | Kh=%a Kl=%a\n", bump, Sh, Sl, Rh, Rl, Kh, Kl);
      sweep("scale2", ref_scale, cand_scale2, 127, 1);
      sweep("shrink2", ref_shrink, cand_shrink2, 127, 1);
      sweep("divk2", ref_divk, cand_divk2, 127, 1);
   }
   split(S, &Sh, &Sl, 0); split(R, &Rh, &Rl, 0); split(K, &Kh, &Kl, 0);
   total += sweep("divk3", ref_divk, cand_divk3, 127, 1);
   total += sweep("scale4", ref_scale, cand_scale4, 127, 1);
   total += sweep("shrink4", ref_shrink, cand_shrink4, 127, 1);
   return total != 0;
}

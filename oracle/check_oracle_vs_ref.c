/*
 * check_oracle_vs_ref.c -- TEST INFRASTRUCTURE ONLY.
 * Pins the CPU restatement (liboracle.so) against the REAL reference objects
 * (_ref/libjpegref_strict.so), bit for bit.  Runs in the build container only.
 *   usage: check_oracle_vs_ref [n_million_blocks=4]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "jpeg_oracle.h"

/* from ref_driver.c (calls the reference's own functions) */
int  ref_backend_run(const jpeggpu_frame_desc_t *, const jpeggpu_blocl16_t *const[3], const jpeggpu_qtable_t *const[4], uint8_t *, size_t);
void ref_iquant_blocks(jpeggpu_blocl16_t *, size_t, int, int, const jpeggpu_qtable_t *);
void ref_izz_blocks(const jpeggpu_blocl16_t *, jpeggpu_bloct16_t *, size_t);
void ref_idct_opt_blocks(const jpeggpu_bloct16_t *, jpeggpu_bloctu8_t *, size_t);
void ref_ycc2rgb_pixels(const uint8_t *, const uint8_t *, const uint8_t *, jpeggpu_rgb_t *, size_t);
void ref_idct_blocks(const jpeggpu_bloct16_t *, jpeggpu_bloctu8_t *, size_t);
void ref_idct_coef(float *);

static int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { fails++; printf("FAILED: " __VA_ARGS__); printf("\n"); } } while (0)

static void check_stage_blocks(size_t n, uint64_t seed, int kind, const jpeggpu_qtable_t *q) {
   jpeggpu_blocl16_t *a = malloc(n * sizeof *a), *b = malloc(n * sizeof *b);
   jpeggpu_bloct16_t *za = malloc(n * sizeof *za), *zb = malloc(n * sizeof *zb);
   jpeggpu_bloctu8_t *pa = malloc(n * sizeof *pa), *pb = malloc(n * sizeof *pb);
   oracle_synth_blocks(a, n, seed, 0, kind);
   memcpy(b, a, n * sizeof *a);
   oracle_iquant_blocks(a, n, 0, 63, q); ref_iquant_blocks(b, n, 0, 63, q);
   CHECK(!memcmp(a, b, n * sizeof *a), "iquant kind %d", kind);
   oracle_izz_blocks(a, za, n); ref_izz_blocks(b, zb, n);
   CHECK(!memcmp(za, zb, n * sizeof *za), "izz kind %d", kind);
   oracle_idct_opt_blocks(za, pa, n); ref_idct_opt_blocks(zb, pb, n);
   size_t bad = 0;
   for (size_t i = 0; i < n; i++) bad += memcmp(&pa[i], &pb[i], 64) != 0;
   CHECK(bad == 0, "idct_opt kind %d: %zu of %zu blocks differ", kind, bad, n);
   printf("stage chain kind %d: %zu blocks, %zu differing\n", kind, n, bad);
   /* the slow transform (src/idct.c) on a sixteenth of the blocks (it is 8x slower) */
   size_t ns = n / 16, bads = 0;
   oracle_idct_blocks(za, pa, ns); ref_idct_blocks(zb, pb, ns);
   for (size_t i = 0; i < ns; i++) bads += memcmp(&pa[i], &pb[i], 64) != 0;
   CHECK(bads == 0, "idct (slow) kind %d: %zu of %zu blocks differ", kind, bads, ns);
   printf("slow idct kind %d: %zu blocks, %zu differing\n", kind, ns, bads);
   free(a); free(b); free(za); free(zb); free(pa); free(pb);
}

static void check_frame(int w, int h, int nc, const int hs[3], const int vs[3], int apply_q, uint64_t seed, int kind,
                        const jpeggpu_qtable_t *const qt[4]) {
   jpeggpu_frame_desc_t d; memset(&d, 0, sizeof d);
   d.width = (uint16_t)w; d.height = (uint16_t)h; d.nbcomp = (uint8_t)nc; d.apply_iquant = (uint8_t)apply_q;
   for (int c = 0; c < nc; c++) { d.comp[c].hs = (uint8_t)hs[c]; d.comp[c].vs = (uint8_t)vs[c]; d.comp[c].idq = (uint8_t)(c ? 1 : 0); }
   oracle_frame_geometry(&d);
   jpeggpu_blocl16_t *pl[3] = { 0, 0, 0 };
   for (int c = 0; c < nc; c++) {
      size_t nb = (size_t)d.nbmcuH * d.comp[c].hs * d.nbmcuV * d.comp[c].vs;
      pl[c] = malloc(nb * sizeof(jpeggpu_blocl16_t));
      oracle_synth_blocks(pl[c], nb, seed + (uint64_t)c, 0, kind);
   }
   size_t rows = (size_t)d.nbmcuV * d.max_v * 8, rb = (size_t)w * (nc == 3 ? 3 : 1);
   uint8_t *oa = calloc(rows, rb), *ob = calloc(rows, rb);
   int ra = oracle_backend_run(&d, (const jpeggpu_blocl16_t *const *)pl, qt, oa, 0);
   int rr = ref_backend_run(&d, (const jpeggpu_blocl16_t *const *)pl, qt, ob, 0);
   CHECK(ra == 0 && rr == 0, "frame rc %d %d", ra, rr);
   CHECK(!memcmp(oa, ob, rows * rb), "frame %dx%d nc=%d hs=%d,%d,%d vs=%d,%d,%d", w, h, nc, hs[0], hs[1], hs[2], vs[0], vs[1], vs[2]);
   printf("frame %4dx%-4d nc=%d Y%dx%d Cb%dx%d Cr%dx%d q=%d kind=%d: %s\n", w, h, nc, hs[0], vs[0], hs[1], vs[1], hs[2], vs[2], apply_q, kind,
          memcmp(oa, ob, rows * rb) ? "DIFF" : "identical");
   for (int c = 0; c < nc; c++) free(pl[c]);
   free(oa); free(ob);
}

static void check_coef_table(void) {
   static float a[8][8][8][8], b[4096];
   oracle_idct_coef(a); ref_idct_coef(b);
   CHECK(!memcmp(a, b, sizeof b), "calc_coef table differs");
   printf("calc_coef table: %s\n", memcmp(a, b, sizeof b) ? "DIFFERENT" : "identical (4096 floats)");
}

int main(int argc, char **argv) {
   check_coef_table();
   size_t nmil = argc > 1 ? (size_t)atoi(argv[1]) : 4;
   /* quantisation tables: a JPEG-like ramp and a full-range one (exercises the int16 wrap) */
   jpeggpu_qtable_t qa, qb, qc;
   for (int i = 0; i < 64; i++) { qa.data[i] = (uint16_t)(2 + i); qb.data[i] = (uint16_t)(3 + 2 * i); qc.data[i] = (uint16_t)(40503u * (unsigned)(i + 1)); }
   const jpeggpu_qtable_t *qt[4] = { &qa, &qb, &qc, NULL };

   /* zigzag known-answer: impulses */
   for (int k = 0; k < 64; k++) {
      jpeggpu_blocl16_t z; memset(&z, 0, sizeof z); z.data[k] = (int16_t)(k + 1);
      jpeggpu_bloct16_t a, b; oracle_izz_blocks(&z, &a, 1); ref_izz_blocks(&z, &b, 1);
      CHECK(!memcmp(&a, &b, sizeof a), "izz impulse %d", k);
   }
   /* stage chain on random blocks */
   check_stage_blocks(nmil * 500000, 0x5EED0001, 0, &qa);
   check_stage_blocks(nmil * 250000, 0x5EED0002, 1, &qa);
   check_stage_blocks(nmil * 250000, 0x5EED0003, 2, &qc);

   /* ycc2rgb: every input */
   {
      size_t n = 1u << 24; size_t bad = 0;
      uint8_t *y = malloc(n), *cb = malloc(n), *cr = malloc(n);
      jpeggpu_rgb_t *a = malloc(n * 3), *b = malloc(n * 3);
      for (size_t i = 0; i < n; i++) { y[i] = (uint8_t)i; cb[i] = (uint8_t)(i >> 8); cr[i] = (uint8_t)(i >> 16); }
      oracle_ycc2rgb_pixels(y, cb, cr, a, n); ref_ycc2rgb_pixels(y, cb, cr, b, n);
      for (size_t i = 0; i < n; i++) bad += memcmp(&a[i], &b[i], 3) != 0;
      CHECK(bad == 0, "ycc2rgb: %zu of 2^24 differ", bad);
      printf("ycc2rgb: 2^24 inputs, %zu differing\n", bad);
      free(y); free(cb); free(cr); free(a); free(b);
   }

   /* frames: the sampling combinations of test/upsampler_test.c:29-42 that satisfy
      max % factor == 0, ragged sizes, with and without iquant */
   static const int combos[][6] = {   /* hsY hsCb hsCr vsY vsCb vsCr */
      {1,1,1, 1,1,1}, {2,1,1, 1,1,1}, {2,1,1, 2,1,1}, {1,1,1, 2,1,1}, {2,2,1, 2,1,1}, {2,1,2, 2,2,1},
      {4,1,1, 1,1,1}, {4,2,2, 1,1,1}, {1,1,1, 4,1,1}, {4,2,1, 2,1,2}, {2,2,2, 2,2,2}, {4,4,2, 2,2,1},
   };
   for (size_t i = 0; i < sizeof combos / sizeof combos[0]; i++) {
      const int *k = combos[i];
      int hs[3] = { k[0], k[1], k[2] }, vs[3] = { k[3], k[4], k[5] };
      check_frame(67, 45, 3, hs, vs, 1, 100 + i, 0, qt);
      check_frame(160, 32, 3, hs, vs, 0, 200 + i, 1, qt);
   }
   { int one[3] = { 1, 1, 1 };
     check_frame(8, 8, 1, one, one, 1, 7, 0, qt);
     check_frame(61, 37, 1, one, one, 1, 8, 2, qt);
     check_frame(300, 225, 1, one, one, 0, 9, 0, qt);
     /* gray with sampling factors above 1: save_mcu_ligne_bw only ever reads the first block row of an MCU row
        (src/decoder_utils.c:55-58), so rows 8.. of each band repeat rows 0..7 */
     int g22h[3] = { 2, 1, 1 }, g22v[3] = { 2, 1, 1 }, g12v[3] = { 2, 1, 1 }, g41h[3] = { 4, 1, 1 }, g14v[3] = { 4, 1, 1 };
     check_frame(61, 37, 1, g22h, g22v, 1, 10, 0, qt);
     check_frame(130, 70, 1, one, g12v, 1, 11, 1, qt);
     check_frame(97, 20, 1, g41h, one, 0, 12, 0, qt);
     check_frame(40, 90, 1, one, g14v, 1, 13, 0, qt);
     check_frame(200, 64, 1, g41h, g14v, 1, 14, 2, qt); }

   printf(fails ? "check_oracle_vs_ref: %d FAILURES\n" : "check_oracle_vs_ref: ALL IDENTICAL (%d failures)\n", fails);
   return fails != 0;
}

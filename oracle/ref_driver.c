/*
 * ref_driver.c -- TEST INFRASTRUCTURE ONLY.  Harness around the REAL reference objects.
 *
 * Compiled (by oracle/Makefile, in the build container only) together with the reference's own
 * unmodified sources where they lie under /root/reference, into oracle/_ref/libjpegref_*.so.
 * Nothing here restates reference arithmetic: every pixel comes out of the reference's iquant,
 * izz, idct_opt, save_mcu_ligne_bw / save_mcu_ligne_color and (for real files) decode_entete +
 * decode_bloc_acdc.  The harness only lays out buffers the way src/progressive.c:257-312 does.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* reference headers (-I/root/reference/include at build time) */
#include <jpeg2ppm.h>
#include <img.h>
#include <iqzz.h>
#include <idct_opt.h>
#include <idct.h>
#include <ycc2rgb.h>
#include <decoder_utils.h>
#include <entete.h>
#include <vld.h>
#include <bitstream.h>
#include <options.h>

#include "../include/jpeggpu.h"

extern all_option_t all_option;

/* all_option.idct_fast for the harness (src/baseline.c:190-191, src/progressive.c:284-285) */
static int g_ref_idct_fast = 1;
static float g_ref_coef[8][8][8][8];
static int g_ref_coef_ready = 0;
void ref_set_idct_fast(int fast) { g_ref_idct_fast = fast != 0; }
void ref_idct_coef(float *out4096) { calc_coef((float(*)[8][8][8])out4096); }
static bloctu8_t *ref_transform(bloct16_t *zz) {
   if (g_ref_idct_fast) return idct_opt(zz);
   if (!g_ref_coef_ready) { calc_coef(g_ref_coef); g_ref_coef_ready = 1; }
   return idct(zz, g_ref_coef);
}
void ref_idct_blocks(const jpeggpu_bloct16_t *in, jpeggpu_bloctu8_t *out, size_t n) {
   if (!g_ref_coef_ready) { calc_coef(g_ref_coef); g_ref_coef_ready = 1; }
   for (size_t i = 0; i < n; i++) { bloctu8_t *r = idct((bloct16_t *)&in[i], g_ref_coef); memcpy(&out[i], r, sizeof *r); free(r); }
}

/* Reference back end over full-frame planes; same contract as jpeggpu_backend_run.
 * MCU rows [m0, m1).  Mirrors the loop structure of src/progressive.c:257-312.                 */
int ref_backend_rows(const jpeggpu_frame_desc_t *d, const jpeggpu_blocl16_t *const coef[3],
                     const jpeggpu_qtable_t *const qtables[4], uint8_t *out, size_t out_stride,
                     uint32_t m0, uint32_t m1) {
   const int nc = d->nbcomp;
   if (nc != 1 && nc != 3) return JPEGGPU_ERR_NB_COMP;
   const size_t row_bytes = (size_t)d->width * (nc == 3 ? 3 : 1);
   if (out_stride == 0) out_stride = row_bytes;
   const size_t band_rows = (size_t)d->max_v * 8;

   img_t *img = init_img();
   img->width = d->width; img->height = d->height;
   img->max_hsampling = d->max_h; img->max_vsampling = d->max_v;
   img->nbmcuH = d->nbmcuH; img->nbmcuV = d->nbmcuV; img->nbMCU = (uint64_t)d->nbmcuH * d->nbmcuV;
   img->comps->nb = (uint8_t)nc;
   for (int c = 0; c < nc; c++) {
      img->comps->comps[c] = (idcomp_t *)calloc(1, sizeof(idcomp_t));
      img->comps->comps[c]->idc = (uint8_t)(c + 1);
      img->comps->ordre[c] = (uint8_t)(c + 1);
      img->comps->comps[c]->hsampling = d->comp[c].hs;
      img->comps->comps[c]->vsampling = d->comp[c].vs;
      img->comps->comps[c]->idq = d->comp[c].idq;
   }
   uint8_t y_id = 0, cb_id = 0, cr_id = 0, yhf = 0, yvf = 0, cbhf = 0, cbvf = 0, crhf = 0, crvf = 0;
   uint64_t nbY = 0, nbCb = 0, nbCr = 0;
   char *rgb = NULL;
   if (nc == 3) get_ycc_info(img, &y_id, &cb_id, &cr_id, &yhf, &yvf, &cbhf, &cbvf, &crhf, &crvf, &nbY, &nbCb, &nbCr, &rgb);

   bloctu8_t ***ycc = (bloctu8_t ***)malloc(sizeof(bloctu8_t **) * nc);
   for (int c = 0; c < nc; c++) ycc[c] = (bloctu8_t **)calloc((size_t)d->nbmcuH * d->comp[c].hs * d->comp[c].vs, sizeof(bloctu8_t *));

   char *band = NULL; size_t band_len = 0;
   int rc = 0;
   for (uint32_t m = m0; m < m1 && rc == 0; m++) {
      for (int c = 0; c < nc; c++) {
         size_t nbH = (size_t)d->nbmcuH * d->comp[c].hs;
         for (int by = 0; by < d->comp[c].vs; by++) {
            for (size_t bx = 0; bx < nbH; bx++) {
               blocl16_t q;
               memcpy(q.data, coef[c][((size_t)m * d->comp[c].vs + by) * nbH + bx].data, sizeof q.data);
               if (d->apply_iquant) iquant(&q, 0, 63, (qtable_t *)qtables[d->comp[c].idq]);
               bloct16_t *zz = izz(&q);
               bloctu8_t *px = ref_transform(zz);
               free(zz);
               free(ycc[c][by * nbH + bx]);
               ycc[c][by * nbH + bx] = px;
            }
         }
      }
      FILE *f = open_memstream(&band, &band_len);
      erreur_t e;
      if (nc == 1) e = save_mcu_ligne_bw(f, img, ycc);
      else e = save_mcu_ligne_color(f, img, ycc, rgb, yhf, yvf, y_id, nbY, cbhf, cbvf, cb_id, nbCb, crhf, crvf, cr_id, nbCr);
      fclose(f);
      if (e.code) rc = (int)e.code;
      else if (band_len != band_rows * row_bytes) rc = -100;
      else for (size_t r = 0; r < band_rows; r++)
         memcpy(out + ((size_t)m * band_rows + r) * out_stride, band + r * row_bytes, row_bytes);
      free(band); band = NULL;
   }
   for (int c = 0; c < nc; c++) {
      size_t n = (size_t)d->nbmcuH * d->comp[c].hs * d->comp[c].vs;
      for (size_t j = 0; j < n; j++) free(ycc[c][j]);
      free(ycc[c]);
   }
   free(ycc);
   if (nc == 3) free(rgb);
   free_img(img);
   return rc;
}

int ref_backend_run(const jpeggpu_frame_desc_t *d, const jpeggpu_blocl16_t *const coef[3],
                    const jpeggpu_qtable_t *const qtables[4], uint8_t *out, size_t out_stride) {
   return ref_backend_rows(d, coef, qtables, out, out_stride, 0, d->nbmcuV);
}

/* stage-level pass-throughs to the reference functions (array forms) */
void ref_iquant_blocks(jpeggpu_blocl16_t *b, size_t n, int s0, int s1, const jpeggpu_qtable_t *q) {
   for (size_t i = 0; i < n; i++) iquant((blocl16_t *)&b[i], (uint8_t)s0, (uint8_t)s1, (qtable_t *)q);
}
void ref_izz_blocks(const jpeggpu_blocl16_t *in, jpeggpu_bloct16_t *out, size_t n) {
   for (size_t i = 0; i < n; i++) { bloct16_t *r = izz((blocl16_t *)&in[i]); memcpy(&out[i], r, sizeof *r); free(r); }
}
void ref_idct_opt_blocks(const jpeggpu_bloct16_t *in, jpeggpu_bloctu8_t *out, size_t n) {
   for (size_t i = 0; i < n; i++) { bloctu8_t *r = idct_opt((bloct16_t *)&in[i]); memcpy(&out[i], r, sizeof *r); free(r); }
}
void ref_ycc2rgb_pixels(const uint8_t *y, const uint8_t *cb, const uint8_t *cr, jpeggpu_rgb_t *o, size_t n) {
   for (size_t i = 0; i < n; i++) { rgb_t p; ycc2rgb_pixel(y[i], cb[i], cr[i], &p); o[i].r = p.r; o[i].g = p.g; o[i].b = p.b; }
}

/* Decode the entropy-coded data of a BASELINE file with the reference's own header parser and
 * VLD (decode_entete, decode_bloc_acdc) and return full-frame QUANTISED zigzag planes, laid out
 * like init_sortie (src/progressive.c:73-84) but contiguous.  The MCU walk follows
 * src/baseline.c:160-199 minus the pixel stages.  Caller frees planes[c].                       */
int ref_dump_baseline_planes(const char *path, jpeggpu_frame_desc_t *d, jpeggpu_blocl16_t *planes[3],
                             jpeggpu_qtable_t qt[4], int qt_present[4]) {
   memset(&all_option, 0, sizeof all_option);
   all_option.idct_fast = 1;
   FILE *f = fopen(path, "rb");
   if (!f) return -1;
   img_t *img = init_img();
   erreur_t e = decode_entete(f, true, img);
   if (e.code) { fclose(f); free_img(img); return (int)e.code; }
   if (img->section->num_sof != 0) { fclose(f); free_img(img); return -2; }
   memset(d, 0, sizeof *d);
   d->width = img->width; d->height = img->height; d->nbcomp = img->comps->nb;
   d->max_h = img->max_hsampling; d->max_v = img->max_vsampling;
   d->nbmcuH = (uint32_t)img->nbmcuH; d->nbmcuV = (uint32_t)img->nbmcuV; d->apply_iquant = 1;
   for (int c = 0; c < d->nbcomp; c++) {
      d->comp[c].hs = img->comps->comps[c]->hsampling;
      d->comp[c].vs = img->comps->comps[c]->vsampling;
      d->comp[c].idq = img->comps->comps[c]->idq;
      size_t nb = (size_t)d->nbmcuH * d->comp[c].hs * d->nbmcuV * d->comp[c].vs;
      planes[c] = (jpeggpu_blocl16_t *)calloc(nb, sizeof(jpeggpu_blocl16_t));
   }
   for (int i = 0; i < 4; i++) {
      qt_present[i] = img->qtables[i] != NULL;
      if (qt_present[i]) memcpy(qt[i].data, img->qtables[i]->qtable->data, sizeof qt[i].data);
   }
   int16_t dc_prec[3] = { 0, 0, 0 };
   bitstream_t bs; init_bitstream(&bs, f);
   int rc = 0;
   for (uint64_t i = 0; i < img->nbMCU && rc == 0; i++) {
      uint64_t mx = i % img->nbmcuH, my = i / img->nbmcuH;
      for (int k = 0; k < d->nbcomp && rc == 0; k++) {
         int c = get_composante(img, (uint8_t)k);
         if (c < 0) break;
         idcomp_t *ci = img->comps->comps[c];
         size_t nbH = (size_t)img->nbmcuH * ci->hsampling;
         for (int by = 0; by < ci->vsampling && rc == 0; by++)
            for (int bx = 0; bx < ci->hsampling && rc == 0; bx++) {
               blocl16_t blk; memset(&blk, 0, sizeof blk);
               uint16_t skip = 0;
               e = decode_bloc_acdc(&bs, img, img->htables->dc[ci->idhdc], img->htables->ac[ci->idhac], &blk, dc_prec + c, &skip);
               if (e.code) { rc = (int)e.code; break; }
               memcpy(planes[c][(my * ci->vsampling + by) * nbH + mx * ci->hsampling + bx].data, blk.data, sizeof blk.data);
            }
      }
   }
   fclose(f);
   free_img(img);
   return rc;
}

/*
 * baseline_jpeggpu.c -- the same binding for src/baseline.c.  The reference streams block by block (VLD + iquant ->
 * izz -> idct_opt -> one MCU row of ycc -> save_mcu_ligne_*, :160-217); here every block, already dequantised by the
 * reference's own decode_bloc_baseline (:31-69, which calls iquant at :66), is stored at its frame position
 * (mcuY*vs + by)*nbH + mcuX*hs + bx of a contiguous plane and the whole frame goes through ONE jpeggpu_backend_run
 * with apply_iquant = 0.  decode_bloc_baseline is used unmodified: this file #includes the reference's baseline.c
 * (build container only) with its driver renamed out of the way.  A scan that fails half-way still leaves the header
 * and the complete MCU rows in the file, as the reference does (:176-179, :201-216).
 */
#define decode_baseline_image decode_baseline_image_cpu
#include <baseline.c>
#undef decode_baseline_image

#include <string.h>
#include <jpeggpu.h>
#include "jpeggpu_erreur.h"

erreur_t decode_baseline_image(FILE *infile, img_t *img) {
   if (all_option.print_tables) print_huffman_quant_table(img);
   const uint8_t nbcomp = img->comps->nb;
   jpeggpu_set_idct_fast(all_option.idct_fast ? 1 : 0);

   FILE *outputfile = ouverture_fichier_out(nbcomp, 0);
   if (nbcomp == 1) fprintf(outputfile, "P5\n");
   else if (nbcomp == 3) fprintf(outputfile, "P6\n");
   else { fclose(outputfile); return (erreur_t) {.code = ERR_NB_COMP, .com = "Il faut une ou trois composante", .must_free = false}; }
   fprintf(outputfile, "%d %d\n", img->width, img->height);
   fprintf(outputfile, "255\n");

   jpeggpu_frame_desc_t d;
   memset(&d, 0, sizeof d);
   d.width = img->width; d.height = img->height; d.nbcomp = nbcomp;
   d.max_h = img->max_hsampling; d.max_v = img->max_vsampling;
   d.apply_iquant = 0;                       /* iquant already ran inside decode_bloc_baseline (:66) */
   d.nbmcuH = (uint32_t)img->nbmcuH; d.nbmcuV = (uint32_t)img->nbmcuV;
   jpeggpu_blocl16_t *plane[3] = { NULL, NULL, NULL };
   erreur_t err = {.code = SUCCESS};
   for (uint8_t k = 0; k < nbcomp; k++) {
      d.comp[k].hs = img->comps->comps[k]->hsampling;
      d.comp[k].vs = img->comps->comps[k]->vsampling;
      d.comp[k].idq = img->comps->comps[k]->idq;
      size_t n = jpeggpu_plane_blocks(&d, k) * sizeof(jpeggpu_blocl16_t);
      plane[k] = (jpeggpu_blocl16_t *)jpeggpu_host_alloc(n);
      if (plane[k] == NULL) { err = jpeggpu_erreur(JPEGGPU_ERR_HIP); break; }
      memset(plane[k], 0, n);
   }
   const size_t rb = jpeggpu_raster_row_bytes(&d);
   uint8_t *raster = err.code ? NULL : (uint8_t *)jpeggpu_host_alloc(jpeggpu_raster_rows(&d) * rb);
   if (!err.code && raster == NULL) err = jpeggpu_erreur(JPEGGPU_ERR_HIP);

   int16_t *dc_prec = (int16_t *)calloc(nbcomp, sizeof(int16_t));
   bitstream_t *bs = (bitstream_t *)malloc(sizeof(bitstream_t));
   init_bitstream(bs, infile);
   uint64_t rows_done = 0;                   /* complete MCU rows decoded */
   for (uint64_t i = 0; i < img->nbMCU && !err.code; i++) {
      const uint64_t mcuX = i % img->nbmcuH, mcuY = i / img->nbmcuH;
      for (uint8_t k = 0; k < nbcomp && !err.code; k++) {
         int16_t indice_comp = get_composante(img, k);
         if (indice_comp == -1) break;
         const uint8_t hs = img->comps->comps[indice_comp]->hsampling, vs = img->comps->comps[indice_comp]->vsampling;
         const uint64_t nbH = img->nbmcuH * hs;
         for (uint8_t by = 0; by < vs && !err.code; by++)
            for (uint8_t bx = 0; bx < hs && !err.code; bx++) {
               /* the reference indexes ycc by scan position k (:195-196); frame position == scan position for every
                  file whose SOS lists the components in SOF order, the only order get_ycc_info is coherent for */
               blocl16_t *bloc = (blocl16_t *)&plane[k][(mcuY * vs + by) * nbH + mcuX * hs + bx];
               memset(bloc, 0, sizeof *bloc);
               err = decode_bloc_baseline(bs, img, indice_comp, bloc, dc_prec);
            }
      }
      if (!err.code && mcuX == img->nbmcuH - 1) rows_done = mcuY + 1;
   }
   erreur_t err_end = {.code = SUCCESS};
   if (!err.code) err_end = finir_octet(bs);
   free(bs);
   free(dc_prec);

   /* ONE call for every complete MCU row */
   if (rows_done > 0 && raster != NULL) {
      d.nbmcuV = (uint32_t)rows_done;
      int rc = jpeggpu_backend_run(&d, (const jpeggpu_blocl16_t *const *)plane, NULL, raster, 0);
      if (rc) { if (!err.code) err = jpeggpu_erreur(rc); }
      else fwrite(raster, 1, jpeggpu_raster_rows(&d) * rb, outputfile);
   }
   fclose(outputfile);
   for (uint8_t k = 0; k < 3; k++) jpeggpu_host_free(plane[k]);
   jpeggpu_host_free(raster);
   if (err.code) return err;
   return err_end;
}

/*
 * progressive_jpeggpu.c -- the binding a maintainer of beldaphilippe/jpeg-decoder adds to src/progressive.c to run the
 * per-scan pixel back end on an MI355X, as a file that compiles and runs.
 *
 * Two changes against src/progressive.c:
 *   1. init_sortie (:73-84) hands out the same blocl16_t* per block, but the blocks of a component lie in ONE pinned
 *      slab, block (blocX, blocY) at slab + blocY*nbH + blocX -- the scan decoders index it exactly as before
 *      (:118, :157, :163);
 *   2. the block loop and the writers (:246-315: iquant on a copy, izz, idct_opt, save_mcu_ligne_*) become ONE
 *      jpeggpu_backend_run per scan plus one fwrite.
 * Everything else is the reference's own code, compiled UNMODIFIED from where it lies: this file #includes the
 * reference's progressive.c (build container only, -I$(REF)/src) with its driver renamed out of the way, which
 * gives it the static scan decoders decode_progressif_dc / decode_progressif_ac (:98-170).  Nothing is copied.
 */
#define decode_progressive_image decode_progressive_image_cpu
#include <progressive.c>
#undef decode_progressive_image

#include <jpeggpu.h>
#include "jpeggpu_erreur.h"

/* change 1: contiguous, pinned coefficient planes behind the reference's blocl16_t*** */
static blocl16_t ***init_sortie_contigu(img_t *img, jpeggpu_blocl16_t *slab[3]) {
   blocl16_t ***sortie = (blocl16_t ***)calloc(img->comps->nb, sizeof(blocl16_t **));
   for (uint8_t i = 0; i < img->comps->nb && i < 3; i++) {
      uint64_t n = img->nbmcuH * img->comps->comps[i]->hsampling * img->nbmcuV * img->comps->comps[i]->vsampling;
      slab[i] = (jpeggpu_blocl16_t *)jpeggpu_host_alloc(n * sizeof(jpeggpu_blocl16_t));
      if (slab[i] == NULL) return sortie;
      memset(slab[i], 0, n * sizeof(jpeggpu_blocl16_t));
      sortie[i] = (blocl16_t **)malloc(n * sizeof(blocl16_t *));
      for (uint64_t j = 0; j < n; j++) sortie[i][j] = (blocl16_t *)&slab[i][j];   /* same 128-byte record, include/jpeg2ppm.h:31-34 */
   }
   return sortie;
}

static void free_sortie_contigu(img_t *img, blocl16_t ***sortie, jpeggpu_blocl16_t *slab[3], uint8_t *raster) {
   for (uint8_t i = 0; i < img->comps->nb && i < 3; i++) {
      free(sortie[i]);
      jpeggpu_host_free(slab[i]);
   }
   free(sortie);
   jpeggpu_host_free(raster);
}

erreur_t decode_progressive_image(FILE *infile, img_t *img) {
   const uint8_t nbcomp = img->comps->nb;
   if (nbcomp != 1 && nbcomp != 3) return (erreur_t) {.code = ERR_NB_COMP, .com = "Il faut une ou trois composante", .must_free = false};
   jpeggpu_blocl16_t *slab[3] = { NULL, NULL, NULL };
   blocl16_t ***sortie = init_sortie_contigu(img, slab);
   for (uint8_t i = 0; i < nbcomp; i++)
      if (slab[i] == NULL) { erreur_t e = jpeggpu_erreur(JPEGGPU_ERR_HIP); free_sortie_contigu(img, sortie, slab, NULL); return e; }
   jpeggpu_set_idct_fast(all_option.idct_fast ? 1 : 0);          /* "-f", src/options.c:76-79 */

   /* what L2 has in hand at src/progressive.c:257 */
   jpeggpu_frame_desc_t d;
   memset(&d, 0, sizeof d);
   d.width = img->width; d.height = img->height; d.nbcomp = nbcomp;
   d.max_h = img->max_hsampling; d.max_v = img->max_vsampling;
   d.apply_iquant = 1;                                           /* sortie holds QUANTISED coefficients (:277-279) */
   d.nbmcuH = (uint32_t)img->nbmcuH; d.nbmcuV = (uint32_t)img->nbmcuV;
   for (uint8_t k = 0; k < nbcomp; k++) {
      d.comp[k].hs = img->comps->comps[k]->hsampling;
      d.comp[k].vs = img->comps->comps[k]->vsampling;
      d.comp[k].idq = img->comps->comps[k]->idq;
   }
   const size_t rows = jpeggpu_raster_rows(&d), rb = jpeggpu_raster_row_bytes(&d);
   uint8_t *raster = (uint8_t *)jpeggpu_host_alloc(rows * rb);
   if (raster == NULL) { erreur_t e = jpeggpu_erreur(JPEGGPU_ERR_HIP); free_sortie_contigu(img, sortie, slab, NULL); return e; }

   uint64_t nb_passage_sos = 1;
   my_timer_t timer_image, timer_ecriture;
   while (!img->section->eoi_done) {
      init_timer(&timer_image);
      start_timer(&timer_image);
      init_timer(&timer_ecriture);

      /* the reference's scan decoding, unchanged (:215-236) */
      bitstream_t *bs = (bitstream_t *)malloc(sizeof(bitstream_t));
      init_bitstream(bs, infile);
      erreur_t err;
      if (img->other->se == 0) err = decode_progressif_dc(bs, img, sortie);
      else err = decode_progressif_ac(bs, img, sortie);
      if (!err.code) err = finir_octet(bs);
      free(bs);
      if (err.code) { free_sortie_contigu(img, sortie, slab, raster); return err; }

      FILE *outputfile = ouverture_fichier_out(nbcomp, nb_passage_sos);
      fprintf(outputfile, nbcomp == 1 ? "P5\n" : "P6\n");
      fprintf(outputfile, "%d %d\n", img->width, img->height);
      fprintf(outputfile, "255\n");

      /* change 2: the whole back end of this scan */
      const jpeggpu_qtable_t *qt[4] = { NULL, NULL, NULL, NULL };
      for (int i = 0; i < 4; i++)
         if (img->qtables[i] != NULL) qt[i] = (const jpeggpu_qtable_t *)img->qtables[i]->qtable;   /* qtable_t, include/img.h:8-11 */
      int rc = jpeggpu_backend_run(&d, (const jpeggpu_blocl16_t *const *)slab, qt, raster, 0);
      stop_timer(&timer_image);
      if (rc) { fclose(outputfile); err = jpeggpu_erreur(rc); free_sortie_contigu(img, sortie, slab, raster); return err; }
      start_timer(&timer_ecriture);
      fwrite(raster, 1, rows * rb, outputfile);                  /* every padded row, like save_mcu_ligne_* (src/decoder_utils.c:51,68) */
      stop_timer(&timer_ecriture);
      fclose(outputfile);

      char str[43];
      sprintf(str, "Décodage de l'image n°%ld", (long)nb_passage_sos);
      print_timer(str, &timer_image);
      sprintf(str, "Ecriture de l'image n°%ld", (long)nb_passage_sos);
      print_timer(str, &timer_ecriture);

      err = decode_entete(infile, false, img);
      if (err.code) { free_sortie_contigu(img, sortie, slab, raster); return err; }
      nb_passage_sos++;
   }

   /* symbolic link to the last image, as :333-342 */
   char *base_out_name = out_file_name(nbcomp, 0);
   char *out_name = out_file_name(nbcomp, nb_passage_sos - 1);
   char *sl = strrchr(out_name, '/');
   const char *target = sl ? sl + 1 : out_name;
   char *cmd = malloc(16 + strlen(target) + strlen(base_out_name));
   sprintf(cmd, "ln -f -s %s %s", target, base_out_name);
   (void)!system(cmd);
   free(out_name); free(base_out_name); free(cmd);

   free_sortie_contigu(img, sortie, slab, raster);
   return (erreur_t) {.code = SUCCESS};
}

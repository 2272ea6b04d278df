// compat.cpp -- libjpeggpu_compat.so: the reference's per-block symbols routed to the GPU
// (include/jpeggpu_compat.h).  No arithmetic here.
#include <cstdio>
#include <cstdlib>

#include "../../include/jpeggpu_compat.h"

static void must(int rc, const char *what) {
   if (rc == 0) return;
   fprintf(stderr, "jpeggpu_compat: %s: %d: %s\n", what, rc, jpeggpu_last_error());
   abort();
}

extern "C" {

void iquant(jpeggpu_blocl16_t *entree, uint8_t s_start, uint8_t s_end, jpeggpu_qtable_t *qtable) {
   must(jpeggpu_iquant_blocks(entree, 1, s_start, s_end, qtable), "iquant");
}

jpeggpu_bloct16_t *izz(jpeggpu_blocl16_t *entree) {
   jpeggpu_bloct16_t *res = (jpeggpu_bloct16_t *)malloc(sizeof *res);
   must(jpeggpu_izz_blocks(entree, res, 1), "izz");
   return res;
}

jpeggpu_bloctu8_t *idct_opt(jpeggpu_bloct16_t *mcu) {
   jpeggpu_bloctu8_t *res = (jpeggpu_bloctu8_t *)malloc(sizeof *res);
   must(jpeggpu_idct_opt_blocks(mcu, res, 1), "idct_opt");
   return res;
}

void ycc2rgb_pixel(uint8_t y, uint8_t cb, uint8_t cr, jpeggpu_rgb_t *rgb) {
   must(jpeggpu_ycc2rgb_pixels(&y, &cb, &cr, rgb, 1), "ycc2rgb_pixel");
}

}  // extern "C"

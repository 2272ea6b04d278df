/*
 * jpeggpu_compat.h -- link-compatibility face of the boundary (first row): the four
 * per-block symbols of the reference under their ORIGINAL names and ownership rules, so that the
 * reference's unmodified baseline.o / progressive.o / decoder_utils.o link against
 * libjpeggpu_compat.so instead of iqzz.o, idct_opt.o and ycc2rgb.o.
 *
 * Each call is one GPU round trip through the array forms of jpeggpu.h with n = 1 (there is no
 * CPU implementation in the product).  That is slow by construction -- 64 samples per launch --
 * and exists to prove drop-in parity function by function; the fast path is jpeggpu_backend_run*.
 * A GPU failure cannot be reported through these signatures (the reference's functions cannot
 * fail): the shim prints jpeggpu_last_error() and aborts.
 */
#ifndef JPEGGPU_COMPAT_H
#define JPEGGPU_COMPAT_H
#include "jpeggpu.h"
#ifdef __cplusplus
extern "C" {
#endif

/* include/iqzz.h:9 -- in place, int16 wrap (src/iqzz.c:9-14).                                    */
void iquant(jpeggpu_blocl16_t *entree, uint8_t s_start, uint8_t s_end, jpeggpu_qtable_t *qtable);
/* include/iqzz.h:13 -- returns a malloc'ed bloct16_t the caller frees (src/baseline.c:185-196).   */
jpeggpu_bloct16_t *izz(jpeggpu_blocl16_t *entree);
/* include/idct_opt.h:8 -- returns a malloc'ed bloctu8_t the caller frees (src/baseline.c:194-196). */
jpeggpu_bloctu8_t *idct_opt(jpeggpu_bloct16_t *mcu);
/* include/ycc2rgb.h:13.                                                                           */
void ycc2rgb_pixel(uint8_t y, uint8_t cb, uint8_t cr, jpeggpu_rgb_t *rgb);

#ifdef __cplusplus
}
#endif
#endif

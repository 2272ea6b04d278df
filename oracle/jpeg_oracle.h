/*
 * jpeg_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's pixel back end (iquant, izz, idct_opt, ycc2rgb_pixel,
 * save_mcu_ligne_bw/color).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this; the product library (libjpeggpu.so) never links or calls it.
 *
 * Parity pin: proven bit-identical to the reference's own objects (oracle/_ref, strict -O2 build)
 * by oracle/check_oracle_vs_ref.c in the build container, and to the reference's golden files
 * valid_out/invader.pgm and valid_out/shaun_the_sheep.ppm through tests/golden/.
 */
#ifndef JPEG_ORACLE_H
#define JPEG_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "../include/jpeggpu.h"   /* layout structs + frame descriptor only */

#ifdef __cplusplus
extern "C" {
#endif

void oracle_iquant(int16_t zz[64], int s_start, int s_end, const uint16_t q[64]);
void oracle_izz(const int16_t zz[64], int16_t out[8][8]);
void oracle_idct_opt(const int16_t in[8][8], uint8_t out[8][8]);
void oracle_ycc2rgb_pixel(uint8_t y, uint8_t cb, uint8_t cr, uint8_t rgb[3]);
/* The slow transform selected by "-f" (src/idct.c:35-71): direct 64-term sum per sample with the
 * coefficient table of calc_coef (cos/sqrt of the platform's libm, like the reference).          */
void oracle_idct_coef(float coef[8][8][8][8]);
void oracle_idct(const int16_t in[8][8], uint8_t out[8][8]);
void oracle_idct_blocks(const jpeggpu_bloct16_t *in, jpeggpu_bloctu8_t *out, size_t n);
/* all_option.idct_fast (src/options.c:76-79): 1 = idct_opt (default), 0 = idct, for oracle_backend_*. */
void oracle_set_idct_fast(int fast);

/* n-item forms (same signatures as the jpeggpu_*_blocks stage entries, minus the GPU). */
void oracle_iquant_blocks(jpeggpu_blocl16_t *blocks, size_t n, int s_start, int s_end,
                          const jpeggpu_qtable_t *q);
void oracle_izz_blocks(const jpeggpu_blocl16_t *in, jpeggpu_bloct16_t *out, size_t n);
void oracle_idct_opt_blocks(const jpeggpu_bloct16_t *in, jpeggpu_bloctu8_t *out, size_t n);
void oracle_ycc2rgb_pixels(const uint8_t *y, const uint8_t *cb, const uint8_t *cr,
                           jpeggpu_rgb_t *rgb, size_t n);

/* Whole back end over one frame; same contract as jpeggpu_backend_run.  MCU rows
 * [mcu_row0, mcu_row1) only (0, nbmcuV for the full frame) so that callers can thread it.     */
int oracle_backend_rows(const jpeggpu_frame_desc_t *desc, const jpeggpu_blocl16_t *const coef[3],
                        const jpeggpu_qtable_t *const qtables[4], uint8_t *out, size_t out_stride,
                        uint32_t mcu_row0, uint32_t mcu_row1);
int oracle_backend_run(const jpeggpu_frame_desc_t *desc, const jpeggpu_blocl16_t *const coef[3],
                       const jpeggpu_qtable_t *const qtables[4], uint8_t *out, size_t out_stride);

/* Geometry exactly as calcul_image_information (src/entete.c:176-197). */
int oracle_frame_geometry(jpeggpu_frame_desc_t *desc);

/* Deterministic synthetic coefficients: counter-based splitmix64, block-indexed.
 * Fills n blocks starting at global block index <first_block> of stream <seed>.
 * kind 0: JPEG-like sparse Laplacian; 1: uniform +-1024; 2: full-range int16 (wrap stress).   */
void oracle_synth_blocks(jpeggpu_blocl16_t *blocks, size_t n, uint64_t seed, uint64_t first_block,
                         int kind);

#ifdef __cplusplus
}
#endif
#endif

// host_mirror_check.cpp -- TEST TOOL.  Compiles jpeg_math.hpp (the device arithmetic) for the HOST
// and checks it against the oracle, so the CPU-only suite pins the zigzag tables, the deferred
// halvings / folded constants and the integer colour formulas before any GPU is involved.
//   usage: host_mirror_check [n_blocks_per_kind=200000]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "jpeg_math.hpp"
#include "../../oracle/jpeg_oracle.h"

using namespace jpegmath;

static size_t check_blocks(size_t n, uint64_t seed, int kind, const jpeggpu_qtable_t &q) {
   std::vector<jpeggpu_blocl16_t> in(n);
   oracle_synth_blocks(in.data(), n, seed, 0, kind);
   uint32_t qw[32];
   for (int i = 0; i < 32; i++) qw[i] = q.data[2 * i] | ((uint32_t)q.data[2 * i + 1] << 16);
   size_t bad = 0;
   for (size_t b = 0; b < n; b++) {
      jpeggpu_blocl16_t z = in[b];
      jpeggpu_bloct16_t sq; jpeggpu_bloctu8_t want;
      oracle_iquant(z.data, 0, 63, q.data);
      oracle_izz(z.data, sq.data);
      oracle_idct_opt(sq.data, want.data);
      uint32_t zw[32]; memcpy(zw, in[b].data, 128);
      float m[8][8]; uint2 rows[8], rows2[8];
      dequant_izz(zw, qw, m);
      idct_block(m, rows);
      f32x2 p1[4][8];
      dequant_izz_pk(zw, qw, p1);           // the packed form the fused kernel uses
      idct_block_pk(p1, rows2);
      if (memcmp(rows, rows2, sizeof rows)) { bad++; continue; }
      f32x2 p2[4][8]; uint2 rows3[8];
      dequant_izz_rp(zw, qw, p2);           // pass 2 packed within a row (pairing (0,4)(2,6)(1,7)(3,5))
      idct_block_rp(p2, rows3);
      if (memcmp(rows, rows3, sizeof rows)) { bad++; continue; }
      for (int r = 0; r < 8; r++)
         for (int c = 0; c < 8; c++) {
            uint8_t got = (uint8_t)(((c < 4 ? rows[r].x : rows[r].y) >> (8 * (c & 3))) & 0xFF);
            if (got != want.data[c][r]) { bad++; r = 8; break; }
         }
   }
   printf("host mirror idct kind %d: %zu blocks, %zu differing\n", kind, n, bad);
   return bad;
}

int main(int argc, char **argv) {
   size_t n = argc > 1 ? (size_t)atol(argv[1]) : 200000;
   jpeggpu_qtable_t qa, qc;
   for (int i = 0; i < 64; i++) { qa.data[i] = (uint16_t)(2 + i); qc.data[i] = (uint16_t)(40503u * (unsigned)(i + 1)); }
   size_t bad = 0;
   bad += check_blocks(n, 11, 0, qa);
   bad += check_blocks(n, 12, 1, qa);
   bad += check_blocks(n, 13, 2, qc);
   size_t cbad = 0;
   for (int y = 0; y < 256; y++)
      for (int cb = 0; cb < 256; cb++)
         for (int cr = 0; cr < 256; cr++) {
            uint8_t want[3];
            oracle_ycc2rgb_pixel((uint8_t)y, (uint8_t)cb, (uint8_t)cr, want);
            uint32_t v = ycc_to_rgb(y, chroma_offsets(cb, cr));
            cbad += (v != ((uint32_t)want[0] | ((uint32_t)want[1] << 8) | ((uint32_t)want[2] << 16)));
            if (!(cb == 28 && cr == 228)) {   // the fused kernel's fast path (everything but the special pair)
               Off3 o = chroma_off3(cb - 128, cr - 128);
               cbad += (clamp8(y + o.r) != want[0]) + (clamp8(y + o.g) != want[1]) + (clamp8(y + o.b) != want[2]);
            }
         }
   printf("host mirror ycc2rgb: 2^24 inputs, %zu differing\n", cbad);
   // the packed 16-bit form (rgb_quad_pk): every (cb, cr) under pixels 0,1 with a second pair under pixels 2,3,
   // every luma value, both chroma layouts
   size_t pbad = 0, pn = 0;
   for (int cb = 0; cb < 256; cb++)
      for (int cr = 0; cr < 256; cr++) {
         const int cb2 = (cb * 7 + 3) & 255, cr2 = (cr * 13 + 5) & 255;
         if ((cb == 28 && cr == 228) || (cb2 == 28 && cr2 == 228)) continue;
         const uint32_t cbw1 = (uint32_t)cb | ((uint32_t)cb << 8) | ((uint32_t)cb2 << 16) | ((uint32_t)cb2 << 24);
         const uint32_t crw1 = (uint32_t)cr | ((uint32_t)cr << 8) | ((uint32_t)cr2 << 16) | ((uint32_t)cr2 << 24);
         const uint32_t cbw2 = (uint32_t)cb | ((uint32_t)cb2 << 8), crw2 = (uint32_t)cr | ((uint32_t)cr2 << 8);
         const ChromaPairs o1 = chroma_pairs<1>(cbw1 ^ 0x80808080u, crw1 ^ 0x80808080u);
         const ChromaPairs o2 = chroma_pairs<2>(cbw2 ^ 0x80808080u, crw2 ^ 0x80808080u);
         for (int q = 0; q < 64; q++) {
            const uint32_t yw = (uint32_t)(4 * q) | ((uint32_t)(4 * q + 1) << 8) | ((uint32_t)(4 * q + 2) << 16) | ((uint32_t)(4 * q + 3) << 24);
            uint8_t want[12];
            for (int i = 0; i < 4; i++)
               oracle_ycc2rgb_pixel((uint8_t)(4 * q + i), (uint8_t)(i < 2 ? cb : cb2), (uint8_t)(i < 2 ? cr : cr2), want + 3 * i);
            uint32_t w1[3], w2[3];
            rgb_quad_pk(yw, o1, w1);
            rgb_quad_pk(yw, o2, w2);
            pbad += (memcmp(w1, want, 12) != 0) + (memcmp(w2, want, 12) != 0);
            pn += 2;
         }
      }
   printf("host mirror packed colour path: %zu quads, %zu differing\n", pn, pbad);
   bad += pbad;
   bad += cbad;
   printf(bad ? "host_mirror_check: FAILED\n" : "host_mirror_check: OK\n");
   return bad != 0;
}

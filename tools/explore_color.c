/*
 * explore_color.c -- finds and verifies integer formulas that reproduce ycc2rgb_pixel
 * (src/ycc2rgb.c:18-22, all-double arithmetic with clamp + truncation) on ALL 2^24 inputs.
 *
 *   r = clamp8(y + ((RA*(cr-128) + RB) >> S))
 *   b = clamp8(y + ((BA*(cb-128) + BB) >> S))
 *   g = clamp8(y + ((GB - GA1*(cb-128) - GA2*(cr-128)) >> S))
 *
 *   gcc -O2 -ffp-contract=off -o explore_color explore_color.c
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

static double clampd(double v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
static void ref(int y, int cb, int cr, int *r, int *g, int *b) {
   *r = (uint8_t)clampd((double)y + 1.402 * ((double)cr - 128));
   *g = (uint8_t)clampd((double)y - 0.34414 * ((double)cb - 128) - 0.71414 * ((double)cr - 128));
   *b = (uint8_t)clampd((double)y + 1.772 * ((double)cb - 128));
}
static int clamp8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

int main(void) {
   /* 1. every channel is clamp8(y + off(chroma)), except ONE chroma pair whose green offset
         depends on y: cb=28, cr=228, where -0.34414*(-100) - 0.71414*100 is exactly -37 in real
         arithmetic and the double evaluation lands just below the integer for 94 <= y <= 165. */
   static int roff[256], boff[256], goff[256][256];
   for (int c = 0; c < 256; c++) {
      roff[c] = boff[c] = 1 << 30;
      for (int y = 0; y < 256; y++) {
         int r, g, b; ref(y, c, c, &r, &g, &b);
         if (r > 0 && r < 255 && roff[c] == 1 << 30) roff[c] = r - y;
         if (b > 0 && b < 255 && boff[c] == 1 << 30) boff[c] = b - y;
      }
   }
   for (int cb = 0; cb < 256; cb++) for (int cr = 0; cr < 256; cr++) {
      goff[cb][cr] = 1 << 30;
      for (int y = 0; y < 256; y++) { int r, g, b; ref(y, cb, cr, &r, &g, &b); if (g > 0 && g < 255) { goff[cb][cr] = g - y; break; } }
      if (goff[cb][cr] == 1 << 30) { printf("no unclamped g for cb=%d cr=%d\n", cb, cr); return 1; }
   }
   long bad = 0;
   for (int y = 0; y < 256; y++) for (int cb = 0; cb < 256; cb++) for (int cr = 0; cr < 256; cr++) {
      int r, g, b; ref(y, cb, cr, &r, &g, &b);
      int fix = (cb == 28 && cr == 228 && (unsigned)(y - 94) < 72u);
      bad += r != clamp8(y + roff[cr]);
      bad += b != clamp8(y + boff[cb]);
      bad += g != clamp8(y + goff[cb][cr] - fix);
   }
   printf("offset form (+ the one y-dependent pair): %ld mismatches over 2^24 inputs\n", bad);
   if (bad) return 1;

   /* 2. fit multiply-shift formulas: for each (S, A) intersect the admissible bias intervals */
   for (int S = 14; S <= 23; S++) {
      long one = 1L << S;
      long ra0 = (long)(1.402 * one + 0.5), ba0 = (long)(1.772 * one + 0.5);
      for (long A = ra0 - 1; A <= ra0 + 1; A++) {
         long lo = -(1L << 62), hi = 1L << 62;
         for (int c = 0; c < 256; c++) { long l = roff[c] * one - A * (c - 128), h = l + one; if (l > lo) lo = l; if (h < hi) hi = h; }
         if (lo < hi) printf("R: S=%d A=%ld B in [%ld, %ld)\n", S, A, lo, hi);
      }
      for (long A = ba0 - 1; A <= ba0 + 1; A++) {
         long lo = -(1L << 62), hi = 1L << 62;
         for (int c = 0; c < 256; c++) { long l = boff[c] * one - A * (c - 128), h = l + one; if (l > lo) lo = l; if (h < hi) hi = h; }
         if (lo < hi) printf("B: S=%d A=%ld B in [%ld, %ld)\n", S, A, lo, hi);
      }
      long g10 = (long)(0.34414 * one + 0.5), g20 = (long)(0.71414 * one + 0.5);
      for (long A1 = g10 - 1; A1 <= g10 + 1; A1++) for (long A2 = g20 - 1; A2 <= g20 + 1; A2++) {
         long lo = -(1L << 62), hi = 1L << 62;
         for (int cb = 0; cb < 256; cb++) for (int cr = 0; cr < 256; cr++) {
            long l = goff[cb][cr] * one + A1 * (cb - 128) + A2 * (cr - 128), h = l + one;
            if (l > lo) lo = l; if (h < hi) hi = h;
         }
         if (lo < hi) printf("G: S=%d A1=%ld A2=%ld B in [%ld, %ld)\n", S, A1, A2, lo, hi);
      }
   }
   return 0;
}

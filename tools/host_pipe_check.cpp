// tools/host_pipe_check.cpp -- the host-pointer entries (jpeggpu_backend_run, jpeggpu_backend_run_batch: stripes, helper threads,
// staging rings) driven from C++ so that the whole library can run with its HOST code under AddressSanitizer on the GPU box
// (make -C jpeg-decoder_amd/csrc asan builds tools/host_pipe_asan from this file + every library source).  Pageable and pinned
// caller memory must give the same bytes; guard bytes around the rasters must stay untouched.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <unistd.h>
#include "../include/jpeggpu.h"
extern "C" void __sanitizer_purge_allocator(void);   // ASan builds only (weakly referenced below)

static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

static int check(int rc, const char *what) { if (rc) { fprintf(stderr, "%s: %d: %s\n", what, rc, jpeggpu_last_error()); exit(2); } return rc; }

int main() {
   check(jpeggpu_init(0), "jpeggpu_init");
   jpeggpu_qtable_t q[2];
   for (int t = 0; t < 2; t++) for (int i = 0; i < 64; i++) q[t].data[i] = (uint16_t)(1 + (i * (t + 2)) % 37);
   const jpeggpu_qtable_t *qt[4] = { &q[0], &q[1], nullptr, nullptr };
   struct Case { uint16_t w, h; uint8_t nb, hs0, vs0; uint32_t frames; };
   const Case cases[] = { { 2048, 2048, 3, 1, 1, 1 }, { 4096, 1400, 3, 2, 2, 1 }, { 1920, 1080, 3, 2, 2, 9 }, { 3000, 1999, 1, 1, 1, 2 }, { 517, 130, 3, 2, 1, 40 } };
   for (const Case &c : cases) {
      jpeggpu_frame_desc_t d;
      memset(&d, 0, sizeof d);
      d.width = c.w; d.height = c.h; d.nbcomp = c.nb; d.apply_iquant = 1;
      d.comp[0].hs = c.hs0; d.comp[0].vs = c.vs0; d.comp[0].idq = 0;
      for (int k = 1; k < c.nb; k++) { d.comp[k].hs = 1; d.comp[k].vs = 1; d.comp[k].idq = 1; }
      check(jpeggpu_frame_geometry(&d), "jpeggpu_frame_geometry");
      const size_t rows = jpeggpu_raster_rows(&d), rb = jpeggpu_raster_row_bytes(&d), out_bytes = rows * rb * c.frames, guard = 4096;
      size_t nb[3] = { 0, 0, 0 };
      std::vector<std::vector<jpeggpu_blocl16_t>> planes(3);
      jpeggpu_blocl16_t *pinned[3] = { nullptr, nullptr, nullptr };
      const jpeggpu_blocl16_t *pp[3] = { nullptr, nullptr, nullptr }, *pg[3] = { nullptr, nullptr, nullptr };
      for (int k = 0; k < c.nb; k++) {
         nb[k] = jpeggpu_plane_blocks(&d, k);
         planes[k].resize(nb[k] * c.frames);
         for (auto &b : planes[k]) for (int i = 0; i < 64; i++) b.data[i] = (rnd() % 8 < (unsigned)(i < 6 ? 7 : 2)) ? (int16_t)((int)(rnd() % 61) - 30) : 0;
         pinned[k] = (jpeggpu_blocl16_t *)jpeggpu_host_alloc(planes[k].size() * sizeof(jpeggpu_blocl16_t));
         if (!pinned[k]) { fprintf(stderr, "jpeggpu_host_alloc failed\n"); return 2; }
         memcpy(pinned[k], planes[k].data(), planes[k].size() * sizeof(jpeggpu_blocl16_t));
         pg[k] = planes[k].data(); pp[k] = pinned[k];
      }
      std::vector<uint8_t> out_a(out_bytes + 2 * guard, 0xA5), out_b(out_bytes + 2 * guard, 0xA5), out_c(out_bytes + 2 * guard, 0xA5);
      uint8_t *out_p = (uint8_t *)jpeggpu_host_alloc(out_bytes);
      if (!out_p) return 2;
      check(jpeggpu_backend_run_batch(&d, c.frames, pg, nullptr, qt, out_a.data() + guard, 0, 0), "batch, pageable");
      check(jpeggpu_backend_run_batch(&d, c.frames, pp, nullptr, qt, out_p, 0, 0), "batch, pinned");
      check(jpeggpu_backend_run_batch(&d, c.frames, pp, nullptr, qt, out_b.data() + guard, 0, 0), "batch, pinned in / pageable out");
      for (uint32_t f = 0; f < c.frames; f++) {
         const jpeggpu_blocl16_t *one[3] = { nullptr, nullptr, nullptr };
         for (int k = 0; k < c.nb; k++) one[k] = pg[k] + f * nb[k];
         check(jpeggpu_backend_run(&d, one, qt, out_c.data() + guard + f * rows * rb, 0), "single frame, pageable");
      }
      int bad = memcmp(out_a.data() + guard, out_p, out_bytes) != 0 || out_a != out_b || out_a != out_c;
      for (size_t i = 0; i < guard && !bad; i++) bad = out_a[i] != 0xA5 || out_a[guard + out_bytes + i] != 0xA5;
      printf("%ux%u x%u comps, %u frame(s): %s\n", c.w, c.h, c.nb, c.frames, bad ? "MISMATCH" : "ok");
      jpeggpu_host_free(out_p);
      for (int k = 0; k < c.nb; k++) jpeggpu_host_free(pinned[k]);
      if (bad) return 1;
   }
   jpeggpu_shutdown();
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
   __sanitizer_purge_allocator();   // freed HSA allocations leave ASan's quarantine while the HSA runtime is alive (see jpeg2ppm_gpu.cpp)
#endif
#endif
   printf("host_pipe_check: all ok\n");
   return 0;
}

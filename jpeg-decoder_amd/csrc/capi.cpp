// capi.cpp -- the C-ABI layer declared in include/jpeggpu.h.
//
// Host-side logic only: argument validation with the reference's error convention, tile planning,
// pinned staging for the host-pointer forms, and kernel launches.  No pixel arithmetic lives here
// and there is no CPU fallback: every entry point that produces data needs a HIP device.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/jpeggpu.h"
#include "kernels.h"

using namespace jpeggpu_k;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
   va_list ap;
   va_start(ap, fmt);
   vsnprintf(g_err, sizeof g_err, fmt, ap);
   va_end(ap);
   return code;
}

struct State {
   bool ready = false;
   int device = -1;
   hipStream_t stream = nullptr;
   uint32_t num_cus = 256;
   // device + pinned staging for the host-pointer entry points, grown on demand
   void *d_in = nullptr;  size_t d_in_cap = 0;
   void *d_out = nullptr; size_t d_out_cap = 0;
   void *h_pin = nullptr; size_t h_pin_cap = 0;
   // "-f" mode (all_option.idct_fast == 0): calc_coef's table on the device, built on first use
   bool idct_fast = true;
   float *d_slow_coef = nullptr;
};
State g;
std::mutex g_mu;

#define HIP_TRY(expr)                                                                            \
   do {                                                                                          \
      hipError_t e_ = (expr);                                                                    \
      if (e_ != hipSuccess) return fail(JPEGGPU_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
   } while (0)

int ensure_init() {
   if (g.ready) return 0;
   return jpeggpu_init(0);
}

int grow(void **p, size_t *cap, size_t need, bool pinned) {
   if (*cap >= need) return 0;
   if (*p) { if (pinned) (void)hipHostFree(*p); else (void)hipFree(*p); *p = nullptr; *cap = 0; }
   size_t want = need + need / 4 + 4096;
   if (pinned) HIP_TRY(hipHostMalloc(p, want, hipHostMallocDefault));
   else HIP_TRY(hipMalloc(p, want));
   *cap = want;
   return 0;
}

// calc_coef (src/idct.c:35-48) with f_C (:19-24) and calc_cos (:26-32): the double cosine is stored as
// float, C(lambda)C(mu) is returned as float, and the three floats are multiplied left to right.
// cos/sqrt come from the platform's libm, as they do for the reference on the same host.
void calc_coef_host(float *t) {
   float cs[8][8];
   for (int x = 0; x < 8; x++)
      for (int l = 0; l < 8; l++) cs[x][l] = (float)std::cos((2 * x + 1) * l * 3.14159265358979323846 / 16);
   const float r2 = (float)(1 / std::sqrt(2.0));
   for (int x = 0; x < 8; x++)
      for (int y = 0; y < 8; y++)
         for (int l = 0; l < 8; l++)
            for (int m = 0; m < 8; m++) {
               const float fc = l == 0 ? (m == 0 ? 0.5f : r2) : (m == 0 ? r2 : 1.0f);
               float v = fc * cs[x][l];
               v = v * cs[y][m];
               t[((x * 8 + y) * 8 + l) * 8 + m] = v;
            }
}

int ensure_slow_coef() {
   if (g.d_slow_coef) return 0;
   static float table[4096];
   calc_coef_host(table);
   HIP_TRY(hipMalloc((void **)&g.d_slow_coef, sizeof table));
   HIP_TRY(hipMemcpy(g.d_slow_coef, table, sizeof table, hipMemcpyHostToDevice));
   return 0;
}

void pack_qtable(const jpeggpu_qtable_t *q, uint32_t w[32]) {
   for (int i = 0; i < 32; i++) w[i] = (uint32_t)q->data[2 * i] | ((uint32_t)q->data[2 * i + 1] << 16);
}

// Validation shared by all frame-level entries.  Error order follows the reference: component
// count first (src/decoder_utils.c:50,67), then tables (src/progressive.c:262-267).
int validate(const jpeggpu_frame_desc_t *d, const jpeggpu_qtable_t *const qt[4]) {
   if (!d) return fail(JPEGGPU_ERR_INVALID_ARG, "desc is NULL");
   if (d->nbcomp != 1 && d->nbcomp != 3)
      return fail(JPEGGPU_ERR_NB_COMP, "Il faut une ou trois composantes (nbcomp=%d)", d->nbcomp);
   if (d->width == 0 || d->nbmcuH == 0 || d->nbmcuV == 0 || d->max_h == 0 || d->max_v == 0)
      return fail(JPEGGPU_ERR_INVALID_ARG, "empty geometry (width=%u nbmcuH=%u nbmcuV=%u max_h=%u max_v=%u)", d->width, d->nbmcuH,
                  d->nbmcuV, d->max_h, d->max_v);
   uint32_t per_mcu = 0;
   for (int c = 0; c < d->nbcomp; c++) {
      const jpeggpu_component_t &k = d->comp[c];
      if (k.hs == 0 || k.vs == 0 || k.hs > 4 || k.vs > 4)
         return fail(JPEGGPU_ERR_SAMPLING, "component %d: sampling factors %ux%u out of 1..4", c, k.hs, k.vs);
      if (d->max_h % k.hs || d->max_v % k.vs || k.hs > d->max_h || k.vs > d->max_v)
         return fail(JPEGGPU_ERR_SAMPLING, "component %d: %ux%u does not divide max %ux%u", c, k.hs, k.vs, d->max_h, d->max_v);
      if (d->apply_iquant && (k.idq > 3 || !qt || !qt[k.idq]))
         return fail(JPEGGPU_ERR_NO_QT, "Pas de table de quantification pour la composante %d", c);
      per_mcu += (uint32_t)k.hs * k.vs;
   }
   if ((uint64_t)d->nbmcuH * d->max_h * 8 < d->width)
      return fail(JPEGGPU_ERR_INVALID_ARG, "width %u exceeds nbmcuH*max_h*8 = %llu", d->width,
                  (unsigned long long)d->nbmcuH * d->max_h * 8);
   (void)per_mcu;
   return 0;
}


// Tile planning: T MCUs per workgroup, one thread per block.
void plan_backend(const jpeggpu_frame_desc_t *d, const jpeggpu_qtable_t *const qt[4], BackendParams &p, LaunchDims &ld) {
   memset(&p, 0, sizeof p);
   uint32_t per_mcu = 0;
   for (int c = 0; c < d->nbcomp; c++) per_mcu += (uint32_t)d->comp[c].hs * d->comp[c].vs;
   uint32_t target_blocks = 256;
   if (const char *e = getenv("JPEGGPU_TARGET_BLOCKS")) target_blocks = (uint32_t)atoi(e);   // tuning knob
   if (target_blocks < 48) target_blocks = 48;          // one MCU can hold 3 * 4*4 blocks
   if (target_blocks > kMaxThreads) target_blocks = kMaxThreads;   // __launch_bounds__
   // T MCUs per tile: the largest power of two that fits the block budget (4:4:4 -> 64 MCUs = 192
   // blocks, 4:2:0 -> 64 = 384 (below), 4:2:2 -> 64 = 256, gray -> 256), never wider than the frame needs
   uint32_t T = 1;
   while (T * 2 * per_mcu <= target_blocks) T *= 2;
   // 6 blocks per MCU (4:2:0, 4:1:1): 32 MCUs would put Cb and Cr into one wave; 64 MCUs = 384 blocks keep every
   // wave on one component (the kernel's scalar prologue) and measured 1.7% faster (profiles/r01_kernel_experiments.txt)
   auto wave_uniform = [&](uint32_t t) {
      uint32_t b = 0;
      for (int c = 0; c < d->nbcomp; c++) { b += t * d->comp[c].hs * d->comp[c].vs; if (b % 64) return false; }
      return true;
   };
   if (!getenv("JPEGGPU_TARGET_BLOCKS") && !wave_uniform(T) && T * 2 * per_mcu <= kMaxThreads && wave_uniform(T * 2)) T *= 2;
   // a frame narrower than the tile keeps a power-of-two tile (the unused lanes idle): the sub-sampling-specific
   // kernels turn the item index into (row, quad) with a shift and a mask
   while (T / 2 >= d->nbmcuH && T > 1) T /= 2;
   if (const char *e = getenv("JPEGGPU_TILE_MCUS")) {   // tuning knob: any T with T*blocks_per_MCU <= 256 (not a power of two: generic kernel)
      uint32_t v = (uint32_t)atoi(e);
      if (v >= 1 && v * per_mcu <= kMaxThreads) T = v;
   }
   p.T = T; p.B = T * per_mcu;
   p.nbcomp = d->nbcomp; p.apply_iquant = d->apply_iquant; p.width = d->width;
   p.tile_w = T * d->max_h * 8; p.tile_rows = (uint32_t)d->max_v * 8;
   p.wq_shift = 0xFFFFFFFFu;
   for (uint32_t sft = 0; sft < 16; sft++)
      if ((p.tile_w >> 2) == (1u << sft)) p.wq_shift = sft;
   uint32_t boff = 0, poff = 0;
   for (int c = 0; c < d->nbcomp; c++) {
      const jpeggpu_component_t &k = d->comp[c];
      p.nbH[c] = d->nbmcuH * k.hs; p.vs[c] = k.vs; p.bw[c] = T * k.hs;
      p.hf[c] = d->max_h / k.hs; p.vf[c] = d->max_v / k.vs; p.idq[c] = k.idq;
      p.boff[c] = boff; p.plane_off[c] = poff;
      boff += T * k.hs * k.vs; poff += T * k.hs * k.vs * 64;
   }
   for (int c = d->nbcomp; c < 4; c++) p.boff[c] = boff;
   p.wave_uniform = 1;
   for (int c = 1; c <= d->nbcomp; c++)
      if (p.boff[c] % 64) p.wave_uniform = 0;
   if (getenv("JPEGGPU_NO_UNIFORM")) p.wave_uniform = 0;   // diagnostic: per-lane prologue everywhere
   if (d->apply_iquant)
      for (int i = 0; i < 4; i++)
         if (qt[i]) pack_qtable(qt[i], p.qt[i]);
   ld.tiles_per_row = (d->nbmcuH + T - 1) / T;
   ld.mcu_rows = d->nbmcuV;
   ld.threads = (p.B + 63) / 64 * 64;                 // one thread per block
   ld.lds_bytes = (size_t)ld.threads * 64 + 512;
   if (const char *e = getenv("JPEGGPU_DIAG_LDS_BYTES")) {   // diagnostic: larger allocation = fewer resident workgroups per CU
      size_t v = (size_t)atol(e);
      if (v > ld.lds_bytes && v <= 64 * 1024) ld.lds_bytes = v;
   }
}

int run_device(const jpeggpu_frame_desc_t *d, uint32_t nframes, const void *const d_coef[3], const size_t cfs[3],
               const jpeggpu_qtable_t *const qt[4], void *d_out, size_t out_stride, size_t out_frame_stride, hipStream_t s) {
   BackendParams p; LaunchDims ld;
   plan_backend(d, qt, p, ld);
   const size_t row_bytes = jpeggpu_raster_row_bytes(d), rows = jpeggpu_raster_rows(d);
   if (out_stride == 0) out_stride = row_bytes;
   if (out_stride < row_bytes) return fail(JPEGGPU_ERR_INVALID_ARG, "out_stride %zu < row bytes %zu", out_stride, row_bytes);
   if (out_stride > ((size_t)1 << 26)) return fail(JPEGGPU_ERR_INVALID_ARG, "out_stride %zu above 2^26 (the kernel addresses a tile's rows with 32-bit offsets)", out_stride);
   if (out_frame_stride == 0) out_frame_stride = out_stride * rows;
   for (int c = 0; c < d->nbcomp; c++) {
      if (!d_coef || !d_coef[c]) return fail(JPEGGPU_ERR_INVALID_ARG, "coef[%d] is NULL", c);
      if ((uintptr_t)d_coef[c] & 15) return fail(JPEGGPU_ERR_INVALID_ARG, "coef[%d] must be 16-byte aligned", c);
      p.coef[c] = d_coef[c];
      size_t dense = jpeggpu_plane_blocks(d, c) * sizeof(jpeggpu_blocl16_t);
      p.coef_frame_stride[c] = (cfs && cfs[c]) ? cfs[c] : dense;
      if (p.coef_frame_stride[c] & 15) return fail(JPEGGPU_ERR_INVALID_ARG, "coef_frame_stride[%d] must be a multiple of 16", c);
   }
   if (!d_out) return fail(JPEGGPU_ERR_INVALID_ARG, "out is NULL");
   p.out = d_out; p.out_stride = out_stride; p.out_frame_stride = out_frame_stride;
   if (const char *e = getenv("JPEGGPU_DIAG_PTR")) p.diag = (void *)strtoull(e, nullptr, 0);   // diagnostic builds only
   if (nframes == 0) return 0;
   if (!g.idct_fast) { int rc = ensure_slow_coef(); if (rc) return rc; p.slow_coef = g.d_slow_coef; }
   if (nframes > 65535 || d->nbmcuV > 65535) return fail(JPEGGPU_ERR_INVALID_ARG, "nframes / nbmcuV above 65535");
   ld.frames = nframes;
   HIP_TRY(launch_backend(p, ld, s));
   return 0;
}

}  // namespace

extern "C" {

const char *jpeggpu_last_error(void) { return g_err; }

// used by decode_api.cpp (same library); not part of the public headers
void jpeggpu_internal_set_error(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg ? msg : ""); }
int jpeggpu_internal_ensure_init(void) { return ensure_init(); }

int jpeggpu_init(int device) {
   std::lock_guard<std::mutex> lk(g_mu);
   if (g.ready && g.device == device) return 0;
   if (g.ready) return fail(JPEGGPU_ERR_INVALID_ARG, "already bound to device %d (one process per GPU)", g.device);
   int n = 0;
   hipError_t e = hipGetDeviceCount(&n);
   if (e != hipSuccess || n <= 0) return fail(JPEGGPU_ERR_NO_DEVICE, "no HIP device (%s)", e == hipSuccess ? "count is 0" : hipGetErrorString(e));
   if (device < 0 || device >= n) return fail(JPEGGPU_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, n);
   HIP_TRY(hipSetDevice(device));
   HIP_TRY(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
   int cus = 0;
   if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) g.num_cus = (uint32_t)cus;
   g.device = device; g.ready = true;
   g_err[0] = 0;
   return 0;
}

void jpeggpu_shutdown(void) {
   std::lock_guard<std::mutex> lk(g_mu);
   if (!g.ready) return;
   (void)hipStreamSynchronize(g.stream);
   if (g.d_in) (void)hipFree(g.d_in);
   if (g.d_out) (void)hipFree(g.d_out);
   if (g.h_pin) (void)hipHostFree(g.h_pin);
   if (g.d_slow_coef) (void)hipFree(g.d_slow_coef);
   (void)hipStreamDestroy(g.stream);
   g = State();
}

int jpeggpu_frame_geometry(jpeggpu_frame_desc_t *d) {
   if (!d) return fail(JPEGGPU_ERR_INVALID_ARG, "desc is NULL");
   if (d->nbcomp != 1 && d->nbcomp != 3) return fail(JPEGGPU_ERR_NB_COMP, "Il faut une ou trois composantes (nbcomp=%d)", d->nbcomp);
   // calcul_image_information (src/entete.c:176-197): float ceil of float quotients
   int nbh = (int)std::ceil((float)d->width / 8), nbv = (int)std::ceil((float)d->height / 8);
   uint8_t mh = 0, mv = 0;
   for (int c = 0; c < d->nbcomp; c++) {
      if (d->comp[c].hs > mh) mh = d->comp[c].hs;
      if (d->comp[c].vs > mv) mv = d->comp[c].vs;
   }
   if (!mh || !mv) return fail(JPEGGPU_ERR_SAMPLING, "sampling factors are zero");
   d->max_h = mh; d->max_v = mv;
   d->nbmcuH = (uint32_t)std::ceil((float)nbh / mh);
   d->nbmcuV = (uint32_t)std::ceil((float)nbv / mv);
   return 0;
}

size_t jpeggpu_plane_blocks(const jpeggpu_frame_desc_t *d, int c) {
   if (!d || c < 0 || c >= d->nbcomp) return 0;
   return (size_t)d->nbmcuH * d->comp[c].hs * (size_t)d->nbmcuV * d->comp[c].vs;
}
size_t jpeggpu_raster_rows(const jpeggpu_frame_desc_t *d) { return d ? (size_t)d->nbmcuV * d->max_v * 8 : 0; }
size_t jpeggpu_raster_row_bytes(const jpeggpu_frame_desc_t *d) { return d ? (size_t)d->width * (d->nbcomp == 3 ? 3 : 1) : 0; }

int jpeggpu_synchronize(void *hip_stream) {
   int rc = ensure_init(); if (rc) return rc;
   HIP_TRY(hipStreamSynchronize(hip_stream ? (hipStream_t)hip_stream : g.stream));
   return 0;
}

int jpeggpu_backend_run_batch_device(const jpeggpu_frame_desc_t *d, uint32_t nframes, const void *const d_coef[3],
                                     const size_t coef_frame_stride[3], const jpeggpu_qtable_t *const qt[4], void *d_out,
                                     size_t out_stride, size_t out_frame_stride, void *hip_stream) {
   int rc = validate(d, qt); if (rc) return rc;
   rc = ensure_init(); if (rc) return rc;
   return run_device(d, nframes, d_coef, coef_frame_stride, qt, d_out, out_stride, out_frame_stride,
                     hip_stream ? (hipStream_t)hip_stream : g.stream);
}

int jpeggpu_backend_run_device(const jpeggpu_frame_desc_t *d, const void *const d_coef[3], const jpeggpu_qtable_t *const qt[4],
                               void *d_out, size_t out_stride, void *hip_stream) {
   return jpeggpu_backend_run_batch_device(d, 1, d_coef, nullptr, qt, d_out, out_stride, 0, hip_stream);
}

int jpeggpu_backend_run(const jpeggpu_frame_desc_t *d, const jpeggpu_blocl16_t *const coef[3], const jpeggpu_qtable_t *const qt[4],
                        uint8_t *out, size_t out_stride) {
   int rc = validate(d, qt); if (rc) return rc;
   if (!out) return fail(JPEGGPU_ERR_INVALID_ARG, "out is NULL");
   for (int c = 0; c < d->nbcomp; c++) if (!coef || !coef[c]) return fail(JPEGGPU_ERR_INVALID_ARG, "coef[%d] is NULL", c);
   rc = ensure_init(); if (rc) return rc;
   std::lock_guard<std::mutex> lk(g_mu);
   const size_t row_bytes = jpeggpu_raster_row_bytes(d), rows = jpeggpu_raster_rows(d);
   if (out_stride == 0) out_stride = row_bytes;
   if (out_stride < row_bytes) return fail(JPEGGPU_ERR_INVALID_ARG, "out_stride %zu < row bytes %zu", out_stride, row_bytes);
   size_t off[3] = { 0, 0, 0 }, in_bytes = 0;
   for (int c = 0; c < d->nbcomp; c++) { off[c] = in_bytes; in_bytes += jpeggpu_plane_blocks(d, c) * sizeof(jpeggpu_blocl16_t); }
   const size_t out_bytes = row_bytes * rows;
   rc = grow(&g.d_in, &g.d_in_cap, in_bytes, false); if (rc) return rc;
   rc = grow(&g.d_out, &g.d_out_cap, out_bytes, false); if (rc) return rc;
   rc = grow(&g.h_pin, &g.h_pin_cap, in_bytes > out_bytes ? in_bytes : out_bytes, true); if (rc) return rc;
   // uploaded once with pinned hipMemcpyAsync (north star); the caller's buffers are pageable
   // malloc memory (src/progressive.c:73-84), so they are gathered into the pinned area first.
   const void *dc[3] = { nullptr, nullptr, nullptr };
   for (int c = 0; c < d->nbcomp; c++) {
      memcpy((char *)g.h_pin + off[c], coef[c], jpeggpu_plane_blocks(d, c) * sizeof(jpeggpu_blocl16_t));
      dc[c] = (char *)g.d_in + off[c];
   }
   HIP_TRY(hipMemcpyAsync(g.d_in, g.h_pin, in_bytes, hipMemcpyHostToDevice, g.stream));
   rc = run_device(d, 1, dc, nullptr, qt, g.d_out, row_bytes, 0, g.stream); if (rc) return rc;
   HIP_TRY(hipMemcpyAsync(g.h_pin, g.d_out, out_bytes, hipMemcpyDeviceToHost, g.stream));
   HIP_TRY(hipStreamSynchronize(g.stream));
   for (size_t r = 0; r < rows; r++) memcpy(out + r * out_stride, (char *)g.h_pin + r * row_bytes, row_bytes);
   return 0;
}

// ---- stage-level array forms -----------------------------------------------------------------

static int stage_roundtrip(const void *in, size_t in_bytes, void *out, size_t out_bytes, hipError_t (*launch)(void *, void *, void *), void *ctx) {
   int rc = ensure_init(); if (rc) return rc;
   std::lock_guard<std::mutex> lk(g_mu);
   rc = grow(&g.d_in, &g.d_in_cap, in_bytes, false); if (rc) return rc;
   rc = grow(&g.d_out, &g.d_out_cap, out_bytes, false); if (rc) return rc;
   HIP_TRY(hipMemcpyAsync(g.d_in, in, in_bytes, hipMemcpyHostToDevice, g.stream));
   HIP_TRY(launch(g.d_in, g.d_out, ctx));
   HIP_TRY(hipMemcpyAsync(out, out_bytes ? g.d_out : g.d_in, out_bytes ? out_bytes : in_bytes, hipMemcpyDeviceToHost, g.stream));
   HIP_TRY(hipStreamSynchronize(g.stream));
   return 0;
}

int jpeggpu_iquant_blocks(jpeggpu_blocl16_t *blocks, size_t n, uint8_t s_start, uint8_t s_end, const jpeggpu_qtable_t *q) {
   if (n == 0) return 0;
   if (!blocks || !q) return fail(JPEGGPU_ERR_INVALID_ARG, "NULL argument");
   if (s_end > 63) return fail(JPEGGPU_ERR_INVALID_ARG, "s_end %u > 63", s_end);
   struct Ctx { size_t n; int s0, s1; uint32_t q[32]; } ctx = { n, s_start, s_end, {} };
   pack_qtable(q, ctx.q);
   return stage_roundtrip(blocks, n * 128, blocks, 0, [](void *di, void *, void *c) {
      Ctx *x = (Ctx *)c; return launch_iquant(di, x->n, x->s0, x->s1, x->q, g.stream); }, &ctx);
}

int jpeggpu_izz_blocks(const jpeggpu_blocl16_t *in, jpeggpu_bloct16_t *out, size_t n) {
   if (n == 0) return 0;
   if (!in || !out) return fail(JPEGGPU_ERR_INVALID_ARG, "NULL argument");
   return stage_roundtrip(in, n * 128, out, n * 128, [](void *di, void *dout, void *c) { return launch_izz(di, dout, *(size_t *)c, g.stream); }, &n);
}

int jpeggpu_idct_opt_blocks(const jpeggpu_bloct16_t *in, jpeggpu_bloctu8_t *out, size_t n) {
   if (n == 0) return 0;
   if (!in || !out) return fail(JPEGGPU_ERR_INVALID_ARG, "NULL argument");
   return stage_roundtrip(in, n * 128, out, n * 64, [](void *di, void *dout, void *c) { return launch_idct(di, dout, *(size_t *)c, g.stream); }, &n);
}

int jpeggpu_idct_blocks(const jpeggpu_bloct16_t *in, jpeggpu_bloctu8_t *out, size_t n) {
   if (n == 0) return 0;
   if (!in || !out) return fail(JPEGGPU_ERR_INVALID_ARG, "NULL argument");
   int rc = ensure_init(); if (rc) return rc;
   { std::lock_guard<std::mutex> lk(g_mu); rc = ensure_slow_coef(); if (rc) return rc; }
   return stage_roundtrip(in, n * 128, out, n * 64, [](void *di, void *dout, void *c) { return launch_idct_slow(di, dout, *(size_t *)c, g.d_slow_coef, g.stream); }, &n);
}

void jpeggpu_set_idct_fast(int fast) { g.idct_fast = fast != 0; }
int jpeggpu_get_idct_fast(void) { return g.idct_fast ? 1 : 0; }
void jpeggpu_idct_coef_table(float *out4096) { if (out4096) calc_coef_host(out4096); }

int jpeggpu_ycc2rgb_pixels(const uint8_t *y, const uint8_t *cb, const uint8_t *cr, jpeggpu_rgb_t *rgb, size_t n) {
   if (n == 0) return 0;
   if (!y || !cb || !cr || !rgb) return fail(JPEGGPU_ERR_INVALID_ARG, "NULL argument");
   int rc = ensure_init(); if (rc) return rc;
   std::lock_guard<std::mutex> lk(g_mu);
   rc = grow(&g.d_in, &g.d_in_cap, 3 * n, false); if (rc) return rc;
   rc = grow(&g.d_out, &g.d_out_cap, 3 * n, false); if (rc) return rc;
   uint8_t *di = (uint8_t *)g.d_in;
   HIP_TRY(hipMemcpyAsync(di, y, n, hipMemcpyHostToDevice, g.stream));
   HIP_TRY(hipMemcpyAsync(di + n, cb, n, hipMemcpyHostToDevice, g.stream));
   HIP_TRY(hipMemcpyAsync(di + 2 * n, cr, n, hipMemcpyHostToDevice, g.stream));
   HIP_TRY(launch_ycc2rgb(di, di + n, di + 2 * n, (uint8_t *)g.d_out, n, g.stream));
   HIP_TRY(hipMemcpyAsync(rgb, g.d_out, 3 * n, hipMemcpyDeviceToHost, g.stream));
   HIP_TRY(hipStreamSynchronize(g.stream));
   return 0;
}

}  // extern "C"

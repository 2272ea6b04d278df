// capi.cpp -- the C-ABI layer declared in include/jpeggpu.h.
//
// Host-side logic only: argument validation with the reference's error convention, tile planning,
// per-device contexts (stream, staging, "-f" table, option), the pipelined host-pointer form, and kernel
// launches.  No pixel arithmetic lives here and there is no CPU fallback: every entry point that
// produces data needs a HIP device.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/jpeggpu.h"
#include "kernels.h"

using namespace jpeggpu_k;

// One context per device and caller: its own stream, staging and option.  The process-wide entry points
// of the header (jpeggpu_backend_run ...) run on a default context bound by jpeggpu_init().
constexpr int kEventRing = 32;   // events of the host-pointer pipeline: recorded again only long after their unit has finished

struct jpeggpu_ctx_s {
   int device = -1;
   hipStream_t stream = nullptr;                 // default stream of the context (device-pointer forms with hip_stream == NULL)
   hipStream_t pipe[2] = { nullptr, nullptr };   // host-pointer pipeline: [0] uploads, [1] kernels and downloads
   hipEvent_t up_done[kEventRing] = {}, pipe_done[kEventRing] = {};   // unit k % kEventRing: uploaded / downloaded
   std::mutex mu;                                // host-pointer and stage-level entries: they share the staging below
   void *d_in = nullptr;  size_t d_in_cap = 0;
   void *d_out = nullptr; size_t d_out_cap = 0;
   void *h_in[2] = { nullptr, nullptr };  size_t h_in_cap[2] = { 0, 0 };     // pinned stripe staging
   void *h_out[2] = { nullptr, nullptr }; size_t h_out_cap[2] = { 0, 0 };
   std::atomic<int> idct_fast{ 1 };              // all_option.idct_fast of this context
   std::mutex slow_mu;
   std::atomic<float *> d_slow_coef{ nullptr };  // calc_coef's table on the device, built on first use
   void *static_scratch = nullptr;               // device address of kernels.hip's static staging array on this device
};

namespace {

constexpr size_t kDeviceStagingBytes = (size_t)1 << 30;   // device-side staging of one host-pointer call (run_host)
constexpr uint32_t kPipeDepth = 3;                        // units in flight at most

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
   va_list ap;
   va_start(ap, fmt);
   vsnprintf(g_err, sizeof g_err, fmt, ap);
   va_end(ap);
   return code;
}

#define HIP_TRY(expr)                                                                            \
   do {                                                                                          \
      hipError_t e_ = (expr);                                                                    \
      if (e_ != hipSuccess) {                                                                    \
         (void)hipGetLastError();   /* reported here: must not resurface from a later launch's hipGetLastError() */ \
         return fail(JPEGGPU_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));                    \
      }                                                                                          \
   } while (0)

// ---- tuning knobs: read from the environment ONCE (first use), never per launch --------------------
struct Tuning {
   uint32_t target_blocks = 0;   // JPEGGPU_TARGET_BLOCKS (0 = default 256)
   uint32_t tile_mcus = 0;       // JPEGGPU_TILE_MCUS: any T (not a power of two: generic pixel phase)
   bool no_uniform = false;      // JPEGGPU_NO_UNIFORM: per-lane prologue everywhere
   bool no_linear = false;       // JPEGGPU_NO_LINEAR: tiles never run on over the end of an MCU row
   bool no_restage = false;      // JPEGGPU_NO_RESTAGE: never the 16-byte-store pixel phase (kernels.hip, RS)
   size_t stripe_bytes = 0;      // JPEGGPU_STRIPE_BYTES: stripe size of the host-pointer pipeline
   size_t diag_lds_bytes = 0;    // JPEGGPU_DIAG_LDS_BYTES: larger LDS allocation = fewer resident workgroups per CU
   size_t staging_bytes = 0;     // JPEGGPU_DEVICE_STAGING_BYTES: device-side staging of one host-pointer call (0 = 1 GiB)
   uint32_t pipe_depth = 0;      // JPEGGPU_PIPE_DEPTH: units of the host-pointer pipeline in flight at most (0 = default)
};
Tuning g_tune;
std::once_flag g_tune_once;

void read_tuning() {
   Tuning t;
   if (const char *e = getenv("JPEGGPU_TARGET_BLOCKS")) t.target_blocks = (uint32_t)atoi(e);
   if (const char *e = getenv("JPEGGPU_TILE_MCUS")) t.tile_mcus = (uint32_t)atoi(e);
   t.no_uniform = getenv("JPEGGPU_NO_UNIFORM") != nullptr;
   t.no_linear = getenv("JPEGGPU_NO_LINEAR") != nullptr;
   t.no_restage = getenv("JPEGGPU_NO_RESTAGE") != nullptr;
   if (const char *e = getenv("JPEGGPU_STRIPE_BYTES")) t.stripe_bytes = (size_t)atoll(e);
   if (const char *e = getenv("JPEGGPU_DIAG_LDS_BYTES")) t.diag_lds_bytes = (size_t)atol(e);
   if (const char *e = getenv("JPEGGPU_DEVICE_STAGING_BYTES")) t.staging_bytes = (size_t)atoll(e);
   if (const char *e = getenv("JPEGGPU_PIPE_DEPTH")) t.pipe_depth = (uint32_t)atoi(e);
   g_tune = t;
}
const Tuning &tuning() { std::call_once(g_tune_once, read_tuning); return g_tune; }

// ---- contexts ----------------------------------------------------------------------------------------
std::mutex g_mu;                        // creation / destruction of the default context
std::atomic<jpeggpu_ctx_s *> g_default{ nullptr };
std::atomic<int> g_idct_fast{ 1 };      // the process-wide option (jpeggpu_set_idct_fast); survives jpeggpu_shutdown

void ctx_delete(jpeggpu_ctx_s *c);

int ctx_new(int device, jpeggpu_ctx_s **out) {
   int n = 0;
   hipError_t e = hipGetDeviceCount(&n);
   if (e != hipSuccess || n <= 0) return fail(JPEGGPU_ERR_NO_DEVICE, "no HIP device (%s)", e == hipSuccess ? "count is 0" : hipGetErrorString(e));
   if (device < 0 || device >= n) return fail(JPEGGPU_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, n);
   HIP_TRY(hipSetDevice(device));
   jpeggpu_ctx_s *c = new jpeggpu_ctx_s;
   c->device = device;
   hipError_t rc = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
   for (int i = 0; i < 2 && rc == hipSuccess; i++) rc = hipStreamCreateWithFlags(&c->pipe[i], hipStreamNonBlocking);
   for (int i = 0; i < kEventRing && rc == hipSuccess; i++) {
      rc = hipEventCreateWithFlags(&c->pipe_done[i], hipEventDisableTiming);
      if (rc == hipSuccess) rc = hipEventCreateWithFlags(&c->up_done[i], hipEventDisableTiming);
   }
   if (rc != hipSuccess) {
      for (int i = 0; i < kEventRing; i++) {
         if (c->pipe_done[i]) (void)hipEventDestroy(c->pipe_done[i]);
         if (c->up_done[i]) (void)hipEventDestroy(c->up_done[i]);
      }
      for (int i = 0; i < 2; i++) if (c->pipe[i]) (void)hipStreamDestroy(c->pipe[i]);
      if (c->stream) (void)hipStreamDestroy(c->stream);
      delete c;
      return fail(JPEGGPU_ERR_HIP, "stream/event creation on device %d: %s", device, hipGetErrorString(rc));
   }
   c->idct_fast.store(g_idct_fast.load());
   // jpeggpu_backend_run_noalloc: the address of the static staging array on THIS device, looked up here, at set-up
   hipError_t se = static_scratch(device, &c->static_scratch);
   if (se != hipSuccess) {
      ctx_delete(c);
      return fail(JPEGGPU_ERR_HIP, "hipGetSymbolAddress(scratch) on device %d: %s", device, hipGetErrorString(se));
   }
   *out = c;
   return 0;
}

void ctx_delete(jpeggpu_ctx_s *c) {
   if (!c) return;
   (void)hipSetDevice(c->device);
   (void)hipStreamSynchronize(c->stream);
   for (int i = 0; i < 2; i++) {
      (void)hipStreamSynchronize(c->pipe[i]);
      if (c->h_in[i]) (void)hipHostFree(c->h_in[i]);
      if (c->h_out[i]) (void)hipHostFree(c->h_out[i]);
      (void)hipStreamDestroy(c->pipe[i]);
   }
   for (int i = 0; i < kEventRing; i++) { (void)hipEventDestroy(c->pipe_done[i]); (void)hipEventDestroy(c->up_done[i]); }
   if (c->d_in) (void)hipFree(c->d_in);
   if (c->d_out) (void)hipFree(c->d_out);
   if (float *t = c->d_slow_coef.load()) (void)hipFree(t);
   (void)hipStreamDestroy(c->stream);
   delete c;
}

// HIP's current device is a per-THREAD setting: every entry point binds the calling thread to the context's device.
int bind(jpeggpu_ctx_s *c) {
   if (!c) return fail(JPEGGPU_ERR_INVALID_ARG, "ctx is NULL");
   HIP_TRY(hipSetDevice(c->device));
   return 0;
}

int default_ctx(jpeggpu_ctx_s **out) {
   jpeggpu_ctx_s *c = g_default.load();
   if (!c) {
      int rc = jpeggpu_init(0);
      if (rc) return rc;
      c = g_default.load();
   }
   *out = c;
   return bind(c);
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

int ensure_slow_coef(jpeggpu_ctx_s *c, const float **out) {
   float *t = c->d_slow_coef.load(std::memory_order_acquire);
   if (!t) {
      std::lock_guard<std::mutex> lk(c->slow_mu);
      t = c->d_slow_coef.load(std::memory_order_relaxed);
      if (!t) {
         static float table[4096];
         static std::once_flag once;
         std::call_once(once, [] { calc_coef_host(table); });
         HIP_TRY(hipMalloc((void **)&t, sizeof table));
         hipError_t e = hipMemcpy(t, table, sizeof table, hipMemcpyHostToDevice);
         if (e != hipSuccess) { (void)hipFree(t); return fail(JPEGGPU_ERR_HIP, "upload of the idct table: %s", hipGetErrorString(e)); }
         c->d_slow_coef.store(t, std::memory_order_release);
      }
   }
   *out = t;
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
   for (int c = 0; c < d->nbcomp; c++) {
      const jpeggpu_component_t &k = d->comp[c];
      if (k.hs == 0 || k.vs == 0 || k.hs > 4 || k.vs > 4)
         return fail(JPEGGPU_ERR_SAMPLING, "component %d: sampling factors %ux%u out of 1..4", c, k.hs, k.vs);
      if (d->max_h % k.hs || d->max_v % k.vs || k.hs > d->max_h || k.vs > d->max_v)
         return fail(JPEGGPU_ERR_SAMPLING, "component %d: %ux%u does not divide max %ux%u", c, k.hs, k.vs, d->max_h, d->max_v);
      if (d->apply_iquant && (k.idq > 3 || !qt || !qt[k.idq]))
         return fail(JPEGGPU_ERR_NO_QT, "Pas de table de quantification pour la composante %d", c);
   }
   if ((uint64_t)d->nbmcuH * d->max_h * 8 < d->width)
      return fail(JPEGGPU_ERR_INVALID_ARG, "width %u exceeds nbmcuH*max_h*8 = %llu", d->width,
                  (unsigned long long)d->nbmcuH * d->max_h * 8);
   return 0;
}

// Tile planning: T MCUs per workgroup, one thread per block.
void plan_backend(const jpeggpu_frame_desc_t *d, const jpeggpu_qtable_t *const qt[4], BackendParams &p, LaunchDims &ld) {
   const Tuning &tn = tuning();
   memset(&p, 0, sizeof p);
   uint32_t per_mcu = 0;
   for (int c = 0; c < d->nbcomp; c++) per_mcu += (uint32_t)d->comp[c].hs * d->comp[c].vs;
   uint32_t target_blocks = tn.target_blocks ? tn.target_blocks : 256;
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
   if (!tn.target_blocks && !wave_uniform(T) && T * 2 * per_mcu <= kMaxThreads && wave_uniform(T * 2)) T *= 2;
   // a frame narrower than the tile keeps a power-of-two tile (the unused lanes idle): the sub-sampling-specific
   // kernels turn the item index into (row, quad) with a shift and a mask
   while (T / 2 >= d->nbmcuH && T > 1) T /= 2;
   if (tn.tile_mcus >= 1 && tn.tile_mcus * per_mcu <= kMaxThreads) T = tn.tile_mcus;
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
   if (tn.no_uniform) p.wave_uniform = 0;
   if (d->apply_iquant)
      for (int i = 0; i < 4; i++)
         if (qt[i]) pack_qtable(qt[i], p.qt[i]);
   auto log2_exact = [](uint32_t v) { uint32_t s = 0; while ((1u << s) < v) s++; return (1u << s) == v ? s : 0xFFu; };
   bool pow2 = log2_exact((uint32_t)d->max_h * 2) != 0xFFu;
   for (int c = 0; c < d->nbcomp; c++) { const uint32_t s = log2_exact(d->comp[c].hs); p.hs_shift[c] = s == 0xFFu ? 0 : s; pow2 = pow2 && s != 0xFFu; }
   p.mcu_q_shift = pow2 ? log2_exact((uint32_t)d->max_h * 2) : 0;
   p.nbmcuH = d->nbmcuH; p.total_mcus = d->nbmcuH * d->nbmcuV;
   p.inv_nbmcuH = (uint32_t)((1ull << 32) / (d->nbmcuH < 2 ? 2u : (uint32_t)d->nbmcuH));
   ld.tiles_per_row = (d->nbmcuH + T - 1) / T;
   ld.mcu_rows = d->nbmcuV;
   // Linear tiling: when an MCU row is not a whole number of tiles (1920 pixels of 4:2:0 = 120 MCUs = 1.875 tiles) the
   // last tile of every row would run with idle lanes; instead the frame's MCUs are tiled in raster order and a tile
   // may run on into the next MCU row (one wrap point at most, since nbmcuH > T).
   if (!tn.no_linear && pow2 && p.wave_uniform && p.wq_shift != 0xFFFFFFFFu && p.wq_shift >= 6 && fast_colour_layout(p) &&
       d->nbmcuH > T && d->nbmcuH % T != 0 && (uint64_t)d->nbmcuH * d->nbmcuV < (1ull << 31)) {
      p.linear = 1;
      ld.tiles_per_row = (p.total_mcus + T - 1) / T;
      ld.mcu_rows = 1;
   }
   ld.threads = (p.B + 63) / 64 * 64;                 // one thread per block
   ld.lds_bytes = (size_t)ld.threads * 64 + 512;
   if (tn.diag_lds_bytes > ld.lds_bytes && tn.diag_lds_bytes <= 64 * 1024) ld.lds_bytes = tn.diag_lds_bytes;
}

int run_device(jpeggpu_ctx_s *ctx, const jpeggpu_frame_desc_t *d, uint32_t nframes, const void *const d_coef[3], const size_t cfs[3],
               const jpeggpu_qtable_t *const qt[4], void *d_out, size_t out_stride, size_t out_frame_stride, hipStream_t s,
               bool may_alloc = true, int idct_fast = -1) {   // idct_fast: -1 = read the context's option now, else the caller's snapshot
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
#if defined(JM_DIAG_STAMP)   // diagnostic builds only (tools/stamp.py): where the kernel leaves its phase time stamps
   if (const char *e = getenv("JPEGGPU_DIAG_PTR")) p.diag = (void *)strtoull(e, nullptr, 0);
#endif
   if (nframes == 0) return 0;
   if (!(idct_fast < 0 ? ctx->idct_fast.load() : idct_fast)) {   // the option is read ONCE per public call
      if (!may_alloc && !ctx->d_slow_coef.load())
         return fail(JPEGGPU_ERR_INVALID_ARG, "idct_fast == 0 needs its coefficient table on the device: make one ordinary call in this mode first");
      int rc = ensure_slow_coef(ctx, &p.slow_coef); if (rc) return rc;
   }
   if (nframes > 65535 || d->nbmcuV > 65535) return fail(JPEGGPU_ERR_INVALID_ARG, "nframes / nbmcuV above 65535");
   // 4:4:4, uncropped, in 64-MCU tiles (whole tiles per MCU row, or linear tiling with an even number of MCUs per row) and a
   // 16-byte aligned raster: row-pair units with 16-byte stores (kernels.hip, RS = 1); the planes then lie in LDS interleaved
   // by row pair
   const bool aligned16 = !(((uintptr_t)d_out | out_stride | out_frame_stride) & 15);
   if (!tuning().no_restage && !p.slow_coef && p.wave_uniform && d->nbcomp == 3 && p.T == 64 && ld.threads == 192 && fast_colour_layout(p) &&
       p.hf[1] == 1 && p.vf[1] == 1 && d->width == (uint32_t)d->nbmcuH * 8 && aligned16 &&
       (p.linear ? (d->nbmcuH % 2 == 0 && d->nbmcuH > 64) : d->nbmcuH % 64 == 0)) {
      p.restage = p.linear ? 3 : 1;
      p.plane_off[0] = 0; p.plane_off[1] = 1024; p.plane_off[2] = 2048;
   }
   // 4:2:0 (Y 2x2, Cb and Cr 1x1 on one table), uncropped width, 16-byte aligned raster: 32-MCU tiles of 192 lanes, Cb and
   // Cr sharing the third wave (kernels.hip, RS = 2); frames whose MCU rows are not a whole number of tiles are tiled linearly
   if (!tuning().no_restage && !tuning().tile_mcus && !tuning().target_blocks && !tuning().no_uniform && !p.slow_coef && d->nbcomp == 3 &&
       d->comp[0].hs == 2 && d->comp[0].vs == 2 && d->comp[1].hs == 1 && d->comp[1].vs == 1 && d->comp[2].hs == 1 && d->comp[2].vs == 1 && d->max_h == 2 && d->max_v == 2 &&
       (!d->apply_iquant || d->comp[1].idq == d->comp[2].idq) && d->nbmcuH >= 32 && d->width == (uint32_t)d->nbmcuH * 16 && aligned16 &&
       (uint64_t)d->nbmcuH * d->nbmcuV < (1ull << 31) && (d->nbmcuH % 32 == 0 || !tuning().no_linear)) {
      p.restage = 2;
      p.T = 32; p.B = 192;
      p.linear = d->nbmcuH % 32 != 0;
      ld.threads = 192;
      ld.lds_bytes = 12288;
      ld.tiles_per_row = p.linear ? (p.total_mcus + 31) / 32 : d->nbmcuH / 32;
      ld.mcu_rows = p.linear ? 1 : d->nbmcuV;
   }
   if (p.slow_coef && p.linear) { p.linear = 0; ld.tiles_per_row = (d->nbmcuH + p.T - 1) / p.T; ld.mcu_rows = d->nbmcuV; }   // "-f": generic pixel phase
   ld.frames = nframes;
   HIP_TRY(launch_backend(p, ld, s));
   return 0;
}

bool is_pinned_host(const void *p) {
   hipPointerAttribute_t a;
   if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
   return a.type == hipMemoryTypeHost;
}
// [p, p + bytes) is pinned host memory: its first AND its last byte are (a buffer that was pinned or registered only in
// part must not be handed to the DMA engines in place; include/jpeggpu.h: the WHOLE range a call touches has to be pinned)
bool range_pinned(const void *p, size_t bytes) {
   return bytes == 0 || (is_pinned_host(p) && is_pinned_host((const char *)p + bytes - 1));
}
// the frames first, first + step, ... (count of them), `bytes` each, `stride` apart: every one of them when they are few,
// an even sample of 256 of them otherwise (always the first and the last)
bool frames_pinned(const void *base, size_t stride, size_t bytes, uint32_t first, uint32_t step, uint32_t count) {
   const uint32_t probes = count < 256 ? count : 256;
   for (uint32_t i = 0; i < probes; i++) {
      const uint32_t k = probes > 1 ? (uint32_t)((uint64_t)i * (count - 1) / (probes - 1)) : 0;
      if (!range_pinned((const char *)base + (size_t)(first + k * step) * stride, bytes)) return false;
   }
   return true;
}

// The host-pointer forms.  Every frame is cut into stripes of whole MCU rows (a stripe is a frame of its own to the
// kernel: replicate upsampling has no halo, src/decoder_utils.c:72-80); the stripes of all frames of the call form one
// sequence of units.  The link to the host is full duplex (profiles/r02_probe_pcie.txt: 57 GB/s up and 57 GB/s down
// at the same time), so the uploads of all units queue back to back on one stream and nothing else ever sits between
// them; a second stream takes, per unit, the wait for that unit's upload, its kernel and its download.  Upload of unit
// k+1, kernel of unit k and download of unit k-1 overlap -- across the frames of a batch as well -- and the call costs
// the upload time plus one unit's kernel and download.  Caller memory that
// is already pinned (jpeggpu_host_alloc, hipHostMalloc, the caller's own hipHostRegister) is read and written by the DMA engines
// in place; pageable memory (the reference's malloc'ed buffers, src/progressive.c:73-84) goes through two pinned
// buffers per direction, filled and emptied by host threads while the other unit is in flight.
//
// Frames first, first + step, ... (count of them) of the caller's arrays: plane c of frame f at coef[c] + f * cfs[c],
// its raster at out + f * out_frame_stride.
int run_host(jpeggpu_ctx_s *ctx, const jpeggpu_frame_desc_t *d, uint32_t first, uint32_t step, uint32_t count,
             const jpeggpu_blocl16_t *const coef[3], const size_t cfs[3], const jpeggpu_qtable_t *const qt[4],
             uint8_t *out, size_t out_stride, size_t out_frame_stride) {
   const size_t row_bytes = jpeggpu_raster_row_bytes(d), rows = jpeggpu_raster_rows(d);
   if (out_stride == 0) out_stride = row_bytes;
   if (out_stride < row_bytes) return fail(JPEGGPU_ERR_INVALID_ARG, "out_stride %zu < row bytes %zu", out_stride, row_bytes);
   if (out_frame_stride == 0) out_frame_stride = out_stride * rows;
   if (count == 0) return 0;
   std::lock_guard<std::mutex> lk(ctx->mu);
   const int idct_fast = ctx->idct_fast.load();              // one snapshot for every unit of this call
   hipStream_t s_up = ctx->pipe[0], s_down = ctx->pipe[1];
   // whatever way this function is left, nothing may still be copying from / into the caller's memory or the staging
   struct Drain {
      jpeggpu_ctx_s *c; bool armed = true;
      ~Drain() { if (armed) { (void)hipStreamSynchronize(c->pipe[0]); (void)hipStreamSynchronize(c->pipe[1]); } }
   } drain{ ctx };
   size_t fstride[3] = { 0, 0, 0 }, row_in[3] = { 0, 0, 0 }, in_per_mcu_row = 0;
   for (int c = 0; c < d->nbcomp; c++) {
      fstride[c] = (cfs && cfs[c]) ? cfs[c] : jpeggpu_plane_blocks(d, c) * sizeof(jpeggpu_blocl16_t);
      row_in[c] = (size_t)d->nbmcuH * d->comp[c].hs * d->comp[c].vs * sizeof(jpeggpu_blocl16_t);   // one MCU row of plane c
      in_per_mcu_row += row_in[c];
   }
   const size_t band = (size_t)d->max_v * 8;                 // raster rows per MCU row
   bool in_pinned = true;
   for (int c = 0; c < d->nbcomp; c++)
      in_pinned = in_pinned && frames_pinned(coef[c], fstride[c], jpeggpu_plane_blocks(d, c) * sizeof(jpeggpu_blocl16_t), first, step, count);
   const bool out_pinned = frames_pinned(out, out_frame_stride, (rows - 1) * out_stride + row_bytes, first, step, count);
   // stripe height: ~16 MiB of coefficients when the DMA engines work in place, ~8 MiB when a host thread stages
   // (tools/sweep_stripe.py, profiles/r02_sweep_stripe.txt), at least one MCU row, at most the frame
   const size_t target = tuning().stripe_bytes ? tuning().stripe_bytes : ((size_t)(in_pinned && out_pinned ? 16 : 8) << 20);
   uint32_t srows = (uint32_t)(target / (in_per_mcu_row ? in_per_mcu_row : 1));
   if (srows < 1) srows = 1;
   if (srows > d->nbmcuV) srows = d->nbmcuV;
   const uint32_t nstripes = (d->nbmcuV + srows - 1) / srows;
   if ((uint64_t)nstripes * count > 0xFFFFFFF0ull) return fail(JPEGGPU_ERR_INVALID_ARG, "too many stripes in one call");
   const uint32_t nunits = nstripes * count;
   const size_t slot_in = (size_t)srows * in_per_mcu_row, slot_out = (size_t)srows * band * row_bytes;
   // Device side: every unit of the call has a slot of its own (coefficients + raster band) as long as that stays within
   // kDeviceStagingBytes; a longer call goes round the slots, and waits for everything queued each time it starts over
   // (1 GiB of traffic apart: 13 ms of transfers for one drained pipeline).  No finer-grained reuse: an upload stream that
   // waits for an event of the download stream more than halves the throughput (the two directions stop overlapping;
   // tools/probe_pcie.hip, profiles/r02_probe_pcie.txt).
   uint32_t nslots = (uint32_t)std::min<size_t>((tuning().staging_bytes ? tuning().staging_bytes : kDeviceStagingBytes) / (slot_in + slot_out), 0x7FFFFFFF);
   if (nslots < 2) nslots = 2;
   if (nslots > nunits) nslots = nunits;
   int rc = grow(&ctx->d_in, &ctx->d_in_cap, nslots * slot_in, false); if (rc) return rc;
   rc = grow(&ctx->d_out, &ctx->d_out_cap, nslots * slot_out, false); if (rc) return rc;
   if (!in_pinned) for (int b = 0; b < 2; b++) { rc = grow(&ctx->h_in[b], &ctx->h_in_cap[b], slot_in, true); if (rc) return rc; }
   if (!out_pinned) for (int b = 0; b < 2; b++) { rc = grow(&ctx->h_out[b], &ctx->h_out_cap[b], slot_out, true); if (rc) return rc; }

   uint32_t depth = tuning().pipe_depth ? tuning().pipe_depth : kPipeDepth;
   if (depth > (uint32_t)kEventRing - 2) depth = kEventRing - 2;
   struct Unit { uint32_t frame, m0, n; };
   auto unit = [&](uint32_t k) {
      const uint32_t fi = k / nstripes, m0 = (k - fi * nstripes) * srows;
      return Unit{ first + fi * step, m0, (m0 + srows <= d->nbmcuV ? srows : d->nbmcuV - m0) };
   };
   auto copy_out = [&](uint32_t k) {   // staged raster band of unit k -> the caller's rows
      const Unit u = unit(k);
      const uint8_t *src = (const uint8_t *)ctx->h_out[k & 1];
      uint8_t *dst = out + (size_t)u.frame * out_frame_stride + (size_t)u.m0 * band * out_stride;
      if (out_stride == row_bytes) memcpy(dst, src, (size_t)u.n * band * row_bytes);
      else for (size_t r = 0; r < (size_t)u.n * band; r++) memcpy(dst + r * out_stride, src + r * row_bytes, row_bytes);
   };
   // Pageable output of some size: a second host thread empties the download staging while this one fills the upload
   // staging (one thread copies ~29 GB/s, profiles/r02_sweep_stripe.txt, and would otherwise move in + out = 1.5x
   // the coefficients by itself).  It lives for this call only and is joined on every way out.
   struct OutCopier {
      std::thread th;
      std::atomic<uint32_t> enqueued{ 0 }, copied{ 0 };        // units whose download is queued / is in the caller's rows
      std::atomic<int> err{ 0 };
      std::atomic<bool> stop{ false };
      ~OutCopier() { stop.store(true); if (th.joinable()) th.join(); }
   } oc;
   bool helper = !out_pinned && nunits >= 3 && row_bytes * rows * count >= ((size_t)16 << 20);
   if (helper) {
      try {
         oc.th = std::thread([&, nunits] {
            if (hipSetDevice(ctx->device) != hipSuccess) { oc.err.store((int)hipErrorInvalidDevice); return; }
            for (uint32_t k = 0; k < nunits; k++) {
               while (oc.enqueued.load(std::memory_order_acquire) <= k) { if (oc.stop.load()) return; std::this_thread::yield(); }
               hipError_t e = hipEventSynchronize(ctx->pipe_done[k % kEventRing]);
               if (e != hipSuccess) { oc.err.store((int)e); return; }
               copy_out(k);
               oc.copied.store(k + 1, std::memory_order_release);
            }
         });
      } catch (const std::system_error &) { helper = false; }    // no thread to be had: this thread copies, as for small calls
   }
   // Pageable input of some size: a third host thread takes the second half of every copy into the upload staging (same
   // reason as above: one thread copies ~29 GB/s, the link takes 57).  One task at a time, handed over through two counters.
   struct InHelper {
      std::thread th;
      std::atomic<uint64_t> posted{ 0 }, done{ 0 };
      std::atomic<bool> stop{ false };
      char *dst = nullptr; const char *src = nullptr; size_t n = 0;   // the task: written before posted is raised, read after
      void post(char *d_, const char *s_, size_t n_) { dst = d_; src = s_; n = n_; posted.fetch_add(1, std::memory_order_release); }
      void wait() const { while (done.load(std::memory_order_acquire) < posted.load(std::memory_order_relaxed)) std::this_thread::yield(); }
      ~InHelper() { stop.store(true); if (th.joinable()) th.join(); }
   } ih;
   bool in_helper = false;
   if (!in_pinned && (uint64_t)in_per_mcu_row * d->nbmcuV * count >= ((uint64_t)16 << 20)) {
      try {
         ih.th = std::thread([&ih] {
            uint64_t seen = 0;
            for (;;) {
               while (ih.posted.load(std::memory_order_acquire) == seen) { if (ih.stop.load()) return; std::this_thread::yield(); }
               seen++;
               memcpy(ih.dst, ih.src, ih.n);
               ih.done.store(seen, std::memory_order_release);
            }
         });
         in_helper = true;
      } catch (const std::system_error &) { in_helper = false; }
   }
   auto stage_in = [&](char *dst, const char *src, size_t n) {   // caller memory -> upload staging
      if (in_helper && n >= ((size_t)1 << 20)) {
         const size_t half = (n / 2) & ~(size_t)63;
         ih.post(dst + half, src + half, n - half);
         memcpy(dst, src, half);
         ih.wait();
      } else {
         memcpy(dst, src, n);
      }
   };
   // events are used in a ring: a hipStreamWaitEvent refers to the record that preceded it, recording the event again
   // for a later unit does not disturb it
   for (uint32_t k = 0; k < nunits; k++) {
      const int b = (int)(k & 1), e = (int)(k % kEventRing);   // staging buffer, events
      const uint32_t sl = k % nslots;
      const Unit u = unit(k);
      if (!in_pinned && k >= 2) HIP_TRY(hipEventSynchronize(ctx->up_done[(k - 2) % kEventRing]));   // unit k-2 has left this staging buffer
      if (k != 0 && sl == 0) HIP_TRY(hipStreamSynchronize(s_down));   // the ring starts over: every kernel so far has read its slot
      // At most <depth> units in flight: the host waits for the download of unit k - depth before it queues unit k.  Measured
      // (profiles/r02_host_pipeline_depth.txt): with 64 frame-sized units queued as fast as the host can, the enqueue calls
      // start to block after some 20 units and the whole pipeline falls from 76 to 45 GB/s; depth 2-3 keeps 76.
      if (k >= depth) HIP_TRY(hipEventSynchronize(ctx->pipe_done[(k - depth) % kEventRing]));
      const void *dc[3] = { nullptr, nullptr, nullptr };
      size_t hoff = 0;
      char *const d_slot = (char *)ctx->d_in + (size_t)sl * slot_in;
      for (int c = 0; c < d->nbcomp; c++) {
         const char *src = (const char *)coef[c] + (size_t)u.frame * fstride[c] + (size_t)u.m0 * row_in[c];
         if (!in_pinned) {
            stage_in((char *)ctx->h_in[b] + hoff, src, (size_t)u.n * row_in[c]);
            src = (const char *)ctx->h_in[b] + hoff;
         }
         HIP_TRY(hipMemcpyAsync(d_slot + hoff, src, (size_t)u.n * row_in[c], hipMemcpyHostToDevice, s_up));
         dc[c] = d_slot + hoff;
         hoff += (size_t)u.n * row_in[c];                        // a multiple of 128: the next plane stays 16-byte aligned
      }
      HIP_TRY(hipEventRecord(ctx->up_done[e], s_up));
      if (!out_pinned && k >= 2) {                               // unit k-2 sits in the staging buffer unit k downloads into
         if (helper) {
            while (oc.copied.load(std::memory_order_acquire) < k - 1) {
               if (oc.err.load()) return fail(JPEGGPU_ERR_HIP, "download of a stripe: %s", hipGetErrorString((hipError_t)oc.err.load()));
               std::this_thread::yield();
            }
         } else {
            HIP_TRY(hipEventSynchronize(ctx->pipe_done[(k - 2) % kEventRing]));
            copy_out(k - 2);
         }
      }
      HIP_TRY(hipStreamWaitEvent(s_down, ctx->up_done[e], 0));
      jpeggpu_frame_desc_t sd = *d;
      sd.nbmcuV = u.n;
      uint8_t *d_band = (uint8_t *)ctx->d_out + (size_t)sl * slot_out;
      rc = run_device(ctx, &sd, 1, dc, nullptr, qt, d_band, row_bytes, 0, s_down, true, idct_fast); if (rc) return rc;
      uint8_t *dst = out + (size_t)u.frame * out_frame_stride + (size_t)u.m0 * band * out_stride;
      if (!out_pinned) HIP_TRY(hipMemcpyAsync(ctx->h_out[b], d_band, (size_t)u.n * band * row_bytes, hipMemcpyDeviceToHost, s_down));
      else if (out_stride == row_bytes) HIP_TRY(hipMemcpyAsync(dst, d_band, (size_t)u.n * band * row_bytes, hipMemcpyDeviceToHost, s_down));
      else HIP_TRY(hipMemcpy2DAsync(dst, out_stride, d_band, row_bytes, row_bytes, (size_t)u.n * band, hipMemcpyDeviceToHost, s_down));
      HIP_TRY(hipEventRecord(ctx->pipe_done[e], s_down));
      oc.enqueued.store(k + 1, std::memory_order_release);
   }
   if (helper) {
      oc.th.join();                                           // it has seen every unit queued, so it runs to the end
      if (oc.err.load()) return fail(JPEGGPU_ERR_HIP, "download of a stripe: %s", hipGetErrorString((hipError_t)oc.err.load()));
      HIP_TRY(hipStreamSynchronize(s_down));
   } else {
      // drain, oldest unit first
      for (uint32_t k = (nunits >= 2 ? nunits - 2 : 0); k < nunits; k++) {
         HIP_TRY(hipEventSynchronize(ctx->pipe_done[k % kEventRing]));
         if (!out_pinned) copy_out(k);
      }
   }
   HIP_TRY(hipStreamSynchronize(s_up));                       // long since idle: every download waited for its upload
   drain.armed = false;                                       // both streams have been waited for
   return 0;
}

int ctx_backend_run_batch(jpeggpu_ctx_s *ctx, const jpeggpu_frame_desc_t *d, uint32_t nframes, const jpeggpu_blocl16_t *const coef[3],
                          const size_t cfs[3], const jpeggpu_qtable_t *const qt[4], uint8_t *out, size_t out_stride, size_t out_frame_stride) {
   int rc = validate(d, qt); if (rc) return rc;
   if (!out) return fail(JPEGGPU_ERR_INVALID_ARG, "out is NULL");
   for (int c = 0; c < d->nbcomp; c++) {
      if (!coef || !coef[c]) return fail(JPEGGPU_ERR_INVALID_ARG, "coef[%d] is NULL", c);
      if (cfs && cfs[c] & 15) return fail(JPEGGPU_ERR_INVALID_ARG, "coef_frame_stride[%d] must be a multiple of 16", c);
   }
   return run_host(ctx, d, 0, 1, nframes, coef, cfs, qt, out, out_stride, out_frame_stride);
}

int ctx_backend_run(jpeggpu_ctx_s *ctx, const jpeggpu_frame_desc_t *d, const jpeggpu_blocl16_t *const coef[3],
                    const jpeggpu_qtable_t *const qt[4], uint8_t *out, size_t out_stride) {
   return ctx_backend_run_batch(ctx, d, 1, coef, nullptr, qt, out, out_stride, 0);
}

}  // namespace

extern "C" {

const char *jpeggpu_last_error(void) { return g_err; }

// used by decode_api.cpp / batch_api.cpp (same library); not part of the public headers
int jpeggpu_internal_range_pinned(const void *p, size_t bytes) { return range_pinned(p, bytes) ? 1 : 0; }   // batch_api.cpp, decode_api.cpp
void jpeggpu_internal_set_error(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg ? msg : ""); }
int jpeggpu_internal_ensure_init(void) { jpeggpu_ctx_s *c; return default_ctx(&c); }   // also binds the calling thread to the device
int jpeggpu_internal_device(void) { jpeggpu_ctx_s *c = g_default.load(); return c ? c->device : 0; }
// tests and tools change the knobs inside one process: they ask for the environment to be read again
void jpeggpu_internal_reload_tuning(void) { (void)tuning(); read_tuning(); }
size_t jpeggpu_internal_stripe_bytes(void) { return tuning().stripe_bytes; }   // decode_api.cpp: rows handed on per step of a streamed baseline file

int jpeggpu_init(int device) {
   std::lock_guard<std::mutex> lk(g_mu);
   (void)tuning();
   if (jpeggpu_ctx_s *c = g_default.load()) {
      if (c->device == device) return bind(c);
      return fail(JPEGGPU_ERR_INVALID_ARG, "the default context is bound to device %d; use jpeggpu_ctx_create() for further devices", c->device);
   }
   jpeggpu_ctx_s *c = nullptr;
   int rc = ctx_new(device, &c);
   if (rc) return rc;
   g_default.store(c);
   g_err[0] = 0;
   return 0;
}

void jpeggpu_internal_pool_drain(void);   // decode_api.cpp: buffers of closed files kept for the next one

void jpeggpu_shutdown(void) {
   std::lock_guard<std::mutex> lk(g_mu);
   jpeggpu_internal_pool_drain();
   jpeggpu_ctx_s *c = g_default.exchange(nullptr);
   ctx_delete(c);
}

int jpeggpu_ctx_create(int device, jpeggpu_ctx_t **ctx) {
   if (!ctx) return fail(JPEGGPU_ERR_INVALID_ARG, "ctx is NULL");
   *ctx = nullptr;
   (void)tuning();
   return ctx_new(device, ctx);
}
void jpeggpu_ctx_destroy(jpeggpu_ctx_t *ctx) { ctx_delete(ctx); }
int jpeggpu_ctx_device(const jpeggpu_ctx_t *ctx) { return ctx ? ctx->device : -1; }
void jpeggpu_ctx_set_idct_fast(jpeggpu_ctx_t *ctx, int fast) { if (ctx) ctx->idct_fast.store(fast != 0); }
int jpeggpu_ctx_get_idct_fast(const jpeggpu_ctx_t *ctx) { return ctx ? ctx->idct_fast.load() : -1; }

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

// ---- per-context forms -------------------------------------------------------------------------------

int jpeggpu_ctx_synchronize(jpeggpu_ctx_t *ctx, void *hip_stream) {
   int rc = bind(ctx); if (rc) return rc;
   HIP_TRY(hipStreamSynchronize(hip_stream ? (hipStream_t)hip_stream : ctx->stream));
   return 0;
}

int jpeggpu_ctx_backend_run_batch_device(jpeggpu_ctx_t *ctx, const jpeggpu_frame_desc_t *d, uint32_t nframes, const void *const d_coef[3],
                                         const size_t coef_frame_stride[3], const jpeggpu_qtable_t *const qt[4], void *d_out,
                                         size_t out_stride, size_t out_frame_stride, void *hip_stream) {
   int rc = validate(d, qt); if (rc) return rc;
   rc = bind(ctx); if (rc) return rc;
   return run_device(ctx, d, nframes, d_coef, coef_frame_stride, qt, d_out, out_stride, out_frame_stride,
                     hip_stream ? (hipStream_t)hip_stream : ctx->stream);
}

int jpeggpu_ctx_backend_run_device(jpeggpu_ctx_t *ctx, const jpeggpu_frame_desc_t *d, const void *const d_coef[3],
                                   const jpeggpu_qtable_t *const qt[4], void *d_out, size_t out_stride, void *hip_stream) {
   return jpeggpu_ctx_backend_run_batch_device(ctx, d, 1, d_coef, nullptr, qt, d_out, out_stride, 0, hip_stream);
}

int jpeggpu_ctx_backend_run(jpeggpu_ctx_t *ctx, const jpeggpu_frame_desc_t *d, const jpeggpu_blocl16_t *const coef[3],
                            const jpeggpu_qtable_t *const qt[4], uint8_t *out, size_t out_stride) {
   int rc = bind(ctx); if (rc) return rc;
   return ctx_backend_run(ctx, d, coef, qt, out, out_stride);
}

int jpeggpu_ctx_backend_run_batch(jpeggpu_ctx_t *ctx, const jpeggpu_frame_desc_t *d, uint32_t nframes, const jpeggpu_blocl16_t *const coef[3],
                                  const size_t coef_frame_stride[3], const jpeggpu_qtable_t *const qt[4], uint8_t *out,
                                  size_t out_stride, size_t out_frame_stride) {
   int rc = bind(ctx); if (rc) return rc;
   return ctx_backend_run_batch(ctx, d, nframes, coef, coef_frame_stride, qt, out, out_stride, out_frame_stride);
}

// ---- several GPUs from one process: independent shards, one host thread per context, no exchange step ----

// (ii) ONE frame cut into contiguous MCU-row stripes: context g decodes rows [g*nbmcuV/n, (g+1)*nbmcuV/n) of every component
// and writes the matching band of the caller's raster (replicate upsampling has no halo, src/decoder_utils.c:72-80).
// (i)  A batch of independent frames: frame i goes to context i mod n.
// Both run the pipelined host-pointer form of each context on its own thread; the first error wins.
static int run_sharded(jpeggpu_ctx_t *const *ctxs, int nctx, const jpeggpu_frame_desc_t *d, uint32_t nframes,
                       const jpeggpu_blocl16_t *const coef[3], const size_t cfs_in[3], const jpeggpu_qtable_t *const qt[4],
                       uint8_t *out, size_t out_stride, size_t out_frame_stride, bool stripes) {
   int rc = validate(d, qt); if (rc) return rc;
   if (!ctxs || nctx < 1) return fail(JPEGGPU_ERR_INVALID_ARG, "no contexts");
   for (int g = 0; g < nctx; g++) if (!ctxs[g]) return fail(JPEGGPU_ERR_INVALID_ARG, "ctxs[%d] is NULL", g);
   if (!out) return fail(JPEGGPU_ERR_INVALID_ARG, "out is NULL");
   for (int c = 0; c < d->nbcomp; c++) if (!coef || !coef[c]) return fail(JPEGGPU_ERR_INVALID_ARG, "coef[%d] is NULL", c);
   const size_t row_bytes = jpeggpu_raster_row_bytes(d), rows = jpeggpu_raster_rows(d), band = (size_t)d->max_v * 8;
   if (out_stride == 0) out_stride = row_bytes;
   if (out_stride < row_bytes) return fail(JPEGGPU_ERR_INVALID_ARG, "out_stride %zu < row bytes %zu", out_stride, row_bytes);
   if (out_frame_stride == 0) out_frame_stride = out_stride * rows;
   size_t cfs[3] = { 0, 0, 0 }, row_in[3] = { 0, 0, 0 };
   for (int c = 0; c < d->nbcomp; c++) {
      cfs[c] = (cfs_in && cfs_in[c]) ? cfs_in[c] : jpeggpu_plane_blocks(d, c) * sizeof(jpeggpu_blocl16_t);
      row_in[c] = (size_t)d->nbmcuH * d->comp[c].hs * d->comp[c].vs * sizeof(jpeggpu_blocl16_t);
   }
   std::vector<int> status((size_t)nctx, 0);
   std::vector<std::string> text((size_t)nctx);
   std::vector<std::thread> th;
   auto shard = [&](int g) {
         int r = bind(ctxs[g]);
         if (!r && stripes) {
            const uint32_t m0 = (uint32_t)(((uint64_t)g * d->nbmcuV) / nctx), m1 = (uint32_t)(((uint64_t)(g + 1) * d->nbmcuV) / nctx);
            if (m1 > m0) {
               jpeggpu_frame_desc_t sd = *d;
               sd.nbmcuV = m1 - m0;
               const jpeggpu_blocl16_t *sc[3] = { nullptr, nullptr, nullptr };
               for (int c = 0; c < d->nbcomp; c++) sc[c] = (const jpeggpu_blocl16_t *)((const char *)coef[c] + (size_t)m0 * row_in[c]);
               r = run_host(ctxs[g], &sd, 0, 1, 1, sc, nullptr, qt, out + (size_t)m0 * band * out_stride, out_stride, 0);
            }
         } else if (!r) {
            // frames g, g + nctx, ...: one pipelined call, so that the upload of a frame overlaps the download of the one before
            if ((uint32_t)g < nframes)
               r = run_host(ctxs[g], d, (uint32_t)g, (uint32_t)nctx, (nframes - (uint32_t)g + (uint32_t)nctx - 1) / (uint32_t)nctx, coef, cfs, qt,
                            out, out_stride, out_frame_stride);
         }
         status[(size_t)g] = r;
         if (r) text[(size_t)g] = g_err;     // g_err is thread-local: carry the text back to the caller's thread
   };
   for (int g = 0; g < nctx; g++) {
      try { th.emplace_back(shard, g); }
      catch (...) { shard(g); }              // no thread to be had: this shard runs on the caller's thread
   }
   for (auto &t : th) t.join();
   for (int g = 0; g < nctx; g++)
      if (status[(size_t)g]) return fail(status[(size_t)g], "context %d (device %d): %s", g, ctxs[g]->device, text[(size_t)g].c_str());
   return 0;
}

int jpeggpu_multi_backend_run(jpeggpu_ctx_t *const *ctxs, int nctx, const jpeggpu_frame_desc_t *d,
                              const jpeggpu_blocl16_t *const coef[3], const jpeggpu_qtable_t *const qt[4], uint8_t *out, size_t out_stride) {
   return run_sharded(ctxs, nctx, d, 1, coef, nullptr, qt, out, out_stride, 0, true);
}

int jpeggpu_multi_backend_run_batch(jpeggpu_ctx_t *const *ctxs, int nctx, const jpeggpu_frame_desc_t *d, uint32_t nframes,
                                    const jpeggpu_blocl16_t *const coef[3], const size_t coef_frame_stride[3],
                                    const jpeggpu_qtable_t *const qt[4], uint8_t *out, size_t out_stride, size_t out_frame_stride) {
   return run_sharded(ctxs, nctx, d, nframes, coef, coef_frame_stride, qt, out, out_stride, out_frame_stride, false);
}

// ---- process-wide forms: the same on the default context -------------------------------------------

int jpeggpu_synchronize(void *hip_stream) {
   jpeggpu_ctx_s *c; int rc = default_ctx(&c); if (rc) return rc;
   return jpeggpu_ctx_synchronize(c, hip_stream);
}

int jpeggpu_backend_run_batch_device(const jpeggpu_frame_desc_t *d, uint32_t nframes, const void *const d_coef[3],
                                     const size_t coef_frame_stride[3], const jpeggpu_qtable_t *const qt[4], void *d_out,
                                     size_t out_stride, size_t out_frame_stride, void *hip_stream) {
   int rc = validate(d, qt); if (rc) return rc;           // argument errors are reported without a device
   jpeggpu_ctx_s *c; rc = default_ctx(&c); if (rc) return rc;
   return run_device(c, d, nframes, d_coef, coef_frame_stride, qt, d_out, out_stride, out_frame_stride,
                     hip_stream ? (hipStream_t)hip_stream : c->stream);
}

int jpeggpu_backend_run_device(const jpeggpu_frame_desc_t *d, const void *const d_coef[3], const jpeggpu_qtable_t *const qt[4],
                               void *d_out, size_t out_stride, void *hip_stream) {
   return jpeggpu_backend_run_batch_device(d, 1, d_coef, nullptr, qt, d_out, out_stride, 0, hip_stream);
}

int jpeggpu_backend_run_batch(const jpeggpu_frame_desc_t *d, uint32_t nframes, const jpeggpu_blocl16_t *const coef[3],
                              const size_t coef_frame_stride[3], const jpeggpu_qtable_t *const qt[4], uint8_t *out,
                              size_t out_stride, size_t out_frame_stride) {
   int rc = validate(d, qt); if (rc) return rc;           // argument errors are reported without a device
   if (!out) return fail(JPEGGPU_ERR_INVALID_ARG, "out is NULL");
   for (int c = 0; c < d->nbcomp; c++) if (!coef || !coef[c]) return fail(JPEGGPU_ERR_INVALID_ARG, "coef[%d] is NULL", c);
   jpeggpu_ctx_s *c; rc = default_ctx(&c); if (rc) return rc;
   return ctx_backend_run_batch(c, d, nframes, coef, coef_frame_stride, qt, out, out_stride, out_frame_stride);
}

int jpeggpu_backend_run(const jpeggpu_frame_desc_t *d, const jpeggpu_blocl16_t *const coef[3], const jpeggpu_qtable_t *const qt[4],
                        uint8_t *out, size_t out_stride) {
   return jpeggpu_backend_run_batch(d, 1, coef, nullptr, qt, out, out_stride, 0);
}

// The allocation-free variant of jpeggpu_backend_run: same bytes, no hipMalloc / hipFree / hipHostMalloc anywhere on
// its path.  The device staging (planes in, packed raster out) is kernels.hip's static array; its address on this
// device was looked up when the context was created.  Nothing in the array has to start at zero: the planes are
// overwritten by the upload before the kernel reads them and the kernel writes every byte of the packed raster.
// All calls on one GPU of this process share the array, so they are serialised here; the previous call's bytes stay
// in it until overwritten.
int jpeggpu_backend_run_noalloc(jpeggpu_ctx_t *ctx, const jpeggpu_frame_desc_t *d, const jpeggpu_blocl16_t *const coef[3],
                                const jpeggpu_qtable_t *const qt[4], uint8_t *out, size_t out_stride) {
   static std::mutex device_mu[kMaxDevices];
   int rc = validate(d, qt); if (rc) return rc;
   if (!out) return fail(JPEGGPU_ERR_INVALID_ARG, "out is NULL");
   for (int c = 0; c < d->nbcomp; c++) if (!coef || !coef[c]) return fail(JPEGGPU_ERR_INVALID_ARG, "coef[%d] is NULL", c);
   if (!ctx) ctx = g_default.load();
   if (!ctx) return fail(JPEGGPU_ERR_INVALID_ARG, "no context: jpeggpu_init() / jpeggpu_ctx_create() first (creating one allocates)");
   rc = bind(ctx); if (rc) return rc;
   const size_t row_bytes = jpeggpu_raster_row_bytes(d), rows = jpeggpu_raster_rows(d), out_bytes = row_bytes * rows;
   if (out_stride == 0) out_stride = row_bytes;
   if (out_stride < row_bytes) return fail(JPEGGPU_ERR_INVALID_ARG, "out_stride %zu < row bytes %zu", out_stride, row_bytes);
   size_t off[3] = { 0, 0, 0 }, in_bytes = 0;
   for (int c = 0; c < d->nbcomp; c++) { off[c] = in_bytes; in_bytes += (jpeggpu_plane_blocks(d, c) * sizeof(jpeggpu_blocl16_t) + 255) & ~(size_t)255; }
   if (in_bytes + out_bytes > kStaticScratchBytes)
      return fail(JPEGGPU_ERR_INVALID_ARG, "frame needs %zu bytes of device staging (planes %zu + raster %zu); the static buffer of jpeggpu_backend_run_noalloc holds %zu",
                  in_bytes + out_bytes, in_bytes, out_bytes, (size_t)kStaticScratchBytes);
   uint8_t *base = (uint8_t *)ctx->static_scratch;     // the DEVICE address, never the array's name
   std::lock_guard<std::mutex> lk(device_mu[ctx->device]);
   // Whatever way this section is left, nothing of it may still be queued when the mutex goes: the next caller on this GPU
   // overwrites the shared array, and the caller's planes and raster must not be touched after the return.  (Declared
   // after the lock: destroyed before it.)
   struct Drain { hipStream_t s; ~Drain() { (void)hipStreamSynchronize(s); } } drain{ ctx->stream };
   const void *dc[3] = { nullptr, nullptr, nullptr };
   for (int c = 0; c < d->nbcomp; c++) {
      HIP_TRY(hipMemcpyAsync(base + off[c], coef[c], jpeggpu_plane_blocks(d, c) * sizeof(jpeggpu_blocl16_t), hipMemcpyHostToDevice, ctx->stream));
      dc[c] = base + off[c];
   }
   uint8_t *d_raster = base + in_bytes;
   rc = run_device(ctx, d, 1, dc, nullptr, qt, d_raster, row_bytes, 0, ctx->stream, false);
   if (rc) return rc;
   if (out_stride == row_bytes) HIP_TRY(hipMemcpyAsync(out, d_raster, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
   else HIP_TRY(hipMemcpy2DAsync(out, out_stride, d_raster, row_bytes, row_bytes, rows, hipMemcpyDeviceToHost, ctx->stream));
   HIP_TRY(hipStreamSynchronize(ctx->stream));
   return 0;
}

// Pinned host memory for planes and rasters: jpeggpu_backend_run moves it by DMA in place (no staging copy).
void *jpeggpu_host_alloc(size_t bytes) {
   jpeggpu_ctx_s *c; if (default_ctx(&c)) return nullptr;
   void *p = nullptr;
   hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
   if (e != hipSuccess) { fail(JPEGGPU_ERR_HIP, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e)); return nullptr; }
   return p;
}
void jpeggpu_host_free(void *ptr) { if (ptr) (void)hipHostFree(ptr); }

// ---- stage-level array forms -----------------------------------------------------------------

static int stage_roundtrip(const void *in, size_t in_bytes, void *out, size_t out_bytes,
                           hipError_t (*launch)(void *, void *, void *, hipStream_t), void *arg) {
   jpeggpu_ctx_s *c; int rc = default_ctx(&c); if (rc) return rc;
   std::lock_guard<std::mutex> lk(c->mu);
   rc = grow(&c->d_in, &c->d_in_cap, in_bytes, false); if (rc) return rc;
   rc = grow(&c->d_out, &c->d_out_cap, out_bytes, false); if (rc) return rc;
   HIP_TRY(hipMemcpyAsync(c->d_in, in, in_bytes, hipMemcpyHostToDevice, c->stream));
   HIP_TRY(launch(c->d_in, c->d_out, arg, c->stream));
   HIP_TRY(hipMemcpyAsync(out, out_bytes ? c->d_out : c->d_in, out_bytes ? out_bytes : in_bytes, hipMemcpyDeviceToHost, c->stream));
   HIP_TRY(hipStreamSynchronize(c->stream));
   return 0;
}

int jpeggpu_iquant_blocks(jpeggpu_blocl16_t *blocks, size_t n, uint8_t s_start, uint8_t s_end, const jpeggpu_qtable_t *q) {
   if (n == 0) return 0;
   if (!blocks || !q) return fail(JPEGGPU_ERR_INVALID_ARG, "NULL argument");
   if (s_end > 63) return fail(JPEGGPU_ERR_INVALID_ARG, "s_end %u > 63", s_end);
   struct Arg { size_t n; int s0, s1; uint32_t q[32]; } arg = { n, s_start, s_end, {} };
   pack_qtable(q, arg.q);
   return stage_roundtrip(blocks, n * 128, blocks, 0, [](void *di, void *, void *a, hipStream_t s) {
      Arg *x = (Arg *)a; return launch_iquant(di, x->n, x->s0, x->s1, x->q, s); }, &arg);
}

int jpeggpu_izz_blocks(const jpeggpu_blocl16_t *in, jpeggpu_bloct16_t *out, size_t n) {
   if (n == 0) return 0;
   if (!in || !out) return fail(JPEGGPU_ERR_INVALID_ARG, "NULL argument");
   return stage_roundtrip(in, n * 128, out, n * 128, [](void *di, void *dout, void *a, hipStream_t s) { return launch_izz(di, dout, *(size_t *)a, s); }, &n);
}

int jpeggpu_idct_opt_blocks(const jpeggpu_bloct16_t *in, jpeggpu_bloctu8_t *out, size_t n) {
   if (n == 0) return 0;
   if (!in || !out) return fail(JPEGGPU_ERR_INVALID_ARG, "NULL argument");
   return stage_roundtrip(in, n * 128, out, n * 64, [](void *di, void *dout, void *a, hipStream_t s) { return launch_idct(di, dout, *(size_t *)a, s); }, &n);
}

int jpeggpu_idct_blocks(const jpeggpu_bloct16_t *in, jpeggpu_bloctu8_t *out, size_t n) {
   if (n == 0) return 0;
   if (!in || !out) return fail(JPEGGPU_ERR_INVALID_ARG, "NULL argument");
   jpeggpu_ctx_s *c; int rc = default_ctx(&c); if (rc) return rc;
   struct Arg { size_t n; const float *coef; } arg = { n, nullptr };
   rc = ensure_slow_coef(c, &arg.coef); if (rc) return rc;
   return stage_roundtrip(in, n * 128, out, n * 64, [](void *di, void *dout, void *a, hipStream_t s) {
      Arg *x = (Arg *)a; return launch_idct_slow(di, dout, x->n, x->coef, s); }, &arg);
}

// all_option.idct_fast: process-wide in the reference (src/options.c:13,76-79).  Sets the default context's option and
// the value every context created later starts with; kept across jpeggpu_shutdown().
void jpeggpu_set_idct_fast(int fast) {
   g_idct_fast.store(fast != 0);
   if (jpeggpu_ctx_s *c = g_default.load()) c->idct_fast.store(fast != 0);
}
int jpeggpu_get_idct_fast(void) { return g_idct_fast.load(); }
void jpeggpu_idct_coef_table(float *out4096) { if (out4096) calc_coef_host(out4096); }

int jpeggpu_ycc2rgb_pixels(const uint8_t *y, const uint8_t *cb, const uint8_t *cr, jpeggpu_rgb_t *rgb, size_t n) {
   if (n == 0) return 0;
   if (!y || !cb || !cr || !rgb) return fail(JPEGGPU_ERR_INVALID_ARG, "NULL argument");
   jpeggpu_ctx_s *c; int rc = default_ctx(&c); if (rc) return rc;
   std::lock_guard<std::mutex> lk(c->mu);
   rc = grow(&c->d_in, &c->d_in_cap, 3 * n, false); if (rc) return rc;
   rc = grow(&c->d_out, &c->d_out_cap, 3 * n, false); if (rc) return rc;
   uint8_t *di = (uint8_t *)c->d_in;
   HIP_TRY(hipMemcpyAsync(di, y, n, hipMemcpyHostToDevice, c->stream));
   HIP_TRY(hipMemcpyAsync(di + n, cb, n, hipMemcpyHostToDevice, c->stream));
   HIP_TRY(hipMemcpyAsync(di + 2 * n, cr, n, hipMemcpyHostToDevice, c->stream));
   HIP_TRY(launch_ycc2rgb(di, di + n, di + 2 * n, (uint8_t *)c->d_out, n, c->stream));
   HIP_TRY(hipMemcpyAsync(rgb, c->d_out, 3 * n, hipMemcpyDeviceToHost, c->stream));
   HIP_TRY(hipStreamSynchronize(c->stream));
   return 0;
}

}  // extern "C"

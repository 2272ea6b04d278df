// batch_api.cpp -- jpeggpu_decode_batch (include/jpeggpu_decode.h): many same-geometry files, host
// threads for the serial entropy decoding, one batched launch of the fused kernel per chunk.
//
//   thread pool : image i of the chunk -> Decoder on the caller's bytes, planes = its slot of the pinned
//                 chunk buffer (frame-major per component, the layout of jpeggpu_backend_run_batch_device)
//   main thread : chunk k decoded -> H2D, kernel, D2H queued on stream k%2 -> decode chunk k+1 meanwhile
//   output      : a raster array that is pinned already (jpeggpu_host_alloc) or can be pinned in place
//                 (hipHostRegister) receives the D2H directly; else a pinned staging buffer and a parallel memcpy.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/jpeggpu_decode.h"
#include "frontend.hpp"

extern "C" void jpeggpu_internal_set_error(const char *msg);
extern "C" int jpeggpu_internal_ensure_init(void);
extern "C" int jpeggpu_internal_device(void);
extern "C" int jpeggpu_internal_range_pinned(const void *p, size_t bytes);   // capi.cpp: first and last byte are pinned host memory
extern "C" void *jpeggpu_internal_pool_take(int host, size_t bytes, int device);   // decode_api.cpp: buffers kept between calls
extern "C" void jpeggpu_internal_pool_give(int host, void *p, size_t bytes, int device);

namespace {

using clk = std::chrono::steady_clock;
double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }

int bfail(int code, const char *msg) { jpeggpu_internal_set_error(msg); return code; }

#define HIP_TRYB(expr)                                                          \
   do {                                                                         \
      hipError_t e_ = (expr);                                                   \
      if (e_ != hipSuccess) {                                                   \
         char b_[300];                                                          \
         snprintf(b_, sizeof b_, "%s: %s", #expr, hipGetErrorString(e_));       \
         jpeggpu_internal_set_error(b_);                                        \
         return JPEGGPU_ERR_HIP;                                                \
      }                                                                         \
   } while (0)

// Everything the pipeline allocates; released whatever way the call ends.  The chunk buffers go back to the library's
// buffer cache (decode_api.cpp): a second batch of the same shape finds them there, touched and mapped -- 600 MB of
// fresh pinned memory cost 0.1-0.3 ms per MB at the first CPU write (profiles/r02_probe_alloc.txt).
struct Resources {
   uint8_t *h_in[2] = { nullptr, nullptr }, *h_out[2] = { nullptr, nullptr };
   void *d_in[2] = { nullptr, nullptr }, *d_out[2] = { nullptr, nullptr };
   size_t in_cap = 0, out_cap = 0;
   int device = -1;
   hipStream_t st[2] = { nullptr, nullptr };
   hipEvent_t ev_done[2] = { nullptr, nullptr };
   void *registered = nullptr;
   hipError_t host_alloc(uint8_t **p, size_t bytes) {
      if ((*p = (uint8_t *)jpeggpu_internal_pool_take(1, bytes, device))) return hipSuccess;
      return hipHostMalloc((void **)p, bytes, hipHostMallocDefault);
   }
   hipError_t dev_alloc(void **p, size_t bytes) {
      if ((*p = jpeggpu_internal_pool_take(0, bytes, device))) return hipSuccess;
      return hipMalloc(p, bytes);
   }
   ~Resources() {
      for (int b = 0; b < 2; b++) if (st[b]) (void)hipStreamSynchronize(st[b]);
      if (registered) (void)hipHostUnregister(registered);
      for (int b = 0; b < 2; b++) {
         jpeggpu_internal_pool_give(1, h_in[b], in_cap, device);
         jpeggpu_internal_pool_give(1, h_out[b], out_cap, device);
         jpeggpu_internal_pool_give(0, d_in[b], in_cap, device);
         jpeggpu_internal_pool_give(0, d_out[b], out_cap, device);
         if (ev_done[b]) (void)hipEventDestroy(ev_done[b]);
         if (st[b]) (void)hipStreamDestroy(st[b]);
      }
   }
};

bool same_geometry(const jpeggpu_frame_desc_t &a, const jpegfe::Decoder &d) {
   if (a.width != d.width || a.height != d.height || a.nbcomp != d.nbcomp) return false;
   for (int c = 0; c < a.nbcomp; c++)
      if (a.comp[c].hs != d.comp[c].hs || a.comp[c].vs != d.comp[c].vs) return false;
   return true;
}

void desc_of(const jpegfe::Decoder &d, jpeggpu_frame_desc_t *desc) {
   memset(desc, 0, sizeof *desc);
   desc->width = d.width; desc->height = d.height; desc->nbcomp = d.nbcomp;
   desc->max_h = d.max_h; desc->max_v = d.max_v; desc->apply_iquant = 0;   // the threads dequantise
   desc->nbmcuH = d.nbmcuH; desc->nbmcuV = d.nbmcuV;
   for (int c = 0; c < d.nbcomp && c < 3; c++) { desc->comp[c].hs = d.comp[c].hs; desc->comp[c].vs = d.comp[c].vs; desc->comp[c].idq = d.comp[c].idq; }
}

// A small fork-join pool: run(n, fn) executes fn(i) for i in [0, n) on the workers and returns when all are done.
class Pool {
 public:
   explicit Pool(unsigned nthreads) {
      for (unsigned t = 0; t < nthreads; t++) workers_.emplace_back([this] { loop(); });
   }
   ~Pool() {
      { std::lock_guard<std::mutex> lk(mu_); stop_ = true; gen_++; }
      cv_.notify_all();
      for (auto &w : workers_) w.join();
   }
   void run(size_t n, const std::function<void(size_t)> &fn) {
      if (n == 0) return;
      {
         std::lock_guard<std::mutex> lk(mu_);
         fn_ = &fn; n_ = n; next_.store(0); pending_ = workers_.size(); gen_++;
      }
      cv_.notify_all();
      std::unique_lock<std::mutex> lk(mu_);
      done_.wait(lk, [this] { return pending_ == 0; });
   }
 private:
   void loop() {
      uint64_t seen = 0;
      while (true) {
         const std::function<void(size_t)> *fn;
         size_t n;
         {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return gen_ != seen; });
            seen = gen_;
            if (stop_) return;
            fn = fn_; n = n_;
         }
         for (size_t i; (i = next_.fetch_add(1)) < n;) (*fn)(i);
         std::lock_guard<std::mutex> lk(mu_);
         if (--pending_ == 0) done_.notify_all();
      }
   }
   std::vector<std::thread> workers_;
   std::mutex mu_;
   std::condition_variable cv_, done_;
   const std::function<void(size_t)> *fn_ = nullptr;
   std::atomic<size_t> next_{ 0 };
   size_t n_ = 0, pending_ = 0;
   uint64_t gen_ = 0;
   bool stop_ = false;
};

}  // namespace

extern "C" int jpeggpu_decode_batch(const jpeggpu_blob_t *jpegs, size_t n, uint8_t *rasters, size_t raster_frame_stride, int nthreads,
                                    int *status, jpeggpu_frame_desc_t *desc_out, jpeggpu_batch_stats_t *stats) {
   const auto t_begin = clk::now();
   if (stats) memset(stats, 0, sizeof *stats);
   if (n == 0) return 0;
   if (!jpegs || !rasters) return bfail(JPEGGPU_ERR_INVALID_ARG, "NULL argument");
   std::vector<int> own_status;
   if (!status) { own_status.assign(n, 0); status = own_status.data(); }

   // geometry = that of the first file whose header is acceptable
   jpeggpu_frame_desc_t desc;
   bool have_desc = false;
   int first_rc = JPEGGPU_ERR_INVALID_ARG;
   std::string first_err = "aucune image décodable";
   for (size_t i = 0; i < n && !have_desc; i++) {
      jpegfe::Decoder d;
      d.data = jpegs[i].data; d.len = jpegs[i].len;
      int r = (d.data && d.len) ? d.open() : JPEGGPU_ERR_INVALID_ARG;
      if (r == 0) { desc_of(d, &desc); have_desc = true; }
      else if (i == 0) { first_rc = r; first_err = d.err.empty() ? first_err : d.err; }
   }
   if (!have_desc) {
      for (size_t i = 0; i < n; i++) status[i] = first_rc;
      return bfail(first_rc, first_err.c_str());
   }
   if (desc_out) *desc_out = desc;
   const size_t rows = jpeggpu_raster_rows(&desc), rb = jpeggpu_raster_row_bytes(&desc), out_bytes = rows * rb;
   if (raster_frame_stride == 0) raster_frame_stride = out_bytes;
   if (raster_frame_stride < out_bytes) return bfail(JPEGGPU_ERR_INVALID_ARG, "raster_frame_stride plus petit qu'une image");
   size_t plane_bytes[3] = { 0, 0, 0 }, in_bytes = 0;
   for (int c = 0; c < desc.nbcomp; c++) { plane_bytes[c] = jpeggpu_plane_blocks(&desc, c) * sizeof(jpeggpu_blocl16_t); in_bytes += plane_bytes[c]; }
   if (out_bytes == 0 || in_bytes == 0) return bfail(JPEGGPU_ERR_INVALID_ARG, "image vide");

   int rc = jpeggpu_internal_ensure_init();
   if (rc) return rc;

   unsigned hw = std::thread::hardware_concurrency();
   unsigned T = nthreads > 0 ? (unsigned)nthreads : (hw ? (hw < 16 ? hw : 16) : 4);
   if (T > 256) T = 256;
   if (T > n) T = (unsigned)n;
   // chunk: about 256 MiB of coefficients, at least 2 images per thread, at most the batch
   static const size_t chunk_bytes = [] { const char *e = getenv("JPEGGPU_BATCH_CHUNK_BYTES"); return e && atoll(e) > 0 ? (size_t)atoll(e) : (size_t(256) << 20); }();
   size_t F = chunk_bytes / (in_bytes > out_bytes ? in_bytes : out_bytes);
   if (F < 2 * T) F = 2 * T;
   F = (F + T - 1) / T * T;                 // whole rounds of the thread pool
   if (F > 4096) F = 4096;
   if (F > n) F = n;
   const size_t nchunks = (n + F - 1) / F;
   const int nbuf = nchunks > 1 ? 2 : 1;

   std::atomic<uint64_t> vld_ns{ 0 };
   std::atomic<uint32_t> ok{ 0 };
   bool registered = false;
   auto pipeline = [&]() -> int {
      Resources R;          // destroyed after the pool below has joined
      Pool pool(T);
      bool queued[2] = { false, false };
      size_t queued_base[2] = { 0, 0 }, queued_cnt[2] = { 0, 0 };
      R.device = jpeggpu_internal_device();
      R.in_cap = F * in_bytes; R.out_cap = F * out_bytes;
      for (int b = 0; b < nbuf; b++) {
         HIP_TRYB(R.host_alloc(&R.h_in[b], R.in_cap));
         HIP_TRYB(R.dev_alloc(&R.d_in[b], R.in_cap));
         HIP_TRYB(R.dev_alloc(&R.d_out[b], R.out_cap));
         HIP_TRYB(hipStreamCreateWithFlags(&R.st[b], hipStreamNonBlocking));
         HIP_TRYB(hipEventCreateWithFlags(&R.ev_done[b], hipEventDisableTiming));
      }
      // the caller's raster array: already pinned (jpeggpu_host_alloc, hipHostMalloc) -> the downloads land there; else
      // pin it in place; where the driver refuses that, stage
      if (raster_frame_stride == out_bytes && jpeggpu_internal_range_pinned(rasters, n * raster_frame_stride)) {
         registered = true;
      } else if (raster_frame_stride == out_bytes && hipHostRegister(rasters, n * raster_frame_stride, hipHostRegisterDefault) == hipSuccess) {
         R.registered = rasters;
         registered = true;
      } else {
         (void)hipGetLastError();
         for (int b = 0; b < nbuf; b++) HIP_TRYB(R.host_alloc(&R.h_out[b], R.out_cap));
      }
      // Iteration k runs ONE fork-join over the pool: the decode tasks of chunk k followed by the copy-out
      // tasks of chunk k-1 (staging only).  The copy tasks sit at the end of the list, so the threads reach
      // them when the GPU has long finished chunk k-1; the first one to get there waits for its event.
      std::atomic<int> hip_error{ 0 };
      const int dev = jpeggpu_internal_device();   // the default context's device, not whatever this thread happens to be bound to
      for (size_t k = 0; k <= nchunks; k++) {
         const int b = (int)(k % nbuf);
         const bool have_chunk = k < nchunks;
         const size_t base = k * F, cnt = have_chunk ? ((base + F <= n) ? F : n - base) : 0;
         // the buffers of slot b are free once the launch that used them last (chunk k - nbuf) has finished
         if (have_chunk && queued[b] && (registered || nbuf == 1)) {
            HIP_TRYB(hipEventSynchronize(R.ev_done[b]));
            queued[b] = false;
         }
         // copy-out of the previous chunk (staging): slot pb
         const int pb = (int)((k + nbuf - 1) % nbuf);
         const bool copy_prev = !registered && k >= 1 && queued[pb] && !(nbuf == 1 && have_chunk);
         const size_t pbase = queued_base[pb], pcnt = copy_prev ? queued_cnt[pb] : 0;
         const uint8_t *psrc = R.h_out[pb];
         hipEvent_t pev = R.ev_done[pb];
         // plane c of frame i of the chunk: h_in[b] + off[c] + i * plane_bytes[c]
         size_t off[3] = { 0, 0, 0 };
         for (int c = 1; c < desc.nbcomp; c++) off[c] = off[c - 1] + F * plane_bytes[c - 1];
         uint8_t *hin = R.h_in[b];
         pool.run(cnt + pcnt, [&](size_t task) {
            if (task >= cnt) {                          // ---- copy-out of image (task - cnt) of the previous chunk
               const size_t i = task - cnt;
               (void)hipSetDevice(dev);                 // HIP's current device is per thread: workers start on device 0
               if (hipEventSynchronize(pev) != hipSuccess) { hip_error.store(1); return; }
               memcpy(rasters + (pbase + i) * raster_frame_stride, psrc + i * out_bytes, out_bytes);
               return;
            }
            const size_t i = task;                      // ---- entropy decoding of image i of this chunk
            const auto t0 = clk::now();
            jpegfe::Decoder d;
            d.data = jpegs[base + i].data; d.len = jpegs[base + i].len;
            int r = (d.data && d.len) ? d.open() : JPEGGPU_ERR_INVALID_ARG;
            for (int c = 0; c < desc.nbcomp; c++) memset(hin + off[c] + i * plane_bytes[c], 0, plane_bytes[c]);
            if (r == 0 && !same_geometry(desc, d)) r = JPEGGPU_ERR_INVALID_ARG;
            if (r == 0) {
               for (int c = 0; c < desc.nbcomp; c++) d.plane[c] = (int16_t *)(hin + off[c] + i * plane_bytes[c]);
               d.dequant_on_store = d.num_sof == 0;
               while (r == 0 && !d.eoi_done) {
                  r = d.decode_scan();
                  if (r == 0 && !d.eoi_done) r = d.next_header();
               }
               if (r == 0 && d.num_sof != 0) {
                  for (int c = 0; c < d.nbcomp; c++) if (d.comp[c].idq > 3 || !d.qt_present[d.comp[c].idq]) r = JPEGGPU_ERR_NO_HT;
                  if (r == 0) d.dequantise_planes();
               }
            }
            status[base + i] = r;
            if (r == 0) ok.fetch_add(1);
            vld_ns.fetch_add((uint64_t)(secs(t0, clk::now()) * 1e9));
         });
         if (hip_error.load()) return bfail(JPEGGPU_ERR_HIP, "hipEventSynchronize a échoué dans un thread de copie");
         if (copy_prev) queued[pb] = false;
         if (!have_chunk) break;
         const void *dc[3] = { nullptr, nullptr, nullptr };
         size_t cfs[3] = { 0, 0, 0 };
         for (int c = 0; c < desc.nbcomp; c++) { dc[c] = (char *)R.d_in[b] + off[c]; cfs[c] = plane_bytes[c]; }
         for (int c = 0; c < desc.nbcomp; c++)   // the cnt frames actually filled of each component region
            HIP_TRYB(hipMemcpyAsync((char *)R.d_in[b] + off[c], hin + off[c], cnt * plane_bytes[c], hipMemcpyHostToDevice, R.st[b]));
         int r1 = jpeggpu_backend_run_batch_device(&desc, (uint32_t)cnt, dc, cfs, nullptr, R.d_out[b], 0, 0, R.st[b]);
         if (r1) return r1;
         HIP_TRYB(hipMemcpyAsync(registered ? rasters + base * raster_frame_stride : R.h_out[b], R.d_out[b], cnt * out_bytes,
                                 hipMemcpyDeviceToHost, R.st[b]));
         HIP_TRYB(hipEventRecord(R.ev_done[b], R.st[b]));
         queued[b] = true; queued_base[b] = base; queued_cnt[b] = cnt;
      }
      // whatever is still queued (registered output: the last launches; staging with one buffer: its copy-out)
      for (int b2 = 0; b2 < nbuf; b2++) {
         if (!queued[b2]) continue;
         HIP_TRYB(hipEventSynchronize(R.ev_done[b2]));
         if (!registered) {
            const size_t qb = queued_base[b2];
            const uint8_t *src = R.h_out[b2];
            pool.run(queued_cnt[b2], [&](size_t i) { memcpy(rasters + (qb + i) * raster_frame_stride, src + i * out_bytes, out_bytes); });
         }
         queued[b2] = false;
      }
      return 0;
   };
   rc = pipeline();
   if (stats) {
      stats->seconds = secs(t_begin, clk::now());
      stats->vld_seconds = vld_ns.load() * 1e-9;
      stats->images_ok = ok.load();
      stats->threads = T; stats->chunk_frames = (uint32_t)F; stats->chunks = (uint32_t)nchunks;
      stats->output_registered = registered ? 1 : 0;
   }
   return rc;
}

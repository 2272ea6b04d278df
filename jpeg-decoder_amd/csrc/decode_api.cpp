// decode_api.cpp -- C ABI of include/jpeggpu_decode.h: the host front end (frontend.cpp) wired to
// the GPU back end, the per-scan hand-off of progressive files and the PPM/PGM output stage.
//
// Coefficient planes live in pinned host memory when a GPU is present (the entropy decoder writes
// straight into the buffer hipMemcpyAsync reads), stay resident on the device between scans, and
// only the components a scan touched are uploaded again.  Upload, kernel and download of scan k
// are queued on one stream; the host goes on to parse and decode scan k+1 as soon as the upload
// has been read (event), so the GPU work of a scan overlaps the entropy decoding of the next.
#include <hip/hip_runtime_api.h>

#include <sys/stat.h>
#include <unistd.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/jpeggpu_decode.h"
#include "kernels.h"
#include "frontend.hpp"

namespace {

thread_local char g_derr[512] = "";

int dfail(int code, const char *fmt, ...) {
   va_list ap;
   va_start(ap, fmt);
   vsnprintf(g_derr, sizeof g_derr, fmt, ap);
   va_end(ap);
   return code;
}

}  // namespace

// capi.cpp: lets this file report through jpeggpu_last_error()
extern "C" void jpeggpu_internal_set_error(const char *msg);
extern "C" int jpeggpu_internal_ensure_init(void);
extern "C" int jpeggpu_internal_range_pinned(const void *p, size_t bytes);   // capi.cpp: first and last byte are pinned host memory
extern "C" size_t jpeggpu_internal_stripe_bytes(void);

struct jpeggpu_jpeg {
   jpegfe::Decoder d;
   bool pinned = false;
   int device = -1;               // the device the buffers belong to (the library's default context at open time)
   size_t plane_bytes[3] = { 0, 0, 0 };
   // device side, created by the first render
   bool gpu = false;
   hipStream_t stream = nullptr;
   hipEvent_t uploaded = nullptr;
   void *d_plane[3] = { nullptr, nullptr, nullptr };
   void *d_pack[3] = { nullptr, nullptr, nullptr };   // device side of Decoder::dc_pack (progressive files)
   bool d_zeroed[3] = { false, false, false };        // the device plane has been initialised (uploaded or cleared)
   bool pack_on_device[3] = { false, false, false };  // d_pack[c] holds what dc_pack[c] held at the last render that uploaded it
   size_t uploaded_bytes = 0;                         // host-to-device bytes queued so far (jpeggpu_jpeg_uploaded_bytes)
   void *d_out = nullptr;
   uint8_t *h_out = nullptr;      // pinned raster
   uint8_t *h_alt[3] = { nullptr, nullptr, nullptr };   // further pinned rasters (jpeggpu_jpeg2ppm: images are written to their files while later ones are decoded and rendered)
   size_t out_bytes = 0;
   bool in_flight = false;        // a render has been queued and not yet waited for
};

namespace {

int report(jpeggpu_jpeg *j, int rc) {
   if (rc) jpeggpu_internal_set_error(j->d.err.c_str());
   return rc;
}

#define HIP_TRYD(expr)                                                                       \
   do {                                                                                      \
      hipError_t e_ = (expr);                                                                \
      if (e_ != hipSuccess) {                                                                \
         dfail(JPEGGPU_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));                      \
         jpeggpu_internal_set_error(g_derr);                                                 \
         return JPEGGPU_ERR_HIP;                                                             \
      }                                                                                      \
   } while (0)

// ---- buffers kept between files -------------------------------------------------------------------------------------
// A fresh pinned buffer costs little to allocate but 0.1-0.3 ms per MB when the CPU first touches it (page faults:
// 30 ms for the planes of a 4096x4096 4:4:4 file, profiles/r02_probe_alloc.txt), against 1 ms to clear one that has been
// touched before.  A closed file's buffers (pinned planes and raster, device planes and raster) are therefore kept, up
// to JPEGGPU_POOL_BYTES per kind (default 2 GiB, 0 = keep nothing), and handed to the next file that needs one of
// exactly that size on the same device -- the serving pattern of many files of one geometry.  jpeggpu_shutdown frees them.
// jpeggpu_decode_batch (batch_api.cpp) keeps its chunk buffers the same way.
struct Cached { void *p; size_t bytes; int device; bool host; };
std::mutex g_pool_mu;
std::vector<Cached> g_pool;                 // oldest first
size_t g_pool_held[2] = { 0, 0 };           // [host]

size_t pool_limit() {
   static const size_t v = [] { const char *e = getenv("JPEGGPU_POOL_BYTES"); return e ? (size_t)atoll(e) : ((size_t)2 << 30); }();
   return v;
}

void pool_release(const Cached &c) {
   if (c.host) (void)hipHostFree(c.p);
   else (void)hipFree(c.p);
}

void *pool_take(bool host, size_t bytes, int device) {
   std::lock_guard<std::mutex> lk(g_pool_mu);
   for (size_t i = g_pool.size(); i-- > 0;)
      if (g_pool[i].host == host && g_pool[i].device == device && g_pool[i].bytes == bytes) {
         void *p = g_pool[i].p;
         g_pool_held[host] -= bytes;
         g_pool.erase(g_pool.begin() + (long)i);
         return p;
      }
   return nullptr;
}

void pool_give(bool host, void *p, size_t bytes, int device) {
   if (!p) return;
   std::vector<Cached> out;
   {
      std::lock_guard<std::mutex> lk(g_pool_mu);
      if (bytes > pool_limit() || device < 0) {
         out.push_back(Cached{ p, bytes, device, host });
      } else {
         g_pool.push_back(Cached{ p, bytes, device, host });
         g_pool_held[host] += bytes;
         for (size_t i = 0; g_pool_held[host] > pool_limit() && i < g_pool.size();) {
            if (g_pool[i].host != host) { i++; continue; }
            out.push_back(g_pool[i]);
            g_pool_held[host] -= g_pool[i].bytes;
            g_pool.erase(g_pool.begin() + (long)i);
         }
      }
   }
   for (const Cached &c : out) pool_release(c);
}

int current_device() {
   int dev = -1;
   if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return -1; }
   return dev;
}

hipError_t host_alloc(void **p, size_t bytes, int device) {
   if ((*p = pool_take(true, bytes, device))) return hipSuccess;
   return hipHostMalloc(p, bytes, hipHostMallocDefault);
}

hipError_t dev_alloc(void **p, size_t bytes, int device) {
   if ((*p = pool_take(false, bytes, device))) return hipSuccess;
   return hipMalloc(p, bytes);
}

bool have_device() {
   int n = 0;
   return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}

int alloc_planes(jpeggpu_jpeg *j) {
   jpegfe::Decoder &d = j->d;
   j->pinned = have_device() && jpeggpu_internal_ensure_init() == 0;
   j->device = j->pinned ? current_device() : -1;
   for (int c = 0; c < d.nbcomp; c++) {
      size_t bytes = d.plane_blocks(c) * sizeof(jpeggpu_blocl16_t);
      j->plane_bytes[c] = bytes;
      void *p = nullptr;
      if (bytes == 0) bytes = 128;
      if (j->pinned) {
         if (host_alloc(&p, bytes, j->device) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
      } else {
         p = malloc(bytes);
      }
      if (!p) return dfail(JPEGGPU_ERR_IO, "allocation de %zu octets impossible", bytes);
      memset(p, 0, bytes);
      d.plane[c] = (int16_t *)p;
      // progressive files: DC scans also leave their coefficients packed (2 bytes per block) for the per-scan hand-off
      if (j->pinned && d.num_sof == 2 && j->plane_bytes[c] >= 128) {
         void *q = nullptr;
         if (host_alloc(&q, j->plane_bytes[c] / 64, j->device) == hipSuccess) { memset(q, 0, j->plane_bytes[c] / 64); d.dc_pack[c] = (int16_t *)q; }
         else (void)hipGetLastError();   // no packed side: whole planes are uploaded
      }
   }
   // DC scans then write coefficient 0 into the packed arrays only (2 bytes per block instead of a 128-byte line of the
   // plane per block: the scans are memory-bound otherwise); render_begin puts the values in place on the device
   d.dc_in_pack = j->pinned && d.num_sof == 2;
   return 0;
}

void free_all(jpeggpu_jpeg *j) {
   // every member is released on its own: a render that failed half-way through its allocations leaves some set
   if (j->stream) (void)hipStreamSynchronize(j->stream);
   // buffers go to the cache (pool_give frees what it will not keep); sizes as they were asked for
   const int dev = j->device;
   for (int c = 0; c < 3; c++) pool_give(false, j->d_plane[c], j->plane_bytes[c] ? j->plane_bytes[c] : 128, dev);
   for (int c = 0; c < 3; c++) pool_give(false, j->d_pack[c], j->plane_bytes[c] / 64, dev);
   for (int c = 0; c < 3; c++) pool_give(true, j->d.dc_pack[c], j->plane_bytes[c] / 64, dev);
   pool_give(false, j->d_out, j->out_bytes ? j->out_bytes : 16, dev);
   pool_give(true, j->h_out, j->out_bytes ? j->out_bytes : 16, dev);
   for (uint8_t *p : j->h_alt) pool_give(true, p, j->out_bytes ? j->out_bytes : 16, dev);
   if (j->uploaded) (void)hipEventDestroy(j->uploaded);
   if (j->stream) (void)hipStreamDestroy(j->stream);
   for (int c = 0; c < 3; c++)
      if (j->d.plane[c]) { if (j->pinned) pool_give(true, j->d.plane[c], j->plane_bytes[c] ? j->plane_bytes[c] : 128, dev); else free(j->d.plane[c]); }
   delete j;
}

int open_common(jpeggpu_jpeg *j, jpeggpu_jpeg_t **out) {
   int rc = j->d.open();
   if (rc) { report(j, rc); free_all(j); return rc; }
   rc = alloc_planes(j);
   if (rc) { jpeggpu_internal_set_error(g_derr); free_all(j); return rc; }
   *out = j;
   return 0;
}

void fill_desc(const jpegfe::Decoder &d, jpeggpu_frame_desc_t *desc) {
   memset(desc, 0, sizeof *desc);
   desc->width = d.width; desc->height = d.height; desc->nbcomp = d.nbcomp;
   desc->max_h = d.max_h; desc->max_v = d.max_v; desc->apply_iquant = 1;
   desc->nbmcuH = d.nbmcuH; desc->nbmcuV = d.nbmcuV;
   for (int c = 0; c < d.nbcomp && c < 3; c++) {
      desc->comp[c].hs = d.comp[c].hs; desc->comp[c].vs = d.comp[c].vs; desc->comp[c].idq = d.comp[c].idq;
   }
}

// The device side of a handle: stream, event, device planes and raster (and the pinned raster unless the caller brings one).
int prepare_gpu(jpeggpu_jpeg *j, const jpeggpu_frame_desc_t &desc, bool own_raster) {
   jpegfe::Decoder &d = j->d;
   int rc = jpeggpu_internal_ensure_init();
   if (rc) return rc;
   if (!j->gpu) {   // allocate what is still missing (a retry after a failed attempt keeps what it already has)
      if (!j->stream) HIP_TRYD(hipStreamCreateWithFlags(&j->stream, hipStreamNonBlocking));
      if (!j->uploaded) HIP_TRYD(hipEventCreateWithFlags(&j->uploaded, hipEventDisableTiming));
      for (int c = 0; c < d.nbcomp; c++) {
         if (!j->d_plane[c]) HIP_TRYD(dev_alloc(&j->d_plane[c], j->plane_bytes[c] ? j->plane_bytes[c] : 128, j->device));
         if (d.dc_pack[c] && !j->d_pack[c]) HIP_TRYD(dev_alloc(&j->d_pack[c], j->plane_bytes[c] / 64, j->device));
         d.dirty[c] = true;
         if (d.dirty_kind[c] == jpegfe::Decoder::kClean) d.dirty_kind[c] = jpegfe::Decoder::kWhole;   // nothing decoded yet: the (zero) plane as it is
      }
      j->out_bytes = jpeggpu_raster_rows(&desc) * jpeggpu_raster_row_bytes(&desc);
      if (j->device < 0) j->device = current_device();
      if (!j->d_out) HIP_TRYD(dev_alloc(&j->d_out, j->out_bytes ? j->out_bytes : 16, j->device));
      j->gpu = true;
   }
   if (own_raster && !j->h_out) HIP_TRYD(host_alloc((void **)&j->h_out, j->out_bytes ? j->out_bytes : 16, j->device));
   return 0;
}

// Queue upload of the dirty planes, the fused kernel and the download of the raster.
// direct != NULL: pinned caller memory the raster is downloaded into in place (rows direct_stride bytes apart) instead of
// into the handle's own pinned raster.
int render_begin(jpeggpu_jpeg *j, uint8_t *direct = nullptr, size_t direct_stride = 0) {
   jpegfe::Decoder &d = j->d;
   jpeggpu_frame_desc_t desc;
   fill_desc(d, &desc);
   const jpeggpu_qtable_t *qt[4];
   for (int i = 0; i < 4; i++) qt[i] = d.qt_present[i] ? (const jpeggpu_qtable_t *)d.qt[i] : nullptr;
   // src/progressive.c:262-267: a component without its table is ERR_NO_HT there
   for (int c = 0; c < d.nbcomp; c++)
      if (d.comp[c].idq > 3 || !qt[d.comp[c].idq]) {
         dfail(JPEGGPU_ERR_NO_HT, "Pas de table de quantification pour la composante %d", c);
         jpeggpu_internal_set_error(g_derr);
         return JPEGGPU_ERR_NO_HT;
      }
   int rc = prepare_gpu(j, desc, direct == nullptr);
   if (rc) return rc;
   // Per-scan hand-off (src/progressive.c:209-331 runs the whole back end after every scan): a component that only DC
   // scans touched since its last upload sends the dirty spectral band -- coefficient 0, 2 of every 128 bytes -- and a
   // small kernel puts it in place; the first time round the device plane is cleared instead of uploaded (everything
   // but coefficient 0 is still zero on the host as well).  Anything else sends the component's plane.
   for (int c = 0; c < d.nbcomp; c++)
      if (d.dirty[c] && j->plane_bytes[c]) {
         const size_t nblocks = j->plane_bytes[c] / 128;
         const bool have_pack = j->d_pack[c] && d.dc_pack[c];
         if (d.dirty_kind[c] == jpegfe::Decoder::kDcOnly && have_pack) {
            if (!j->d_zeroed[c]) HIP_TRYD(hipMemsetAsync(j->d_plane[c], 0, j->plane_bytes[c], j->stream));
            HIP_TRYD(hipMemcpyAsync(j->d_pack[c], d.dc_pack[c], nblocks * 2, hipMemcpyHostToDevice, j->stream));
            HIP_TRYD(jpeggpu_k::launch_scatter_dc(j->d_plane[c], j->d_pack[c], nblocks, j->stream));
            j->uploaded_bytes += nblocks * 2;
            j->pack_on_device[c] = true;
         } else {
            HIP_TRYD(hipMemcpyAsync(j->d_plane[c], d.plane[c], j->plane_bytes[c], hipMemcpyHostToDevice, j->stream));
            j->uploaded_bytes += j->plane_bytes[c];
            if (d.dc_in_pack && have_pack) {
               // the host plane carries no coefficient 0 (Decoder::dc_in_pack): put the packed values in place behind the upload
               if (d.dc_dirty[c] || !j->pack_on_device[c]) {
                  HIP_TRYD(hipMemcpyAsync(j->d_pack[c], d.dc_pack[c], nblocks * 2, hipMemcpyHostToDevice, j->stream));
                  j->uploaded_bytes += nblocks * 2;
                  j->pack_on_device[c] = true;
               }
               HIP_TRYD(jpeggpu_k::launch_scatter_dc(j->d_plane[c], j->d_pack[c], nblocks, j->stream));
            }
         }
         j->d_zeroed[c] = true;
         d.dirty[c] = false;
         d.dc_dirty[c] = false;
         d.dirty_kind[c] = jpegfe::Decoder::kClean;
      }
   HIP_TRYD(hipEventRecord(j->uploaded, j->stream));
   rc = jpeggpu_backend_run_device(&desc, (const void *const *)j->d_plane, qt, j->d_out, 0, j->stream);
   if (rc) return rc;
   const size_t rb = jpeggpu_raster_row_bytes(&desc);
   if (!direct) HIP_TRYD(hipMemcpyAsync(j->h_out, j->d_out, j->out_bytes, hipMemcpyDeviceToHost, j->stream));
   else if (direct_stride == rb) HIP_TRYD(hipMemcpyAsync(direct, j->d_out, j->out_bytes, hipMemcpyDeviceToHost, j->stream));
   else if (rb) HIP_TRYD(hipMemcpy2DAsync(direct, direct_stride, j->d_out, rb, rb, j->out_bytes / rb, hipMemcpyDeviceToHost, j->stream));
   j->in_flight = true;
   return 0;
}

// The planes may be written again (next scan) once the upload has been consumed.
int render_planes_free(jpeggpu_jpeg *j) {
   if (j->in_flight) HIP_TRYD(hipEventSynchronize(j->uploaded));
   return 0;
}

int render_end(jpeggpu_jpeg *j) {
   if (j->in_flight) HIP_TRYD(hipStreamSynchronize(j->stream));
   j->in_flight = false;
   return 0;
}

void write_pnm_header(FILE *f, const jpeggpu_frame_desc_t *desc) {   // src/baseline.c:121-130
   fprintf(f, desc->nbcomp == 1 ? "P5\n" : "P6\n");
   fprintf(f, "%d %d\n", desc->width, desc->height);
   fprintf(f, "255\n");
}

int write_pnm(const char *path, const jpeggpu_frame_desc_t *desc, const uint8_t *raster, size_t stride, size_t rows) {
   FILE *f = fopen(path, "w");
   if (!f) return dfail(JPEGGPU_ERR_IO, "ouverture de '%s' impossible", path);
   const size_t rb = jpeggpu_raster_row_bytes(desc);
   write_pnm_header(f, desc);
   bool ok = true;
   if (rows && rb) {
      if (stride == rb) ok = fwrite(raster, 1, rb * rows, f) == rb * rows;
      else for (size_t r = 0; r < rows && ok; r++) ok = fwrite(raster + r * stride, 1, rb, f) == rb;
   }
   if (fclose(f) != 0) ok = false;
   return ok ? 0 : dfail(JPEGGPU_ERR_IO, "écriture de '%s' incomplète", path);
}

// A baseline file is handed on while it is being decoded: every few MCU rows the finished rows are uploaded, converted
// and downloaded on the handle's stream, and a thread of its own appends them to the output file as their downloads
// complete -- so at the end of the entropy decoding only the last rows are left to do.  (What the file holds is what
// decode_baseline_image writes, src/baseline.c:89-231: the header and every complete MCU row, also of a damaged scan.)
struct BandWriter {
   struct Job { hipEvent_t ready; const uint8_t *p; size_t bytes; };
   FILE *f; int device;
   std::thread th;
   std::mutex mu;
   std::condition_variable cv;
   std::deque<Job> q;
   uint64_t posted = 0, done = 0;
   bool failed = false, started = false, threaded = false, stop = false;
   bool want_thread = true;               // false: a file that is handed on in one piece is written by the caller's thread
   BandWriter(FILE *f_, int dev) : f(f_), device(dev) {}
   void run_one(const Job &job) {   // mu NOT held
      bool ok = hipEventSynchronize(job.ready) == hipSuccess;
      if (ok && job.bytes) ok = fwrite(job.p, 1, job.bytes, f) == job.bytes;
      std::lock_guard<std::mutex> lk(mu);
      if (!ok) failed = true;
      done++;
      cv.notify_all();
   }
   void loop() {
      (void)hipSetDevice(device);
      for (;;) {
         Job job;
         { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || !q.empty(); }); if (q.empty()) return; job = q.front(); q.pop_front(); }
         run_one(job);
      }
   }
   void post(hipEvent_t ready, const uint8_t *p, size_t bytes) {
      if (!started) {
         started = true;
         if (want_thread) { try { th = std::thread([this] { loop(); }); threaded = true; } catch (const std::system_error &) { threaded = false; } }
      }
      { std::lock_guard<std::mutex> lk(mu); posted++; if (threaded) q.push_back(Job{ ready, p, bytes }); }
      if (threaded) cv.notify_all(); else run_one(Job{ ready, p, bytes });
   }
   bool drain() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done >= posted; }); return !failed; }   // everything posted is in the file
   ~BandWriter() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); if (th.joinable()) th.join(); }
};

struct BaselineStream {
   jpeggpu_jpeg *j;
   jpeggpu_frame_desc_t desc;
   const jpeggpu_qtable_t *qt[4];
   size_t band_bytes = 0, row_in[3] = { 0, 0, 0 };   // raster bytes / plane bytes of one MCU row
   uint32_t sent = 0;                                 // MCU rows handed on so far
   int rc = 0;                                        // first GPU error (its text is in jpeggpu_last_error())
   std::vector<hipEvent_t> events;
   BandWriter *writer = nullptr;
   // MCU rows [sent, upto): upload, kernel, download, and a job for the writer
   int send(uint32_t upto) {
      if (rc || upto <= sent) return rc;
      const uint32_t m0 = sent, n = upto - sent;
      const void *dc[3] = { nullptr, nullptr, nullptr };
      for (int c = 0; c < desc.nbcomp; c++) {
         char *dst = (char *)j->d_plane[c] + (size_t)m0 * row_in[c];
         hipError_t e = hipMemcpyAsync(dst, (const char *)j->d.plane[c] + (size_t)m0 * row_in[c], (size_t)n * row_in[c], hipMemcpyHostToDevice, j->stream);
         if (e != hipSuccess) { dfail(JPEGGPU_ERR_HIP, "hipMemcpyAsync (rows %u..%u): %s", m0, upto, hipGetErrorString(e)); jpeggpu_internal_set_error(g_derr); return rc = JPEGGPU_ERR_HIP; }
         dc[c] = dst;
      }
      jpeggpu_frame_desc_t sd = desc;
      sd.nbmcuV = n;
      uint8_t *d_band = (uint8_t *)j->d_out + (size_t)m0 * band_bytes;
      int r = jpeggpu_backend_run_device(&sd, dc, qt, d_band, 0, j->stream);
      if (r) return rc = r;
      hipEvent_t ev = nullptr;
      hipError_t e = hipMemcpyAsync(j->h_out + (size_t)m0 * band_bytes, d_band, (size_t)n * band_bytes, hipMemcpyDeviceToHost, j->stream);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
      if (e == hipSuccess) { events.push_back(ev); e = hipEventRecord(ev, j->stream); }
      if (e != hipSuccess) { dfail(JPEGGPU_ERR_HIP, "download of rows %u..%u: %s", m0, upto, hipGetErrorString(e)); jpeggpu_internal_set_error(g_derr); return rc = JPEGGPU_ERR_HIP; }
      j->in_flight = true;
      writer->post(ev, j->h_out + (size_t)m0 * band_bytes, (size_t)n * band_bytes);
      sent = upto;
      return 0;
   }
   static int hook(void *ctx, uint32_t rows_done) { return static_cast<BaselineStream *>(ctx)->send(rows_done); }
   ~BaselineStream() { for (hipEvent_t e : events) (void)hipEventDestroy(e); }
};

std::string out_name(const char *filepath, const char *outfile, uint8_t nbcomp, uint8_t nb) {
   std::string base = outfile ? outfile : filepath;
   size_t dot = base.rfind('.');
   std::string ext = dot == std::string::npos ? std::string() : base.substr(dot);
   std::string name = dot == std::string::npos ? base : base.substr(0, dot);
   if (nb != 0) name += "-" + std::to_string((unsigned)nb);
   if (!outfile) name += nbcomp == 1 ? ".pgm" : (nbcomp == 3 ? ".ppm" : "");
   else name += ext;
   return name;
}

}  // namespace

extern "C" {

// batch_api.cpp
void *jpeggpu_internal_pool_take(int host, size_t bytes, int device) { return pool_take(host != 0, bytes, device); }
void jpeggpu_internal_pool_give(int host, void *p, size_t bytes, int device) { pool_give(host != 0, p, bytes, device); }

// jpeggpu_shutdown (capi.cpp): nothing of the cache outlives the library's default context
void jpeggpu_internal_pool_drain(void) {
   std::vector<Cached> all;
   {
      std::lock_guard<std::mutex> lk(g_pool_mu);
      all.swap(g_pool);
      g_pool_held[0] = g_pool_held[1] = 0;
   }
   for (const Cached &c : all) pool_release(c);
}

int jpeggpu_jpeg_open_memory(const uint8_t *data, size_t len, jpeggpu_jpeg_t **out) {
   if (!data || !out) { jpeggpu_internal_set_error("NULL argument"); return JPEGGPU_ERR_INVALID_ARG; }
   *out = nullptr;
   jpeggpu_jpeg *j = new jpeggpu_jpeg();
   j->d.data = data;
   j->d.len = len;
   return open_common(j, out);
}

int jpeggpu_jpeg_open_file(const char *path, jpeggpu_jpeg_t **out) {
   if (!path || !out) { jpeggpu_internal_set_error("NULL argument"); return JPEGGPU_ERR_INVALID_ARG; }
   *out = nullptr;
   // ouverture_fichier_in, src/utils.c:43-54
   const char *dot = strrchr(path, '.');
   if (!dot || !(strcmp(dot + 1, "jpeg") == 0 || strcmp(dot + 1, "jpg") == 0)) {
      jpeggpu_internal_set_error("Mauvaise extension de fichier.");
      return JPEGGPU_ERR_INVALID_FILE_EXT;
   }
   FILE *f = fopen(path, "r");
   if (!f) { jpeggpu_internal_set_error("Fichier introuvable."); return JPEGGPU_ERR_INVALID_FILE_PATH; }
   jpeggpu_jpeg *j = new jpeggpu_jpeg();
   fseek(f, 0, SEEK_END);
   long sz = ftell(f);
   fseek(f, 0, SEEK_SET);
   j->d.owned.resize(sz > 0 ? (size_t)sz : 0);
   size_t got = sz > 0 ? fread(j->d.owned.data(), 1, (size_t)sz, f) : 0;
   fclose(f);
   j->d.owned.resize(got);
   j->d.data = j->d.owned.data();
   j->d.len = got;
   return open_common(j, out);
}

void jpeggpu_jpeg_close(jpeggpu_jpeg_t *j) { if (j) free_all(j); }
size_t jpeggpu_jpeg_uploaded_bytes(const jpeggpu_jpeg_t *j) { return j ? j->uploaded_bytes : 0; }

int jpeggpu_jpeg_frame_desc(const jpeggpu_jpeg_t *j, jpeggpu_frame_desc_t *desc) {
   if (!j || !desc) return JPEGGPU_ERR_INVALID_ARG;
   fill_desc(j->d, desc);
   return 0;
}

int jpeggpu_jpeg_scan_info(const jpeggpu_jpeg_t *j, jpeggpu_scan_info_t *info) {
   if (!j || !info) return JPEGGPU_ERR_INVALID_ARG;
   memset(info, 0, sizeof *info);
   const jpegfe::Decoder &d = j->d;
   info->num_sof = d.num_sof;
   for (unsigned k = 0; k < 3; k++) {
      int c = d.get_composante(k);
      if (c < 0) break;
      info->scan_comp[info->nb_scan_comp++] = (uint8_t)c;
   }
   info->ss = d.ss; info->se = d.se; info->ah = d.ah; info->al = d.al;
   info->eoi = d.eoi_done;
   info->scans_decoded = (uint16_t)d.scans_decoded;
   info->mcu_rows_done = d.mcu_rows_done;
   return 0;
}

const jpeggpu_qtable_t *jpeggpu_jpeg_qtable(const jpeggpu_jpeg_t *j, int idq) {
   if (!j || idq < 0 || idq > 3 || !j->d.qt_present[idq]) return nullptr;
   return (const jpeggpu_qtable_t *)j->d.qt[idq];
}

const jpeggpu_blocl16_t *jpeggpu_jpeg_plane(const jpeggpu_jpeg_t *j, int c) {
   if (!j || c < 0 || c >= j->d.nbcomp || c > 2) return nullptr;
   const_cast<jpeggpu_jpeg_t *>(j)->d.merge_dc_into_planes(c);   // progressive files on a GPU keep coefficient 0 packed while decoding
   return (const jpeggpu_blocl16_t *)j->d.plane[c];
}

int jpeggpu_jpeg_nb_comments(const jpeggpu_jpeg_t *j) { return j ? (int)j->d.comments.size() : 0; }
const char *jpeggpu_jpeg_comment(const jpeggpu_jpeg_t *j, int i) {
   return (j && i >= 0 && i < (int)j->d.comments.size()) ? j->d.comments[i].c_str() : nullptr;
}

int jpeggpu_jpeg_decode_scan(jpeggpu_jpeg_t *j) {
   if (!j) return JPEGGPU_ERR_INVALID_ARG;
   int rc = render_planes_free(j);
   if (rc) return rc;
   return report(j, j->d.decode_scan());
}

int jpeggpu_jpeg_next_header(jpeggpu_jpeg_t *j) {
   if (!j) return JPEGGPU_ERR_INVALID_ARG;
   return report(j, j->d.next_header());
}

int jpeggpu_jpeg_render(jpeggpu_jpeg_t *j, uint8_t *out, size_t out_stride) {
   if (!j || !out) { jpeggpu_internal_set_error("NULL argument"); return JPEGGPU_ERR_INVALID_ARG; }
   jpeggpu_frame_desc_t desc;
   fill_desc(j->d, &desc);
   const size_t rb = jpeggpu_raster_row_bytes(&desc), rows = jpeggpu_raster_rows(&desc);
   if (out_stride == 0) out_stride = rb;
   if (out_stride < rb) { jpeggpu_internal_set_error("out_stride plus petit qu'une ligne"); return JPEGGPU_ERR_INVALID_ARG; }
   const bool direct = j->pinned && rb * rows != 0 && jpeggpu_internal_range_pinned(out, (rows - 1) * out_stride + rb);
   int rc = render_begin(j, direct ? out : nullptr, out_stride);
   if (rc) return rc;
   rc = render_end(j);
   if (rc || direct) return rc;
   if (out_stride == rb) memcpy(out, j->h_out, rb * rows);
   else for (size_t r = 0; r < rows; r++) memcpy(out + r * out_stride, j->h_out + r * rb, rb);
   return 0;
}

int jpeggpu_write_pnm(const char *path, const jpeggpu_frame_desc_t *desc, const uint8_t *raster, size_t stride) {
   if (!path || !desc || !raster) { jpeggpu_internal_set_error("NULL argument"); return JPEGGPU_ERR_INVALID_ARG; }
   if (desc->nbcomp != 1 && desc->nbcomp != 3) { jpeggpu_internal_set_error("Il faut une ou trois composante"); return JPEGGPU_ERR_NB_COMP; }
   const size_t rb = jpeggpu_raster_row_bytes(desc);
   if (stride == 0) stride = rb;
   int rc = write_pnm(path, desc, raster, stride, jpeggpu_raster_rows(desc));
   if (rc) jpeggpu_internal_set_error(g_derr);
   return rc;
}

int jpeggpu_write_pnm_rows(const char *path, const jpeggpu_frame_desc_t *desc, const uint8_t *raster, size_t stride, size_t rows) {
   if (!path || !desc || !raster) { jpeggpu_internal_set_error("NULL argument"); return JPEGGPU_ERR_INVALID_ARG; }
   if (desc->nbcomp != 1 && desc->nbcomp != 3) { jpeggpu_internal_set_error("Il faut une ou trois composante"); return JPEGGPU_ERR_NB_COMP; }
   if (rows > jpeggpu_raster_rows(desc)) { jpeggpu_internal_set_error("plus de lignes que le raster n'en contient"); return JPEGGPU_ERR_INVALID_ARG; }
   if (stride == 0) stride = jpeggpu_raster_row_bytes(desc);
   int rc = write_pnm(path, desc, raster, stride, rows);
   if (rc) jpeggpu_internal_set_error(g_derr);
   return rc;
}

int jpeggpu_out_file_name(const char *filepath, const char *outfile, uint8_t nbcomp, uint8_t nb, char *buf, size_t cap) {
   if (!filepath || !buf) return JPEGGPU_ERR_INVALID_ARG;
   std::string n = out_name(filepath, outfile, nbcomp, nb);
   if (n.size() + 1 > cap) return JPEGGPU_ERR_INVALID_ARG;
   memcpy(buf, n.c_str(), n.size() + 1);
   return (int)n.size();
}

int jpeggpu_jpeg2ppm(const char *filepath, const char *outfile, int *nb_images) {
   return jpeggpu_jpeg2ppm_timed(filepath, outfile, nb_images, nullptr);
}

int jpeggpu_jpeg2ppm_timed(const char *filepath, const char *outfile, int *nb_images, jpeggpu_timing_t *timing) {
   using clk = std::chrono::steady_clock;
   jpeggpu_timing_t tm = {0, 0, 0, 0, 0, 0};
   const clk::time_point t_start = clk::now();
   // adds the wall time of a stage to its accumulator when it goes out of scope
   struct Lap {
      double &acc; clk::time_point t0;
      explicit Lap(double &a) : acc(a), t0(clk::now()) {}
      ~Lap() { acc += std::chrono::duration<double>(clk::now() - t0).count(); }
   };
   struct Total {   // fills *timing on every exit path
      jpeggpu_timing_t &tm; jpeggpu_timing_t *out; clk::time_point t0;
      ~Total() { tm.total_s = std::chrono::duration<double>(clk::now() - t0).count(); if (out) *out = tm; }
   } total{tm, timing, t_start};
   if (nb_images) *nb_images = 0;
   // verif_option_io, src/jpeg2ppm.c:25-60
   if (!filepath) { jpeggpu_internal_set_error("Pas de fichier jpeg/jpg"); return JPEGGPU_ERR_INVALID_FILE_PATH; }
   if (access(filepath, R_OK)) {
      dfail(JPEGGPU_ERR_INVALID_FILE_PATH, "Pas de fichier '%s'", filepath);
      jpeggpu_internal_set_error(g_derr);
      return JPEGGPU_ERR_INVALID_FILE_PATH;
   }
   if (outfile) {
      std::string o = outfile;
      size_t sl = o.rfind('/');
      std::string folder = sl == std::string::npos ? "." : (sl == 0 ? "/" : o.substr(0, sl));
      struct stat sb;
      if (stat(folder.c_str(), &sb) == -1) mkdir(folder.c_str(), S_IRWXU | S_IRGRP | S_IXGRP | S_IROTH | S_IXOTH);
   }
   jpeggpu_jpeg_t *j = nullptr;
   int rc;
   { Lap lap(tm.init_s); (void)jpeggpu_internal_ensure_init(); }   // a missing device is reported by the first stage that needs it
   { Lap lap(tm.header_s); rc = jpeggpu_jpeg_open_file(filepath, &j); }
   if (rc) return rc;
   jpeggpu_frame_desc_t desc;
   fill_desc(j->d, &desc);
   const size_t rb = jpeggpu_raster_row_bytes(&desc);
   int written = 0;

   if (j->d.num_sof == 0) {
      // decode_baseline_image, src/baseline.c:89-231: the output file exists (header + the complete MCU
      // rows) even when the scan turns out to be corrupt
      std::string name = out_name(filepath, outfile, desc.nbcomp, 0);
      const size_t total_rows = jpeggpu_raster_rows(&desc);
      int gpu_rc = 0;
      if (total_rows && rb) { Lap lap(tm.gpu_wait_s); gpu_rc = prepare_gpu(j, desc, true); }
      FILE *f = nullptr;
      int io_rc = 0;
      if (!gpu_rc) {
         f = fopen(name.c_str(), "w");
         if (!f) { io_rc = dfail(JPEGGPU_ERR_IO, "ouverture de '%s' impossible", name.c_str()); }
         else write_pnm_header(f, &desc);
      }
      int drc, wr_ok = 1;
      {
         BandWriter writer(f, j->device);                       // declared before the stream state: joined after the last job
         BaselineStream S;
         S.j = j; S.desc = desc; S.writer = &writer;
         for (int i = 0; i < 4; i++) S.qt[i] = j->d.qt_present[i] ? (const jpeggpu_qtable_t *)j->d.qt[i] : nullptr;
         S.band_bytes = (size_t)desc.max_v * 8 * rb;
         size_t in_per_row = 0;
         for (int c = 0; c < desc.nbcomp; c++) { S.row_in[c] = (size_t)desc.nbmcuH * desc.comp[c].hs * desc.comp[c].vs * sizeof(jpeggpu_blocl16_t); in_per_row += S.row_in[c]; }
         const bool streaming = f && total_rows && rb;
         if (streaming) {
            const size_t step = jpeggpu_internal_stripe_bytes() ? jpeggpu_internal_stripe_bytes() : ((size_t)8 << 20);
            j->d.row_hook = &BaselineStream::hook;
            j->d.row_hook_ctx = &S;
            j->d.row_hook_every = (uint32_t)std::max<size_t>(1, step / (in_per_row ? in_per_row : 1));
            writer.want_thread = j->d.row_hook_every < desc.nbmcuV;
         }
         { Lap lap(tm.vld_s); drc = j->d.decode_scan(); }
         j->d.row_hook = nullptr;
         if (drc && S.rc && drc == S.rc) drc = 0;                // the hook ended the scan: a GPU error (gpu_rc below), not a decoding error
         else if (drc) report(j, drc);
         std::string derr = j->d.err;
         const uint32_t mcu_rows = (drc || S.rc) ? j->d.mcu_rows_done : desc.nbmcuV;
         if (streaming) {
            { Lap lap(tm.gpu_wait_s); (void)S.send(mcu_rows); }  // the rows after the last hook (all of them for a small file)
            { Lap lap(tm.write_s); wr_ok = writer.drain(); }
            { Lap lap(tm.gpu_wait_s); int r = render_end(j); if (!S.rc) S.rc = r; }
            if (S.rc) gpu_rc = S.rc;
         }
         if (f) {
            Lap lap(tm.write_s);
            if (fclose(f) != 0) wr_ok = 0;
            if (!wr_ok && !io_rc) io_rc = dfail(JPEGGPU_ERR_IO, "écriture de '%s' incomplète", name.c_str());
            if (!io_rc && !gpu_rc) written = 1;
         }
         rc = gpu_rc ? gpu_rc : io_rc;
         if (io_rc && !gpu_rc) jpeggpu_internal_set_error(g_derr);
         if (drc) { jpeggpu_internal_set_error(derr.c_str()); rc = drc; }
      }
   } else {
      // decode_progressive_image, src/progressive.c:188-349: one image per scan.  Writing an image (10 ms for 50 MB) is
      // left to a thread of its own, so that it overlaps the entropy decoding of the following scans like the GPU work
      // does (a scan of a chroma component decodes faster than its image is written: the writer may lag up to three
      // images behind, images go round four pinned rasters).  Two files are written at a time; after a failed write no
      // further file is started.
      struct Writer {
         struct Job { std::string name; const uint8_t *raster; uint64_t id; };
         const jpeggpu_frame_desc_t &desc; size_t rb, nrows;
         std::thread th[2];                 // two files at a time: a 50 MB image takes 10 ms to write, a chroma scan 3 ms to decode
         std::mutex mu;
         std::condition_variable cv;
         std::deque<Job> q;
         std::vector<char> finished;        // per job
         uint64_t posted = 0, done = 0;     // done: jobs 0 .. done-1 are all finished
         int written = 0, rc = 0;
         std::string err;
         bool started = false, threaded = false, stop = false;
         Writer(const jpeggpu_frame_desc_t &d, size_t rb_, size_t nrows_) : desc(d), rb(rb_), nrows(nrows_) {}
         void run_one(const Job &job) {   // mu NOT held
            int r = 0; std::string e;
            bool skip; { std::lock_guard<std::mutex> lk(mu); skip = rc != 0; }
            if (!skip) { r = write_pnm(job.name.c_str(), &desc, job.raster, rb, nrows); if (r) e = g_derr; }   // g_derr is thread-local
            std::lock_guard<std::mutex> lk(mu);
            if (!skip) { if (r) { if (!rc) { rc = r; err = e; } } else written++; }
            finished[job.id] = 1;
            while (done < finished.size() && finished[done]) done++;
            cv.notify_all();
         }
         void loop() {
            for (;;) {
               Job job;
               { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || !q.empty(); }); if (q.empty()) return; job = q.front(); q.pop_front(); }
               run_one(job);
            }
         }
         void post(const std::string &name, const uint8_t *raster) {
            if (!started) {
               started = true;
               try { th[0] = std::thread([this] { loop(); }); threaded = true; } catch (const std::system_error &) { threaded = false; }
               if (threaded) { try { th[1] = std::thread([this] { loop(); }); } catch (const std::system_error &) {} }
            }
            Job job{ name, raster, 0 };
            { std::lock_guard<std::mutex> lk(mu); job.id = posted++; finished.push_back(0); if (threaded) q.push_back(job); }
            if (threaded) cv.notify_all(); else run_one(job);   // no thread to be had: write here
         }
         int wait_done(uint64_t n) { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done >= n; }); return rc; }   // the first n files are on disk (or failed)
         int status() { std::lock_guard<std::mutex> lk(mu); return rc; }
         ~Writer() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for (auto &t : th) if (t.joinable()) t.join(); }
      } wr(desc, rb, jpeggpu_raster_rows(&desc));
      uint64_t k = 1;                       // number of the image the next render produces (1-based, like the file names)
      std::string pending_name;
      const uint8_t *pending_raster = nullptr;
      auto writer_failed = [&](int r) { if (r) jpeggpu_internal_set_error(wr.err.c_str()); return r; };
      auto flush = [&]() -> int {   // wait for the queued render, then hand its raster to the writer
         if (pending_name.empty()) return 0;
         int r;
         { Lap lap(tm.gpu_wait_s); r = render_end(j); }
         if (!r) r = writer_failed(wr.status());
         if (!r) wr.post(pending_name, pending_raster);
         pending_name.clear();
         return r;
      };
      constexpr uint64_t kRasters = 4;
      uint64_t nrasters = 1;                // rasters in use: the handle's own + the h_alt that could be allocated
      while (!j->d.eoi_done) {
         int drc;
         { Lap lap(tm.vld_s); drc = report(j, j->d.decode_scan()); }   // overlaps the kernel + download of the previous scan
         std::string derr = j->d.err;
         rc = flush();
         if (drc) { jpeggpu_internal_set_error(derr.c_str()); rc = drc; }
         if (rc) break;
         // image k goes to raster (k-1) % nrasters: 0 = the handle's own, i = h_alt[i-1], allocated the first time round
         if (k - 1 == nrasters && nrasters < kRasters) {
            const size_t ob = jpeggpu_raster_rows(&desc) * rb;
            if (host_alloc((void **)&j->h_alt[nrasters - 1], ob ? ob : 16, j->device) == hipSuccess) nrasters++;
            else { (void)hipGetLastError(); j->h_alt[nrasters - 1] = nullptr; }
         }
         const uint64_t slot = (k - 1) % nrasters;
         uint8_t *const alt = slot ? j->h_alt[slot - 1] : nullptr;
         if (k > nrasters) {                // the file written from this raster (image k - nrasters) is complete
            Lap lap(tm.write_s);
            rc = writer_failed(wr.wait_done(k - nrasters));
            if (rc) break;
         }
         { Lap lap(tm.gpu_wait_s); rc = render_begin(j, alt, rb); }
         if (rc) break;
         pending_raster = alt ? alt : j->h_out;
         pending_name = out_name(filepath, outfile, desc.nbcomp, (uint8_t)k);   // uint8_t parameter, src/utils.c:56
         { Lap lap(tm.gpu_wait_s); rc = render_planes_free(j); }
         if (rc) break;
         int hrc;
         { Lap lap(tm.header_s); hrc = report(j, j->d.next_header()); }
         if (hrc) { std::string herr = j->d.err; rc = flush(); jpeggpu_internal_set_error(herr.c_str()); rc = hrc; break; }
         k++;
      }
      if (!rc) rc = flush();
      { Lap lap(tm.write_s); int w = wr.wait_done(wr.posted); if (!rc) rc = writer_failed(w); }   // every file handed over is on disk (or failed)
      written = wr.written;
      if (!rc && k > 1) {
         // src/progressive.c:332-343: "ln -f -s <last image> <name>"
         std::string base = out_name(filepath, outfile, desc.nbcomp, 0), last = out_name(filepath, outfile, desc.nbcomp, (uint8_t)(k - 1));
         size_t sl = last.rfind('/');
         std::string target = sl == std::string::npos ? last : last.substr(sl + 1);
         unlink(base.c_str());
         if (symlink(target.c_str(), base.c_str()) != 0) { /* the reference ignores ln's status */ }
      }
   }
   if (nb_images) *nb_images = written;
   jpeggpu_jpeg_close(j);
   return rc;
}

}  // extern "C"

// tools/probe_pcie.hip -- what the host link delivers for the host-pointer entry point's traffic shape:
// 100 MB up + 50 MB down (workload c2), as single copies, as concurrent copies on two streams, and cut into stripes.
// Build: hipcc -O2 --offload-arch=gfx950 -o tools/probe_pcie tools/probe_pcie.hip      Run on the GPU box.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void touch(unsigned *p, unsigned n) { unsigned i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = i; }

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class F> static double median_ms(int reps, F &&f) {
   std::vector<double> t;
   for (int i = 0; i < reps; i++) { double a = now_ms(); f(); t.push_back(now_ms() - a); }
   std::sort(t.begin(), t.end());
   return t[t.size() / 2];
}

int main() {
   const size_t IN = (size_t)4096 * 4096 * 3 * 2, OUT = (size_t)4096 * 4096 * 3;
   void *h_in, *h_out, *d_in, *d_out;
   CK(hipHostMalloc(&h_in, IN, hipHostMallocDefault)); CK(hipHostMalloc(&h_out, OUT, hipHostMallocDefault));
   CK(hipMalloc(&d_in, IN)); CK(hipMalloc(&d_out, OUT));
   memset(h_in, 1, IN); memset(h_out, 0, OUT);
   hipStream_t s[4];
   for (auto &x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
   auto sync_all = [&] { for (auto &x : s) (void)hipStreamSynchronize(x); };
   const int R = 15;
   auto report = [&](const char *name, double ms, size_t bytes) { printf("%-58s %7.3f ms  %6.1f GB/s\n", name, ms, bytes / ms / 1e6); };

   for (int w = 0; w < 3; w++) { (void)hipMemcpyAsync(d_in, h_in, IN, hipMemcpyHostToDevice, s[0]); (void)hipMemcpyAsync(h_out, d_out, OUT, hipMemcpyDeviceToHost, s[1]); sync_all(); }

   report("H2D 100 MB, one copy", median_ms(R, [&] { (void)hipMemcpyAsync(d_in, h_in, IN, hipMemcpyHostToDevice, s[0]); sync_all(); }), IN);
   report("D2H 50 MB, one copy", median_ms(R, [&] { (void)hipMemcpyAsync(h_out, d_out, OUT, hipMemcpyDeviceToHost, s[0]); sync_all(); }), OUT);
   report("H2D 100 MB || D2H 50 MB, two streams", median_ms(R, [&] {
      (void)hipMemcpyAsync(d_in, h_in, IN, hipMemcpyHostToDevice, s[0]); (void)hipMemcpyAsync(h_out, d_out, OUT, hipMemcpyDeviceToHost, s[1]); sync_all(); }), IN + OUT);
   report("H2D 100 MB split over two streams", median_ms(R, [&] {
      (void)hipMemcpyAsync(d_in, h_in, IN / 2, hipMemcpyHostToDevice, s[0]);
      (void)hipMemcpyAsync((char *)d_in + IN / 2, (char *)h_in + IN / 2, IN / 2, hipMemcpyHostToDevice, s[1]); sync_all(); }), IN);
   report("H2D 100 MB split over four streams", median_ms(R, [&] {
      for (int i = 0; i < 4; i++) (void)hipMemcpyAsync((char *)d_in + i * (IN / 4), (char *)h_in + i * (IN / 4), IN / 4, hipMemcpyHostToDevice, s[i]);
      sync_all(); }), IN);
   for (size_t chunk : { (size_t)1 << 20, (size_t)2 << 20, (size_t)4 << 20, (size_t)8 << 20, (size_t)16 << 20, (size_t)32 << 20 }) {
      char name[96];
      snprintf(name, sizeof name, "H2D 100 MB in %zu MiB copies, one stream", chunk >> 20);
      report(name, median_ms(R, [&] {
         for (size_t o = 0; o < IN; o += chunk) (void)hipMemcpyAsync((char *)d_in + o, (char *)h_in + o, std::min(chunk, IN - o), hipMemcpyHostToDevice, s[0]);
         sync_all(); }), IN);
      // the pipeline's shape: per stripe, upload on the stripe's stream, then download of half as many bytes on the same
      // stream (the kernel between them left out); stripes alternate between L streams
      for (int L : { 2, 3, 4 }) {
         snprintf(name, sizeof name, "stripes of %zu MiB up + half down, %d lanes", chunk >> 20, L);
         report(name, median_ms(R, [&] {
            int k = 0;
            for (size_t o = 0; o < IN; o += chunk, k++) {
               size_t n = std::min(chunk, IN - o);
               (void)hipMemcpyAsync((char *)d_in + o, (char *)h_in + o, n, hipMemcpyHostToDevice, s[k % L]);
               (void)hipMemcpyAsync((char *)h_out + o / 2, (char *)d_out + o / 2, n / 2, hipMemcpyDeviceToHost, s[k % L]);
            }
            sync_all(); }), IN + OUT);
      }
      // uploads all on one stream, downloads all on another, an event per stripe between them
      snprintf(name, sizeof name, "stripes of %zu MiB: up-stream + down-stream, events", chunk >> 20);
      std::vector<hipEvent_t> ev((IN + chunk - 1) / chunk);
      for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      report(name, median_ms(R, [&] {
         int k = 0;
         for (size_t o = 0; o < IN; o += chunk, k++) {
            size_t n = std::min(chunk, IN - o);
            (void)hipMemcpyAsync((char *)d_in + o, (char *)h_in + o, n, hipMemcpyHostToDevice, s[0]);
            (void)hipEventRecord(ev[k], s[0]);
            (void)hipStreamWaitEvent(s[1], ev[k], 0);
            (void)hipMemcpyAsync((char *)h_out + o / 2, (char *)d_out + o / 2, n / 2, hipMemcpyDeviceToHost, s[1]);
         }
         sync_all(); }), IN + OUT);
      for (auto &e : ev) (void)hipEventDestroy(e);
   }
   // the batch shape: units of one 1920x1080 4:2:0 frame = uploads of 4.18 + 1.04 + 1.04 MB, download of 6.27 MB.
   // (a) all uploads on one stream  (b) one upload stream per component  (c) two frames per unit on one stream
   {
      const size_t Y = (size_t)240 * 136 * 128, Cc = (size_t)120 * 68 * 128, O = (size_t)1920 * 1088 * 3;
      const int N = 8;    // frames that fit the 50 MB raster buffer (8 x 6 266 880 bytes)
      static_assert((size_t)8 * 1920 * 1088 * 3 <= (size_t)4096 * 4096 * 3, "raster buffer too small");
      hipEvent_t ev[3][16];
      for (auto &r : ev) for (auto &e : r) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      hipEvent_t kd[16];
      for (auto &e : kd) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      // the library's pipeline: + a kernel between the wait and the download, + the upload of unit k waiting for the
      // kernel of unit k-3 (device slots)
      auto run_lib = [&](bool kernel, bool back_edge, int reps) {
         for (int r = 0; r < reps; r++)
         for (int k = 0; k < N; k++) {
            size_t o = (size_t)k * (Y + 2 * Cc);
            const size_t sz[3] = { Y, Cc, Cc };
            if (back_edge && (k >= 3 || r > 0)) (void)hipStreamWaitEvent(s[0], kd[(k + N - 3) % N], 0);
            for (int c = 0; c < 3; c++) { (void)hipMemcpyAsync((char *)d_in + o, (char *)h_in + o, sz[c], hipMemcpyHostToDevice, s[0]); o += sz[c]; }
            (void)hipEventRecord(ev[0][k], s[0]);
            (void)hipStreamWaitEvent(s[3], ev[0][k], 0);
            if (kernel) touch<<<256, 256, 0, s[3]>>>((unsigned *)d_out + (size_t)k * (O / 4), 1024);
            (void)hipEventRecord(kd[k], s[3]);
            (void)hipMemcpyAsync((char *)h_out + (size_t)k * O, (char *)d_out + (size_t)k * O, O, hipMemcpyDeviceToHost, s[3]);
         }
         sync_all();
      };
      auto run = [&](int nup, int group) {
         for (int k = 0; k < N; k += group) {
            const size_t in0 = (size_t)k * (Y + 2 * Cc), sz[3] = { Y * group, Cc * group, Cc * group };
            size_t o = in0;
            for (int c = 0; c < 3; c++) {
               hipStream_t su = s[c % nup];
               (void)hipMemcpyAsync((char *)d_in + o, (char *)h_in + o, sz[c], hipMemcpyHostToDevice, su);
               o += sz[c];
               if (c == 2 || nup > 1) { (void)hipEventRecord(ev[c][k], su); (void)hipStreamWaitEvent(s[3], ev[c][k], 0); }
            }
            (void)hipMemcpyAsync((char *)h_out + (size_t)k * O, (char *)d_out + (size_t)k * O, O * group, hipMemcpyDeviceToHost, s[3]);
         }
         sync_all();
      };
      const size_t bytes = (size_t)N * (Y + 2 * Cc + O);
      report("1080p units: 3 uploads on ONE stream + download stream", median_ms(R, [&] { run(1, 1); }), bytes);
      report("1080p units: one upload stream per component + download", median_ms(R, [&] { run(3, 1); }), bytes);
      report("1080p units, 2 frames each: one upload stream + download", median_ms(R, [&] { run(1, 2); }), bytes);
      report("1080p units, 4 frames each: one upload stream + download", median_ms(R, [&] { run(1, 4); }), bytes);
      report("1080p units, 2 frames each: three upload streams + download", median_ms(R, [&] { run(3, 2); }), bytes);
      report("8 x 1080p units x 8 rounds: as (a)", median_ms(R, [&] { run_lib(false, false, 8); }), bytes * 8);
      report("8 x 1080p units x 8 rounds: + kernel before the download", median_ms(R, [&] { run_lib(true, false, 8); }), bytes * 8);
      report("8 x 1080p units x 8 rounds: + upload waits for kernel k-3", median_ms(R, [&] { run_lib(false, true, 8); }), bytes * 8);
      report("8 x 1080p units x 8 rounds: + both", median_ms(R, [&] { run_lib(true, true, 8); }), bytes * 8);
   }
   return 0;
}

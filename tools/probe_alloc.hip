// tools/probe_alloc.hip -- what the allocations of one 4096x4096 4:4:4 decode cost (jpeggpu_jpeg_open_* + first render):
// pinned planes, their clearing, device planes, device raster, pinned raster.  Build: hipcc -O2 --offload-arch=gfx950 -o tools/probe_alloc tools/probe_alloc.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define T(name, stmt) do { double a_ = now_ms(); stmt; printf("  %-44s %8.3f ms\n", name, now_ms() - a_); } while (0)
int main() {
   const size_t IN = (size_t)4096 * 4096 * 3 * 2, OUT = (size_t)4096 * 4096 * 3;
   (void)hipFree(nullptr);
   for (int rep = 0; rep < 3; rep++) {
      printf("pass %d\n", rep);
      void *h_in = nullptr, *h_out = nullptr, *d_in = nullptr, *d_out = nullptr, *m = nullptr;
      T("hipHostMalloc 100 MB", (void)hipHostMalloc(&h_in, IN, hipHostMallocDefault));
      T("memset 100 MB (pinned, first touch)", memset(h_in, 0, IN));
      T("memset 100 MB (pinned, again)", memset(h_in, 0, IN));
      T("malloc + memset 100 MB (pageable)", (m = malloc(IN), memset(m, 0, IN)));
      T("hipMalloc 100 MB", (void)hipMalloc(&d_in, IN));
      T("hipMalloc 50 MB", (void)hipMalloc(&d_out, OUT));
      T("hipHostMalloc 50 MB", (void)hipHostMalloc(&h_out, OUT, hipHostMallocDefault));
      T("H2D 100 MB (first)", (void)hipMemcpy(d_in, h_in, IN, hipMemcpyHostToDevice));
      T("H2D 100 MB (again)", (void)hipMemcpy(d_in, h_in, IN, hipMemcpyHostToDevice));
      T("D2H 50 MB (first)", (void)hipMemcpy(h_out, d_out, OUT, hipMemcpyDeviceToHost));
      T("D2H 50 MB (again)", (void)hipMemcpy(h_out, d_out, OUT, hipMemcpyDeviceToHost));
      T("hipFree x2", ((void)hipFree(d_in), (void)hipFree(d_out)));
      T("hipHostFree x2", ((void)hipHostFree(h_in), (void)hipHostFree(h_out)));
      free(m);
   }
   return 0;
}

// ubench_policy.hip -- cache-policy bits on the kernel's 12-byte raster stores (c2-shaped skeleton M1 of ubench_skel.hip):
// which of nt / sc0 / sc1 combinations streams fastest?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
#define ST(mods) asm volatile("global_store_dwordx3 %0, %1, off " mods :: "v"(dst), "v"(val) : "memory")
template <int POL> __global__ void __launch_bounds__(192) k(const u32x4 *in, uint8_t *out, uint32_t tiles_per_row) {
   const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
   const size_t wg = blockIdx.x;
   const size_t plane = (size_t)gridDim.x * 64 * 8;
   const u32x4 *src = in + wave * plane + (wg * 64 + lane) * 8;
   u32x4 v[8];
#pragma unroll
   for (int i = 0; i < 8; i++) v[i] = src[i];
   const size_t trow = wg / tiles_per_row, tcol = wg % tiles_per_row;
   uint8_t *tile_out = out + trow * (8 * (size_t)tiles_per_row * 1536) + tcol * 1536;
   const size_t stride = (size_t)tiles_per_row * 1536;
   int k2 = 0;
   for (uint32_t run = wave; run < 16; run += 3, k2++) {
      const uint32_t row = run >> 1, half = run & 1;
      uint8_t *dst = tile_out + row * stride + half * 768 + lane * 12;
      const u32x3 val = { v[k2 & 7].x ^ v[(k2 + 1) & 7].y, v[k2 & 7].z, v[(k2 + 2) & 7].w };
      if (POL == 0) ST("");
      else if (POL == 1) ST("nt");
      else if (POL == 2) ST("sc0");
      else if (POL == 3) ST("sc1");
      else if (POL == 4) ST("sc0 sc1");
      else if (POL == 5) ST("nt sc0");
      else if (POL == 6) ST("nt sc1");
      else ST("nt sc0 sc1");
   }
}
template <int POL> int run(const char *name, const u32x4 *d, uint8_t *o, unsigned wgs) {
   hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
   for (int w = 0; w < 40; w++) hipLaunchKernelGGL((k<POL>), dim3(wgs), dim3(192), 0, 0, d, o, 8u);
   CHECK(hipDeviceSynchronize());
   const int REP = 20;
   CHECK(hipEventRecord(a));
   for (int r = 0; r < REP; r++) hipLaunchKernelGGL((k<POL>), dim3(wgs), dim3(192), 0, 0, d, o, 8u);
   CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
   float ms; CHECK(hipEventElapsedTime(&ms, a, b));
   double us = ms * 1e3 / REP, bytes = (double)wgs * 192 * (128 + 64);
   printf("stores: %-12s %8.1f us  %6.0f GB/s  %.1f %%\n", name, us, bytes / (us * 1e-6) / 1e9, bytes / (us * 1e-6) / 8e12 * 100);
   return 0;
}
int main() {
   const unsigned wgs = 16 * 4096;
   const size_t in_bytes = (size_t)wgs * 192 * 128, out_bytes = (size_t)wgs * 192 * 64;
   u32x4 *d; uint8_t *o; CHECK(hipMalloc((void **)&d, in_bytes)); CHECK(hipMalloc((void **)&o, out_bytes)); CHECK(hipMemset(d, 1, in_bytes));
   for (int rep = 0; rep < 2; rep++) {
      run<0>("(default)", d, o, wgs); run<1>("nt", d, o, wgs); run<2>("sc0", d, o, wgs); run<3>("sc1", d, o, wgs);
      run<4>("sc0 sc1", d, o, wgs); run<5>("nt sc0", d, o, wgs); run<6>("nt sc1", d, o, wgs); run<7>("nt sc0 sc1", d, o, wgs);
   }
   return 0;
}

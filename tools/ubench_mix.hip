// ubench_mix.hip -- what the HBM system of an MI355X delivers for fully coalesced streams at the back end's
// read:write mix (2 bytes of coefficients in per byte of pixels out) compared with a 1:1 copy, read-only and write-only.
// 16 bytes per lane per access, grid-stride-free (one pass), sized like workload c2 at 16 frames (1.6 GB in, 0.8 GB out).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// R reads and W writes of 16 bytes per lane; each wave works on R (and W) consecutive 1 KiB lines
template <int R, int W, int NT> __global__ void __launch_bounds__(256) k(const u32x4 *in, u32x4 *out) {
   const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
   const int lane = threadIdx.x & 63;
   u32x4 acc = { 0, 0, 0, 0 };
#pragma unroll
   for (int i = 0; i < R; i++) acc ^= in[(wave * R + i) * 64 + lane];
   if (W == 0) { if (acc.x == 0x12345678u) out[0] = acc; }
#pragma unroll
   for (int j = 0; j < W; j++) {
      u32x4 v = acc + (uint32_t)j;
      if (NT) __builtin_nontemporal_store(v, &out[(wave * W + j) * 64 + lane]); else out[(wave * W + j) * 64 + lane] = v;
   }
}
template <int R, int W, int NT> int run(const char *name, const u32x4 *d, u32x4 *o, size_t units) {
   // a unit = one wave's work: R KiB in, W KiB out
   unsigned blocks = (unsigned)(units / 4);
   hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
   for (int w = 0; w < 100; w++) hipLaunchKernelGGL((k<R, W, NT>), dim3(blocks), dim3(256), 0, 0, d, o);
   CHECK(hipDeviceSynchronize());
   const int REP = 30;
   CHECK(hipEventRecord(a));
   for (int r = 0; r < REP; r++) hipLaunchKernelGGL((k<R, W, NT>), dim3(blocks), dim3(256), 0, 0, d, o);
   CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
   float ms; CHECK(hipEventElapsedTime(&ms, a, b));
   double us = ms * 1e3 / REP, bytes = (double)units * 1024 * (R + W);
   printf("%-40s %8.1f us  %6.0f GB/s  %.1f %% of 8 TB/s\n", name, us, bytes / (us * 1e-6) / 1e9, bytes / (us * 1e-6) / 8e12 * 100);
   return 0;
}
int main() {
   const size_t in_bytes = (size_t)16 * 786432 * 128, out_bytes = in_bytes;   // buffers: 1.6 GB each
   u32x4 *d, *o; CHECK(hipMalloc((void **)&d, in_bytes)); CHECK(hipMalloc((void **)&o, out_bytes)); CHECK(hipMemset(d, 1, in_bytes));
   const size_t kib = in_bytes / 1024;
   for (int rep = 0; rep < 2; rep++) {
      run<2, 1, 1>("coalesced 2 reads : 1 write, nt stores", d, o, kib / 2);
      run<2, 1, 0>("coalesced 2 reads : 1 write", d, o, kib / 2);
      run<8, 4, 1>("coalesced 8 reads : 4 writes, nt stores", d, o, kib / 8);
      run<1, 1, 1>("copy 1 : 1, nt stores", d, o, kib);
      run<1, 1, 0>("copy 1 : 1", d, o, kib);
      run<4, 4, 0>("copy 4 : 4", d, o, kib / 4);
      run<4, 0, 0>("read only (4 per wave)", d, o, kib / 4);
      run<0, 4, 1>("write only (4 per wave), nt", d, o, kib / 4);
      run<0, 4, 0>("write only (4 per wave)", d, o, kib / 4);
   }
   return 0;
}

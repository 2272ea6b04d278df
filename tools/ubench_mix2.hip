// ubench_mix2.hip -- follow-up to ubench_mix.hip: why does a wave that reads 8 KiB and then writes 4 KiB (the shape of the
// back end's waves) stream at 71 % of the peak when waves of 2 KiB + 1 KiB reach 75.6 %?  Varies the work per wave, the
// number of resident waves (LDS allocation) and the order of issue inside a wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// MODE 0: all R reads, then all W writes.  MODE 1: (2 reads, 1 write) repeated.  MODE 2: record-shaped reads (lane L reads
// the 8 x 16 bytes of its own 128-byte record), then W writes of 16 B per lane.
template <int R, int W, int MODE> __global__ void __launch_bounds__(256) k(const u32x4 *in, u32x4 *out) {
   extern __shared__ uint32_t pad[];
   const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
   const int lane = threadIdx.x & 63;
   if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < W; j++) {
         u32x4 a = in[(wave * R + 2 * j) * 64 + lane] ^ in[(wave * R + 2 * j + 1) * 64 + lane];
         __builtin_nontemporal_store(a, &out[(wave * W + j) * 64 + lane]);
      }
      return;
   }
   u32x4 v[R];
#pragma unroll
   for (int i = 0; i < R; i++) v[i] = MODE == 2 ? in[wave * R * 64 + lane * R + i] : in[(wave * R + i) * 64 + lane];
#pragma unroll
   for (int j = 0; j < W; j++) __builtin_nontemporal_store(v[2 * j] ^ v[2 * j + 1], &out[(wave * W + j) * 64 + lane]);
   if (threadIdx.x == 1000) pad[0] = 1;
}
template <int R, int W, int MODE> int run(const char *name, const u32x4 *d, u32x4 *o, size_t kib_in, size_t lds) {
   const size_t units = kib_in / R;
   unsigned blocks = (unsigned)(units / 4);
   if (lds > 48 * 1024) CHECK(hipFuncSetAttribute((const void *)k<R, W, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
   hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
   for (int w = 0; w < 60; w++) hipLaunchKernelGGL((k<R, W, MODE>), dim3(blocks), dim3(256), lds, 0, d, o);
   CHECK(hipDeviceSynchronize());
   const int REP = 20;
   CHECK(hipEventRecord(a));
   for (int r = 0; r < REP; r++) hipLaunchKernelGGL((k<R, W, MODE>), dim3(blocks), dim3(256), lds, 0, d, o);
   CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
   float ms; CHECK(hipEventElapsedTime(&ms, a, b));
   double us = ms * 1e3 / REP, bytes = (double)units * 1024 * (R + W);
   printf("%-58s lds=%6zu %8.1f us  %6.0f GB/s  %.1f %%\n", name, lds, us, bytes / (us * 1e-6) / 1e9, bytes / (us * 1e-6) / 8e12 * 100);
   return 0;
}
int main() {
   const size_t in_bytes = (size_t)16 * 786432 * 128;
   u32x4 *d, *o; CHECK(hipMalloc((void **)&d, in_bytes)); CHECK(hipMalloc((void **)&o, in_bytes)); CHECK(hipMemset(d, 1, in_bytes));
   const size_t kib = in_bytes / 1024;
   run<2, 1, 0>("2 reads then 1 write per wave", d, o, kib, 0);
   run<4, 2, 0>("4 reads then 2 writes", d, o, kib, 0);
   run<8, 4, 0>("8 reads then 4 writes", d, o, kib, 0);
   run<16, 8, 0>("16 reads then 8 writes", d, o, kib, 0);
   run<8, 4, 1>("8 : 4 issued as 4 x (2 reads, 1 write)", d, o, kib, 0);
   run<8, 4, 2>("8 : 4, record-shaped reads (lane = 128-byte record)", d, o, kib, 0);
   // fewer resident waves: 160 KiB of LDS per CU, 4 waves per workgroup
   for (size_t lds : { (size_t)20000, (size_t)26000, (size_t)40000, (size_t)64000 }) {
      run<8, 4, 0>("8 reads then 4 writes", d, o, kib, lds);
      run<8, 4, 2>("8 : 4, record-shaped reads", d, o, kib, lds);
   }
   run<2, 1, 0>("2 reads then 1 write per wave", d, o, kib, 40000);
   return 0;
}

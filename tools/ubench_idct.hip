// ubench_idct.hip -- pure-VALU throughput of the block transform: scalar vs packed-fp32 form.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../jpeg-decoder_amd/csrc/jpeg_math.hpp"
using namespace jpegmath;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 64;
template <int KIND> __global__ void __launch_bounds__(256) k(uint2 *out, int seed) {
   float m[8][8];
   uint2 rows[8] = {};
   uint32_t acc = 0;
   for (int it = 0; it < ITER; it++) {
#pragma unroll
      for (int c = 0; c < 8; c++)
#pragma unroll
         for (int r = 0; r < 8; r++) m[c][r] = (float)(int)((threadIdx.x * 37 + c * 8 + r + it * 11 + seed + (rows[r].x & 255)) % 211) - 100.0f;
      if (KIND == 0) idct_block(m, rows); else idct_block_pk(m, rows);
#pragma unroll
      for (int r = 0; r < 8; r++) acc += rows[r].x ^ rows[r].y;
   }
   out[blockIdx.x * blockDim.x + threadIdx.x] = make_uint2(acc, rows[3].y);
}
template <int KIND> int run(const char *name, uint2 *d) {
   int blocks = 256 * 4 * 8;
   hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
   hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1);
   CHECK(hipDeviceSynchronize());
   CHECK(hipEventRecord(a));
   hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1);
   CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
   float ms; CHECK(hipEventElapsedTime(&ms, a, b));
   double wave_blocks_per_simd = (double)blocks * 4 / 1024 * ITER;
   unsigned h[2]; CHECK(hipMemcpy(h, d, 8, hipMemcpyDeviceToHost));
   printf("%-10s %8.3f ms  %.1f ns per 64-block wave-transform per SIMD   (check %08x)\n", name, ms, ms * 1e6 / wave_blocks_per_simd, h[0]);
   return 0;
}
int main() {
   uint2 *d; CHECK(hipMalloc(&d, 256 * 4 * 8 * 256 * 8));
   run<0>("scalar", d); run<1>("packed", d); run<0>("scalar", d); run<1>("packed", d);
   return 0;
}

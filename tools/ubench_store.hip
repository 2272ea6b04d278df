// ubench_store.hip -- the store side of the back end's skeleton: every lane reads one 128-byte record, a workgroup of
// NW waves writes ROWS raster rows of SEG bytes each (rows `pitch` bytes apart), one store instruction = 64 lanes x LB bytes.
// Which (SEG, LB) shapes does the memory system like?  2 bytes read per byte written throughout.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
// NW waves; each lane reads 128 B and the workgroup writes NW*64*64 bytes = ROWS * SEG
template <int NW, int LB, int SEG> __global__ void __launch_bounds__(NW * 64) k(const u32x4 *in, uint8_t *out, uint32_t tiles_per_row) {
   constexpr int RUN = 64 * LB;                   // bytes per store instruction
   constexpr int TOTAL = NW * 64 * 64;            // bytes per workgroup
   constexpr int NRUNS = TOTAL / RUN;
   constexpr int ROWS = TOTAL / SEG;
   constexpr int RPR = SEG / RUN;                 // runs per row
   static_assert(TOTAL % SEG == 0 && SEG % RUN == 0, "shape");
   const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
   const size_t wg = blockIdx.x;
   const u32x4 *src = in + ((size_t)wg * NW * 64 + tid) * 8;
   u32x4 v[8];
#pragma unroll
   for (int i = 0; i < 8; i++) v[i] = src[i];
   const size_t pitch = (size_t)tiles_per_row * SEG;
   uint8_t *tile_out = out + (wg / tiles_per_row) * (ROWS * pitch) + (wg % tiles_per_row) * SEG;
   int k2 = 0;
   for (int run = wave; run < NRUNS; run += NW, k2++) {
      const int row = run / RPR, part = run % RPR;
      uint8_t *dst = tile_out + row * pitch + part * RUN + lane * LB;
      const u32x4 a = v[k2 & 7] ^ v[(k2 + 3) & 7];
      if (LB == 16) __builtin_nontemporal_store(a, reinterpret_cast<u32x4 *>(dst));
      else __builtin_nontemporal_store(u32x3{a.x, a.y, a.z}, reinterpret_cast<u32x3 __attribute__((aligned(4))) *>(dst));
   }
}
template <int NW, int LB, int SEG> int run(const u32x4 *d, uint8_t *o, size_t in_bytes, unsigned row_bytes) {
   unsigned tpr = row_bytes / SEG;
   unsigned wgs = (unsigned)(in_bytes / (NW * 64 * 128)) / tpr * tpr;
   hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
   for (int w = 0; w < 40; w++) hipLaunchKernelGGL((k<NW, LB, SEG>), dim3(wgs), dim3(NW * 64), 0, 0, d, o, tpr);
   CHECK(hipDeviceSynchronize());
   const int REP = 20;
   CHECK(hipEventRecord(a));
   for (int r = 0; r < REP; r++) hipLaunchKernelGGL((k<NW, LB, SEG>), dim3(wgs), dim3(NW * 64), 0, 0, d, o, tpr);
   CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
   float ms; CHECK(hipEventElapsedTime(&ms, a, b));
   double us = ms * 1e3 / REP, bytes = (double)wgs * NW * 64 * 192;
   printf("waves/wg=%d lane=%2d B  segment=%5d B (%2d rows/wg, pitch %6u)  %8.1f us  %6.0f GB/s  %.1f %%\n", NW, LB, SEG, NW * 4096 / SEG, row_bytes, us,
          bytes / (us * 1e-6) / 1e9, bytes / (us * 1e-6) / 8e12 * 100);
   return 0;
}
int main() {
   const size_t in_bytes = (size_t)16 * 786432 * 128;
   u32x4 *d; uint8_t *o; CHECK(hipMalloc((void **)&d, in_bytes)); CHECK(hipMalloc((void **)&o, in_bytes / 2 + (1 << 20))); CHECK(hipMemset(d, 1, in_bytes));
   // 3 waves (c2's 192-lane workgroups: 12288 B per workgroup)
   run<3, 12, 768>(d, o, in_bytes, 12288);
   run<3, 12, 1536>(d, o, in_bytes, 12288);      // c2 today: 8 rows x 1536 B
   run<3, 12, 3072>(d, o, in_bytes, 12288);
   run<3, 12, 6144>(d, o, in_bytes, 12288);
   run<3, 12, 12288>(d, o, in_bytes, 12288);     // one whole row per workgroup
   run<3, 16, 1024>(d, o, in_bytes, 12288);
   run<3, 16, 2048>(d, o, in_bytes, 12288);
   run<3, 16, 4096>(d, o, in_bytes, 12288);
   run<3, 16, 12288>(d, o, in_bytes, 12288);
   // 6 waves (384 lanes: 24576 B per workgroup)
   run<6, 12, 1536>(d, o, in_bytes, 12288);
   run<6, 12, 3072>(d, o, in_bytes, 12288);      // T = 128 for 4:4:4; also 4:2:0 / 4:2:2 today (with 24576-byte rows)
   run<6, 12, 6144>(d, o, in_bytes, 12288);
   run<6, 12, 12288>(d, o, in_bytes, 12288);
   run<6, 12, 3072>(d, o, in_bytes, 24576);
   // 4 waves
   run<4, 16, 2048>(d, o, in_bytes, 12288);
   run<4, 16, 4096>(d, o, in_bytes, 12288);
   return 0;
}

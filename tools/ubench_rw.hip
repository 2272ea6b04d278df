// ubench_rw.hip -- the back end's data-movement skeleton without LDS, barrier or arithmetic: every lane
// pulls one 128-byte record (8 x 16 B) and the wave writes 64 (or 60) bytes per lane in different
// store shapes.  Sized like workload c2 (402 MB in, 201 MB out).  Answers: which store shape / cache
// policy reaches the copy rate, and how far the real kernel (LDS + barrier + phases) is from that.
//   S0  16 B per lane, lanes contiguous (1 KiB per instruction), 4 instructions
//   S1  12 B per lane, lanes contiguous (768 B per instruction), 5 instructions (60 B per lane)
//   S2  as S1 but the 5 instructions go to 5 raster rows 12288 B apart (what the kernel's tile does)
//   S3  as S0 but rows 12288 B apart
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int SHAPE, int NT> __global__ void __launch_bounds__(256) k(const uint4 *in, uint8_t *out) {
   const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
   const int lane = threadIdx.x & 63;
   const uint4 *base = in + wave * 512;
   uint4 v[8];
#pragma unroll
   for (int i = 0; i < 8; i++) v[i] = base[lane * 8 + i];
   if (SHAPE == 0 || SHAPE == 3) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
         u32x4 w = { v[2 * j].x ^ v[2 * j + 1].x, v[2 * j].y ^ v[2 * j + 1].y, v[2 * j].z ^ v[2 * j + 1].z, v[2 * j].w ^ v[2 * j + 1].w };
         uint8_t *dst = SHAPE == 0 ? out + wave * 4096 + j * 1024 + lane * 16
                                   : out + (wave >> 3) * (8 * 4096) + (size_t)j * 8192 + (wave & 7) * 1024 + lane * 16;
         if (NT) __builtin_nontemporal_store(w, reinterpret_cast<u32x4 *>(dst)); else *reinterpret_cast<u32x4 *>(dst) = w;
      }
   } else {
#pragma unroll
      for (int j = 0; j < 5; j++) {
         u32x3 w = { v[j].x ^ v[j + 3].x, v[j].y ^ v[j + 3].y, v[j].z ^ v[j + 3].w };
         // S1: wave-contiguous 3840 B; S2: 5 rows of a 16-wave-wide band (row = 16 * 768 = 12288 B)
         uint8_t *dst = SHAPE == 1 ? out + wave * 3840 + j * 768 + lane * 12
                                   : out + (wave >> 4) * (5 * 12288) + (size_t)j * 12288 + (wave & 15) * 768 + lane * 12;
         if (NT) __builtin_nontemporal_store(w, reinterpret_cast<u32x3 __attribute__((aligned(4))) *>(dst));
         else *reinterpret_cast<u32x3 __attribute__((aligned(4))) *>(dst) = w;
      }
   }
}
template <int SHAPE, int NT> int run(const char *name, const uint4 *d, uint8_t *o, size_t in_bytes, int threads) {
   size_t nrec = in_bytes / 128; unsigned blocks = (unsigned)(nrec / threads);
   size_t out_bytes = nrec * ((SHAPE == 0 || SHAPE == 3) ? 64 : 60);
   hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
   for (int w = 0; w < 200; w++) hipLaunchKernelGGL((k<SHAPE, NT>), dim3(blocks), dim3(threads), 0, 0, d, o);   // clocks up
   CHECK(hipDeviceSynchronize());
   const int R = 50;
   CHECK(hipEventRecord(a));
   for (int r = 0; r < R; r++) hipLaunchKernelGGL((k<SHAPE, NT>), dim3(blocks), dim3(threads), 0, 0, d, o);
   CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
   float ms; CHECK(hipEventElapsedTime(&ms, a, b));
   double us = ms * 1e3 / R;
   printf("%-44s threads=%3d  %.1f us  %.0f GB/s (in+out)\n", name, threads, us, (in_bytes + out_bytes) / (us * 1e-6) / 1e9);
   return 0;
}
int main() {
   size_t in_bytes = (size_t)4 * 786432 * 128;   // c2: 4 frames
   uint4 *d; uint8_t *o; CHECK(hipMalloc(&d, in_bytes)); CHECK(hipMalloc(&o, in_bytes)); CHECK(hipMemset(d, 1, in_bytes));
   for (int t : {192, 256}) {
      run<0, 0>("S0 16B/lane contiguous", d, o, in_bytes, t);
      run<0, 1>("S0 16B/lane contiguous, nt", d, o, in_bytes, t);
      run<1, 0>("S1 12B/lane contiguous", d, o, in_bytes, t);
      run<1, 1>("S1 12B/lane contiguous, nt", d, o, in_bytes, t);
      run<2, 0>("S2 12B/lane, 5 rows 12288 B apart", d, o, in_bytes, t);
      run<2, 1>("S2 12B/lane, 5 rows 12288 B apart, nt", d, o, in_bytes, t);
      run<3, 0>("S3 16B/lane, 4 rows 8192 B apart", d, o, in_bytes, t);
      run<3, 1>("S3 16B/lane, 4 rows 8192 B apart, nt", d, o, in_bytes, t);
   }
   return 0;
}

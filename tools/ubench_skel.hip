// ubench_skel.hip -- which part of the back end's data-movement skeleton costs bandwidth?  c2-shaped: workgroups of 192
// lanes, every lane reads one 128-byte record, the workgroup writes 8 raster rows x 1536 bytes (rows 12288 B apart).
//   M0  records in, 16-byte contiguous stores out (4 per lane... 64 B per lane), no LDS           [reference point]
//   M1  records in, 12-byte stores to the 8 strided rows (5.33 per lane), no LDS
//   M2  M1 + every lane parks 64 bytes in LDS and the stores read 3 x 4 bytes back; wave-local (no barrier; 3 x 64-lane tiles)
//   M3  M2 with the workgroup barrier between the phases (the kernel's structure)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
template <int M> __global__ void __launch_bounds__(192) k(const u32x4 *in, uint8_t *out, uint32_t tiles_per_row) {
   __shared__ __attribute__((aligned(16))) uint8_t lds[192 * 64];
   const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
   const size_t wg = blockIdx.x;
   // three planes of 64 records each per workgroup (Y, Cb, Cr): plane c at in + c * plane_size
   const size_t plane = (size_t)gridDim.x * 64 * 8;          // u32x4 units per plane
   const u32x4 *src = in + wave * plane + (wg * 64 + lane) * 8;
   u32x4 v[8];
#pragma unroll
   for (int i = 0; i < 8; i++) v[i] = src[i];
   uint32_t w[8][2];
#pragma unroll
   for (int i = 0; i < 8; i++) { w[i][0] = v[i].x ^ v[i].z; w[i][1] = v[i].y ^ v[i].w; }
   const size_t trow = wg / tiles_per_row, tcol = wg % tiles_per_row;
   uint8_t *tile_out = out + trow * (8 * (size_t)tiles_per_row * 1536) + tcol * 1536;
   const size_t stride = (size_t)tiles_per_row * 1536;
   if (M == 0) {
      uint8_t *o = out + (wg * 192 + tid) * 64;
#pragma unroll
      for (int j = 0; j < 4; j++) __builtin_nontemporal_store(u32x4{w[2 * j][0], w[2 * j][1], w[2 * j + 1][0], w[2 * j + 1][1]}, reinterpret_cast<u32x4 *>(out + ((wg * 4 + j) * 192 + tid) * 16));
      (void)o;
      return;
   }
   if (M == 4) {   // 16-byte stores, 48 lanes per run, same 8 strided rows (what an LDS re-shuffle of the RGB bytes would give)
      int k2 = 0;
      for (uint32_t run = wave; run < 16; run += 3, k2++) {
         const uint32_t row = run >> 1, half = run & 1;
         if (lane < 48)
            __builtin_nontemporal_store(u32x4{w[k2][0], w[k2][1], w[k2 + 1][0], w[k2 + 1][1]},
                                        reinterpret_cast<u32x4 *>(tile_out + row * stride + half * 768 + lane * 16));
      }
      return;
   }
   if (M == 5) {   // 12-byte stores, but the tile's 8 x 1536 bytes are contiguous (isolates the row stride)
      int k2 = 0;
      uint8_t *o = out + wg * 12288;
      for (uint32_t run = wave; run < 16; run += 3, k2++)
         __builtin_nontemporal_store(u32x3{w[k2][0], w[k2][1], w[k2 + 1][0]},
                                     reinterpret_cast<u32x3 __attribute__((aligned(4))) *>(o + run * 768 + lane * 12));
      return;
   }
   if (M == 6) {   // 16-byte stores by all 64 lanes (1024-byte runs), 12 runs per workgroup, rows 12288 B apart, 2 runs... rows of 2048 B
      int k2 = 0;
      uint8_t *t2 = out + (wg / 6) * (6 * (size_t)12288) + (wg % 6) * 2048;
      for (uint32_t run = wave; run < 12; run += 3, k2++) {
         const uint32_t row = run >> 1, half = run & 1;
         __builtin_nontemporal_store(u32x4{w[k2][0], w[k2][1], w[k2 + 1][0], w[k2 + 1][1]},
                                     reinterpret_cast<u32x4 *>(t2 + row * 12288 + half * 1024 + lane * 16));
      }
      return;
   }
   if (M == 7) {   // M1 without the non-temporal hint
      int k2 = 0;
      for (uint32_t run = wave; run < 16; run += 3, k2++) {
         const uint32_t row = run >> 1, half = run & 1;
         *reinterpret_cast<u32x3 __attribute__((aligned(4))) *>(tile_out + row * stride + half * 768 + lane * 12) = u32x3{w[k2][0], w[k2][1], w[k2 + 1][0]};
      }
      return;
   }
   if (M == 8) {   // M1 with every wave writing whole rows: wave w takes rows w, w+3, ... (both halves back to back)
      int k2 = 0;
      for (uint32_t row = wave; row < 8; row += 3, k2 += 2) {
         __builtin_nontemporal_store(u32x3{w[k2][0], w[k2][1], w[k2 + 1][0]}, reinterpret_cast<u32x3 __attribute__((aligned(4))) *>(tile_out + row * stride + lane * 12));
         __builtin_nontemporal_store(u32x3{w[k2 + 1][1], w[k2][1], w[k2 + 1][0]}, reinterpret_cast<u32x3 __attribute__((aligned(4))) *>(tile_out + row * stride + 768 + lane * 12));
      }
      return;
   }
   if (M == 9) {   // row-pair units: 4 units of 2 rows x 1536 B = 3 stores of 64 x 16 B (the middle one half in row A, half in row B);
      // wave 0 takes units 0 and 3, waves 1 and 2 one unit each.  No LDS.
      const uint32_t o0 = lane * 16, o1 = lane < 32 ? 1024 + lane * 16 : (uint32_t)stride + (lane - 32) * 16, o2 = (uint32_t)stride + 512 + lane * 16;
      int k2 = 0;
      for (uint32_t u = wave; u < 4; u += 3, k2 += 3) {
         uint8_t *ub = tile_out + 2 * u * stride;
         __builtin_nontemporal_store(u32x4{w[k2][0], w[k2][1], w[k2 + 1][0], w[k2 + 1][1]}, reinterpret_cast<u32x4 *>(ub + o0));
         __builtin_nontemporal_store(u32x4{w[k2 + 1][0], w[k2 + 1][1], w[k2 + 2][0], w[k2 + 2][1]}, reinterpret_cast<u32x4 *>(ub + o1));
         __builtin_nontemporal_store(u32x4{w[k2 + 2][0], w[k2 + 2][1], w[k2][0], w[k2][1]}, reinterpret_cast<u32x4 *>(ub + o2));
      }
      return;
   }
   if (M == 12) {   // M9 balanced: every wave one whole unit + one of the three stores of unit 3 (only a data-free benchmark can do that)
      const uint32_t o[3] = { lane * 16, lane < 32 ? 1024 + lane * 16 : (uint32_t)stride + (lane - 32) * 16, (uint32_t)stride + 512 + lane * 16 };
      uint8_t *ub = tile_out + 2 * wave * stride;
      __builtin_nontemporal_store(u32x4{w[0][0], w[0][1], w[1][0], w[1][1]}, reinterpret_cast<u32x4 *>(ub + o[0]));
      __builtin_nontemporal_store(u32x4{w[1][0], w[1][1], w[2][0], w[2][1]}, reinterpret_cast<u32x4 *>(ub + o[1]));
      __builtin_nontemporal_store(u32x4{w[2][0], w[2][1], w[0][0], w[0][1]}, reinterpret_cast<u32x4 *>(ub + o[2]));
      ub = tile_out + 6 * stride;
      __builtin_nontemporal_store(u32x4{w[3][0], w[3][1], w[4][0], w[4][1]}, reinterpret_cast<u32x4 *>(ub + (wave == 0 ? o[0] : wave == 1 ? o[1] : o[2])));
      return;
   }
   if (M == 13) {   // M6's geometry (rows of 2048 B per tile, no store spans two rows) with M9's 2/1/1 assignment: unit = 3 runs
      uint8_t *t2 = out + (wg / 6) * (6 * (size_t)12288) + (wg % 6) * 2048;
      int k2 = 0;
      for (uint32_t u = wave; u < 4; u += 3)
         for (uint32_t r3 = 0; r3 < 3; r3++, k2++) {
            const uint32_t run = u * 3 + r3, row = run >> 1, half = run & 1;
            __builtin_nontemporal_store(u32x4{w[k2][0], w[k2][1], w[k2 + 1][0], w[k2 + 1][1]}, reinterpret_cast<u32x4 *>(t2 + row * 12288 + half * 1024 + lane * 16));
         }
      return;
   }
   if (M == 14) {   // M9 with the units dealt 1/2/1 ... no: M9 with the heavy wave rotating from workgroup to workgroup
      const uint32_t o0 = lane * 16, o1 = lane < 32 ? 1024 + lane * 16 : (uint32_t)stride + (lane - 32) * 16, o2 = (uint32_t)stride + 512 + lane * 16;
      const uint32_t wv = (wave + (uint32_t)wg) % 3;
      int k2 = 0;
      for (uint32_t u = wv; u < 4; u += 3, k2 += 3) {
         uint8_t *ub = tile_out + 2 * u * stride;
         __builtin_nontemporal_store(u32x4{w[k2][0], w[k2][1], w[k2 + 1][0], w[k2 + 1][1]}, reinterpret_cast<u32x4 *>(ub + o0));
         __builtin_nontemporal_store(u32x4{w[k2 + 1][0], w[k2 + 1][1], w[k2 + 2][0], w[k2 + 2][1]}, reinterpret_cast<u32x4 *>(ub + o1));
         __builtin_nontemporal_store(u32x4{w[k2 + 2][0], w[k2 + 2][1], w[k2][0], w[k2][1]}, reinterpret_cast<u32x4 *>(ub + o2));
      }
      return;
   }
   if (M == 15) {   // rows dealt 3/3/2 (wave w: rows w, w+3, w+6); a wave's first two rows share the middle store (tail of the first,
      // head of the second), a third row takes one full and one half store: 13 stores per tile, 5/5/3
      const uint32_t r1 = wave, r2 = wave + 3, r3 = wave + 6;
      uint8_t *a = tile_out + r1 * stride, *b = tile_out + r2 * stride;
      __builtin_nontemporal_store(u32x4{w[0][0], w[0][1], w[1][0], w[1][1]}, reinterpret_cast<u32x4 *>(a + lane * 16));
      __builtin_nontemporal_store(u32x4{w[1][0], w[1][1], w[2][0], w[2][1]}, reinterpret_cast<u32x4 *>(lane < 32 ? a + 1024 + lane * 16 : b + (lane - 32) * 16));
      __builtin_nontemporal_store(u32x4{w[2][0], w[2][1], w[0][0], w[0][1]}, reinterpret_cast<u32x4 *>(b + 512 + lane * 16));
      if (r3 < 8) {
         uint8_t *c = tile_out + r3 * stride;
         __builtin_nontemporal_store(u32x4{w[3][0], w[3][1], w[4][0], w[4][1]}, reinterpret_cast<u32x4 *>(c + lane * 16));
         if (lane < 32) __builtin_nontemporal_store(u32x4{w[4][0], w[4][1], w[5][0], w[5][1]}, reinterpret_cast<u32x4 *>(c + 1024 + lane * 16));
      }
      return;
   }
   if (M == 16) {   // every row on its own: one full and one half store per row, rows dealt 3/3/2: 16 stores per tile
      for (uint32_t r = wave; r < 8; r += 3) {
         uint8_t *c = tile_out + r * stride;
         __builtin_nontemporal_store(u32x4{w[r][0], w[r][1], w[0][0], w[0][1]}, reinterpret_cast<u32x4 *>(c + lane * 16));
         if (lane < 32) __builtin_nontemporal_store(u32x4{w[1][0], w[1][1], w[r][0], w[r][1]}, reinterpret_cast<u32x4 *>(c + 1024 + lane * 16));
      }
      return;
   }
   if (M == 10 || M == 11) {   // the kernel's skeleton with the wave-private restage: planes interleaved per row pair in LDS
      // (unit u = [Y rows 2u,2u+1 | Cb | Cr] = 3072 contiguous bytes), barrier, then per unit: every lane reads 3 x 16 bytes
      // (16 pixels of each plane), writes 48 bytes of "RGB" back over the unit at 48 * lane, reads 16-byte pieces at
      // 1024 s + 16 lane and stores them.  M11: the same with four runs of 12-byte stores per unit (store shape only).
      uint8_t *pl = lds + wave * 1024 + lane * 8;
#pragma unroll
      for (int r = 0; r < 8; r++) *reinterpret_cast<uint2 *>(pl + (r >> 1) * 3072 + (r & 1) * 512) = make_uint2(w[r][0], w[r][1]);
      __syncthreads();
      const uint32_t o0 = lane * 16, o1 = lane < 32 ? 1024 + lane * 16 : (uint32_t)stride + (lane - 32) * 16, o2 = (uint32_t)stride + 512 + lane * 16;
      for (uint32_t u = wave; u < 4; u += 3) {
         uint8_t *ul = lds + u * 3072;
         const u32x4 a = *reinterpret_cast<const u32x4 *>(ul + lane * 16);
         const u32x4 b = *reinterpret_cast<const u32x4 *>(ul + 1024 + lane * 16);
         const u32x4 c = *reinterpret_cast<const u32x4 *>(ul + 2048 + lane * 16);
         uint8_t *ub = tile_out + 2 * u * stride;
         if (M == 11) {
            const uint32_t half = lane >> 5, q = (lane & 31) * 48;
            uint8_t *d = ub + half * stride + q;
            __builtin_nontemporal_store(u32x3{a.x, b.x, c.x}, reinterpret_cast<u32x3 __attribute__((aligned(4))) *>(d));
            __builtin_nontemporal_store(u32x3{a.y, b.y, c.y}, reinterpret_cast<u32x3 __attribute__((aligned(4))) *>(d + 12));
            __builtin_nontemporal_store(u32x3{a.z, b.z, c.z}, reinterpret_cast<u32x3 __attribute__((aligned(4))) *>(d + 24));
            __builtin_nontemporal_store(u32x3{a.w, b.w, c.w}, reinterpret_cast<u32x3 __attribute__((aligned(4))) *>(d + 36));
            continue;
         }
         // all three reads of every lane are complete before any write of this wave (data dependence through registers)
         *reinterpret_cast<u32x4 *>(ul + lane * 48) = u32x4{a.x, b.x, c.x, a.y};
         *reinterpret_cast<u32x4 *>(ul + lane * 48 + 16) = u32x4{b.y, c.y, a.z, b.z};
         *reinterpret_cast<u32x4 *>(ul + lane * 48 + 32) = u32x4{c.z, a.w, b.w, c.w};
         __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
         __builtin_amdgcn_wave_barrier();
         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
         const u32x4 s0 = *reinterpret_cast<const u32x4 *>(ul + lane * 16);
         const u32x4 s1 = *reinterpret_cast<const u32x4 *>(ul + 1024 + lane * 16);
         const u32x4 s2 = *reinterpret_cast<const u32x4 *>(ul + 2048 + lane * 16);
         __builtin_nontemporal_store(s0, reinterpret_cast<u32x4 *>(ub + o0));
         __builtin_nontemporal_store(s1, reinterpret_cast<u32x4 *>(ub + o1));
         __builtin_nontemporal_store(s2, reinterpret_cast<u32x4 *>(ub + o2));
      }
      return;
   }
   if (M == 1) {
      // 16 runs of 64 quads (row = run / 2, half = run % 2), wave w takes runs w, w+3, ...
      int k2 = 0;
      for (uint32_t run = wave; run < 16; run += 3, k2++) {
         const uint32_t row = run >> 1, half = run & 1;
         __builtin_nontemporal_store(u32x3{w[k2][0], w[k2][1], w[k2 + 1][0]},
                                     reinterpret_cast<u32x3 __attribute__((aligned(4))) *>(tile_out + row * stride + half * 768 + lane * 12));
      }
      return;
   }
   // LDS: plane c = wave, 8 rows of 512 bytes; block `lane` owns columns [8 lane, 8 lane + 8)
   uint8_t *pl = lds + wave * 4096 + lane * 8;
#pragma unroll
   for (int r = 0; r < 8; r++) *reinterpret_cast<uint2 *>(pl + r * 512) = make_uint2(w[r][0], w[r][1]);
   if (M == 3) __syncthreads();
   for (uint32_t run = wave; run < 16; run += 3) {
      const uint32_t row = run >> 1, half = run & 1;
      const uint32_t off = row * 512 + half * 256 + lane * 4;
      // M2 (no barrier): only this wave's own plane is certain to be written -> read it three times
      const uint32_t a = *reinterpret_cast<const uint32_t *>(lds + (M == 3 ? 0 : wave * 4096) + off);
      const uint32_t b = *reinterpret_cast<const uint32_t *>(lds + (M == 3 ? 4096 : wave * 4096) + (off ^ (M == 3 ? 0 : 4)));
      const uint32_t c = *reinterpret_cast<const uint32_t *>(lds + (M == 3 ? 8192 : wave * 4096) + (off ^ (M == 3 ? 0 : 8)));
      __builtin_nontemporal_store(u32x3{a, b, c}, reinterpret_cast<u32x3 __attribute__((aligned(4))) *>(tile_out + row * stride + half * 768 + lane * 12));
   }
}
template <int M> int run(const char *name, const u32x4 *d, uint8_t *o, unsigned wgs, unsigned tpr) {
   hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
   for (int w = 0; w < 60; w++) hipLaunchKernelGGL((k<M>), dim3(wgs), dim3(192), 0, 0, d, o, tpr);
   CHECK(hipDeviceSynchronize());
   const int REP = 20;
   CHECK(hipEventRecord(a));
   for (int r = 0; r < REP; r++) hipLaunchKernelGGL((k<M>), dim3(wgs), dim3(192), 0, 0, d, o, tpr);
   CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
   float ms; CHECK(hipEventElapsedTime(&ms, a, b));
   double us = ms * 1e3 / REP, bytes = (double)wgs * 192 * (128 + 64);
   printf("%-70s %8.1f us  %6.0f GB/s  %.1f %%\n", name, us, bytes / (us * 1e-6) / 1e9, bytes / (us * 1e-6) / 8e12 * 100);
   return 0;
}
int main() {
   const unsigned wgs = 16 * 4096;                       // c2 at 16 frames: 4096 tiles per frame
   const size_t in_bytes = (size_t)wgs * 192 * 128, out_bytes = (size_t)wgs * 192 * 64;
   u32x4 *d; uint8_t *o; CHECK(hipMalloc((void **)&d, in_bytes)); CHECK(hipMalloc((void **)&o, out_bytes)); CHECK(hipMemset(d, 1, in_bytes));
   for (int rep = 0; rep < 2; rep++) {
      run<0>("M0 records in, contiguous 16-byte stores", d, o, wgs, 8);
      run<1>("M1 records in, 12-byte stores to 8 rows 12288 B apart", d, o, wgs, 8);
      run<3>("M3 M1 + LDS round trip + workgroup barrier (the kernel's skeleton)", d, o, wgs, 8);
      run<4>("M4 16-byte stores from 48 lanes per run, same strided rows", d, o, wgs, 8);
      run<5>("M5 12-byte stores, the tile's 12288 bytes contiguous", d, o, wgs, 8);
      run<6>("M6 16-byte stores from 64 lanes, rows of 2048 B per tile, 12288 B apart", d, o, wgs / 6 * 6, 6);
      run<7>("M7 M1 without the non-temporal hint", d, o, wgs, 8);
      run<8>("M8 M1, every wave writes whole rows (both halves)", d, o, wgs, 8);
      run<9>("M9 16-byte stores, row-pair units (3 x 1024 B per unit), waves 2/1/1", d, o, wgs, 8);
      run<10>("M10 skeleton with wave-private restage in LDS (M9's stores)", d, o, wgs, 8);
      run<12>("M12 M9 balanced 4/4/4 (unit 3's stores one per wave)", d, o, wgs, 8);
      run<13>("M13 M6's rows of 2048 B with M9's 2/1/1 units", d, o, wgs / 6 * 6, 6);
      run<14>("M14 M9, heavy wave rotating with the workgroup", d, o, wgs, 8);
      run<15>("M15 rows dealt 3/3/2, pairs share the middle store: 13 stores, 5/5/3", d, o, wgs, 8);
      run<16>("M16 rows dealt 3/3/2, one full + one half store per row: 16 stores", d, o, wgs, 8);
   }
   return 0;
}

// ubench_valu.hip -- VALU issue-rate microbenchmarks on gfx950: what does one wave64 instruction
// cost on this chip (per SIMD), for the instruction kinds the back end is made of?
//   hipcc -O3 --offload-arch=gfx950 -o ubench_valu ubench_valu.hip && ./ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITER = 2048, UNROLL = 16;

#define BODY16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

template <int KIND>
__global__ void __launch_bounds__(256) k(float *out, float seed) {
   float r[16]; float2 p[16];
   typedef float f2 __attribute__((ext_vector_type(2)));
   f2 q[16];
   int ii[16];
   double dd[16];
   for (int i = 0; i < 16; i++) { r[i] = seed + i + threadIdx.x; q[i] = f2{r[i], r[i] + 1}; ii[i] = (int)r[i]; dd[i] = r[i]; }
   float c = seed * 0.5f; f2 c2 = f2{c, c};
   long long t0 = __builtin_amdgcn_s_memtime();
   for (int it = 0; it < ITER; it++) {
      if (KIND == 0) {
#define OP(n) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[n]) : "v"(c));
         BODY16(OP)
#undef OP
      } else if (KIND == 1) {
#define OP(n) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(q[n]) : "v"(c2));
         BODY16(OP)
#undef OP
      } else if (KIND == 2) {
#define OP(n) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[n]) : "v"(c));
         BODY16(OP)
#undef OP
      } else if (KIND == 3) {
#define OP(n) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(q[n]) : "v"(c2));
         BODY16(OP)
#undef OP
      } else if (KIND == 4) {
#define OP(n) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(ii[n]) : "v"(ii[(n + 1) & 15]));
         BODY16(OP)
#undef OP
      } else if (KIND == 5) {
#define OP(n) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(ii[n]) : "v"(ii[(n + 1) & 15]));
         BODY16(OP)
#undef OP
      } else if (KIND == 6) {
#define OP(n) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(r[n]) : "v"(ii[n]));
         BODY16(OP)
#undef OP
      } else if (KIND == 7) {
#define OP(n) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(r[n]) : "v"(c));
         BODY16(OP)
#undef OP
      } else if (KIND == 8) {
#define OP(n) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(dd[n]));
         BODY16(OP)
#undef OP
      } else if (KIND == 9) {
#define OP(n) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(ii[n]) : "v"(ii[(n + 1) & 15]));
         BODY16(OP)
#undef OP
      } else if (KIND == 10) {
#define OP(n) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(ii[n]) : "v"(r[n]));
         BODY16(OP)
#undef OP
      } else if (KIND == 11) {
#define OP(n) asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(ii[n]) : "v"(ii[(n + 1) & 15]));
         BODY16(OP)
#undef OP
      } else if (KIND == 12) {
#define OP(n) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(ii[n]) : "v"(ii[(n + 1) & 15]));
         BODY16(OP)
#undef OP
      } else if (KIND == 13) {
#define OP(n) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(ii[n]) : "v"(ii[(n + 1) & 15]));
         BODY16(OP)
#undef OP
      } else if (KIND == 14) {
#define OP(n) asm volatile("v_sat_pk_u8_i16 %0, %1" : "=v"(ii[n]) : "v"(ii[(n + 1) & 15]));
         BODY16(OP)
#undef OP
      } else if (KIND == 15) {
#define OP(n) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(q[n]) : "v"(c2));
         BODY16(OP)
#undef OP
      } else if (KIND == 16) {
#define OP(n) asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r[n]) : "v"(ii[n]));
         BODY16(OP)
#undef OP
      } else if (KIND == 17) {
#define OP(n) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(ii[n]) : "v"(r[n]));
         BODY16(OP)
#undef OP
      } else if (KIND == 18) {
#define OP(n) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[n]) : "v"(c));
         BODY16(OP)
#undef OP
      } else if (KIND == 19) {
#define OP(n) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[n]) : "v"(c));
         BODY16(OP)
#undef OP
      } else if (KIND == 20) {
#define OP(n) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(r[n]) : "v"(c));
         BODY16(OP)
#undef OP
      } else if (KIND == 21) {
#define OP(n) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ii[n]) : "v"(ii[(n + 1) & 15]));
         BODY16(OP)
#undef OP
      } else if (KIND == 22) {
#define OP(n) asm volatile("v_floor_f32 %0, %1" : "=v"(r[n]) : "v"(r[(n + 1) & 15]));
         BODY16(OP)
#undef OP
      } else if (KIND == 23) {
#define OP(n) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(ii[n]) : "v"(ii[(n + 1) & 15]));
         BODY16(OP)
#undef OP
      } else if (KIND == 24) {
#define OP(n) asm volatile("v_add_f32 %0, %1, %2" : "=v"(r[n]) : "v"(r[(n + 1) & 15]), "v"(r[(n + 2) & 15]));
         BODY16(OP)
#undef OP
      } else if (KIND == 25) {
#define OP(n) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(q[n]) : "v"(q[(n + 1) & 15]), "v"(q[(n + 2) & 15]));
         BODY16(OP)
#undef OP
      } else if (KIND == 26) {
#define OP(n) asm volatile("v_med3_i32 %0, %0, %1, %1" : "+v"(ii[n]) : "v"(ii[(n + 1) & 15]));
         BODY16(OP)
#undef OP
      } else if (KIND == 27) {
#define OP(n) asm volatile("v_mov_b32 %0, %1" : "=v"(ii[n]) : "v"(ii[(n + 1) & 15]));
         BODY16(OP)
#undef OP
      } else if (KIND == 28) {   // mix: pk_mul then add, alternating
#define OP(n) asm volatile("v_pk_mul_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3" : "+v"(q[n]), "+v"(r[n]) : "v"(c2), "v"(c));
         BODY16(OP)
#undef OP
      }
   }
   long long t1 = __builtin_amdgcn_s_memtime();
   if (threadIdx.x == 0 && blockIdx.x == 0) ((long long *)out)[256 * 8 * 256 / 2 - 1] = t1 - t0;
   float acc = 0;
   for (int i = 0; i < 16; i++) acc += r[i] + q[i].x + q[i].y + ii[i] + (float)dd[i];
   out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ void probe_cvt(const float *in, unsigned *out, int n) {
   int i = threadIdx.x;
   if (i < n) { unsigned r = 0; asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(r) : "v"(in[i])); out[i] = r; }
}

template <int KIND> int run(const char *name, float *d_out, int ops_per_instr) {
   int cus = 256, blocks = cus * 8;   // 8 blocks x 4 waves = 32 waves per CU = 8 per SIMD
   hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
   hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f);
   CHECK(hipDeviceSynchronize());
   CHECK(hipEventRecord(a));
   hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f);
   CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
   float ms; CHECK(hipEventElapsedTime(&ms, a, b));
   double winstr_per_simd = (double)blocks * 4 / (cus * 4) * ITER * UNROLL;   // wave-instructions each SIMD issued
   double ns_per = ms * 1e6 / winstr_per_simd;
   long long cyc = 0; CHECK(hipMemcpy(&cyc, (char *)d_out + (256 * 8 * 256 / 2 - 1) * 8, 8, hipMemcpyDeviceToHost));
   printf("%-22s %8.3f ms   %.3f ns per wave64-instr per SIMD; block0 wave0 memtime %lld ticks -> %.2f ticks per instr-slot (8 waves/SIMD)\n", name, ms, ns_per, cyc, (double)cyc / (ITER * UNROLL * 8.0));
   (void)ops_per_instr;
   return 0;
}

int main() {
   float *d_out; CHECK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float)));
   run<0>("v_add_f32", d_out, 1);
   run<1>("v_pk_add_f32", d_out, 2);
   run<2>("v_fma_f32", d_out, 1);
   run<15>("v_pk_fma_f32", d_out, 2);
   run<3>("v_pk_mul_f32", d_out, 2);
   run<4>("v_mul_lo_u32", d_out, 1);
   run<5>("v_mul_i32_i24", d_out, 1);
   run<11>("v_mad_i32_i24", d_out, 1);
   run<9>("v_pk_mul_lo_u16", d_out, 2);
   run<6>("v_cvt_f32_i32", d_out, 1);
   run<16>("v_cvt_f32_i32_sdwa", d_out, 1);
   run<17>("v_cvt_i32_f32", d_out, 1);
   run<7>("v_med3_f32", d_out, 1);
   run<8>("v_mul_f64", d_out, 1);
   run<10>("v_cvt_pk_u8_f32", d_out, 1);
   run<12>("v_perm_b32", d_out, 1);
   run<13>("v_pk_add_i16", d_out, 2);
   run<14>("v_sat_pk_u8_i16", d_out, 2);
   run<18>("v_mul_f32", d_out, 1);
   run<19>("v_sub_f32", d_out, 1);
   run<20>("v_fmac_f32", d_out, 1);
   run<21>("v_add_u32", d_out, 1);
   run<22>("v_floor_f32", d_out, 1);
   run<23>("v_lshl_or_b32", d_out, 1);
   run<24>("v_add_f32 3op", d_out, 1);
   run<25>("v_pk_mov_b32", d_out, 1);
   run<26>("v_med3_i32", d_out, 1);
   run<27>("v_mov_b32", d_out, 1);
   run<28>("pk_mul+add (2 instr)", d_out, 1);
   // semantics probe: v_cvt_pk_u8_f32 rounding / saturation
   {
      float h[] = { -300.f, -1.5f, -0.5f, -0.0f, 0.3f, 0.5f, 0.7f, 0.99999f, 1.0f, 1.5f, 2.5f, 3.5f, 127.99999f, 254.5f, 254.99998f, 255.0f, 255.5f, 256.0f, 300.f, 1e9f };
      int n = sizeof h / sizeof h[0];
      float *din; unsigned *dout; unsigned ho[32];
      CHECK(hipMalloc(&din, sizeof h)); CHECK(hipMalloc(&dout, 32 * 4));
      CHECK(hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(probe_cvt, dim3(1), dim3(64), 0, 0, din, dout, n);
      CHECK(hipMemcpy(ho, dout, n * 4, hipMemcpyDeviceToHost));
      for (int i = 0; i < n; i++) printf("cvt_pk_u8_f32(%g) = %u\n", h[i], ho[i]);
   }
   return 0;
}

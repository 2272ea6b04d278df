// probe: does v_cvt_pk_u8_f32 follow MODE.fp_round?  (round toward zero wanted: src/idct_opt.c:113 truncates)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(const float *in, unsigned *out, int n) {
   int i = threadIdx.x;
   if (i >= n) return;
   unsigned r = 0, r2 = 0; float a = in[i], s;
   asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                "v_cvt_pk_u8_f32 %0, %2, 0, %0\n\t"
                "v_add_f32 %1, %2, %2\n\t"
                "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\t" : "+v"(r), "=v"(s) : "v"(a));
   asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(r2) : "v"(a));
   out[2 * i] = r; out[2 * i + 1] = r2;
}
int main() {
   float h[] = { -300.f, -1.5f, -0.5f, -0.0f, 0.3f, 0.5f, 0.7f, 0.99999f, 1.0f, 1.5f, 2.5f, 3.5f, 127.99999f, 254.5f, 254.99998f, 255.0f, 255.5f, 256.0f, 300.f, 1e9f, 1.9999999f, 3.9999998f };
   int n = sizeof h / sizeof h[0];
   float *din; unsigned *dout; unsigned ho[64];
   hipMalloc(&din, sizeof h); hipMalloc(&dout, 64 * 4);
   hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
   hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, din, dout, n);
   hipMemcpy(ho, dout, 2 * n * 4, hipMemcpyDeviceToHost);
   for (int i = 0; i < n; i++) printf("x=%-12g rtz-mode=%3u  default=%3u  trunc-clamp=%d\n", h[i], ho[2 * i], ho[2 * i + 1], (int)(h[i] < 0 ? 0 : h[i] > 255 ? 255 : h[i]));
   return 0;
}

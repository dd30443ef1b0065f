// Micro-benchmark: VALU issue rate of v_fma_f32 / v_pk_fma_f32 / v_min3_f32 on gfx950, as a function of waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float float2_ __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void k(float* out, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float2_ p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  const float c = 0.999f, d = 1e-3f;
  const float2_ c2 = {c, c}, d2 = {d, d};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (OP == 0) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
      } else if (OP == 1) {
        asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                     "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2), "v"(d2));
      } else if (OP == 2) {
        asm volatile("v_min3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n"
                     "v_min3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
      } else if (OP == 3) {
        asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                     "v_pk_add_f32 %4, %4, %9\n v_pk_add_f32 %5, %5, %9\n v_pk_add_f32 %6, %6, %9\n v_pk_add_f32 %7, %7, %9\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2), "v"(d2));
      } else {
        asm volatile("v_sub_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                     "v_sub_f32 %4, %4, %9\n v_mul_f32 %5, %5, %8\n v_sub_f32 %6, %6, %9\n v_mul_f32 %7, %7, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template <int OP>
void run(const char* name, int waves_per_simd) {
  const int threads = 256;                       // 4 waves = 1 per SIMD
  const int blocks = 256 * waves_per_simd;       // blocks per CU = waves per SIMD
  const int iters = 20000;
  float* out;
  hipMalloc(&out, sizeof(float) * threads * blocks);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, threads>>>(out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<blocks, threads>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double winstr = (double)blocks * 4 * iters * 64.0;     // wave-instructions
  const double per_simd_per_ns = winstr / 1024.0 / (ms * 1e6);
  printf("%-14s waves/SIMD=%d  %.3f ms  %.3f wave-instr/ns/SIMD  (= %.2f cycles per wave-instr at 2.4 GHz)\n", name,
         waves_per_simd, ms, per_simd_per_ns, 2.4 / per_simd_per_ns);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma_f32", w);
    run<1>("v_pk_fma_f32", w);
    run<2>("v_min3_f32", w);
    run<3>("v_pk_mul/add", w);
    run<4>("v_sub/v_mul", w);
  }
  return 0;
}

// Micro-benchmark: sustained v_fma_f32 rate on gfx950 with the multiplier operand in a VGPR vs in an SGPR, in the shape
// of the Chamfer filter sweep (acc = q * ref + acc; 8 independent accumulators, 3 query registers, 4 reference values).
// Runs ~100+ ms per variant so that the clock settles.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, const float* refs, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float q0 = 0.999f + threadIdx.x * 1e-9f, q1 = 0.998f, q2 = 0.997f;
  float r0 = refs[0], r1 = refs[1], r2 = refs[2], r3 = refs[3];     // uniform: SGPRs unless forced into VGPRs
  if (OP == 0) asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
  else asm volatile("" : "+s"(r0), "+s"(r1), "+s"(r2), "+s"(r3));
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (OP == 0) {
        asm volatile("v_fmac_f32 %0, %8, %11\n v_fmac_f32 %1, %9, %12\n v_fmac_f32 %2, %10, %13\n v_fmac_f32 %3, %8, %14\n"
                     "v_fmac_f32 %4, %9, %11\n v_fmac_f32 %5, %10, %12\n v_fmac_f32 %6, %8, %13\n v_fmac_f32 %7, %9, %14\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(q0), "v"(q1), "v"(q2), "v"(r0), "v"(r1), "v"(r2), "v"(r3));
      } else {
        asm volatile("v_fmac_f32 %0, %11, %8\n v_fmac_f32 %1, %12, %9\n v_fmac_f32 %2, %13, %10\n v_fmac_f32 %3, %14, %8\n"
                     "v_fmac_f32 %4, %11, %9\n v_fmac_f32 %5, %12, %10\n v_fmac_f32 %6, %13, %8\n v_fmac_f32 %7, %14, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(q0), "v"(q1), "v"(q2), "s"(r0), "s"(r1), "s"(r2), "s"(r3));
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int OP>
void run(const char* name, int waves_per_simd, float* refs) {
  const int threads = 256, blocks = 256 * waves_per_simd, iters = 400000;
  float* out;
  hipMalloc(&out, sizeof(float) * threads * blocks);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, threads>>>(out, refs, iters / 4);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<blocks, threads>>>(out, refs, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double winstr = (double)blocks * 4 * iters * 64.0;
  printf("%-22s waves/SIMD=%d  %8.2f ms  %.3f wave-instr/ns/SIMD (%.2f clk per instr at 2.4 GHz)\n", name, waves_per_simd, ms,
         winstr / 1024.0 / (ms * 1e6), 2.4 / (winstr / 1024.0 / (ms * 1e6)));
  hipFree(out);
}

int main() {
  float h[4] = {0.5f, 0.25f, 0.125f, 1e-3f}, *refs;
  hipMalloc(&refs, 16);
  hipMemcpy(refs, h, 16, hipMemcpyHostToDevice);
  for (int w : {4, 8}) {
    run<0>("fmac acc,q(v),ref(v)", w, refs);
    run<1>("fmac acc,ref(s),q(v)", w, refs);
  }
  return 0;
}

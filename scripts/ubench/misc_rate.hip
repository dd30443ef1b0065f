// Micro-benchmark: issue rate of the bookkeeping instructions around the sweeps on gfx950 (sustained, 4 waves per SIMD):
// v_cndmask with the mask in VCC vs in an SGPR pair, v_cmp to VCC vs to an SGPR pair, v_med3, v_min, v_mov of a literal.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  float b0 = 0.5f + a0, b1 = 0.25f + a0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (OP == 0) asm volatile(REP8("v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e32 %1, %1, %5, vcc\n v_cndmask_b32_e32 %2, %2, %4, vcc\n v_cndmask_b32_e32 %3, %3, %5, vcc\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc");
      else if (OP == 1) asm volatile(REP8("v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cndmask_b32_e64 %1, %1, %5, s[10:11]\n v_cndmask_b32_e64 %2, %2, %4, s[10:11]\n v_cndmask_b32_e64 %3, %3, %5, s[10:11]\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "s10", "s11");
      else if (OP == 2) asm volatile(REP8("v_cmp_lt_f32_e32 vcc, %0, %4\n v_cmp_lt_f32_e32 vcc, %1, %5\n v_cmp_lt_f32_e32 vcc, %2, %4\n v_cmp_lt_f32_e32 vcc, %3, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc");
      else if (OP == 3) asm volatile(REP8("v_cmp_lt_f32_e64 s[10:11], %0, %4\n v_cmp_lt_f32_e64 s[12:13], %1, %5\n v_cmp_lt_f32_e64 s[14:15], %2, %4\n v_cmp_lt_f32_e64 s[16:17], %3, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17");
      else if (OP == 4) asm volatile(REP8("v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %4, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
      else if (OP == 5) asm volatile(REP8("v_min_f32_e32 %0, %0, %4\n v_min_f32_e32 %1, %1, %5\n v_min_f32_e32 %2, %2, %4\n v_min_f32_e32 %3, %3, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
      else if (OP == 6) asm volatile(REP8("v_mov_b32_e32 %0, 0x7f800000\n v_mov_b32_e32 %1, 0x7f800000\n v_mov_b32_e32 %2, 0x7f800000\n v_mov_b32_e32 %3, 0x7f800000\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
      else if (OP == 7) asm volatile(REP8("v_mov_b32_e32 %0, %4\n v_mov_b32_e32 %1, %5\n v_mov_b32_e32 %2, %4\n v_mov_b32_e32 %3, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
      else if (OP == 8) asm volatile(REP8("v_min_u32_e32 %0, %0, %4\n v_min_u32_e32 %1, %1, %5\n v_min_u32_e32 %2, %2, %4\n v_min_u32_e32 %3, %3, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
      else if (OP == 9) asm volatile(REP8("v_sub_f32_e32 %0, %4, %0\n v_sub_f32_e32 %1, %5, %1\n v_mul_f32_e32 %2, %2, %4\n v_mul_f32_e32 %3, %3, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
      else if (OP == 10) asm volatile(REP8("v_fma_f32 %0, %4, %4, %0\n v_fma_f32 %1, %5, %5, %1\n v_fma_f32 %2, %4, %4, %2\n v_fma_f32 %3, %5, %5, %3\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
      else if (OP == 11) asm volatile(REP8("v_min3_f32 %0, %0, %4, %5\n v_min3_f32 %1, %1, %4, %5\n v_min3_f32 %2, %2, %4, %5\n v_min3_f32 %3, %3, %4, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
      else if (OP == 12) asm volatile(REP8("v_add_u32_e32 %0, %0, %4\n v_and_b32_e32 %1, %1, %5\n v_xor_b32_e32 %2, %2, %4\n v_lshlrev_b32_e32 %3, 4, %3\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
      else if (OP == 13) asm volatile(REP8("v_max_f32_e32 %0, %0, %4\n v_min_f32_e32 %1, %1, %5\n v_max_f32_e32 %2, %2, %4\n v_min_f32_e32 %3, %3, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

template <int OP>
void run(const char* name) {
  const int threads = 256, blocks = 256 * 4, iters = 40000;
  float* out;
  hipMalloc(&out, sizeof(float) * threads * blocks);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, threads>>>(out, iters / 4);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<blocks, threads>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double winstr = (double)blocks * 4 * iters * 8 * 8 * 4;
  printf("%-34s %8.2f ms  %.3f wave-instr/ns/SIMD (%.2f clk per instr at 2.4 GHz)\n", name, ms, winstr / 1024.0 / (ms * 1e6),
         2.4 / (winstr / 1024.0 / (ms * 1e6)));
  hipFree(out);
}

int main() {
  run<0>("v_cndmask_b32_e32 (vcc)");
  run<1>("v_cndmask_b32_e64 (sgpr pair)");
  run<2>("v_cmp_lt_f32_e32 (-> vcc)");
  run<3>("v_cmp_lt_f32_e64 (-> sgpr pair)");
  run<4>("v_med3_f32");
  run<5>("v_min_f32_e32");
  run<6>("v_mov_b32 literal");
  run<7>("v_mov_b32 vgpr");
  run<8>("v_min_u32_e32");
  run<9>("v_sub_f32 / v_mul_f32");
  run<10>("v_fma_f32 (a*a+c)");
  run<11>("v_min3_f32");
  run<12>("v_add_u32/and/xor/lshl");
  run<13>("v_max_f32/v_min_f32");
  return 0;
}

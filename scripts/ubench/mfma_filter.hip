// Feasibility probe: the Chamfer filter e = |r|^2 - 2 q.r as a K=4 fp32 MFMA (2 x v_mfma_f32_32x32x2_f32 per 32x32 tile of
// (reference, query) pairs) followed by the in-lane min over the accumulator (8 v_min3 per tile): how many clocks per MFMA
// does a wave sustain when the min3s are interleaved, at 1 / 2 / 4 waves per SIMD?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int C>
__global__ __launch_bounds__(256) void k(float* out, const float* refs, int tiles) {
  __shared__ float s_ref[2048 * 4];
  for (int i = threadIdx.x; i < 2048 * 4; i += 256) s_ref[i] = refs[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, rl = lane & 31, half = lane >> 5;
  float b0[C], b1[C], tm[C];
#pragma unroll
  for (int c = 0; c < C; ++c) { b0[c] = 0.001f * (lane + c); b1[c] = half ? 1.0f : 0.002f * (lane + c); tm[c] = 1e30f; }
  for (int t = 0; t < tiles; ++t) {
    const float a0 = s_ref[((t & 63) * 32 + rl) * 4 + half], a1 = s_ref[((t & 63) * 32 + rl) * 4 + 2 + half];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[c], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[c], acc, 0, 0, 0);
      float m = tm[c];
#pragma unroll
      for (int r = 0; r < 16; r += 2) m = __builtin_fminf(__builtin_fminf(m, acc[r]), acc[r + 1]);
      tm[c] = m;
    }
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) s += tm[c];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int C>
void run(int waves_per_simd, float* refs) {
  const int blocks = 256 * waves_per_simd, tiles = 64 * 40;
  float* out;
  hipMalloc(&out, sizeof(float) * 256 * blocks);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<C><<<blocks, 256>>>(out, refs, tiles / 4);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<C><<<blocks, 256>>>(out, refs, tiles);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)blocks * 4 * tiles * C * 2;            // wave-level MFMA instructions
  const double per_simd_clk = ms * 1e-3 * 2.4e9 / (mfma / 1024.0);
  printf("C=%d waves/SIMD=%d: %.2f ms, %.1f clk per MFMA per SIMD at 2.4 GHz (64 = matrix pipe full), %.2f T pairs/s\n", C,
         waves_per_simd, ms, per_simd_clk, mfma / 2 * 1024 / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  float* refs;
  hipMalloc(&refs, 2048 * 16);
  hipMemset(refs, 0, 2048 * 16);
  for (int w : {1, 2, 4}) { run<4>(w, refs); run<8>(w, refs); }
  return 0;
}

// kabsch.hip -- batched rigid alignment: 3x3 (weighted) covariance reduction + register-resident
// one-sided Jacobi SVD.  Replaces SVDHead.forward (registration/model_utils.py:220-255), whose
// per-sample Python loop of torch.svd calls (:232-240) becomes one workgroup per sample.
//
// HBM-bound: each sample reads src+corr (+w) once from HBM (2 x 3 x N x 4 B); the centred second
// pass re-reads the same lines from L2.  Reads are coalesced along N (the reference's [B,3,N]
// channel-major layout is already SoA).
#include "houv_common.h"

namespace houv {
namespace {

constexpr int kKBlock = 256;

template <int NV>
__device__ __forceinline__ void block_reduce(float (&v)[NV], float* s_red /* [kKBlock/64][NV] */, float* s_out /*[NV]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
  __syncthreads();   // s_red / s_out free
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) s_red[wave * NV + i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    float a = 0.f;
    for (int w = 0; w < kKBlock / 64; ++w) a += s_red[w * NV + threadIdx.x];
    s_out[threadIdx.x] = a;
  }
  __syncthreads();
}

__global__ __launch_bounds__(kKBlock) void kabsch_kernel(const float* __restrict__ src, const float* __restrict__ corr,
                                                         const float* __restrict__ w, int N, float* __restrict__ R,
                                                         float* __restrict__ t) {
  __shared__ float s_red[(kKBlock / 64) * 12];
  __shared__ float s_out[12];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* s0 = src + (size_t)b * 3 * N;
  const float* c0 = corr + (size_t)b * 3 * N;
  const float* wb = w ? w + (size_t)b * N : nullptr;

  // pass 1: unweighted sums (model_utils.py:221-222) and, when weighted, the weighted sums used for t (:254)
  float a[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) a[i] = 0.f;
  for (int i = tid; i < N; i += kKBlock) {
    const float sx = s0[i], sy = s0[N + i], sz = s0[2 * N + i];
    const float cx = c0[i], cy = c0[N + i], cz = c0[2 * N + i];
    a[0] += sx; a[1] += sy; a[2] += sz; a[3] += cx; a[4] += cy; a[5] += cz;
    if (wb) {
      const float ww = wb[i];
      a[6] += ww * sx; a[7] += ww * sy; a[8] += ww * sz; a[9] += ww * cx; a[10] += ww * cy; a[11] += ww * cz;
    }
  }
  block_reduce<12>(a, s_red, s_out);
  const float inv_n = 1.0f / (float)N;
  const float ms[3] = {s_out[0] * inv_n, s_out[1] * inv_n, s_out[2] * inv_n};
  const float mc[3] = {s_out[3] * inv_n, s_out[4] * inv_n, s_out[5] * inv_n};
  const float ws[3] = {s_out[6], s_out[7], s_out[8]};
  const float wc[3] = {s_out[9], s_out[10], s_out[11]};

  // pass 2: H = (src_c * w) corr_c^T  (:224-227)
  float h[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) h[i] = 0.f;
  for (int i = tid; i < N; i += kKBlock) {
    const float ww = wb ? wb[i] : 1.0f;
    const float sx = (s0[i] - ms[0]) * ww, sy = (s0[N + i] - ms[1]) * ww, sz = (s0[2 * N + i] - ms[2]) * ww;
    const float cx = c0[i] - mc[0], cy = c0[N + i] - mc[1], cz = c0[2 * N + i] - mc[2];
    h[0] += sx * cx; h[1] += sx * cy; h[2] += sx * cz;
    h[3] += sy * cx; h[4] += sy * cy; h[5] += sy * cz;
    h[6] += sz * cx; h[7] += sz * cy; h[8] += sz * cz;
  }
  block_reduce<9>(h, s_red, s_out);
  if (tid == 0) {
    float H[9], Rm[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) H[i] = s_out[i];
    kabsch_rotation<float>(H, Rm);
    // t = -R mean(src) + mean(corr)  (:252), or with the weighted sums (:254)
    const float* ps = wb ? ws : ms;
    const float* pc = wb ? wc : mc;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      t[b * 3 + i] = -(Rm[i * 3 + 0] * ps[0] + Rm[i * 3 + 1] * ps[1] + Rm[i * 3 + 2] * ps[2]) + pc[i];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) R[b * 9 + i] = Rm[i];
  }
}

}  // namespace
}  // namespace houv

extern "C" int houv_kabsch(const float* src, const float* corr, const float* w_or_null, int B, int N, float* R,
                           float* t, void* stream) {
  using namespace houv;
  if (B < 0 || N <= 0 || !src || !corr || !R || !t) {
    set_error("houv_kabsch: bad argument B=%d N=%d", B, N);
    return 0;
  }
  if (B == 0) return 1;
  kabsch_kernel<<<B, kKBlock, 0, (hipStream_t)stream>>>(src, corr, w_or_null, N, R, t);
  return check_launch("houv_kabsch") ? 1 : 0;
}

// kabsch.hip -- batched rigid alignment: 3x3 (weighted) covariance reduction + register-resident
// one-sided Jacobi SVD.  Replaces SVDHead.forward (registration/model_utils.py:220-255), whose
// per-sample Python loop of torch.svd calls (:232-240) goes away:
//
// HBM-bound: each sample is read exactly once (2 x 3 x N x 4 B, + N x 4 B of weights).  Reads are coalesced along N (the reference's [B,3,N]
// channel-major layout is already SoA).
#include "houv_common.h"

namespace houv {
namespace {

constexpr int kKBlock = 256;          // 4 waves = 4 samples per workgroup
constexpr int kSamplesPerBlock = kKBlock / 64;

// One WAVE per sample: the two reductions are pure wave shuffles (no LDS, no workgroup barriers), so a CU keeps
// up to 32 samples in flight instead of 8 and the serial 3x3 SVD of one sample never stalls the loads of another.
__global__ __launch_bounds__(kKBlock) void kabsch_kernel(const float* __restrict__ src, const float* __restrict__ corr,
                                                         const float* __restrict__ w, int B, int N,
                                                         float* __restrict__ R, float* __restrict__ t) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * kSamplesPerBlock + (threadIdx.x >> 6);
  if (b >= B) return;   // wave-uniform
  const float* s0 = src + (size_t)b * 3 * N;
  const float* c0 = corr + (size_t)b * 3 * N;
  const float* wb = w ? w + (size_t)b * N : nullptr;

  // ONE pass over the sample (a second, centred pass would re-read it from HBM: with 32 samples in flight per CU the
  // 49 KB samples do not survive in the 4 MB L2).  Moments are taken about the sample's first point (a, b) -- any point
  // of the cloud is within one diameter of the mean, so the shifted sums lose at most a few ulp to cancellation:
  //   S1 = sum (s-a), C1 = sum (c-b), W0 = sum w, WS = sum w (s-a), WC = sum w (c-b), P = sum w (s-a)(c-b)^T
  //   H  = sum w (s-ms)(c-mc)^T = P - WS mc'^T - ms' WC^T + W0 ms' mc'^T        (ms' = S1/N, mc' = C1/N)
  // which is (src_c * w) corr_c^T with the UNWEIGHTED means of model_utils.py:221-227.
  const float ax = s0[0], ay = s0[N], az = s0[2 * N];
  const float bx = c0[0], by = c0[N], bz = c0[2 * N];
  float m[7], P[9];   // S1[3], C1[3], W0 ; P
  float ws_[3] = {0.f, 0.f, 0.f}, wc_[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 7; ++i) m[i] = 0.f;
#pragma unroll
  for (int i = 0; i < 9; ++i) P[i] = 0.f;
  auto acc_point = [&](float sx_, float sy_, float sz_, float cx_, float cy_, float cz_, float ww) {
    const float sx = sx_ - ax, sy = sy_ - ay, sz = sz_ - az;
    const float cx = cx_ - bx, cy = cy_ - by, cz = cz_ - bz;
    m[0] += sx; m[1] += sy; m[2] += sz; m[3] += cx; m[4] += cy; m[5] += cz;
    const float wx = sx * ww, wy = sy * ww, wz = sz * ww;
    if (wb) {
      m[6] += ww;
      ws_[0] += wx; ws_[1] += wy; ws_[2] += wz;
      wc_[0] += cx * ww; wc_[1] += cy * ww; wc_[2] += cz * ww;
    }
    P[0] += wx * cx; P[1] += wx * cy; P[2] += wx * cz;
    P[3] += wy * cx; P[4] += wy * cy; P[5] += wy * cz;
    P[6] += wz * cx; P[7] += wz * cy; P[8] += wz * cz;
  };
  const bool vec = ((N & 3) == 0) && ((reinterpret_cast<uintptr_t>(s0) | reinterpret_cast<uintptr_t>(c0) |
                                       reinterpret_cast<uintptr_t>(wb)) & 15) == 0;
  if (vec) {   // 16-byte loads: 4 points per lane per load
    const int n4 = N >> 2;
    const float4* s4 = reinterpret_cast<const float4*>(s0);
    const float4* c4 = reinterpret_cast<const float4*>(c0);
    const float4* w4 = reinterpret_cast<const float4*>(wb);
#pragma unroll 2
    for (int i = lane; i < n4; i += 64) {
      const float4 X = s4[i], Y = s4[n4 + i], Z = s4[2 * n4 + i];
      const float4 U = c4[i], V = c4[n4 + i], W = c4[2 * n4 + i];
      const float4 q = wb ? w4[i] : make_float4(1.f, 1.f, 1.f, 1.f);
      acc_point(X.x, Y.x, Z.x, U.x, V.x, W.x, q.x);
      acc_point(X.y, Y.y, Z.y, U.y, V.y, W.y, q.y);
      acc_point(X.z, Y.z, Z.z, U.z, V.z, W.z, q.z);
      acc_point(X.w, Y.w, Z.w, U.w, V.w, W.w, q.w);
    }
  } else {
    for (int i = lane; i < N; i += 64)
      acc_point(s0[i], s0[N + i], s0[2 * N + i], c0[i], c0[N + i], c0[2 * N + i], wb ? wb[i] : 1.0f);
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) m[i] = wave_sum(m[i]);     // butterfly: every lane holds the totals
#pragma unroll
  for (int i = 0; i < 9; ++i) P[i] = wave_sum(P[i]);
  if (wb) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { ws_[i] = wave_sum(ws_[i]); wc_[i] = wave_sum(wc_[i]); }
  } else {
    m[6] = (float)N;
#pragma unroll
    for (int i = 0; i < 3; ++i) { ws_[i] = m[i]; wc_[i] = m[3 + i]; }
  }
  const float inv_n = 1.0f / (float)N;
  const float msp[3] = {m[0] * inv_n, m[1] * inv_n, m[2] * inv_n};     // means of the shifted clouds
  const float mcp[3] = {m[3] * inv_n, m[4] * inv_n, m[5] * inv_n};
  float h[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) h[i * 3 + j] = P[i * 3 + j] - ws_[i] * mcp[j] - msp[i] * wc_[j] + m[6] * msp[i] * mcp[j];
  const float sh_a[3] = {ax, ay, az}, sh_b[3] = {bx, by, bz};
  const float ms[3] = {ax + msp[0], ay + msp[1], az + msp[2]};
  const float mc[3] = {bx + mcp[0], by + mcp[1], bz + mcp[2]};
  float ws[3], wc[3];   // sum w s, sum w c
#pragma unroll
  for (int i = 0; i < 3; ++i) { ws[i] = ws_[i] + m[6] * sh_a[i]; wc[i] = wc_[i] + m[6] * sh_b[i]; }
  if (lane == 0) {
    float Rm[9];
    kabsch_rotation<float>(h, Rm);
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
  if (B < 0 || N <= 0) {
    set_error("houv_kabsch: bad argument B=%d N=%d", B, N);
    return 0;
  }
  if (B == 0) return 1;
  if (!src || !corr || !R || !t) {
    set_error("houv_kabsch: null pointer");
    return 0;
  }
  kabsch_kernel<<<(B + kSamplesPerBlock - 1) / kSamplesPerBlock, kKBlock, 0, (hipStream_t)stream>>>(src, corr, w_or_null, B, N, R, t);
  return check_launch("houv_kabsch") ? 1 : 0;
}

// attention.hip -- fused multi-head attention for the DCP pointer network (registration/models/dcp.py:26-32, :198-229)
// in fp32 on the MFMA pipe of gfx950:   O = softmax(Q K^T / sqrt(d_k)) V   per (pair, head), d_k = 128.
//
// The un-fused form (houv_gemm_f32 -> houv_softmax_rows -> houv_gemm_f32) writes and re-reads the [P,4,Nq,Nk] score
// tensor three times -- 67 MB per pair and attention at 2048 points, ~0.3 ms of the 1.85 ms a pair takes.  Here the
// scores never leave the chip (online softmax over 32-key blocks).
//
// Mapping (v_mfma_f32_32x32x2_f32; C/D layout: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)):
//   * a workgroup = 4 waves = 128 queries of one (pair, head); a wave owns 32 queries, whose Q rows live in registers for
//     the whole kernel (64 floats per lane: lane (q, half) holds dims half*64 .. half*64+63);
//   * S^T = K_blk Q^T  (keys x queries): A operand = K from LDS (dim-major tile), B operand = the Q registers.  The MFMA's
//     summation index may be permuted freely as long as A and B agree: step s pairs dim half*64+s of both operands, which
//     is what makes the contiguous Q registers usable;
//   * in the S^T accumulator a lane holds 16 keys of ONE query (the other half-wave holds the other 16): the row maximum
//     and sum are in-lane reductions plus one exchange with lane ^ 32;
//   * O^T += V^T P^T  (dims x queries): B operand = the P registers exactly as they sit in the accumulator (step r pairs
//     key (r & 3) + 8 (r >> 2) + 4 half, again a permuted summation index), A operand = V rows from LDS.  No transposition
//     of P through LDS.
// fp32 MFMA products are exact k-ordered fmaf chains; the online softmax differs from torch's two-pass softmax only by
// fp32 rounding order (tests/test_gpu_dcp.py: 2e-5 against a float64 reference).
#include "../../include/houv_hip.h"
#include "houv_common.h"

namespace houv {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kDk = 128;       // head dimension (dcp.py: 512 / 4 heads)
constexpr int kBq = 128;       // queries per workgroup
constexpr int kBk = 32;        // keys per block
constexpr int kLdK = kBk + 1;  // dim-major K tile: Ks[d][key], +1 spreads the transposing writes; dims 64..127 sit 32 floats
                               // further (k_at) so that the two half-waves of an operand fetch use disjoint banks
constexpr int kLdV = kDk + 8;  // key-major V tile: Vs[key][d]; 4 rows = 32 banks apart, again for the two half-waves
__device__ __forceinline__ int k_at(int d, int key) { return d * kLdK + (d >> 6) * 32 + key; }

struct AttnArgs {
  const float* Q; const float* K; const float* V; float* O;
  int Nq, Nk, ldq, ldk, ldv, ldo;
  long long sQ, sK, sV, sO;
  float scale;
};

// FULL: Nq % 128 == 0 and Nk % 32 == 0 (the launcher checks): no row / key guards -- no exec-mask branches around the loads, no
// masking compares in the softmax.
template <bool FULL>
__global__ __launch_bounds__(256, 2) void attention_f32_kernel(AttnArgs g) {
  __shared__ float Ks[kDk * kLdK + 32];
  __shared__ __attribute__((aligned(16))) float Vs[kBk * kLdV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, half = lane >> 5;
  const int head = blockIdx.y, pair = blockIdx.z;
  const float* __restrict__ Q = g.Q + pair * g.sQ + head * kDk;
  const float* __restrict__ K = g.K + pair * g.sK + head * kDk;
  const float* __restrict__ V = g.V + pair * g.sV + head * kDk;
  float* __restrict__ O = g.O + pair * g.sO + head * kDk;
  const int q = blockIdx.x * kBq + wave * 32 + ql;

  // this lane's query row, dims half*64 .. +63 (zeros for rows beyond Nq: computed, never stored)
  float qf[64];
  {
    const float* qp = Q + (size_t)(FULL || q < g.Nq ? q : 0) * g.ldq + half * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float4 v = (FULL || q < g.Nq) ? *reinterpret_cast<const float4*>(qp + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
      qf[4 * i + 0] = v.x; qf[4 * i + 1] = v.y; qf[4 * i + 2] = v.z; qf[4 * i + 3] = v.w;
    }
  }
  f32x16 o[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // staging: 32 keys x 128 dims = 1024 float4 per tile, 4 per thread: key = tid/32 + 8 i, dims 4 (tid % 32) ..
  float4 rk[4], rv[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = k0 + (tid >> 5) + 8 * i, d = (tid & 31) * 4;
      const bool ok = FULL || key < g.Nk;
      rk[i] = ok ? *reinterpret_cast<const float4*>(K + (size_t)key * g.ldk + d) : make_float4(0.f, 0.f, 0.f, 0.f);
      rv[i] = ok ? *reinterpret_cast<const float4*>(V + (size_t)key * g.ldv + d) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = (tid >> 5) + 8 * i, d = (tid & 31) * 4;
      Ks[k_at(d + 0, key)] = rk[i].x; Ks[k_at(d + 1, key)] = rk[i].y;
      Ks[k_at(d + 2, key)] = rk[i].z; Ks[k_at(d + 3, key)] = rk[i].w;
      *reinterpret_cast<float4*>(&Vs[key * kLdV + d]) = rv[i];
    }
  };

  fetch(0);
  for (int k0 = 0; k0 < g.Nk; k0 += kBk) {
    __syncthreads();          // previous block fully consumed
    stage();
    __syncthreads();
    if (k0 + kBk < g.Nk) fetch(k0 + kBk);   // next block's global loads fly under the MFMAs

    // ---- S^T = K_blk Q^T: 64 steps, step s pairs dim half*64 + s ----
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const float* kp = Ks + k_at(half * 64, ql);
#pragma unroll
    for (int st = 0; st < 64; ++st) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[st * kLdK], qf[st], s, 0, 0, 0);

    // ---- online softmax: this lane holds keys k0 + (r&3) + 8 (r>>2) + 4 half of query ql ----
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      s[r] = (FULL || key < g.Nk) ? s[r] * g.scale : -INFINITY;
      mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, kWave));
    const float m_new = fmaxf(m_run, mx);                     // finite: every block holds at least one valid key
    const float alpha = __expf(m_run - m_new);                // exp(-inf) = 0 on the first block
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __expf(s[r] - m_new);
      rs += s[r];
    }
    rs += __shfl_xor(rs, 32, kWave);
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= alpha;

    // ---- O^T += V_blk^T P^T: 16 steps per dim tile, step r pairs key (r&3) + 8 (r>>2) + 4 half ----
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* vp = Vs + ((r & 3) + 8 * (r >> 2) + 4 * half) * kLdV + ql;
#pragma unroll
      for (int t = 0; t < 4; ++t) o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[t * 32], s[r], o[t], 0, 0, 0);
    }
  }

  // ---- O[q][dims] = O^T / l: a lane holds, per dim tile t and group gq = r >> 2, dims t*32 + 8 gq + 4 half .. +3 ----
  if (FULL || q < g.Nq) {
    const float inv = 1.0f / l_run;
    float* op = O + (size_t)q * g.ldo;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const float4 v = make_float4(o[t][4 * gq + 0] * inv, o[t][4 * gq + 1] * inv, o[t][4 * gq + 2] * inv, o[t][4 * gq + 3] * inv);
        *reinterpret_cast<float4*>(op + t * 32 + 8 * gq + 4 * half) = v;
      }
  }
}

}  // namespace
}  // namespace houv

extern "C" int houv_attention_f32(const float* Q, const float* K, const float* V, float* O, int P, int H, int Nq, int Nk,
                                  int dk, int ldq, int ldk, int ldv, int ldo, long long sQ, long long sK, long long sV,
                                  long long sO, float scale, void* stream) {
  using namespace houv;
  if (P < 0 || H <= 0 || Nq <= 0 || Nk <= 0 || dk != kDk || ldq < H * dk || ldk < H * dk || ldv < H * dk || ldo < H * dk ||
      (ldq & 3) || (ldk & 3) || (ldv & 3) || (ldo & 3) || (sQ & 3) || (sK & 3) || (sV & 3) || (sO & 3) || P > 65535 || H > 65535) {
    set_error("houv_attention_f32: bad argument P=%d H=%d Nq=%d Nk=%d dk=%d (dk must be 128; strides multiples of 4)", P, H,
              Nq, Nk, dk);
    return 0;
  }
  if (P == 0) return 1;
  if (!Q || !K || !V || !O || ((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(K) |
                                reinterpret_cast<uintptr_t>(V) | reinterpret_cast<uintptr_t>(O)) & 15)) {
    set_error("houv_attention_f32: null or unaligned pointer (16-byte alignment required)");
    return 0;
  }
  AttnArgs g{Q, K, V, O, Nq, Nk, ldq, ldk, ldv, ldo, sQ, sK, sV, sO, scale};
  dim3 grid((Nq + kBq - 1) / kBq, H, P);
  if (Nq % kBq == 0 && Nk % kBk == 0) attention_f32_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(g);
  else attention_f32_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(g);
  return check_launch("houv_attention_f32") ? 1 : 0;
}

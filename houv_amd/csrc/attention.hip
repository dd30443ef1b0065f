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
#include <atomic>
#include "../../include/houv_hip.h"
#include "houv_common.h"
#include "houv_split.h"

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


// ---------------------------------------------------------------------------------------------------------------
// The same attention on the bf16 matrix pipe (houv_split.h): Q, K, V and the probabilities are split into three bf16 parts and
// both products are summed from six part products in fp32 -- fp32-grade results (tests/test_gpu_dcp.py: the same 2e-5 against a
// float64 reference) at 6/16 of the fp32-input MFMA time.
//   * a pre-pass (attention_split_k_kernel, attention_split_v_kernel) splits K and V once per call (every query tile reads them): K planes [pair][head][part][key][128 dims], V planes
//     TRANSPOSED [pair][head][part][128 dims][key'], keys permuted inside every group of 16 (quads 1 and 2 swapped) so that an
//     operand fragment of V^T is 8 consecutive bf16 -- see (c);
//   * Q is split by the workgroup itself, once, into registers: lane (query, half) holds dims 16 s + 8 half .. +7 of every step s;
//   * (a) S^T = K_blk Q^T: v_mfma_f32_32x32x16_bf16, A = K fragments from LDS (one ds_read_b128 per part), B = the Q registers;
//   * (b) online softmax on the accumulator exactly as in the fp32 kernel;
//   * (c) O^T += V_blk^T P^T: the probabilities stay where the accumulator holds them (cdna_hip_programming.md, "an accumulator
//     tile as the next MFMA's operand"): registers 8 s .. 8 s + 7 of lane (query, half) are the B fragment of step s, i.e. slot
//     (half, j) stands for key 16 s + 8 (j >> 2) + 4 half + (j & 3) -- the permutation the pre-pass applied to V^T's keys.
// 128 queries per workgroup (4 waves), 32 keys per block, two workgroups per CU; full tiles only (Nq % 128 == 0, Nk % 32 == 0).
// ---------------------------------------------------------------------------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kKRowB = kDk * 2 + 16;     // bytes per K row in LDS: 128 bf16 + 16 B of padding (17 slots: conflict-free b128 reads)
constexpr int kVRowB = kBk * 2 + 16;     // bytes per V^T row in LDS: 32 bf16 + 16 B (5 slots)

struct AttnSplitArgs {
  const float* Q; float* O;
  const unsigned char* Kp; const unsigned char* Vp;      // the pre-pass's planes
  int Nq, Nk, ldq, ldo, H;
  long long sQ, sO;
  float scale;
};

// pre-pass, K: one thread = 8 consecutive dims of one key (16-byte loads and stores, both coalesced)
__global__ __launch_bounds__(256) void attention_split_k_kernel(const float* __restrict__ K, int Nk, int ldk, long long sK, int H,
                                                                unsigned char* __restrict__ Kp) {
  const int head = blockIdx.y, pair = blockIdx.z;
  const size_t plane = (size_t)Nk * kDk * 2;                                    // bytes of one part of one (pair, head)
  unsigned char* kp = Kp + ((size_t)pair * H + head) * 3 * plane;
  const int t = blockIdx.x * 256 + threadIdx.x;                                 // Nk * 16 threads
  if (t >= Nk * 16) return;
  const int key = t >> 4, oct = t & 15;
  const float* src = K + pair * sK + (size_t)key * ldk + head * kDk + oct * 8;
  const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  uint4 parts[3];
  split8<3>(v, parts);
#pragma unroll
  for (int p = 0; p < 3; ++p) *reinterpret_cast<uint4*>(kp + p * plane + ((size_t)key * kDk + oct * 8) * 2) = parts[p];
}

// pre-pass, V: a workgroup transposes 64 keys x 128 dims through LDS -- rows of V are read with coalesced float4 loads, and every
// (dim, 8-key slot group) fragment is written as 16 bytes with the 8 groups of a dim side by side (128-byte runs per dim and part)
__global__ __launch_bounds__(256) void attention_split_v_kernel(const float* __restrict__ V, int Nk, int ldv, long long sV, int H,
                                                                unsigned char* __restrict__ Vp) {
  constexpr int kKeys = 64, kLd = kKeys + 1;
  __shared__ float T[kDk * kLd];                                                // T[dim][key]
  const int head = blockIdx.y, pair = blockIdx.z, k0 = blockIdx.x * kKeys, tid = threadIdx.x;
  const size_t plane = (size_t)Nk * kDk * 2;
  unsigned char* vp = Vp + ((size_t)pair * H + head) * 3 * plane;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = tid + 256 * i, key = idx >> 5, d = (idx & 31) * 4;
    const float4 v = (k0 + key < Nk) ? *reinterpret_cast<const float4*>(V + pair * sV + (size_t)(k0 + key) * ldv + head * kDk + d)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    T[(d + 0) * kLd + key] = v.x; T[(d + 1) * kLd + key] = v.y; T[(d + 2) * kLd + key] = v.z; T[(d + 3) * kLd + key] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int oct = tid & 7, dim = (tid >> 3) + 32 * i;
    const int base = (oct >> 1) * 16 + 4 * (oct & 1);                           // slot (half = oct & 1, j) <- key base + 8 (j >> 2) + (j & 3)
    if (k0 + (oct >> 1) * 16 >= Nk) continue;                                   // Nk is a multiple of 32: whole 16-key groups
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = T[dim * kLd + base + 8 * (j >> 2) + (j & 3)];
    uint4 parts[3];
    split8<3>(v, parts);
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<uint4*>(vp + p * plane + ((size_t)dim * Nk + k0 + oct * 8) * 2) = parts[p];
  }
}

__global__ __launch_bounds__(256, 2) void attention_split_kernel(AttnSplitArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned char Ks[3][kBk * kKRowB];
  __shared__ __attribute__((aligned(16))) unsigned char Vs[3][kDk * kVRowB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, half = lane >> 5;
  const int head = blockIdx.y, pair = blockIdx.z;
  const float* __restrict__ Q = g.Q + pair * g.sQ + head * kDk;
  float* __restrict__ O = g.O + pair * g.sO + head * kDk;
  const size_t plane = (size_t)g.Nk * kDk * 2;
  const unsigned char* __restrict__ Kp = g.Kp + ((size_t)pair * g.H + head) * 3 * plane;
  const unsigned char* __restrict__ Vp = g.Vp + ((size_t)pair * g.H + head) * 3 * plane;
  const int q = blockIdx.x * kBq + wave * 32 + ql;

  // Q fragments: step s, part p -> 8 bf16 (dims 16 s + 8 half .. +7 of this lane's query)
  uint4 qf[8][3];
  {
    const float* qp = Q + (size_t)q * g.ldq + half * 8;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float4 a = *reinterpret_cast<const float4*>(qp + 16 * s), b = *reinterpret_cast<const float4*>(qp + 16 * s + 4);
      const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      split8<3>(v, qf[s]);
    }
  }
  f32x16 o[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // staging: per part 32 keys x 256 B of K and 128 dims x 64 B of V^T = 512 + 512 16-byte pieces; 2 + 2 per thread and part.
  // K and V travel separately (24 registers each instead of 48 together: with both in flight the kernel spilled, and the reloads'
  // s_waitcnt vmcnt drained the prefetch): V(k0) flies under S^T(k0) + softmax, K(k0+1) under P V(k0)
  u32x4 rk[6], rv[6];                     // plain vectors: HIP's uint4 struct copies become memcpys that keep the arrays in scratch
  // per-lane byte offsets are 32-bit and block-independent; the block / part offsets go to the (scalar) base pointers
  const unsigned koff0 = (unsigned)((tid >> 4) * kDk * 2 + (tid & 15) * 16), koff1 = koff0 + 16u * kDk * 2;   // key = c / 16, piece c % 16
  const unsigned voff0 = (unsigned)((tid >> 2) * g.Nk * 2 + (tid & 3) * 16), voff1 = voff0 + 64u * (unsigned)g.Nk * 2;   // dim = c / 4, piece c % 4
  auto fetch_k = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const unsigned char* base = Kp + p * plane + (size_t)k0 * kDk * 2;
      rk[p * 2 + 0] = *reinterpret_cast<const u32x4*>(base + koff0);
      rk[p * 2 + 1] = *reinterpret_cast<const u32x4*>(base + koff1);
    }
  };
  auto fetch_v = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const unsigned char* base = Vp + p * plane + (size_t)k0 * 2;
      rv[p * 2 + 0] = *reinterpret_cast<const u32x4*>(base + voff0);
      rv[p * 2 + 1] = *reinterpret_cast<const u32x4*>(base + voff1);
    }
  };
  auto stage_k = [&]() {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = tid + 256 * i;
        *reinterpret_cast<u32x4*>(&Ks[p][(c >> 4) * kKRowB + (c & 15) * 16]) = rk[p * 2 + i];
      }
  };
  auto stage_v = [&]() {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = tid + 256 * i;
        *reinterpret_cast<u32x4*>(&Vs[p][(c >> 2) * kVRowB + (c & 3) * 16]) = rv[p * 2 + i];
      }
  };
  auto frag = [](const uint4& u) { return __builtin_bit_cast(bf16x8, u); };
  // six part products, smallest first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi); written out -- indexing the part arrays
  // through a table makes hipcc promote them to LDS (24 KB, one workgroup per CU less)
  auto mma6 = [&](const uint4 (&a)[3], const uint4 (&b)[3], f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(a[2]), frag(b[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(a[0]), frag(b[2]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(a[1]), frag(b[1]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(a[1]), frag(b[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(a[0]), frag(b[1]), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(a[0]), frag(b[0]), c, 0, 0, 0);
  };

  fetch_k(0);
  for (int k0 = 0; k0 < g.Nk; k0 += kBk) {
    stage_k();                // Ks is free: every wave passed the barrier behind S^T of the previous block
    fetch_v(k0);
    __syncthreads();

    // ---- (a) S^T = K_blk Q^T: 8 steps of 16 dims ----
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      uint4 kf[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) kf[p] = *reinterpret_cast<const uint4*>(&Ks[p][ql * kKRowB + st * 32 + half * 16]);
      s = mma6(kf, qf[st], s);
    }

    if (k0 + kBk < g.Nk) fetch_k(k0 + kBk);   // the next K block flies under the softmax and P V

    // ---- (b) online softmax: this lane holds keys k0 + (r&3) + 8 (r>>2) + 4 half of query ql ----
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] *= g.scale;
      mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, kWave));
    const float m_new = fmaxf(m_run, mx);
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

    stage_v();                // Vs is free: every wave passed this block's first barrier, i.e. finished P V of the previous block
    __syncthreads();

    // ---- (c) O^T += V_blk^T P^T: 2 steps of 16 keys per 32-dim tile ----
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const float pv[8] = {s[8 * st + 0], s[8 * st + 1], s[8 * st + 2], s[8 * st + 3], s[8 * st + 4], s[8 * st + 5], s[8 * st + 6], s[8 * st + 7]};
      uint4 pf[3];
      split8<3>(pv, pf);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        uint4 vf[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) vf[p] = *reinterpret_cast<const uint4*>(&Vs[p][(t * 32 + ql) * kVRowB + st * 32 + half * 16]);
        o[t] = mma6(vf, pf, o[t]);
      }
    }
  }

  // ---- O[q][dims] = O^T / l: a lane holds, per dim tile t and group gq = r >> 2, dims t*32 + 8 gq + 4 half .. +3 ----
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
  dim3 grid((Nq + kBq - 1) / kBq, H, P);
  if (g_debug.attn_split.load() && Nq % kBq == 0 && Nk % kBk == 0) {
    // bf16 matrix pipe: K and V are split once into a stream-ordered workspace (2 x 3 planes of bf16), freed behind the kernel
    const size_t bytes = (size_t)P * H * 3 * Nk * kDk * 2;
    unsigned char* ws = nullptr;
    static std::atomic<unsigned> pool_kept{0};                 // per device, once: freed workspace stays in the stream-ordered pool
    int devid = 0;
    if (hipGetDevice(&devid) == hipSuccess && devid < 32 && !(pool_kept.fetch_or(1u << devid) & (1u << devid))) {
      hipMemPool_t pool;
      unsigned long long keep = ~0ull;
      if (hipDeviceGetDefaultMemPool(&pool, devid) == hipSuccess) (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    }
    if (hipMallocAsync(reinterpret_cast<void**>(&ws), 2 * bytes, (hipStream_t)stream) == hipSuccess && ws) {
      attention_split_k_kernel<<<dim3((Nk * 16 + 255) / 256, H, P), 256, 0, (hipStream_t)stream>>>(K, Nk, ldk, sK, H, ws);
      attention_split_v_kernel<<<dim3((Nk + 63) / 64, H, P), 256, 0, (hipStream_t)stream>>>(V, Nk, ldv, sV, H, ws + bytes);
      AttnSplitArgs a{Q, O, ws, ws + bytes, Nq, Nk, ldq, ldo, H, sQ, sO, scale};
      attention_split_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a);
      const bool ok = check_launch("houv_attention_f32");
      (void)hipFreeAsync(ws, (hipStream_t)stream);
      return ok ? 1 : 0;
    }
    (void)hipGetLastError();   // no workspace: the fp32-input kernel below needs none
  }
  AttnArgs g{Q, K, V, O, Nq, Nk, ldq, ldk, ldv, ldo, sQ, sK, sV, sO, scale};
  if (Nq % kBq == 0 && Nk % kBk == 0) attention_f32_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(g);
  else attention_f32_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(g);
  return check_launch("houv_attention_f32") ? 1 : 0;
}

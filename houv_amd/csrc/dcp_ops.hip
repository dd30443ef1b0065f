// dcp_ops.hip -- the non-GEMM kernels of the DCP feature head (registration/models/dcp.py), fp32, inference.
//   houv_knn            dcp.py:35-42   k nearest neighbours (self included), the brute-force sweep generalised to k > 1
//   houv_edgeconv1      dcp.py:44-66 + conv1/bn1/relu (:272,277,285-286): edge features (neighbour, centre) -> 64 channels
//   houv_max_over_k     dcp.py:287,290,293,296  x.max(dim=-1) over the k neighbours, written into the concat buffer
//   houv_layernorm      dcp.py:144-154 (a_2 (x-mean)/(std+eps) + b_2 with the UNBIASED std of torch.std) [+ residual]
//   houv_softmax_rows   dcp.py:31  softmax over the last dimension, in place
//   houv_softmax_corr   dcp.py:346-348  softmax(scores) and src_corr = tgt . scores^T fused: one pass over each row
// All HBM-bound except knn (VALU, like the Chamfer sweep).
#include "../../include/houv_hip.h"
#include "houv_common.h"

namespace houv {
namespace {

// ------------------------------------------------------------------------------------------------------------------
// k-NN: one lane per query, references broadcast from LDS, a sorted top-K list per lane in registers.
// Inserting into the list (a K-step compare / select bubble, ~5 K instructions) is what costs: a lane needs it ~K ln(N/K)
// times, but a wave runs it whenever ANY of its 64 lanes does -- for nearly every reference (measured: 94 % of the kernel).
// So a lane only APPENDS a qualifying reference (d below its K-th distance as of the last flush -- a stale, hence
// conservative, threshold) to a small queue in LDS, and the wave flushes the queues through the bubble when one of them
// is full: ~8x fewer bubbles.  Entries are inserted in scan order with the same strict <, and an entry that no longer
// qualifies falls through the bubble unchanged, so the lists are identical to inserting on the spot.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kKnnQueue = 8;

template <int K>
__global__ __launch_bounds__(256) void knn_kernel(const float* __restrict__ xyz, int N, int* __restrict__ idx) {
  __shared__ float4 s_ref[1024];
  __shared__ float2 s_q[kKnnQueue][256];   // [slot][lane]: (distance, index bits)
  const int b = blockIdx.y, tid = threadIdx.x;
  const int qi = blockIdx.x * 256 + tid;
  const float* __restrict__ p = xyz + (size_t)b * N * 3;
  const bool ok = qi < N;
  const float qx = ok ? p[qi * 3 + 0] : 0.f, qy = ok ? p[qi * 3 + 1] : 0.f, qz = ok ? p[qi * 3 + 2] : 0.f;
  float bd[K];
  int bi[K];
#pragma unroll
  for (int j = 0; j < K; ++j) { bd[j] = INFINITY; bi[j] = 0; }
  int qn = 0;                 // entries in this lane's queue
  float thr = INFINITY;       // bd[K-1] as of the last flush
  auto flush = [&]() {
#pragma unroll 1
    for (int s = 0; s < kKnnQueue; ++s) {
      if (!__any(s < qn)) break;
      const float2 e = s_q[s][tid];
      float d = s < qn ? e.x : INFINITY;
      int id = __float_as_int(e.y);
#pragma unroll
      for (int t = 0; t < K; ++t) {   // bubble the candidate through the sorted list (strict <: earlier index first)
        const bool lt = d < bd[t];
        const float td = bd[t]; const int ti = bi[t];
        bd[t] = lt ? d : td;  bi[t] = lt ? id : ti;
        d = lt ? td : d;      id = lt ? ti : id;
      }
    }
    qn = 0;
    thr = bd[K - 1];
  };
  for (int r0 = 0; r0 < N; r0 += 1024) {
    const int cnt = min(1024, N - r0);
    __syncthreads();
    for (int j = tid; j < cnt; j += 256) s_ref[j] = make_float4(p[(r0 + j) * 3], p[(r0 + j) * 3 + 1], p[(r0 + j) * 3 + 2], 0.f);
    __syncthreads();
    for (int j = 0; j < cnt; ++j) {
      const float4 r = s_ref[j];
      const float d = metric_sqdist<0>(r.x - qx, r.y - qy, r.z - qz);
      if (d < thr) {
        s_q[qn][tid] = make_float2(d, __int_as_float(r0 + j));
        ++qn;
      }
      if (__any(qn == kKnnQueue)) flush();
    }
  }
  flush();
  if (ok) {
#pragma unroll
    for (int j = 0; j < K; ++j) idx[((size_t)b * N + qi) * K + j] = bi[j];
  }
}

// The same search with the REFERENCES split four ways (round 3): the kernel above puts one wave on 64 queries x all N references --
// at 16 clouds x 2048 points that is 512 waves for 1024 SIMDs, each running for 380 us.  Here a workgroup of four waves owns 64
// queries; wave w scans the w-th quarter of the references (its own LDS stage, same queue / flush), the four sorted partial lists
// meet in LDS and wave 0 merges them (20 steps over four list heads).  A partial list is sorted by (distance, index) -- insertion in
// scan order with strict < --, the merge takes the smallest head and on ties the lower quarter, i.e. the lower index: the result
// is the (distance, index)-lexicographic top-K, exactly what the single scan produces.
template <int K>
__global__ __launch_bounds__(256) void knn_split_kernel(const float* __restrict__ xyz, int N, int* __restrict__ idx) {
  constexpr int kSlices = 4, kChunk = 512;
  __shared__ __attribute__((aligned(16))) unsigned char smem[kSlices * kChunk * 16 + kKnnQueue * 256 * 8];   // 32 KB stages + 16 KB queues
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float4* s_ref = reinterpret_cast<float4*>(smem) + wave * kChunk;
  float2 (*s_q)[256] = reinterpret_cast<float2 (*)[256]>(smem + kSlices * kChunk * 16);
  const int qi = blockIdx.x * 64 + lane;
  const float* __restrict__ p = xyz + (size_t)b * N * 3;
  const bool ok = qi < N;
  const float qx = ok ? p[qi * 3 + 0] : 0.f, qy = ok ? p[qi * 3 + 1] : 0.f, qz = ok ? p[qi * 3 + 2] : 0.f;
  float bd[K];
  int bi[K];
#pragma unroll
  for (int j = 0; j < K; ++j) { bd[j] = INFINITY; bi[j] = 0; }
  int qn = 0;
  float thr = INFINITY;
  auto flush = [&]() {
#pragma unroll 1
    for (int s = 0; s < kKnnQueue; ++s) {
      if (!__any(s < qn)) break;
      const float2 e = s_q[s][tid];
      float d = s < qn ? e.x : INFINITY;
      int id = __float_as_int(e.y);
#pragma unroll
      for (int t = 0; t < K; ++t) {
        const bool lt = d < bd[t];
        const float td = bd[t]; const int ti = bi[t];
        bd[t] = lt ? d : td;  bi[t] = lt ? id : ti;
        d = lt ? td : d;      id = lt ? ti : id;
      }
    }
    qn = 0;
    thr = bd[K - 1];
  };
  const int len = (N + kSlices - 1) / kSlices, r_begin = wave * len, r_end = min(N, r_begin + len);
  for (int r0 = r_begin; r0 < r_end; r0 += kChunk) {
    const int cnt = min(kChunk, r_end - r0);
    __builtin_amdgcn_wave_barrier();                       // this wave's stage is its own: LDS ops of a wave execute in order
    for (int j = lane; j < cnt; j += 64) s_ref[j] = make_float4(p[(r0 + j) * 3], p[(r0 + j) * 3 + 1], p[(r0 + j) * 3 + 2], 0.f);
    __builtin_amdgcn_wave_barrier();
    for (int j = 0; j < cnt; ++j) {
      const float4 r = s_ref[j];
      const float d = metric_sqdist<0>(r.x - qx, r.y - qy, r.z - qz);
      if (d < thr) {
        s_q[qn][tid] = make_float2(d, __int_as_float(r0 + j));
        ++qn;
      }
      if (__any(qn == kKnnQueue)) flush();
    }
  }
  flush();
  __syncthreads();                                         // every wave is done with its stage and queue: the lists take their place
  float2 (*L)[K][64] = reinterpret_cast<float2 (*)[K][64]>(smem);   // [slice][rank][query], 4 x K x 64 x 8 B <= 48 KB for K <= 24
  static_assert(kSlices * K * 64 * 8 <= (int)sizeof(smem), "the partial lists must fit the stage + queue area");
#pragma unroll
  for (int j = 0; j < K; ++j) L[wave][j][lane] = make_float2(bd[j], __int_as_float(bi[j]));
  __syncthreads();
  if (wave == 0 && ok) {
    float cd[kSlices];
    int ci[kSlices], h[kSlices];
#pragma unroll
    for (int s = 0; s < kSlices; ++s) { const float2 e = L[s][0][lane]; cd[s] = e.x; ci[s] = __float_as_int(e.y); h[s] = 0; }
#pragma unroll 1
    for (int j = 0; j < K; ++j) {
      int best = 0;
      float d = cd[0];
      int id = ci[0];
#pragma unroll
      for (int s = 1; s < kSlices; ++s) {                  // strict <: on equal distances the lower quarter (lower index) stays
        const bool lt = cd[s] < d;
        d = lt ? cd[s] : d; id = lt ? ci[s] : id; best = lt ? s : best;
      }
      idx[((size_t)b * N + qi) * K + j] = id;
      int hb = 0;
#pragma unroll
      for (int s = 0; s < kSlices; ++s) { h[s] += (s == best); hb = (s == best) ? h[s] : hb; }
      const float2 e = (hb < K) ? L[best][hb][lane] : make_float2(INFINITY, 0.f);
#pragma unroll
      for (int s = 0; s < kSlices; ++s) { cd[s] = (s == best) ? e.x : cd[s]; ci[s] = (s == best) ? __float_as_int(e.y) : ci[s]; }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// conv1 on the 6-channel edge feature (neighbour xyz, centre xyz) + folded BN + ReLU.  K = 6 is too thin for MFMA.
// One thread per (point, neighbour, 4 output channels).  out[(b*N+n)*k + j][64]
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void edgeconv1_kernel(const float* __restrict__ xyz, const int* __restrict__ idx,
                                                        size_t total_edges, int N, int k, const float* __restrict__ W /*[64,6]*/,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        float* __restrict__ out) {
  __shared__ float sW[64 * 6], sS[64], sH[64];
  for (int i = threadIdx.x; i < 64 * 6; i += 256) sW[i] = W[i];
  if (threadIdx.x < 64) { sS[threadIdx.x] = scale[threadIdx.x]; sH[threadIdx.x] = shift[threadIdx.x]; }
  __syncthreads();
  const size_t work = total_edges * 16;                 // 16 channel quads per edge
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < work; e += (size_t)gridDim.x * 256) {
    const size_t edge = e >> 4;
    const int cq = (int)(e & 15) * 4;
    const size_t pt = edge / k;                         // b*N + n
    const size_t b = pt / N;
    const int nb = idx[edge];
    const float* c = xyz + pt * 3;
    const float* q = xyz + (b * N + nb) * 3;
    const float f[6] = {q[0], q[1], q[2], c[0], c[1], c[2]};   // cat((feature, x), dim=3): neighbour first (dcp.py:64)
    float4 o;
    float* po = &o.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* w = sW + (cq + u) * 6;
      float a = 0.f;
#pragma unroll
      for (int t = 0; t < 6; ++t) a = __builtin_fmaf(f[t], w[t], a);
      po[u] = fmaxf(a * sS[cq + u] + sH[cq + u], 0.f);
    }
    *reinterpret_cast<float4*>(out + edge * 64 + cq) = o;
  }
}

// out[pt*ldo + c] = max_j act[(pt*k + j)*C + c]
__global__ __launch_bounds__(256) void max_over_k_kernel(const float* __restrict__ act, size_t npts, int k, int C,
                                                         float* __restrict__ out, int ldo) {
  const int c4 = C / 4;
  const size_t work = npts * c4;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < work; e += (size_t)gridDim.x * 256) {
    const size_t pt = e / c4;
    const int c = (int)(e - pt * c4) * 4;
    float4 m = *reinterpret_cast<const float4*>(act + (pt * k) * C + c);
    for (int j = 1; j < k; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(act + (pt * k + j) * C + c);
      m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
    *reinterpret_cast<float4*>(out + pt * ldo + c) = m;
  }
}

// One wave per row of D (multiple of 4, <= 2048) channels: a (x-mean)/(std_unbiased+eps) + b  [+ residual]
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, size_t rows, int D,
                                                        const float* __restrict__ a, const float* __restrict__ b2,
                                                        float eps, const float* __restrict__ residual,
                                                        float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * D;
  float s = 0.f;
  for (int i = lane * 4; i < D; i += 256) {
    const float4 v = *reinterpret_cast<const float4*>(xr + i);
    s += (v.x + v.y) + (v.z + v.w);
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int i = lane * 4; i < D; i += 256) {
    const float4 v = *reinterpret_cast<const float4*>(xr + i);
    const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
    q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
  }
  const float stdv = sqrtf(wave_sum(q) / (float)(D - 1));          // torch.std: Bessel-corrected (dcp.py:153)
  const float inv = 1.0f / (stdv + eps);
  for (int i = lane * 4; i < D; i += 256) {
    const float4 v = *reinterpret_cast<const float4*>(xr + i);
    const float4 aa = *reinterpret_cast<const float4*>(a + i);
    const float4 bb = *reinterpret_cast<const float4*>(b2 + i);
    float4 o = make_float4(aa.x * (v.x - mean) * inv + bb.x, aa.y * (v.y - mean) * inv + bb.y,
                           aa.z * (v.z - mean) * inv + bb.z, aa.w * (v.w - mean) * inv + bb.w);
    if (residual) {
      const float4 r = *reinterpret_cast<const float4*>(residual + row * D + i);
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    *reinterpret_cast<float4*>(out + row * D + i) = o;
  }
}

// One wave per row, in place: x = softmax(x) over L columns.  Rows of up to 4096 floats (multiple of 4, 16-byte aligned)
// are held in registers -- one 16-byte read and one 16-byte write per element; longer/odd rows take the 3-pass path.
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ x, size_t rows, int L) {
  const int lane = threadIdx.x & 63;
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* xr = x + row * L;
  if ((L & 3) == 0 && L <= 4096 && (reinterpret_cast<uintptr_t>(xr) & 15) == 0) {
    constexpr int MAXV = 16;                       // 16 float4 per lane x 64 lanes = 4096 floats
    const int n4 = L >> 2;
    float4 v[MAXV];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      v[i] = (c < n4) ? reinterpret_cast<const float4*>(xr)[c] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      m = fmaxf(fmaxf(fmaxf(m, v[i].x), fmaxf(v[i].y, v[i].z)), v[i].w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      v[i].x = expf(v[i].x - m); v[i].y = expf(v[i].y - m); v[i].z = expf(v[i].z - m); v[i].w = expf(v[i].w - m);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float inv = 1.0f / wave_sum(s);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < n4) reinterpret_cast<float4*>(xr)[c] = make_float4(v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv);
    }
    return;
  }
  float m = -INFINITY;
  for (int i = lane; i < L; i += 64) m = fmaxf(m, xr[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float s = 0.f;
  for (int i = lane; i < L; i += 64) {
    const float e = expf(xr[i] - m);
    xr[i] = e;
    s += e;
  }
  const float inv = 1.0f / wave_sum(s);
  for (int i = lane; i < L; i += 64) xr[i] *= inv;
}

// One wave per score row n of pair b: p = softmax(scores[b,n,:]); corr[b,c,n] = sum_m p[m] * pts[b,m,c]
__global__ __launch_bounds__(256) void softmax_corr_kernel(const float* __restrict__ scores, int P, int N, int M,
                                                           const float* __restrict__ pts /*[P,M,3]*/,
                                                           float* __restrict__ corr /*[P,3,N]*/) {
  const int lane = threadIdx.x & 63;
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (size_t)P * N) return;
  const size_t b = row / N;
  const int n = (int)(row - b * N);
  const float* sr = scores + row * M;
  const float* pb = pts + b * M * 3;
  float m = -INFINITY;
  for (int i = lane; i < M; i += 64) m = fmaxf(m, sr[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float s = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
  for (int i = lane; i < M; i += 64) {
    const float e = expf(sr[i] - m);
    s += e;
    cx += e * pb[i * 3 + 0]; cy += e * pb[i * 3 + 1]; cz += e * pb[i * 3 + 2];
  }
  s = wave_sum(s); cx = wave_sum(cx); cy = wave_sum(cy); cz = wave_sum(cz);
  if (lane == 0) {
    const float inv = 1.0f / s;
    corr[(b * 3 + 0) * N + n] = cx * inv;
    corr[(b * 3 + 1) * N + n] = cy * inv;
    corr[(b * 3 + 2) * N + n] = cz * inv;
  }
}

inline unsigned grid_for(size_t work, unsigned cap = 16384) {
  size_t g = (work + 255) / 256;
  return (unsigned)(g > cap ? cap : (g ? g : 1));
}

}  // namespace
}  // namespace houv

extern "C" int houv_knn(const float* xyz, int B, int N, int k, int32_t* idx, void* stream) {
  using namespace houv;
  if (B <= 0 || N <= 0 || !xyz || !idx || k > N) {
    set_error("houv_knn: bad argument B=%d N=%d k=%d", B, N, k);
    return 0;
  }
  dim3 grid((N + 255) / 256, B);
  hipStream_t s = (hipStream_t)stream;
  const bool split = N >= 512 && g_debug.knn_split.load() != 0;   // four waves per 64 queries, a quarter of the references each
  if (k == 20 && split) knn_split_kernel<20><<<dim3((N + 63) / 64, B), 256, 0, s>>>(xyz, N, idx);
  else if (k == 16 && split) knn_split_kernel<16><<<dim3((N + 63) / 64, B), 256, 0, s>>>(xyz, N, idx);
  else if (k == 20) knn_kernel<20><<<grid, 256, 0, s>>>(xyz, N, idx);
  else if (k == 16) knn_kernel<16><<<grid, 256, 0, s>>>(xyz, N, idx);
  else if (k == 8) knn_kernel<8><<<grid, 256, 0, s>>>(xyz, N, idx);
  else if (k == 3) knn_kernel<3><<<grid, 256, 0, s>>>(xyz, N, idx);
  else if (k == 1) knn_kernel<1><<<grid, 256, 0, s>>>(xyz, N, idx);
  else {
    set_error("houv_knn: k must be one of 1, 3, 8, 16, 20 (got %d)", k);
    return 0;
  }
  return check_launch("houv_knn") ? 1 : 0;
}

extern "C" int houv_edgeconv1(const float* xyz, const int32_t* idx, int B, int N, int k, const float* W,
                              const float* scale, const float* shift, float* out, void* stream) {
  using namespace houv;
  if (B <= 0 || N <= 0 || k <= 0 || !xyz || !idx || !W || !scale || !shift || !out) {
    set_error("houv_edgeconv1: bad argument");
    return 0;
  }
  const size_t edges = (size_t)B * N * k;
  edgeconv1_kernel<<<grid_for(edges * 16), 256, 0, (hipStream_t)stream>>>(xyz, idx, edges, N, k, W, scale, shift, out);
  return check_launch("houv_edgeconv1") ? 1 : 0;
}

extern "C" int houv_max_over_k(const float* act, long long npts, int k, int C, float* out, int ldo, void* stream) {
  using namespace houv;
  if (npts <= 0 || k <= 0 || C <= 0 || (C & 3) || (ldo & 3) || !act || !out) {
    set_error("houv_max_over_k: bad argument (C and ldo must be multiples of 4)");
    return 0;
  }
  max_over_k_kernel<<<grid_for((size_t)npts * (C / 4)), 256, 0, (hipStream_t)stream>>>(act, (size_t)npts, k, C, out, ldo);
  return check_launch("houv_max_over_k") ? 1 : 0;
}

extern "C" int houv_layernorm(const float* x, long long rows, int D, const float* a, const float* b, float eps,
                              const float* residual_or_null, float* out, void* stream) {
  using namespace houv;
  if (rows <= 0 || D < 4 || (D & 3) || !x || !a || !b || !out) {
    set_error("houv_layernorm: bad argument (D must be a multiple of 4)");
    return 0;
  }
  layernorm_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(x, (size_t)rows, D, a, b, eps,
                                                                              residual_or_null, out);
  return check_launch("houv_layernorm") ? 1 : 0;
}

extern "C" int houv_softmax_rows(float* x, long long rows, int L, void* stream) {
  using namespace houv;
  if (rows <= 0 || L <= 0 || !x) {
    set_error("houv_softmax_rows: bad argument");
    return 0;
  }
  softmax_rows_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(x, (size_t)rows, L);
  return check_launch("houv_softmax_rows") ? 1 : 0;
}

extern "C" int houv_softmax_corr(const float* scores, int P, int N, int M, const float* pts, float* corr, void* stream) {
  using namespace houv;
  if (P <= 0 || N <= 0 || M <= 0 || !scores || !pts || !corr) {
    set_error("houv_softmax_corr: bad argument");
    return 0;
  }
  softmax_corr_kernel<<<(unsigned)(((size_t)P * N + 3) / 4), 256, 0, (hipStream_t)stream>>>(scores, P, N, M, pts, corr);
  return check_launch("houv_softmax_corr") ? 1 : 0;
}

// pointops.hip -- the sibling brute-force point ops the reference pulls from utils/mm3d_pn2:
//   houv_furthest_point_sample  utils/mm3d_pn2/ops/furthest_point_sample/src/furthest_point_sample_cuda.cu:22-118
//   houv_knn_cross (three_nn)   utils/mm3d_pn2/ops/interpolate/src/three_nn_cuda.cu (k nearest of ANOTHER cloud, k = 3)
//   houv_gather_points          utils/mm3d_pn2/ops/gather_points/src/gather_points_cuda.cu
// Call site on the registration side: train_utils.combine (registration/train_utils.py:459-464; not used by `solve`).
// These CUDA extensions cannot run here and the reference holds no fixtures for them: parity unpinned; the tests
// check them against direct numpy/torch restatements of the (simple, deterministic) algorithms.
#include "../../include/houv_hip.h"
#include "houv_common.h"

namespace houv {
namespace {

// Furthest point sampling: one workgroup per sample; every lane keeps its points and their running minimum distance
// in registers (the reference round-trips a [B,N] temp array through global memory every step); each of the npoint
// steps is one distance update + a (distance, lowest index) arg-max: wave butterfly, then 16 values through LDS.
template <int BLOCK, int PPT>
__global__ __launch_bounds__(BLOCK) void fps_kernel(const float* __restrict__ xyz, int N, int npoint,
                                                    int* __restrict__ idx) {
  __shared__ unsigned long long s_key[BLOCK / 64];
  __shared__ float s_pt[3];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* __restrict__ p = xyz + (size_t)b * N * 3;
  float px[PPT], py[PPT], pz[PPT], md[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int i = k * BLOCK + tid;
    const bool ok = i < N;
    px[k] = ok ? p[i * 3 + 0] : 0.f; py[k] = ok ? p[i * 3 + 1] : 0.f; pz[k] = ok ? p[i * 3 + 2] : 0.f;
    md[k] = ok ? 1e10f : -1.f;                       // the reference's temp.fill_(1e10); padding can never win
  }
  int old = 0;                                       // the first sample is point 0 (furthest_point_sample_cuda.cu:43-44)
  if (tid == 0) idx[(size_t)b * npoint] = 0;
  for (int j = 1; j < npoint; ++j) {
    if (tid == 0) { s_pt[0] = p[old * 3 + 0]; s_pt[1] = p[old * 3 + 1]; s_pt[2] = p[old * 3 + 2]; }
    __syncthreads();
    const float x1 = s_pt[0], y1 = s_pt[1], z1 = s_pt[2];
    unsigned long long best = 0ull;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const float dx = px[k] - x1, dy = py[k] - y1, dz = pz[k] - z1;
      const float d = dx * dx + dy * dy + dz * dz;
      md[k] = fminf(md[k], d);
      if (md[k] >= 0.f) {
        // max over (distance, then LOWEST index): key = dist bits | ~index
        const unsigned long long key = ((unsigned long long)__float_as_uint(md[k]) << 32) | (unsigned)(0x7fffffff - (k * BLOCK + tid));
        best = key > best ? key : best;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(best, o, 64);
      best = other > best ? other : best;
    }
    if (lane == 0) s_key[wave] = best;
    __syncthreads();
    unsigned long long m = s_key[0];
#pragma unroll
    for (int w = 1; w < BLOCK / 64; ++w) m = s_key[w] > m ? s_key[w] : m;
    old = 0x7fffffff - (int)(unsigned)(m & 0xffffffffull);
    if (tid == 0) idx[(size_t)b * npoint + j] = old;
    __syncthreads();                                  // s_pt / s_key reused next step
  }
}

// k nearest points of ref[B,M,3] for every query[B,N,3] (k <= 8): sorted ascending, squared distances out.
template <int K>
__global__ __launch_bounds__(256) void knn_cross_kernel(const float* __restrict__ query, const float* __restrict__ ref,
                                                        int N, int M, float* __restrict__ dist2, int* __restrict__ idx) {
  __shared__ float4 s_ref[1024];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int qi = blockIdx.x * 256 + tid;
  const float* __restrict__ q = query + (size_t)b * N * 3;
  const float* __restrict__ r = ref + (size_t)b * M * 3;
  const bool ok = qi < N;
  const float qx = ok ? q[qi * 3 + 0] : 0.f, qy = ok ? q[qi * 3 + 1] : 0.f, qz = ok ? q[qi * 3 + 2] : 0.f;
  float bd[K];
  int bi[K];
#pragma unroll
  for (int j = 0; j < K; ++j) { bd[j] = INFINITY; bi[j] = 0; }
  for (int r0 = 0; r0 < M; r0 += 1024) {
    const int cnt = min(1024, M - r0);
    __syncthreads();
    for (int j = tid; j < cnt; j += 256) s_ref[j] = make_float4(r[(r0 + j) * 3], r[(r0 + j) * 3 + 1], r[(r0 + j) * 3 + 2], 0.f);
    __syncthreads();
    for (int j = 0; j < cnt; ++j) {
      const float4 v = s_ref[j];
      float d = metric_sqdist<0>(v.x - qx, v.y - qy, v.z - qz);
      if (__any(d < bd[K - 1])) {
        int id = r0 + j;
#pragma unroll
        for (int s = 0; s < K; ++s) {
          const bool lt = d < bd[s];
          const float td = bd[s]; const int ti = bi[s];
          bd[s] = lt ? d : td;  bi[s] = lt ? id : ti;
          d = lt ? td : d;      id = lt ? ti : id;
        }
      }
    }
  }
  if (ok) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      dist2[((size_t)b * N + qi) * K + j] = bd[j];
      idx[((size_t)b * N + qi) * K + j] = bi[j];
    }
  }
}

// The same with queued insertions, as in houv_knn (dcp_ops.hip): a wave runs the insertion bubble for all 64 lanes whenever one
// lane needs it, so a lane only queues its candidates and the queues are flushed in bulk.  Same lists, same order.  Pays for
// K = 8 (332 -> 232 us at 64 x 2048 x 1900); for K <= 3 the bubble is cheaper than the queue (181 -> 214 us), which keep the
// kernel above.
constexpr int kCrossQueue = 8;

template <int K>
__global__ __launch_bounds__(256) void knn_cross_queued_kernel(const float* __restrict__ query, const float* __restrict__ ref,
                                                               int N, int M, float* __restrict__ dist2, int* __restrict__ idx) {
  __shared__ float4 s_ref[1024];
  __shared__ float2 s_q[kCrossQueue][256];   // [slot][lane]: (distance, index bits)
  const int b = blockIdx.y, tid = threadIdx.x;
  const int qi = blockIdx.x * 256 + tid;
  const float* __restrict__ q = query + (size_t)b * N * 3;
  const float* __restrict__ r = ref + (size_t)b * M * 3;
  const bool ok = qi < N;
  const float qx = ok ? q[qi * 3 + 0] : 0.f, qy = ok ? q[qi * 3 + 1] : 0.f, qz = ok ? q[qi * 3 + 2] : 0.f;
  float bd[K];
  int bi[K];
#pragma unroll
  for (int j = 0; j < K; ++j) { bd[j] = INFINITY; bi[j] = 0; }
  int qn = 0;                 // entries in this lane's queue
  float thr = INFINITY;       // bd[K-1] as of the last flush: stale, hence conservative
  auto flush = [&]() {
#pragma unroll 1
    for (int s = 0; s < kCrossQueue; ++s) {
      if (!__any(s < qn)) break;
      const float2 e = s_q[s][tid];
      float d = s < qn ? e.x : INFINITY;
      int id = __float_as_int(e.y);
#pragma unroll
      for (int t = 0; t < K; ++t) {   // strict <: the earlier index stays first among equal distances
        const bool lt = d < bd[t];
        const float td = bd[t]; const int ti = bi[t];
        bd[t] = lt ? d : td;  bi[t] = lt ? id : ti;
        d = lt ? td : d;      id = lt ? ti : id;
      }
    }
    qn = 0;
    thr = bd[K - 1];
  };
  for (int r0 = 0; r0 < M; r0 += 1024) {
    const int cnt = min(1024, M - r0);
    __syncthreads();
    for (int j = tid; j < cnt; j += 256) s_ref[j] = make_float4(r[(r0 + j) * 3], r[(r0 + j) * 3 + 1], r[(r0 + j) * 3 + 2], 0.f);
    __syncthreads();
    for (int j = 0; j < cnt; ++j) {
      const float4 v = s_ref[j];
      const float d = metric_sqdist<0>(v.x - qx, v.y - qy, v.z - qz);
      if (d < thr) {
        s_q[qn][tid] = make_float2(d, __int_as_float(r0 + j));
        ++qn;
      }
      if (__any(qn == kCrossQueue)) flush();
    }
  }
  flush();
  if (ok) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      dist2[((size_t)b * N + qi) * K + j] = bd[j];
      idx[((size_t)b * N + qi) * K + j] = bi[j];
    }
  }
}

// out[b,c,m] = features[b,c,idx[b,m]]
__global__ __launch_bounds__(256) void gather_points_kernel(const float* __restrict__ feat, const int* __restrict__ idx,
                                                            size_t total, int C, int N, int Mo, float* __restrict__ out) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t bc = e / Mo;
    const int m = (int)(e - bc * Mo);
    const size_t b = bc / C;
    out[e] = feat[bc * N + idx[b * Mo + m]];
  }
}

}  // namespace
}  // namespace houv

extern "C" int houv_furthest_point_sample(const float* xyz, int B, int N, int npoint, int32_t* idx, void* stream) {
  using namespace houv;
  if (B < 0 || N <= 0 || npoint <= 0 || npoint > N) {
    set_error("houv_furthest_point_sample: bad argument B=%d N=%d npoint=%d", B, N, npoint);
    return 0;
  }
  if (B == 0) return 1;
  if (!xyz || !idx) { set_error("houv_furthest_point_sample: null pointer"); return 0; }
  hipStream_t s = (hipStream_t)stream;
  if (N <= 1024) fps_kernel<256, 4><<<B, 256, 0, s>>>(xyz, N, npoint, idx);
  else if (N <= 4096) fps_kernel<512, 8><<<B, 512, 0, s>>>(xyz, N, npoint, idx);
  else if (N <= 16384) fps_kernel<1024, 16><<<B, 1024, 0, s>>>(xyz, N, npoint, idx);
  else { set_error("houv_furthest_point_sample: N=%d > 16384 not supported", N); return 0; }
  return check_launch("houv_furthest_point_sample") ? 1 : 0;
}

extern "C" int houv_knn_cross(const float* query, const float* ref, int B, int N, int M, int k, float* dist2,
                              int32_t* idx, void* stream) {
  using namespace houv;
  if (B < 0 || N <= 0 || M <= 0 || k > M) {
    set_error("houv_knn_cross: bad argument B=%d N=%d M=%d k=%d", B, N, M, k);
    return 0;
  }
  if (B == 0) return 1;
  if (!query || !ref || !dist2 || !idx) { set_error("houv_knn_cross: null pointer"); return 0; }
  dim3 grid((N + 255) / 256, B);
  hipStream_t s = (hipStream_t)stream;
  if (k == 3) knn_cross_kernel<3><<<grid, 256, 0, s>>>(query, ref, N, M, dist2, idx);
  else if (k == 1) knn_cross_kernel<1><<<grid, 256, 0, s>>>(query, ref, N, M, dist2, idx);
  else if (k == 8) knn_cross_queued_kernel<8><<<grid, 256, 0, s>>>(query, ref, N, M, dist2, idx);
  else { set_error("houv_knn_cross: k must be 1, 3 or 8 (got %d)", k); return 0; }
  return check_launch("houv_knn_cross") ? 1 : 0;
}

extern "C" int houv_gather_points(const float* features, const int32_t* idx, int B, int C, int N, int M, float* out,
                                  void* stream) {
  using namespace houv;
  if (B < 0 || C <= 0 || N <= 0 || M <= 0) { set_error("houv_gather_points: bad argument"); return 0; }
  if (B == 0) return 1;
  if (!features || !idx || !out) { set_error("houv_gather_points: null pointer"); return 0; }
  const size_t total = (size_t)B * C * M;
  size_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  gather_points_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(features, idx, total, C, N, M, out);
  return check_launch("houv_gather_points") ? 1 : 0;
}

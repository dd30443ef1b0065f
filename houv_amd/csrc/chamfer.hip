// chamfer.hip -- stand-alone Chamfer nearest-neighbour op for gfx950 (MI355X).
//
// Replaces utils/metrics/CD/chamfer3D/chamfer3D.cu (NmDistanceKernel :12-134, NmDistanceGradKernel
// :155-174) behind the same contract (caller-allocated outputs, lowest index wins ties, 1/0 return).
// Design (not a translation of the CUDA kernel):
//   * one launch covers both directions and the whole batch; grid = (B * query-blocks, 2);
//   * every lane owns Q query points in registers, the reference cloud is staged through LDS as
//     float4 and read back with wave-uniform (broadcast) ds_read_b128, so one LDS read feeds
//     64 x Q distance evaluations;
//   * the inner loop is min-only (v_min3_f32 over two references at a time); the arg-min is
//     tracked per 32-reference sub-tile and recovered exactly afterwards by re-evaluating that one
//     sub-tile with bit-identical arithmetic ("deferred index");
//   * the running minimum lives in registers for the whole sweep (the reference spills it to global
//     memory every 512 references, chamfer3D.cu:126-129).
#include "houv_common.h"

namespace houv {
namespace {

constexpr int kBlock = 256;
constexpr int kRefTile = 2048;   // references staged per LDS pass: 2048 * 16 B = 32 KiB

template <int Q>
__global__ __launch_bounds__(kBlock) void chamfer_nn_kernel(const float* __restrict__ xyz1,
                                                            const float* __restrict__ xyz2, int N, int M, int nqb,
                                                            float* __restrict__ dist1, float* __restrict__ dist2,
                                                            int* __restrict__ idx1, int* __restrict__ idx2) {
  __shared__ float4 s_ref[kRefTile];
  const bool fwd = blockIdx.y == 0;
  const int nq = fwd ? N : M, nr = fwd ? M : N;
  const int b = blockIdx.x / nqb;
  const int q0 = (blockIdx.x - b * nqb) * (kBlock * Q);
  if (q0 >= nq) return;   // uniform for the whole workgroup (grid is sized for max(N, M))
  const float* __restrict__ q = (fwd ? xyz1 : xyz2) + (size_t)b * nq * 3;
  const float* __restrict__ r = (fwd ? xyz2 : xyz1) + (size_t)b * nr * 3;
  float* __restrict__ dist = (fwd ? dist1 : dist2) + (size_t)b * nq;
  int* __restrict__ idx = (fwd ? idx1 : idx2) + (size_t)b * nq;
  const int tid = threadIdx.x;

  float qx[Q], qy[Q], qz[Q], best[Q];
  int btile[Q];
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    const int qi = q0 + k * kBlock + tid;
    const bool ok = qi < nq;
    qx[k] = ok ? q[qi * 3 + 0] : 0.f;
    qy[k] = ok ? q[qi * 3 + 1] : 0.f;
    qz[k] = ok ? q[qi * 3 + 2] : 0.f;
    best[k] = INFINITY;
    btile[k] = 0;
  }

  for (int r0 = 0; r0 < nr; r0 += kRefTile) {
    const int cnt = min(kRefTile, nr - r0);
    const int ntile = (cnt + kSub - 1) / kSub;
    __syncthreads();
    for (int j = tid; j < ntile * kSub; j += kBlock) {
      float4 v = make_float4(INFINITY, INFINITY, INFINITY, 0.f);   // padding never wins
      if (j < cnt) {
        const float* p = r + (size_t)(r0 + j) * 3;
        v = make_float4(p[0], p[1], p[2], 0.f);
      }
      s_ref[j] = v;
    }
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
      float tm[Q];
#pragma unroll
      for (int k = 0; k < Q; ++k) tm[k] = INFINITY;
      const float4* rp = s_ref + t * kSub;
#pragma unroll 8
      for (int j = 0; j < kSub; j += 2) {
        const float4 a = rp[j], c = rp[j + 1];
        asm volatile("" ::"v"(a.w), "v"(c.w));   // keep the loads ds_read_b128 (4 LDS cycles, not b96's 8)
#pragma unroll
        for (int k = 0; k < Q; ++k) {
          const float d0 = metric_sqdist<0>(a.x - qx[k], a.y - qy[k], a.z - qz[k]);
          const float d1 = metric_sqdist<0>(c.x - qx[k], c.y - qy[k], c.z - qz[k]);
          tm[k] = min3f(tm[k], d0, d1);
        }
      }
#pragma unroll
      for (int k = 0; k < Q; ++k) {
        const bool lt = tm[k] < best[k];   // strict: the earlier sub-tile keeps ties
        best[k] = lt ? tm[k] : best[k];
        btile[k] = lt ? (r0 / kSub + t) : btile[k];
      }
    }
  }

  // exact index recovery: re-evaluate the winning sub-tile (descending, so the lowest index wins)
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    const int qi = q0 + k * kBlock + tid;
    if (qi >= nq) continue;
    const int base = btile[k] * kSub;
    int found = base;
    float bd = best[k];
    if (!(bd < INFINITY)) {
      // no finite distance at all (NaN / overflowing input): the reference reports ref 0 (chamfer3D.cu:37)
      bd = metric_sqdist<0>(r[0] - qx[k], r[1] - qy[k], r[2] - qz[k]);
      found = 0;
    } else {
      for (int j = kSub - 1; j >= 0; --j) {
        const int jj = base + j;
        if (jj < nr) {
          const float d = metric_sqdist<0>(r[jj * 3 + 0] - qx[k], r[jj * 3 + 1] - qy[k], r[jj * 3 + 2] - qz[k]);
          found = (d == bd) ? jj : found;
        }
      }
    }
    dist[qi] = bd;
    idx[qi] = found;
  }
}

// Backward: same arithmetic and accumulate-into-zeroed-buffers contract as NmDistanceGradKernel
// (chamfer3D.cu:155-174), but one thread per (batch, point) over the whole batch in one launch per
// direction pair instead of a single block column walking the batch serially.
__global__ __launch_bounds__(256) void chamfer_grad_kernel(const float* __restrict__ xyz1,
                                                           const float* __restrict__ xyz2, int B, int N, int M,
                                                           const float* __restrict__ g1, const float* __restrict__ g2,
                                                           const int* __restrict__ idx1, const int* __restrict__ idx2,
                                                           float* __restrict__ gx1, float* __restrict__ gx2) {
  const bool fwd = blockIdx.y == 0;
  const int nq = fwd ? N : M, nr = fwd ? M : N;
  const float* __restrict__ q = fwd ? xyz1 : xyz2;
  const float* __restrict__ r = fwd ? xyz2 : xyz1;
  const float* __restrict__ g = fwd ? g1 : g2;
  const int* __restrict__ idx = fwd ? idx1 : idx2;
  float* __restrict__ gq = fwd ? gx1 : gx2;
  float* __restrict__ gr = fwd ? gx2 : gx1;
  const size_t total = (size_t)B * nq;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const size_t b = e / nq;
    const int j2 = idx[e];
    const float* qp = q + e * 3;
    const float* rp = r + (b * nr + j2) * 3;
    const float gg = g[e] * 2.f;
    const float dx = gg * (qp[0] - rp[0]), dy = gg * (qp[1] - rp[1]), dz = gg * (qp[2] - rp[2]);
    atomicAdd(gq + e * 3 + 0, dx);
    atomicAdd(gq + e * 3 + 1, dy);
    atomicAdd(gq + e * 3 + 2, dz);
    float* grp = gr + (b * nr + j2) * 3;
    atomicAdd(grp + 0, -dx);
    atomicAdd(grp + 1, -dy);
    atomicAdd(grp + 2, -dz);
  }
}

// LDS form of the backward pass, one workgroup per batch instance: both directions' contributions are summed in
// two LDS accumulators (ds_add_f32, no global atomics) and then added to the caller's buffers with coalesced
// read-modify-writes (the workgroup owns its instance, so the accumulate-into-zeroed contract needs no atomics).
// HBM-bound: 176 KB per 2048^2 instance.  Used when (N+M)*12 B fits in LDS; chamfer_grad_kernel otherwise.
__global__ __launch_bounds__(512) void chamfer_grad_lds_kernel(const float* __restrict__ xyz1,
                                                               const float* __restrict__ xyz2, int N, int M,
                                                               const float* __restrict__ g1, const float* __restrict__ g2,
                                                               const int* __restrict__ idx1, const int* __restrict__ idx2,
                                                               float* __restrict__ gx1, float* __restrict__ gx2) {
  extern __shared__ float s_acc[];          // [3N] grad of cloud 1, then [3M] grad of cloud 2
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* __restrict__ p1 = xyz1 + (size_t)b * N * 3;
  const float* __restrict__ p2 = xyz2 + (size_t)b * M * 3;
  float* a1 = s_acc;
  float* a2 = s_acc + 3 * N;
  for (int i = tid; i < 3 * (N + M); i += blockDim.x) s_acc[i] = 0.f;
  __syncthreads();
  for (int j = tid; j < N; j += blockDim.x) {
    const int j2 = idx1[(size_t)b * N + j];
    const float gg = g1[(size_t)b * N + j] * 2.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = gg * (p1[j * 3 + c] - p2[j2 * 3 + c]);
      atomicAdd(&a1[j * 3 + c], v);
      atomicAdd(&a2[j2 * 3 + c], -v);
    }
  }
  for (int j = tid; j < M; j += blockDim.x) {
    const int j2 = idx2[(size_t)b * M + j];
    const float gg = g2[(size_t)b * M + j] * 2.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = gg * (p2[j * 3 + c] - p1[j2 * 3 + c]);
      atomicAdd(&a2[j * 3 + c], v);
      atomicAdd(&a1[j2 * 3 + c], -v);
    }
  }
  __syncthreads();
  float* o1 = gx1 + (size_t)b * N * 3;
  float* o2 = gx2 + (size_t)b * M * 3;
  for (int i = tid; i < 3 * N; i += blockDim.x) o1[i] += a1[i];
  for (int i = tid; i < 3 * M; i += blockDim.x) o2[i] += a2[i];
}

}  // namespace
}  // namespace houv

extern "C" int houv_chamfer_forward(const float* xyz1, const float* xyz2, int B, int N, int M, float* dist1,
                                    float* dist2, int32_t* idx1, int32_t* idx2, void* stream) {
  using namespace houv;
  if (B < 0 || N <= 0 || M <= 0) {
    set_error("houv_chamfer_forward: bad shape B=%d N=%d M=%d (N, M must be >= 1)", B, N, M);
    return 0;
  }
  if (B == 0) return 1;
  if (!xyz1 || !xyz2 || !dist1 || !dist2 || !idx1 || !idx2) {
    set_error("houv_chamfer_forward: null pointer");
    return 0;
  }
  hipStream_t s = (hipStream_t)stream;
  const int mx = N > M ? N : M;
  const int q = mx <= kBlock ? 1 : (mx <= 2 * kBlock ? 2 : 4);
  const int nqb = (mx + kBlock * q - 1) / (kBlock * q);
  if ((long long)B * nqb > 0x7fffffffLL) {
    set_error("houv_chamfer_forward: batch too large");
    return 0;
  }
  dim3 grid((unsigned)(B * nqb), 2, 1);
  if (q == 1) chamfer_nn_kernel<1><<<grid, kBlock, 0, s>>>(xyz1, xyz2, N, M, nqb, dist1, dist2, idx1, idx2);
  else if (q == 2) chamfer_nn_kernel<2><<<grid, kBlock, 0, s>>>(xyz1, xyz2, N, M, nqb, dist1, dist2, idx1, idx2);
  else chamfer_nn_kernel<4><<<grid, kBlock, 0, s>>>(xyz1, xyz2, N, M, nqb, dist1, dist2, idx1, idx2);
  return check_launch("houv_chamfer_forward") ? 1 : 0;
}

extern "C" int houv_chamfer_backward(const float* xyz1, const float* xyz2, int B, int N, int M,
                                     const float* graddist1, const float* graddist2, const int32_t* idx1,
                                     const int32_t* idx2, float* gradxyz1, float* gradxyz2, void* stream) {
  using namespace houv;
  if (B < 0 || N <= 0 || M <= 0) {
    set_error("houv_chamfer_backward: bad shape B=%d N=%d M=%d", B, N, M);
    return 0;
  }
  if (B == 0) return 1;
  if (!xyz1 || !xyz2 || !graddist1 || !graddist2 || !idx1 || !idx2 || !gradxyz1 || !gradxyz2) {
    set_error("houv_chamfer_backward: null pointer");
    return 0;
  }
  const size_t lds = (size_t)3 * ((size_t)N + (size_t)M) * sizeof(float);
  if (lds <= 150 * 1024) {
    if (hipFuncSetAttribute((const void*)chamfer_grad_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) ==
        hipSuccess) {
      chamfer_grad_lds_kernel<<<B, 512, lds, (hipStream_t)stream>>>(xyz1, xyz2, N, M, graddist1, graddist2, idx1, idx2,
                                                                   gradxyz1, gradxyz2);
      return check_launch("houv_chamfer_backward") ? 1 : 0;
    }
    (void)hipGetLastError();
  }
  const size_t total = (size_t)B * (size_t)(N > M ? N : M);
  size_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  dim3 grid((unsigned)blocks, 2, 1);
  chamfer_grad_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(xyz1, xyz2, B, N, M, graddist1, graddist2, idx1, idx2,
                                                            gradxyz1, gradxyz2);
  return check_launch("houv_chamfer_backward") ? 1 : 0;
}

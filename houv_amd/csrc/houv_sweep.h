// houv_sweep.h -- the LDS-broadcast brute-force nearest-neighbour sweep shared by the fused HOUV loop (solve.hip)
// and the ICP refinement kernel (icp.hip), plus the workgroup reduction they both use.
#pragma once
#include "houv_common.h"

namespace houv {

constexpr int kAccStride = 16;   // row stride (floats) of the per-wave reduction scratch

// ------------------------------------------------------------------------------------------------
// The brute-force sweep: for each of this lane's Q queries, min over all references of the NMET
// squared distances, plus the id of the 32-reference sub-tile that produced each minimum.
// ------------------------------------------------------------------------------------------------
template <int Q, int NMET>
__device__ __forceinline__ void sweep(const float4* __restrict__ refs, int ntile, const float (&qx)[Q],
                                      const float (&qy)[Q], const float (&qz)[Q], float (&best)[Q][NMET],
                                      int (&btile)[Q][NMET]) {
#pragma unroll
  for (int k = 0; k < Q; ++k)
#pragma unroll
    for (int m = 0; m < NMET; ++m) {
      best[k][m] = INFINITY;
      btile[k][m] = 0;
    }
  for (int t = 0; t < ntile; ++t) {
    float tm[Q][NMET];
#pragma unroll
    for (int k = 0; k < Q; ++k)
#pragma unroll
      for (int m = 0; m < NMET; ++m) tm[k][m] = INFINITY;
    const float4* rp = refs + t * kSub;
#pragma unroll 4
    for (int j = 0; j < kSub; j += 2) {
      const float4 a = rp[j], c = rp[j + 1];
      // keep .w "used" so the loads stay ds_read_b128 (4 LDS cycles) instead of ds_read_b96 (8)
      asm volatile("" ::"v"(a.w), "v"(c.w));
#pragma unroll
      for (int k = 0; k < Q; ++k) {
        const float ax = a.x - qx[k], ay = a.y - qy[k], az = a.z - qz[k];
        const float cx = c.x - qx[k], cy = c.y - qy[k], cz = c.z - qz[k];
        if constexpr (NMET == 4) {
          const float axx = ax * ax, ayy = ay * ay, cxx = cx * cx, cyy = cy * cy;
          const float a3 = __builtin_fmaf(ay, ay, axx), c3 = __builtin_fmaf(cy, cy, cxx);   // z dropped
          const float a1 = __builtin_fmaf(az, az, ayy), c1 = __builtin_fmaf(cz, cz, cyy);   // x dropped
          const float a2 = __builtin_fmaf(az, az, axx), c2 = __builtin_fmaf(cz, cz, cxx);   // y dropped
          const float a0 = __builtin_fmaf(az, az, a3), c0 = __builtin_fmaf(cz, cz, c3);     // full
          tm[k][0] = min3f(tm[k][0], a0, c0);
          tm[k][1] = min3f(tm[k][1], a1, c1);
          tm[k][2] = min3f(tm[k][2], a2, c2);
          tm[k][3] = min3f(tm[k][3], a3, c3);
        } else {
          tm[k][0] = min3f(tm[k][0], metric_sqdist<0>(ax, ay, az), metric_sqdist<0>(cx, cy, cz));
        }
      }
    }
#pragma unroll
    for (int k = 0; k < Q; ++k)
#pragma unroll
      for (int m = 0; m < NMET; ++m) {
        const bool lt = tm[k][m] < best[k][m];   // strict: earlier sub-tile keeps ties (lowest index wins)
        best[k][m] = lt ? tm[k][m] : best[k][m];
        btile[k][m] = lt ? t : btile[k][m];
      }
  }
}

// Sum NV per-thread values over the workgroup into out[0..NV) (LDS).
template <int BLOCK, int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* red, float* out) {
  constexpr int NW = BLOCK / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[wave * kAccStride + i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) a += red[w * kAccStride + threadIdx.x];
    out[threadIdx.x] = a;
  }
}


// Exact NN recovery for one query: re-evaluate the winning 32-reference sub-tile with the bit-identical expression
// and return the lowest matching reference.  The scan order is rotated by `rot` (= lane id & 31): sub-tile bases are
// 512 B apart, so an un-rotated scan puts all lanes of a ds_read_b128 group on the same LDS bank quad.
template <int MET, int BATCH>
__device__ __forceinline__ float4 recover_nn(const float4* __restrict__ rp, float qx, float qy, float qz, float bd, int rot) {
  int jb = kSub;
#pragma unroll 1
  for (int c = 0; c < kSub; c += BATCH) {
    float4 r[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) r[u] = rp[(c + u + rot) & (kSub - 1)];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) asm volatile("" ::"v"(r[u].x), "v"(r[u].y), "v"(r[u].z), "v"(r[u].w));   // keep b128
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const float d = metric_sqdist<MET>(r[u].x - qx, r[u].y - qy, r[u].z - qz);
      jb = min(jb, (d == bd) ? ((c + u + rot) & (kSub - 1)) : kSub);   // lowest matching index, whatever the order
    }
  }
  return rp[jb & (kSub - 1)];
}

}  // namespace houv

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
__device__ __forceinline__ float4 recover_nn(const float4* __restrict__ rp, float qx, float qy, float qz, float bd, int rot,
                                             int& j_out) {
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
  j_out = jb & (kSub - 1);
  return rp[j_out];
}


// ---------------------------------------------------------------------------------------------------------------
// EXACT pruned sweep (opt-in, houv_solve_iterate_pruned).  References are grouped in the same 32-point sub-tiles as
// the brute-force sweep; every sub-tile carries an axis-aligned bounding box.  For a query with an upper bound ub[m]
// on its nearest-neighbour distance under metric m (the distance to the point that was its NN in the previous
// iteration -- an actual point, so the bound is attained), a sub-tile whose box is farther than ub[m] for every
// metric cannot contain any metric's NN and is skipped.  The surviving sub-tiles are visited in ascending order with
// the same min3 / strict-< bookkeeping as sweep(), so (best, btile) come out BIT-IDENTICAL to the brute-force sweep.
// Per-lane sub-tile lists are 64-bit masks; lanes walk their own lists (LDS gathers, scan order rotated per lane).
// ---------------------------------------------------------------------------------------------------------------
template <int Q, int NMET>
__device__ __forceinline__ void pruned_sweep(const float4* __restrict__ refs, const float4* __restrict__ boxes, int ntile,
                                             const float (&qx)[Q], const float (&qy)[Q], const float (&qz)[Q],
                                             const short* __restrict__ prev, int prev_stride, int count, int block,
                                             int rot, float (&best)[Q][NMET], int (&btile)[Q][NMET]) {
#pragma unroll   // static k: runtime-indexed register arrays would be demoted to scratch
  for (int k = 0; k < Q; ++k) {
    // upper bounds: the distance to last iteration's NN (an actual point, so the bound is attained); computed just in
    // time per query to keep the register footprint of the tile walk small
    float ubs[NMET];
    {
      const int i = k * block + (int)threadIdx.x;
      const bool ok = i < count;
#pragma unroll
      for (int m = 0; m < NMET; ++m) ubs[m] = INFINITY;
      if (ok) {
        { const float4 r = refs[prev[0 * prev_stride + i]]; ubs[0] = metric_sqdist<0>(r.x - qx[k], r.y - qy[k], r.z - qz[k]); }
        if constexpr (NMET == 4) {
          { const float4 r = refs[prev[1 * prev_stride + i]]; ubs[1] = metric_sqdist<1>(r.x - qx[k], r.y - qy[k], r.z - qz[k]); }
          { const float4 r = refs[prev[2 * prev_stride + i]]; ubs[2] = metric_sqdist<2>(r.x - qx[k], r.y - qy[k], r.z - qz[k]); }
          { const float4 r = refs[prev[3 * prev_stride + i]]; ubs[3] = metric_sqdist<3>(r.x - qx[k], r.y - qy[k], r.z - qz[k]); }
        }
      }
#pragma unroll
      for (int m = 0; m < NMET; ++m) ubs[m] = ubs[m] * 1.00001f + 1e-30f;   // LB is rounded: keep the test conservative
    }
    unsigned long long un = 0ull;
    for (int t = 0; t < ntile; ++t) {                 // wave-uniform: box reads are LDS broadcasts
      const float4 lo = boxes[2 * t], hi = boxes[2 * t + 1];
      const float dx = fmaxf(fmaxf(lo.x - qx[k], qx[k] - hi.x), 0.f);
      const float dy = fmaxf(fmaxf(lo.y - qy[k], qy[k] - hi.y), 0.f);
      const float dz = fmaxf(fmaxf(lo.z - qz[k], qz[k] - hi.z), 0.f);
      const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
      bool in = (xx + yy + zz) <= ubs[0];
      if constexpr (NMET == 4) in = in || (yy + zz) <= ubs[1] || (xx + zz) <= ubs[2] || (xx + yy) <= ubs[3];
      un |= in ? (1ull << t) : 0ull;
    }
    float bk[NMET];
    int tk[NMET];
#pragma unroll
    for (int m = 0; m < NMET; ++m) { bk[m] = INFINITY; tk[m] = 0; }
    while (__any(un != 0ull)) {
      const bool act = un != 0ull;
      const int t = act ? (__ffsll((long long)un) - 1) : 0;
      un = act ? (un & (un - 1ull)) : 0ull;
      const float4* rp = refs + t * kSub;
      float tm[NMET];
#pragma unroll
      for (int m = 0; m < NMET; ++m) tm[m] = INFINITY;
#pragma unroll 1
      for (int j0 = 0; j0 < kSub; j0 += 8) {
        // 8 gathered reads issued back to back, ONE wait, then 4 x (two references) of arithmetic
        float4 r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = rp[(j0 + u + rot) & (kSub - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) asm volatile("" ::"v"(r[u].x), "v"(r[u].y), "v"(r[u].z), "v"(r[u].w));   // keep b128
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
          const float4 a = r[u], c = r[u + 1];
          const float ax = a.x - qx[k], ay = a.y - qy[k], az = a.z - qz[k];
          const float cx = c.x - qx[k], cy = c.y - qy[k], cz = c.z - qz[k];
          if constexpr (NMET == 4) {
            const float axx = ax * ax, ayy = ay * ay, cxx = cx * cx, cyy = cy * cy;
            const float a3 = __builtin_fmaf(ay, ay, axx), c3 = __builtin_fmaf(cy, cy, cxx);
            const float a1 = __builtin_fmaf(az, az, ayy), c1 = __builtin_fmaf(cz, cz, cyy);
            const float a2 = __builtin_fmaf(az, az, axx), c2 = __builtin_fmaf(cz, cz, cxx);
            const float a0 = __builtin_fmaf(az, az, a3), c0 = __builtin_fmaf(cz, cz, c3);
            tm[0] = min3f(tm[0], a0, c0);
            tm[1] = min3f(tm[1], a1, c1);
            tm[2] = min3f(tm[2], a2, c2);
            tm[3] = min3f(tm[3], a3, c3);
          } else {
            tm[0] = min3f(tm[0], metric_sqdist<0>(ax, ay, az), metric_sqdist<0>(cx, cy, cz));
          }
        }
      }
#pragma unroll
      for (int m = 0; m < NMET; ++m) {
        const bool lt = act && (tm[m] < bk[m]);
        bk[m] = lt ? tm[m] : bk[m];
        tk[m] = lt ? t : tk[m];
      }
    }
#pragma unroll
    for (int m = 0; m < NMET; ++m) { best[k][m] = bk[m]; btile[k][m] = tk[m]; }
  }
}

// Axis-aligned boxes of the 32-point sub-tiles of a cloud whose point (k*BLOCK + tid) lives in this lane's registers:
// a sub-tile is one 32-lane half of a wave, so five xor-shuffles per coordinate reduce it.  box[2t] = lo, box[2t+1] = hi.
template <int BLOCK, int Q>
__device__ __forceinline__ void tile_boxes(const float (&x)[Q], const float (&y)[Q], const float (&z)[Q], int count,
                                           int ntile, float4* __restrict__ box) {
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    const int i = k * BLOCK + threadIdx.x;
    const bool ok = i < count;
    float lx = ok ? x[k] : INFINITY, ly = ok ? y[k] : INFINITY, lz = ok ? z[k] : INFINITY;
    float hx = ok ? x[k] : -INFINITY, hy = ok ? y[k] : -INFINITY, hz = ok ? z[k] : -INFINITY;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
      lx = fminf(lx, __shfl_xor(lx, o, 64)); ly = fminf(ly, __shfl_xor(ly, o, 64)); lz = fminf(lz, __shfl_xor(lz, o, 64));
      hx = fmaxf(hx, __shfl_xor(hx, o, 64)); hy = fmaxf(hy, __shfl_xor(hy, o, 64)); hz = fmaxf(hz, __shfl_xor(hz, o, 64));
    }
    const int t = i >> 5;
    if ((threadIdx.x & 31) == 0 && t < ntile) {
      box[2 * t] = make_float4(lx, ly, lz, 0.f);
      box[2 * t + 1] = make_float4(hx, hy, hz, 0.f);
    }
  }
}

}  // namespace houv

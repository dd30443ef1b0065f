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
#include <stdlib.h>

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

// ---------------------------------------------------------------------------------------------------------------
// Filtered sweep (the default): the SAME result as chamfer_nn_kernel, bit for bit, for 4.4 instead of 7.1 VALU issue
// slots per point pair.
//   * filter: e_j = |r_j|^2 - 2 q.r_j  (= |q - r_j|^2 - |q|^2) as three FMAs on (x, y, z, |r|^2) float4 references --
//     the .w slot of the LDS tile, unused by the direct sweep, carries |r|^2, so the LDS traffic is unchanged;
//   * e differs from the exact direct-difference d (the reference's arithmetic, chamfer3D.cu:31-36) by rounding, so
//     it only SELECTS: per query the two smallest sub-tile minima of e (and their sub-tiles) plus the third smallest
//     value are tracked (v_med3 updates, 8 instructions per query and 32 references); afterwards
//       - the best sub-tile is re-evaluated with the exact expression (min d, lowest index) from the LDS tile,
//       - the second one too when its minimum lies within tau of the best,
//       - and when even the third does, the query falls back to an exact scan of all references (rare: ~1e-4);
//   * tau bounds how far the ranking by e can disagree with the ranking by d.  With u = 2^-24, R = max |r|:
//       |e~ - E| <= 6.02 u (R + |q|)^2   (3 roundings for |r|^2, 3 for the FMA chain),   |d - D| <= 5.01 u D,
//     so the oracle's arg-min j* satisfies  e~(j*) <= e~(j) + 2 * 6.02 u (R+|q|)^2 + 10.1 u D_j  for every j;
//     tau = 25 u (R + |q|)^2 covers it (derivation in DESIGN.md 3.2).  NaN / Inf coordinates behave as in the
//     direct kernel: such references never win (v_min3 drops NaN), and a query without any finite distance reports
//     reference 0 (chamfer3D.cu:37).
// ---------------------------------------------------------------------------------------------------------------
constexpr float kUlpHalf = 5.9604645e-8f;   // u = 2^-24
constexpr int kList2 = 1024, kList3 = 256;   // capacities of the per-workgroup recovery work lists (LDS)

// exact (direct-difference) minimum of one 32-reference sub-tile and the lowest index attaining it
__device__ __forceinline__ float filter_e(float ex, float ey, float ez, float px, float py, float pz, float pw) {
  return __builtin_fmaf(ex, px, __builtin_fmaf(ey, py, __builtin_fmaf(ez, pz, pw)));
}

template <bool FROM_LDS>
__device__ __forceinline__ void exact_tile(const float4* __restrict__ s_ref, const float* __restrict__ r, int tile, int nr,
                                           float qx, float qy, float qz, int rot, float& bd, int& jb) {
  const int base = tile * kSub;
#pragma unroll 4
  for (int jr = 0; jr < kSub; ++jr) {
    const int j = FROM_LDS ? ((jr + rot) & (kSub - 1)) : jr;   // rotated: lanes sit on different bank quads whatever their tile
    const int jj = base + j;
    float px, py, pz;
    if constexpr (FROM_LDS) {
      const float4 p = s_ref[jj];
      px = p.x; py = p.y; pz = p.z;
    } else {
      const int jc = jj < nr ? jj : 0;
      px = r[jc * 3 + 0]; py = r[jc * 3 + 1]; pz = r[jc * 3 + 2];
    }
    const float d = metric_sqdist<0>(px - qx, py - qy, pz - qz);
    const bool better = (jj < nr) && (d < bd || (d == bd && jj < jb));
    bd = better ? d : bd;
    jb = better ? jj : jb;
  }
}

// ---- the filter sweep of one staged chunk for Q = 8, in assembly with hand-allocated registers ----
// gfx950 issues a wave64 v_fma / v_fmac every ~2.6 clk -- unless operands share a VGPR bank (bank = register index mod 4):
// `v_fma d, s0, s1, s2` with s0 and s1 in one bank, or `v_fmac d, s0, s1` with s0 in d's bank, takes 4.5 clk
// (scripts/ubench/gen_vgpr_banks.py -> profiles/r02_vgpr_banks.txt).  hipcc's register allocator does not know; in the
// compiled loop 185 of the 768 FMAs per sub-tile paid twice, and this loop is nothing but FMAs.  Register plan:
//   v20-27 ex[k]   v28-35 ey[k]   v36+4k: ez[k] (bank 0), tm[k] (1), e0 (2), e1 (3)
//   v68-75 / v76-83: two reference pairs (x, y, z, |r|^2 in banks 0, 1, 2, 3), double-buffered ds_read_b128
//   v84-91 best  v92-99 sec  v100-107 third  v108-115 bt  v116-123 bt2  v124 sub-tile id  v125 LDS address  v127 +inf
// so that  v_fma e, z(2), ez(0), w(3);  v_fmac e, y(1), ey;  v_fmac e, x(0), ex  with e in bank 2 / 3 never conflict.
// Same instructions on the same values as the C++ loop below (which serves every other Q): bit-identical results.
typedef float chf4 __attribute__((ext_vector_type(4)));
typedef int chi4 __attribute__((ext_vector_type(4)));
#define HC_S_(x) #x
#define HC_S(x) HC_S_(x)
#define HC_K ".irp k,0,1,2,3,4,5,6,7\n"
#define HC_E0 "v[38+4*\\k]"
#define HC_E1 "v[39+4*\\k]"
#define HC_TM "v[37+4*\\k]"
#define HC_EZ "v[36+4*\\k]"
#define HC_LOAD(buf, pair) \
  "ds_read_b128 v[" HC_S(buf) ":" HC_S(buf) "+3], v125 offset:32*" HC_S(pair) "\n ds_read_b128 v[" HC_S(buf) "+4:" HC_S(buf) "+7], v125 offset:32*" HC_S(pair) "+16\n"
#define HC_EVAL(buf, acc)                                                                                                  \
  HC_K "v_fma_f32 " HC_E0 ", v[" HC_S(buf) "+2], " HC_EZ ", v[" HC_S(buf) "+3]\n v_fma_f32 " HC_E1 ", v[" HC_S(buf) "+6], " HC_EZ ", v[" HC_S(buf) "+7]\n.endr\n" \
  HC_K "v_fmac_f32 " HC_E0 ", v[" HC_S(buf) "+1], v[28+\\k]\n v_fmac_f32 " HC_E1 ", v[" HC_S(buf) "+5], v[28+\\k]\n.endr\n"                         \
  HC_K "v_fmac_f32 " HC_E0 ", v[" HC_S(buf) "+0], v[20+\\k]\n v_fmac_f32 " HC_E1 ", v[" HC_S(buf) "+4], v[20+\\k]\n.endr\n"                         \
  HC_K "v_min3_f32 " HC_TM ", " acc ", " HC_E0 ", " HC_E1 "\n.endr\n"
#define HC_STEP(buf, pair) HC_EVAL(buf, HC_TM) HC_LOAD(buf, pair) "s_waitcnt lgkmcnt(2)\n"

__device__ __forceinline__ void filter_sweep8(unsigned lds_addr, int ntile, int tile0, const float (&ex)[8], const float (&ey)[8],
                                              const float (&ez)[8], float (&best)[8], float (&sec)[8], float (&third)[8],
                                              int (&bt)[8], int (&bt2)[8]) {
  chf4 x0 = {ex[0], ex[1], ex[2], ex[3]}, x1 = {ex[4], ex[5], ex[6], ex[7]};
  chf4 y0 = {ey[0], ey[1], ey[2], ey[3]}, y1 = {ey[4], ey[5], ey[6], ey[7]};
  chf4 b0 = {best[0], best[1], best[2], best[3]}, b1 = {best[4], best[5], best[6], best[7]};
  chf4 s0 = {sec[0], sec[1], sec[2], sec[3]}, s1 = {sec[4], sec[5], sec[6], sec[7]};
  chf4 t0 = {third[0], third[1], third[2], third[3]}, t1 = {third[4], third[5], third[6], third[7]};
  chi4 i0 = {bt[0], bt[1], bt[2], bt[3]}, i1 = {bt[4], bt[5], bt[6], bt[7]};
  chi4 j0 = {bt2[0], bt2[1], bt2[2], bt2[3]}, j1 = {bt2[4], bt2[5], bt2[6], bt2[7]};
  unsigned long long m;
  asm volatile(
      "v_mov_b32 v127, 0x7f800000\n"
      "1:\n" HC_LOAD(68, 0) HC_LOAD(76, 1) "s_waitcnt lgkmcnt(2)\n"
      HC_EVAL(68, "v127") HC_LOAD(68, 2) "s_waitcnt lgkmcnt(2)\n"
      HC_STEP(76, 3) HC_STEP(68, 4) HC_STEP(76, 5) HC_STEP(68, 6) HC_STEP(76, 7) HC_STEP(68, 8) HC_STEP(76, 9) HC_STEP(68, 10)
      HC_STEP(76, 11) HC_STEP(68, 12) HC_STEP(76, 13) HC_STEP(68, 14) HC_STEP(76, 15)
      HC_EVAL(68, HC_TM) "s_waitcnt lgkmcnt(0)\n" HC_EVAL(76, HC_TM)
      // per query: the two smallest sub-tile minima with their sub-tiles, and the third smallest value
      HC_K "v_cmp_lt_f32_e32 vcc, " HC_TM ", v[84+\\k]\n v_cmp_lt_f32_e64 %[m], " HC_TM ", v[92+\\k]\n"
      "v_med3_f32 v[100+\\k], " HC_TM ", v[92+\\k], v[100+\\k]\n v_med3_f32 v[92+\\k], " HC_TM ", v[84+\\k], v[92+\\k]\n"
      "v_cndmask_b32_e64 " HC_E0 ", v[116+\\k], v124, %[m]\n v_cndmask_b32_e32 v[116+\\k], " HC_E0 ", v[108+\\k], vcc\n"
      "v_cndmask_b32_e32 v[108+\\k], v[108+\\k], v124, vcc\n v_cndmask_b32_e32 v[84+\\k], v[84+\\k], " HC_TM ", vcc\n.endr\n"
      "v_add_u32_e32 v124, 1, v124\n v_add_u32_e32 v125, 0x200, v125\n"
      "s_sub_u32 %[n], %[n], 1\n s_cmp_lg_u32 %[n], 0\n s_cbranch_scc1 1b\n"
      : "+{v[84:87]}"(b0), "+{v[88:91]}"(b1), "+{v[92:95]}"(s0), "+{v[96:99]}"(s1), "+{v[100:103]}"(t0), "+{v[104:107]}"(t1),
        "+{v[108:111]}"(i0), "+{v[112:115]}"(i1), "+{v[116:119]}"(j0), "+{v[120:123]}"(j1), "+{v124}"(tile0), "+{v125}"(lds_addr),
        [n] "+s"(ntile), [m] "=&s"(m)
      : "{v[20:23]}"(x0), "{v[24:27]}"(x1), "{v[28:31]}"(y0), "{v[32:35]}"(y1), "{v36}"(ez[0]), "{v40}"(ez[1]), "{v44}"(ez[2]),
        "{v48}"(ez[3]), "{v52}"(ez[4]), "{v56}"(ez[5]), "{v60}"(ez[6]), "{v64}"(ez[7])
      : "v37", "v38", "v39", "v41", "v42", "v43", "v45", "v46", "v47", "v49", "v50", "v51", "v53", "v54", "v55", "v57", "v58", "v59",
        "v61", "v62", "v63", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",
        "v80", "v81", "v82", "v83", "v127", "vcc", "scc", "memory");
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    best[k] = b0[k]; best[k + 4] = b1[k]; sec[k] = s0[k]; sec[k + 4] = s1[k]; third[k] = t0[k]; third[k + 4] = t1[k];
    bt[k] = i0[k]; bt[k + 4] = i1[k]; bt2[k] = j0[k]; bt2[k + 4] = j1[k];
  }
}

template <int Q>
__global__ __launch_bounds__(kBlock, 4) void chamfer_nn_filter_kernel(const float* __restrict__ xyz1,
                                                                   const float* __restrict__ xyz2, int N, int M, int nqb,
                                                                   float* dist1, float* dist2, int* idx1, int* idx2) {
  __shared__ float4 s_ref[kRefTile];
  __shared__ unsigned s_rmax;   // bit pattern of max |r|^2 (non-negative floats order like unsigned integers)
  __shared__ unsigned s_list2[kList2];          // work list "second sub-tile": query (11 bits) | sub-tile << 11
  __shared__ unsigned short s_list3[kList3];    // work list "exact full scan": query
  __shared__ int s_n2, s_n3;
  const bool fwd = blockIdx.y == 0;
  const int nq = fwd ? N : M, nr = fwd ? M : N;
  const int b = blockIdx.x / nqb;
  const int q0 = (blockIdx.x - b * nqb) * (kBlock * Q);
  if (q0 >= nq) return;   // uniform for the whole workgroup (grid is sized for max(N, M))
  const float* __restrict__ q = (fwd ? xyz1 : xyz2) + (size_t)b * nq * 3;
  const float* __restrict__ r = (fwd ? xyz2 : xyz1) + (size_t)b * nr * 3;
  float* dist = (fwd ? dist1 : dist2) + (size_t)b * nq;     // no __restrict__: phase B re-reads what phase A wrote
  int* idx = (fwd ? idx1 : idx2) + (size_t)b * nq;
  const int tid = threadIdx.x;

  // the sweep keeps only e = -2 q (exact scaling); the recovery gets q back as -0.5 e
  float ex[Q], ey[Q], ez[Q], best[Q], sec[Q], third[Q];
  int bt[Q], bt2[Q];
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    const int qi = q0 + k * kBlock + tid;
    const bool ok = qi < nq;
    ex[k] = ok ? -2.0f * q[qi * 3 + 0] : 0.f;
    ey[k] = ok ? -2.0f * q[qi * 3 + 1] : 0.f;
    ez[k] = ok ? -2.0f * q[qi * 3 + 2] : 0.f;
    best[k] = sec[k] = third[k] = INFINITY;
    bt[k] = bt2[k] = 0;
  }
  if (tid == 0) {
    s_rmax = 0u;
    s_n2 = 0;
    s_n3 = 0;
  }

  for (int r0 = 0; r0 < nr; r0 += kRefTile) {
    const int cnt = min(kRefTile, nr - r0);
    const int ntile = (cnt + kSub - 1) / kSub;
    __syncthreads();
    float lmax = 0.f;
    for (int j = tid; j < ntile * kSub; j += kBlock) {
      float4 v = make_float4(0.f, 0.f, 0.f, INFINITY);   // padding: e = +inf (or NaN), never wins
      if (j < cnt) {
        const float* p = r + (size_t)(r0 + j) * 3;
        v = make_float4(p[0], p[1], p[2], metric_sqdist<0>(p[0], p[1], p[2]));
        lmax = fmaxf(lmax, v.w);
      }
      s_ref[j] = v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o, kWave));
    if ((tid & 63) == 0) atomicMax(&s_rmax, __float_as_uint(lmax));
    __syncthreads();
#ifndef HOUV_CHAMFER_COMPILED_SWEEP
    if constexpr (Q == 8) {
      filter_sweep8((unsigned)(size_t)(const __attribute__((address_space(3))) float4*)s_ref, ntile, r0 / kSub, ex, ey, ez, best, sec, third, bt, bt2);
      continue;
    }
#endif
    for (int t = 0; t < ntile; ++t) {
      float tm[Q];
#pragma unroll
      for (int k = 0; k < Q; ++k) tm[k] = INFINITY;
      const float4* rp = s_ref + t * kSub;
      // broadcast reads, software-pipelined: the next four references are in flight while four are evaluated
      constexpr int kB = 4;
      float4 ra[kB], rb[kB];
      auto eval = [&](const float4 (&rr)[kB]) {
#pragma unroll
        for (int u = 0; u < kB; u += 2) {
          const float4 a = rr[u], c = rr[u + 1];
#pragma unroll
          for (int k = 0; k < Q; ++k) {
            const float e0 = __builtin_fmaf(ex[k], a.x, __builtin_fmaf(ey[k], a.y, __builtin_fmaf(ez[k], a.z, a.w)));
            const float e1 = __builtin_fmaf(ex[k], c.x, __builtin_fmaf(ey[k], c.y, __builtin_fmaf(ez[k], c.z, c.w)));
            tm[k] = min3f(tm[k], e0, e1);
          }
        }
      };
#pragma unroll
      for (int u = 0; u < kB; ++u) ra[u] = rp[u];
#pragma unroll
      for (int j0 = 0; j0 < kSub; j0 += 2 * kB) {
#pragma unroll
        for (int u = 0; u < kB; ++u) rb[u] = rp[j0 + kB + u];
        eval(ra);
        if (j0 + 2 * kB < kSub) {
#pragma unroll
          for (int u = 0; u < kB; ++u) ra[u] = rp[j0 + 2 * kB + u];
        }
        eval(rb);
      }
      const int tile = r0 / kSub + t;
#pragma unroll
      for (int k = 0; k < Q; ++k) {
        const float v = tm[k];
        const bool lt1 = v < best[k], lt2 = v < sec[k];
        third[k] = __builtin_amdgcn_fmed3f(v, sec[k], third[k]);   // sec <= third: the middle one is the new third smallest
        sec[k] = __builtin_amdgcn_fmed3f(v, best[k], sec[k]);      // best <= sec: likewise
        int s2 = lt2 ? tile : bt2[k];
        asm volatile("" : "+v"(s2));                               // keep straight-line selects (a nested ?: becomes branches)
        bt2[k] = lt1 ? bt[k] : s2;
        bt[k] = lt1 ? tile : bt[k];
        best[k] = lt1 ? v : best[k];
      }
    }
  }

  // ---- exact recovery ------------------------------------------------------------------------------------------
  // Phase A (dense, every lane): the best sub-tile -> provisional (dist, idx) in global memory.  Queries that owe more work
  // -- an ambiguous best sub-tile, a second / third sub-tile within tau -- are NOT finished in place (a wave would run the
  // extra work for all 64 lanes whenever one lane asks for it) but appended to two LDS work lists.
  // Phase B (after a barrier): the lists are worked off densely: one lane per "one more sub-tile" entry, 8 to 64 lanes per
  // "exact full scan" entry.  Lists that overflow (pathological inputs: everything tied) spill to the in-lane path.
  const float rmax = sqrtf(__uint_as_float(s_rmax));   // visible: written before the last tile's barrier
  const bool resident = nr <= kRefTile;                // single LDS pass: every sub-tile is still in s_ref
  const int rot = tid & (kSub - 1);
  unsigned ovf2 = 0u, ovf3 = 0u, ovfa = 0u;   // list overflows; ovfa: the owed sub-tile is the best one, not the second
#pragma unroll 1
  for (int k = 0; k < Q; ++k) {
    const int ql = k * kBlock + tid, qi = q0 + ql;
    const bool valid = qi < nq;
    const bool finite = best[k] < INFINITY;
    const float qx = -0.5f * ex[k], qy = -0.5f * ey[k], qz = -0.5f * ez[k];
    const float qn = sqrtf(metric_sqdist<0>(qx, qy, qz));
    const float tau = 25.0f * kUlpHalf * (rmax + qn) * (rmax + qn) + 1e-30f;
    // !finite: the FILTER saw no finite value -- for NaN / Inf inputs, but also for finite coordinates beyond ~1.3e19, where
    // |r|^2 overflows and e = -inf + inf = NaN for every reference while a direct difference (r - q)^2 may still be finite:
    // such a query takes the exact full scan (which leaves the provisional reference-0 answer of chamfer3D.cu:37 in place
    // when no distance is finite).
    const bool need3 = valid && (!finite || !(third[k] > best[k] + tau));        // written so that NaN / inf thresholds say "needed"
    const bool need2 = valid && finite && !need3 && !(sec[k] > best[k] + tau);
    float bd = INFINITY;
    int jb = 0x7fffffff;
    // The exact arg-min of the best sub-tile is one of its references with e <= min e + tau (the bound above, applied to the
    // sub-tile; every reference tying with it in d is one too).  Counting them costs 3 FMAs + 3 half-rate instructions per
    // reference against ~12 for the exact scan with its index tie-break; nearly always there is exactly one, whose exact
    // distance is then the answer.  Any other count (ties, lattices, non-finite thresholds) still owes the exact scan of the
    // sub-tile: the query joins the "one more sub-tile" work list of phase B.1 with its BEST sub-tile.
    int cand = 0, jc = 0;
    bool amb = false;                      // ambiguous: the exact scan of the best sub-tile is still owed (work list below)
    if (resident) {
      const float thr = best[k] + tau;
      const int base = bt[k] * kSub;
#pragma unroll 8
      for (int jr = 0; jr < kSub; ++jr) {
        const int jj = base + ((jr + rot) & (kSub - 1));   // rotated: lanes sit on different bank quads whatever their tile
        const float4 p = s_ref[jj];
        const bool c = filter_e(ex[k], ey[k], ez[k], p.x, p.y, p.z, p.w) <= thr;   // false for NaN e, for padding (e = +inf)
        jc = c ? jj : jc;
        cand += c ? 1 : 0;
      }
      const float4 p = s_ref[jc];          // jc = 0 without a candidate: still a reference, its exact distance a valid provisional
      bd = metric_sqdist<0>(p.x - qx, p.y - qy, p.z - qz);
      jb = jc;
      amb = valid && finite && cand != 1;
    } else {
      exact_tile<false>(s_ref, r, bt[k], nr, qx, qy, qz, rot, bd, jb);
    }
    if (!finite) {
      // no finite distance at all (NaN / overflowing input): the reference reports ref 0 (chamfer3D.cu:37)
      bd = metric_sqdist<0>(r[0] - qx, r[1] - qy, r[2] - qz);
      jb = 0;
    }
    if (valid) {
      dist[qi] = bd;
      idx[qi] = jb;
    }
    if (need3 || (need2 && amb)) {         // two sub-tiles owed to one query would race on its result: take the full scan
      const int slot = atomicAdd(&s_n3, 1);
      if (slot < kList3) s_list3[slot] = (unsigned short)ql;
      else ovf3 |= 1u << k;
    } else if (need2 || amb) {
      const int slot = atomicAdd(&s_n2, 1);
      if (slot < kList2) s_list2[slot] = (unsigned)ql | ((unsigned)(need2 ? bt2[k] : bt[k]) << 11);
      else { ovf2 |= 1u << k; ovfa |= amb ? 1u << k : 0u; }
    }
  }
  __syncthreads();   // lists complete; the provisional results are visible to the whole workgroup

  // Phase B.1: one more sub-tile (the second one, or the best one of an ambiguous query), one lane per entry
  const int n2 = min(s_n2, kList2);
#pragma unroll 1
  for (int base = 0; base < n2; base += kBlock) {
    const int i = base + tid;
    const bool act = i < n2;
    const unsigned e = act ? s_list2[i] : 0u;
    const int qi = q0 + (int)(e & 2047u), t2 = (int)(e >> 11);
    const float qx = act ? q[qi * 3 + 0] : 0.f, qy = act ? q[qi * 3 + 1] : 0.f, qz = act ? q[qi * 3 + 2] : 0.f;
    const float bd = act ? dist[qi] : 0.f;
    const int jb = act ? idx[qi] : 0;
    float d2 = INFINITY;
    int j2 = 0x7fffffff;
    if (resident) exact_tile<true>(s_ref, r, t2, nr, qx, qy, qz, rot, d2, j2);
    else exact_tile<false>(s_ref, r, t2, nr, qx, qy, qz, rot, d2, j2);
    if (act && (d2 < bd || (d2 == bd && j2 < jb))) {
      dist[qi] = d2;
      idx[qi] = j2;
    }
  }
  // Phase B.2: exact scan of every reference, `lpe` lanes per entry -- a whole wave when the list holds at most four
  // queries (the usual case: ~0.1 % of the queries), eight when it is long.  Lane s of an entry takes references s, s+lpe, ...
  // in ascending order with strict <, then the partial results are merged: smaller distance, lower index on ties.
  const int n3 = min(s_n3, kList3);
  const int lsh = n3 <= 4 ? 6 : (n3 <= 8 ? 5 : (n3 <= 16 ? 4 : 3)), lpe = 1 << lsh;
#pragma unroll 1
  for (int base = 0; base < n3; base += kBlock >> lsh) {
    const int i = base + (tid >> lsh), sub = tid & (lpe - 1);
    const bool act = i < n3;
    if (!__any(act)) continue;                       // waves without an entry in this pass
    const int qi = q0 + (act ? (int)s_list3[i] : 0);
    const float qx = act ? q[qi * 3 + 0] : 0.f, qy = act ? q[qi * 3 + 1] : 0.f, qz = act ? q[qi * 3 + 2] : 0.f;
    float d3 = INFINITY;
    int j3 = 0x7fffffff;
    if (resident) {
#pragma unroll 4
      for (int j = sub; j < nr; j += lpe) {
        const float4 p = s_ref[j];
        const float d = metric_sqdist<0>(p.x - qx, p.y - qy, p.z - qz);
        const bool lt = d < d3;
        d3 = lt ? d : d3;
        j3 = lt ? j : j3;
      }
    } else {
#pragma unroll 4
      for (int j = sub; j < nr; j += lpe) {
        const float d = metric_sqdist<0>(r[j * 3 + 0] - qx, r[j * 3 + 1] - qy, r[j * 3 + 2] - qz);
        const bool lt = d < d3;
        d3 = lt ? d : d3;
        j3 = lt ? j : j3;
      }
    }
    for (int o = 1; o < lpe; o <<= 1) {
      const float dq = __shfl_xor(d3, o, kWave);
      const int jo = __shfl_xor(j3, o, kWave);
      const bool take = dq < d3 || (dq == d3 && jo < j3);
      d3 = take ? dq : d3;
      j3 = take ? jo : j3;
    }
    if (act && sub == 0 && j3 != 0x7fffffff) {
      dist[qi] = d3;
      idx[qi] = j3;
    }
  }
  // list overflow (more than kList2 / kList3 uncertain queries in one workgroup): finish those in place
  if (__any((ovf2 | ovf3) != 0u)) {
#pragma unroll 1
    for (int k = 0; k < Q; ++k) {
      const bool o2 = (ovf2 >> k) & 1u, o3 = (ovf3 >> k) & 1u;
      if (!__any(o2 || o3)) continue;
      const int qi = q0 + k * kBlock + tid;
      const float qx = -0.5f * ex[k], qy = -0.5f * ey[k], qz = -0.5f * ez[k];
      float bd = (o2 || o3) ? dist[qi] : 0.f;
      int jb = (o2 || o3) ? idx[qi] : 0;
      if (__any(o2)) {
        float d2 = INFINITY;
        int j2 = 0x7fffffff;
        const int t2 = ((ovfa >> k) & 1u) ? bt[k] : bt2[k];
        if (resident) exact_tile<true>(s_ref, r, t2, nr, qx, qy, qz, rot, d2, j2);
        else exact_tile<false>(s_ref, r, t2, nr, qx, qy, qz, rot, d2, j2);
        const bool take = o2 && (d2 < bd || (d2 == bd && j2 < jb));
        bd = take ? d2 : bd;
        jb = take ? j2 : jb;
      }
      if (__any(o3)) {
        float d3 = INFINITY;
        int j3 = 0x7fffffff;
#pragma unroll 4
        for (int j = 0; j < nr; ++j) {
          const float d = metric_sqdist<0>(r[j * 3 + 0] - qx, r[j * 3 + 1] - qy, r[j * 3 + 2] - qz);
          const bool lt = d < d3;
          d3 = lt ? d : d3;
          j3 = lt ? j : j3;
        }
        const bool take = o3 && j3 != 0x7fffffff;
        bd = take ? d3 : bd;
        jb = take ? j3 : jb;
      }
      if (o2 || o3) {
        dist[qi] = bd;
        idx[qi] = jb;
      }
    }
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
  const bool direct = g_debug.chamfer_direct.load() != 0;   // houv_debug_set: A/B diagnostics only
  const int qmax = g_debug.chamfer_q.load();
  int q = mx <= kBlock ? 1 : (mx <= 2 * kBlock ? 2 : (mx <= 4 * kBlock || direct ? 4 : 8));
  if (q > qmax) q = qmax;
  const int nqb = (mx + kBlock * q - 1) / (kBlock * q);
  if ((long long)B * nqb > 0x7fffffffLL) {
    set_error("houv_chamfer_forward: batch too large");
    return 0;
  }
  dim3 grid((unsigned)(B * nqb), 2, 1);
  // HOUV_CHAMFER_DIRECT=1 (diagnostics / A-B only): the direct-difference sweep without the expanded-form filter
  if (direct) {
    if (q == 1) chamfer_nn_kernel<1><<<grid, kBlock, 0, s>>>(xyz1, xyz2, N, M, nqb, dist1, dist2, idx1, idx2);
    else if (q == 2) chamfer_nn_kernel<2><<<grid, kBlock, 0, s>>>(xyz1, xyz2, N, M, nqb, dist1, dist2, idx1, idx2);
    else chamfer_nn_kernel<4><<<grid, kBlock, 0, s>>>(xyz1, xyz2, N, M, nqb, dist1, dist2, idx1, idx2);
  } else {
    if (q == 1) chamfer_nn_filter_kernel<1><<<grid, kBlock, 0, s>>>(xyz1, xyz2, N, M, nqb, dist1, dist2, idx1, idx2);
    else if (q == 2) chamfer_nn_filter_kernel<2><<<grid, kBlock, 0, s>>>(xyz1, xyz2, N, M, nqb, dist1, dist2, idx1, idx2);
    else if (q == 4) chamfer_nn_filter_kernel<4><<<grid, kBlock, 0, s>>>(xyz1, xyz2, N, M, nqb, dist1, dist2, idx1, idx2);
    else chamfer_nn_filter_kernel<8><<<grid, kBlock, 0, s>>>(xyz1, xyz2, N, M, nqb, dist1, dist2, idx1, idx2);
  }
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

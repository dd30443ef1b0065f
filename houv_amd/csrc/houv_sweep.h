// houv_sweep.h -- the LDS-broadcast brute-force nearest-neighbour sweep shared by the fused HOUV loop (solve.hip)
// and the ICP refinement kernel (icp.hip), plus the workgroup reduction they both use.
#pragma once
#include "houv_common.h"

namespace houv {

constexpr int kAccStride = 16;   // row stride (floats) of the per-wave reduction scratch

typedef float houv_f4v __attribute__((ext_vector_type(4)));
typedef const houv_f4v __attribute__((address_space(3))) * lds_f4;   // LDS pointer usable from a 32-bit byte address

// ------------------------------------------------------------------------------------------------
// The brute-force sweep: for each of this lane's Q queries, min over all references of the NMET
// squared distances, plus the id of the 16-reference tracking unit (kTrk) that produced each minimum.
// ------------------------------------------------------------------------------------------------
// One sub-tile of the sweep: out = min(in, the sub-tile's distances), per query and metric.  `in` and `out` are DIFFERENT
// register sets (the first v_min3 of a chain is three-address: out = min3(in, a, c)), so the caller can compare out against in
// afterwards without having saved a copy.
template <int Q, int NMET>
__device__ __forceinline__ void sweep_tile(const float4* __restrict__ rp, const float (&qx)[Q], const float (&qy)[Q],
                                           const float (&qz)[Q], const float (&in)[Q][NMET], float (&out)[Q][NMET]) {
#pragma unroll
  for (int j = 0; j < kTrk; j += 2) {
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
        out[k][0] = min3f(j == 0 ? in[k][0] : out[k][0], a0, c0);
        out[k][1] = min3f(j == 0 ? in[k][1] : out[k][1], a1, c1);
        out[k][2] = min3f(j == 0 ? in[k][2] : out[k][2], a2, c2);
        out[k][3] = min3f(j == 0 ? in[k][3] : out[k][3], a3, c3);
      } else {
        out[k][0] = min3f(j == 0 ? in[k][0] : out[k][0], metric_sqdist<0>(ax, ay, az), metric_sqdist<0>(cx, cy, cz));
      }
    }
  }
}

template <int Q, int NMET>
__device__ __forceinline__ void sweep(const float4* __restrict__ refs, int ntile, const float (&qx)[Q],
                                      const float (&qy)[Q], const float (&qz)[Q], float (&best)[Q][NMET],
                                      int (&btile)[Q][NMET]) {
  // The running minimum is threaded through the sub-tiles, ping-ponging between two register sets: per query, metric and
  // sub-tile the bookkeeping is one v_cmp + one v_cndmask ("did this sub-tile lower the minimum?"; strict <: the earlier
  // sub-tile keeps ties, so the lowest index wins) instead of v_cmp + two v_cndmask -- all half-rate instructions.
  float other[Q][NMET];
#pragma unroll
  for (int k = 0; k < Q; ++k)
#pragma unroll
    for (int m = 0; m < NMET; ++m) {
      best[k][m] = INFINITY;
      btile[k][m] = 0;
    }
  int t = 0;
  for (; t + 1 < ntile; t += 2) {
    sweep_tile<Q, NMET>(refs + t * kTrk, qx, qy, qz, best, other);
#pragma unroll
    for (int k = 0; k < Q; ++k)
#pragma unroll
      for (int m = 0; m < NMET; ++m) btile[k][m] = (other[k][m] < best[k][m]) ? t : btile[k][m];
    sweep_tile<Q, NMET>(refs + (t + 1) * kTrk, qx, qy, qz, other, best);
#pragma unroll
    for (int k = 0; k < Q; ++k)
#pragma unroll
      for (int m = 0; m < NMET; ++m) btile[k][m] = (best[k][m] < other[k][m]) ? t + 1 : btile[k][m];
  }
  if (t < ntile) {   // odd number of sub-tiles
    sweep_tile<Q, NMET>(refs + t * kTrk, qx, qy, qz, best, other);
#pragma unroll
    for (int k = 0; k < Q; ++k)
#pragma unroll
      for (int m = 0; m < NMET; ++m) {
        btile[k][m] = (other[k][m] < best[k][m]) ? t : btile[k][m];
        best[k][m] = other[k][m];
      }
  }
}

// Single-metric form of sweep() for metric MET (used by the fused loop's mis-prediction repair): the same expression
// tree per metric (metric_sqdist<MET> == the a0..a3 of the fused form), the same min3 / strict-< bookkeeping, hence
// bit-identical (best, btile) for that metric.
template <int Q, int MET>
__device__ __forceinline__ void sweep_one(const float4* __restrict__ refs, int ntile, const float (&qx)[Q],
                                          const float (&qy)[Q], const float (&qz)[Q], float (&best)[Q], int (&btile)[Q]) {
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    best[k] = INFINITY;
    btile[k] = 0;
  }
  for (int t = 0; t < ntile; ++t) {
    float tm[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) tm[k] = INFINITY;
    const float4* rp = refs + t * kTrk;
#pragma unroll 4
    for (int j = 0; j < kTrk; j += 2) {
      const float4 a = rp[j], c = rp[j + 1];
      asm volatile("" ::"v"(a.w), "v"(c.w));
#pragma unroll
      for (int k = 0; k < Q; ++k)
        tm[k] = min3f(tm[k], metric_sqdist<MET>(a.x - qx[k], a.y - qy[k], a.z - qz[k]),
                      metric_sqdist<MET>(c.x - qx[k], c.y - qy[k], c.z - qz[k]));
    }
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      const bool lt = tm[k] < best[k];
      best[k] = lt ? tm[k] : best[k];
      btile[k] = lt ? t : btile[k];
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


// acc = 2 * acc + bit in ONE instruction: v_addc_co_u32 adds the lane's bit of a 64-bit lane mask (the compare's SGPR pair) as
// carry-in.  Replaces v_cndmask + v_or (+ v_min) wherever per-lane bits are collected: box tests, rescans.
__device__ __forceinline__ unsigned shift_in_mask(unsigned acc, unsigned long long lanes) {
  unsigned long long carry_out;
  asm("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(acc), "=s"(carry_out) : "s"(lanes));
  return acc;
}
__device__ __forceinline__ unsigned shift_in(unsigned acc, bool bit) { return shift_in_mask(acc, __builtin_amdgcn_ballot_w64(bit)); }

// Exact NN recovery for one query: re-evaluate the winning 16-reference tracking unit (kTrk; half a sub-tile -- the sweeps and the
// pruned walks track the arg-min at this granularity since the end of round 3: half the re-evaluations) with the bit-identical
// expression and return the lowest matching reference.  The scan order is rotated per lane: units are 256 B apart, so an
// un-rotated scan puts all lanes of a ds_read_b128 group on the same LDS bank quad.
// XOR256 (the cloud is 256-B aligned in LDS): lane l reads slot (i ^ (l & 15)) -- 16 distinct slots in each of the instruction's
// 16-lane groups -- and collects the matches as one bit per read (shift_in); the lowest matching index is decoded after the scan
// (one match unless two references tie exactly).  Otherwise: (j + rot) mod 16, min-tracked.
template <int MET, int BATCH, bool XOR256 = false>
__device__ __forceinline__ float4 recover_nn(const float4* __restrict__ rp, float qx, float qy, float qz, float bd, int rot,
                                             int& j_out) {
  int jb = kTrk;
  static_assert(kTrk == 16 && kTrk % BATCH == 0, "one 256-B unit of 16 slots");
  if constexpr (XOR256) {
    const unsigned r16 = (unsigned)rot & 15u;
    const unsigned xa = (unsigned)(size_t)(lds_f4)rp + (r16 << 4);
    unsigned match = 0u;                          // bit (31 - e) = read e matched
#pragma unroll 1
    for (int c = 0; c < kTrk; c += BATCH) {
      float4 r[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const houv_f4v v = *(lds_f4)(size_t)(xa ^ ((unsigned)(c + u) << 4));
        r[u] = make_float4(v.x, v.y, v.z, v.w);
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u) asm volatile("" ::"v"(r[u].x), "v"(r[u].y), "v"(r[u].z), "v"(r[u].w));   // keep b128
#pragma unroll
      for (int u = 0; u < BATCH; ++u)
        match = shift_in(match, metric_sqdist<MET>(r[u].x - qx, r[u].y - qy, r[u].z - qz) == bd);
    }
    match <<= 32 - kTrk;                          // read e at bit 31 - e
    while (match != 0u) {                         // one round unless references tie (or none: not a point)
      const int e = __clz((int)match);
      match &= ~(0x80000000u >> e);
      jb = min(jb, (int)((unsigned)e ^ r16));
    }
  } else {
#pragma unroll 1
    for (int c = 0; c < kTrk; c += BATCH) {
      float4 r[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) r[u] = rp[(c + u + rot) & (kTrk - 1)];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) asm volatile("" ::"v"(r[u].x), "v"(r[u].y), "v"(r[u].z), "v"(r[u].w));   // keep b128
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const float d = metric_sqdist<MET>(r[u].x - qx, r[u].y - qy, r[u].z - qz);
        jb = min(jb, (d == bd) ? ((c + u + rot) & (kTrk - 1)) : kTrk);   // lowest matching index, whatever the order
      }
    }
  }
  j_out = jb & (kTrk - 1);
  return rp[j_out];
}

// ---------------------------------------------------------------------------------------------------------------
// EXACT pruned sweep (houv_solve_iterate_pruned; this first form, walked by the OWNING lanes, was round 2's and is kept for A/B --
// the product runs pruned_sweep_sorted further down).  References are grouped in the same 32-point sub-tiles as
// the brute-force sweep; every sub-tile carries an axis-aligned bounding box.  For every query and metric m the
// distance to the point that was its NN in the previous iteration is an upper bound ub[m] that is attained; a sub-tile
// whose box is farther from the query than ub[m] for every metric cannot contain any of its NNs -- nor a point tying
// with one -- and is skipped.  The surviving sub-tiles are visited in ascending order with the same min3 / strict-<
// bookkeeping as sweep(), so (best, btile) come out BIT-IDENTICAL to the brute-force sweep.
//   * per-lane sub-tile lists are 64-bit masks (<= 64 sub-tiles, i.e. clouds of <= 2048 points);
//   * lanes walk their own lists: LDS gathers, scan order XOR-rotated per lane (conflict-free, one v_xor per read),
//     reads software-pipelined in batches of 4;
//   * a lane's Q lists are walked back to back inside ONE loop, so a wave runs for the max over lanes of the SUMMED
//     list lengths (measured 60 steps for 48.8 asked at 2048^2 with views) instead of the sum of per-list maxima (70).
// Measured alternatives (profiles/r01_pruned_variants.txt): G neighbouring queries of a lane sharing one list and every
// gathered reference (HOUV_PRUNE_OWN = HOUV_PRUNE_GROUP = 2 or 4) does fewer, fatter steps but neighbours' lists are
// correlated, which costs more in lane imbalance than the shared reads save: 1.05 / 1.19 vs 1.04 us per hypothesis-
// iteration.  The walk is ~60 % of the pruned iteration, the box tests ~10 %, the bounds ~2 %.
// ---------------------------------------------------------------------------------------------------------------
#ifdef HOUV_STAMPS
// one record per workgroup (modulo kStampWgs), summed on the host: same-address global atomics from every wave at every stamp
// serialise in L2 (the stamped build ran 2.7x slower than the product, distorting what it measures), and an LDS array would cost the
// <512,4> variant its second workgroup per CU.  [0..15] solve.hip's phases, [16..23]: asked, steps, waves, cycles of the sweep's phases
constexpr int kStampWgs = 4096;
__device__ unsigned long long g_stamp_wg[kStampWgs * 24];
#define HOUV_STAMP_ADD(i, v) atomicAdd(&g_stamp_wg[(blockIdx.x % kStampWgs) * 24 + (i)], (unsigned long long)(v))
#define HOUV_PSTAT(i, v) HOUV_STAMP_ADD(16 + (i), v)
#define HOUV_PSTAMP(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); if ((threadIdx.x & 63) == 0) HOUV_PSTAT(i, n_ - pst_); pst_ = n_; } while (0)
#else
#define HOUV_PSTAMP(i) do {} while (0)
#endif

// point owned by (thread, k): a lane owns Q/OWN chunks of OWN consecutive points; chunk c of all lanes covers points
// [c*BLOCK*OWN, (c+1)*BLOCK*OWN).  OWN = 1: strided (brute-force default), OWN = Q: Q consecutive points per lane.
template <int BLOCK, int Q, int OWN>
__device__ __forceinline__ int pt_index(int k) {
  static_assert(OWN >= 1 && Q % OWN == 0 && kSub % OWN == 0, "ownership chunk must divide Q and the sub-tile");
  return (k / OWN) * (BLOCK * OWN) + (int)threadIdx.x * OWN + (k % OWN);
}

#ifndef HOUV_PRUNE_GROUP
#define HOUV_PRUNE_GROUP 1
#endif

// Minima of the NMET squared distances between G queries and the 32 references of ONE sub-tile, gathered per lane, SEPARATELY for the
// sub-tile's two tracking units (references 0..15 -> ta, 16..31 -> tb).  The lane's sub-tile starts at LDS byte address
// (xa & ~511); the scan order is rotated per lane by XOR over the 16 slots of a 256-B half -- xa carries (lane & 15) << 4 in bits
// 4..7 -- and both halves are read through one address (the second read is an immediate offset: one v_xor per TWO reads); needs the
// clouds 512-B aligned in LDS (solve.hip aligns the dynamic segment); conflict-free as (lane ^ i) % 16 takes 16 distinct slots in
// every ds_read_b128 lane group.  A min3 takes two references of the SAME half, so the two accumulators cost no instruction more
// than one.  The reads are software-pipelined: while a batch of four references is being evaluated the next batch is in flight
// (ping-pong register sets; the trailing prefetch wraps around and is dropped).  Same expression trees as sweep_tile(); the order
// inside a tracking unit does not matter to a minimum.  ta comes in holding whatever the caller threads through the first unit
// (its running minima, or +inf); tb is set here.
template <int G, int NMET>
__device__ __forceinline__ void gather_tile_min(unsigned xa, const float (&cx)[G], const float (&cy)[G], const float (&cz)[G],
                                                float (&ta)[G][NMET], float (&tb)[G][NMET]) {
#pragma unroll
  for (int k = 0; k < G; ++k)
#pragma unroll
    for (int m = 0; m < NMET; ++m) tb[k][m] = INFINITY;
  constexpr int kBatch = 4, kHalf = kSub / 2;
  static_assert(kSub == 32 && kHalf == kTrk, "two 256-B tracking units of 16 slots");
  auto fetch = [&](float4 (&r)[kBatch], int i0) {
#pragma unroll
    for (int u = 0; u < kBatch; u += 2) {
      lds_f4 p = (lds_f4)(size_t)(xa ^ ((unsigned)((i0 + u / 2) & (kHalf - 1)) << 4));
      const houv_f4v v0 = p[0], v1 = p[kHalf];
      r[u] = make_float4(v0.x, v0.y, v0.z, v0.w);           // first unit
      r[u + 1] = make_float4(v1.x, v1.y, v1.z, v1.w);       // second unit
    }
  };
  auto eval2 = [&](const float4 a, const float4 c, float (&t)[G][NMET]) {   // two references of one unit
#pragma unroll
    for (int k = 0; k < G; ++k) {
      const float ax = a.x - cx[k], ay = a.y - cy[k], az = a.z - cz[k];
      const float bx = c.x - cx[k], by = c.y - cy[k], bz = c.z - cz[k];
      if constexpr (NMET == 4) {
        const float axx = ax * ax, ayy = ay * ay, bxx = bx * bx, byy = by * by;
        const float a3 = __builtin_fmaf(ay, ay, axx), b3 = __builtin_fmaf(by, by, bxx);
        const float a1 = __builtin_fmaf(az, az, ayy), b1 = __builtin_fmaf(bz, bz, byy);
        const float a2 = __builtin_fmaf(az, az, axx), b2 = __builtin_fmaf(bz, bz, bxx);
        const float a0 = __builtin_fmaf(az, az, a3), b0 = __builtin_fmaf(bz, bz, b3);
        t[k][0] = min3f(t[k][0], a0, b0);
        t[k][1] = min3f(t[k][1], a1, b1);
        t[k][2] = min3f(t[k][2], a2, b2);
        t[k][3] = min3f(t[k][3], a3, b3);
      } else {
        t[k][0] = min3f(t[k][0], metric_sqdist<0>(ax, ay, az), metric_sqdist<0>(bx, by, bz));
      }
    }
  };
  auto eval = [&](float4 (&r)[kBatch]) {
#pragma unroll
    for (int u = 0; u < kBatch; ++u) asm volatile("" ::"v"(r[u].x), "v"(r[u].y), "v"(r[u].z), "v"(r[u].w));   // keep b128
    static_assert(kBatch == 4, "a batch = two slots x two units");
    eval2(r[0], r[2], ta);
    eval2(r[1], r[3], tb);
  };
  float4 ra[kBatch], rb[kBatch];
  fetch(ra, 0);
#pragma unroll 1
  for (int i0 = 0; i0 < kHalf; i0 += kBatch) {
    fetch(rb, i0 + kBatch / 2);
    eval(ra);
    fetch(ra, i0 + kBatch);
    eval(rb);
  }
}

// (running minimum, tracking unit) of one query and metric after a sub-tile whose unit minima are ta -- threaded: already
// min(running, first unit) -- and tb (second unit).  Units are taken in ascending order with strict <, as sweep() takes them.
__device__ __forceinline__ void take_units(float ta, float tb, int unit0, float& cb, int& ct) {
  const bool la = ta < cb, lb = tb < ta;
  ct = lb ? unit0 + 1 : (la ? unit0 : ct);
  cb = lb ? tb : ta;
}

// Which sub-tiles each of this lane's queries must visit: the bound per metric is the distance to the point that was the
// query's nearest neighbour in the previous iteration (`prev`, attained), the test a point-to-box distance per metric.
template <int BLOCK, int Q, int NMET, int OWN, int L, int G>
__device__ __forceinline__ void prune_masks(const float4* __restrict__ refs, const float4* __restrict__ boxes, int ntile,
                                            const float (&qx)[Q], const float (&qy)[Q], const float (&qz)[Q],
                                            const short* __restrict__ prev, int prev_stride, int count,
                                            unsigned long long (&un)[L]) {
#ifdef HOUV_STAMPS
  unsigned long long pst_ = __builtin_readcyclecounter();
#endif
  {
    float ub[Q][NMET];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      const int i = pt_index<BLOCK, Q, OWN>(k);
      const bool ok = i < count;
      const int ii = ok ? i : 0;
      { const float4 r = refs[prev[0 * prev_stride + ii]]; ub[k][0] = metric_sqdist<0>(r.x - qx[k], r.y - qy[k], r.z - qz[k]); }
      if constexpr (NMET == 4) {
        { const float4 r = refs[prev[1 * prev_stride + ii]]; ub[k][1] = metric_sqdist<1>(r.x - qx[k], r.y - qy[k], r.z - qz[k]); }
        { const float4 r = refs[prev[2 * prev_stride + ii]]; ub[k][2] = metric_sqdist<2>(r.x - qx[k], r.y - qy[k], r.z - qz[k]); }
        { const float4 r = refs[prev[3 * prev_stride + ii]]; ub[k][3] = metric_sqdist<3>(r.x - qx[k], r.y - qy[k], r.z - qz[k]); }
      }
#pragma unroll
      for (int m = 0; m < NMET; ++m) ub[k][m] = ok ? (ub[k][m] * 1.00001f + 1e-30f) : -1.f;   // box distances are rounded: stay conservative
    }
    unsigned alo[L], ahi[L];
#pragma unroll
    for (int g = 0; g < L; ++g) alo[g] = ahi[g] = 0u;
    HOUV_PSTAMP(3);
    // per query and box (17 instructions; wave-uniform t: the box reads are LDS broadcasts): the box point nearest to the query is
    // the query clamped into the box (v_med3), its offset squared per axis and summed per metric; the verdicts are collected
    // one bit per box by shift_in, so the boxes run in DESCENDING order, 32 per mask word
    auto test = [&](const float4 lo, const float4 hi, unsigned (&acc)[L]) {
#pragma unroll
      for (int g = 0; g < L; ++g) {
        unsigned long long in = 0ull;                     // lane masks of the compares, OR-ed on the scalar unit (no branches)
#pragma unroll
        for (int k = g * G; k < (g + 1) * G; ++k) {
          const float dx = qx[k] - __builtin_amdgcn_fmed3f(qx[k], lo.x, hi.x);
          const float dy = qy[k] - __builtin_amdgcn_fmed3f(qy[k], lo.y, hi.y);
          const float dz = qz[k] - __builtin_amdgcn_fmed3f(qz[k], lo.z, hi.z);
          if constexpr (NMET == 4) {
            const float xx = dx * dx, yy = dy * dy;
            const float s3 = __builtin_fmaf(dy, dy, xx), s1 = __builtin_fmaf(dz, dz, yy), s2 = __builtin_fmaf(dz, dz, xx);
            const float s0 = __builtin_fmaf(dz, dz, s3);
            in |= __builtin_amdgcn_ballot_w64(s0 <= ub[k][0]) | __builtin_amdgcn_ballot_w64(s1 <= ub[k][1]) |
                  __builtin_amdgcn_ballot_w64(s2 <= ub[k][2]) | __builtin_amdgcn_ballot_w64(s3 <= ub[k][3]);
          } else {
            in |= __builtin_amdgcn_ballot_w64(__builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)) <= ub[k][0]);
          }
        }
        acc[g] = shift_in_mask(acc[g], in);
      }
    };
    auto run = [&](int t_first, int t_last, unsigned (&acc)[L]) {      // t_first >= t_last; the next box is in flight while this one is tested
      float4 lo = boxes[2 * t_first], hi = boxes[2 * t_first + 1];
      for (int t = t_first; t >= t_last; --t) {
        const int tn = t > 0 ? t - 1 : 0;
        const float4 nlo = boxes[2 * tn], nhi = boxes[2 * tn + 1];
        test(lo, hi, acc);
        lo = nlo; hi = nhi;
      }
    };
    if (ntile > 32) run(ntile - 1, 32, ahi);
    run((ntile < 32 ? ntile : 32) - 1, 0, alo);
#pragma unroll
    for (int g = 0; g < L; ++g) un[g] = ((unsigned long long)ahi[g] << 32) | alo[g];
  }
  HOUV_PSTAMP(4);
}

template <int BLOCK, int Q, int NMET, int OWN>
__device__ __forceinline__ void pruned_sweep(const float4* __restrict__ refs, const float4* __restrict__ boxes, int ntile,
                                             const float (&qx)[Q], const float (&qy)[Q], const float (&qz)[Q],
                                             const short* __restrict__ prev, int prev_stride, int count, int rot,
                                             float (&best)[Q][NMET], int (&btile)[Q][NMET],
                                             unsigned long long* __restrict__ stats = nullptr, int cap_slack = -1) {
  // G queries share one sub-tile list and every gathered reference (G = 1 by default, see above); a lane's Q/G lists
  // are walked back to back inside ONE loop (a lane moves on to its next list while others are still on their first)
  constexpr int G = (OWN < HOUV_PRUNE_GROUP) ? OWN : HOUV_PRUNE_GROUP;
  constexpr int L = Q / G;
  unsigned long long un[L];
#ifdef HOUV_STAMPS
  unsigned long long pst_ = __builtin_readcyclecounter();
#endif
  prune_masks<BLOCK, Q, NMET, OWN, L, G>(refs, boxes, ntile, qx, qy, qz, prev, prev_stride, count, un);
#pragma unroll
  for (int k = 0; k < Q; ++k)
#pragma unroll
    for (int m = 0; m < NMET; ++m) { best[k][m] = INFINITY; btile[k][m] = 0; }
  int nsteps = 0;   // wave-uniform
  if (stats) {      // selectivity counters for bench.py's executed-work accounting (houv_debug_set("solve_stats", ptr)); wave-uniform branch
    int asked = 0;
#pragma unroll
    for (int g = 0; g < L; ++g) asked += __popcll(un[g]);
    asked = wave_incl_scan_dpp(asked);
    if ((threadIdx.x & 63) == 63) {
      atomicAdd(&stats[0], (unsigned long long)asked);
      atomicAdd(&stats[2], 1ull);
    }
  }
#ifdef HOUV_STAMPS
  {
    int asked = 0;
#pragma unroll
    for (int g = 0; g < L; ++g) asked += __popcll(un[g]);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) asked += __shfl_xor(asked, o, 64);
    // union over the wave's 64 lanes of list g (64 consecutive points when OWN == 1), and over all of the wave's lists
    int uni_g = 0;
    unsigned long long all = 0ull;
#pragma unroll
    for (int g = 0; g < L; ++g) {
      unsigned long long u = un[g];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) u |= ((unsigned long long)__shfl_xor((int)(u >> 32), o, 64) << 32) | (unsigned)__shfl_xor((int)u, o, 64);
      uni_g += __popcll(u);
      all |= u;
    }
    if ((threadIdx.x & 63) == 0) {
      HOUV_PSTAT(0, (unsigned long long)asked);           // sub-tile visits the wave's lanes asked for
      HOUV_PSTAT(2, 1ull);                                // waves
      HOUV_PSTAT(6, (unsigned long long)uni_g);           // sum over lists of the per-list wave unions
      HOUV_PSTAT(7, (unsigned long long)__popcll(all));   // union over the whole wave
    }
  }
  pst_ = __builtin_readcyclecounter();
  unsigned long long steps_ = 0;
#endif
  // ---- capped lock-step passes (round 3) --------------------------------------------------------------------------
  // Pass g: every lane walks ITS list g, all lanes in step, for `cap` steps (wave-uniform).  The query, its running minima
  // and sub-tile ids are then compile-time registers: no "which list is this lane on" selects, and a lane whose list has
  // run out simply re-evaluates its last sub-tile -- harmless, an evaluation can only repeat or exceed what `best` already
  // holds (and a sub-tile that is NOT on a lane's list cannot hold a point at or below its bound), so no activity predicate
  // either: 22 instead of ~110 bookkeeping instructions per step.  Pure lock-step (cap = the wave's longest list) needs
  // 70 steps where the fused loop below needs 60 (lane imbalance); so the passes are CAPPED near the wave's mean list
  // length and what is left of the long lists goes to the fused loop, whose steps cost more but are now few.
  const int slack = cap_slack;
  if (slack >= 0) {
#pragma unroll
    for (int g = 0; g < L; ++g) {
      const int len = __popcll(un[g]);
      const int wsum = __builtin_amdgcn_readlane(wave_incl_scan_dpp(len), 63);
      const int wmax = __builtin_amdgcn_readlane(wave_max_to_lane63(len), 63);
      const int cap = min(wmax, ((wsum + 63) >> 6) + slack);
      float cx[G], cy[G], cz[G];
#pragma unroll
      for (int k = 0; k < G; ++k) { cx[k] = qx[g * G + k]; cy[k] = qy[g * G + k]; cz[k] = qz[g * G + k]; }
      int t = 0;
#pragma unroll 1
      for (int s = 0; s < cap; ++s) {
        const unsigned long long mm = un[g];
        t = (mm != 0ull) ? (__ffsll((long long)mm) - 1) : t;
        un[g] = mm & (mm - 1ull);                              // 0 stays 0
        const unsigned xa = (unsigned)(size_t)(lds_f4)refs + (unsigned)t * (kSub * 16u) + (((unsigned)rot & 15u) << 4);
        float ta[G][NMET], tb[G][NMET];
#pragma unroll
        for (int k = 0; k < G; ++k)
#pragma unroll
          for (int m = 0; m < NMET; ++m) ta[k][m] = best[g * G + k][m];   // the running minima threaded through the first unit
        gather_tile_min<G, NMET>(xa, cx, cy, cz, ta, tb);
#pragma unroll
        for (int k = 0; k < G; ++k)
#pragma unroll
          for (int m = 0; m < NMET; ++m) take_units(ta[k][m], tb[k][m], 2 * t, best[g * G + k][m], btile[g * G + k][m]);
      }
      nsteps += cap;
    }
  }
  // ---- fused loop: whatever the passes left (everything when slack < 0) ---------------------------------------------
  for (;;) {
    // current list of this lane: the first one that still has sub-tiles
    unsigned long long mm = 0ull;
    int cur = 0;
#pragma unroll
    for (int g = L - 1; g >= 0; --g) {
      const bool has = un[g] != 0ull;
      mm = has ? un[g] : mm;
      cur = has ? g : cur;
    }
    if (!__any(mm != 0ull)) break;
    ++nsteps;
#ifdef HOUV_STAMPS
    ++steps_;   // sub-tile steps the wave executed
#endif
    const bool act = mm != 0ull;
    const int t = act ? (__ffsll((long long)mm) - 1) : 0;
    mm = act ? (mm & (mm - 1ull)) : 0ull;
    float cx[G], cy[G], cz[G];
#pragma unroll
    for (int k = 0; k < G; ++k) { cx[k] = qx[k]; cy[k] = qy[k]; cz[k] = qz[k]; }
#pragma unroll
    for (int g = 0; g < L; ++g) {
      un[g] = (cur == g) ? mm : un[g];
      if (g > 0) {
#pragma unroll
        for (int k = 0; k < G; ++k) {
          cx[k] = (cur == g) ? qx[g * G + k] : cx[k];
          cy[k] = (cur == g) ? qy[g * G + k] : cy[k];
          cz[k] = (cur == g) ? qz[g * G + k] : cz[k];
        }
      }
    }
    const unsigned xa = (unsigned)(size_t)(lds_f4)refs + (unsigned)t * (kSub * 16u) + (((unsigned)rot & 15u) << 4);
    float ta[G][NMET], tb[G][NMET];
#pragma unroll
    for (int k = 0; k < G; ++k)
#pragma unroll
      for (int m = 0; m < NMET; ++m) ta[k][m] = INFINITY;               // which list the lane is on is only known per lane: fresh minima
    gather_tile_min<G, NMET>(xa, cx, cy, cz, ta, tb);
#pragma unroll
    for (int g = 0; g < L; ++g)
#pragma unroll
      for (int k = 0; k < G; ++k)
#pragma unroll
        for (int m = 0; m < NMET; ++m) {
          const bool on = act && (cur == g);
          const bool lb = tb[k][m] < ta[k][m];                          // the second unit holds the sub-tile's minimum (strictly)
          const float tmin = lb ? tb[k][m] : ta[k][m];
          const bool lt = on && (tmin < best[g * G + k][m]);
          best[g * G + k][m] = lt ? tmin : best[g * G + k][m];
          btile[g * G + k][m] = lt ? 2 * t + (lb ? 1 : 0) : btile[g * G + k][m];
        }
  }
  if (stats && (threadIdx.x & 63) == 0) atomicAdd(&stats[1], (unsigned long long)nsteps);
#ifdef HOUV_STAMPS
  if ((threadIdx.x & 63) == 0) HOUV_PSTAT(1, steps_);
  HOUV_PSTAMP(5);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Pruned sweep, BALANCED form (round 3; solve_kernel<.., PRUNE = 2>).  The owner-walk above loses ~20 % of its steps to
// lane imbalance (a wave runs for its longest lane) and ~25 % of its instructions to "which of my lists am I on" selects.
// Here the walk is detached from ownership:
//   1. every thread computes the visit masks of ITS queries (prune_masks) and parks them in LDS -- in the .w lanes of the
//      two clouds' float4 slots, which nothing else uses (8 bytes per point index: exactly one 64-bit mask);
//   2. the workgroup's queries are counting-sorted by list length, longest first (65 bins, LDS atomics, 2 bytes per query);
//   3. waves take blocks of 64 consecutive sorted queries from a shared counter (longest blocks first: LPT scheduling); the
//      64 lists of a block have (nearly) the same length, so all lanes walk in lock-step with no idle lanes, the query, its
//      running minima and sub-tile ids in fixed registers (~22 bookkeeping instructions per step instead of ~110); a lane
//      whose list is a step shorter re-evaluates its last sub-tile, which cannot change its result;
//   4. per query the sub-tile ids of the minima go back to the owner through the query's (now consumed) mask slot, the four
//      minima through a per-hypothesis scratch record in global memory (16 bytes per query; L2-resident).
// A query's sub-tiles are still visited in ascending order with strict <, so (best, btile) are bit-identical to sweep().
// LDS cost: 2 bytes per query + 0.5 KB, so two 512-thread workgroups still share a CU (the phases of one hide behind
// the sweeps of the other: a single 1024-thread workgroup per CU measured 14 % slower on the brute-force kernel).
// ---------------------------------------------------------------------------------------------------------------
struct SortedStage {
  unsigned short* order;      // [BLOCK * Q]  query ids, longest list first
  int* hist;                  // [65 + 65 + 2]  bin counts (by 64 - length) | bin bases | next block
};

__device__ __forceinline__ unsigned& w_slot(float4* cloud, int q) { return reinterpret_cast<unsigned*>(cloud + q)[3]; }

// wlo / whi: the two LDS clouds (both hold >= count entries); .w of wlo[q] carries the low half of query q's mask and,
// after the walk, the tracking-unit ids (8 bits each: <= 256 units of 16 references) of its minima; .w of whi[q] the high half.
// TS = 1 (clouds of 2049..4096 points): the masks are over 64 SUPER-tiles of two sub-tiles each (`boxes`, `ntile` count
// super-tiles); a visit evaluates both sub-tiles in ascending order, the minima carry tracking-unit ids (0..255) for the rescans.
template <int BLOCK, int Q, int NMET, int TS = 0>
__device__ __forceinline__ void pruned_sweep_sorted(const float4* __restrict__ refs, const float4* __restrict__ boxes, int ntile,
                                                    const float4* __restrict__ qarr, float4* wlo, float4* whi,
                                                    const float (&qx)[Q], const float (&qy)[Q], const float (&qz)[Q],
                                                    const short* __restrict__ prev, int prev_stride, int count, int rot,
                                                    const SortedStage& st, float4* __restrict__ res, float (&best)[Q][NMET],
                                                    int (&btile)[Q][NMET], unsigned long long* __restrict__ stats) {
  static_assert(BLOCK % 64 == 0 && BLOCK >= 128, "the bin prefix runs on one wave while another resets the block counter");
  const int tid = threadIdx.x, lane = tid & 63;
  unsigned long long un[Q];
  prune_masks<BLOCK, Q, NMET, 1, Q, 1>(refs, boxes, ntile, qx, qy, qz, prev, prev_stride, count, un);
#ifdef HOUV_STAMPS
  unsigned long long pst_ = __builtin_readcyclecounter();   // diagnostic build: [3] bounds, [4] box tests, [5] sort, [6] walk, [7] end barrier
  if ((threadIdx.x & 63) == 0) HOUV_PSTAT(2, 1ull);
#endif
  int len[Q], rnk[Q];
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    const int q = pt_index<BLOCK, Q, 1>(k);
    len[k] = __popcll(un[k]);
    rnk[k] = 0;
    if (q < count) {
      w_slot(wlo, q) = (unsigned)un[k];
      w_slot(whi, q) = (unsigned)(un[k] >> 32);
      rnk[k] = atomicAdd(&st.hist[64 - len[k]], 1);        // place inside its bin (any order: results do not depend on it)
    }
  }
  __syncthreads();
  if (tid < 64) {                                           // exclusive prefix over the 65 bins, then clear them for the next sweep
    const int c = st.hist[tid];
    const int incl = wave_incl_scan_dpp(c);
    st.hist[65 + tid] = incl - c;
    if (tid == 63) st.hist[65 + 64] = incl;                 // bin 64 = empty lists (cannot happen for a valid query; harmless)
    st.hist[tid] = 0;
    if (tid == 0) st.hist[64] = 0;
  } else if (tid == 64) {
    st.hist[130] = 0;                                       // next block
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    const int q = pt_index<BLOCK, Q, 1>(k);
    if (q < count) st.order[st.hist[65 + 64 - len[k]] + rnk[k]] = (unsigned short)q;
  }
  __syncthreads();
  const int nblk = (count + 63) >> 6;
  int asked = 0, nsteps = 0;
  HOUV_PSTAMP(5);
  for (;;) {
    int b = 0;
    if (lane == 0) b = atomicAdd(&st.hist[130], 1);
    b = __builtin_amdgcn_readfirstlane(b);
    if (b >= nblk) break;
    const int e = b * 64 + lane;
    const bool valid = e < count;                                       // the last block may be partly filled
    const int q = st.order[valid ? e : count - 1];
    unsigned long long mm = valid ? (((unsigned long long)w_slot(whi, q) << 32) | w_slot(wlo, q)) : 0ull;
    const int cap = __builtin_amdgcn_readfirstlane(__popcll(mm));      // sorted: lane 0 holds the block's longest list
    const float4 qp = qarr[q];
    float cx[1] = {qp.x}, cy[1] = {qp.y}, cz[1] = {qp.z};
    float cb[NMET];
    int ct[NMET];
#pragma unroll
    for (int m = 0; m < NMET; ++m) { cb[m] = INFINITY; ct[m] = 0; }
    asked += __popcll(mm);
    nsteps += cap;
    int t = 0;
#pragma unroll 1
    for (int s = 0; s < cap; ++s) {
      t = (mm != 0ull) ? (__ffsll((long long)mm) - 1) : t;
      mm &= mm - 1ull;                                                  // 0 stays 0
#pragma unroll
      for (int h = 0; h < (1 << TS); ++h) {
        const int ts = (t << TS) | h;                                   // sub-tile
        const unsigned xa = (unsigned)(size_t)(lds_f4)refs + (unsigned)ts * (kSub * 16u) + (((unsigned)rot & 15u) << 4);
        float ta[1][NMET], tb[1][NMET];                                 // the running minima threaded through the first tracking unit
#pragma unroll
        for (int m = 0; m < NMET; ++m) ta[0][m] = cb[m];
        gather_tile_min<1, NMET>(xa, cx, cy, cz, ta, tb);
#pragma unroll
        for (int m = 0; m < NMET; ++m) take_units(ta[0][m], tb[0][m], 2 * ts, cb[m], ct[m]);
      }
    }
    if (valid) {
      if constexpr (NMET == 4) {
        res[q] = make_float4(cb[0], cb[1], cb[2], cb[3]);
        w_slot(wlo, q) = (unsigned)ct[0] | ((unsigned)ct[1] << 8) | ((unsigned)ct[2] << 16) | ((unsigned)ct[3] << 24);
      } else {
        res[q].x = cb[0];
        w_slot(wlo, q) = (unsigned)ct[0];
      }
    }
  }
  if (stats) {
    asked = wave_incl_scan_dpp(asked);
    if (lane == 63) atomicAdd(&stats[0], (unsigned long long)asked);
    if (lane == 0) {
      atomicAdd(&stats[1], (unsigned long long)nsteps);
      atomicAdd(&stats[2], 1ull);                                       // one wave-sweep = 64 x Q queries, as in the owner walk
    }
  }
  HOUV_PSTAMP(6);
  __syncthreads();   // the workgroup's waves share one L1: its global stores above are visible to its loads below
  HOUV_PSTAMP(7);
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    const int q = pt_index<BLOCK, Q, 1>(k);
    const bool ok = q < count && len[k] > 0;                            // padding queries were never walked
    const int qq = ok ? q : 0;
    const unsigned tl = w_slot(wlo, qq);
    if constexpr (NMET == 4) {
      const float4 r = res[qq];
      best[k][0] = ok ? r.x : INFINITY; best[k][1] = ok ? r.y : INFINITY; best[k][2] = ok ? r.z : INFINITY; best[k][3] = ok ? r.w : INFINITY;
      btile[k][0] = ok ? (int)(tl & 255u) : 0; btile[k][1] = ok ? (int)((tl >> 8) & 255u) : 0;
      btile[k][2] = ok ? (int)((tl >> 16) & 255u) : 0; btile[k][3] = ok ? (int)(tl >> 24) : 0;
    } else {
      const float r = res[qq].x;
      best[k][0] = ok ? r : INFINITY;
      btile[k][0] = ok ? (int)(tl & 255u) : 0;
    }
  }
}

// Axis-aligned boxes of the 32-point sub-tiles of a cloud whose points live in this lane's registers (ownership as
// pt_index): within a chunk a sub-tile spans 32/OWN consecutive lanes x OWN points; an in-lane min/max plus a few
// xor-shuffles reduce it.  box[2t] = lo, box[2t+1] = hi.
// TS = 1: boxes of SUPER-tiles of 64 points (two sub-tiles; a whole wave when OWN == 1) instead of sub-tiles.
template <int BLOCK, int Q, int OWN, int TS = 0>
__device__ __forceinline__ void tile_boxes(const float (&x)[Q], const float (&y)[Q], const float (&z)[Q], int count,
                                           int ntile, float4* __restrict__ box) {
  constexpr int kTile = kSub << TS;
  static_assert(kTile / OWN <= 64, "a tile's owners must sit in one wave");
#pragma unroll
  for (int c = 0; c < Q / OWN; ++c) {
    float lx = INFINITY, ly = INFINITY, lz = INFINITY, hx = -INFINITY, hy = -INFINITY, hz = -INFINITY;
#pragma unroll
    for (int o = 0; o < OWN; ++o) {
      const int k = c * OWN + o;
      if (pt_index<BLOCK, Q, OWN>(k) < count) {
        lx = fminf(lx, x[k]); ly = fminf(ly, y[k]); lz = fminf(lz, z[k]);
        hx = fmaxf(hx, x[k]); hy = fmaxf(hy, y[k]); hz = fmaxf(hz, z[k]);
      }
    }
#pragma unroll
    for (int o = 1; o < kTile / OWN; o <<= 1) {
      lx = fminf(lx, __shfl_xor(lx, o, 64)); ly = fminf(ly, __shfl_xor(ly, o, 64)); lz = fminf(lz, __shfl_xor(lz, o, 64));
      hx = fmaxf(hx, __shfl_xor(hx, o, 64)); hy = fmaxf(hy, __shfl_xor(hy, o, 64)); hz = fmaxf(hz, __shfl_xor(hz, o, 64));
    }
    const int t = pt_index<BLOCK, Q, OWN>(c * OWN) / kTile;
    if (((int)threadIdx.x % (kTile / OWN)) == 0 && t < ntile) {
      box[2 * t] = make_float4(lx, ly, lz, 0.f);
      box[2 * t + 1] = make_float4(hx, hy, hz, 0.f);
    }
  }
}

}  // namespace houv

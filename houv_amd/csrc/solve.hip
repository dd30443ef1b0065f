// solve.hip -- the fused HOUV optimisation loop for gfx950 (MI355X).
//
// One workgroup owns one hypothesis (pair p, restart k) for the whole loop:
//   pose from 8 scalars -> move the source cloud -> 4-metric bidirectional Chamfer (two LDS-resident
//   brute-force sweeps) -> top-k robust loss -> closed-form gradient -> Adam step, `n_iters` times,
// with both clouds resident in LDS and NOTHING but the 24-double state touching HBM in between.
// It replaces the PyTorch loop of predict_model (registration/models/houv.py:106-138; loss :209-222,
// model_utils_completion.py:83-100,157-166) and of getPredict_angle (registration/train_utils.py:359-456),
// which per iteration launches 8 NmDistanceKernel + 8 NmDistanceGradKernel + 8 topk + ~150 small kernels
// on K-fold replicated clouds.
//
// Design notes (DESIGN.md has the long form):
//   * sweep: every lane owns Q query points in registers; reference points are read from LDS with
//     wave-uniform ds_read_b128 (broadcast), two at a time; the four squared distances (full + three
//     axis-dropped views) share dx,dy,dz: 3 sub + 2 mul + 4 fma + 4 min3/2 = 11 VALU ops per point pair;
//   * arg-min is deferred: a per-32-reference sub-tile id is tracked (3 ops per 32 refs) and the exact
//     NN is recovered by re-evaluating that sub-tile with bit-identical arithmetic;
//   * no distance/index arrays are ever materialised: the epilogue of each sweep turns (NN, dist)
//     straight into the 13 sums the parameter gradient needs (sum sqrt d, sum G, sum G p^T);
//   * top-k (k = N/2 for the full metric) = exact 4-pass 8-bit radix select on the fp32 bit patterns
//     of the register-resident distances, LDS histogram;
//   * the un-moved source point needed for sum G p^T in the target->moved direction is R^T(p' - T);
//   * PRUNE = 0 is that brute-force sweep (north_star's formulation).  PRUNE = 2 / 3 -- the product default for clouds of
//     257..2048 / 2049..4096 points -- replace the two sweeps by the EXACT pruned search of houv_sweep.h (remembered-NN
//     bounds + boxes of k-d-leaf sub-tiles + a balanced, sorted-block walk): same (minimum, sub-tile) per query and metric,
//     hence the same bits everywhere downstream, for ~1/8 of the point pairs.
#include <stddef.h>
#include <stdlib.h>
#ifdef HOUV_STAMPS
#include <vector>
#endif

#include "../../include/houv_hip.h"
#include "houv_common.h"
#include "houv_sweep.h"

namespace houv {
namespace {

struct SolveArgs {
  const float* src;
  const float* tgt;
  int P, N, M, K;
  double* state;
  int steps_done, n_iters, angle_base, trans_mode, f64_params, k_full, k_view;
  double lr, beta1, beta2, eps;
  float loss_scale;
  float* out_score;
  float* out_loss;
  float* out_R;
  float* out_T;
  float* out_grad;
  float* out_cd;
  short* nn_ws;      // pruned mode, per hypothesis 16 rows of ws_stride int16: rows [dir*4 + metric] = index of each query's NN
                     // in the last iteration; rows 8..15 = ws_stride float4 of scratch (the balanced walk's minima per query)
  int ws_valid;      //   1: nn_ws holds the NNs of the iteration before this launch's first one
  int ws_stride;
  int pred_mode;     // diagnostics (HOUV_SOLVE_PREDICT): 0 normal; 1 always predict direction B (every A-win takes the
                     // repair path); 2 rescan everything (no skipping: the round-1 epilogue's work)
  int ws_refresh;    // pruned mode: every ws_refresh-th iteration rescans everything (refreshes every remembered NN)
  int cap_slack;     // pruned walk: lock-step passes capped at the wave's mean list length + cap_slack (< 0: fused loop only)
  unsigned long long* stats;   // houv_debug_set("solve_stats", device pointer): [0] sub-tile visits the lanes of the pruned sweeps
                               // asked for, [1] sub-tile steps their waves executed, [2] pruned wave-sweeps, [3] brute wave-sweeps,
                               // [4] shader clocks (s_memtime) and [5] 100-MHz ticks (s_memrealtime) summed over the workgroups'
                               // loops: [4]/[5] x 100 MHz = the clock the chip sustained under THIS kernel's load
};

#ifdef HOUV_STAMPS
// Diagnostic build only (scripts/stamps.sh): per-phase wave-cycle totals, never read by the kernel itself.
#define HOUV_STAMP(i)                                                        \
  do {                                                                       \
    const unsigned long long now_ = __builtin_readcyclecounter();           \
    if ((threadIdx.x & 63) == 0) HOUV_STAMP_ADD(i, now_ - t_stamp_);   \
    t_stamp_ = now_;                                                         \
  } while (0)
#define HOUV_STAMP_PARAM , unsigned long long& t_stamp_
#define HOUV_STAMP_ARG , t_stamp_
#else
#define HOUV_STAMP(i) do {} while (0)
#define HOUV_STAMP_PARAM
#define HOUV_STAMP_ARG
#endif

#ifndef HOUV_RESCAN_BATCH
#define HOUV_RESCAN_BATCH 4
#endif
constexpr int kRescanBatch = HOUV_RESCAN_BATCH;
constexpr int kAccN = 13;      // sum sqrt(d), G[3], (G p^T)[9]

constexpr int kRedStride = 4 * kAccN;   // per-wave partial sums of one direction: [metric][13]
constexpr int kHistBins = 256;          // 8-bit radix digits
constexpr int kHistSets = 3;            // rotating histograms: one barrier per radix pass (see select_smallest)
constexpr int kPoseFloats = 28;         // sizeof(Pose) / 4 rounded up
static_assert(sizeof(Pose) <= kPoseFloats * 4 && offsetof(Pose, T) == 36, "sm.pose[0..11] must be R | T");

struct Smem {
  float4* tgt;     // [Mpad]
  float4* mov;     // [Npad]
  double* state;   // [24]
  double* adam;    // [2][2] step size and sqrt(bias correction 2) of Adam: slot = step parity (see the scalar tail)
  float* pose;     // [kPoseFloats] the whole Pose of the current parameters: R row-major [0..8], T [9..11], backward intermediates
  float* acc;      // [8][kAccStride]   slot = metric*2 + dir
  float* red;      // [2 dirs][NW][kRedStride]
  unsigned* hist;  // [kHistSets][256]
  int* ctl;        // [8 + NW]
  float4* tbox;    // [2*64] lo/hi boxes of the target's 32-point sub-tiles   (pruned mode only)
  float4* mbox;    // [2*64] same for the moved cloud, rebuilt every iteration
  SortedStage st;  // staging of the balanced pruned sweep (PRUNE == 2 only)
};

// prune: 0 brute force, 1 pruned (owner walk), 2 pruned (balanced walk: + staging for block * q queries)
// PRUNE == 3: the balanced walk over 64-point SUPER-tiles (pairs of sub-tiles) for clouds of 2049..4096 points: the clouds are
// padded to multiples of 64 points
__host__ __device__ inline int pad_unit(int prune) { return prune == 3 ? 2 * kSub : kSub; }

__host__ __device__ inline size_t smem_bytes(int N, int M, int block, int prune, int q) {
  const int pu = pad_unit(prune);
  int npad = (N + pu - 1) / pu * pu, mpad = (M + pu - 1) / pu * pu;
  if (prune >= 2) npad = mpad = (npad > mpad ? npad : mpad);   // the balanced walk parks a mask half in EITHER cloud's .w lanes
  const int nw = block / 64;
  const size_t nq = (size_t)block * q;
  return (size_t)(npad + mpad) * 16 + 28 * 8 + kPoseFloats * 4 + 8 * kAccStride * 4 + (size_t)2 * nw * kRedStride * 4 +
         kHistSets * kHistBins * 4 + (8 + nw) * 4 + 64 + (prune ? 2 * 128 * 16 : 0) +
         (prune >= 2 ? nq * 2 + 132 * 4 : 0);
}

// nq = BLOCK * Q for the balanced pruned sweep (PRUNE >= 2), 0 otherwise; pu = pad_unit(PRUNE)
__device__ inline Smem carve(unsigned char* base, int N, int M, int block, int nq, int pu) {
  int npad = (N + pu - 1) / pu * pu, mpad = (M + pu - 1) / pu * pu;
  if (nq) npad = mpad = (npad > mpad ? npad : mpad);
  const int nw = block / 64;
  Smem s;
  s.tgt = reinterpret_cast<float4*>(base);
  s.mov = s.tgt + mpad;
  s.state = reinterpret_cast<double*>(s.mov + npad);
  s.adam = s.state + 24;
  s.pose = reinterpret_cast<float*>(s.adam + 4);
  s.acc = s.pose + kPoseFloats;
  s.red = s.acc + 8 * kAccStride;
  s.hist = reinterpret_cast<unsigned*>(s.red + 2 * nw * kRedStride);
  s.ctl = reinterpret_cast<int*>(s.hist + kHistSets * kHistBins);
  // 16-byte alignment by OFFSET arithmetic on the shared segment (a pointer -> integer -> pointer round trip hides the LDS
  // address space from the compiler: the box reads of the pruned sweep became flat_load_dwordx3 + s_waitcnt vmcnt(0))
  const size_t box_off = ((size_t)(reinterpret_cast<unsigned char*>(s.ctl + 8 + nw) - base) + 15) & ~(size_t)15;
  s.tbox = reinterpret_cast<float4*>(base + box_off);
  s.mbox = s.tbox + 128;
  // balanced pruned sweep only (the pointers are never used otherwise)
  s.st.hist = reinterpret_cast<int*>(s.mbox + 128);
  s.st.order = reinterpret_cast<unsigned short*>(s.st.hist + 132);
  return s;
}

__device__ __forceinline__ void store_pose(float* dst, const Pose& f) {
  const float* src = reinterpret_cast<const float*>(&f);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(Pose) / 4); ++i) dst[i] = src[i];
}
__device__ __forceinline__ void load_pose(Pose& f, const float* src) {
  float* dst = reinterpret_cast<float*>(&f);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(Pose) / 4); ++i) dst[i] = src[i];
}

// Exact selection of the `ksel` smallest of the BLOCK*Q keys (fp32 bit patterns of non-negative
// distances; 0xFFFFFFFF marks "not a point").  4-pass 8-bit radix select on LDS histograms.
// Ties at the threshold are taken in (thread, k) order -- torch.topk leaves tie order unspecified.
//   * ONE barrier per pass: three histograms rotate (`hrot` = the one this pass fills, all-zero on entry).  While pass p
//     fills set hrot, every thread also clears set hrot+1, whose last readers (pass p-2) are all past the barrier of
//     pass p-1; after the barrier EVERY wave scans the 256 bins itself (one ds_read_b128 per lane + a DPP prefix sum),
//     so no broadcast through LDS and no second barrier is needed.
//   * pass 0 (sign + 7 exponent bits) sees a handful of distinct digits: plain LDS atomics would serialise 64 lanes
//     on one address, so the wave counts each digit with a ballot and ONE lane adds the count.
template <int BLOCK, int Q>
__device__ __forceinline__ void select_smallest(const unsigned (&key)[Q], int ksel, unsigned* hist, int* ctl,
                                                bool (&sel)[Q], int& hrot) {
  constexpr int NW = BLOCK / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned prefix = 0u, mask = 0u;
  int remaining = ksel, neq = 0;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    unsigned* h = hist + hrot * kHistBins;
    const int nxt = (hrot == kHistSets - 1) ? 0 : hrot + 1;
    for (int i = tid; i < kHistBins; i += BLOCK) hist[nxt * kHistBins + i] = 0u;
    if (pass == 0) {
#pragma unroll
      for (int k = 0; k < Q; ++k) {
        const unsigned digit = key[k] >> 24;
        unsigned long long todo = __ballot(1);
        while (todo) {                                           // wave-uniform loop over the distinct digits
          const int leader = __ffsll((long long)todo) - 1;
          const unsigned d = (unsigned)__builtin_amdgcn_readlane((int)digit, leader);
          const unsigned long long m = __ballot(digit == d);
          if (lane == leader) atomicAdd(&h[d], (unsigned)__popcll(m));
          todo &= ~m;
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < Q; ++k)
        if ((key[k] & mask) == prefix) atomicAdd(&h[(key[k] >> shift) & 255u], 1u);
    }
    __syncthreads();
    {
      const uint4 hv = *reinterpret_cast<const uint4*>(h + 4 * lane);
      const int hh[4] = {(int)hv.x, (int)hv.y, (int)hv.z, (int)hv.w};
      const int tot = hh[0] + hh[1] + hh[2] + hh[3];
      int c = wave_incl_scan_dpp(tot) - tot;
      int fbin = 0, fc = 0, fn = 0;
      bool found = false;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const bool hit = c < remaining && remaining <= c + hh[b];
        fbin = hit ? 4 * lane + b : fbin;
        fc = hit ? c : fc;
        fn = hit ? hh[b] : fn;
        found = found || hit;
        c += hh[b];
      }
      const int src_lane = __ffsll((long long)__ballot(found)) - 1;   // exactly one lane holds the bin (1 <= remaining <= total)
      const int bin = __builtin_amdgcn_readlane(fbin, src_lane);
      prefix |= (unsigned)bin << shift;
      mask |= 255u << shift;
      remaining -= __builtin_amdgcn_readlane(fc, src_lane);
      neq = __builtin_amdgcn_readlane(fn, src_lane);
    }
    hrot = nxt;
  }
  if (neq == remaining) {
#pragma unroll
    for (int k = 0; k < Q; ++k) sel[k] = key[k] <= prefix;
  } else {
    int e = 0;
#pragma unroll
    for (int k = 0; k < Q; ++k) e += (key[k] == prefix) ? 1 : 0;
    const int incl = wave_incl_scan_dpp(e);
    __syncthreads();
    if (lane == 63) ctl[8 + wave] = incl;
    __syncthreads();
    int rank = incl - e;
    for (int w = 0; w < NW; ++w) rank += (w < wave) ? ctl[8 + w] : 0;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      const bool eq = key[k] == prefix;
      sel[k] = key[k] < prefix || (eq && rank < remaining);
      rank += eq ? 1 : 0;
    }
  }
}

// ---- epilogue building blocks ------------------------------------------------------------------------------------
// Per metric and direction the scalar tail needs   S = sum sqrt(d)   over the selected queries, and -- for the ONE
// direction that wins the min of houv.py:212-221 -- G = sum c, GP = sum c p^T with c = mask * (moved - target) / sqrt(d),
// p = un-moved source point.  S needs only the distances the sweep already holds; G and GP need the identity of the
// nearest neighbour, i.e. the exact rescan of the winning sub-tile (32 references x ~11 instructions per query and
// metric: 3/4 of the epilogue's instruction count).  Since round 2 the rescans run only where the gradient flows.
//   DIR == 1 ("A"): queries are this lane's moved points (count = N), references the target cloud.
//   DIR == 0 ("B"): queries are this lane's target points (count = M), references the moved cloud.
constexpr int kGradN = 12;

// wave-level DPP sum; lane 63 parks the total
__device__ __forceinline__ void park(float v, float* dst) {
  v = wave_sum_to_lane63(v);
  if ((threadIdx.x & 63) == 63) *dst = v;
}

// selections of one direction: bit k of bits[m] = query k of this lane takes part in metric m's mean
template <int BLOCK, int Q, int NMET, int OWN>
__device__ __forceinline__ void select_all(const Smem& sm, const float (&best)[Q][NMET], int count, int k_full, int k_view,
                                           int& hrot, unsigned (&bits)[NMET]) {
  bool valid[Q], sel[Q];
  unsigned key[Q];
#pragma unroll
  for (int m = 0; m < NMET; ++m) {
    const int ksel = (m == 0) ? k_full : k_view;     // the view terms take all points in every caller (k_view == count)
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      valid[k] = pt_index<BLOCK, Q, OWN>(k) < count;
      key[k] = valid[k] ? __float_as_uint(best[k][m]) : 0xFFFFFFFFu;
      sel[k] = valid[k];
    }
    if (ksel < count) select_smallest<BLOCK, Q>(key, ksel, sm.hist, sm.ctl, sel, hrot);
    unsigned b = 0u;
#pragma unroll
    for (int k = 0; k < Q; ++k) b |= sel[k] ? (1u << k) : 0u;
    bits[m] = b;
  }
}

// this lane's share of S for one metric (k order; no finite distance -> NaN like torch's min/topk/sqrt chain)
template <int Q>
__device__ __forceinline__ float lane_sqrt_sum(const float (&bd)[Q], unsigned selbits) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < Q; ++k)
    if ((selbits >> k) & 1u) s += (bd[k] < INFINITY) ? sqrtf(bd[k]) : NAN;
  return s;
}

// this lane's share of G[3], GP[9] for one metric: exact NN recovery + products, per query (nothing is kept per query)
template <int BLOCK, int Q, int MET, int DIR, int OWN>
__device__ __forceinline__ void lane_grad_sums(const Smem& sm, const float4* __restrict__ refs, const float (&qx)[Q],
                                               const float (&qy)[Q], const float (&qz)[Q], const float (&bd)[Q],
                                               const int (&bt)[Q], unsigned selbits, int count, const float (&px)[Q],
                                               const float (&py)[Q], const float (&pz)[Q], float (&g)[kGradN], short* ws) {
  const int rot = threadIdx.x & (kSub - 1);
#pragma unroll
  for (int i = 0; i < kGradN; ++i) g[i] = 0.f;
  float R[9], T[3];
  if constexpr (DIR == 0) {
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = sm.pose[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) T[i] = sm.pose[9 + i];
  }
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    int jn;
    const float4 nn = recover_nn<MET, kRescanBatch, true>(refs + bt[k] * kTrk, qx[k], qy[k], qz[k], bd[k], rot, jn);
    if (ws && pt_index<BLOCK, Q, OWN>(k) < count) ws[pt_index<BLOCK, Q, OWN>(k)] = (short)(bt[k] * kTrk + jn);
    if ((selbits >> k) & 1u) {
      const float s = (bd[k] < INFINITY) ? sqrtf(bd[k]) : NAN;
      const float inv = 1.0f / s;   // d == 0 -> inf, and 0*inf = NaN below, as torch's sqrt backward gives
      float dx, dy, dz, sx, sy, sz;
      if constexpr (DIR == 1) {
        dx = qx[k] - nn.x; dy = qy[k] - nn.y; dz = qz[k] - nn.z;
        sx = px[k]; sy = py[k]; sz = pz[k];
      } else {
        dx = nn.x - qx[k]; dy = nn.y - qy[k]; dz = nn.z - qz[k];
        const float ux = nn.x - T[0], uy = nn.y - T[1], uz = nn.z - T[2];
        sx = R[0] * ux + R[3] * uy + R[6] * uz;   // R^T (p' - T)
        sy = R[1] * ux + R[4] * uy + R[7] * uz;
        sz = R[2] * ux + R[5] * uy + R[8] * uz;
      }
      if constexpr (MET == 1) dx = 0.f;
      if constexpr (MET == 2) dy = 0.f;
      if constexpr (MET == 3) dz = 0.f;
      const float cx = dx * inv, cy = dy * inv, cz = dz * inv;
      g[0] += cx; g[1] += cy; g[2] += cz;
      g[3] += cx * sx; g[4] += cx * sy; g[5] += cx * sz;
      g[6] += cy * sx; g[7] += cy * sy; g[8] += cy * sz;
      g[9] += cz * sx; g[10] += cz * sy; g[11] += cz * sz;
    }
  }
}

// S of every metric of one direction: per-lane sums -> wave totals parked in red[dir][wave][m*13]
template <int BLOCK, int Q, int NMET>
__device__ __forceinline__ void park_sqrt_sums(const float (&best)[Q][NMET], const unsigned (&selbits)[NMET], float* red_wave) {
  float bd[Q];
#pragma unroll
  for (int m = 0; m < NMET; ++m) {
#pragma unroll
    for (int k = 0; k < Q; ++k) bd[k] = best[k][m];
    park(lane_sqrt_sum<Q>(bd, selbits[m]), red_wave + m * kAccN);
  }
}

// G, GP of the metrics in `mask` of one direction -> red[dir][wave][m*13 + 1 ..]; no barrier in here
template <int BLOCK, int Q, int NMET, int DIR, int OWN>
__device__ __forceinline__ void park_grad_sums(const Smem& sm, const float4* __restrict__ refs, const float (&qx)[Q],
                                               const float (&qy)[Q], const float (&qz)[Q], const float (&best)[Q][NMET],
                                               const int (&btile)[Q][NMET], const unsigned (&selbits)[NMET], unsigned mask,
                                               int count, const float (&px)[Q], const float (&py)[Q], const float (&pz)[Q],
                                               float* red_wave, short* ws, int ws_stride) {
  float bd[Q], g[kGradN];
  int bt[Q];
#define HOUV_GRAD(MET)                                                                                              \
  if ((mask >> MET) & 1u) {                                                                                         \
    _Pragma("unroll") for (int k = 0; k < Q; ++k) {                                                                 \
      bd[k] = best[k][MET];                                                                                         \
      bt[k] = btile[k][MET];                                                                                        \
    }                                                                                                               \
    lane_grad_sums<BLOCK, Q, MET, DIR, OWN>(sm, refs, qx, qy, qz, bd, bt, selbits[MET], count, px, py, pz, g,       \
                                            ws ? ws + (size_t)MET * ws_stride : nullptr);                           \
    _Pragma("unroll") for (int i = 0; i < kGradN; ++i) park(g[i], red_wave + MET * kAccN + 1 + i);                  \
  }
  HOUV_GRAD(0)
  if constexpr (NMET == 4) {
    HOUV_GRAD(1)
    HOUV_GRAD(2)
    HOUV_GRAD(3)
  }
#undef HOUV_GRAD
}

// cross-wave sums (wave order) of the parked partials into sm.acc[(metric*2+dir)][..]; call after a barrier
template <int BLOCK, int NMET>
__device__ __forceinline__ void final_sums(const Smem& sm, int dir, bool want_s, unsigned grad_mask) {
  constexpr int NW = BLOCK / 64;
  const int tid = threadIdx.x;
  if (tid < NMET * kAccN) {
    const int m = tid / kAccN, i = tid % kAccN;
    if ((i == 0) ? want_s : (((grad_mask >> m) & 1u) != 0u)) {
      const float* r = sm.red + (size_t)dir * NW * kRedStride + tid;
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) a += r[w * kRedStride];
      sm.acc[(m * 2 + dir) * kAccStride + i] = a;
    }
  }
}

// which direction wins each metric's min: bit m set = direction 1 (over the moved points).  The rule of the scalar tail
// (torch.min(cat([first, second])): first wins ties), evaluated by every thread on the same LDS values.
template <int NMET>
__device__ __forceinline__ unsigned picked_direction(const Smem& sm, int k_full, int k_view) {
  unsigned pick = 0u;
#pragma unroll
  for (int m = 0; m < NMET; ++m) {
    const float kk = (float)((m == 0) ? k_full : k_view);
    const float cd0 = sm.acc[(m * 2 + 0) * kAccStride] / kk, cd1 = sm.acc[(m * 2 + 1) * kAccStride] / kk;
    pick |= (cd0 <= cd1) ? 0u : (1u << m);
  }
  return pick;
}

// Mis-prediction repair (rare): metric MET's gradient flows through direction A, but A's rescans were skipped because the
// previous iteration's winner was B and A's sweep state is gone.  Redo A for this one metric: moved points, single-metric
// sweep (bit-identical minima and sub-tiles: same expression tree, same tie rule), selection, rescan, sums.
template <int BLOCK, int Q, int MET, int OWN>
__device__ __forceinline__ void repair_direction_a(const Smem& sm, const float* __restrict__ src, int N, int mpad, int k_sel,
                                                   int& hrot, float* red_wave) {
  float sx[Q], sy[Q], sz[Q], mx[Q], my[Q], mz[Q];
  float R[9], T[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = sm.pose[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) T[i] = sm.pose[9 + i];
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    const int i = pt_index<BLOCK, Q, OWN>(k);
    const bool ok = i < N;
    sx[k] = ok ? src[i * 3 + 0] : 0.f;
    sy[k] = ok ? src[i * 3 + 1] : 0.f;
    sz[k] = ok ? src[i * 3 + 2] : 0.f;
    mx[k] = __builtin_fmaf(sz[k], R[2], __builtin_fmaf(sy[k], R[1], sx[k] * R[0])) + T[0];
    my[k] = __builtin_fmaf(sz[k], R[5], __builtin_fmaf(sy[k], R[4], sx[k] * R[3])) + T[1];
    mz[k] = __builtin_fmaf(sz[k], R[8], __builtin_fmaf(sy[k], R[7], sx[k] * R[6])) + T[2];
  }
  float bd[Q];
  int bt[Q];
  sweep_one<Q, MET>(sm.tgt, mpad / kTrk, mx, my, mz, bd, bt);
  bool sel[Q];
  unsigned key[Q], bits = 0u;
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    sel[k] = pt_index<BLOCK, Q, OWN>(k) < N;
    key[k] = sel[k] ? __float_as_uint(bd[k]) : 0xFFFFFFFFu;
  }
  if (k_sel < N) select_smallest<BLOCK, Q>(key, k_sel, sm.hist, sm.ctl, sel, hrot);
#pragma unroll
  for (int k = 0; k < Q; ++k) bits |= sel[k] ? (1u << k) : 0u;
  float g[kGradN];
  lane_grad_sums<BLOCK, Q, MET, 1, OWN>(sm, sm.tgt, mx, my, mz, bd, bt, bits, N, sx, sy, sz, g, nullptr);
#pragma unroll
  for (int i = 0; i < kGradN; ++i) park(g[i], red_wave + MET * kAccN + 1 + i);
}

// PRUNE: the exact pruned search of houv_sweep.h.  OWN: a lane owns Q/OWN chunks of OWN consecutive points (pt_index);
// 1 (strided, coalesced loads) everywhere by default -- other values are build-time experiments of the pruned mode
// (HOUV_PRUNE_OWN), for which <PRUNE=false, OWN> is the brute-force sweep under the same summation order (ws_valid=-1).
// Waves per SIMD the register budget is set for: 4 (128 VGPRs); 8 (64 VGPRs) where a lane owns one point.
template <int BLOCK, int Q, int NMET, int PRUNE, int OWN>
__global__ __launch_bounds__(BLOCK, (Q == 1 ? 8 : 4)) void solve_kernel(SolveArgs a) {
  static_assert(PRUNE < 2 || OWN == 1, "the balanced pruned sweep keeps the strided point ownership");
  constexpr int TS = (PRUNE == 3) ? 1 : 0;                      // visit masks over super-tiles of 32 << TS references
  constexpr int kPad = kSub << TS;
  extern __shared__ __attribute__((aligned(512))) unsigned char smem_raw[];   // 512 B: pruned_sweep's XOR-rotated gathers
  const int N = a.N, M = a.M;
  const Smem sm = carve(smem_raw, N, M, BLOCK, PRUNE >= 2 ? BLOCK * Q : 0, kPad);
  const int tid = threadIdx.x;
  const int ninst = a.P * a.K;
  // XCD-aware placement: workgroups b and b+8 share an XCD (and its L2), so give each XCD a contiguous
  // range of hypotheses -> the K restarts of one pair read the pair's clouds through ONE L2.
  int inst = blockIdx.x;
  if ((ninst & 7) == 0) inst = (blockIdx.x & 7) * (ninst >> 3) + (blockIdx.x >> 3);
  const int pair = inst / a.K;
  const float* __restrict__ src = a.src + (size_t)pair * N * 3;
  const float* __restrict__ tgt = a.tgt + (size_t)pair * M * 3;
  const int npad = (N + kPad - 1) / kPad * kPad, mpad = (M + kPad - 1) / kPad * kPad;
  const float4 pad4 = make_float4(INFINITY, INFINITY, INFINITY, 0.f);   // padding references never win

  for (int j = tid; j < mpad; j += BLOCK) sm.tgt[j] = (j < M) ? make_float4(tgt[j * 3], tgt[j * 3 + 1], tgt[j * 3 + 2], 0.f) : pad4;
  for (int j = N + tid; j < npad; j += BLOCK) sm.mov[j] = pad4;
  if (tid < 24) sm.state[tid] = a.state[(size_t)inst * 24 + tid];
  for (int j = tid; j < kHistBins; j += BLOCK) sm.hist[j] = 0u;   // radix-select histogram set 0 (select_smallest rotates)
  if constexpr (PRUNE >= 2) {
    if (tid < 132) sm.st.hist[tid] = 0;                           // list-length bins of the balanced pruned sweep
  }
  int hrot = 0;
  __syncthreads();
  const int rot = tid & (kSub - 1);
  short* ws_a = nullptr;   // NN of the moved points in the target (direction 1)
  short* ws_b = nullptr;   // NN of the target points in the moved cloud (direction 0)
  float4* ws_res = nullptr;   // balanced walk: per-query minima on their way back to the owning lanes
  if constexpr (PRUNE) {
    ws_b = a.nn_ws + ((size_t)inst * 16 + 0) * a.ws_stride;
    ws_a = a.nn_ws + ((size_t)inst * 16 + 4) * a.ws_stride;
    ws_res = reinterpret_cast<float4*>(a.nn_ws + ((size_t)inst * 16 + 8) * a.ws_stride);
    float tx0[Q], ty0[Q], tz0[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      const int i = pt_index<BLOCK, Q, OWN>(k);
      const float4 v = (i < M) ? sm.tgt[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      tx0[k] = v.x; ty0[k] = v.y; tz0[k] = v.z;
    }
    tile_boxes<BLOCK, Q, OWN, TS>(tx0, ty0, tz0, M, mpad / kPad, sm.tbox);   // the target is static: boxes once per launch
  }
  // Adam's step-dependent scalars (two double pow()) are computed off the critical path: by thread kAdamTid (another wave,
  // hence another SIMD, when the workgroup has one) one iteration ahead, into the slot of the step's parity.
  constexpr int kAdamTid = BLOCK >= 128 ? 64 : 0;
  if (tid == 0) {
    float p[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k] = (float)sm.state[k];
    Pose f;
    pose_forward(p, a.angle_base, a.trans_mode, f);
    store_pose(sm.pose, f);
  }
  if (tid == kAdamTid) {
    const int step = a.steps_done + 1;
    const AdamScalars asc = adam_scalars(step, a.lr, a.beta1, a.beta2);
    sm.adam[(step & 1) * 2 + 0] = asc.step_size;
    sm.adam[(step & 1) * 2 + 1] = asc.bc2_sqrt;
  }
  __syncthreads();

#ifdef HOUV_STAMPS
  unsigned long long t_stamp_ = __builtin_readcyclecounter();
#endif
  unsigned long long clk0 = 0ull, rt0 = 0ull;   // two stamps per LAUNCH (not per iteration), only when the counters are on
  if (a.stats) {
    clk0 = __builtin_amdgcn_s_memtime();
    rt0 = __builtin_amdgcn_s_memrealtime();
  }
  constexpr int NW = BLOCK / 64;
  constexpr unsigned kAllMet = (1u << NMET) - 1u;
  // Gradient-direction prediction: bit m = "metric m's min was won by direction A (over the moved points) in the previous
  // iteration".  A's rescans + G/GP sums run only for predicted-A metrics (A's sweep state is gone by the time the winner
  // is known); B's run exactly for the metrics B wins; a metric predicted B but won by A is repaired (rare).  Results do
  // not depend on the prediction.  The pruned kernel's bounds are distances to REMEMBERED nearest neighbours (nn_ws): any
  // remembered point gives a valid, attained bound, so a skipped rescan only leaves an older neighbour in place (a
  // slightly looser bound); every ws_refresh-th iteration rescans everything to keep them fresh.
  unsigned pred_a = kAllMet;
  float* red_a = sm.red + ((size_t)1 * NW + (tid >> 6)) * kRedStride;
  float* red_b = sm.red + ((size_t)0 * NW + (tid >> 6)) * kRedStride;
#pragma unroll 1
  for (int it = 0; it < a.n_iters; ++it) {
    float best[Q][NMET];
    int btile[Q][NMET];
    if (a.pred_mode == 1) pred_a = 0u;
    const bool allgrad = a.pred_mode == 2 || ((PRUNE != 0) && (a.ws_refresh <= 1 || ((a.steps_done + it) % a.ws_refresh) == 0 ||
                                                        (a.ws_valid == 0 && it == 0)));
    const unsigned grad_a = allgrad ? kAllMet : pred_a;
    {
      // ---- move this lane's source points, publish them as references for sweep B ----
      float sx[Q], sy[Q], sz[Q], mx[Q], my[Q], mz[Q];
      float R[9], T[3];
#pragma unroll
      for (int i = 0; i < 9; ++i) R[i] = sm.pose[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) T[i] = sm.pose[9 + i];
#pragma unroll
      for (int k = 0; k < Q; ++k) {
        const int i = pt_index<BLOCK, Q, OWN>(k);
        const bool ok = i < N;
        sx[k] = ok ? src[i * 3 + 0] : 0.f;
        sy[k] = ok ? src[i * 3 + 1] : 0.f;
        sz[k] = ok ? src[i * 3 + 2] : 0.f;
        // src @ R^T + T (houv.py:102)
        mx[k] = __builtin_fmaf(sz[k], R[2], __builtin_fmaf(sy[k], R[1], sx[k] * R[0])) + T[0];
        my[k] = __builtin_fmaf(sz[k], R[5], __builtin_fmaf(sy[k], R[4], sx[k] * R[3])) + T[1];
        mz[k] = __builtin_fmaf(sz[k], R[8], __builtin_fmaf(sy[k], R[7], sx[k] * R[6])) + T[2];
        if (ok) sm.mov[i] = make_float4(mx[k], my[k], mz[k], 0.f);
      }
      __syncthreads();
      HOUV_STAMP(0);
      // ---- sweep A: moved -> target ----
      bool pruned_now = false;
      if constexpr (PRUNE) {
        tile_boxes<BLOCK, Q, OWN, TS>(mx, my, mz, N, npad / kPad, sm.mbox);   // read by sweep B after the next barriers
        pruned_now = (a.ws_valid != 0) || (it > 0);
      }
      if (pruned_now) {
        if constexpr (PRUNE >= 2) {
          pruned_sweep_sorted<BLOCK, Q, NMET, TS>(sm.tgt, sm.tbox, mpad / kPad, sm.mov, sm.tgt, sm.mov, mx, my, mz, ws_a, a.ws_stride, N,
                                              rot, sm.st, ws_res, best, btile, a.stats);
        } else if constexpr (PRUNE == 1) {
          pruned_sweep<BLOCK, Q, NMET, OWN>(sm.tgt, sm.tbox, mpad / kSub, mx, my, mz, ws_a, a.ws_stride, N, rot, best, btile, a.stats, a.cap_slack);
        }
      } else {
        sweep<Q, NMET>(sm.tgt, mpad / kTrk, mx, my, mz, best, btile);
        if (a.stats && (tid & 63) == 0) atomicAdd(&a.stats[3], 1ull);
      }
      HOUV_STAMP(1);
      // ---- epilogue A: selection, S of every metric, G/GP of the predicted-A metrics; one barrier ----
      unsigned sel[NMET];
      select_all<BLOCK, Q, NMET, OWN>(sm, best, N, a.k_full, a.k_view, hrot, sel);
      HOUV_STAMP(9);
      park_sqrt_sums<BLOCK, Q, NMET>(best, sel, red_a);
      park_grad_sums<BLOCK, Q, NMET, 1, OWN>(sm, sm.tgt, mx, my, mz, best, btile, sel, grad_a, N, sx, sy, sz, red_a, ws_a,
                                            a.ws_stride);
      HOUV_STAMP(8);
      __syncthreads();
      final_sums<BLOCK, NMET>(sm, 1, true, grad_a);
      HOUV_STAMP(2);
    }
    unsigned pick_a;
    {
      // ---- sweep B: target -> moved ----
      float tx[Q], ty[Q], tz[Q];
#pragma unroll
      for (int k = 0; k < Q; ++k) {
        const int i = pt_index<BLOCK, Q, OWN>(k);
        const float4 v = (i < M) ? sm.tgt[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        tx[k] = v.x; ty[k] = v.y; tz[k] = v.z;
      }
      const bool pruned_now = (PRUNE != 0) && ((a.ws_valid != 0) || (it > 0));
      if (pruned_now) {
        if constexpr (PRUNE >= 2) {
          pruned_sweep_sorted<BLOCK, Q, NMET, TS>(sm.mov, sm.mbox, npad / kPad, sm.tgt, sm.tgt, sm.mov, tx, ty, tz, ws_b, a.ws_stride, M,
                                              rot, sm.st, ws_res, best, btile, a.stats);
        } else if constexpr (PRUNE == 1) {
          pruned_sweep<BLOCK, Q, NMET, OWN>(sm.mov, sm.mbox, npad / kSub, tx, ty, tz, ws_b, a.ws_stride, M, rot, best, btile, a.stats, a.cap_slack);
        }
      } else {
        sweep<Q, NMET>(sm.mov, npad / kTrk, tx, ty, tz, best, btile);
        if (a.stats && (tid & 63) == 0) atomicAdd(&a.stats[3], 1ull);
      }
      HOUV_STAMP(3);
      // ---- epilogue B: selection and S first; then the winners are known to every thread ----
      unsigned sel[NMET];
      select_all<BLOCK, Q, NMET, OWN>(sm, best, M, a.k_full, a.k_view, hrot, sel);
      HOUV_STAMP(9);
      park_sqrt_sums<BLOCK, Q, NMET>(best, sel, red_b);
      __syncthreads();
      final_sums<BLOCK, NMET>(sm, 0, true, 0u);
      __syncthreads();
      pick_a = picked_direction<NMET>(sm, a.k_full, a.k_view);
      const unsigned grad_b = allgrad ? kAllMet : (~pick_a & kAllMet);
      park_grad_sums<BLOCK, Q, NMET, 0, OWN>(sm, sm.mov, tx, ty, tz, best, btile, sel, grad_b, M, tx, ty, tz, red_b, ws_b,
                                            a.ws_stride);
      HOUV_STAMP(8);
      // ---- repair: won by A, but A's rescans were skipped ----
      const unsigned miss = pick_a & ~grad_a & kAllMet;
      if (miss) {
        if (miss & 1u) repair_direction_a<BLOCK, Q, 0, OWN>(sm, src, N, mpad, a.k_full, hrot, red_a);
        if constexpr (NMET == 4) {
          if (miss & 2u) repair_direction_a<BLOCK, Q, 1, OWN>(sm, src, N, mpad, a.k_view, hrot, red_a);
          if (miss & 4u) repair_direction_a<BLOCK, Q, 2, OWN>(sm, src, N, mpad, a.k_view, hrot, red_a);
          if (miss & 8u) repair_direction_a<BLOCK, Q, 3, OWN>(sm, src, N, mpad, a.k_view, hrot, red_a);
        }
      }
      __syncthreads();
      final_sums<BLOCK, NMET>(sm, 0, false, grad_b);
      if (miss) final_sums<BLOCK, NMET>(sm, 1, false, miss);
      HOUV_STAMP(4);
    }
#ifdef HOUV_STAMPS
    if (tid == 0) {   // prediction statistics: metric-iterations with A rescanned / won by A / repaired / total
      HOUV_STAMP_ADD(12, (unsigned long long)__popc(grad_a));
      HOUV_STAMP_ADD(13, (unsigned long long)__popc(pick_a));
      HOUV_STAMP_ADD(14, (unsigned long long)__popc(pick_a & ~grad_a & kAllMet));
      HOUV_STAMP_ADD(15, (unsigned long long)NMET);
    }
#endif
    pred_a = pick_a;
    __syncthreads();
    HOUV_STAMP(5);

    // ---- per-hypothesis scalar tail: loss, closed-form gradient, Adam, next pose ----
    if (kAdamTid != 0 && tid == kAdamTid && it + 1 < a.n_iters) {   // next iteration's Adam scalars, while thread 0 works below
      const int step = a.steps_done + it + 2;
      const AdamScalars asc = adam_scalars(step, a.lr, a.beta1, a.beta2);
      sm.adam[(step & 1) * 2 + 0] = asc.step_size;
      sm.adam[(step & 1) * 2 + 1] = asc.bc2_sqrt;
    }
    if (tid == 0) {
      float p[8];
      Pose f;
      load_pose(f, sm.pose);        // the forward of the current parameters, kept from the end of the previous tail / the prologue
      float cd[NMET][2], val[NMET];
      int pick[NMET];
      float gT[3] = {0.f, 0.f, 0.f}, Mm[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float loss = 0.f;
      bool bad = false;
#pragma unroll
      for (int m = 0; m < NMET; ++m) {
        const float kk = (float)((m == 0) ? a.k_full : a.k_view);
        cd[m][0] = sm.acc[(m * 2 + 0) * kAccStride] / kk;   // over target points   (calc_cd_percent's 1st output)
        cd[m][1] = sm.acc[(m * 2 + 1) * kAccStride] / kk;   // over moved points    (2nd output)
        // torch.min(cat([first, second])): first wins ties; NaN propagates
        pick[m] = (cd[m][0] <= cd[m][1]) ? 0 : 1;
        val[m] = cd[m][pick[m]];
        if (cd[m][0] != cd[m][0] || cd[m][1] != cd[m][1]) { val[m] = NAN; bad = true; }
        const float w = ((m == 0) ? 6.0f : 1.0f) * a.loss_scale / kk;
        const float* ac = sm.acc + (m * 2 + pick[m]) * kAccStride;
#pragma unroll
        for (int i = 0; i < 3; ++i) gT[i] += w * ac[1 + i];
#pragma unroll
        for (int i = 0; i < 9; ++i) Mm[i] += w * ac[4 + i];
      }
      loss = val[0] * 6.0f;                                  // houv.py:222 / train_utils.py:433
      if constexpr (NMET == 4) loss = loss + (val[1] + val[2] + val[3]);
      if (bad) {
#pragma unroll
        for (int i = 0; i < 3; ++i) gT[i] = NAN;
#pragma unroll
        for (int i = 0; i < 9; ++i) Mm[i] = NAN;
      }
      float g[8];
      pose_backward(f, a.trans_mode, gT, Mm, g);
      if (it == a.n_iters - 1) {
        // outputs of the LAST forward (houv.py:134-136: the final step is never observed)
        if (a.out_score) a.out_score[inst] = val[0];
        if (a.out_loss) a.out_loss[inst] = loss;
        if (a.out_R)
          for (int k = 0; k < 9; ++k) a.out_R[(size_t)inst * 9 + k] = f.R[k];
        if (a.out_T)
          for (int k = 0; k < 3; ++k) a.out_T[(size_t)inst * 3 + k] = f.T[k];
        if (a.out_grad)
          for (int k = 0; k < 8; ++k) a.out_grad[(size_t)inst * 8 + k] = g[k];
        if (a.out_cd)
          for (int m = 0; m < 4; ++m)
            for (int d = 0; d < 2; ++d) a.out_cd[(size_t)inst * 8 + m * 2 + d] = (m < NMET) ? cd[m < NMET ? m : 0][d] : 0.f;
      }
      const int step = a.steps_done + it + 1;
      const AdamScalars asc{sm.adam[(step & 1) * 2 + 0], sm.adam[(step & 1) * 2 + 1]};
      if (a.f64_params) {
        for (int k = 0; k < 8; ++k)
          adam_step<double>(sm.state[k], sm.state[8 + k], sm.state[16 + k], (double)g[k], asc, a.beta1, a.beta2, a.eps);
      } else {
        for (int k = 0; k < 8; ++k) {
          float pp = (float)sm.state[k], mm = (float)sm.state[8 + k], vv = (float)sm.state[16 + k];
          adam_step<float>(pp, mm, vv, g[k], asc, a.beta1, a.beta2, a.eps);
          sm.state[k] = pp; sm.state[8 + k] = mm; sm.state[16 + k] = vv;
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) p[k] = (float)sm.state[k];
      pose_forward(p, a.angle_base, a.trans_mode, f);
      store_pose(sm.pose, f);
      if (kAdamTid == 0 && it + 1 < a.n_iters) {                       // single-wave workgroups: no other wave to do it
        const AdamScalars nxt = adam_scalars(step + 1, a.lr, a.beta1, a.beta2);
        sm.adam[((step + 1) & 1) * 2 + 0] = nxt.step_size;
        sm.adam[((step + 1) & 1) * 2 + 1] = nxt.bc2_sqrt;
      }
    }
    __syncthreads();
    HOUV_STAMP(6);
  }
  if (tid < 24) a.state[(size_t)inst * 24 + tid] = sm.state[tid];
  if (a.stats) {
    const unsigned long long dc = __builtin_amdgcn_s_memtime() - clk0, dr = __builtin_amdgcn_s_memrealtime() - rt0;
    if (tid == 0) {
      atomicAdd(&a.stats[4], dc);
      atomicAdd(&a.stats[5], dr);
    }
  }
}

template <int BLOCK, int Q, int PRUNE, int OWN>
int launch(const SolveArgs& a, int use_views, hipStream_t s) {
  const size_t bytes = smem_bytes(a.N, a.M, BLOCK, PRUNE, Q);
  const int grid = a.P * a.K;
  hipError_t e;
  if (use_views) {
    e = hipFuncSetAttribute((const void*)solve_kernel<BLOCK, Q, 4, PRUNE, OWN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { set_error("houv_solve_iterate: cannot reserve %zu B of LDS: %s", bytes, hipGetErrorString(e)); return 0; }
    solve_kernel<BLOCK, Q, 4, PRUNE, OWN><<<grid, BLOCK, bytes, s>>>(a);
  } else {
    e = hipFuncSetAttribute((const void*)solve_kernel<BLOCK, Q, 1, PRUNE, OWN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { set_error("houv_solve_iterate: cannot reserve %zu B of LDS: %s", bytes, hipGetErrorString(e)); return 0; }
    solve_kernel<BLOCK, Q, 1, PRUNE, OWN><<<grid, BLOCK, bytes, s>>>(a);
  }
  return check_launch("houv_solve_iterate") ? 1 : 0;
}

}  // namespace
}  // namespace houv

#ifdef HOUV_STAMPS
// sums of the per-workgroup stamp records (diagnostic build): first = 16 for the sweep's counters, 0 for the kernel's phases
static int read_stamp_records(unsigned long long* host_out, int first, int n, int reset) {
  static std::vector<unsigned long long> h(houv::kStampWgs * 24);
  if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(houv::g_stamp_wg), h.size() * sizeof(unsigned long long)) != hipSuccess) return 0;
  for (int i = 0; i < n; ++i) {
    unsigned long long acc = 0ull;
    for (int w = 0; w < houv::kStampWgs; ++w) acc += h[(size_t)w * 24 + first + i];
    host_out[i] = acc;
  }
  if (reset) {
    for (int w = 0; w < houv::kStampWgs; ++w)
      for (int i = 0; i < n; ++i) h[(size_t)w * 24 + first + i] = 0ull;
    if (hipMemcpyToSymbol(HIP_SYMBOL(houv::g_stamp_wg), h.data(), h.size() * sizeof(unsigned long long)) != hipSuccess) return 0;
  }
  return 1;
}
extern "C" int houv_debug_read_prune_stats(unsigned long long* host_out, int reset) { return read_stamp_records(host_out, 16, 8, reset); }
extern "C" int houv_debug_read_stamps(unsigned long long* host_out, int reset) { return read_stamp_records(host_out, 0, 16, reset); }
#endif

// pruned mode: consecutive points a lane owns per chunk (pt_index), for kernels with Q = 2 / Q = 4 points per lane
#ifndef HOUV_PRUNE_OWN
#define HOUV_PRUNE_OWN 1
#endif
constexpr int kOwn4 = HOUV_PRUNE_OWN;

// The variant table: which solve_kernel<BLOCK, Q> serves clouds of max(N, M) points -- the SAME (BLOCK, Q) for the brute-force
// sweep and for the pruned search, so that both sum in the same order and agree bit for bit.  Q = 3 points per lane covers
// the sizes between the powers of two without idle lanes (768, 1536, 3072).  The pruned search (balanced walk,
// pruned_sweep_sorted) serves 257..2048 points with one visit-mask bit per 32-point sub-tile (PRUNE = 2) and 2049..4096 points with
// one bit per 64-point super-tile (PRUNE = 3; one 1024-thread workgroup per CU there, like the brute-force kernel).  tests/test_host_logic.py enumerates this table and fails when a variant
// has no size that the GPU tests compare with the CPU oracle.
extern "C" int houv_solve_variant(int N, int M, int pruned, int* block, int* points_per_lane, int* prune_mode) {
  using namespace houv;
  const int mx = N > M ? N : M;
  if (N <= 0 || M <= 0) {
    set_error("houv_solve_variant: bad cloud sizes N=%d M=%d", N, M);
    return 0;
  }
  if (mx > 4096) {
    set_error("houv_solve_iterate: clouds larger than 4096 points do not fit in 160 KiB of LDS (N=%d M=%d)", N, M);
    return 0;
  }
  // Small clouds: no idle waves (every wave of a workgroup runs the whole epilogue, points or not).  Measured and NOT kept
  // (profiles/r03_sizes.txt): one wave with 2 points per lane at 65..128 points (+7 %, but it regroups the sums, and the
  // statistical G14 rung is calibrated on this grouping), one wave with 3-4 points per lane up to 256 points and two waves up to
  // 512 points (5-6 % SLOWER than 4 waves with 1-2 points per lane).
  int b, q;
  if (mx <= 64) { b = 64; q = 1; }
  else if (mx <= 128) { b = 128; q = 1; }
  else if (mx <= 256) { b = 256; q = 1; }
  else if (mx <= 512) { b = 256; q = 2; }
  else if (mx <= 768) { b = 256; q = 3; }
  else if (mx <= 1024) { b = 256; q = 4; }
  else if (mx <= 1536) { b = 512; q = 3; }
  else if (mx <= 2048) { b = 512; q = 4; }
  else if (mx <= 3072) { b = 1024; q = 3; }
  else { b = 1024; q = 4; }
  if (block) *block = b;
  if (points_per_lane) *points_per_lane = q;
  // Up to 256 points (8 sub-tiles or fewer, one point per lane) the pruned search is not built: houv_solve_iterate_pruned then runs
  // the brute-force kernel -- the same result.  With Morton-ordered sub-tiles it did not pay up to 512 points either; with k-d leaves
  // it does from 257 on (profiles/r03_sizes.txt: 512 points 0.119 -> 0.093 us, 320 points 0.084 -> 0.071).
  if (prune_mode) *prune_mode = (!pruned || mx <= g_debug.prune_min_points.load() - 1) ? 0 : (mx > 2048) ? 3 :
                               ((b == 512 && q == 4 && g_debug.prune_owner_walk.load()) ? 1 : 2);
  return 1;
}

static int solve_dispatch(const float* src, const float* tgt, int P, int N, int M, int K, double* state, int steps_done,
                          int n_iters, int angle_base, int trans_mode, int use_views, int f64_params, int k_full,
                          int k_view, double lr, double beta1, double beta2, double eps, float loss_scale,
                          float* out_score, float* out_loss, float* out_R, float* out_T, float* out_grad, float* out_cd,
                          short* nn_ws, int ws_valid, int ws_stride, bool prune, void* stream, const char* who) {
  using namespace houv;
  if (P < 0 || N <= 0 || M <= 0 || K <= 0 || n_iters <= 0 || steps_done < 0 || angle_base < 0 || angle_base > 3 ||
      trans_mode < 0 || trans_mode > 1) {
    set_error("%s: bad argument P=%d N=%d M=%d K=%d n_iters=%d steps_done=%d base=%d trans_mode=%d", who, P, N, M, K,
              n_iters, steps_done, angle_base, trans_mode);
    return 0;
  }
  if (P == 0) return 1;
  if (!src || !tgt || !state) {
    set_error("%s: null pointer", who);
    return 0;
  }
  // topk(k) over a direction with fewer than k points raises in the reference (model_utils_completion.py:91-92)
  const int kv = use_views ? k_view : 1;
  if (k_full < 1 || k_full > N || k_full > M || kv < 1 || kv > N || kv > M) {
    set_error("%s: top-k size out of range (k_full=%d k_view=%d N=%d M=%d)", who, k_full, k_view, N, M);
    return 0;
  }
  if ((long long)P * K > 0x7fffffffLL) {
    set_error("%s: too many hypotheses", who);
    return 0;
  }
  // pruned mode refreshes every remembered NN on every 4th iteration (same-device A/B with the balanced walk,
  // profiles/r03_ab_refresh.txt: 1 -> 0.697, 2 -> 0.667, 4 -> 0.658, 6 -> 0.657, 8 -> 0.658 us per hypothesis-iteration; round 2's
  // owner walk, whose steps cost more, preferred 2: profiles/r02_ab_pruned_refresh.txt; results identical in all).
  // pred_mode / ws_refresh / cap_slack / stats are diagnostics set through houv_debug_set(), never through the environment.
  SolveArgs a{src, tgt, P, N, M, K, state, steps_done, n_iters, angle_base, trans_mode, f64_params, k_full, k_view,
              lr, beta1, beta2, eps, loss_scale, out_score, out_loss, out_R, out_T, out_grad, out_cd, nn_ws, ws_valid,
              ws_stride, g_debug.pred_mode.load(), g_debug.ws_refresh.load(), g_debug.prune_cap_slack.load(),
              reinterpret_cast<unsigned long long*>(g_debug.stats.load())};
  hipStream_t s = (hipStream_t)stream;
  const int mx = N > M ? N : M;
  int block = 0, q = 0, mode = 0;
  if (!houv_solve_variant(N, M, prune ? 1 : 0, &block, &q, &mode)) return 0;
  if (prune && (!nn_ws || ws_stride < mx || (ws_stride & 7))) {
    set_error("%s: pruned mode needs a workspace of 16 x ws_stride int16 per hypothesis, ws_stride >= max(N,M) and a multiple of 8", who);
    return 0;
  }
  if (prune && ws_valid < 0) {   // test aid: the pruned entry point with the search switched off = the brute-force kernel of this size
    a.ws_valid = 0;
    mode = 0;
  }
  if (mode == 0) a.nn_ws = nullptr;
  // one instantiation per row of the variant table (houv_solve_variant), x {views, no views}, x {brute force, pruned}
#define HOUV_GO(B_, Q_)                                                          \
  if (block == B_ && q == Q_) {                                                  \
    if (mode == 2) return launch<B_, Q_, ((B_ >= 256 && Q_ >= 2) ? 2 : 0), 1>(a, use_views, s); \
    return launch<B_, Q_, 0, 1>(a, use_views, s);                                \
  }
  if (mx <= 2048) {
    if (mode == 1) return launch<512, 4, 1, kOwn4>(a, use_views, s);   // round 2's owner walk, A/B only (prune_owner_walk)
    HOUV_GO(64, 1) HOUV_GO(128, 1) HOUV_GO(256, 1) HOUV_GO(256, 2) HOUV_GO(256, 3) HOUV_GO(256, 4)
    HOUV_GO(512, 3) HOUV_GO(512, 4)
  } else {
    if (block == 1024 && q == 3) return mode == 3 ? launch<1024, 3, 3, 1>(a, use_views, s) : launch<1024, 3, 0, 1>(a, use_views, s);
    if (block == 1024 && q == 4) return mode == 3 ? launch<1024, 4, 3, 1>(a, use_views, s) : launch<1024, 4, 0, 1>(a, use_views, s);
  }
#undef HOUV_GO
  set_error("%s: no kernel variant <%d,%d>", who, block, q);
  return 0;
}

extern "C" int houv_solve_iterate(const float* src, const float* tgt, int P, int N, int M, int K, double* state,
                                  int steps_done, int n_iters, int angle_base, int trans_mode, int use_views,
                                  int f64_params, int k_full, int k_view, double lr, double beta1, double beta2,
                                  double eps, float loss_scale, float* out_score, float* out_loss, float* out_R,
                                  float* out_T, float* out_grad, float* out_cd, void* stream) {
  return solve_dispatch(src, tgt, P, N, M, K, state, steps_done, n_iters, angle_base, trans_mode, use_views, f64_params,
                        k_full, k_view, lr, beta1, beta2, eps, loss_scale, out_score, out_loss, out_R, out_T, out_grad,
                        out_cd, nullptr, 0, 0, false, stream, "houv_solve_iterate");
}

extern "C" int houv_solve_iterate_pruned(const float* src, const float* tgt, int P, int N, int M, int K, double* state,
                                         int steps_done, int n_iters, int angle_base, int trans_mode, int use_views,
                                         int f64_params, int k_full, int k_view, double lr, double beta1, double beta2,
                                         double eps, float loss_scale, float* out_score, float* out_loss, float* out_R,
                                         float* out_T, float* out_grad, float* out_cd, int16_t* nn_ws, int ws_valid,
                                         int ws_stride, void* stream) {
  return solve_dispatch(src, tgt, P, N, M, K, state, steps_done, n_iters, angle_base, trans_mode, use_views, f64_params,
                        k_full, k_view, lr, beta1, beta2, eps, loss_scale, out_score, out_loss, out_R, out_T, out_grad,
                        out_cd, (short*)nn_ws, ws_valid, ws_stride, true, stream, "houv_solve_iterate_pruned");
}

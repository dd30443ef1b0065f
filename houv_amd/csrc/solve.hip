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
//   * the un-moved source point needed for sum G p^T in the target->moved direction is R^T(p' - T).
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
  short* nn_ws;      // pruned mode: [P*K][2 dirs][4 metrics][ws_stride] index of each query's NN in the last iteration
  int ws_valid;      //   1: nn_ws holds the NNs of the iteration before this launch's first one
  int ws_stride;
};

#ifdef HOUV_STAMPS
// Diagnostic build only (scripts/stamps.sh): per-phase wave-cycle totals, never read by the kernel itself.
__device__ unsigned long long g_stamp[16];
#define HOUV_STAMP(i)                                                        \
  do {                                                                       \
    const unsigned long long now_ = __builtin_readcyclecounter();           \
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_stamp[i], now_ - t_stamp_);   \
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

struct Smem {
  float4* tgt;     // [Mpad]
  float4* mov;     // [Npad]
  double* state;   // [24]
  float* pose;     // [12] R row-major, T
  float* acc;      // [8][kAccStride]   slot = metric*2 + dir
  float* red;      // [NW][kAccStride]
  unsigned* hist;  // [256]
  int* ctl;        // [8 + NW]
  float4* tbox;    // [2*64] lo/hi boxes of the target's 32-point sub-tiles   (pruned mode)
  float4* mbox;    // [2*64] same for the moved cloud, rebuilt every iteration
};

__host__ __device__ inline size_t smem_bytes(int N, int M, int block) {
  const int npad = (N + kSub - 1) / kSub * kSub, mpad = (M + kSub - 1) / kSub * kSub;
  const int nw = block / 64;
  return (size_t)(npad + mpad) * 16 + 24 * 8 + 12 * 4 + 8 * kAccStride * 4 + (size_t)nw * kAccStride * 4 + 256 * 4 +
         (8 + nw) * 4 + 64 + 2 * 128 * 16;
}

__device__ inline Smem carve(unsigned char* base, int N, int M, int block) {
  const int npad = (N + kSub - 1) / kSub * kSub, mpad = (M + kSub - 1) / kSub * kSub;
  const int nw = block / 64;
  Smem s;
  s.tgt = reinterpret_cast<float4*>(base);
  s.mov = s.tgt + mpad;
  s.state = reinterpret_cast<double*>(s.mov + npad);
  s.pose = reinterpret_cast<float*>(s.state + 24);
  s.acc = s.pose + 12;
  s.red = s.acc + 8 * kAccStride;
  s.hist = reinterpret_cast<unsigned*>(s.red + nw * kAccStride);
  s.ctl = reinterpret_cast<int*>(s.hist + 256);
  s.tbox = reinterpret_cast<float4*>((reinterpret_cast<uintptr_t>(s.ctl + 8 + nw) + 15) & ~(uintptr_t)15);
  s.mbox = s.tbox + 128;
  return s;
}

// Exact selection of the `ksel` smallest of the BLOCK*Q keys (fp32 bit patterns of non-negative
// distances; 0xFFFFFFFF marks "not a point").  4-pass 8-bit radix select, LDS histogram.
// Ties at the threshold are taken in (thread, k) order -- torch.topk leaves tie order unspecified.
template <int BLOCK, int Q>
__device__ __forceinline__ void select_smallest(const unsigned (&key)[Q], int ksel, unsigned* hist, int* ctl,
                                                bool (&sel)[Q]) {
  constexpr int NW = BLOCK / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned prefix = 0u, mask = 0u;
  int remaining = ksel, neq = 0;
#pragma unroll 1
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    __syncthreads();
    for (int i = tid; i < 256; i += BLOCK) hist[i] = 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < Q; ++k)
      if ((key[k] & mask) == prefix) atomicAdd(&hist[(key[k] >> shift) & 255u], 1u);
    __syncthreads();
    if (wave == 0) {
      const int h0 = (int)hist[4 * lane + 0], h1 = (int)hist[4 * lane + 1], h2 = (int)hist[4 * lane + 2],
                h3 = (int)hist[4 * lane + 3];
      const int tot = h0 + h1 + h2 + h3;
      int c = wave_incl_scan_i(tot) - tot;
      const int hh[4] = {h0, h1, h2, h3};
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (c < remaining && remaining <= c + hh[b]) {
          ctl[0] = 4 * lane + b;
          ctl[1] = c;
          ctl[2] = hh[b];
        }
        c += hh[b];
      }
    }
    __syncthreads();
    prefix |= (unsigned)ctl[0] << shift;
    mask |= 255u << shift;
    remaining -= ctl[1];
    neq = ctl[2];
  }
  if (neq == remaining) {
#pragma unroll
    for (int k = 0; k < Q; ++k) sel[k] = key[k] <= prefix;
  } else {
    int e = 0;
#pragma unroll
    for (int k = 0; k < Q; ++k) e += (key[k] == prefix) ? 1 : 0;
    const int incl = wave_incl_scan_i(e);
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

// Epilogue of one sweep for metric MET.
//   DIR == 1: queries are this lane's moved points (count = N), references the target cloud.
//   DIR == 0: queries are this lane's target points (count = M), references the moved cloud.
// Recovers the exact NN, selects the ksel smallest distances, and reduces
//   S = sum sqrt(d),  G = sum c,  GP = sum c p^T,   c = mask * (moved - target) / sqrt(d),  p = un-moved source point
// over the selection into acc_out[0..13).
template <int BLOCK, int Q, int MET, int DIR, int OWN>
__device__ __forceinline__ void epilogue_metric(const Smem& sm, const float4* __restrict__ refs, const float (&qx)[Q],
                                                const float (&qy)[Q], const float (&qz)[Q], const float (&bestm)[Q],
                                                const int (&btilem)[Q], int count, int ksel, const float (&px)[Q],
                                                const float (&py)[Q], const float (&pz)[Q], float* acc_out, short* ws HOUV_STAMP_PARAM) {
  const int tid = threadIdx.x;
  const int rot = tid & (kSub - 1);
  float nx[Q], ny[Q], nz[Q];
  unsigned key[Q];
  bool sel[Q];
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    const float bd = bestm[k];
    int jn;
    const float4 nn = recover_nn<MET, kRescanBatch, true>(refs + btilem[k] * kSub, qx[k], qy[k], qz[k], bd, rot, jn);
    if (ws && pt_index<BLOCK, Q, OWN>(k) < count) ws[pt_index<BLOCK, Q, OWN>(k)] = (short)(btilem[k] * kSub + jn);
    nx[k] = nn.x; ny[k] = nn.y; nz[k] = nn.z;
    const bool valid = pt_index<BLOCK, Q, OWN>(k) < count;
    key[k] = valid ? __float_as_uint(bd) : 0xFFFFFFFFu;
    sel[k] = valid;
  }
  HOUV_STAMP(8);
  if (ksel < count) select_smallest<BLOCK, Q>(key, ksel, sm.hist, sm.ctl, sel);
  HOUV_STAMP(9);

  float acc[kAccN];
#pragma unroll
  for (int i = 0; i < kAccN; ++i) acc[i] = 0.f;
  float R[9], T[3];
  if constexpr (DIR == 0) {
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = sm.pose[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) T[i] = sm.pose[9 + i];
  }
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    if (sel[k]) {
      // no finite distance (NaN pose from an earlier sqrt'(0)): torch's min/topk/sqrt chain yields NaN, not inf
      const float s = (bestm[k] < INFINITY) ? sqrtf(bestm[k]) : NAN;
      const float inv = 1.0f / s;   // d == 0 -> inf, and 0*inf = NaN below, as torch's sqrt backward gives
      float dx, dy, dz, sx, sy, sz;
      if constexpr (DIR == 1) {
        dx = qx[k] - nx[k]; dy = qy[k] - ny[k]; dz = qz[k] - nz[k];
        sx = px[k]; sy = py[k]; sz = pz[k];
      } else {
        dx = nx[k] - qx[k]; dy = ny[k] - qy[k]; dz = nz[k] - qz[k];
        const float ux = nx[k] - T[0], uy = ny[k] - T[1], uz = nz[k] - T[2];
        sx = R[0] * ux + R[3] * uy + R[6] * uz;   // R^T (p' - T)
        sy = R[1] * ux + R[4] * uy + R[7] * uz;
        sz = R[2] * ux + R[5] * uy + R[8] * uz;
      }
      if constexpr (MET == 1) dx = 0.f;
      if constexpr (MET == 2) dy = 0.f;
      if constexpr (MET == 3) dz = 0.f;
      const float cx = dx * inv, cy = dy * inv, cz = dz * inv;
      acc[0] += s;
      acc[1] += cx; acc[2] += cy; acc[3] += cz;
      acc[4] += cx * sx; acc[5] += cx * sy; acc[6] += cx * sz;
      acc[7] += cy * sx; acc[8] += cy * sy; acc[9] += cy * sz;
      acc[10] += cz * sx; acc[11] += cz * sy; acc[12] += cz * sz;
    }
  }
  HOUV_STAMP(10);
  block_sum<BLOCK, kAccN>(acc, sm.red, acc_out);
  HOUV_STAMP(11);
}

template <int BLOCK, int Q, int NMET, int DIR, int OWN>
__device__ __forceinline__ void epilogue(const Smem& sm, const float4* __restrict__ refs, const float (&qx)[Q],
                                         const float (&qy)[Q], const float (&qz)[Q], const float (&best)[Q][NMET],
                                         const int (&btile)[Q][NMET], int count, int k_full, int k_view,
                                         const float (&px)[Q], const float (&py)[Q], const float (&pz)[Q], short* ws,
                                         int ws_stride HOUV_STAMP_PARAM) {
  float bm[Q];
  int bt[Q];
#define HOUV_EPI(MET)                                                                                              \
  {                                                                                                                \
    _Pragma("unroll") for (int k = 0; k < Q; ++k) {                                                                \
      bm[k] = best[k][MET];                                                                                        \
      bt[k] = btile[k][MET];                                                                                       \
    }                                                                                                              \
    epilogue_metric<BLOCK, Q, MET, DIR, OWN>(sm, refs, qx, qy, qz, bm, bt, count, (MET == 0) ? k_full : k_view, px, py, \
                                        pz, sm.acc + (MET * 2 + DIR) * kAccStride,                                 \
                                        ws ? ws + (size_t)MET * ws_stride : nullptr HOUV_STAMP_ARG);               \
  }
  HOUV_EPI(0)
  if constexpr (NMET == 4) {
    HOUV_EPI(1)
    HOUV_EPI(2)
    HOUV_EPI(3)
  }
#undef HOUV_EPI
}

// PRUNE: the exact pruned search of houv_sweep.h.  OWN: a lane owns Q/OWN chunks of OWN consecutive points (pt_index);
// 1 (strided, coalesced loads) everywhere by default -- other values are build-time experiments of the pruned mode
// (HOUV_PRUNE_OWN), for which <PRUNE=false, OWN> is the brute-force sweep under the same summation order (ws_valid=-1).
template <int BLOCK, int Q, int NMET, bool PRUNE, int OWN>
__global__ __launch_bounds__(BLOCK, 4) void solve_kernel(SolveArgs a) {
  extern __shared__ __attribute__((aligned(512))) unsigned char smem_raw[];   // 512 B: pruned_sweep's XOR-rotated gathers
  const int N = a.N, M = a.M;
  const Smem sm = carve(smem_raw, N, M, BLOCK);
  const int tid = threadIdx.x;
  const int ninst = a.P * a.K;
  // XCD-aware placement: workgroups b and b+8 share an XCD (and its L2), so give each XCD a contiguous
  // range of hypotheses -> the K restarts of one pair read the pair's clouds through ONE L2.
  int inst = blockIdx.x;
  if ((ninst & 7) == 0) inst = (blockIdx.x & 7) * (ninst >> 3) + (blockIdx.x >> 3);
  const int pair = inst / a.K;
  const float* __restrict__ src = a.src + (size_t)pair * N * 3;
  const float* __restrict__ tgt = a.tgt + (size_t)pair * M * 3;
  const int npad = (N + kSub - 1) / kSub * kSub, mpad = (M + kSub - 1) / kSub * kSub;
  const float4 pad4 = make_float4(INFINITY, INFINITY, INFINITY, 0.f);   // padding references never win

  for (int j = tid; j < mpad; j += BLOCK) sm.tgt[j] = (j < M) ? make_float4(tgt[j * 3], tgt[j * 3 + 1], tgt[j * 3 + 2], 0.f) : pad4;
  for (int j = N + tid; j < npad; j += BLOCK) sm.mov[j] = pad4;
  if (tid < 24) sm.state[tid] = a.state[(size_t)inst * 24 + tid];
  __syncthreads();
  const int rot = tid & (kSub - 1);
  short* ws_a = nullptr;   // NN of the moved points in the target (direction 1)
  short* ws_b = nullptr;   // NN of the target points in the moved cloud (direction 0)
  if constexpr (PRUNE) {
    ws_b = a.nn_ws + ((size_t)inst * 2 + 0) * 4 * a.ws_stride;
    ws_a = a.nn_ws + ((size_t)inst * 2 + 1) * 4 * a.ws_stride;
    float tx0[Q], ty0[Q], tz0[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      const int i = pt_index<BLOCK, Q, OWN>(k);
      const float4 v = (i < M) ? sm.tgt[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      tx0[k] = v.x; ty0[k] = v.y; tz0[k] = v.z;
    }
    tile_boxes<BLOCK, Q, OWN>(tx0, ty0, tz0, M, mpad / kSub, sm.tbox);   // the target is static: boxes once per launch
  }
  if (tid == 0) {
    float p[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k] = (float)sm.state[k];
    Pose f;
    pose_forward(p, a.angle_base, a.trans_mode, f);
#pragma unroll
    for (int k = 0; k < 9; ++k) sm.pose[k] = f.R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) sm.pose[9 + k] = f.T[k];
  }
  __syncthreads();

#ifdef HOUV_STAMPS
  unsigned long long t_stamp_ = __builtin_readcyclecounter();
#endif
#pragma unroll 1
  for (int it = 0; it < a.n_iters; ++it) {
    float best[Q][NMET];
    int btile[Q][NMET];
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
        tile_boxes<BLOCK, Q, OWN>(mx, my, mz, N, npad / kSub, sm.mbox);   // read by sweep B after the next barriers
        pruned_now = (a.ws_valid != 0) || (it > 0);
      }
      if (pruned_now) {
        if constexpr (PRUNE) {
          pruned_sweep<BLOCK, Q, NMET, OWN>(sm.tgt, sm.tbox, mpad / kSub, mx, my, mz, ws_a, a.ws_stride, N, rot, best, btile);
        }
      } else {
        sweep<Q, NMET>(sm.tgt, mpad / kSub, mx, my, mz, best, btile);
      }
      HOUV_STAMP(1);
      epilogue<BLOCK, Q, NMET, 1, OWN>(sm, sm.tgt, mx, my, mz, best, btile, N, a.k_full, a.k_view, sx, sy, sz, ws_a,
                                  a.ws_stride HOUV_STAMP_ARG);
      HOUV_STAMP(2);
    }
    {
      // ---- sweep B: target -> moved ----
      float tx[Q], ty[Q], tz[Q];
#pragma unroll
      for (int k = 0; k < Q; ++k) {
        const int i = pt_index<BLOCK, Q, OWN>(k);
        const float4 v = (i < M) ? sm.tgt[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        tx[k] = v.x; ty[k] = v.y; tz[k] = v.z;
      }
      const bool pruned_now = PRUNE && ((a.ws_valid != 0) || (it > 0));
      if (pruned_now) {
        if constexpr (PRUNE) {
          pruned_sweep<BLOCK, Q, NMET, OWN>(sm.mov, sm.mbox, npad / kSub, tx, ty, tz, ws_b, a.ws_stride, M, rot, best, btile);
        }
      } else {
        sweep<Q, NMET>(sm.mov, npad / kSub, tx, ty, tz, best, btile);
      }
      HOUV_STAMP(3);
      epilogue<BLOCK, Q, NMET, 0, OWN>(sm, sm.mov, tx, ty, tz, best, btile, M, a.k_full, a.k_view, tx, ty, tz, ws_b,
                                  a.ws_stride HOUV_STAMP_ARG);
      HOUV_STAMP(4);
    }
    __syncthreads();
    HOUV_STAMP(5);

    // ---- per-hypothesis scalar tail: loss, closed-form gradient, Adam, next pose ----
    if (tid == 0) {
      float p[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) p[k] = (float)sm.state[k];
      Pose f;
      pose_forward(p, a.angle_base, a.trans_mode, f);
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
      const AdamScalars asc = adam_scalars(step, a.lr, a.beta1, a.beta2);   // two double pow() per step, not per parameter
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
#pragma unroll
      for (int k = 0; k < 9; ++k) sm.pose[k] = f.R[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) sm.pose[9 + k] = f.T[k];
    }
    __syncthreads();
    HOUV_STAMP(6);
  }
  if (tid < 24) a.state[(size_t)inst * 24 + tid] = sm.state[tid];
}

template <int BLOCK, int Q, bool PRUNE, int OWN>
int launch(const SolveArgs& a, int use_views, hipStream_t s) {
  const size_t bytes = smem_bytes(a.N, a.M, BLOCK);
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
extern "C" int houv_debug_read_prune_stats(unsigned long long* host_out, int reset) {
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(houv::g_prune_stat), sizeof(unsigned long long) * 8) != hipSuccess) return 0;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(houv::g_prune_stat), z, sizeof(z)) != hipSuccess) return 0;
  }
  return 1;
}
extern "C" int houv_debug_read_stamps(unsigned long long* host_out, int reset) {
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(houv::g_stamp), sizeof(unsigned long long) * 16) != hipSuccess) return 0;
  if (reset) {
    unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(houv::g_stamp), z, sizeof(z)) != hipSuccess) return 0;
  }
  return 1;
}
#endif

// pruned mode: consecutive points a lane owns per chunk (pt_index), for kernels with Q = 2 / Q = 4 points per lane
#ifndef HOUV_PRUNE_OWN
#define HOUV_PRUNE_OWN 1
#endif
constexpr int kOwn2 = HOUV_PRUNE_OWN < 2 ? HOUV_PRUNE_OWN : 2;
constexpr int kOwn4 = HOUV_PRUNE_OWN;

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
  SolveArgs a{src, tgt, P, N, M, K, state, steps_done, n_iters, angle_base, trans_mode, f64_params, k_full, k_view,
              lr, beta1, beta2, eps, loss_scale, out_score, out_loss, out_R, out_T, out_grad, out_cd, nn_ws, ws_valid,
              ws_stride};
  hipStream_t s = (hipStream_t)stream;
  const int mx = N > M ? N : M;
  if (smem_bytes(N, M, 1024) > 160 * 1024) {
    set_error("%s: clouds of %d + %d points do not fit in 160 KiB of LDS", who, N, M);
    return 0;
  }
  if (prune) {
    if (mx > 2048 || !nn_ws || ws_stride < mx) {
      set_error("%s: pruned mode needs clouds of <= 2048 points (64 sub-tiles) and a workspace (ws_stride >= max(N,M))", who);
      return 0;
    }
    if (ws_valid < 0) {   // test aid: the brute-force sweep under the pruned kernel's point ownership / summation order
      a.ws_valid = 0;
      if (mx <= 256) return launch<256, 1, false, 1>(a, use_views, s);
      if (mx <= 512) return launch<256, 2, false, kOwn2>(a, use_views, s);
      if (mx <= 768) return launch<256, 3, false, 1>(a, use_views, s);
      if (mx <= 1024) return launch<256, 4, false, kOwn4>(a, use_views, s);
      if (mx <= 1536) return launch<512, 3, false, 1>(a, use_views, s);
      return launch<512, 4, false, kOwn4>(a, use_views, s);
    }
    if (mx <= 256) return launch<256, 1, true, 1>(a, use_views, s);
    if (mx <= 512) return launch<256, 2, true, kOwn2>(a, use_views, s);
    if (mx <= 768) return launch<256, 3, true, 1>(a, use_views, s);
    if (mx <= 1024) return launch<256, 4, true, kOwn4>(a, use_views, s);
    if (mx <= 1536) return launch<512, 3, true, 1>(a, use_views, s);
    return launch<512, 4, true, kOwn4>(a, use_views, s);
  }
  if (mx <= 256) return launch<256, 1, false, 1>(a, use_views, s);
  // Q = 3 points per lane covers the sizes between the powers of two without idle lanes (768, 1536, 3072)
  if (mx <= 512) return launch<256, 2, false, 1>(a, use_views, s);
  if (mx <= 768) return launch<256, 3, false, 1>(a, use_views, s);
  if (mx <= 1024) return launch<256, 4, false, 1>(a, use_views, s);
  if (mx <= 1536) return launch<512, 3, false, 1>(a, use_views, s);
  if (mx <= 2048) return launch<512, 4, false, 1>(a, use_views, s);
  if (mx <= 3072) return launch<1024, 3, false, 1>(a, use_views, s);
  if (mx <= 4096) return launch<1024, 4, false, 1>(a, use_views, s);
  set_error("%s: clouds larger than 4096 points are not supported (N=%d M=%d)", who, N, M);
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

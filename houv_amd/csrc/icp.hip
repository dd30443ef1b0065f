// icp.hip -- batched point-to-point ICP refinement for gfx950 (BASELINE configs[3]).
//
// What it replaces: the reference's ICP baseline calls Open3D 0.9.0 per pair from Python
// (registration/train_ICP.py:137-153):
//     registration_icp(source, target, max_correspondence_distance = 0.02, init,
//                      TransformationEstimationPointToPoint(), ICPConvergenceCriteria(max_iteration = 500))
// Open3D is a third-party dependency that is NOT under /root/reference and not installed here, so this kernel
// restates its published algorithm (open3d/registration/Registration.cpp RegistrationICP +
// TransformationEstimationPointToPoint = Eigen::umeyama without scaling):
//     result = correspondences(T source, target)          # per source point: NN in target if dist < max_dist
//     repeat up to max_iteration times:
//         update = Kabsch(corresponded pairs);  T = update * T
//         new = correspondences(T source, target)
//         stop when |fitness - new.fitness| < relative_fitness and |rmse - new.rmse| < relative_rmse (1e-6 both)
// PARITY UNPINNED: no Open3D output is available to check against; tests compare with oracle/icp_ref.py, a numpy
// restatement of the same published algorithm.
//
// One workgroup per pair: target cloud resident in LDS, this lane's source points in registers, the same
// LDS-broadcast single-metric sweep + deferred exact arg-min as the Chamfer kernels, two workgroup reductions
// (means, centred covariance) and a register-resident Jacobi SVD per iteration.  No HBM traffic inside the loop.
#include "../../include/houv_hip.h"
#include "houv_common.h"
#include "houv_sweep.h"

namespace houv {
namespace {

struct IcpArgs {
  const float* src;
  const float* tgt;
  int P, N, M;
  const float* init;   // [P,16] row-major 4x4, or null = identity
  float max_dist2;
  int max_iter;
  float rel_fitness, rel_rmse;
  float* out_T;        // [P,16]
  float* out_fitness;  // [P]
  float* out_rmse;     // [P]
  int* out_iters;      // [P]
};

__host__ __device__ inline size_t icp_smem_bytes(int M, int block) {
  const int mpad = (M + kSub - 1) / kSub * kSub;
  return (size_t)mpad * 16 + (size_t)(block / 64) * kAccStride * 4 + kAccStride * 4 + 16 * 4 + 64;
}

template <int BLOCK, int Q>
__global__ __launch_bounds__(BLOCK) void icp_kernel(IcpArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int N = a.N, M = a.M, tid = threadIdx.x, pair = blockIdx.x;
  const int mpad = (M + kSub - 1) / kSub * kSub;
  float4* s_tgt = reinterpret_cast<float4*>(smem_raw);
  float* s_red = reinterpret_cast<float*>(s_tgt + mpad);           // [NW][kAccStride]
  float* s_out = s_red + (BLOCK / 64) * kAccStride;                // [kAccStride]
  float* s_T = s_out + kAccStride;                                 // [12] current R (row-major) | t ; [12] = stop flag
  const float* __restrict__ src = a.src + (size_t)pair * N * 3;
  const float* __restrict__ tgt = a.tgt + (size_t)pair * M * 3;
  const float4 pad4 = make_float4(INFINITY, INFINITY, INFINITY, 0.f);
  for (int j = tid; j < mpad; j += BLOCK) s_tgt[j] = (j < M) ? make_float4(tgt[j * 3], tgt[j * 3 + 1], tgt[j * 3 + 2], 0.f) : pad4;
  if (tid < 12) {
    float v = (tid == 0 || tid == 4 || tid == 8) ? 1.f : 0.f;      // identity R, zero t
    if (a.init) {
      const float* T0 = a.init + (size_t)pair * 16;
      v = (tid < 9) ? T0[(tid / 3) * 4 + (tid % 3)] : T0[(tid - 9) * 4 + 3];
    }
    s_T[tid] = v;
  }
  float sx[Q], sy[Q], sz[Q];
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    const int i = k * BLOCK + tid;
    const bool ok = i < N;
    sx[k] = ok ? src[i * 3 + 0] : 0.f;
    sy[k] = ok ? src[i * 3 + 1] : 0.f;
    sz[k] = ok ? src[i * 3 + 2] : 0.f;
  }
  __syncthreads();

  const int rot = tid & (kSub - 1);
  float prev_fit = 0.f, prev_rmse = 0.f, fit = 0.f, rmse = 0.f;
  int it = 0;
#pragma unroll 1
  for (;; ++it) {
    float R[9], T[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = s_T[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) T[i] = s_T[9 + i];
    float px[Q], py[Q], pz[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      px[k] = __builtin_fmaf(sz[k], R[2], __builtin_fmaf(sy[k], R[1], sx[k] * R[0])) + T[0];
      py[k] = __builtin_fmaf(sz[k], R[5], __builtin_fmaf(sy[k], R[4], sx[k] * R[3])) + T[1];
      pz[k] = __builtin_fmaf(sz[k], R[8], __builtin_fmaf(sy[k], R[7], sx[k] * R[6])) + T[2];
    }
    float best[Q][1];
    int btile[Q][1];
    sweep<Q, 1>(s_tgt, mpad / kTrk, px, py, pz, best, btile);
    // correspondences: NN strictly inside the search radius
    float nx[Q], ny[Q], nz[Q];
    bool in[Q];
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      int jn;
      const float4 nn = recover_nn<0, 4>(s_tgt + btile[k][0] * kTrk, px[k], py[k], pz[k], best[k][0], rot, jn);
      nx[k] = nn.x; ny[k] = nn.y; nz[k] = nn.z;
      in[k] = ((k * BLOCK + tid) < N) && (best[k][0] < a.max_dist2);
      if (in[k]) {
        acc[0] += 1.f; acc[1] += px[k]; acc[2] += py[k]; acc[3] += pz[k];
        acc[4] += nx[k]; acc[5] += ny[k]; acc[6] += nz[k]; acc[7] += best[k][0];
      }
    }
    block_sum<BLOCK, 8>(acc, s_red, s_out);
    __syncthreads();
    const float cnt = s_out[0];
    const float inv = cnt > 0.f ? 1.0f / cnt : 0.f;
    const float mp[3] = {s_out[1] * inv, s_out[2] * inv, s_out[3] * inv};
    const float mn[3] = {s_out[4] * inv, s_out[5] * inv, s_out[6] * inv};
    fit = cnt / (float)N;
    rmse = cnt > 0.f ? sqrtf(s_out[7] * inv) : 0.f;
    // Open3D's stop test compares the result before and after an update
    bool stop = (it > 0) && (fabsf(prev_fit - fit) < a.rel_fitness) && (fabsf(prev_rmse - rmse) < a.rel_rmse);
    stop = stop || (it >= a.max_iter) || !(cnt > 0.f);
    if (stop) break;   // uniform: every thread computed the same values from LDS
    prev_fit = fit;
    prev_rmse = rmse;
    float h[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      if (in[k]) {
        const float ax = px[k] - mp[0], ay = py[k] - mp[1], az = pz[k] - mp[2];
        const float bx = nx[k] - mn[0], by = ny[k] - mn[1], bz = nz[k] - mn[2];
        h[0] += ax * bx; h[1] += ax * by; h[2] += ax * bz;
        h[3] += ay * bx; h[4] += ay * by; h[5] += ay * bz;
        h[6] += az * bx; h[7] += az * by; h[8] += az * bz;
      }
    }
    block_sum<BLOCK, 9>(h, s_red, s_out);
    __syncthreads();
    if (tid == 0) {
      float H[9], Ru[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) H[i] = s_out[i];
      kabsch_rotation<float>(H, Ru);          // maps (moved source) onto (target): R = V diag(1,1,det) U^T
      float tu[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) tu[i] = mn[i] - (Ru[i * 3 + 0] * mp[0] + Ru[i * 3 + 1] * mp[1] + Ru[i * 3 + 2] * mp[2]);
      float Rn[9], tn[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) Rn[i * 3 + j] = Ru[i * 3 + 0] * R[0 * 3 + j] + Ru[i * 3 + 1] * R[1 * 3 + j] + Ru[i * 3 + 2] * R[2 * 3 + j];
        tn[i] = Ru[i * 3 + 0] * T[0] + Ru[i * 3 + 1] * T[1] + Ru[i * 3 + 2] * T[2] + tu[i];
      }
#pragma unroll
      for (int i = 0; i < 9; ++i) s_T[i] = Rn[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) s_T[9 + i] = tn[i];
    }
    __syncthreads();
  }
  if (tid == 0) {
    float* o = a.out_T + (size_t)pair * 16;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) o[i * 4 + j] = s_T[i * 3 + j];
      o[i * 4 + 3] = s_T[9 + i];
    }
    o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
    if (a.out_fitness) a.out_fitness[pair] = fit;
    if (a.out_rmse) a.out_rmse[pair] = rmse;
    if (a.out_iters) a.out_iters[pair] = it;
  }
}

template <int BLOCK, int Q>
int launch_icp(const IcpArgs& a, hipStream_t s) {
  const size_t bytes = icp_smem_bytes(a.M, BLOCK);
  hipError_t e = hipFuncSetAttribute((const void*)icp_kernel<BLOCK, Q>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) {
    set_error("houv_icp_refine: cannot reserve %zu B of LDS: %s", bytes, hipGetErrorString(e));
    return 0;
  }
  icp_kernel<BLOCK, Q><<<a.P, BLOCK, bytes, s>>>(a);
  return check_launch("houv_icp_refine") ? 1 : 0;
}

}  // namespace
}  // namespace houv

extern "C" int houv_icp_refine(const float* src, const float* tgt, int P, int N, int M, const float* init_or_null,
                               float max_correspondence_distance, int max_iteration, float relative_fitness,
                               float relative_rmse, float* out_T, float* out_fitness, float* out_rmse, int* out_iters,
                               void* stream) {
  using namespace houv;
  if (P < 0 || N <= 0 || M <= 0 || max_iteration < 0 || !(max_correspondence_distance > 0.f)) {
    set_error("houv_icp_refine: bad argument P=%d N=%d M=%d max_iteration=%d max_dist=%g", P, N, M, max_iteration,
              (double)max_correspondence_distance);
    return 0;
  }
  if (P == 0) return 1;
  if (!src || !tgt || !out_T) {
    set_error("houv_icp_refine: null pointer");
    return 0;
  }
  if (icp_smem_bytes(M, 1024) > 160 * 1024 || N > 8192) {
    set_error("houv_icp_refine: clouds too large for the LDS-resident kernel (N=%d <= 8192, M=%d <= ~10000)", N, M);
    return 0;
  }
  IcpArgs a{src, tgt, P, N, M, init_or_null, max_correspondence_distance * max_correspondence_distance, max_iteration,
            relative_fitness, relative_rmse, out_T, out_fitness, out_rmse, out_iters};
  hipStream_t s = (hipStream_t)stream;
  if (N <= 256) return launch_icp<256, 1>(a, s);
  if (N <= 512) return launch_icp<256, 2>(a, s);
  if (N <= 1024) return launch_icp<256, 4>(a, s);
  if (N <= 2048) return launch_icp<512, 4>(a, s);
  if (N <= 4096) return launch_icp<1024, 4>(a, s);
  return launch_icp<1024, 8>(a, s);
}

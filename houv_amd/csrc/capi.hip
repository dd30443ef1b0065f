// capi.hip -- error plumbing + the small per-hypothesis ops of libhouv_hip.so.
#include <stdarg.h>
#include <string.h>

#include "../../include/houv_hip.h"
#include "houv_common.h"

namespace houv {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  // the reference prints kernel errors and carries on (chamfer3D.cu:147); keep the message visible too
  fprintf(stderr, "[houv_hip] %s\n", g_err);
}

DebugKnobs g_debug;

namespace {

// HOUV.forward (registration/models/houv.py:94-103): one thread per hypothesis builds R,T; the
// optional cloud transform is a second, coalesced pass.
__global__ void pose_kernel(const float* __restrict__ params, int n, int angle_base, int trans_mode,
                            float* __restrict__ R, float* __restrict__ T) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float p[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) p[k] = params[i * 8 + k];
  Pose f;
  pose_forward(p, angle_base, trans_mode, f);
#pragma unroll
  for (int k = 0; k < 9; ++k) R[i * 9 + k] = f.R[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) T[i * 3 + k] = f.T[k];
}

__global__ void move_kernel(const float* __restrict__ src, const float* __restrict__ R, const float* __restrict__ T,
                            size_t total, int N, float* __restrict__ moved) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const size_t b = e / N;
    const float* r = R + b * 9;
    const float x = src[e * 3 + 0], y = src[e * 3 + 1], z = src[e * 3 + 2];
    // src @ R^T + T  (houv.py:102): row i of R dotted with the point, k-ordered like bmm
    moved[e * 3 + 0] = __builtin_fmaf(z, r[2], __builtin_fmaf(y, r[1], x * r[0])) + T[b * 3 + 0];
    moved[e * 3 + 1] = __builtin_fmaf(z, r[5], __builtin_fmaf(y, r[4], x * r[3])) + T[b * 3 + 1];
    moved[e * 3 + 2] = __builtin_fmaf(z, r[8], __builtin_fmaf(y, r[7], x * r[6])) + T[b * 3 + 2];
  }
}

}  // namespace
}  // namespace houv

extern "C" int houv_abi_version(void) { return HOUV_ABI_VERSION; }

extern "C" const char* houv_last_error(void) { return houv::g_err; }

extern "C" int houv_debug_set(const char* name, long long value) {
  using namespace houv;
  const struct { const char* name; std::atomic<int>* knob; long long lo, hi; } ints[] = {
      {"solve_predict", &g_debug.pred_mode, 0, 2},   {"prune_refresh", &g_debug.ws_refresh, 0, 1 << 30},
      {"prune_cap_slack", &g_debug.prune_cap_slack, -1, 64}, {"prune_owner_walk", &g_debug.prune_owner_walk, 0, 1}, {"prune_min_points", &g_debug.prune_min_points, 257, 2049},
      {"chamfer_direct", &g_debug.chamfer_direct, 0, 1}, {"chamfer_q", &g_debug.chamfer_q, 1, 8},
      {"gemm_4w", &g_debug.gemm_4w, 0, 1},           {"gemm_guarded", &g_debug.gemm_guarded, 0, 1},
      {"gemm_split", &g_debug.gemm_split, 0, 6},     {"attn_split", &g_debug.attn_split, 0, 1},
      {"knn_split", &g_debug.knn_split, 0, 1}};
  if (name && !strcmp(name, "solve_stats")) {
    g_debug.stats = (unsigned long long)value;
    return 1;
  }
  for (const auto& k : ints)
    if (name && !strcmp(name, k.name) && value >= k.lo && value <= k.hi) {
      *k.knob = (int)value;
      return 1;
    }
  set_error("houv_debug_set: unknown switch or value out of range: %s = %lld", name ? name : "(null)", value);
  return 0;
}

#ifndef HOUV_BUILD_ID
#define HOUV_BUILD_ID "unknown"
#endif
extern "C" const char* houv_build_id(void) { return HOUV_BUILD_ID; }

extern "C" int houv_pose_forward(const float* params, int n, int angle_base, int trans_mode, const float* src, int N,
                                 float* R, float* T, float* moved, void* stream) {
  using namespace houv;
  if (n < 0 || !params || !R || !T || angle_base < 0 || angle_base > 3 || trans_mode < 0 || trans_mode > 1) {
    set_error("houv_pose_forward: bad argument");
    return 0;
  }
  if (n == 0) return 1;
  hipStream_t s = (hipStream_t)stream;
  pose_kernel<<<(n + 255) / 256, 256, 0, s>>>(params, n, angle_base, trans_mode, R, T);
  if (src && moved && N > 0) {
    const size_t total = (size_t)n * N;
    size_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    move_kernel<<<(unsigned)blocks, 256, 0, s>>>(src, R, T, total, N, moved);
  }
  return check_launch("houv_pose_forward") ? 1 : 0;
}

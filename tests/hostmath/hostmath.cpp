// Host build of houv_amd/csrc/houv_math.h for CPU unit tests (test infrastructure only:
// lets the exact code the HIP kernels inline be checked against the oracle without a GPU).
#include "../../houv_amd/csrc/houv_math.h"
#include <string.h>

extern "C" {

// params[n,8] -> R[n,9], T[n,3]
void hm_pose_forward(const float* params, int n, int angle_base, int trans_mode, float* R, float* T) {
  for (int i = 0; i < n; ++i) {
    houv::Pose f;
    houv::pose_forward(params + 8 * i, angle_base, trans_mode, f);
    memcpy(R + 9 * i, f.R, sizeof(f.R));
    memcpy(T + 3 * i, f.T, sizeof(f.T));
  }
}

// gT[n,3], M[n,9] -> g[n,8]
void hm_pose_backward(const float* params, int n, int angle_base, int trans_mode, const float* gT, const float* M, float* g) {
  for (int i = 0; i < n; ++i) {
    houv::Pose f;
    houv::pose_forward(params + 8 * i, angle_base, trans_mode, f);
    houv::pose_backward(f, trans_mode, gT + 3 * i, M + 9 * i, g + 8 * i);
  }
}

void hm_adam_f32(float* p, float* m, float* v, const float* g, int n, int step, double lr, double b1, double b2, double eps) {
  for (int i = 0; i < n; ++i) houv::adam_step<float>(p[i], m[i], v[i], g[i], step, lr, b1, b2, eps);
}

void hm_adam_f64(double* p, double* m, double* v, const double* g, int n, int step, double lr, double b1, double b2, double eps) {
  for (int i = 0; i < n; ++i) houv::adam_step<double>(p[i], m[i], v[i], g[i], step, lr, b1, b2, eps);
}

void hm_svd3x3_f32(const float* H, int n, float* U, float* S, float* V) {
  for (int i = 0; i < n; ++i) houv::svd3x3<float>(H + 9 * i, U + 9 * i, S + 3 * i, V + 9 * i);
}

void hm_kabsch_rotation_f32(const float* H, int n, float* R) {
  for (int i = 0; i < n; ++i) houv::kabsch_rotation<float>(H + 9 * i, R + 9 * i);
}
}

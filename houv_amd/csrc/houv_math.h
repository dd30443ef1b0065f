// houv_math.h -- small per-instance math shared by the HIP kernels.
//
// Everything here is `HOUV_HD inline`, written against plain C++ so the very
// same code can be compiled for the host by tests/hostmath (g++) and unit
// tested against the oracle without a GPU.  Citations are to the reference
// tree (Dizzy-cell/HOUV):
//   registration/models/houv.py:69-103      Rodrigues + translation reparam
//   registration/train_utils.py:397-407      the `solve` twin (sigma = sin(s pi))
//   registration/model_utils.py:220-255      SVDHead (Kabsch)
//   torch.optim.Adam (reference: houv.py:118, train_utils.py:389)
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define HOUV_HD __host__ __device__
#else
#define HOUV_HD
#endif

namespace houv {

// houv.py:19 -- pi = acos(0)*2 evaluated in fp32 (= 3.1415927410125732), then used as a Python float
// that multiplies fp32 tensors, i.e. rounded back to fp32 at each use.
constexpr float kPiF = 3.14159274101257324f;

enum TransMode { kTransHouv = 0, kTransSolve = 1 };

HOUV_HD inline float tsqrt(float x) { return sqrtf(x); }     // correctly rounded (no -ffast-math)
HOUV_HD inline double tsqrt(double x) { return sqrt(x); }

// Parameter block layout (8 scalars per hypothesis): V[0..2], a, c[0..2], s
struct Pose {
  float R[9];   // row-major
  float T[3];
  // forward intermediates kept for the backward pass
  float u[3], inv_vnorm, sin_t, cos_t, chat[3], inv_cnorm, sigma, a, s;
};

// houv.py:94-103 (HOUV.forward) / train_utils.py:403-407.
HOUV_HD inline void pose_forward(const float p[8], int angle_base, int trans_mode, Pose& o) {
  const float vx = p[0], vy = p[1], vz = p[2];
  const float vn = sqrtf(vx * vx + vy * vy + vz * vz);           // houv.py:71
  o.inv_vnorm = 1.0f / vn;
  o.u[0] = vx / vn; o.u[1] = vy / vn; o.u[2] = vz / vn;
  o.a = p[3];
  // theta = sin(a*pi)*pi/8 + pi/8 + base*pi/4   (houv.py:96), all fp32 roundings as torch does them
  const float theta = sinf(p[3] * kPiF) * kPiF / 8.0f + kPiF / 8.0f + (float)angle_base * kPiF / 4.0f;
  const float st = sinf(theta), ct = cosf(theta);
  o.sin_t = st; o.cos_t = ct;
  const float ux = o.u[0], uy = o.u[1], uz = o.u[2];
  // A = [u]x (houv.py:78-83);  A*A = u u^T - I (|u|=1) but the reference multiplies it out: keep the product form
  const float A[9] = {0.f, -uz, uy, uz, 0.f, -ux, -uy, ux, 0.f};
  float AA[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      AA[i * 3 + j] = A[i * 3 + 0] * A[0 * 3 + j] + A[i * 3 + 1] * A[1 * 3 + j] + A[i * 3 + 2] * A[2 * 3 + j];
  const float omc = 1.0f - ct;
  for (int i = 0; i < 9; ++i) o.R[i] = ((i % 4 == 0) ? 1.0f : 0.0f) + st * A[i] + omc * AA[i];   // houv.py:85
  o.s = p[7];
  const float sp = sinf(p[7] * kPiF);
  o.sigma = (trans_mode == kTransHouv) ? (sp * 0.125f + 0.125f) : (sp * 1.0f);      // houv.py:99 / train_utils.py:404
  const float cx = p[4], cy = p[5], cz = p[6];
  const float cn = sqrtf(cx * cx + cy * cy + cz * cz);           // houv.py:89
  o.inv_cnorm = 1.0f / cn;
  o.chat[0] = cx / cn; o.chat[1] = cy / cn; o.chat[2] = cz / cn;
  o.T[0] = o.chat[0] * o.sigma; o.T[1] = o.chat[1] * o.sigma; o.T[2] = o.chat[2] * o.sigma;
}

// Closed-form backward of pose_forward (what autograd does through houv.py:69-103).
//   gT[3]  = dL/dT = sum_i G_i
//   M[9]   = dL/dR = sum_i G_i p_i^T   (row-major, p = un-moved source point)
// -> g[8] = dL/d(V, a, c, s)
HOUV_HD inline void pose_backward(const Pose& f, int trans_mode, const float gT[3], const float M[9], float g[8]) {
  const float ux = f.u[0], uy = f.u[1], uz = f.u[2];
  const float A[9] = {0.f, -uz, uy, uz, 0.f, -ux, -uy, ux, 0.f};
  float AA[9], MAt[9], AtM[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      AA[i * 3 + j] = A[i * 3 + 0] * A[0 * 3 + j] + A[i * 3 + 1] * A[1 * 3 + j] + A[i * 3 + 2] * A[2 * 3 + j];
      // (M A^T)_ij = sum_k M_ik A_jk ;  (A^T M)_ij = sum_k A_ki M_kj
      MAt[i * 3 + j] = M[i * 3 + 0] * A[j * 3 + 0] + M[i * 3 + 1] * A[j * 3 + 1] + M[i * 3 + 2] * A[j * 3 + 2];
      AtM[i * 3 + j] = A[0 * 3 + i] * M[0 * 3 + j] + A[1 * 3 + i] * M[1 * 3 + j] + A[2 * 3 + i] * M[2 * 3 + j];
    }
  // dL/dtheta = <M, cos A + sin A^2>
  float dth = 0.f;
  for (int i = 0; i < 9; ++i) dth += M[i] * (f.cos_t * A[i] + f.sin_t * AA[i]);
  // theta = sin(a pi) pi/8 + ...  ->  dtheta/da = cos(a pi) * pi * pi/8
  g[3] = dth * cosf(f.a * kPiF) * kPiF * kPiF / 8.0f;
  // dL/dA = sin M + (1-cos)(M A^T + A^T M)
  float Qm[9];
  const float omc = 1.0f - f.cos_t;
  for (int i = 0; i < 9; ++i) Qm[i] = f.sin_t * M[i] + omc * (MAt[i] + AtM[i]);
  // A01=-u2 A02=u1 A10=u2 A12=-u0 A20=-u1 A21=u0
  const float du0 = Qm[7] - Qm[5], du1 = Qm[2] - Qm[6], du2 = Qm[3] - Qm[1];
  // u = v/|v| -> dv = (I - u u^T)/|v| du
  const float dot_u = ux * du0 + uy * du1 + uz * du2;
  g[0] = (du0 - ux * dot_u) * f.inv_vnorm;
  g[1] = (du1 - uy * dot_u) * f.inv_vnorm;
  g[2] = (du2 - uz * dot_u) * f.inv_vnorm;
  // T = sigma * chat
  const float dot_c = f.chat[0] * gT[0] + f.chat[1] * gT[1] + f.chat[2] * gT[2];
  g[4] = f.sigma * (gT[0] - f.chat[0] * dot_c) * f.inv_cnorm;
  g[5] = f.sigma * (gT[1] - f.chat[1] * dot_c) * f.inv_cnorm;
  g[6] = f.sigma * (gT[2] - f.chat[2] * dot_c) * f.inv_cnorm;
  const float dsig = (trans_mode == kTransHouv) ? (cosf(f.s * kPiF) * kPiF * 0.125f) : (cosf(f.s * kPiF) * kPiF);
  g[7] = dot_c * dsig;
}

// One torch.optim.Adam step (no weight decay / amsgrad) on one scalar, arithmetic type T = float
// (HOUV module, fp32 parameters: houv.py:54-61,118) or double (`solve` twin keeps float64 leaves:
// train_utils.py:381-389).  `step` is 1-based.  Mirrors torch's single-tensor formulation:
//   m += (g-m)(1-b1);  v = v b2 + (1-b2) g g;  p -= (lr/(1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// The step-dependent scalars (torch computes them in Python doubles): shared by all parameters of a step.
struct AdamScalars {
  double step_size, bc2_sqrt;
};
HOUV_HD inline AdamScalars adam_scalars(int step, double lr, double b1, double b2) {
  const double bc1 = 1.0 - pow(b1, (double)step);
  const double bc2 = 1.0 - pow(b2, (double)step);
  return AdamScalars{lr / bc1, sqrt(bc2)};
}
template <typename T>
HOUV_HD inline void adam_step(T& p, T& m, T& v, T g, const AdamScalars& sc, double b1, double b2, double eps) {
  m = m + (g - m) * (T)(1.0 - b1);
  v = v * (T)b2 + ((T)(1.0 - b2) * g) * g;
  const T denom = tsqrt(v) / (T)sc.bc2_sqrt + (T)eps;
  p = p - (T)sc.step_size * (m / denom);
}
template <typename T>
HOUV_HD inline void adam_step(T& p, T& m, T& v, T g, int step, double lr, double b1, double b2, double eps) {
  adam_step<T>(p, m, v, g, adam_scalars(step, lr, b1, b2), b1, b2, eps);
}

// ---------------------------------------------------------------------------------------------
// 3x3 SVD by one-sided (Hestenes) Jacobi, register resident.  H = U diag(S) V^T, S sorted
// descending like torch.svd (model_utils.py:233).
// ---------------------------------------------------------------------------------------------
template <typename T>
HOUV_HD inline void svd3x3(const T H[9], T U[9], T S[3], T V[9]) {
  T B[9];
  for (int i = 0; i < 9; ++i) { B[i] = H[i]; V[i] = (i % 4 == 0) ? (T)1 : (T)0; }
  const int P[3] = {0, 0, 1}, Qi[3] = {1, 2, 2};
  for (int sweep = 0; sweep < 12; ++sweep) {
    T off = 0;
    for (int r = 0; r < 3; ++r) {
      const int p = P[r], q = Qi[r];
      const T a = B[0 * 3 + p] * B[0 * 3 + p] + B[1 * 3 + p] * B[1 * 3 + p] + B[2 * 3 + p] * B[2 * 3 + p];
      const T b = B[0 * 3 + q] * B[0 * 3 + q] + B[1 * 3 + q] * B[1 * 3 + q] + B[2 * 3 + q] * B[2 * 3 + q];
      const T g = B[0 * 3 + p] * B[0 * 3 + q] + B[1 * 3 + p] * B[1 * 3 + q] + B[2 * 3 + p] * B[2 * 3 + q];
      // converged for this pair when |g| <= eps * |b_p||b_q|
      const T eps2 = sizeof(T) == 4 ? (T)1e-14 : (T)1e-31;
      if (g * g <= eps2 * a * b) continue;
      off += (T)1;
      const T zeta = (b - a) / ((T)2 * g);
      const T az = zeta < 0 ? -zeta : zeta;
      const T t = (zeta < 0 ? (T)-1 : (T)1) / (az + tsqrt((T)1 + zeta * zeta));
      const T c = (T)1 / tsqrt((T)1 + t * t);
      const T s = c * t;
      for (int i = 0; i < 3; ++i) {
        const T bp = B[i * 3 + p], bq = B[i * 3 + q];
        B[i * 3 + p] = c * bp - s * bq;
        B[i * 3 + q] = s * bp + c * bq;
        const T vp = V[i * 3 + p], vq = V[i * 3 + q];
        V[i * 3 + p] = c * vp - s * vq;
        V[i * 3 + q] = s * vp + c * vq;
      }
    }
    if (off == (T)0) break;
  }
  for (int j = 0; j < 3; ++j)
    S[j] = tsqrt(B[0 * 3 + j] * B[0 * 3 + j] + B[1 * 3 + j] * B[1 * 3 + j] + B[2 * 3 + j] * B[2 * 3 + j]);
  // sort columns by S descending (3-element network)
  auto swapcol = [&](int x, int y) {
    T ts = S[x]; S[x] = S[y]; S[y] = ts;
    for (int i = 0; i < 3; ++i) {
      T tb = B[i * 3 + x]; B[i * 3 + x] = B[i * 3 + y]; B[i * 3 + y] = tb;
      T tv = V[i * 3 + x]; V[i * 3 + x] = V[i * 3 + y]; V[i * 3 + y] = tv;
    }
  };
  if (S[0] < S[1]) swapcol(0, 1);
  if (S[1] < S[2]) swapcol(1, 2);
  if (S[0] < S[1]) swapcol(0, 1);
  // U columns = B columns / S; rank-deficient columns are completed to a right-handed orthonormal set
  const T tiny = S[0] * (sizeof(T) == 4 ? (T)1e-6 : (T)1e-13);
  for (int j = 0; j < 2; ++j) {
    if (S[j] > tiny && S[j] > (T)0) {
      for (int i = 0; i < 3; ++i) U[i * 3 + j] = B[i * 3 + j] / S[j];
    } else if (j == 0) {
      U[0] = 1; U[3] = 0; U[6] = 0;
    } else {
      // any unit vector orthogonal to u0
      const T x = U[0], y = U[3], z = U[6];
      const T ax = x < 0 ? -x : x, ay = y < 0 ? -y : y, az = z < 0 ? -z : z;
      T e[3] = {0, 0, 0};
      if (ax <= ay && ax <= az) e[0] = 1; else if (ay <= az) e[1] = 1; else e[2] = 1;
      T w[3] = {y * e[2] - z * e[1], z * e[0] - x * e[2], x * e[1] - y * e[0]};
      const T n = tsqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
      U[1] = w[0] / n; U[4] = w[1] / n; U[7] = w[2] / n;
    }
  }
  if (S[2] > tiny && S[2] > (T)0) {
    for (int i = 0; i < 3; ++i) U[i * 3 + 2] = B[i * 3 + 2] / S[2];
  } else {
    U[2] = U[3] * U[7] - U[6] * U[4];
    U[5] = U[6] * U[1] - U[0] * U[7];
    U[8] = U[0] * U[4] - U[3] * U[1];
  }
}

template <typename T>
HOUV_HD inline T det3(const T m[9]) {
  return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// model_utils.py:232-240: r = v u^T; if det(r) < 0 flip the last column of v and recompute.
template <typename T>
HOUV_HD inline void kabsch_rotation(const T H[9], T R[9]) {
  T U[9], S[3], V[9];
  svd3x3<T>(H, U, S, V);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i * 3 + j] = V[i * 3 + 0] * U[j * 3 + 0] + V[i * 3 + 1] * U[j * 3 + 1] + V[i * 3 + 2] * U[j * 3 + 2];
  if (det3<T>(R) < (T)0) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) R[i * 3 + j] = V[i * 3 + 0] * U[j * 3 + 0] + V[i * 3 + 1] * U[j * 3 + 1] - V[i * 3 + 2] * U[j * 3 + 2];
  }
}

}  // namespace houv

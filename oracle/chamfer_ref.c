/* chamfer_ref.c -- CPU oracle, TEST INFRASTRUCTURE ONLY (never linked into or called by houv_amd/).
 *
 * Plain-C restatement of the arithmetic of the reference's CUDA Chamfer op
 *   utils/metrics/CD/chamfer3D/chamfer3D.cu:12-134  (NmDistanceKernel)   -> oracle_nm_distance
 *   utils/metrics/CD/chamfer3D/chamfer3D.cu:155-174 (NmDistanceGradKernel) -> oracle_nm_distance_grad
 * i.e. fp32 DIRECT differences d = x2*x2 + y2*y2 + z2*z2 with (x2,y2,z2) = reference point - query point
 * (:31-34), strict `<` so the lowest index wins ties (:35, :44 ...; tiles are visited in ascending order and
 * a later tile only replaces on strictly smaller, :126).  nvcc contracts the sum into
 * fma(z2,z2, fma(y2,y2, x2*x2)); that contraction is written out with fmaf here and this file is compiled with
 * -ffp-contract=off, so the HIP kernels can be compared BIT FOR BIT (dist bits and indices).
 *
 * Parity status: pinned -- tests/test_oracle_golden.py checks it against golden vectors G1 captured from the
 * reference's own pure-torch Chamfer (the op's parity oracle in utils/metrics/CD/unit_test.py:22-33): indices
 * exactly equal, distances to fp32 rounding of the float64 value.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

/* One direction: for every query point of xyz[b,n,3] the squared distance to, and index of, its nearest point in
 * xyz2[b,m,3].  Matches the kernel's outputs result[b,n], result_i[b,n]. */
void oracle_nm_distance(int b, int n, const float* xyz, int m, const float* xyz2, float* result, int32_t* result_i) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int i = 0; i < b; ++i) {
    for (int j = 0; j < n; ++j) {
      const float x1 = xyz[((size_t)i * n + j) * 3 + 0];
      const float y1 = xyz[((size_t)i * n + j) * 3 + 1];
      const float z1 = xyz[((size_t)i * n + j) * 3 + 2];
      float best = 0.f;
      int best_i = 0;
      for (int k = 0; k < m; ++k) {
        const float x2 = xyz2[((size_t)i * m + k) * 3 + 0] - x1;
        const float y2 = xyz2[((size_t)i * m + k) * 3 + 1] - y1;
        const float z2 = xyz2[((size_t)i * m + k) * 3 + 2] - z1;
        const float d = fmaf(z2, z2, fmaf(y2, y2, x2 * x2));
        if (k == 0 || d < best) {
          best = d;
          best_i = k;
        }
      }
      result[(size_t)i * n + j] = best;
      result_i[(size_t)i * n + j] = best_i;
    }
  }
}

/* Both directions, the layout of chamfer_cuda_forward (chamfer3D.cu:136-154). Returns 1 like the reference. */
int oracle_chamfer_forward(const float* xyz1, const float* xyz2, int b, int n, int m, float* dist1, float* dist2,
                           int32_t* idx1, int32_t* idx2) {
  oracle_nm_distance(b, n, xyz1, m, xyz2, dist1, idx1);
  oracle_nm_distance(b, m, xyz2, n, xyz1, dist2, idx2);
  return 1;
}

/* chamfer3D.cu:155-174: accumulate into (pre-zeroed) grad_xyz1 / grad_xyz2. */
void oracle_nm_distance_grad(int b, int n, const float* xyz1, int m, const float* xyz2, const float* grad_dist1,
                             const int32_t* idx1, float* grad_xyz1, float* grad_xyz2) {
  for (int i = 0; i < b; ++i) {
    for (int j = 0; j < n; ++j) {
      const size_t e = (size_t)i * n + j;
      const int j2 = idx1[e];
      const float g = grad_dist1[e] * 2;
      for (int c = 0; c < 3; ++c) {
        const float v = g * (xyz1[e * 3 + c] - xyz2[((size_t)i * m + j2) * 3 + c]);
        grad_xyz1[e * 3 + c] += v;
        grad_xyz2[((size_t)i * m + j2) * 3 + c] += -v;
      }
    }
  }
}

int oracle_chamfer_backward(const float* xyz1, const float* xyz2, int b, int n, int m, const float* graddist1,
                            const float* graddist2, const int32_t* idx1, const int32_t* idx2, float* gradxyz1,
                            float* gradxyz2) {
  oracle_nm_distance_grad(b, n, xyz1, m, xyz2, graddist1, idx1, gradxyz1, gradxyz2);
  oracle_nm_distance_grad(b, m, xyz2, n, xyz1, graddist2, idx2, gradxyz2, gradxyz1);
  return 1;
}

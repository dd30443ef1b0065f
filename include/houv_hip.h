/* houv_hip.h -- C ABI of libhouv_hip.so: the MI355X (gfx950) HOUV registration hot path.
 *
 * This is the drop-in boundary.  Every entry point replaces one interface of the reference
 * (Dizzy-cell/HOUV; paths relative to the reference root) and keeps its conventions:
 *   - the CALLER allocates every buffer; nothing is allocated or freed in here;
 *   - all pointers are DEVICE pointers to contiguous row-major arrays (fp32 / int32 / fp64 as
 *     declared) on the current device, unless a parameter says "host";
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the null stream) and
 *     the call returns without synchronising;
 *   - return value 1 = enqueued OK, 0 = error (message via houv_last_error()); this is the
 *     reference's own convention (utils/metrics/CD/chamfer3D/chamfer3D.cu:145-153).
 *
 * No torch types appear in any signature.
 */
#ifndef HOUV_HIP_H_
#define HOUV_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HOUV_ABI_VERSION 2

/* ABI version of the loaded library (== HOUV_ABI_VERSION). */
int houv_abi_version(void);

/* Thread-local text of the last error returned by any call below ("" if none). */
const char* houv_last_error(void);

/* sha256 (first 16 hex digits) of the sources this library was built from (houv_amd/csrc/Makefile): lets a committed
 * profile say which build it measured.  No counterpart in the reference. */
const char* houv_build_id(void);

/* Diagnostic switches for tests, A/B scripts and bench.py (process-wide; never read from the environment).  Names and
 * meanings: houv_amd/csrc/houv_common.h `DebugKnobs`.  Results do not depend on any of them except the ones that select
 * an alternative kernel of the same result for timing.  Returns 0 on an unknown name / out-of-range value.
 * No counterpart in the reference. */
int houv_debug_set(const char* name, long long value);

/* ---------------------------------------------------------------------------------------------
 * Chamfer nearest-neighbour op.
 * Replaces: pybind `chamfer_3D.forward` -> chamfer_cuda_forward
 *           (utils/metrics/CD/chamfer3D/chamfer_cuda.cpp:17-19,31; chamfer3D.cu:136-154; kernel :12-134).
 *   dist1[b,i] = min_j |xyz1[b,i]-xyz2[b,j]|^2   idx1[b,i] = argmin_j (lowest j on ties)
 *   dist2[b,j] = min_i |xyz2[b,j]-xyz1[b,i]|^2   idx2[b,j] = argmin_i
 * fp32 direct-difference arithmetic d = fma(dz,dz,fma(dy,dy,dx*dx)).
 * xyz1[B,N,3] xyz2[B,M,3] dist1[B,N] dist2[B,M] idx1[B,N] idx2[B,M].  N or M == 0 is an error. */
int houv_chamfer_forward(const float* xyz1, const float* xyz2, int B, int N, int M,
                         float* dist1, float* dist2, int32_t* idx1, int32_t* idx2, void* stream);

/* Replaces: pybind `chamfer_3D.backward` -> chamfer_cuda_backward
 *           (chamfer_cuda.cpp:22-26,32; chamfer3D.cu:176-195; kernel :155-174).
 * ACCUMULATES into gradxyz1[B,N,3] / gradxyz2[B,M,3], which the caller must have zero-filled
 * (dist_chamfer_3D.py:56-60), exactly like the reference:
 *   g = 2*graddist1[b,i]; gradxyz1[b,i] += g (x1_i - x2_idx1);  gradxyz2[b,idx1] -= same;  and symmetrically. */
int houv_chamfer_backward(const float* xyz1, const float* xyz2, int B, int N, int M,
                          const float* graddist1, const float* graddist2,
                          const int32_t* idx1, const int32_t* idx2,
                          float* gradxyz1, float* gradxyz2, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Kabsch rigid solve: 3x3 (weighted) covariance reduction + register-resident Jacobi SVD.
 * Replaces: SVDHead.forward (registration/model_utils.py:220-255).
 *   src, corr: [B,3,N] (channel-major like the reference); w: [B,1,N] or NULL.
 *   R[B,3,3], t[B,3].  Centres by the UNWEIGHTED means; reflection fix on det<0. */
int houv_kabsch(const float* src, const float* corr, const float* w_or_null, int B, int N,
                float* R, float* t, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused HOUV optimisation loop.
 * Replaces the body of predict_model (registration/models/houv.py:106-138) and of getPredict_angle
 * (registration/train_utils.py:359-456): for every hypothesis (pair p, restart k) it runs
 * `n_iters` x { pose from the 8 unconstrained scalars -> move the source cloud -> robust Chamfer
 * loss (Predict_loss, houv.py:209-222) -> closed-form gradient -> Adam step } entirely on chip.
 *
 *   src[P,N,3], tgt[P,M,3]   the P pairs (NOT replicated K times)
 *   state[P*K,24] fp64       per hypothesis: param[8] = (V0,V1,V2, a, c0,c1,c2, s), adam_m[8], adam_v[8];
 *                            in/out.  With f64_params == 0 the values are fp32 numbers stored widened.
 *   steps_done               Adam steps already applied to `state` (0 for a fresh stage)
 *   n_iters                  iterations to run now (>= 1)
 *   angle_base               0..3: rotation-angle window base*45deg .. base*45+45deg (houv.py:96)
 *   trans_mode               0: sigma = sin(s pi)/8 + 1/8 (houv.py:99)   1: sigma = sin(s pi) (train_utils.py:404)
 *   use_views                1: loss = 6 min_1 + three projected Chamfer terms (houv.py:222)
 *                            0: loss = 6 min_1 (train_utils.py:433)
 *   f64_params               0: fp32 parameters + fp32 Adam (HOUV module)   1: fp64 leaves + fp64 Adam (`solve` twin)
 *   k_full, k_view           top-k sizes: int(N*0.5) and int(N*1) (model_utils_completion.py:85-86)
 *   lr, beta1, beta2, eps    Adam hyper-parameters
 *   loss_scale               d(objective)/d(per-hypothesis loss) = 1/(P*K) for `.mean()` (houv.py:124)
 * Outputs (each may be NULL), all from the LAST forward pass, i.e. before the last Adam step
 * (houv.py:134-136):
 *   out_score[P*K] = min_1   out_loss[P*K]   out_R[P*K,9]   out_T[P*K,3]
 *   out_grad[P*K,8]  d(objective)/d(param) of the last forward
 *   out_cd[P*K,8]    the 8 Chamfer terms: metric m (0 = full, 1..3 = view dropping x,y,z) x
 *                    direction (0: over target points, 1: over moved points) at [2*m+dir]
 * Limits: (roundup32(N)+roundup32(M))*16 B + 8 KiB must fit in 160 KiB of LDS; K >= 1. */
int houv_solve_iterate(const float* src, const float* tgt, int P, int N, int M, int K,
                       double* state, int steps_done, int n_iters,
                       int angle_base, int trans_mode, int use_views, int f64_params,
                       int k_full, int k_view,
                       double lr, double beta1, double beta2, double eps, float loss_scale,
                       float* out_score, float* out_loss, float* out_R, float* out_T,
                       float* out_grad, float* out_cd, void* stream);

/* EXACT accelerated form of houv_solve_iterate (same arguments, same search result; what houv_amd.solver runs by default): every
 * query remembers its nearest neighbour of an earlier iteration; the distance to that point is an
 * attained upper bound, and 32-point sub-tiles whose bounding box lies farther than the bound for all metrics are
 * skipped.  Works best on spatially sorted clouds (houv_amd.solver reorders them so that runs of 32 points are k-d-tree leaves).  The search result
 * and the summation order are those of houv_solve_iterate: same outputs BIT FOR BIT when given the same clouds.
 *   nn_ws[P*K, 16, ws_stride] int16  workspace, in/out, opaque to the caller: per hypothesis rows [dir*4 + metric] hold the
 *              previous nearest-neighbour index of every point, rows 8..15 are scratch (ws_stride x 16 bytes);
 *              ws_stride >= max(N, M) and a multiple of 8
 *   ws_valid   0: nn_ws holds nothing yet (the first iteration of this call runs the brute-force sweep)
 *              1: nn_ws was left by the previous call on the same hypotheses (chunked launches)
 *             -1: verification mode: every iteration runs the brute-force sweep (nn_ws is not used)
 * Limits: as houv_solve_iterate (N, M <= 4096).  Up to 256 points the brute-force kernel runs (nothing to prune); up to 2048 points
 * a visit mask has one bit per 32-point sub-tile, above one bit per pair of sub-tiles. */
int houv_solve_iterate_pruned(const float* src, const float* tgt, int P, int N, int M, int K,
                              double* state, int steps_done, int n_iters,
                              int angle_base, int trans_mode, int use_views, int f64_params,
                              int k_full, int k_view,
                              double lr, double beta1, double beta2, double eps, float loss_scale,
                              float* out_score, float* out_loss, float* out_R, float* out_T,
                              float* out_grad, float* out_cd,
                              int16_t* nn_ws, int ws_valid, int ws_stride, void* stream);

/* Which kernel variant the two entry points above launch for clouds of N and M points (host-only query, no GPU work):
 * *block = threads per workgroup (256 / 512 / 1024), *points_per_lane = query points a lane owns (1..4), *prune_mode =
 * 0 brute-force sweep, 1 pruned search walked by the owning lanes, 2 / 3 pruned search with the balanced (sorted-block) walk over
 * 32-point sub-tiles / 64-point super-tiles --
 * the template arguments of houv::solve_kernel<block, points_per_lane, metrics, prune_mode, 1>.  Any out pointer may be NULL.
 * Returns 0 with houv_last_error() set when no variant serves the size (max(N,M) > 4096).
 * No counterpart in the reference (its kernel has one fixed launch shape, chamfer3D.cu:142-143); exported so that
 * the test-suite can prove that every variant is compared with the CPU oracle. */
int houv_solve_variant(int N, int M, int pruned, int* block, int* points_per_lane, int* prune_mode);

/* ---------------------------------------------------------------------------------------------
 * Point-to-point ICP refinement, one pair per workgroup (BASELINE configs[3]).
 * Replaces: the per-pair Open3D call of registration/train_ICP.py:137-153
 *   o3d.registration.registration_icp(pcd, pcd2, threshold=0.02, trans_init,
 *       TransformationEstimationPointToPoint(), ICPConvergenceCriteria(max_iteration=500))
 * (open3d==0.9.0, not under /root/reference: its published algorithm is restated; parity unpinned).
 *   src[P,N,3], tgt[P,M,3]; init [P,16] row-major 4x4 or NULL (identity)
 *   max_correspondence_distance: a source point corresponds to its NN in tgt iff dist < this
 *   relative_fitness / relative_rmse: Open3D's ICPConvergenceCriteria defaults are 1e-6 / 1e-6
 * Outputs: out_T[P,16] row-major 4x4 (bottom row 0,0,0,1; maps src into tgt's frame),
 *   out_fitness[P] = #correspondences/N, out_rmse[P] = inlier RMSE, out_iters[P] = updates applied (each may be NULL
 *   except out_T). */
int houv_icp_refine(const float* src, const float* tgt, int P, int N, int M, const float* init_or_null,
                    float max_correspondence_distance, int max_iteration, float relative_fitness,
                    float relative_rmse, float* out_T, float* out_fitness, float* out_rmse, int32_t* out_iters,
                    void* stream);

/* ---------------------------------------------------------------------------------------------
 * DCP feature head building blocks (BASELINE configs[4]): registration/models/dcp.py in fp32,
 * inference (eval-mode BatchNorm folded into per-channel scale/shift by the caller).  The model-level forward
 * (DGCNN -> Transformer -> soft correspondences -> houv_kabsch) is composed from these in houv_amd/models/dcp.py.
 * Layout convention: activations are row-major [rows, channels] ("token-major"), clouds [B,N,3]. */

/* dcp.py:35-42 `knn`: idx[B,N,k] = the k nearest points of xyz[B,N,3] to each point, nearest first, self included.
 * k in {1,3,8,16,20}. */
int houv_knn(const float* xyz, int B, int N, int k, int32_t* idx, void* stream);

/* dcp.py:44-66 + :285: edge feature cat(neighbour, centre)[6] -> relu(scale*(W[64,6] f) + shift) for every (point, neighbour):
 * out[(B*N*k), 64]. */
int houv_edgeconv1(const float* xyz, const int32_t* idx, int B, int N, int k, const float* W, const float* scale,
                   const float* shift, float* out, void* stream);

/* dcp.py:287/290/293/296 `x.max(dim=-1)`: out[p*ldo + c] = max_j act[(p*k+j)*C + c], p < npts (C, ldo multiples of 4). */
int houv_max_over_k(const float* act, long long npts, int k, int C, float* out, int ldo, void* stream);

/* fp32 GEMM on the matrix pipe with fused epilogue (1x1 conv / nn.Linear / attention products):
 *   C = relu?( (alpha * A[M,K] op(B)) * scale[n] + shift[n] + residual[m,n] ),  trans_b=1: B is [N,K] (C = A B^T), 0: [K,N].
 * Batched over outer*inner problems with element strides (s?o, s?i).  scale/shift/residual may be NULL.
 * fp32 in, fp32 out, fp32-grade products: full, 16-byte aligned [N,K] tiles run on the bf16 MFMAs with every operand split into
 * three bf16 parts and six part products per product (error at or below the fp32-input MFMA kernel's, which serves every other
 * shape); an Inf operand yields NaN there.  houv_debug_set("gemm_split", 0) selects the fp32-input kernel everywhere. */
int houv_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                  int trans_b, int outer, int inner, long long sAo, long long sAi, long long sBo, long long sBi,
                  long long sCo, long long sCi, float alpha, const float* scale_or_null, const float* shift_or_null,
                  const float* residual_or_null, int ldr, long long sRo, long long sRi, int relu, void* stream);

/* dcp.py:26-32 `attention` of MultiHeadedAttention (:198-229), fused: O = softmax(Q K^T * scale) V per (pair, head); the
 * [Nq,Nk] scores stay on chip (online softmax).  Token-major operands: head h of row n of pair p starts at
 * X + p*sX + n*ldX + h*dk.  dk must be 128 (DCP: 512 / 4 heads); strides multiples of 4 floats, pointers 16-byte aligned.
 * Full tiles (Nq % 128 == 0, Nk % 32 == 0) run on the bf16 MFMAs with three-part operand splits (fp32-grade, as houv_gemm_f32) and
 * take a stream-ordered workspace of 12 * P * H * Nk * 128 bytes (hipMallocAsync / hipFreeAsync on `stream`; without it, or with
 * houv_debug_set("attn_split", 0), the fp32-input kernel runs). */
int houv_attention_f32(const float* Q, const float* K, const float* V, float* O, int P, int H, int Nq, int Nk, int dk,
                       int ldq, int ldk, int ldv, int ldo, long long sQ, long long sK, long long sV, long long sO,
                       float scale, void* stream);

/* dcp.py:144-154 LayerNorm: out = a*(x-mean)/(std+eps)+b over the last dim D (torch.std: unbiased) [+ residual]. */
int houv_layernorm(const float* x, long long rows, int D, const float* a, const float* b, float eps,
                   const float* residual_or_null, float* out, void* stream);

/* dcp.py:31: x[rows,L] <- softmax over L, in place. */
int houv_softmax_rows(float* x, long long rows, int L, void* stream);

/* dcp.py:346-348: corr[P,3,N] = pts[P,M,3]^T . softmax(scores[P,N,M])^T, one pass per score row. */
int houv_softmax_corr(const float* scores, int P, int N, int M, const float* pts, float* corr, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Sibling point ops of utils/mm3d_pn2 (call site registration/train_utils.py:459-464 `combine`).
 * The reference's CUDA extensions cannot run here and it holds no fixtures for them: parity unpinned. */

/* furthest_point_sample (utils/mm3d_pn2/ops/furthest_point_sample/furthest_point_sample.py:15-36): idx[B,npoint],
 * starting from point 0, each next sample = the point furthest from the chosen set (lowest index on ties). */
int houv_furthest_point_sample(const float* xyz, int B, int N, int npoint, int32_t* idx, void* stream);

/* three_nn generalised (utils/mm3d_pn2/ops/interpolate/three_nn.py:11-37): the k (1, 3 or 8) nearest points of ref[B,M,3]
 * for every query[B,N,3], nearest first: dist2[B,N,k] SQUARED distances, idx[B,N,k]. */
int houv_knn_cross(const float* query, const float* ref, int B, int N, int M, int k, float* dist2, int32_t* idx,
                   void* stream);

/* gather_points (utils/mm3d_pn2/ops/gather_points/gather_points.py:14-35): out[B,C,M] = features[B,C,idx[B,M]]. */
int houv_gather_points(const float* features, const int32_t* idx, int B, int C, int N, int M, float* out, void* stream);

/* Pose only (HOUV.forward, houv.py:94-103): params fp32 [n,8] -> R[n,9], T[n,3]; if src != NULL
 * also moved[n,N,3] = src[n,N,3] @ R^T + T. */
int houv_pose_forward(const float* params, int n, int angle_base, int trans_mode,
                      const float* src_or_null, int N, float* R, float* T, float* moved_or_null, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HOUV_HIP_H_ */

/*
 * pcr_oracle.h — CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the reference's algorithm for the k-NN correspondence +
 * ICP + plane-inlier hot path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product path (libpcr_hip.so) never links, loads or calls it.
 *
 * Pinning status:
 *   - A1/A3/A6 (f32 squared distance, nanoflann result set, 1-NN argmin): PINNED against
 *     the vendored nanoflann 1.3.2 compiled from /root/reference (oracle/_ref) and the
 *     fixtures generated from it (tests/golden/gen_golden.py).
 *   - A2/A4/A11 (f64 sqrt distance, hw2 result sets, k-NN, radius-NN): PINNED against the
 *     hw2 headers compiled from /root/reference and the known-answer vector of
 *     Homework2/hw2/result_cpp.txt:11-33.
 *   - A10 (plane-inlier count): PINNED against the numpy expression of
 *     Homework4/ground_detection_ransac.py:138-139 evaluated by tests/golden/gen_golden.py.
 *   - A7/A9 (Kabsch via Eigen JacobiSVD, ICP loop): hw9 needs PCL+Eigen, which are absent ->
 *     the reference cannot be built here and holds no golden for this step:
 *     **parity unpinned** at the Eigen boundary; this file restates registration.cpp:862-1011
 *     line by line with f64 accumulation.
 *
 * All citations are relative to /root/reference/.
 */
#ifndef PCR_ORACLE_H
#define PCR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- A1: nanoflann L2_Adaptor::evalMetric, dim 3 (Homework9/hw9/include/nanoflann.hpp:403-406) */
float orc_d2_f32(float qx, float qy, float qz, float tx, float ty, float tz);

/* ---- A6 search part: brute-force 1-NN, A1 arithmetic, canonical tie rule
 * (min d2, then lowest target index).  Acceptance follows nanoflann:
 * a candidate is accepted only if d2 < FLT_MAX (nanoflann.hpp:163,1360); a query with no
 * acceptable candidate gets idx = UINT32_MAX, d2 = +inf.  SoA inputs. */
void orc_nn1_f32(const float* tx, const float* ty, const float* tz, size_t nt,
                 const float* sx, const float* sy, const float* sz, size_t ns,
                 uint32_t* idx, float* d2);
/* identical results, queries split over `threads` host threads */
void orc_nn1_f32_mt(const float* tx, const float* ty, const float* tz, size_t nt,
                    const float* sx, const float* sy, const float* sz, size_t ns,
                    uint32_t* idx, float* d2, int threads);
/* size of the tie set {j : d2_j == d2_min} per query (test helper for the tie-set rule). */
void orc_nn1_tiecount_f32(const float* tx, const float* ty, const float* tz, size_t nt,
                          const float* sx, const float* sy, const float* sz, size_t ns,
                          uint32_t* tie_count);

/* ---- A2: hw2 leaf distance (Homework2/hw2/include/kdtree.hpp:341-346):
 * d = sqrt(((0 + (t0-q0)^2) + (t1-q1)^2) + (t2-q2)^2) in f64. */
double orc_dist_f64(const double* t, const double* q, int dim);

/* ---- A2+A4: brute-force k-NN over an n x dim AoS f64 database for m queries, canonical order
 * (distance ascending, then index ascending).  Slots beyond n are (1e10, 0) like the
 * pre-filled hw2 result set (resultSet.hpp:35-42).  idx: m*k int32, dist: m*k f64. */
void orc_knn_f64(const double* db, size_t n, int dim, const double* q, size_t m, int k,
                 int32_t* idx, double* dist);

/* ---- A11: radius search, membership d <= r (inclusive, resultSet.hpp:133), CSR output in
 * ascending index order.  Call with idx==NULL to get row_ptr only (row_ptr has m+1 entries). */
void orc_radius_f64(const double* db, size_t n, int dim, const double* q, size_t m, double r,
                    int64_t* row_ptr, int32_t* idx, double* dist);
/* float variant used by Homework7/hw7/src/kdtree.cpp (ElemType float): d = sqrtf(sum) in f32 */
/* row counts only, on `threads` host threads (same comparison pair by pair) */
void orc_radius_count_f64_mt(const double* db, size_t n, int dim, const double* q, size_t m, double r, int64_t* counts, int threads);
void orc_radius_f32(const float* db, size_t n, int dim, const float* q, size_t m, float r,
                    int64_t* row_ptr, int32_t* idx, float* dist);

/* ---- A4: hw2 KNNResultSet::addPoint restated on caller arrays (resultSet.hpp:65-91).
 * dist/index arrays have `capacity` slots pre-filled (1e10, 0); *count and *worst are state. */
void orc_hw2_knn_add(double* dist, int* index, int capacity, int* count, double* worst,
                     double d, int i);
/* ---- A3: nanoflann KNNResultSet::addPoint (nanoflann.hpp:175-202) incl. the leaf gate
 * `dist < worstDist` (:1360). dists[capacity-1] must be initialised to FLT_MAX (:163). */
void orc_nano_knn_add(float* dists, size_t* indices, size_t capacity, size_t* count,
                      float d, size_t i);

/* ---- A8: transformCloudInplace (Homework9/hw9/src/registration.cpp:165-178), f32, unfused,
 * row-wise ((R_i0*x + R_i1*y) + R_i2*z) + t_i.  R row-major 3x3. SoA in place. */
void orc_transform_f32(float* x, float* y, float* z, size_t n, const float R[9], const float t[3]);

/* ---- A7 accumulation: f64 sums over kept pairs (d2 < max_corr, registration.cpp:936):
 * sums[0..2] = sum p (source), [3..5] = sum q (target), [6..14] = sum q_r * p_c (row-major 3x3,
 * rows = target, cols = source, registration.cpp:985), [15] = count.
 * Also returns the index of the last kept source (for `loss`, :939) or -1. */
int64_t orc_kabsch_accumulate(const float* sx, const float* sy, const float* sz, size_t ns,
                              const float* tx, const float* ty, const float* tz,
                              const uint32_t* idx, const float* d2, float max_corr,
                              double sums[16]);

/* 3x3 SVD (f64, one-sided Jacobi, singular values sorted descending like Eigen::JacobiSVD).
 * A row-major; A = U diag(S) V^T. */
void orc_svd3(const double A[9], double U[9], double S[3], double V[9]);

/* ---- A7 solve: from the 16 sums to (R_delta, t_delta) as f32, registration.cpp:979-998 incl. the
 * det<0 branch `R = V*B*U^T` (:990-996, sic).  Returns 0, or -1 when count == 0. */
int orc_kabsch_solve(const double sums[16], float R[9], float t[3]);
/* the RANSAC twin (registration.cpp:372-392): t from U V^T BEFORE the det < 0 repair, which then replaces R only */
int orc_kabsch_solve_ransac(const double sums[16], float R[9], float t[3]);

/* 4x4 f32 compose T_out = A * B, row-major, sequential k, unfused (registration.cpp:1002). */
void orc_mat4_mul_f32(const float A[16], const float B[16], float out[16]);

typedef struct {
    float max_corr;      /* m_ICP_max_corres_dist, compared against the SQUARED distance (:936) */
    uint64_t max_iter;   /* m_ICP_max_iter */
    float eps;           /* m_ICP_loss_epsilon */
} orc_icp_params;

typedef struct {
    uint64_t iters_run;       /* iterations whose update was applied */
    int converged;            /* 1 if the `unchanged_count > 15` break fired (:954) */
    int empty_pairs;          /* 1 if an iteration kept no pair (reference would divide by zero) */
    uint64_t last_pairs;      /* kept pairs in the last executed NN pass */
    float last_loss;
} orc_icp_stats;

/* ---- A9: ICPpoint2point (registration.cpp:862-1011, Appendix of SURVEY.md).  src is copied,
 * init_T (row-major 4x4, [R t; 0 1]) applied, then the loop.  out_T row-major 4x4.
 * If per_iter_T != NULL it receives max_iter * 16 floats: [R_delta t_delta; 0 1] of each
 * executed iteration; per_iter_pairs (optional) the kept-pair count of each NN pass. */
void orc_icp_p2p_f32(const float* sx, const float* sy, const float* sz, size_t ns,
                     const float* tx, const float* ty, const float* tz, size_t nt,
                     const float init_T[16], const orc_icp_params* prm,
                     float out_T[16], orc_icp_stats* stats,
                     float* per_iter_T, uint64_t* per_iter_pairs);

/* ---- A10: plane-inlier count (Homework4/ground_detection_ransac.py:138-139,152-153):
 * dist_i = |((x_i*a + y_i*b) + z_i*c) + 1.0*d| in f64 on f32-or-f64 points (np.c_ promotes to f64;
 * numpy's dot of an N x 4 f64 matrix with a 4-vector sums k = 0..3 sequentially for each row).
 * counts[h] = #{i : dist_i < thr}.  xyz is SoA. */
void orc_plane_count_f32pts(const float* x, const float* y, const float* z, size_t n,
                            const double* planes4, size_t n_planes, double thr, int64_t* counts);
void orc_plane_count_f64pts(const double* x, const double* y, const double* z, size_t n,
                            const double* planes4, size_t n_planes, double thr, int64_t* counts);
/* final mask (:152-153): mask[i] = dist_i < thr for ONE plane. */
void orc_plane_mask_f32pts(const float* x, const float* y, const float* z, size_t n,
                           const double plane4[4], double thr, uint8_t* mask);

/* ---- estimate_plane_params (ground_detection_ransac.py:158-169) in f64. p: 3 points x 3, row-major. */
void orc_plane_from_3pts(const double p[9], double params[4]);

/* ---- N3: voxel_filter, centroid mode (Homework1 voxel_filter.py:17-52) on an n x 3 f32 cloud (SoA), leaf as f64
 * (a strong np.float64, i.e. the numpy-1.x promotion the author ran: f32 subtraction, f64 division).
 * Quirk kept: a voxel is emitted when the NEXT voxel starts (:43-50), so the last voxel of the sorted order is never
 * emitted.  Centroid = sequential f32 sum in ascending point index / count (np.sum over axis 0 of f32 rows).
 * out: up to n points (SoA, f32 values as the reference stores them before the f64 cast); returns the count. */
size_t orc_voxel_filter_f32(const float* x, const float* y, const float* z, size_t n, double leaf_size,
                            float* ox, float* oy, float* oz);

/* ---- N1: ISS keypoints, ISSKeypoint::compute (Homework7/hw7/src/iss_detector.cpp:38-110) on an n x 3 f32 cloud (SoA).
 * Neighbourhoods use hw7's float kd-tree arithmetic (src/kdtree.cpp:310-316, ElemType float):
 *   s = 0; s = (float)((double)s + pow((double)(t_c - q_c), 2)) for c = 0..2; d = sqrtf(s); member iff d <= r.
 * Covariance (getEigenvalues, :113-152): weighted by 1 / |N(j)| and divided by the weight sum when `weighted`,
 * plain sum otherwise; the reference accumulates in f32 in tree-visit order (order-dependent, Eigen) — restated with
 * f64 accumulation in ascending index order and an f64 Jacobi eigen-solver, eigenvalues rounded to f32.
 * lambda3[i] = smallest eigenvalue if lambda2/lambda1 < gamma21 && lambda3/lambda2 < gamma32 && lambda3 > 0, else -1
 * (needs >= 3 neighbours, :72); keypoint iff lambda3[i] != -1, |N_nms(i)| >= min_neighbors and no neighbour within
 * non_max_radius has a larger lambda3 (:86-105).  is_key: n bytes; lambda3_out (optional): n floats.  Parity with the
 * reference is UNPINNED (hw7 needs PCL + Eigen). */
void orc_iss_f32(const float* x, const float* y, const float* z, size_t n, float local_radius, float non_max_radius,
                 float gamma21, float gamma32, int min_neighbors, int weighted, uint8_t* is_key, float* lambda3_out);

/* ---- N4: global-registration front half (Homework9/hw9/src/registration.cpp:288-434, :535-615).
 * N4a: nanoflann L2_Adaptor::evalMetric for any dim (nanoflann.hpp:382-405), f32, unfused: groups of four
 *   result += ((d0*d0 + d1*d1) + d2*d2) + d3*d3, then the 0-3 tail components one by one.  (The early exit on
 *   worst_dist only ever returns a partial sum for a candidate that is rejected anyway.) */
float orc_d2_dim_f32(const float* a, const float* b, int dim);
/* brute-force 1-NN over an n x dim row-major database, canonical tie rule (min d2, lowest index); acceptance
 * d2 < FLT_MAX as in A6.  PINNED against the vendored nanoflann at dim 33 (tests/golden/desc_match_hw9.npz). */
void orc_nn1_dim_f32(const float* db, size_t n, const float* q, size_t m, int dim, uint32_t* idx, float* d2);
/* N4b: findRANSACCorrespondencesUnion (:535-615): [nn_src(tgt_i), i] for every target, then [i, nn_tgt(src_i)] for every
 * source, sorted by distance (std::sort there: tie order unspecified; here stable = by position), the first
 * floor((1 - rate) * size) kept, the product evaluated in f32 as written (:605).  pairs: 2 x (n_src + n_tgt) u32
 * (src, tgt interleaved), dist likewise; returns the number kept. */
size_t orc_match_union_f32(const float* desc_src, size_t n_src, const float* desc_tgt, size_t n_tgt, int dim,
                           float rejection_rate, uint32_t* pairs, float* dist);
/* findRANSACCorrespondencesInter (:437-533): mutual nearest neighbours only — (s, nn_tgt(s)) is kept iff nn_src(nn_tgt(s)) == s —
 * in ascending s, then sorted by the source->target distance (std::sort there; stable here) and cut like the union. */
size_t orc_match_inter_f32(const float* desc_src, size_t n_src, const float* desc_tgt, size_t n_tgt, int dim,
                           float rejection_rate, uint32_t* pairs, float* dist);
/* N4c: one RANSAC hypothesis (:366-392): Kabsch over the 4 sampled correspondences `quad` (indices into pairs), f64
 * moments + orc_kabsch_solve (the reference: f32 Eigen, JacobiSVD — unpinned, as A7). src/tgt: AoS xyz. */
int orc_ransac_hypothesis(const float* src_xyz, const float* tgt_xyz, const uint32_t* pairs, const uint32_t quad[4],
                          float R[9], float t[3]);
/* consensus set size (:395-421): #{pairs : || tgt - (R src + t) || <= thr}, f32, unfused, row-wise
 * ((R_i0 x + R_i1 y) + R_i2 z) + t_i, norm = sqrtf((ex*ex + ey*ey) + ez*ez). */
uint32_t orc_consensus_count_f32(const float* src_xyz, const float* tgt_xyz, const uint32_t* pairs, size_t n_pairs,
                                 const float R[9], const float t[3], float thr);
/* the RANSAC loop over given quads (n_hyp x 4): first hypothesis with the largest consensus wins (strict >, :423);
 * counts (optional): n_hyp. Returns the winning hypothesis index or -1 when every consensus set is empty. */
int64_t orc_ransac_global_f32(const float* src_xyz, const float* tgt_xyz, const uint32_t* pairs, size_t n_pairs,
                              const uint32_t* quads, size_t n_hyp, float thr, float R[9], float t[3],
                              uint32_t* best_count, uint32_t* counts);

/* ---- N2: PCA ground fit around the inlier count (Homework4/ground_detection_SVD.py:46-92), f64 on f32-valued points
 * (the reference's points are f64 after pcd_preprocessing, :35).
 * FastEigen3x3 (Homework1/.../my_pybind11/src/mylib.cpp:9-189, the closed-form symmetric 3x3 solver the reference took
 * from open3d): unit eigenvector of the SMALLEST eigenvalue of symmetric A (row-major), (0,0,0) when max(A) == 0.
 * UNPINNED (pybind11 + Eigen are absent); cross-checked against numpy.linalg.eigh in the tests. */
void orc_fast_eigen3x3(const double A[9], double normal[3]);
/* extract_initial_seeds (:46-71): candidates z < -1.73 + 0.5; LPR = mean of the lpr_size lowest candidates (all of them
 * when there are fewer); seed iff candidate && z < LPR_z + threshold_seeds.  Only LPR_z is used by the reference.
 * seed_mask: n bytes.  Returns the number of seeds; *upper_bound = LPR_z + threshold_seeds (NaN without candidates).
 * PINNED through the reference function itself (tests/golden/ground_hw4.npz). */
size_t orc_ground_seeds_f64(const float* x, const float* y, const float* z, size_t n, size_t lpr_size,
                            double threshold_seeds, uint8_t* seed_mask, double* upper_bound);
/* estimate_plane (:74-85) over the points with mask != 0: centre = mean, XTX of the centred points (sequential f64 sums in
 * index order; numpy/BLAS order is unspecified), normal = FastEigen3x3(XTX), d = -normal . centre.  Returns the count. */
size_t orc_estimate_plane_f64(const float* x, const float* y, const float* z, size_t n, const uint8_t* mask, double params[4]);
/* ground_detection (:88-101): seeds, then max_iter x { estimate_plane(seeds); inliers = |[p 1] . params| < threshold_dist;
 * seeds = inliers }.  ground_mask: n bytes (the final inliers_filter).  Returns the number of inliers, or (size_t)-1 when
 * a fit had no point (the reference would propagate NaN). */
size_t orc_ground_detection_f64(const float* x, const float* y, const float* z, size_t n, int max_iter, size_t lpr_size,
                                double threshold_dist, double params[4], uint8_t* ground_mask);

/* ---- N1 (second consumer): squared-distance k-NN with the optional hybrid cap, and per-point PCA normals
 * (Homework1/.../pca_normal.py:89-103; search = open3d KDTreeFlann.search_hybrid_vector_3d -> FLANN, absent: UNPINNED).
 * s = ((dx*dx) + dy*dy) + dz*dz in f64 on f32 points (A2 without the sqrt = FLANN / nanoflann L2 at dim 3); canonical
 * order (s, then index); candidates need s < cap_s (strict; +inf = no cap); empty slots (DBL_MAX, -1). */
void orc_knn_sq_f32pts(const float* x, const float* y, const float* z, size_t n, const float* qx, const float* qy,
                       const float* qz, size_t m, int k, double cap_s, int32_t* idx, double* s_out, uint32_t* found);
/* normals[i] = FastEigen3x3 of the scatter matrix (centre = sum / cnt, pca_normal.py:20-22) of point i's <= k nearest
 * points with s < radius^2, in ascending (s, index) order; zeros when fewer than 3 (:97). */
void orc_normals_knn_f64(const float* x, const float* y, const float* z, size_t n, int k, double radius, double* normals);

/* ---- ICPpoint2plane (Homework9/hw9/src/registration.cpp:710-860), the point-to-plane sibling of A9 on the same 1-NN loop.
 * Per kept pair (d2 < max_corr, :778): row A = [n x p (as written :807-812), n], b = n.q - n.p in f32 as written (:814);
 * normal equations accumulated in f64 (the reference: f32 Eigen, (A^T A).inverse() * A^T * b, :818 — UNPINNED), 6x6 solve by
 * Gaussian elimination with partial pivoting, x rounded to f32; loss = |A x - b|^2 (:820) from the moments; the same
 * never-reset `unchanged` counter (:828-838); update R_delta = I + [x0..2]_x (NOT re-orthonormalised, :843), t_delta = x3..5,
 * T_total = T_delta T_total, source moved by (R_delta, t_delta).  tn*: target normals.  Other arguments as orc_icp_p2p_f32;
 * empty_pairs is also set when the 6x6 system is singular. */
void orc_icp_p2plane_f32(const float* sx, const float* sy, const float* sz, size_t ns,
                         const float* tx, const float* ty, const float* tz, size_t nt,
                         const float* tnx, const float* tny, const float* tnz,
                         const float init_T[16], const orc_icp_params* prm, float out_T[16], orc_icp_stats* stats);
/* the 6x6 solve used above: M symmetric (row-major 36), returns 0 or -1 when singular */
int orc_solve6(const double M[36], const double v[6], double x[6]);

#ifdef __cplusplus
}
#endif
#endif /* PCR_ORACLE_H */

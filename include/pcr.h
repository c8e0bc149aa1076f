/*
 * pcr.h — C ABI of libpcr_hip.so: the MI355X-native (gfx950) k-NN correspondence + ICP + plane-inlier
 * hot path of yf26/Hands-On-Point-Cloud-Processing.
 *
 * This is the drop-in boundary: plain pointers and sizes, caller-owned memory, `int`
 * status (0 = ok, negative = error; never exit()), one context per GPU / per process.  The source-level
 * replacement headers in include/pcr/ (kdtree.hpp, resultSet.hpp, KDTreeVectorOfVectorsAdaptor.h,
 * registration.hpp) forward to these entry points; INTEGRATION.md shows the reference-side bindings.
 * All file:line citations are relative to the reference repository root.
 *
 * Arithmetic contracts (bit-exact parity with the reference's CPU path):
 *   A1  f32 squared distance ((dx*dx + dy*dy) + dz*dz), d = q - t, every op rounded, no FMA
 *       — nanoflann::L2_Adaptor::evalMetric, Homework9/hw9/include/nanoflann.hpp:383-408
 *   A2  f64 d = sqrt(((dx*dx) + dy*dy) + dz*dz), d = t - q
 *       — KDTreeKNNSearch leaf loop, Homework2/hw2/include/kdtree.hpp:339-348
 *   ties: minimum distance, then lowest index (canonical rule)
 */
#ifndef PCR_H
#define PCR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCR_OK 0
#define PCR_ERR_ARG (-1)     /* bad argument */
#define PCR_ERR_HIP (-2)     /* a HIP runtime call failed; see pcr_ctx_last_error */
#define PCR_ERR_NOMEM (-3)
#define PCR_ERR_STATE (-4)   /* e.g. collective requested before pcr_comm_init */
#define PCR_ERR_COMM (-5)    /* RCCL failure */
#define PCR_ERR_EMPTY (-6)   /* ICP iteration kept no pair (the reference would divide by zero) */

typedef struct pcr_ctx pcr_ctx;
typedef struct pcr_cloud pcr_cloud;   /* an N-point f32 cloud resident in HBM as SoA x[N] | y[N] | z[N] */

/* host layouts accepted / produced at the boundary */
enum pcr_layout {
    PCR_SOA = 0,   /* x[n], y[n], z[n] contiguous — Eigen column-major N x 3 MatrixXf (registration.cpp:903) */
    PCR_AOS3 = 1,  /* xyzxyz...        — KITTI rows without intensity */
    PCR_AOS4 = 2,  /* xyz?xyz?...      — pcl::PointXYZ (16 B) and KITTI .bin rows (test.hpp:26-28) */
    PCR_AOS6 = 6   /* xyz???xyz???...  — hw9 registration .bin rows: xyz + normal (registration.cpp:25-26) */
};

/* ---- context ------------------------------------------------------------------------------------ */
int pcr_device_count(int* count);                     /* GPUs visible to this process (hipGetDeviceCount) */
int pcr_ctx_create(int device, pcr_ctx** out);
int pcr_ctx_destroy(pcr_ctx* ctx);
int pcr_ctx_sync(pcr_ctx* ctx);                       /* wait for the context's HIP stream */
const char* pcr_ctx_last_error(const pcr_ctx* ctx);   /* text of the last failure on this context */
const char* pcr_version(void);
/* device facts for the bench / roofline: name (e.g. "gfx950"), CU count, HBM bytes */
int pcr_ctx_device_info(const pcr_ctx* ctx, char* arch, size_t arch_cap, int* n_cu, uint64_t* hbm_bytes);

/* ---- clouds --------------------------------------------------------------------------------------- */
int pcr_cloud_create(pcr_ctx* ctx, const float* host_xyz, size_t n, int layout, pcr_cloud** out);
int pcr_cloud_clone(pcr_ctx* ctx, const pcr_cloud* src, pcr_cloud** out);
int pcr_cloud_assign(pcr_ctx* ctx, pcr_cloud* dst, const pcr_cloud* src);   /* dst <- src, same size, on device */
int pcr_cloud_read(pcr_ctx* ctx, const pcr_cloud* c, float* host_xyz, int layout);
size_t pcr_cloud_size(const pcr_cloud* c);
/* synchronises the context's stream, then frees the cloud's HBM (hipFree) — whatever context it is destroyed through */
int pcr_cloud_destroy(pcr_ctx* ctx, pcr_cloud* c);
/* The ICP loops clone the source into a working copy; that copy's buffer is parked on the context (at most two, each <= 2 GB, replaced
 * as soon as a call needs another size) so that the next call's clone costs no hipMalloc / hipFree.  pcr_ctx_trim frees what is parked. */
int pcr_ctx_trim(pcr_ctx* ctx);
int pcr_ctx_parked_bytes(const pcr_ctx* ctx, uint64_t* bytes);   /* HBM held by those parked buffers right now */

/* ---- A6 (search): 1-NN correspondence, brute force over LDS-tiled targets ----------------------------
 * replaces the loop `for i: tar_mat_index.index->findNeighbors(result_set, query, ...)`,
 * Homework9/hw9/src/registration.cpp:925-934, and KDTreeVectorOfVectorsAdaptor::query(q, 1, ...),
 * Homework3/nano_vs_my/include/KDTreeVectorOfVectorsAdaptor.h:80-85.
 * idx[i] = argmin_j d2(src_i, tgt_j) (A1, canonical ties), d2[i] the squared distance.
 * A query with no acceptable candidate (n_tgt == 0, NaN input, d2 >= FLT_MAX — nanoflann.hpp:163,1360)
 * gets idx = UINT32_MAX, d2 = +inf. */
int pcr_nn1_f32(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* src, uint32_t* idx, float* d2);
/* same, results stay in HBM (context workspace) for the Kabsch step; asynchronous on the ctx stream */
int pcr_nn1_f32_async(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* src);
/* ---- a caller's OWN loop over the same pair (what the searches inside pcr_icp_p2p_f32 do, one call at a time) --------------------
 * pcr_cloud_sort_for_target: re-orders `cloud` IN PLACE into the order of the target's index (built if needed), once, before the loop:
 *   the queries of a wave are then neighbours in space (coalesced loads, shared candidates; at >= 4 M target points the tile search
 *   needs it).  orig_index (host, n entries, may be NULL) receives, per position of the re-ordered cloud, the index the point had
 *   before; pcr_kabsch_sums reports `last_kept` in the ORIGINAL numbering for such a cloud (registration.cpp:939's "last pair").
 *   Moving the cloud with pcr_transform_f32 keeps the order valid; results of pcr_nn1_fetch are in the re-ordered numbering.
 * pcr_nn1_f32_loop: one search of such a loop — seeded by the previous call's correspondences, and bounded by the gate of
 *   registration.cpp:936: a pair is only ever kept if d2 < max_corr, so a query with no target inside the gate comes back as
 *   "none" (idx UINT32_MAX, d2 +inf) instead of with a neighbour the caller would discard.  Same kept pairs, same sums, bit for bit. */
int pcr_cloud_sort_for_target(pcr_ctx* ctx, const pcr_cloud* tgt, pcr_cloud* cloud, uint32_t* orig_index);
int pcr_nn1_f32_loop(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* src, float max_corr);
/* fetch the results of the last pcr_nn1_f32_async for n queries */
int pcr_nn1_fetch(pcr_ctx* ctx, size_t n, uint32_t* idx, float* d2);

/* ---- A8: transformCloudInplace, Homework9/hw9/src/registration.cpp:165-178 ---------------------------
 * p <- R p + t per point, f32, unfused, row-wise ((R_i0 x + R_i1 y) + R_i2 z) + t_i.  T row-major 4x4. */
int pcr_transform_f32(pcr_ctx* ctx, pcr_cloud* cloud, const float T[16]);

/* ---- A7 (accumulate): cross-covariance sums over the kept pairs of the last pcr_nn1_f32_async ---------
 * sums[0..2] = sum p, [3..5] = sum q, [6..14] = sum q_r p_c (row-major, rows = target), [15] = count,
 * f64, over pairs with d2 < max_corr (squared distance vs un-squared parameter, registration.cpp:936).
 * last_kept: index of the last kept source or -1; last_d2: its squared distance (loss, :939). */
int pcr_kabsch_sums(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* src, float max_corr,
                    double sums[16], int64_t* last_kept, float* last_d2);
/* ---- A7 (solve): registration.cpp:979-998 incl. the det<0 branch (:990-996). Host, f64. -------------- */
int pcr_kabsch_solve(const double sums[16], float R[9], float t[3]);

/* ---- A9: Registration::ICPpoint2point, Homework9/hw9/src/registration.cpp:862-1011 -------------------
 * (after its normal-space sampling: the clouds passed here are the sampled clouds). */
typedef struct {
    float max_corr;       /* setICPparams max_corres_dist (registration.hpp:126-137); vs SQUARED distance */
    uint64_t max_iter;    /* setICPparams max_iter */
    float eps;            /* setICPparams loss_epsilon */
} pcr_icp_params;

typedef struct {
    uint64_t iters_run;   /* iterations whose update was applied */
    int32_t converged;    /* the `unchanged_count > 15` break fired (:954) */
    int32_t empty_pairs;  /* an iteration kept no pair */
    uint64_t last_pairs;  /* kept pairs (whole job, after the all-reduce) in the last NN pass */
    float last_loss;
    float reserved;
    double ms_total;      /* wall time of the loop */
    double ms_nn;         /* HIP-event time of the 1-NN kernel, summed over iterations; 0 unless pcr_tune_set(ctx, "prof", >= 1) */
    uint64_t nn_launches;
} pcr_icp_stats;

/* src is not modified (a working copy is transformed in place, :872-874). init_T / out_T row-major 4x4.
 * With a communicator attached (pcr_comm_*), src is this rank's shard: the 16 f64 sums are all-reduced
 * once per iteration and every rank computes the identical pose. */
int pcr_icp_p2p_f32(pcr_ctx* ctx, const pcr_cloud* src, const pcr_cloud* tgt, const float init_T[16],
                    const pcr_icp_params* prm, float out_T[16], pcr_icp_stats* stats);

/* ---- A10: plane-inlier count, Homework4/ground_detection_ransac.py:138-139,152-153 -------------------
 * dist_i = |((x a + y b) + z c) + d| in f64; counts[h] = #{i : dist_i < thr} for n_planes hypotheses in
 * ONE pass over the points. planes4: n_planes x 4 f64. */
int pcr_plane_count_f64(pcr_ctx* ctx, const pcr_cloud* pts, const double* planes4, size_t n_planes,
                        double thr, int64_t* counts);
/* final mask of one plane (:152-153): mask[i] = dist_i < thr (uint8), n_inliers optional */
int pcr_plane_mask_f64(pcr_ctx* ctx, const pcr_cloud* pts, const double plane4[4], double thr,
                       uint8_t* mask, int64_t* n_inliers);

/* ---- A2/A4: batched k-NN with the hw2 arithmetic (f64, sqrt), canonical order ------------------------
 * replaces `KNNResultSet rs(k); KDTreeKNNSearch(root, db, rs, query)` per query
 * (Homework2/hw2/include/kdtree.hpp:329-364, benchmark.hpp:59-66).
 * db: n x 3 f64 AoS (vector<vector<double>> flattened), q: m x 3. idx/dist: m x k; empty slots (n < k)
 * hold (1e10, 0) like the pre-filled result set (resultSet.hpp:35-42). k <= 32. */
int pcr_knn_f64(pcr_ctx* ctx, const double* db, size_t n, const double* q, size_t m, int k,
                int32_t* idx, double* dist);
/* ---- A11: radius search (kdtree.hpp:367-402, resultSet.hpp:130-140): all j with d <= r, CSR, ascending
 * index. Pass idx = dist = NULL to get row_ptr (m+1) only, then call again with arrays of row_ptr[m]. */
int pcr_radius_f64(pcr_ctx* ctx, const double* db, size_t n, const double* q, size_t m, double r,
                   int64_t* row_ptr, int32_t* idx, double* dist);

/* ---- the same searches on a database kept resident in HBM: the counterpart of the tree object the
 * reference builds once (KDTreeConstruction, kdtree.hpp:419; nanoflann buildIndex, nanoflann.hpp:1191) and
 * queries many times.  squared = 0: hw2 contract (d = sqrt(s)); squared = 1: nanoflann contract for
 * T = double (squared L2, nanoflann.hpp:403-406; empty slots hold (DBL_MAX, -1)). */
typedef struct pcr_db64 pcr_db64;
int pcr_db64_create(pcr_ctx* ctx, const double* db, size_t n, pcr_db64** out);
int pcr_db64_destroy(pcr_ctx* ctx, pcr_db64* db);
size_t pcr_db64_size(const pcr_db64* db);
int pcr_db64_knn(pcr_ctx* ctx, const pcr_db64* db, const double* q, size_t m, int k, int squared,
                 int32_t* idx, double* dist);
int pcr_db64_radius(pcr_ctx* ctx, const pcr_db64* db, const double* q, size_t m, double r,
                    int64_t* row_ptr, int32_t* idx, double* dist);

/* ---- device-resident radius rows: the CSR of a radius search kept in HBM (12 B per neighbour never cross PCIe unless asked for) --------
 * What the batched consumers of the reference do with radius rows is reduce them — neighbour counts and 1 / count weights
 * (Homework7/hw7/src/iss_detector.cpp:48-76), neighbourhood moments for normals (Homework1 pca_normal.py:89-103) — or walk them in query
 * order (the self-query loop of Homework2/hw2/include/benchmark.hpp:66-70).  q == NULL: every point of db queries db (m is ignored).
 * Same rows as pcr_db64_radius, bit for bit (ascending index inside a row, d <= r inclusive).  The database must outlive the handle. */
typedef struct pcr_rows pcr_rows;
int pcr_db64_radius_rows(pcr_ctx* ctx, const pcr_db64* db, const double* q, size_t m, double r, pcr_rows** out);
int pcr_rows_destroy(pcr_ctx* ctx, pcr_rows* rows);
int pcr_rows_info(const pcr_rows* rows, size_t* m, uint64_t* total);            /* queries, reported neighbours */
int pcr_rows_row_ptr(const pcr_rows* rows, int64_t* row_ptr);                   /* the m + 1 offsets (host copy, no GPU work) */
/* rows [row_begin, row_end) -> idx / dist (either may be NULL), (row_ptr[row_end] - row_ptr[row_begin]) entries: iterate in bounded blocks */
int pcr_rows_fetch(pcr_ctx* ctx, const pcr_rows* rows, size_t row_begin, size_t row_end, int32_t* idx, double* dist);
enum pcr_rows_op { PCR_ROWS_COUNT = 0, PCR_ROWS_SUM_DIST = 1, PCR_ROWS_MAX_DIST = 2 };
int pcr_rows_reduce(pcr_ctx* ctx, const pcr_rows* rows, int op, double* out);    /* out[m]; an empty row gives 0 */
/* per row: mean (m x 3) and scatter matrix sum (p - mean)(p - mean)^T / count as xx xy xz yy yz zz (m x 6) of its neighbours' coordinates */
int pcr_rows_moments(pcr_ctx* ctx, const pcr_rows* rows, double* mean, double* cov);

/* Registration::ICPpoint2plane (registration.hpp:195-202, registration.cpp:710-860): the point-to-plane variant on the same
 * 1-NN loop.  tgt_normals: one normal per target point (a cloud whose x/y/z are normal_x/y/z; e.g. floats 3..5 of hw9's .bin
 * rows).  Rows A = [n x p, n], b = n.q - n.p in f32 as written (:807-814); normal equations in f64; update R_delta = I + [x]_x,
 * NOT re-orthonormalised (:843).  Same parameters, stop rules and stats as pcr_icp_p2p_f32; `empty_pairs` is also set when
 * the 6x6 system is singular.  Sources shard like the point-to-point loop: one all-reduce of 29 f64 per iteration. */
int pcr_icp_p2plane_f32(pcr_ctx* ctx, const pcr_cloud* src, const pcr_cloud* tgt, const pcr_cloud* tgt_normals, const float init_T[16],
                        const pcr_icp_params* prm, float out_T[16], pcr_icp_stats* stats);

/* ---- next row N3: voxel-grid down-sampling, Homework1 voxel_filter.py:17-52 (centroid mode) -----------------
 * One centroid per occupied voxel of edge leaf_size, in ascending voxel-index order; f32 arithmetic and summation
 * order of the reference (bit-exact), including its quirk that the last voxel of the sorted order is never emitted
 * (:41-50).  The result is a new device cloud (feed it to pcr_icp_p2p_f32 without leaving HBM). */
int pcr_voxel_filter_f32(pcr_ctx* ctx, const pcr_cloud* in, double leaf_size, pcr_cloud** out);

/* ---- next row N1: ISS keypoints, ISSKeypoint::compute (Homework7/hw7/src/iss_detector.cpp:38-152) ----------------
 * Batched radius neighbourhoods over a uniform grid instead of one kd-tree search per point.  Membership uses hw7's
 * float arithmetic (src/kdtree.cpp:310-316) bit for bit; the neighbourhood covariance is accumulated in f64 (the
 * reference: f32 in tree-visit order through Eigen) and its eigenvalues come from an f64 Jacobi solver, rounded to
 * f32 before the gamma tests (:79).  Setters mirrored: setLocalRadius / setNonMaxRadius / setThreshold(g21, g32) /
 * setMinNeighbors / useWeightedCovMat (iss_detector.cpp:8-31).
 * is_key[i] = 1 iff input point i is a keypoint (the reference emits them in ascending i, :103); lambda3 (optional, n
 * floats) is lambda3_vec_forall (:67); neighbor_counts (optional, n) is rnn_idx[i].size() (:47-57); n_keypoints
 * (optional) the number of ones. */
typedef struct {
    float local_radius;
    float non_max_radius;
    float gamma21, gamma32;
    int min_neighbors;
    int weighted_covariance;
} pcr_iss_params;
int pcr_iss_keypoints_f32(pcr_ctx* ctx, const pcr_cloud* cloud, const pcr_iss_params* prm, uint8_t* is_key, float* lambda3,
                          uint32_t* neighbor_counts, uint64_t* n_keypoints);

/* ---- next row N1 (second consumer): batched exact k-NN on resident clouds + per-point PCA normals ------------------
 * k-NN of every point of `queries` (may be `db` itself) in `db`, k <= 32, over a uniform grid; f64 leaf arithmetic of
 * hw2 / FLANN / nanoflann at dim 3 on the f32 coordinates widened to f64: s = ((dx*dx) + dy*dy) + dz*dz.
 * squared != 0: order and report s (FLANN / open3d / nanoflann contract), empty slots (DBL_MAX, -1); radius >= 0 adds the
 * hybrid-search cap s < radius^2 (strict, FLANN's KNNRadiusResultSet); radius < 0: none.
 * squared == 0: order and report d = sqrt(s) (hw2, kdtree.hpp:341-346), empty slots (1e10, 0) (resultSet.hpp:35-42).
 * Canonical order: value ascending, then index ascending.  idx, dist: m x k row-major; found (optional): m. */
int pcr_cloud_knn_f64(pcr_ctx* ctx, const pcr_cloud* db, const pcr_cloud* queries, int k, double radius, int squared,
                      int32_t* idx, double* dist, uint32_t* found);
/* pca_normal.py:89-103: normals[i] = eigenvector of the smallest eigenvalue of the scatter matrix of the <= k nearest
 * points within `radius` of point i (itself included; the reference: search_hybrid_vector_3d(radius = 5, max_nn = 10) +
 * PCA, :17-36), zeros when fewer than 3.  Sign and eigen-solver follow FastEigen3x3 (the reference's PCA_faster, :39-45;
 * np.linalg.eig leaves the sign unspecified).  normals: n x 3 f64. */
int pcr_normals_knn_f64(pcr_ctx* ctx, const pcr_cloud* cloud, int k, double radius, double* normals);

/* pca_normal.py:17-36 PCA(data, correlation = False, sort = True) of the whole cloud: eigenvalues descending, eigenvectors in
 * the columns of the row-major 3x3 (signs unspecified, as with np.linalg.eig); centre (optional) = sum / n.  Non-finite
 * points are skipped; PCR_ERR_EMPTY without a finite point. */
int pcr_cloud_pca_f64(pcr_ctx* ctx, const pcr_cloud* cloud, double eigenvalues[3], double eigenvectors[9], double centre[3]);

/* ---- next row N2: PCA ground fit around the inlier count, Homework4/ground_detection_SVD.py:46-101 --------------
 * f64 arithmetic on the f32 points of the cloud (the reference's points are f64 after pcd_preprocessing, :35).
 * pcr_fast_eigen3x3 (host logic, no GPU) = mylib.FastEigen3x3 (Homework1/.../my_pybind11/src/mylib.cpp:105-189): unit
 * eigenvector of the smallest eigenvalue of the symmetric row-major A; (0,0,0) when the signed maximum of A is 0. */
int pcr_fast_eigen3x3(const double A[9], double normal[3]);
/* extract_initial_seeds (:46-71): seed_mask[i] = z_i < -1.73 + 0.5 && z_i < LPR_z + threshold_seeds, LPR_z = mean z of
 * the lpr_size lowest candidates (all of them when fewer).  upper_bound / n_seeds optional. */
int pcr_ground_seeds_f64(pcr_ctx* ctx, const pcr_cloud* cloud, size_t lpr_size, double threshold_seeds, uint8_t* seed_mask,
                         double* upper_bound, uint64_t* n_seeds);
/* ground_detection (:88-101): seeds, then max_iter (>= 1) x { estimate_plane (:74-85); inliers = |[p 1].params| <
 * threshold_dist }.  params = the last plane (normal, d); ground_mask = the last inliers_filter; PCR_ERR_EMPTY when a
 * fit has no point (the reference would propagate NaN). */
int pcr_ground_detection_f64(pcr_ctx* ctx, const pcr_cloud* cloud, int max_iter, size_t lpr_size, double threshold_dist,
                             double params[4], uint8_t* ground_mask, uint64_t* n_ground);

/* ---- next row N4: global-registration front half, Homework9/hw9/src/registration.cpp:288-434, :535-615 -----------
 * N4a: exhaustive 1-NN between two descriptor sets (row-major n x dim / m x dim f32, host memory; dim 33 = FPFH),
 * nanoflann's evalMetric arithmetic for any dim (nanoflann.hpp:382-405: groups of four + tail, f32, unfused), canonical
 * tie rule (min d2, lowest index), acceptance d2 < FLT_MAX; idx = UINT32_MAX / d2 = +inf when nothing is accepted. */
int pcr_nn1_desc_f32(pcr_ctx* ctx, const float* db, size_t n, const float* q, size_t m, int dim, uint32_t* idx, float* d2);
/* N4b: findRANSACCorrespondencesUnion (:535-615).  pairs: room for 2 * (n_src + n_tgt) u32, (src, tgt) interleaved;
 * dist: n_src + n_tgt floats (squared descriptor distance of each kept pair); *n_pairs = floor((1 - rate) * total) as
 * the reference computes it (f32).  Sorted by distance; ties in input order (the reference: std::sort, unspecified). */
int pcr_match_union_f32(pcr_ctx* ctx, const float* desc_src, size_t n_src, const float* desc_tgt, size_t n_tgt, int dim,
                        float rejection_rate, uint32_t* pairs, float* dist, size_t* n_pairs);
/* findRANSACCorrespondencesInter (:437-533): the mutual-nearest-neighbour variant — (s, nn_tgt(s)) kept iff nn_src(nn_tgt(s)) == s,
 * ascending s, then sorted by the source->target distance and cut to floor((1 - rate) * count).  pairs: room for 2 * n_src. */
int pcr_match_inter_f32(pcr_ctx* ctx, const float* desc_src, size_t n_src, const float* desc_tgt, size_t n_tgt, int dim,
                        float rejection_rate, uint32_t* pairs, float* dist, size_t* n_pairs);
/* N4c: Registration::RANSAC (:288-434).  Sampling (host logic, no GPU): n_hyp quads of correspondence indices drawn as
 * :318-352 draws them (std::mt19937 + uniform_int_distribution, non-coplanar SOURCE keypoints), seeded explicitly
 * instead of std::random_device.  PCR_ERR_STATE when no admissible quad exists (the reference would loop forever). */
int pcr_ransac_sample_quads(const float* src_xyz, size_t n_src, const uint32_t* pairs, size_t n_pairs, size_t n_hyp,
                            uint64_t seed, uint32_t* quads);
/* consensus-set size (:395-421) of n_hyp given poses Rt[h] = {R row-major 9, t 3} over all correspondences:
 * counts[h] = #{i : || tgt_i - (R src_i + t) || <= thr}, f32, unfused.  xyz arrays are AoS (n x 3), host memory. */
int pcr_consensus_count_f32(pcr_ctx* ctx, const float* src_xyz, size_t n_src, const float* tgt_xyz, size_t n_tgt,
                            const uint32_t* pairs, size_t n_pairs, const float* Rt, size_t n_hyp, float thr, uint32_t* counts);
/* the whole loop over given quads: 4-point Kabsch per hypothesis (:354-392, f64 moments + the solve of
 * pcr_kabsch_solve) -> consensus counts -> first hypothesis with the largest count (:423-428).  *winner = -1 (R, t
 * untouched) when every consensus set is empty; counts (optional): n_hyp. */
int pcr_ransac_global_f32(pcr_ctx* ctx, const float* src_xyz, size_t n_src, const float* tgt_xyz, size_t n_tgt,
                          const uint32_t* pairs, size_t n_pairs, const uint32_t* quads, size_t n_hyp, float thr,
                          float R[9], float t[3], uint32_t* best_count, int64_t* winner, uint32_t* counts);

/* ---- multi-GPU: one process per GPU, sources sharded, targets replicated ------------------------------
 * Exactly one collective per ICP iteration: all-reduce(sum) of 56 + 2 * nranks f64 (wire format below; 104 at PCR_MAX_RANKS —
 * a pcr_allreduce_fn must accept n up to 128). */
#define PCR_COMM_ID_BYTES 128
#define PCR_MAX_RANKS 24      /* one node has 8 GPUs; the per-iteration reduce buffer holds 56 + 2 * nranks <= 128 f64 */
int pcr_comm_unique_id(char id[PCR_COMM_ID_BYTES]);                     /* rank 0; broadcast it out of band */
int pcr_comm_init_rccl(pcr_ctx* ctx, int nranks, int rank, const char id[PCR_COMM_ID_BYTES]);
/* alternative transport: a host callback that sums buf[0..n) over ranks in place and returns 0
 * (lets the caller use its own process group, e.g. torch.distributed gloo on CPU boxes) */
typedef int (*pcr_allreduce_fn)(void* user, double* buf, int n);
int pcr_comm_init_callback(pcr_ctx* ctx, int nranks, int rank, pcr_allreduce_fn fn, void* user);
int pcr_comm_destroy(pcr_ctx* ctx);
/* one real ncclAllReduce on the attached RCCL communicator (any nranks, also 1), result checked */
int pcr_comm_selftest(pcr_ctx* ctx);
/* Wire format of the per-iteration collective (what a pcr_allreduce_fn sums): 56 + 2 * nranks doubles —
 *   [0..54]  the 16 Kabsch moments as exact 40-bit integer limbs on a fixed-point grid shared by all ranks (unit 2^(e-80) for the
 *            six coordinate sums: 3 doubles each = limb 0, limb 1, carry; unit 2^(2e-120) for the nine products q_r p_c: 4 doubles
 *            each; [54] the pair count), [55] overflow flag, [56 + 2r] / [57 + 2r] (ORDER KEY, d2 of the last kept pair) of rank r: the key
 *            is 0 when the rank kept no pair, else the pair's index in the whole source cloud + 1 for a shard made by
 *            pcr_cloud_shard_spatial, rank + 1 for any other cloud (contiguous blocks in rank order).  The iteration's `loss`
 *            (registration.cpp:939: d2 of the last pair pushed) is taken from the slot with the LARGEST key.
 * Every entry is an integer (or one rank's value) below 2^53, so the sum is exact in any order: the pose does not depend on the
 * number of ranks.  pcr_kabsch_limbs_to_sums (host logic, no GPU) propagates the carries of a summed row in place and returns
 * the 16 moments pcr_kabsch_solve takes; e = pcr_kabsch_grid_exponent(largest finite |target coordinate|, max_corr). */
int pcr_kabsch_grid_exponent(float target_absmax, float max_corr);
int pcr_kabsch_limbs_to_sums(double row[55], int e, double sums[16]);
/* contiguous shard [begin, end) of n items for `rank` of `nranks` (sizes differ by at most one) */
void pcr_shard_range(size_t n, int nranks, int rank, size_t* begin, size_t* end);
/* SPATIALLY COHERENT shards (the reference shards nothing — registration.cpp:925-941 is one serial loop — so the partition is ours to
 * choose, and the exact integer sums make the pose independent of it, bit for bit).  `full` = the WHOLE source cloud, uploaded on
 * every rank (10 M points: 120 MB); out = this rank's share: the cloud in the order of the target's index (cell, then Morton code
 * inside the cell) is cut into nranks x chunks_per_rank runs of equal length (0 = 64 per rank) that are dealt round-robin.  Every rank
 * holds compact pieces of the scene at the scene's own density — the tile search of large targets needs that; a uniformly drawn 1 / N
 * sample (contiguous blocks of a shuffled cloud) spreads its queries N times thinner — and the deal balances expensive regions
 * against cheap ones.  Deterministic: all ranks compute the same order from the same two clouds (shards disjoint, complete).  The
 * shard keeps ascending original order and remembers each point's index in `full` (pcr_cloud_global_index), which is what makes
 * "the last kept pair" well defined across ranks (the wire format above).  `full` may be destroyed afterwards. */
int pcr_cloud_shard_spatial(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* full, int nranks, int rank, int chunks_per_rank, pcr_cloud** out);
int pcr_cloud_global_index(pcr_ctx* ctx, const pcr_cloud* shard, uint32_t* index);   /* host, pcr_cloud_size(shard) entries, ascending */

/* ---- profiling hooks for bench.py: HIP-event timing of the dominant kernel on the ctx stream ----------
 * Off by default (an event pair costs ~6 us of stream time on each side of the kernel it brackets):
 * pcr_tune_set(ctx, "prof", 1) times the correspondence kernels, 2 every kernel, 0 switches it off again. */
int pcr_prof_reset(pcr_ctx* ctx);
int pcr_prof_get(pcr_ctx* ctx, const char* kernel, uint64_t* launches, double* total_ms);
/* the individual durations (ms, launch order, at most 4096 since the last reset): *n = how many exist, ms[0 .. min(*n, cap)) filled */
int pcr_prof_get_each(pcr_ctx* ctx, const char* kernel, double* ms, size_t cap, size_t* n);
/* diagnostics of the last grid search launched with tune "grid_stats" = 1:
 * out = { candidates evaluated, fine x-rows opened, coarse rows tested, far stages run } summed over the queries */
int pcr_grid_stats(pcr_ctx* ctx, uint64_t out[4]);
/* the sixteen diagnostics words of the last 1-NN launch made with tune "grid_stats" = 1.  Cell walk: [0..3] as pcr_grid_stats.
 * Matrix-core exhaustive search (HTRACK / BTRACK): [2] = (wave, query group) pairs that were filtered a second time, [4] = shader
 * cycles (s_memtime) and [5] = 100 MHz real-time ticks (s_memrealtime) summed over the workgroups: [4] / [5] x 100 MHz is the shader
 * clock the chip held under that launch (bench.py: the clock-corrected roofline).  STRACK (the sign form of the f16 filter): [2] =
 * joint evaluations of a wave's list of flagged chunks, [6] = (query, 16-record chunk) pairs evaluated exactly, [4] / [5] as above.
 * STRACK3 (csrc/nn1_sphere.hpp) in addition: [7] level-0 MFMAs, [3] level-1 tiles flagged by level 0, [8] level-1 MFMAs, [9] level-2 tiles flagged
 * by level 1, [10] level-2 MFMAs.  Sign tile search (csrc/grid_stile.hpp) + the walk of its deferred
 * queries: [0] pairs through the sign filter, [1] coarse cells read, [2] tile spheres tested, [3] operand setups, [4] / [5] box / ball (um,
 * summed), [6] queries deferred, [7] passes, [8] exact evaluations, [9] joint evaluations, [10] records under listed tiles, [11] MFMAs,
 * [12] most tiles in a pass, [13] / [14] / [15] deferred because: beyond the ball limit / more clusters than passes / list overflow. */
int pcr_nn1_stats(pcr_ctx* ctx, uint64_t out[16]);
/* Checks on THIS device the arithmetic the matrix-core forms of the exhaustive 1-NN filter rely on (csrc/nn1_brute.hip: BTRACK = two
 * v_mfma_f32_32x32x16_bf16 per tile, three-piece operands; HTRACK = one v_mfma_f32_32x32x16_f16, two-piece scaled operands — the default).
 * `trials` random tiles per mode plus 4 structured (cancellation across K-slots, alternating signs, subnormal pieces, maximal exponent
 * spread) and 8 underflow-regime tiles, through the kernel's own MFMA(s) and operand code; u = 2^-24:
 *   worst[0] = max |D - exact| / (u sum |a_k b_k|), random operands                                   (the bounds assume <= 16)
 *   worst[1] = max (|G - (w - 2 r.t)| - 2 u) / (u (|r|^2 + |t|^2)), the kernel's operand layout       (assume <= 34.2 bf16 / <= 82 f16)
 *   worst[2] = max |G - (w - 2 r.t)| / u over pairs with |r|^2 + |t|^2 <= 2^-6 (f16 underflow regime)  (HTRACK subtracts 4 u)
 *   worst[3] = as worst[0] on the structured tiles                                                    (<= 16) */
int pcr_selftest_mfma_bf16_v2(pcr_ctx* ctx, int trials, double worst[4]);
int pcr_selftest_mfma_f16_v2(pcr_ctx* ctx, int trials, double worst[4]);
/* the round-2 forms of the same self-tests: worst[0] and worst[1] only (a caller built against the older header passes two doubles) */
int pcr_selftest_mfma_bf16(pcr_ctx* ctx, int trials, double worst[2]);
int pcr_selftest_mfma_f16(pcr_ctx* ctx, int trials, double worst[2]);
/* The DECISION of the sign form of the f16 filter (STRACK, csrc/nn1_brute.hip — the default exhaustive search on targets that fit f16:
 * the query's threshold rides in two K-slots, an accumulator is bound - threshold, its sign bit says whether the record can matter),
 * checked on THIS device: `trials` random super-tile tiles + 8 in the f16 underflow regimes, a power-of-two scale per tile, thresholds
 * ON the exact distance of one pair, one ulp below / above it and a factor away, through the kernel's own operand code and MFMA.
 * out = { pairs whose exact f32 distance lies at or below their query's threshold, of those WITHOUT the sign — must be 0 —,
 * pairs with the sign set, pairs in all }.  A short form is part of the once-per-context check below. */
int pcr_selftest_sign_f16(pcr_ctx* ctx, int trials, uint64_t out[4]);
/* The SPHERE statement of the hierarchical form of that filter (STRACK3, csrc/nn1_sphere.hpp: one MFMA row per set of records — its bounding
 * sphere — before any per-record row): `trials` random level-1 tiles of 32 chunks (tight and wide clusters, chunks spread beyond the scaled
 * range, empty chunks, non-finite records) against 32 queries each (near, inside, far, beyond the clamp) with thresholds on / one ulp off / a
 * factor off exact distances and zero, through the index build's operand code, the kernel's query code and the MFMA.
 * out = { (query, chunk) pairs with a record at or below the threshold, of those WITHOUT the sign (must be 0), pairs flagged, pairs }. */
int pcr_selftest_sphere_f16(pcr_ctx* ctx, int trials, uint64_t out[4]);
/* The library runs a short form of the two self-tests ITSELF, once per context, before it first picks a matrix-core kernel, and only
 * uses a form whose four figures stay within HALF of what its bound assumes (f16 -> bf16 -> the f32 filters, whose bounds need IEEE
 * arithmetic only).  This reports the verdicts (-1 = not run yet; run_now != 0 runs them), the figures, the host time the checks
 * took, and which 1-NN kernel family served the last search ("strack3", "strack", "htrack", "btrack", "etrack", "ftrack",
 * "track", "grid", "grid-stile"). */
typedef struct {
    int32_t f16_ok, bf16_ok;
    double f16_worst[4], bf16_worst[4];
    double check_ms;
    char last_nn1_kernel[16];
} pcr_mfma_check;
int pcr_ctx_mfma_check(pcr_ctx* ctx, int run_now, pcr_mfma_check* out);
/* Tuning / diagnostic knobs by name.  A value of 0 means "library default" for every key except "prof"; results never depend on a knob
 * (the parity tests run the switches against each other), only speed and which kernel serves a call.  Defaults in brackets.
 *  dispatch    nn_method [0 auto: api.cpp nn1_auto_grid] 1 exhaustive / 2 exact grid · nn1_variant [0: table above launch_nn1_brute,
 *              csrc/nn1_brute.hip] 1 FTRACK, 2 TRACK (exact arithmetic only), 4 ETRACK, 6 BTRACK, 7 HTRACK, 8 STRACK for every search
 *              that has or can make itself a seed (HTRACK otherwise), 10 STRACK3 (the sphere form on targets of any size; 9 = its old number) ·
 *              nn1_bf16, nn1_f16 [on] 1 force / 2 forbid the matrix-core forms ·
 *              nn1_sign [on: the sign forms for every search on a target that fits f16 — a cold one seeds itself] 2 = never (HTRACK) ·
 *              nn1_sphere [0: STRACK3 from 32 768 target points] 1 always / 2 never (STRACK) ·
 *              mfma_force_fail 1 f16 / 2 bf16 / 3 both (tests: a failing device check) ·
 *              knn_method, radius_method [auto] 1 exhaustive / 2 grid · nn1_async_in_loop [off] 1 = pcr_nn1_f32_async calls of one
 *              caller-side loop seed each other as the searches inside pcr_icp_p2p_f32 do
 *  exhaustive  nn1_btrack_qg [2 up to 49 152 queries, else 4] · nn1_supers_per_slice [from nn1_btrack_blocks = 14 336] ·
 *              nn1_sign_flush [64: entries of a wave's list of flagged chunks from which the end of a super-tile evaluates them] ·
 *              nn1_sign_dense [12: flagged half-lanes of one (group, tile) from which they evaluate their chunk in place] ·
 *              nn1_sphere_qg [1: groups of 32 queries per STRACK3 wave] 2 / 4 · nn1_sphere_flush_end [1: entries from which the end of
 *              a level-1 super-tile evaluates them] · nn1_sphere_l0_per_slice [from nn1_sphere_blocks = 1 024] ·
 *              nn1_xcd [4] 1 / 2 / 4, -1 plain launch · nn1_cold_seed [on] 2 = off, 3 = round 3's rule (sliced launches only) ·
 *              nn1_seed_mode [0: centre of the nearest super-tile + Morton neighbour for a cold search, the centre alone beside stale correspondences]
 *              1 centre / 2 Morton neighbour / 3 both · nn1_seed_levels [2: the centre scan goes through the level-1 super-tiles' centres first] 1 = all centres · nn1_sphere_reseed [on] 2 = a warm search of the sphere forms keeps stale seeds as they are ·
 *              bt_sort_begin_bit [0: low bits of the Morton key the working-cloud sort ignores] ·
 *              nn1_warm_start [on] 2 = off · nn1_chunks_per_slice [from
 *              nn1_etrack_blocks = 32 768] · nn1_tiles_per_slice [from nn1_target_blocks = 16 384] · bt_sort_work [on] 2 = off
 *  exact grid  grid_order [0: Morton + bounding spheres from 256 points] 1 x-sorted / 2 Morton · grid_mode [by index] 1 plain / 2 x-window / 3 spheres ·
 *              grid_lanes [16] · grid_cell_um, grid_cell_scale_x100 [150 for Morton], grid_occupancy_x10 [20], grid_max_cells ·
 *              grid_sort_queries, grid_sort_work, grid_warm_start, grid_wpos, grid_seed_run, grid_far_brute [on] 2 = off ·
 *              grid_sort_fine [auto] 1 / 2 · grid_query_bins_log2 [22], grid_query_bin_min [2] · grid_xcd_run [32 from 4 096 blocks]
 *              -1 = identity · grid_tile [0: the tile search for targets from 4 000 000 points whose working cloud is about as dense as the target] 1 on / 2 off,
 *              grid_tile_reach_pct [200], grid_tile_min_members [8], grid_tile_list_segs [1] (csrc/grid.hip launch_nn1_grid) · knn_cell_scale_x100, knn_slices ·
 *              the sign tile search (csrc/grid_stile.hpp, what grid_tile selects): grid_stile [on] 2 = off, the cell walk ·
 *              grid_stile_cbits [9: bits per axis of the coarse Morton cells] · bt_fine_bits [6 from 2^20 points] · grid_stile_bmax_cm [100],
 *              grid_stile_lim_pct [400], grid_stile_lim_floor_mm [150], grid_stile_split_mm [40] (ball limits of a pass) · grid_stile_keep [768],
 *              grid_stile_keep_small [192], grid_stile_cells [2 048] (tiles / cells a pass may hold) · grid_stile_flush [64], grid_stile_dense [32] ·
 *              grid_stile_passes [3] · grid_stile_queue [on] 2 = static list walk, grid_stile_list_wgs [8 per CU] ·
 *              grid_stile_l1 [on: per-record operands in the scale of the level-1 super-tiles] 2 = the 256-record super-tiles' ·
 *              grid_stile_cold [on] 2 = off, grid_stile_cold_own [32], grid_stile_cold_per [4] (cold seeds from the coarse cells)
 *  ICP loop    icp_pipeline [0 = 1 device-resident] -1 synchronous · icp_chunk [4] · icp_bounded_search [on] 2 = off ·
 *              icp_fused_move [on] 2 = off, icp_fused_max [262 144] · icp_seed_in_move [on] 2 = off · icp_force_slots (tests) ·
 *              kabsch_bfly [on], kabsch_records [on], kabsch_one_pair_blocks [128], kabsch_max_blocks [1 024]
 *  other       plane_group [20: hypotheses per workgroup row of the plane count] · iss_lanes [32] · radius_fused [on] 2 = off · grid_stats 1 = the next 1-NN launch fills pcr_nn1_stats · prof 0 / 1 / 2 */
int pcr_tune_set(pcr_ctx* ctx, const char* key, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* PCR_H */

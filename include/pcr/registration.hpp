// registration.hpp — the ICP end of Homework9/hw9 (include/registration.hpp, src/registration.cpp) on the MI355X.
//
// Two layers:
//  (0) pcr::GlobalRegistration — descriptor matching + feature RANSAC (the initial pose ICP starts from), arrays only
//  (1) pcr::IcpPoint2Point — dependency-free core with the reference's parameter set
//      (Registration::setICPparams, registration.hpp:126-137) and the semantics of
//      Registration::ICPpoint2point (registration.hpp:204-211, registration.cpp:862-1011): R and t are
//      IN/OUT (pre-loaded with the initial guess, registration.cpp:1141-1142 / :874), clouds are the
//      already-sampled clouds (normal-space sampling, :880-881, is upstream of the hot path).
//  (2) When PCL and Eigen are available (__has_include), free functions with the reference's exact types:
//        void transformCloudInplace(PointCloud&, const Eigen::Matrix3f&, const Eigen::Vector3f&)   registration.hpp:58-60
//        void ICPpoint2point(const Matrix3f& init_R, const Vector3f& init_t, const PointCloud& src,
//                            const PointCloud& tgt, Matrix3f& R, Vector3f& t, ...)                registration.hpp:204-211
//      to be called from Registration::compute in place of the CPU loop.
#ifndef PCR_DROPIN_REGISTRATION_HPP
#define PCR_DROPIN_REGISTRATION_HPP

#include <cstddef>
#include <cstdint>
#include <vector>

#include "pcr_host.hpp"

namespace pcr {

class IcpPoint2Point
{
    // defaults = the shipped parameters, Homework9/hw9/main.cpp:88-95
    int m_ICP_normal_bins = 10;
    size_t m_ICP_sampled_size = 4000;
    float m_ICP_max_corres_dist = 1.0f;
    size_t m_ICP_max_iter = 800;
    float m_ICP_loss_epsilon = 1e-8f;

public:
    pcr_icp_stats last_stats{};

    // same argument order and meaning as Registration::setICPparams
    void setICPparams(int normal_bins, size_t sampled_size, float max_corres_dist, size_t max_iter, float loss_epsilon)
    {
        m_ICP_normal_bins = normal_bins;
        m_ICP_sampled_size = sampled_size;
        m_ICP_max_corres_dist = max_corres_dist;
        m_ICP_max_iter = max_iter;
        m_ICP_loss_epsilon = loss_epsilon;
    }

    // src/tgt: host points in `layout` (PCR_AOS4 == pcl::PointXYZ array, PCR_SOA == column-major N x 3 matrix).
    // R (row-major 3x3) and t: in = initial guess, out = final pose.  Returns the C-ABI status.
    int run(const float* src_xyz, size_t n_src, const float* tgt_xyz, size_t n_tgt, int layout, float R[9], float t[3])
    {
        pcr_ctx* ctx = default_ctx();
        pcr_cloud *cs = nullptr, *ct = nullptr;
        int rc = pcr_cloud_create(ctx, src_xyz, n_src, layout, &cs);
        if (rc == PCR_OK) rc = pcr_cloud_create(ctx, tgt_xyz, n_tgt, layout, &ct);
        if (rc == PCR_OK) {
            const float init_T[16] = { R[0], R[1], R[2], t[0], R[3], R[4], R[5], t[1], R[6], R[7], R[8], t[2], 0, 0, 0, 1 };
            float out_T[16];
            pcr_icp_params prm;
            prm.max_corr = m_ICP_max_corres_dist;
            prm.max_iter = m_ICP_max_iter;
            prm.eps = m_ICP_loss_epsilon;
            rc = pcr_icp_p2p_f32(ctx, cs, ct, init_T, &prm, out_T, &last_stats);
            if (rc == PCR_OK) {
                for (int r = 0; r < 3; ++r) {
                    for (int c = 0; c < 3; ++c) R[3 * r + c] = out_T[4 * r + c];
                    t[r] = out_T[4 * r + 3];
                }
            }
        }
        pcr_cloud_destroy(ctx, cs);
        pcr_cloud_destroy(ctx, ct);
        return rc;
    }
};

// The global-registration front half (next row N4): Registration::setRANSACparams (registration.hpp:115-124),
// findRANSACCorrespondencesUnion (:191-193, registration.cpp:535-615) and RANSAC (:179-185, registration.cpp:288-434)
// with plain arrays in place of pcl::PointCloud<FPFHSignature33> / PointCloud.
class GlobalRegistration
{
    // defaults = reg.setRANSACparams(80000, voxel_size * 4, 10, 0.5) with voxel_size 0.3 (Homework9/hw9/main.cpp:30,86)
    uint32_t m_RANSAC_max_iter = 80000;
    float m_RANSAC_dist_threshold = 1.2f;
    float m_RANSAC_angle_threshold = 10.0f;      // unused by the reference too (:419, commented out)
    float m_RANSAC_corres_rejection_rate = 0.5f;

public:
    uint64_t seed = 5489u;                        // the reference seeds std::mt19937 from std::random_device (:298-299)
    uint32_t max_consensus_set_size = 0;          // what the reference prints with DEBUG (:433)

    void setRANSACparams(const uint32_t max_iter, const float dist_threshold, const float angle_threshold, const float corres_rejection_rate)
    {
        m_RANSAC_max_iter = max_iter;
        m_RANSAC_dist_threshold = dist_threshold;
        m_RANSAC_angle_threshold = angle_threshold;
        m_RANSAC_corres_rejection_rate = corres_rejection_rate;
    }

    // descriptors: row-major n x dim (pcl::FPFHSignature33::histogram rows, dim 33).  correspondences[i] = {idx_src, idx_tar}
    void findRANSACCorrespondencesUnion(const float* fpfh_source, size_t N_source, const float* fpfh_target, size_t N_target, int dim,
                                        std::vector<std::vector<size_t>>& correspondences)
    {
        std::vector<uint32_t> pairs(2 * (N_source + N_target) + 2);
        std::vector<float> dist(N_source + N_target + 1);
        size_t kept = 0;
        check(pcr_match_union_f32(default_ctx(), fpfh_source, N_source, fpfh_target, N_target, dim, m_RANSAC_corres_rejection_rate, pairs.data(),
                                  dist.data(), &kept),
              "pcr_match_union_f32");
        correspondences.resize(kept);
        for (size_t i = 0; i < kept; i++) correspondences[i] = std::vector<size_t>{ pairs[2 * i], pairs[2 * i + 1] };
    }

    // the mutual-nearest-neighbour variant (registration.hpp:187-189, registration.cpp:437-533)
    void findRANSACCorrespondencesInter(const float* fpfh_source, size_t N_source, const float* fpfh_target, size_t N_target, int dim,
                                        std::vector<std::vector<size_t>>& correspondences)
    {
        std::vector<uint32_t> pairs(2 * N_source + 2);
        std::vector<float> dist(N_source + 1);
        size_t kept = 0;
        check(pcr_match_inter_f32(default_ctx(), fpfh_source, N_source, fpfh_target, N_target, dim, m_RANSAC_corres_rejection_rate, pairs.data(),
                                  dist.data(), &kept),
              "pcr_match_inter_f32");
        correspondences.resize(kept);
        for (size_t i = 0; i < kept; i++) correspondences[i] = std::vector<size_t>{ pairs[2 * i], pairs[2 * i + 1] };
    }

    // keypoints: AoS xyz (n x 3).  R (row-major) and t are written only when a non-empty consensus set exists, as in the
    // reference (:423-428).  Returns the C-ABI status.
    int RANSAC(const std::vector<std::vector<size_t>>& correspondences, const float* kp_source_xyz, size_t n_source, const float* kp_target_xyz,
               size_t n_target, float R[9], float t[3])
    {
        const size_t M = correspondences.size();
        std::vector<uint32_t> pairs(2 * M + 2);
        for (size_t i = 0; i < M; i++) {
            pairs[2 * i] = (uint32_t)correspondences[i][0];
            pairs[2 * i + 1] = (uint32_t)correspondences[i][1];
        }
        std::vector<uint32_t> quads(4 * (size_t)m_RANSAC_max_iter + 4);
        int rc = pcr_ransac_sample_quads(kp_source_xyz, n_source, pairs.data(), M, m_RANSAC_max_iter, seed, quads.data());
        if (rc != PCR_OK) return rc;
        int64_t winner = -1;
        return pcr_ransac_global_f32(default_ctx(), kp_source_xyz, n_source, kp_target_xyz, n_target, pairs.data(), M, quads.data(), m_RANSAC_max_iter,
                                     m_RANSAC_dist_threshold, R, t, &max_consensus_set_size, &winner, nullptr);
    }
};

}  // namespace pcr

#if defined(__has_include)
#if __has_include(<pcl/point_cloud.h>) && __has_include(<pcl/point_types.h>) && __has_include(<Eigen/Core>)
#include <Eigen/Core>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>

namespace pcr {

typedef pcl::PointXYZ PointT;
typedef pcl::PointCloud<PointT> PointCloud;
static_assert(sizeof(pcl::PointXYZ) == 4 * sizeof(float), "pcl::PointXYZ is expected to be a 16-byte xyz+pad record");

// registration.cpp:165-178 on the GPU (bit-identical f32 arithmetic)
inline void transformCloudInplace(PointCloud& cloud, const Eigen::Matrix3f& R, const Eigen::Vector3f& t)
{
    pcr_ctx* ctx = default_ctx();
    pcr_cloud* c = nullptr;
    check(pcr_cloud_create(ctx, reinterpret_cast<const float*>(cloud.points.data()), cloud.size(), PCR_AOS4, &c), "pcr_cloud_create");
    const float T[16] = { R(0, 0), R(0, 1), R(0, 2), t(0), R(1, 0), R(1, 1), R(1, 2), t(1), R(2, 0), R(2, 1), R(2, 2), t(2), 0, 0, 0, 1 };
    int rc = pcr_transform_f32(ctx, c, T);
    if (rc == PCR_OK) {
        // PCR_AOS4 read-back writes x, y, z and leaves the 4th float of every record untouched
        rc = pcr_cloud_read(ctx, c, reinterpret_cast<float*>(cloud.points.data()), PCR_AOS4);
    }
    pcr_cloud_destroy(ctx, c);
    check(rc, "pcr_transform_f32");
}

// Registration::ICPpoint2point with the reference's types; `icp` carries the parameters of setICPparams.
inline void ICPpoint2point(IcpPoint2Point& icp, const Eigen::Matrix3f& init_R, const Eigen::Vector3f& init_t,
                           const PointCloud& sampled_cloud_src, const PointCloud& sampled_cloud_tar, Eigen::Matrix3f& R,
                           Eigen::Vector3f& t)
{
    float Rr[9], tr[3];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) Rr[3 * r + c] = init_R(r, c);
        tr[r] = init_t(r);
    }
    check(icp.run(reinterpret_cast<const float*>(sampled_cloud_src.points.data()), sampled_cloud_src.size(),
                  reinterpret_cast<const float*>(sampled_cloud_tar.points.data()), sampled_cloud_tar.size(), PCR_AOS4, Rr, tr),
          "pcr_icp_p2p_f32");
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) R(r, c) = Rr[3 * r + c];
        t(r) = tr[r];
    }
}

}  // namespace pcr
#endif
#endif

#endif  // PCR_DROPIN_REGISTRATION_HPP

// registration.hpp — the ICP end of Homework9/hw9 (include/registration.hpp, src/registration.cpp) on the MI355X.
//
// Two layers:
//  (0) pcr::GlobalRegistration — descriptor matching + feature RANSAC (the initial pose ICP starts from), arrays only
//  (1) pcr::IcpPoint2Point — dependency-free core with the reference's parameter set
//      (Registration::setICPparams, registration.hpp:126-137) and the semantics of
//      Registration::ICPpoint2point (registration.hpp:204-211, registration.cpp:862-1011): R and t are
//      IN/OUT (pre-loaded with the initial guess, registration.cpp:1141-1142 / :874), clouds are the
//      already-sampled clouds (normal-space sampling, :880-881, is upstream of the hot path).
//  (2) When PCL and Eigen are available (__has_include): `class pcr::Registration` with the reference's member signatures —
//        setICPparams(int, size_t, float, size_t, float)                                            registration.hpp:126-137
//        compute(const PointCloud&, const PointCloud&, const NormalCloud&, const NormalCloud&, Matrix3f& R, Vector3f& t)   :139-144
//        ICPpoint2point(const Matrix3f& init_R, const Vector3f& init_t, const PointCloud&, const PointCloud&,
//                       const NormalCloud&, const NormalCloud&, Matrix3f& R, Vector3f& t)           :204-211 (and ICPpoint2plane, RANSAC,
//        findRANSACCorrespondencesUnion / Inter) — plus the free transformCloudInplace / transformNormalsInplace (:58-64).
//      
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

#include <functional>
#include <iostream>

// ---- (2) the reference's own types and signatures (Homework9/hw9/include/registration.hpp) --------------------------------
// Everything below compiles only where PCL + Eigen exist (the reference's build environment).  In this repository it is
// compile- and run-checked against minimal test-only stand-ins for the three headers (tests/mock/, tests/test_registration_class.py):
// a compile check of the SIGNATURES, never a pin of Eigen/PCL numerics.
namespace pcr {

using PointT = pcl::PointXYZ;                                   // registration.hpp:42-44
using PointCloud = pcl::PointCloud<PointT>;
using NormalT = pcl::Normal;                                    // registration.hpp:46-48
using NormalCloud = pcl::PointCloud<pcl::Normal>;
static_assert(sizeof(pcl::PointXYZ) == 4 * sizeof(float), "pcl::PointXYZ is expected to be a 16-byte xyz+pad record");

// registration.hpp:58-60 / registration.cpp:165-178 on the GPU (bit-identical f32 arithmetic)
inline void transformCloudInplace(PointCloud& cloud, const Eigen::Matrix3f& R, const Eigen::Vector3f& t)
{
    if (cloud.size() == 0) return;
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

// registration.hpp:62-64 / registration.cpp:181-194: normal <- R * normal (t unused there too).  Only the sampling stage
// upstream of the hot path reads the result, so this stays a host loop with the reference's row-wise f32 arithmetic.
inline void transformNormalsInplace(NormalCloud& cloud, const Eigen::Matrix3f& R, const Eigen::Vector3f& /*t*/)
{
    for (size_t i = 0; i < cloud.size(); i++) {
        const float x = cloud.points[i].normal_x, y = cloud.points[i].normal_y, z = cloud.points[i].normal_z;
        cloud.points[i].normal_x = (R(0, 0) * x + R(0, 1) * y) + R(0, 2) * z;
        cloud.points[i].normal_y = (R(1, 0) * x + R(1, 1) * y) + R(1, 2) * z;
        cloud.points[i].normal_z = (R(2, 0) * x + R(2, 1) * y) + R(2, 2) * z;
    }
}

// class Registration, registration.hpp:67-251 — same public interface (the seven setters and compute) and the same private
// members for the stages this library accelerates (RANSAC, findRANSACCorrespondencesUnion / Inter, ICPpoint2point,
// ICPpoint2plane), each with the reference's exact signature.  The remaining private stages of the reference are PCL
// library calls outside the hot path (Harris3D / ISS keypoints, FPFH / SHOT descriptors, NormalSpaceSampling,
// VoxelGrid); they are pluggable `stages` so that a maintainer passes the reference's own bodies.
class Registration
{
public:
    Registration() {}
    ~Registration() = default;

    void setISSparams(const float iss_salient_radius, const float iss_non_max_radius, const float iss_gamma_21, const float iss_gamma_32,
                      const int iss_min_neighbors, const int iss_threads)
    {
        m_iss_salient_radius = iss_salient_radius; m_iss_non_max_radius = iss_non_max_radius; m_iss_gamma_21 = iss_gamma_21;
        m_iss_gamma_32 = iss_gamma_32; m_iss_min_neighbors = iss_min_neighbors; m_iss_threads = iss_threads;
    }
    void setHarris3Dparams(const float harris3d_radius, const float harris3d_nms_threshold, const int harris3d_threads, const bool harris3d_is_nms,
                           const bool harris3d_is_refine)
    {
        m_harris3d_radius = harris3d_radius; m_harris3d_nms_threshold = harris3d_nms_threshold; m_harris3d_threads = harris3d_threads;
        m_harris3d_is_nms = harris3d_is_nms; m_harris3d_is_refine = harris3d_is_refine;
    }
    void setFPFHparams(const float fpfh_feature_radius) { m_fpfh_feature_radius = fpfh_feature_radius; }
    void setSHOTparams(const float shot_feature_radius) { m_shot_feature_radius = shot_feature_radius; }
    void setRANSACparams(const uint32_t max_iter, const float dist_threshold, const float angle_threshold, const float corres_rejection_rate)
    {
        m_greg.setRANSACparams(max_iter, dist_threshold, angle_threshold, corres_rejection_rate);
    }
    // registration.hpp:126-137
    void setICPparams(const int normal_bins, const size_t sampled_size, const float max_corres_dist, const size_t max_iter, const float loss_epsilon)
    {
        m_ICP_normal_bins = normal_bins; m_ICP_sampled_size = sampled_size; m_ICP_max_corres_dist = max_corres_dist;
        m_ICP_max_iter = max_iter; m_ICP_loss_epsilon = loss_epsilon;
    }

    // The PCL-internal stages of compute() / ICPpoint2point().  Unset keypoints or fpfh33: compute() has no global
    // registration to run and starts ICP from the identity.  Unset normal_space_sampling: ICP runs on the full clouds (the
    // reference's own commented alternative, registration.cpp:883-884).
    struct Stages {
        // getHarris3DKeypoints (registration.hpp:152-154, registration.cpp:214-251)
        std::function<void(const PointCloud& input_cloud, const NormalCloud& input_normals, PointCloud& keypoints_cloud)> keypoints;
        // getFPFH33Descriptors (registration.hpp:156-159, registration.cpp:253-272)
        std::function<void(const PointCloud& input_cloud, const PointCloud& input_keypoints_cloud, const NormalCloud& input_normals,
                           pcl::PointCloud<pcl::FPFHSignature33>& fpfh_descriptors)> fpfh33;
        // normalSpaceSampling (registration.hpp:170-173, registration.cpp:630-662)
        std::function<void(const PointCloud& input_cloud, const NormalCloud& input_normals, PointCloud& sampled_cloud, NormalCloud& sampled_normals)>
            normal_space_sampling;
    } stages;

    uint64_t ransac_seed = 5489u;          // the reference seeds std::mt19937 from std::random_device (registration.cpp:298-299)
    pcr_icp_stats last_icp_stats{};        // iterations run, kept pairs, ... of the last ICP
    bool use_point2plane = false;          // compute() calls ICPpoint2point (registration.cpp:1150); the sibling is selectable

    // registration.hpp:139-144 / registration.cpp:1014-1157
    void compute(const PointCloud& cloud_source, const PointCloud& cloud_target, const NormalCloud& normals_source, const NormalCloud& normals_target,
                 Eigen::Matrix3f& R, Eigen::Vector3f& t)
    {
        Eigen::Matrix3f init_R = Eigen::Matrix3f::Identity();
        Eigen::Vector3f init_t = Eigen::Vector3f::Zero();
        if (stages.keypoints && stages.fpfh33) {
            PointCloud kp_cloud_source, kp_cloud_target;
            stages.keypoints(cloud_source, normals_source, kp_cloud_source);                     // :1030-1031
            stages.keypoints(cloud_target, normals_target, kp_cloud_target);
            pcl::PointCloud<pcl::FPFHSignature33> fpfh_source, fpfh_target;
            stages.fpfh33(cloud_source, kp_cloud_source, normals_source, fpfh_source);           // :1071-1072
            stages.fpfh33(cloud_target, kp_cloud_target, normals_target, fpfh_target);
            std::vector<std::vector<size_t>> correspondences;
            findRANSACCorrespondencesUnion(fpfh_source, fpfh_target, correspondences);           // :1080
            if (correspondences.size() < 4) {                                                    // :1087-1091
                std::cerr << "Correspondences are fewer than 4! Failed!" << std::endl;
                return;
            }
            RANSAC(correspondences, kp_cloud_source, kp_cloud_target, init_R, init_t);           // :1133
        }
        R = init_R;                                                                              // :1136-1137
        t = init_t;
        if (use_point2plane) ICPpoint2plane(init_R, init_t, cloud_source, cloud_target, normals_source, normals_target, R, t);
        else ICPpoint2point(init_R, init_t, cloud_source, cloud_target, normals_source, normals_target, R, t);   // :1150
    }

private:
    // registration.hpp:179-185 / registration.cpp:288-434
    void RANSAC(const std::vector<std::vector<size_t>>& correspondences, const PointCloud& kp_cloud_source, const PointCloud& kp_cloud_target,
                Eigen::Matrix3f& R, Eigen::Vector3f& t)
    {
        std::vector<float> ks = xyz3(kp_cloud_source), kt = xyz3(kp_cloud_target);
        float Rr[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, tr[3] = { 0, 0, 0 };
        to_rows(R, t, Rr, tr);                          // written only when a non-empty consensus set exists (:423-428)
        m_greg.seed = ransac_seed;
        check(m_greg.RANSAC(correspondences, ks.data(), kp_cloud_source.size(), kt.data(), kp_cloud_target.size(), Rr, tr), "pcr_ransac_global_f32");
        from_rows(Rr, tr, R, t);
    }

    // registration.hpp:187-189 / registration.cpp:437-533
    void findRANSACCorrespondencesInter(const pcl::PointCloud<pcl::FPFHSignature33>& fpfh_source, const pcl::PointCloud<pcl::FPFHSignature33>& fpfh_target,
                                        std::vector<std::vector<size_t>>& correspondences)
    {
        std::vector<float> a = rows33(fpfh_source), b = rows33(fpfh_target);
        m_greg.findRANSACCorrespondencesInter(a.data(), fpfh_source.size(), b.data(), fpfh_target.size(), 33, correspondences);
    }

    // registration.hpp:191-193 / registration.cpp:535-615
    void findRANSACCorrespondencesUnion(const pcl::PointCloud<pcl::FPFHSignature33>& fpfh_source, const pcl::PointCloud<pcl::FPFHSignature33>& fpfh_target,
                                        std::vector<std::vector<size_t>>& correspondences)
    {
        std::vector<float> a = rows33(fpfh_source), b = rows33(fpfh_target);
        m_greg.findRANSACCorrespondencesUnion(a.data(), fpfh_source.size(), b.data(), fpfh_target.size(), 33, correspondences);
    }

    // registration.hpp:195-202 / registration.cpp:710-860
    void ICPpoint2plane(const Eigen::Matrix3f& init_R, const Eigen::Vector3f& init_t, const PointCloud& input_cloud_src, const PointCloud& input_cloud_tar,
                        const NormalCloud& input_normals_src, const NormalCloud& input_normals_tar, Eigen::Matrix3f& R, Eigen::Vector3f& t)
    {
        run_icp(true, init_R, init_t, input_cloud_src, input_cloud_tar, input_normals_src, input_normals_tar, R, t);
    }

    // registration.hpp:204-211 / registration.cpp:862-1011
    void ICPpoint2point(const Eigen::Matrix3f& init_R, const Eigen::Vector3f& init_t, const PointCloud& input_cloud_src, const PointCloud& input_cloud_tar,
                        const NormalCloud& input_normals_src, const NormalCloud& input_normals_tar, Eigen::Matrix3f& R, Eigen::Vector3f& t)
    {
        run_icp(false, init_R, init_t, input_cloud_src, input_cloud_tar, input_normals_src, input_normals_tar, R, t);
    }

    // Both ICP members share their prologue in the reference (:870-881 == :718-729): the source is moved by the R, t passed IN
    // (pre-loaded with the initial guess by compute(), :1136-1137), both clouds are normal-space sampled, T_total starts from
    // init_R / init_t (:910-913).  compute() passes the same pose twice; when a caller does not, the in/out pose moves the
    // source and init_R / init_t seed T_total, as written there.
    void run_icp(bool plane, const Eigen::Matrix3f& init_R, const Eigen::Vector3f& init_t, const PointCloud& input_cloud_src,
                 const PointCloud& input_cloud_tar, const NormalCloud& input_normals_src, const NormalCloud& input_normals_tar, Eigen::Matrix3f& R,
                 Eigen::Vector3f& t)
    {
        float Rin[9], tin[3], Ri[9], ti[3];
        to_rows(R, t, Rin, tin);
        to_rows(init_R, init_t, Ri, ti);
        bool same = true;
        for (int k = 0; k < 9; k++) same = same && Rin[k] == Ri[k];
        for (int k = 0; k < 3; k++) same = same && tin[k] == ti[k];
        const PointCloud* src = &input_cloud_src;
        const PointCloud* tgt = &input_cloud_tar;
        const NormalCloud* ntgt = &input_normals_tar;
        PointCloud trans_cloud_src, sampled_cloud_src, sampled_cloud_tar;
        NormalCloud trans_normals_src, sampled_normals_src, sampled_normals_tar;
        bool moved = false;                                   // the source handed to the library has R, t applied already
        if (stages.normal_space_sampling || !same) {
            trans_cloud_src = input_cloud_src;                                                   // :872-875
            trans_normals_src = input_normals_src;
            transformCloudInplace(trans_cloud_src, R, t);
            transformNormalsInplace(trans_normals_src, R, t);
            moved = true;
            src = &trans_cloud_src;
            if (stages.normal_space_sampling) {                                                  // :880-881
                stages.normal_space_sampling(trans_cloud_src, trans_normals_src, sampled_cloud_src, sampled_normals_src);
                stages.normal_space_sampling(input_cloud_tar, input_normals_tar, sampled_cloud_tar, sampled_normals_tar);
                src = &sampled_cloud_src;
                tgt = &sampled_cloud_tar;
                ntgt = &sampled_normals_tar;
            }
        }
        pcr_ctx* ctx = default_ctx();
        pcr_cloud *cs = nullptr, *ct = nullptr, *cn = nullptr;
        int rc = pcr_cloud_create(ctx, reinterpret_cast<const float*>(src->points.data()), src->size(), PCR_AOS4, &cs);
        if (rc == PCR_OK) rc = pcr_cloud_create(ctx, reinterpret_cast<const float*>(tgt->points.data()), tgt->size(), PCR_AOS4, &ct);
        if (rc == PCR_OK && plane) {
            std::vector<float> n3(3 * ntgt->size() + 3);
            for (size_t i = 0; i < ntgt->size(); i++) {
                n3[3 * i] = ntgt->points[i].normal_x; n3[3 * i + 1] = ntgt->points[i].normal_y; n3[3 * i + 2] = ntgt->points[i].normal_z;
            }
            rc = pcr_cloud_create(ctx, n3.data(), ntgt->size(), PCR_AOS3, &cn);
        }
        float out_T[16];
        if (rc == PCR_OK) {
            const float I16[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
            const float init_T[16] = { Ri[0], Ri[1], Ri[2], ti[0], Ri[3], Ri[4], Ri[5], ti[1], Ri[6], Ri[7], Ri[8], ti[2], 0, 0, 0, 1 };
            pcr_icp_params prm;
            prm.max_corr = m_ICP_max_corres_dist;
            prm.max_iter = m_ICP_max_iter;
            prm.eps = m_ICP_loss_epsilon;
            // the library moves the source by init_T itself and starts T_total from it; a source moved here starts from I
            const float* T0 = moved ? I16 : init_T;
            rc = plane ? pcr_icp_p2plane_f32(ctx, cs, ct, cn, T0, &prm, out_T, &last_icp_stats) : pcr_icp_p2p_f32(ctx, cs, ct, T0, &prm, out_T, &last_icp_stats);
            if (rc == PCR_OK && moved) {
                // T_total = (T_delta_k ... T_delta_1) * [init_R init_t]: the product of the deltas times the seed of :910-913.
                // (The reference folds the seed in first; the f32 rounding of the 4x4 products may differ in the last bit.)
                float seeded[16];
                for (int r = 0; r < 4; r++)
                    for (int c = 0; c < 4; c++) {
                        float acc = out_T[4 * r] * init_T[c];
                        acc = acc + out_T[4 * r + 1] * init_T[4 + c];
                        acc = acc + out_T[4 * r + 2] * init_T[8 + c];
                        acc = acc + out_T[4 * r + 3] * init_T[12 + c];
                        seeded[4 * r + c] = acc;
                    }
                for (int k = 0; k < 16; k++) out_T[k] = seeded[k];
            }
        }
        pcr_cloud_destroy(ctx, cs);
        pcr_cloud_destroy(ctx, ct);
        pcr_cloud_destroy(ctx, cn);
        check(rc, plane ? "pcr_icp_p2plane_f32" : "pcr_icp_p2p_f32");
        for (int r = 0; r < 3; ++r) {                                                            // :1008-1009
            for (int c = 0; c < 3; ++c) R(r, c) = out_T[4 * r + c];
            t(r) = out_T[4 * r + 3];
        }
    }

    static void to_rows(const Eigen::Matrix3f& R, const Eigen::Vector3f& t, float Rr[9], float tr[3])
    {
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) Rr[3 * r + c] = R(r, c);
            tr[r] = t(r);
        }
    }
    static void from_rows(const float Rr[9], const float tr[3], Eigen::Matrix3f& R, Eigen::Vector3f& t)
    {
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) R(r, c) = Rr[3 * r + c];
            t(r) = tr[r];
        }
    }
    static std::vector<float> xyz3(const PointCloud& c)
    {
        std::vector<float> v(3 * c.size() + 3);
        for (size_t i = 0; i < c.size(); i++) { v[3 * i] = c.points[i].x; v[3 * i + 1] = c.points[i].y; v[3 * i + 2] = c.points[i].z; }
        return v;
    }
    static std::vector<float> rows33(const pcl::PointCloud<pcl::FPFHSignature33>& f)
    {
        std::vector<float> v(33 * f.size() + 33);
        for (size_t i = 0; i < f.size(); i++)
            for (int k = 0; k < 33; k++) v[33 * i + k] = f.points[i].histogram[k];
        return v;
    }

private:
    float m_iss_salient_radius = 0.f, m_iss_non_max_radius = 0.f, m_iss_gamma_21 = 0.f, m_iss_gamma_32 = 0.f;
    int m_iss_min_neighbors = 0, m_iss_threads = 0;
    float m_harris3d_radius = 0.f, m_harris3d_nms_threshold = 0.f;
    int m_harris3d_threads = 0;
    bool m_harris3d_is_nms = false, m_harris3d_is_refine = false;
    float m_fpfh_feature_radius = 0.f, m_shot_feature_radius = 0.f;
    // defaults = the shipped parameters, Homework9/hw9/main.cpp:88-95
    int m_ICP_normal_bins = 10;
    size_t m_ICP_sampled_size = 4000;
    float m_ICP_max_corres_dist = 1.0f;
    size_t m_ICP_max_iter = 800;
    float m_ICP_loss_epsilon = 1e-8f;
    GlobalRegistration m_greg;
};

}  // namespace pcr
#endif
#endif

#endif  // PCR_DROPIN_REGISTRATION_HPP

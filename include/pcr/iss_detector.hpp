// iss_detector.hpp — drop-in for Homework7/hw7/include/iss_detector.hpp:15-38 (class ISSKeypoint) on the MI355X path.
//
// The public interface of the reference class uses only std types (MyPCDType = vector<vector<float>>), so the same
// driver code (`main.cpp:82-92`: useWeightedCovMat / setInputPointCloud / setLocalRadius / setNonMaxRadius / setThreshold /
// setMinNeighbors / compute) compiles against this header unchanged — and without Eigen: the eigenvalues are computed on
// the GPU (pcr_iss_keypoints_f32, csrc/iss.hip).  Like the reference, compute() clears `keypoints` and appends the
// keypoints in ascending index of the input cloud (iss_detector.cpp:40,103).
//
// Differences a maintainer should know: the defaults of the reference's private fields are indeterminate (no
// initialisers, iss_detector.hpp:30-35) — here they are the values its driver sets; extra read-only accessors expose the
// per-point results that the reference keeps private (rnn_idx sizes, lambda3_vec_forall).
#ifndef PCR_ISS_DETECTOR_HPP
#define PCR_ISS_DETECTOR_HPP

#include <cstdint>
#include <vector>

#include "pcr_host.hpp"

typedef std::vector<std::vector<float>> MyPCDType;

class ISSKeypoint {
    pcr_iss_params prm_ { 0.12f, 0.08f, 0.9f, 0.9f, 5, 1 };
    MyPCDType cloud_;
    std::vector<uint32_t> key_index_, neighbor_count_;
    std::vector<float> lambda3_;

public:
    void useWeightedCovMat(bool use) { prm_.weighted_covariance = use ? 1 : 0; }
    void setLocalRadius(float r) { prm_.local_radius = r; }
    void setNonMaxRadius(float r) { prm_.non_max_radius = r; }
    void setThreshold(float g21, float g32) { prm_.gamma21 = g21; prm_.gamma32 = g32; }
    void setMinNeighbors(int n) { prm_.min_neighbors = n; }
    void setInputPointCloud(MyPCDType& input_point_cloud) { cloud_ = input_point_cloud; }   // a copy, as in the reference (:35)

    void compute(MyPCDType& keypoints)
    {
        keypoints.clear();
        key_index_.clear();
        const size_t n = cloud_.size();
        lambda3_.assign(n, -1.0f);
        neighbor_count_.assign(n, 0);
        if (n == 0) return;
        std::vector<float> xyz(3 * n);
        for (size_t i = 0; i < n; i++)
            for (int c = 0; c < 3; c++) xyz[3 * i + c] = cloud_[i][c];
        pcr_ctx* ctx = pcr::default_ctx();
        pcr_cloud* dev = nullptr;
        pcr::check(pcr_cloud_create(ctx, xyz.data(), n, PCR_AOS3, &dev), "pcr_cloud_create");
        std::vector<uint8_t> is_key(n);
        const int rc = pcr_iss_keypoints_f32(ctx, dev, &prm_, is_key.data(), lambda3_.data(), neighbor_count_.data(), nullptr);
        pcr_cloud_destroy(ctx, dev);
        pcr::check(rc, "pcr_iss_keypoints_f32");
        for (size_t i = 0; i < n; i++)
            if (is_key[i]) {
                keypoints.emplace_back(cloud_[i]);
                key_index_.push_back((uint32_t)i);
            }
    }

    // not in the reference: what compute() found, per input point
    const std::vector<uint32_t>& keypointIndices() const { return key_index_; }
    const std::vector<uint32_t>& neighborCounts() const { return neighbor_count_; }   // rnn_idx[i].size()
    const std::vector<float>& lambda3() const { return lambda3_; }                    // lambda3_vec_forall
};

#endif  // PCR_ISS_DETECTOR_HPP

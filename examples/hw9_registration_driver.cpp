// hw9_registration_driver.cpp — the registration driver of Homework9/hw9/main.cpp (doRegistration :19-122 and the CSV
// row of processDataSet :152-165) reduced to the part that is on the hot path: read a source / target pair, run
// point-to-point ICP with the shipped parameters (main.cpp:88-95) on the MI355X, print
//     idx1,idx2,t_x,t_y,t_z,q_w,q_x,q_y,q_z
// Everything upstream in the reference (voxel grid, Harris, FPFH, feature RANSAC, normal-space sampling) needs PCL and
// is out of scope; the initial pose is the identity unless given.
//   usage: hw9_registration_driver <src.bin> <tgt.bin> <floats_per_point: 4 (KITTI x y z i) | 6 (hw9 x y z nx ny nz)>
//                                  [idx_src idx_tgt [max_iter]]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "registration.hpp"

// file layout: registration.cpp:25-26 (6 floats) or test.hpp:26-28 (4 floats); returns the rows unchanged
static std::vector<float> read_cloud(const std::string& path, int floats_per_point)
{
    std::ifstream in(path, std::ios::binary);
    if (!in.good()) { std::cerr << "Read file " << path << " failed!" << std::endl; std::exit(EXIT_FAILURE); }
    in.seekg(0, std::ios::end);
    const size_t bytes = (size_t)in.tellg();
    in.seekg(0, std::ios::beg);
    std::vector<float> raw(bytes / sizeof(float));
    in.read(reinterpret_cast<char*>(raw.data()), (std::streamsize)(raw.size() * sizeof(float)));
    raw.resize(raw.size() / (size_t)floats_per_point * (size_t)floats_per_point);
    return raw;      // handed to the library as is: PCR_AOS4 / PCR_AOS6 rows
}

// Eigen::Quaternionf(R) (main.cpp:121): the standard trace-based conversion, w >= 0 branch first
static void quaternion_from_R(const float R[9], float q[4])
{
    const float t = R[0] + R[4] + R[8];
    if (t > 0.0f) {
        float s = std::sqrt(t + 1.0f);
        q[0] = 0.5f * s;
        s = 0.5f / s;
        q[1] = (R[7] - R[5]) * s; q[2] = (R[2] - R[6]) * s; q[3] = (R[3] - R[1]) * s;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        float s = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0f);
        float v[3];
        v[i] = 0.5f * s;
        s = 0.5f / s;
        q[0] = (R[3 * k + j] - R[3 * j + k]) * s;
        v[j] = (R[3 * j + i] + R[3 * i + j]) * s;
        v[k] = (R[3 * k + i] + R[3 * i + k]) * s;
        q[1] = v[0]; q[2] = v[1]; q[3] = v[2];
    }
}

int main(int argc, char** argv)
{
    if (argc < 4) {
        std::cerr << "usage: " << argv[0] << " src.bin tgt.bin floats_per_point [idx_src idx_tgt [max_iter]]" << std::endl;
        return 2;
    }
    const int fpp = std::atoi(argv[3]);
    if (fpp != 4 && fpp != 6) { std::cerr << "floats_per_point must be 4 or 6" << std::endl; return 2; }
    const std::string idx_src = argc > 4 ? argv[4] : "0", idx_tgt = argc > 5 ? argv[5] : "1";
    const size_t max_iter = argc > 6 ? (size_t)std::atol(argv[6]) : 800;
    std::vector<float> src = read_cloud(argv[1], fpp), tgt = read_cloud(argv[2], fpp);

    pcr::IcpPoint2Point reg;
    reg.setICPparams(10, 4000, 1.0f, max_iter, 1e-8f);              // main.cpp:88-95
    float R[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, t[3] = { 0, 0, 0 };
    const int rc = reg.run(src.data(), src.size() / (size_t)fpp, tgt.data(), tgt.size() / (size_t)fpp, fpp == 4 ? PCR_AOS4 : PCR_AOS6, R, t);
    if (rc != PCR_OK) { std::cerr << "ICP failed, rc = " << rc << std::endl; return 1; }
    float q[4];
    quaternion_from_R(R, q);
    std::cerr << "ICP: " << reg.last_stats.iters_run << " iterations, " << reg.last_stats.last_pairs << " pairs"
              << (reg.last_stats.converged ? ", converged" : ", max_iter reached") << ", " << reg.last_stats.ms_total << " ms" << std::endl;
    std::printf("idx1,idx2,t_x,t_y,t_z,q_w,q_x,q_y,q_z\n");
    std::printf("%s,%s,%.9g,%.9g,%.9g,%.9g,%.9g,%.9g,%.9g\n", idx_tgt.c_str(), idx_src.c_str(), t[0], t[1], t[2], q[0], q[1], q[2], q[3]);
    return 0;
}

// Compile + run check of the PCL/Eigen-typed block of include/pcr/registration.hpp (class pcr::Registration) against the
// test-only stand-in headers of tests/mock/.  The member signatures are asserted against
// Homework9/hw9/include/registration.hpp:58-64,115-144,179-211 as TYPES; the run drives compute() exactly as
// Homework9/hw9/main.cpp:88-100 does and writes the pose for tests/test_registration_class.py to compare with the C ABI.
//
//   registration_class_check <scene.bin> <out.bin> <mode>
//     scene.bin: u32 n_src, n_tgt, n_kp_src, n_kp_tgt, max_iter; then f32 src[n_src*4], tgt[n_tgt*4] (pcl::PointXYZ records),
//                normals_tgt[n_tgt*3], kp_src[n_kp_src*3], kp_tgt[n_kp_tgt*3], fpfh_src[n_kp_src*33], fpfh_tgt[n_kp_tgt*33]
//     mode 0: no stages (ICP from the identity on the full clouds); 1: keypoints + fpfh33 stages (global registration -> ICP);
//          2: mode 1 + normal_space_sampling = every third point; 3: mode 0 with the point-to-plane sibling
//     out.bin: f32 R[9] row-major, t[3], u64 iters_run, u64 last_pairs
#define private public          // test-only: the reference keeps ICPpoint2point / RANSAC / find* private (registration.hpp:146-211)
#include "registration.hpp"
#undef private

#include <cstdio>
#include <cstring>
#include <type_traits>

using pcr::NormalCloud;
using pcr::PointCloud;
using pcr::Registration;
using M3 = Eigen::Matrix3f;
using V3 = Eigen::Vector3f;
using Fpfh = pcl::PointCloud<pcl::FPFHSignature33>;
using Corr = std::vector<std::vector<size_t>>;

// ---- the signatures, as types ---------------------------------------------------------------------------------------------
static_assert(std::is_same<decltype(&pcr::transformCloudInplace), void (*)(PointCloud&, const M3&, const V3&)>::value, "registration.hpp:58-60");
static_assert(std::is_same<decltype(&pcr::transformNormalsInplace), void (*)(NormalCloud&, const M3&, const V3&)>::value, "registration.hpp:62-64");
static_assert(std::is_same<decltype(&Registration::setRANSACparams), void (Registration::*)(const uint32_t, const float, const float, const float)>::value,
              "registration.hpp:115-124");
static_assert(std::is_same<decltype(&Registration::setICPparams), void (Registration::*)(const int, const size_t, const float, const size_t, const float)>::value,
              "registration.hpp:126-137");
static_assert(std::is_same<decltype(&Registration::compute),
                           void (Registration::*)(const PointCloud&, const PointCloud&, const NormalCloud&, const NormalCloud&, M3&, V3&)>::value,
              "registration.hpp:139-144");
static_assert(std::is_same<decltype(&Registration::RANSAC), void (Registration::*)(const Corr&, const PointCloud&, const PointCloud&, M3&, V3&)>::value,
              "registration.hpp:179-185");
static_assert(std::is_same<decltype(&Registration::findRANSACCorrespondencesInter), void (Registration::*)(const Fpfh&, const Fpfh&, Corr&)>::value,
              "registration.hpp:187-189");
static_assert(std::is_same<decltype(&Registration::findRANSACCorrespondencesUnion), void (Registration::*)(const Fpfh&, const Fpfh&, Corr&)>::value,
              "registration.hpp:191-193");
using IcpMember = void (Registration::*)(const M3&, const V3&, const PointCloud&, const PointCloud&, const NormalCloud&, const NormalCloud&, M3&, V3&);
static_assert(std::is_same<decltype(&Registration::ICPpoint2plane), IcpMember>::value, "registration.hpp:195-202");
static_assert(std::is_same<decltype(&Registration::ICPpoint2point), IcpMember>::value, "registration.hpp:204-211");

static bool read_all(FILE* f, void* p, size_t bytes) { return bytes == 0 || fread(p, 1, bytes, f) == bytes; }

int main(int argc, char** argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s scene.bin out.bin mode\n", argv[0]); return 2; }
    const int mode = atoi(argv[3]);
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("scene"); return 2; }
    uint32_t h[5];
    if (!read_all(f, h, sizeof h)) return 2;
    const size_t ns = h[0], nt = h[1], ks = h[2], kt = h[3];
    PointCloud src, tgt, kp_src, kp_tgt;
    NormalCloud n_src, n_tgt;
    Fpfh d_src, d_tgt;
    src.points.resize(ns); tgt.points.resize(nt); n_src.points.resize(ns); n_tgt.points.resize(nt);
    kp_src.points.resize(ks); kp_tgt.points.resize(kt); d_src.points.resize(ks); d_tgt.points.resize(kt);
    static_assert(sizeof(pcl::PointXYZ) == 16 && sizeof(pcl::FPFHSignature33) == 132, "record layouts");
    std::vector<float> buf;
    bool ok = read_all(f, src.points.data(), ns * 16) && read_all(f, tgt.points.data(), nt * 16);
    buf.resize(3 * nt + 3);
    ok = ok && read_all(f, buf.data(), nt * 12);
    for (size_t i = 0; i < nt; i++) { n_tgt.points[i].normal_x = buf[3 * i]; n_tgt.points[i].normal_y = buf[3 * i + 1]; n_tgt.points[i].normal_z = buf[3 * i + 2]; }
    buf.resize(3 * (ks + kt) + 3);
    ok = ok && read_all(f, buf.data(), (ks + kt) * 12);
    for (size_t i = 0; i < ks; i++) kp_src.points[i] = pcl::PointXYZ(buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]);
    for (size_t i = 0; i < kt; i++) kp_tgt.points[i] = pcl::PointXYZ(buf[3 * (ks + i)], buf[3 * (ks + i) + 1], buf[3 * (ks + i) + 2]);
    ok = ok && read_all(f, d_src.points.data(), ks * 132) && read_all(f, d_tgt.points.data(), kt * 132);
    fclose(f);
    if (!ok) { fprintf(stderr, "short scene file\n"); return 2; }

    Registration reg;                                                         // main.cpp:66-95
    reg.setHarris3Dparams(0.6f, 1e-8f, 4, true, false);
    reg.setFPFHparams(1.2f);
    reg.setRANSACparams(6000, 0.3f, 10, 0.5f);
    reg.setICPparams(10, 4000, 1, h[4], 1e-8f);
    reg.ransac_seed = 99;
    if (mode == 1 || mode == 2) {
        // the PCL stages, replaced here by recorded outputs: the keypoints / descriptors of whichever cloud is asked for
        reg.stages.keypoints = [&](const PointCloud& cloud, const NormalCloud&, PointCloud& out) { out = (&cloud == &src) ? kp_src : kp_tgt; };
        reg.stages.fpfh33 = [&](const PointCloud& cloud, const PointCloud&, const NormalCloud&, Fpfh& out) { out = (&cloud == &src) ? d_src : d_tgt; };
    }
    if (mode == 2)
        reg.stages.normal_space_sampling = [](const PointCloud& c, const NormalCloud& n, PointCloud& sc, NormalCloud& sn) {
            sc.points.clear(); sn.points.clear();
            for (size_t i = 0; i < c.size(); i += 3) { sc.points.push_back(c.points[i]); sn.points.push_back(n.points[i]); }
        };
    reg.use_point2plane = mode == 3;
    Eigen::Matrix3f R;                                                        // main.cpp:97-99
    Eigen::Vector3f t;
    reg.compute(src, tgt, n_src, n_tgt, R, t);

    // transformCloudInplace leaves the 4th float of every record alone
    PointCloud moved = src;
    pcr::transformCloudInplace(moved, R, t);
    for (size_t i = 0; i < moved.size(); i++)
        if (moved.points[i].data[3] != src.points[i].data[3]) { fprintf(stderr, "pad float changed\n"); return 3; }

    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror("out"); return 2; }
    float Rt[12];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rt[3 * r + c] = R(r, c); Rt[9 + r] = t(r); }
    const uint64_t st[2] = { reg.last_icp_stats.iters_run, reg.last_icp_stats.last_pairs };
    fwrite(Rt, sizeof Rt, 1, o);
    fwrite(st, sizeof st, 1, o);
    fwrite(&moved.points[0], 16, moved.size() < 64 ? moved.size() : 64, o);
    fclose(o);
    printf("registration_class_check ok: mode %d, %llu iterations, %llu pairs\n", mode, (unsigned long long)st[0], (unsigned long long)st[1]);
    return 0;
}

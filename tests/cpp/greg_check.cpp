// greg_check.cpp — the sequence Registration::compute runs before ICP (registration.cpp:1082,1138,1141-1145):
// findRANSACCorrespondencesUnion -> RANSAC -> ICPpoint2point, through include/pcr/registration.hpp.
// usage: greg_check in.bin out.bin
//   in : int64 n_src, n_tgt, dim, max_iter; float thr, rate; uint64 seed; float kp_src[n_src][3], kp_tgt[n_tgt][3], desc_src[n_src][dim], desc_tgt[n_tgt][dim]
//   out: int64 n_corr; uint32 max_consensus; float R[9], t[3] (RANSAC); float R[9], t[3] (after ICP on the keypoints)
#include <cstdio>
#include <vector>

#include "registration.hpp"

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    long long hdr[4];
    float thr_rate[2];
    unsigned long long seed;
    if (fread(hdr, 8, 4, f) != 4 || fread(thr_rate, 4, 2, f) != 2 || fread(&seed, 8, 1, f) != 1) return 2;
    const size_t ns = (size_t)hdr[0], nt = (size_t)hdr[1];
    const int dim = (int)hdr[2];
    std::vector<float> ks(3 * ns), kt(3 * nt), ds(ns * dim), dt(nt * dim);
    if (fread(ks.data(), 4, ks.size(), f) != ks.size() || fread(kt.data(), 4, kt.size(), f) != kt.size() ||
        fread(ds.data(), 4, ds.size(), f) != ds.size() || fread(dt.data(), 4, dt.size(), f) != dt.size())
        return 2;
    fclose(f);

    pcr::GlobalRegistration reg;
    reg.setRANSACparams((uint32_t)hdr[3], thr_rate[0], 10, thr_rate[1]);
    reg.seed = seed;
    std::vector<std::vector<size_t>> correspondences;
    reg.findRANSACCorrespondencesUnion(ds.data(), ns, dt.data(), nt, dim, correspondences);
    float R[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, t[3] = { 0, 0, 0 };
    pcr::check(reg.RANSAC(correspondences, ks.data(), ns, kt.data(), nt, R, t), "RANSAC");
    std::printf("max_consensus_set_size = %u\n", reg.max_consensus_set_size);

    float R2[9], t2[3];
    for (int k = 0; k < 9; k++) R2[k] = R[k];
    for (int k = 0; k < 3; k++) t2[k] = t[k];
    pcr::IcpPoint2Point icp;
    icp.setICPparams(10, 4000, 1.0f, 30, 1e-8f);
    pcr::check(icp.run(ks.data(), ns, kt.data(), nt, PCR_AOS3, R2, t2), "ICP");

    FILE* o = fopen(argv[2], "wb");
    if (!o) return 2;
    long long nc = (long long)correspondences.size();
    fwrite(&nc, 8, 1, o);
    fwrite(&reg.max_consensus_set_size, 4, 1, o);
    fwrite(R, 4, 9, o); fwrite(t, 4, 3, o); fwrite(R2, 4, 9, o); fwrite(t2, 4, 3, o);
    fclose(o);
    return 0;
}

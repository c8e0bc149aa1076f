// iss_check.cpp — drives the drop-in ISSKeypoint exactly as Homework7/hw7/main.cpp:82-92 drives the reference class
// (the PLY reader and the viewer need PCL; the cloud comes from a raw float file instead).
// usage: iss_check in.bin out.bin     in: int64 n, float xyz[n][3]     out: int64 k, int32 idx[k], float lambda3[n], uint32 cnt[n], float keypoints[k][3]
#include <cstdio>
#include <cstdlib>

#include "iss_detector.hpp"

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    long long n = 0;
    if (fread(&n, 8, 1, f) != 1) return 2;
    MyPCDType point_cloud((size_t)n, std::vector<float>(3));
    for (long long i = 0; i < n; i++)
        if (fread(point_cloud[i].data(), 4, 3, f) != 3) return 2;
    fclose(f);

    float test = 0.02;
    ISSKeypoint iss_detector;
    iss_detector.useWeightedCovMat(true);
    iss_detector.setInputPointCloud(point_cloud);
    iss_detector.setLocalRadius(6 * test);
    iss_detector.setNonMaxRadius(4 * test);
    iss_detector.setThreshold(0.9, 0.9);
    iss_detector.setMinNeighbors(5);
    MyPCDType keypoints;
    iss_detector.compute(keypoints);
    std::printf("key points size : %zu\n", keypoints.size());

    FILE* o = fopen(argv[2], "wb");
    if (!o) return 2;
    long long k = (long long)keypoints.size();
    fwrite(&k, 8, 1, o);
    fwrite(iss_detector.keypointIndices().data(), 4, (size_t)k, o);
    fwrite(iss_detector.lambda3().data(), 4, (size_t)n, o);
    fwrite(iss_detector.neighborCounts().data(), 4, (size_t)n, o);
    for (const auto& p : keypoints) fwrite(p.data(), 4, 3, o);
    fclose(o);
    return 0;
}

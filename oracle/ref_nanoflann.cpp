// ref_nanoflann.cpp — harness around the REFERENCE's vendored nanoflann 1.3.2
// (Homework9/hw9/include/nanoflann.hpp, NANOFLANN_VERSION 0x132 at :62) and its
// vector-of-vectors adaptor (Homework3/nano_vs_my/include/KDTreeVectorOfVectorsAdaptor.h),
// compiled from where they lie under /root/reference (never copied).  Test infrastructure
// only; output goes to oracle/_ref/.
//
// ref_nano_nn1_f32 instantiates exactly what ICPpoint2point uses
// (Homework9/hw9/src/registration.cpp:903-906,925-934):
//   KDTreeEigenMatrixAdaptor<MatrixXf> == KDTreeSingleIndexAdaptor<L2_Adaptor<float,Self>,Self,-1,size_t>
//   (nanoflann.hpp:1963-1965) over an N x 3 column-major matrix; Eigen is absent here, so the
//   adaptor below supplies the same kdtree_get_pt(idx, dim) = coeff(idx, dim) over three SoA columns.
#include <cstdint>
#include <cstddef>
#include <vector>
#include <thread>
#include <chrono>
#include <algorithm>

#include <nanoflann.hpp>
#include "KDTreeVectorOfVectorsAdaptor.h"

namespace {

struct SoAMat {
    const float* col[3];
    size_t n;
    typedef nanoflann::metric_L2::traits<float, SoAMat>::distance_t metric_t;   // L2_Adaptor<float,SoAMat,float>
    typedef nanoflann::KDTreeSingleIndexAdaptor<metric_t, SoAMat, -1, size_t> index_t;
    inline size_t kdtree_get_point_count() const { return n; }
    inline float kdtree_get_pt(const size_t idx, size_t dim) const { return col[dim][idx]; }
    template <class BBOX> bool kdtree_get_bbox(BBOX&) const { return false; }
};

// row-major n x dim matrix: what KDTreeEigenMatrixAdaptor<MatrixXf> (nanoflann.hpp:1963-2030) presents for the
// 33-D descriptor matrices of findRANSACCorrespondencesUnion (registration.cpp:562,580): same metric, DIM = -1
struct RowMat {
    const float* p;
    size_t n;
    int dim;
    typedef nanoflann::metric_L2::traits<float, RowMat>::distance_t metric_t;
    typedef nanoflann::KDTreeSingleIndexAdaptor<metric_t, RowMat, -1, size_t> index_t;
    inline size_t kdtree_get_point_count() const { return n; }
    inline float kdtree_get_pt(const size_t idx, size_t d) const { return p[idx * (size_t)dim + d]; }
    template <class BBOX> bool kdtree_get_bbox(BBOX&) const { return false; }
};

double now_ms()
{
    return std::chrono::duration<double, std::milli>(
               std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" {

// 1-NN of every source point in the target cloud, f32, leaf_max_size as given (ICP uses 2).
// threads <= 1: single-threaded like the reference; > 1: queries split over std::thread
// (findNeighbors is const).  idx = UINT32_MAX / d2 = FLT_MAX-initialised when nothing is accepted.
int ref_nano_nn1_f32(const float* tx, const float* ty, const float* tz, size_t nt,
                     const float* sx, const float* sy, const float* sz, size_t ns,
                     int leaf_max_size, int threads,
                     uint32_t* idx, float* d2, double* build_ms, double* query_ms)
{
    SoAMat mat{ { tx, ty, tz }, nt };
    double t0 = now_ms();
    SoAMat::index_t index(3, mat, nanoflann::KDTreeSingleIndexAdaptorParams(leaf_max_size));
    index.buildIndex();
    double t1 = now_ms();
    auto work = [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            size_t ri = (size_t)-1;
            float rd = 0.0f;
            nanoflann::KNNResultSet<float> rs(1);
            rs.init(&ri, &rd);
            float q[3] = { sx[i], sy[i], sz[i] };
            index.findNeighbors(rs, &q[0], nanoflann::SearchParams(10));
            idx[i] = (ri == (size_t)-1) ? UINT32_MAX : (uint32_t)ri;
            d2[i] = rd;
        }
    };
    if (threads <= 1) {
        work(0, ns);
    } else {
        std::vector<std::thread> pool;
        size_t chunk = (ns + threads - 1) / threads;
        for (int t = 0; t < threads; t++) {
            size_t lo = std::min(ns, (size_t)t * chunk), hi = std::min(ns, lo + chunk);
            pool.emplace_back(work, lo, hi);
        }
        for (auto& th : pool) th.join();
    }
    double t2 = now_ms();
    if (build_ms) *build_ms = t1 - t0;
    if (query_ms) *query_ms = t2 - t1;
    return 0;
}

// 1-NN of m dim-D queries in an n x dim row-major database through the vendored nanoflann, configured as
// registration.cpp:562-575 does (leaf 2, KNNResultSet<float>(1), SearchParams(10)).
int ref_nano_nn1_dim_f32(const float* db, size_t n, const float* q, size_t m, int dim, int leaf_max_size,
                         uint32_t* idx, float* d2)
{
    RowMat mat{ db, n, dim };
    RowMat::index_t index(dim, mat, nanoflann::KDTreeSingleIndexAdaptorParams(leaf_max_size));
    index.buildIndex();
    for (size_t i = 0; i < m; i++) {
        size_t ri = (size_t)-1;
        float rd = 0.0f;
        nanoflann::KNNResultSet<float> rs(1);
        rs.init(&ri, &rd);
        std::vector<float> query(q + i * (size_t)dim, q + (i + 1) * (size_t)dim);
        index.findNeighbors(rs, &query[0], nanoflann::SearchParams(10));
        idx[i] = (ri == (size_t)-1) ? UINT32_MAX : (uint32_t)ri;
        d2[i] = rd;
    }
    return 0;
}

// k-NN through KDTreeVectorOfVectorsAdaptor<vector<vector<double>>, double>::query
// (KDTreeVectorOfVectorsAdaptor.h:80-85; driver Homework3/nano_vs_my/main.cpp:62-76).
// idx: m*k (uint64), d2: m*k squared distances.
int ref_nano_knn_f64(const double* dbp, size_t n, int dim, const double* qp, size_t m, int k,
                     int leaf_max_size, uint64_t* idx, double* d2)
{
    typedef std::vector<std::vector<double>> VV;
    VV db(n);
    for (size_t i = 0; i < n; i++) db[i].assign(dbp + i * dim, dbp + (i + 1) * dim);
    typedef KDTreeVectorOfVectorsAdaptor<VV, double> kd_t;
    kd_t mat_index(dim, db, leaf_max_size);
    mat_index.index->buildIndex();
    std::vector<size_t> ri(k);
    for (size_t i = 0; i < m; i++) {
        std::fill(ri.begin(), ri.end(), (size_t)-1);
        mat_index.query(qp + i * dim, k, ri.data(), d2 + i * k);
        for (int s = 0; s < k; s++) idx[i * k + s] = (uint64_t)ri[s];
    }
    return 0;
}

}  // extern "C"

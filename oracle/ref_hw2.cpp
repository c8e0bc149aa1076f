// ref_hw2.cpp — harness around the REFERENCE's own hw2 headers, compiled from where they lie
// under /root/reference (never copied).  Test infrastructure only; output goes to oracle/_ref/.
//
// Wraps: KDTreeConstruction / KDTreeKNNSearch / KDTreeRadiusNNSearch / KDTreeDestruction
//        (Homework2/hw2/include/kdtree.hpp:329,367,419,431), the result sets
//        (resultSet.hpp:28-142), the octree twins (octree.hpp) and readBinary (test.hpp:11-33).
// Must be built with -std=c++14 (kdtree.hpp:58 uses `register`).
#include <vector>
#include <string>
#include <cstdint>
#include <cstring>
#include <chrono>

#include "test.hpp"   // Homework2/hw2/include/test.hpp -> resultSet.hpp, bst.hpp, kdtree.hpp, octree.hpp

typedef std::vector<std::vector<double>> Db;

static Db to_db(const double* p, size_t n, int dim)
{
    Db db(n);
    for (size_t i = 0; i < n; i++) db[i].assign(p + i * dim, p + (i + 1) * dim);
    return db;
}

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(
               std::chrono::steady_clock::now().time_since_epoch()).count();
}

extern "C" {

// k-NN through the reference kd-tree. idx/dist: m*k (the result set's distIndexList verbatim,
// i.e. visit-order tie behaviour of resultSet.hpp:65-91). cmp: m comparison counts (may be NULL).
int ref_hw2_kd_knn(const double* dbp, size_t n, int dim, const double* qp, size_t m, int k,
                   int leaf_size, int32_t* idx, double* dist, int32_t* cmp,
                   double* build_ms, double* query_ms)
{
    Db db = to_db(dbp, n, dim);
    double t0 = now_ms();
    Node* root = KDTreeConstruction(db, leaf_size);
    double t1 = now_ms();
    for (size_t i = 0; i < m; i++) {
        std::vector<double> query(qp + i * dim, qp + (i + 1) * dim);
        KNNResultSet rs(k);
        KDTreeKNNSearch(root, db, rs, query);
        for (int s = 0; s < k; s++) {
            idx[i * k + s] = rs.distIndexList[s].index;
            dist[i * k + s] = rs.distIndexList[s].distance;
        }
        if (cmp) cmp[i] = rs.comparisionCount;
    }
    double t2 = now_ms();
    KDTreeDestruction();
    Node::address_set.clear();   // the registry is a global static (kdtree.hpp:33,418); make the call repeatable
    if (build_ms) *build_ms = t1 - t0;
    if (query_ms) *query_ms = t2 - t1;
    return 0;
}

// radius search through the reference kd-tree; CSR in VISIT order (as the reference emits).
// Two-call pattern: idx == NULL -> only row_ptr (m+1) is written.
int ref_hw2_kd_radius(const double* dbp, size_t n, int dim, const double* qp, size_t m, double r,
                      int leaf_size, int64_t* row_ptr, int32_t* idx, double* dist)
{
    Db db = to_db(dbp, n, dim);
    Node* root = KDTreeConstruction(db, leaf_size);
    int64_t w = 0;
    for (size_t i = 0; i < m; i++) {
        std::vector<double> query(qp + i * dim, qp + (i + 1) * dim);
        RadiusNNResultSet rs(r);
        KDTreeRadiusNNSearch(root, db, rs, query);
        row_ptr[i] = w;
        for (auto& di : rs.distIndexList) {
            if (idx) idx[w] = di.index;
            if (dist) dist[w] = di.distance;
            w++;
        }
    }
    row_ptr[m] = w;
    KDTreeDestruction();
    Node::address_set.clear();
    return 0;
}

// the octree twin, as a cross-check oracle (octree.hpp)
int ref_hw2_oct_knn(const double* dbp, size_t n, int dim, const double* qp, size_t m, int k,
                    int leaf_size, double min_extent, int32_t* idx, double* dist)
{
    Db db = to_db(dbp, n, dim);
    Octant* root = OctreeConstruction(db, leaf_size, min_extent);
    for (size_t i = 0; i < m; i++) {
        std::vector<double> query(qp + i * dim, qp + (i + 1) * dim);
        KNNResultSet rs(k);
        OctreeKNNSearch(root, db, rs, query);
        for (int s = 0; s < k; s++) {
            idx[i * k + s] = rs.distIndexList[s].index;
            dist[i * k + s] = rs.distIndexList[s].distance;
        }
    }
    OctreeDestruction();
    Octant::address_set.clear();   // same global-registry pattern as the kd-tree (octree.hpp:32,216)
    return 0;
}

// readBinary (test.hpp:11-33) incl. its EOF behaviour. Returns the point count; copies up to cap points.
int64_t ref_hw2_read_binary(const char* path, double* out_xyz, size_t cap)
{
    Db pts = readBinary(std::string(path));
    size_t n = pts.size();
    for (size_t i = 0; i < n && i < cap; i++)
        for (int c = 0; c < 3; c++) out_xyz[3 * i + c] = pts[i][c];
    return (int64_t)n;
}

}  // extern "C"

// ref_hw7.cpp — harness around the REFERENCE's own hw7 float kd-tree (Homework7/hw7/src/kdtree.cpp + resultSet.cpp,
// include/kdtree.hpp, include/resultSet.hpp), compiled from where it lies under /root/reference (never copied).
// Test infrastructure only; output goes to oracle/_ref/libhw7_ref.so.  This is the neighbourhood half of
// ISSKeypoint::compute (iss_detector.cpp:45-57, :90-92); the Eigen half of hw7 is unbuildable here.
#include <cstdint>
#include <vector>

#include "kdtree.hpp"   // Homework7/hw7/include/kdtree.hpp (ElemType = float)

extern "C" {

// Radius search of every query through the reference kd-tree (leaf_size 12 in iss_detector.cpp:45).
// row_ptr: m + 1 entries.  With idx == NULL only row_ptr is filled; otherwise idx/dist (row_ptr[m] entries) receive the
// result set in the reference's tree-visit order.
int ref_hw7_radius(const float* dbp, size_t n, const float* qp, size_t m, float r, int leaf_size,
                   int64_t* row_ptr, int32_t* idx, float* dist)
{
    std::vector<std::vector<float>> db(n, std::vector<float>(3));
    for (size_t i = 0; i < n; i++) db[i].assign(dbp + 3 * i, dbp + 3 * i + 3);
    Node* root = KDTreeConstruction(db, leaf_size);
    int64_t at = 0;
    for (size_t i = 0; i < m; i++) {
        std::vector<float> query(qp + 3 * i, qp + 3 * i + 3);
        RadiusNNResultSet rs(r);
        KDTreeRadiusNNSearch(root, db, rs, query);
        row_ptr[i] = at;
        if (idx)
            for (const auto& di : rs.distIndexList) { idx[at] = di.index; dist[at] = di.distance; at++; }
        else
            at += rs.size();
    }
    row_ptr[m] = at;
    KDTreeDestruction();
    Node::address_set.clear();   // the registry is a global static (kdtree.cpp:387,400-404); make the call repeatable
    return 0;
}

}  // extern "C"

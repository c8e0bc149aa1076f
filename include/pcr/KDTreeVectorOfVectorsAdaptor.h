// KDTreeVectorOfVectorsAdaptor.h — drop-in for Homework3/nano_vs_my/include/KDTreeVectorOfVectorsAdaptor.h:
// the vector-of-vectors front end of the nanoflann-shaped index (include/pcr/nanoflann.hpp).
//   KDTreeVectorOfVectorsAdaptor<VecOfVec, num_t, DIM, Distance, IndexType>(dim, mat, leaf_max_size = 10)   :59
//   ->index (buildIndex / findNeighbors)                                                                     :53-57
//   query(query_point, num_closest, out_indices, out_distances_sq, nChecks = 10) const                       :80-85
// The adaptor keeps a reference to the caller's container (no copy on the host, :73); dimension mismatch with
// a fixed DIM throws std::runtime_error (:64).
#ifndef PCR_DROPIN_KDTREE_VOV_ADAPTOR_H
#define PCR_DROPIN_KDTREE_VOV_ADAPTOR_H

#include <cassert>
#include <stdexcept>

#include "nanoflann.hpp"

template <class VectorOfVectorsType, typename num_t = double, int DIM = -1, class Distance = nanoflann::metric_L2,
          typename IndexType = size_t>
struct KDTreeVectorOfVectorsAdaptor {
    typedef KDTreeVectorOfVectorsAdaptor<VectorOfVectorsType, num_t, DIM, Distance, IndexType> self_t;
    typedef typename Distance::template traits<num_t, self_t>::distance_t metric_t;
    typedef nanoflann::KDTreeSingleIndexAdaptor<metric_t, self_t, DIM, IndexType> index_t;

    index_t* index;
    const VectorOfVectorsType& m_data;

    KDTreeVectorOfVectorsAdaptor(const size_t /*dimensionality*/, const VectorOfVectorsType& mat, const int leaf_max_size = 10)
        : index(nullptr), m_data(mat)
    {
        assert(mat.size() != 0 && mat[0].size() != 0);
        const size_t dims = mat[0].size();
        if (DIM > 0 && static_cast<int>(dims) != DIM)
            throw std::runtime_error("Data set dimensionality does not match the 'DIM' template argument");
        index = new index_t(static_cast<int>(dims), *this, nanoflann::KDTreeSingleIndexAdaptorParams(leaf_max_size));
        index->buildIndex();
    }
    KDTreeVectorOfVectorsAdaptor(const self_t&) = delete;
    ~KDTreeVectorOfVectorsAdaptor() { delete index; }

    inline void query(const num_t* query_point, const size_t num_closest, IndexType* out_indices, num_t* out_distances_sq,
                      const int /*nChecks_IGNORED*/ = 10) const
    {
        nanoflann::KNNResultSet<num_t, IndexType> resultSet(num_closest);
        resultSet.init(out_indices, out_distances_sq);
        index->findNeighbors(resultSet, query_point, nanoflann::SearchParams());
    }

    const self_t& derived() const { return *this; }
    self_t& derived() { return *this; }
    inline size_t kdtree_get_point_count() const { return m_data.size(); }
    inline num_t kdtree_get_pt(const size_t idx, const size_t dim) const { return m_data[idx][dim]; }
    template <class BBOX> bool kdtree_get_bbox(BBOX&) const { return false; }
};

#endif  // PCR_DROPIN_KDTREE_VOV_ADAPTOR_H

// resultSet.hpp — drop-in replacement for Homework2/hw2/include/resultSet.hpp (same classes, members and
// behaviour), so that the hw2 / nano_vs_my drivers and the reference's own octree.hpp compile unchanged.
//   DistIndex            resultSet.hpp:7-18
//   KNNResultSet         resultSet.hpp:28-93   (capacity slots pre-filled with (1e10, 0); a candidate is rejected
//                                               only when dist > worst, so among equal distances the LAST
//                                               inserted one ends up in front of the run it joins)
//   RadiusNNResultSet    resultSet.hpp:96-142  (inclusive dist <= radius, insertion order)
// The GPU search (kdtree.hpp) fills a set through assign(): canonical order (distance, then index).
#ifndef PCR_DROPIN_RESULTSET_HPP
#define PCR_DROPIN_RESULTSET_HPP

#include <cstddef>
#include <iostream>
#include <vector>

class DistIndex
{
public:
    double distance;
    int index;
    DistIndex(double dist, int idx) : distance(dist), index(idx) {}
    bool operator<(const DistIndex& other) const { return distance < other.distance; }
};

inline std::ostream& operator<<(std::ostream& os, const DistIndex& di)
{
    return os << "Distance = " << di.distance << ", Index = " << di.index;
}

class KNNResultSet
{
    int capacity;
    double worstDist;

public:
    int count = 0;
    int comparisionCount = 0;   // (sic) spelling of the reference's public field
    std::vector<DistIndex> distIndexList;

    explicit KNNResultSet(int capa) : capacity(capa), worstDist(1e10), distIndexList(capa > 0 ? capa : 0, DistIndex(1e10, 0)) {}

    int size() { return capacity; }
    double getWorstDist() { return worstDist; }

    void list()
    {
        std::cout << "Distance-Index list: " << std::endl;
        for (const DistIndex& di : distIndexList) std::cout << di << std::endl;
    }

    void addPoint(double dist, int index)
    {
        ++comparisionCount;
        if (dist > worstDist || capacity <= 0) return;
        if (count < capacity) ++count;
        // open a slot at the tail of the filled part, then walk it towards the front past every entry that is
        // strictly farther than the newcomer
        int slot = count - 1;
        for (; slot > 0 && distIndexList[slot - 1].distance > dist; --slot) distIndexList[slot] = distIndexList[slot - 1];
        distIndexList[slot] = DistIndex(dist, index);
        worstDist = distIndexList[capacity - 1].distance;
    }

    // extension used by the GPU path: take a finished, canonically ordered result (n_valid <= capacity entries)
    void assign(const double* dist, const int* index, int n_valid, int compared)
    {
        for (int s = 0; s < capacity; ++s) distIndexList[s] = s < n_valid ? DistIndex(dist[s], index[s]) : DistIndex(1e10, 0);
        count = n_valid;
        comparisionCount += compared;
        worstDist = capacity > 0 ? distIndexList[capacity - 1].distance : 1e10;
    }
};

class RadiusNNResultSet
{
    double worstDist;
    double radius;

public:
    int count = 0;
    int comparisionCount = 0;
    std::vector<DistIndex> distIndexList;

    explicit RadiusNNResultSet(double r) : worstDist(r), radius(r) {}

    int size() { return count; }
    double getWorstDist() { return worstDist; }

    void list()
    {
        std::cout << "Distance-Index list: " << std::endl;
        for (const DistIndex& di : distIndexList) std::cout << di << std::endl;
    }

    void addPoint(double dist, int index)
    {
        ++comparisionCount;
        if (dist <= worstDist) {
            distIndexList.emplace_back(dist, index);
            ++count;
        }
    }

    // extension used by the GPU path: neighbours in ascending index order
    void assign(const double* dist, const int* index, size_t n, int compared)
    {
        for (size_t s = 0; s < n; ++s) distIndexList.emplace_back(dist[s], index[s]);
        count += (int)n;
        comparisionCount += compared;
    }
};

#endif  // PCR_DROPIN_RESULTSET_HPP

// resultSet.hpp — drop-in replacement for Homework2/hw2/include/resultSet.hpp: the same three types with the same
// public members and behaviour, so that the hw2 / nano_vs_my drivers AND the reference's own octree.hpp (which feeds
// these sets point by point through addPoint) compile unchanged against it.
//
//   DistIndex            resultSet.hpp:7-18    {distance, index}, ordered by distance
//   KNNResultSet         resultSet.hpp:28-93   k slots pre-filled with (1e10, 0); addPoint() rejects only dist > worst,
//                                              so among equal distances the LAST inserted one ends up in front of the
//                                              run it joins; worst = distance in the last slot
//   RadiusNNResultSet    resultSet.hpp:96-142  keeps dist <= radius (inclusive) in insertion order
//
// Extension: assign() installs a finished result computed on the GPU (kdtree.hpp), in canonical order.
#ifndef PCR_DROPIN_RESULTSET_HPP
#define PCR_DROPIN_RESULTSET_HPP

#include <cstddef>
#include <iostream>
#include <vector>

namespace pcr {
namespace dropin {
constexpr double kUnsetDistance = 1e10;   // the reference's placeholder distance for an empty k-NN slot

template <class List>
inline void print_dist_index_list(const List& entries)
{
    std::cout << "Distance-Index list: " << std::endl;
    for (const auto& e : entries) std::cout << e << std::endl;
}
}  // namespace dropin
}  // namespace pcr

class DistIndex
{
public:
    DistIndex(double dist, int idx) : distance(dist), index(idx) {}

    double distance;
    int index;

    bool operator<(const DistIndex& rhs) const { return distance < rhs.distance; }
};

inline std::ostream& operator<<(std::ostream& out, const DistIndex& entry)
{
    out << "Distance = " << entry.distance << ", Index = " << entry.index;
    return out;
}

class KNNResultSet
{
public:
    explicit KNNResultSet(int capa)
        : distIndexList(capa > 0 ? (size_t)capa : 0, DistIndex(pcr::dropin::kUnsetDistance, 0)),
          slots_(capa), farthest_(pcr::dropin::kUnsetDistance)
    {
    }

    // public state, spelled as in the reference (drivers read these directly)
    int count = 0;
    int comparisionCount = 0;
    std::vector<DistIndex> distIndexList;

    int size() { return slots_; }
    double getWorstDist() { return farthest_; }
    void list() { pcr::dropin::print_dist_index_list(distIndexList); }

    void addPoint(double dist, int index)
    {
        comparisionCount += 1;
        if (slots_ <= 0 || dist > farthest_) return;           // equal to the worst is still accepted
        if (count < slots_) count += 1;
        // the newcomer starts in the last filled slot and moves forward past every strictly farther entry
        int at = count - 1;
        while (at > 0 && distIndexList[at - 1].distance > dist) {
            distIndexList[at] = distIndexList[at - 1];
            at -= 1;
        }
        distIndexList[at] = DistIndex(dist, index);
        farthest_ = distIndexList[slots_ - 1].distance;
    }

    // extension (GPU path): n_valid <= capacity entries in canonical order (distance, then index); `compared` points
    // were examined to produce them
    void assign(const double* dist, const int* index, int n_valid, int compared)
    {
        for (int s = 0; s < slots_; ++s)
            distIndexList[s] = s < n_valid ? DistIndex(dist[s], index[s]) : DistIndex(pcr::dropin::kUnsetDistance, 0);
        count = n_valid;
        comparisionCount += compared;
        farthest_ = slots_ > 0 ? distIndexList[slots_ - 1].distance : pcr::dropin::kUnsetDistance;
    }

private:
    int slots_;
    double farthest_;
};

class RadiusNNResultSet
{
public:
    explicit RadiusNNResultSet(double r) : limit_(r) {}

    int count = 0;
    int comparisionCount = 0;
    std::vector<DistIndex> distIndexList;

    int size() { return count; }
    double getWorstDist() { return limit_; }               // the radius never shrinks
    void list() { pcr::dropin::print_dist_index_list(distIndexList); }

    void addPoint(double dist, int index)
    {
        comparisionCount += 1;
        if (dist <= limit_) {                               // inclusive, resultSet.hpp:133
            distIndexList.push_back(DistIndex(dist, index));
            count += 1;
        }
    }

    // extension (GPU path): n neighbours in ascending index order
    void assign(const double* dist, const int* index, size_t n, int compared)
    {
        distIndexList.reserve(distIndexList.size() + n);
        for (size_t s = 0; s < n; ++s) distIndexList.push_back(DistIndex(dist[s], index[s]));
        count += (int)n;
        comparisionCount += compared;
    }

private:
    double limit_;
};

#endif  // PCR_DROPIN_RESULTSET_HPP

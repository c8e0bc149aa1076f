// dropin_check.cpp — exercises the drop-in headers exactly the way the reference drivers use the originals
// (Homework2/hw2/include/test.hpp:162-225 testKDTree, benchmark.hpp:59-66; Homework3/nano_vs_my/main.cpp:62-96)
// and dumps the answers for the pytest side to compare with the oracle.
//   usage: dropin_check <in.bin> <out.bin>
//   in : int64 n, int64 m, int64 k, double r, then n*3 doubles (db), m*3 doubles (queries)
//   out: per query k (int32 idx, double dist) pairs from KDTreeKNNSearch, then the radius set
//        (int64 count, then count x (int32, double)), then k (uint64 idx, double d2) from the nanoflann-shaped
//        adaptor; afterwards the same three blocks for the self-query protocol over the first min(n, 64) points.
#include <cstdint>
#include <cstdio>
#include <vector>

#include <nanoflann.hpp>
#include "KDTreeVectorOfVectorsAdaptor.h"
#include "kdtree.hpp"
#include "resultSet.hpp"
#include "registration.hpp"

typedef std::vector<std::vector<double>> my_vector_of_vectors_t;

static void dump_query(FILE* out, Node*& root, my_vector_of_vectors_t& db, std::vector<double>& query, int k, double r,
                       KDTreeVectorOfVectorsAdaptor<my_vector_of_vectors_t, double>& mat_index)
{
    KNNResultSet result_set(k);
    KDTreeKNNSearch(root, db, result_set, query);
    for (int s = 0; s < k; s++) {
        int32_t i = result_set.distIndexList[s].index;
        double d = result_set.distIndexList[s].distance;
        fwrite(&i, 4, 1, out);
        fwrite(&d, 8, 1, out);
    }
    RadiusNNResultSet result_set_rnn(r);
    KDTreeRadiusNNSearch(root, db, result_set_rnn, query);
    int64_t cnt = result_set_rnn.size();
    fwrite(&cnt, 8, 1, out);
    for (auto& di : result_set_rnn.distIndexList) {
        int32_t i = di.index;
        fwrite(&i, 4, 1, out);
        fwrite(&di.distance, 8, 1, out);
    }
    std::vector<size_t> ret_indexes(k);
    std::vector<double> out_dists_sqr(k);
    nanoflann::KNNResultSet<double> resultSet(k);
    resultSet.init(&ret_indexes[0], &out_dists_sqr[0]);
    mat_index.index->findNeighbors(resultSet, &query[0], nanoflann::SearchParams(10));
    for (int s = 0; s < k; s++) {
        uint64_t i = s < (int)resultSet.size() ? ret_indexes[s] : (uint64_t)-1;
        double d = s < (int)resultSet.size() ? out_dists_sqr[s] : -1.0;
        fwrite(&i, 8, 1, out);
        fwrite(&d, 8, 1, out);
    }
}

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    FILE* in = fopen(argv[1], "rb");
    FILE* out = fopen(argv[2], "wb");
    if (!in || !out) return 3;
    int64_t n, m, k;
    double r;
    if (fread(&n, 8, 1, in) != 1 || fread(&m, 8, 1, in) != 1 || fread(&k, 8, 1, in) != 1 || fread(&r, 8, 1, in) != 1) return 4;
    my_vector_of_vectors_t db(n, std::vector<double>(3)), q(m, std::vector<double>(3));
    for (auto& p : db) if (fread(p.data(), 8, 3, in) != 3) return 4;
    for (auto& p : q) if (fread(p.data(), 8, 3, in) != 3) return 4;
    fclose(in);

    int leaf_size = 32;
    Node* root = KDTreeConstruction(db, leaf_size);
    if (root->isLeaf() && n > leaf_size) { fprintf(stderr, "Error!!!\n"); return 5; }
    typedef KDTreeVectorOfVectorsAdaptor<my_vector_of_vectors_t, double> my_kd_tree_t;
    my_kd_tree_t mat_index(3, db, 10);
    mat_index.index->buildIndex();

    for (auto& query : q) dump_query(out, root, db, query, (int)k, r, mat_index);
    // EXTENSIONS: the same questions in one batch must give the same sets as one call per query
    {
        std::vector<KNNResultSet> ks((size_t)m, KNNResultSet((int)k));
        std::vector<RadiusNNResultSet> rs((size_t)m, RadiusNNResultSet(r));
        KDTreeKNNSearchBatch(root, db, ks, q);
        KDTreeRadiusNNSearchBatch(root, db, rs, q);
        for (int64_t i = 0; i < m; i++) {
            KNNResultSet one((int)k);
            KDTreeKNNSearch(root, db, one, q[i]);
            RadiusNNResultSet oner(r);
            KDTreeRadiusNNSearch(root, db, oner, q[i]);
            if (one.count != ks[i].count || oner.size() != rs[i].size()) { fprintf(stderr, "batch/count mismatch at %lld\n", (long long)i); return 6; }
            for (int s = 0; s < (int)k; s++)
                if (one.distIndexList[s].index != ks[i].distIndexList[s].index || one.distIndexList[s].distance != ks[i].distIndexList[s].distance) {
                    fprintf(stderr, "batch/knn mismatch at %lld\n", (long long)i); return 6;
                }
            for (size_t s = 0; s < oner.distIndexList.size(); s++)
                if (oner.distIndexList[s].index != rs[i].distIndexList[s].index || oner.distIndexList[s].distance != rs[i].distIndexList[s].distance) {
                    fprintf(stderr, "batch/radius mismatch at %lld\n", (long long)i); return 6;
                }
        }
    }
    // benchmark.hpp:59-66 protocol: database points query their own cloud (served from one batched launch)
    int64_t self = n < 64 ? n : 64;
    for (int64_t i = 0; i < self; i++) {
        auto query = db[i];
        dump_query(out, root, db, query, (int)k, r, mat_index);
    }
    // the self-query rows live in HBM and reach the host through a bounded window: a caller that jumps around the cloud (every window
    // re-fetched) must see the rows a batch of the same points gets
    {
        const int64_t jumps = n < 48 ? n : 48;
        my_vector_of_vectors_t jq;
        for (int64_t t = 0; t < jumps; t++) jq.push_back(db[(size_t)((t * 7919 + 13) % n)]);
        std::vector<RadiusNNResultSet> rs((size_t)jumps, RadiusNNResultSet(r));
        KDTreeRadiusNNSearchBatch(root, db, rs, jq);
        for (int64_t t = 0; t < jumps; t++) {
            RadiusNNResultSet oner(r);
            KDTreeRadiusNNSearch(root, db, oner, jq[t]);
            if (oner.size() != rs[t].size()) { fprintf(stderr, "window/radius count mismatch at %lld\n", (long long)t); return 7; }
            for (size_t s2 = 0; s2 < oner.distIndexList.size(); s2++)
                if (oner.distIndexList[s2].index != rs[t].distIndexList[s2].index || oner.distIndexList[s2].distance != rs[t].distIndexList[s2].distance) {
                    fprintf(stderr, "window/radius mismatch at %lld\n", (long long)t); return 7;
                }
        }
    }
    int depth = TreeDepth(root);
    fwrite(&depth, 4, 1, out);
    KDTreeDestruction();

    // ICP front end: identity on identical clouds must stay the identity and keep every pair
    std::vector<float> cloud(4 * n);
    for (int64_t i = 0; i < n; i++) { for (int c = 0; c < 3; c++) cloud[4 * i + c] = (float)db[i][c]; cloud[4 * i + 3] = 1.0f; }
    pcr::IcpPoint2Point icp;
    icp.setICPparams(10, 4000, 1.0f, 3, 1e-8f);
    float R[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, t[3] = { 0, 0, 0 };
    int rc = icp.run(cloud.data(), n, cloud.data(), n, PCR_AOS4, R, t);
    int64_t pairs = (int64_t)icp.last_stats.last_pairs;
    fwrite(&rc, 4, 1, out);
    fwrite(&pairs, 8, 1, out);
    fwrite(R, 4, 9, out);
    fwrite(t, 4, 3, out);
    fclose(out);
    return 0;
}

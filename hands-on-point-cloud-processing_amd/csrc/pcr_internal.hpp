// pcr_internal.hpp — shared host-side declarations of libpcr_hip.so (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "pcr.h"

namespace pcr {

// Every cloud is allocated with its length rounded up to PAD points; x of the padding is +inf so that a
// padded target can never win a nearest-neighbour comparison (d2 = +inf is not < FLT_MAX).
constexpr size_t PAD = 1024;
constexpr int PCR_NSTATS = 16;      // diagnostics words of a 1-NN launch (tune grid_stats; pcr_nn1_stats)

inline size_t padded(size_t n) { return ((n + PAD - 1) / PAD) * PAD + PAD; }   // always >= 1 full pad block

struct ProfEntry {
    uint64_t launches = 0;
    double total_ms = 0.0;
    std::vector<float> each_ms;           // the individual durations since the last reset, in launch order (at most 4096 kept)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

// device-resident state of the pipelined ICP loop (registration.cpp:910-1006); mirrored in pinned host memory
struct IcpState {
    float Rd[9];
    float td[3];
    float T_total[16];
    float last_loss;
    float loss;
    float eps;
    int converged;
    int empty;
    int stop;                    // set by the update kernel: later kernels of the stream become no-ops
    int stop_after_transform;    // max_iter reached: stop once the last transform has been applied
    int overflow;                // a kept source point lay beyond the fixed-point grid of the Kabsch sums (kabsch_plan): PCR_ERR_STATE
    int pad_;
    unsigned long long unchanged;
    unsigned long long iters_run;
    unsigned long long max_iter;
    unsigned long long last_pairs;
};

// fixed-point grid of one exact Kabsch accumulation (kabsch.hip / numerics.hpp): 2^e bounds every coordinate of a kept pair
struct KabschPlan {
    int e = 0;
    float lim = 0.f;      // 2^e as a float (inf when e >= 128)
    double sc = 0.0;      // 2^(80 - e): coordinates -> two 40-bit limbs
    double sp = 0.0;      // 2^(120 - 2e): products -> three 40-bit limbs
};

struct Comm {
    int nranks = 1;
    int rank = 0;
    void* rccl = nullptr;          // ncclComm_t
    pcr_allreduce_fn cb = nullptr;
    void* cb_user = nullptr;
};

}  // namespace pcr

namespace pcr { struct Grid; void grid_free(Grid*); struct BtIndex; void bt_free(BtIndex*); }

struct pcr_cloud {
    pcr::Grid* grid = nullptr;   // exact-NN index over this cloud as a target; built lazily, dropped on modification
    pcr::BtIndex* bt = nullptr;  // the matrix-core brute-force filter's operands over this cloud as a target (nn1_brute.hip); same lifetime
    pcr::Grid* knn_grid = nullptr;   // the same index with the wider cell of the last k-NN batch (knn_grid.hip), same lifetime
    double knn_grid_factor = 0.0;    // its cell edge / grid's cell edge
    pcr::Grid* rad_grid = nullptr;   // the index with cell edge 1.01 r of the last radius search (radius_grid.hip), same lifetime
    double rad_grid_r = 0.0;
    float absmax = -1.f;             // largest finite |coordinate| (cloud_absmax), < 0: not computed; same lifetime as the grids
    mutable uint32_t brute_searches = 0;   // exhaustive searches that found no index on this target (the second one builds it); same lifetime
    size_t n = 0;
    size_t cap = 0;     // padded length of each of x, y, z
    float* base = nullptr;   // device; x = base, y = base + cap, z = base + 2*cap
    // a SHARD of a larger cloud (pcr_cloud_shard_spatial): the index every point has in the whole cloud, ascending (device, n entries).
    // "The last kept pair" of an iteration (registration.cpp:939) is then the kept pair with the largest GLOBAL index over all ranks.
    uint32_t* gidx = nullptr;
    float* x() const { return base; }
    float* y() const { return base + cap; }
    float* z() const { return base + 2 * cap; }
};

struct pcr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    hipDeviceProp_t prop;
    // workspace
    unsigned long long* keys = nullptr;   // (d2_bits << 32 | idx) per query of the last nn1 pass
    size_t keys_cap = 0;
    size_t keys_n = 0;
    // Warm start of the searches inside ICP loops: keys[] still holds correspondences of an earlier search.  Any index < n_tgt
    // is a genuine candidate once re-evaluated exactly against the query, so a stale entry can only make the seed worse, never
    // the result wrong; the bookkeeping below only decides whether the seeds are likely to be GOOD: same source cloud (the
    // previous iteration), or another same-sized source against the same target (the previous ICP run on this pair).
    bool keys_warm = false;               // keys[] was written by a search of an ICP loop
    const pcr_cloud* keys_src = nullptr;  // its source (identity only; reset when the cloud is destroyed)
    const pcr_cloud* keys_tgt = nullptr;  // its target (identity only; reset when the cloud is destroyed)
    size_t keys_warm_n = 0;               // number of queries of that search
    // the last move of an ICP loop already turned keys[] into the seeds of the next exhaustive search of (keys_seed_src, keys_seed_tgt):
    // launch_nn1_brute then skips its own seed kernel (consumed by the next search, whatever it is)
    bool keys_seeded = false;
    const pcr_cloud* keys_seed_src = nullptr;
    const pcr_cloud* keys_seed_tgt = nullptr;
    uint32_t* qperm = nullptr;            // queries grouped by target-grid cell (grid NN)
    size_t qperm_cap = 0;
    size_t qperm_n = 0;
    const pcr_cloud* qperm_src = nullptr;
    // grid ICP: the working cloud is cell-sorted once (work_orig[t] = original index of its point t), and every search leaves
    // the record position of each winner behind (wpos) for the next warm start and for the Kabsch gather
    uint32_t* wpos = nullptr;
    size_t wpos_cap = 0;
    size_t wpos_n = 0;
    bool wpos_valid = false;
    uint32_t* work_orig = nullptr;
    size_t work_orig_cap = 0;
    size_t work_orig_n = 0;
    const pcr_cloud* work_orig_src = nullptr;
    // two spare cloud buffers (base, cap): the working copies an ICP call clones and gives back (cloud_release) are re-used by the next
    // clone of the same size instead of two hipMalloc + two hipFree (each free a device synchronisation) per call
    float* spare_base[2] = { nullptr, nullptr };
    size_t spare_cap[2] = { 0, 0 };
    uint32_t work_cells = 0;              // distinct target-grid cells the sorted working cloud occupies (0: unknown); read through work_cells_now()
    // two counters of the index builds come back through PINNED words behind an event each (a hipMemcpyAsync into pageable memory is ordered only
    // by what the runtime happens to do): [0] work_cells, [1] the occupied cells of the grid built last (Grid::occupied_tag says whose)
    uint32_t* pin_words = nullptr;
    hipEvent_t pin_ev[2] = { nullptr, nullptr };
    bool pin_pending[2] = { false, false };
    uint64_t pin_gen = 0;
    double* partials = nullptr;           // block rows of the Kabsch pass (8192 x 58 doubles)
    size_t partials_cap = 0;
    double* dev_out = nullptr;            // 128 doubles: reduced sums / limbs + bookkeeping
    double* host_out = nullptr;           // pinned mirror
    void* plane_ws = nullptr;             // plane.hip: [ticket | pad to 256 B][partial counts per workgroup]
    size_t plane_ws_bytes = 0, plane_ws_tick = 0;
    void* scratch = nullptr;              // generic device scratch
    size_t scratch_cap = 0;
    void* host_stage = nullptr;           // pinned staging for uploads / downloads
    size_t host_stage_cap = 0;
    void* coop_host = nullptr;            // small coherent host buffer of the one-wave-per-query k-NN (knn_grid.hip): completion word, results
    uint32_t* coop_ticket = nullptr;      // its wave ticket (device memory)
    uint32_t coop_seq = 0;                // value the completion word takes at the end of the next call
    size_t coop_m = 0;                    // queries of the last call (the ticket counts modulo m)
    void* aux = nullptr;                  // second device scratch (partial results of sliced searches), grows on demand
    size_t aux_cap = 0;
    pcr::IcpState* icp_state_dev = nullptr;    // pipelined ICP: device state, pinned snapshots, snapshot events
    pcr::IcpState* icp_state_host = nullptr;
    hipEvent_t icp_events[4] = { nullptr, nullptr, nullptr, nullptr };
    unsigned long long* grid_stats_dev = nullptr;   // diagnostics of the grid search (tune grid_stats)
    const int* stop_flag_dev = nullptr;        // when set, the correspondence kernels exit early once *flag != 0
    uint64_t loop_iters_hint = 0;              // set by an iterated loop (ICP) for its duration: how many searches of one target may follow (nn1_auto_grid)
    uint32_t* far_list = nullptr;              // far queries handed from the grid walk to the exhaustive kernel: [cap] indices + [1] count
    size_t far_cap = 0;
    pcr::Comm comm;
    std::map<std::string, pcr::ProfEntry> prof;
    int prof_level = 0;                   // 0 off (default: an event pair costs ~6 us of stream time on each side of the kernel),
                                          // 1 correspondence kernels only, 2 every kernel
    std::map<std::string, int64_t> tune;
    // what mfma_verdict (nn1_brute.hip) measured on this device before a matrix-core 1-NN kernel was first chosen: -1 not run yet,
    // 1 = within half of every bound the kernel's analysis assumes, 0 = not (the form is never used on this context)
    int mfma_ok16 = -1, mfma_okbf = -1;
    double mfma_worst16[4] = { 0, 0, 0, 0 }, mfma_worstbf[4] = { 0, 0, 0, 0 };
    double mfma_check_ms = 0.0;           // host wall time the checks took (once per context and form)
    const char* last_nn1_kernel = "";     // family of the last 1-NN launch: htrack / btrack / etrack / ftrack / track / grid
};

namespace pcr {

int fail(pcr_ctx* ctx, int code, const char* what, hipError_t e = hipSuccess);

#define PCR_HIP(ctx, call)                                                   \
    do {                                                                     \
        hipError_t e__ = (call);                                             \
        if (e__ != hipSuccess) return pcr::fail((ctx), PCR_ERR_HIP, #call, e__); \
    } while (0)

int ensure_keys(pcr_ctx* ctx, size_t n);
int ensure_scratch(pcr_ctx* ctx, size_t bytes);
int ensure_stage(pcr_ctx* ctx, size_t bytes);
int ensure_aux(pcr_ctx* ctx, size_t bytes);
int64_t tune_get(const pcr_ctx* ctx, const char* key, int64_t dflt);

// profiling: record a (start, stop) event pair around a launch on ctx->stream
struct ProfScope {
    pcr_ctx* ctx;
    const char* name;
    hipEvent_t a = nullptr, b = nullptr;
    ProfScope(pcr_ctx* c, const char* n, int level = 2);
    ~ProfScope();
};
void prof_flush(pcr_ctx* ctx);

// ---- kernel launchers (defined in the .hip files) --------------------------------------------------
int bt_mfma_selftest(pcr_ctx* ctx, int trials, double worst[4]);
int ht_mfma_selftest(pcr_ctx* ctx, int trials, double worst[4]);
int st_sphere_selftest(pcr_ctx* ctx, int trials, unsigned long long res[4]);   // nn1_brute.hip: the level-1 (chunk sphere) statement of STRACK2
int st_sign_selftest(pcr_ctx* ctx, int trials, unsigned long long res[4]);   // nn1_brute.hip: the sign form's decision, checked on the device
bool mfma_verdict(pcr_ctx* ctx, bool f16);     // cached per context; runs the short self-test on first use
int launch_nn1_brute(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* src, bool in_loop);
int launch_nn1_brute_list(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* src, const uint32_t* qlist, const uint32_t* qcount, uint32_t qcap);
// cap2: the caller only uses neighbours with d2 < cap2 (ICP's max_corres_dist gate) — the walk may stop once no such target can exist
int launch_nn1_grid(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* src, bool reuse_perm, float cap2);
// grid radius search with the hw2 contract (radius_grid.hip); *used = false -> the caller runs the exhaustive kernels
struct RadiusRowsDev { uint32_t* rows_dev = nullptr; int32_t* idx_dev = nullptr; double* dist_dev = nullptr; uint64_t total = 0; };
int radius_grid(pcr_ctx* ctx, const pcr_cloud* db, const pcr_cloud* q, double r, double r2max, int64_t* row_ptr_host, int32_t* idx_host, double* dist_host,
                bool* used, RadiusRowsDev* keep = nullptr);
// exact grid k-NN between resident clouds (knn_grid.hip); host outputs idx/val [m x k], found [m] (optional)
int cloud_knn_host(pcr_ctx* ctx, const pcr_cloud* db, const pcr_cloud* q, int k, double cap_s, bool squared, double empty_val, int32_t empty_idx,
                   int32_t* idx, double* val, uint32_t* found);
// the same for a small batch of host queries (f32 rows): one launch, zero-copy in and out
constexpr size_t KNN_SMALL_MAX = 4096;
int cloud_knn_small(pcr_ctx* ctx, const pcr_cloud* db, const float* q_rows, size_t m, int k, double cap_s, bool squared, double empty_val,
                    int32_t empty_idx, int32_t* idx, double* val);
// device-wide exclusive scan of u32 (grid.hip): totals needs ceil(n / SCAN_TILE) + 1 words
constexpr int SCAN_TILE = 2048;
int exclusive_scan_u32(pcr_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n, uint32_t* totals, uint32_t* grand);
// dispatcher: tune "nn_method" 0 = auto (grid for targets >= 2048 points), 1 = brute force, 2 = grid
// (auto: api.cpp nn1_auto_grid — inside an iterated loop the grid from 128 points on; one-shot searches of a target without an index
// stay exhaustive while queries x targets is small)
bool nn1_auto_grid(const pcr_ctx* ctx, const pcr_cloud* tgt, bool in_loop, size_t ns);
// tells the dispatcher for the duration of a loop how many searches of one target may follow
struct LoopHint {
    pcr_ctx* ctx;
    LoopHint(pcr_ctx* c, uint64_t iters) : ctx(c) { ctx->loop_iters_hint = iters; }
    ~LoopHint() { ctx->loop_iters_hint = 0; }
    LoopHint(const LoopHint&) = delete;
    LoopHint& operator=(const LoopHint&) = delete;
};
int launch_nn1(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* src, bool reuse_perm, float cap2 = __builtin_inff());
void cloud_modified(pcr_cloud* c);
// gives a cloud back WITHOUT synchronising the stream: its buffer goes to the context's spare slots (or is freed when both are taken).
// Safe for work enqueued on ctx->stream that still reads the cloud: the buffer stays allocated, and whoever gets it next writes it on
// the same stream.  (pcr_cloud_destroy = the same behind a stream synchronisation: the public contract.)
void cloud_release(pcr_ctx* ctx, pcr_cloud* c);
int grid_sort_working_cloud(pcr_ctx* ctx, const pcr_cloud* tgt, pcr_cloud** work, bool in_place = false);
int bt_sort_working_cloud(pcr_ctx* ctx, const pcr_cloud* tgt, pcr_cloud** work, bool in_place = false);
int nn1_unpack(pcr_ctx* ctx, size_t n, uint32_t* idx_dev, float* d2_dev);
int cloud_shard_spatial(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* full, int nranks, int rank, int chunks_per_rank, pcr_cloud** out,
                        int (*alloc)(pcr_ctx*, size_t, pcr_cloud**));
int launch_transform(pcr_ctx* ctx, pcr_cloud* c, const float R[9], const float t[3]);
// dst (a fresh allocation of src's size: cloud_alloc) = R src + t, padding included — the clone and the transform of a loop's working copy in one launch
int launch_transform_into(pcr_ctx* ctx, const pcr_cloud* src, pcr_cloud* dst, const float R[9], const float t[3]);
int cloud_alloc(pcr_ctx* ctx, size_t n, pcr_cloud** out);     // api.cpp: an uninitialised cloud of n points (from the parked buffers when one fits)
int cloud_absmax(pcr_ctx* ctx, const pcr_cloud* c, float* out);       // largest finite |coordinate|, cached on the cloud (grid.hip)
int kabsch_grid_exponent(float target_absmax, float max_corr);
int kabsch_plan(pcr_ctx* ctx, const pcr_cloud* tgt, float max_corr, KabschPlan* plan);
// dev_out[0..18] = the 16 moments, original index of the last kept pair (or -1), its d2, overflow flag
int launch_kabsch_sums(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* src, float max_corr, const KabschPlan& plan);
int launch_kabsch_partial(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* src, float max_corr, const KabschPlan& plan, uint32_t* n_blocks);
int launch_icp_update(pcr_ctx* ctx, uint32_t n_blocks, IcpState* st_dev, const KabschPlan& plan);
// dev_out[0..54] limbs, [55] overflow flag, [56 + 2r], [57 + 2r] (order key of the last kept pair — 0 = none kept —, its d2) of rank r:
// ICP_NRED(nranks) doubles to all-reduce; gidx: the shard's global indices (key = global index + 1), or nullptr (key = rank + 1: contiguous shards in rank order)
int launch_icp_reduce_slots(pcr_ctx* ctx, uint32_t n_blocks, int nranks, int rank, bool have_points, const uint32_t* gidx = nullptr);
int launch_icp_update_from_sums(pcr_ctx* ctx, int nranks, IcpState* st_dev, const KabschPlan& plan);
inline int icp_nred(int nranks) { return 56 + 2 * nranks; }
// seed_tgt != nullptr: the move also writes the seeds of the next exhaustive search into keys[] (kabsch.hip seed_next_search)
int launch_transform_state(pcr_ctx* ctx, pcr_cloud* c, IcpState* st_dev, const pcr_cloud* seed_tgt = nullptr);
// small clouds, one rank: icp_update + transform_state in one launch; the state alternates between the two buffers st_in / st_out
int launch_icp_update_move(pcr_ctx* ctx, uint32_t n_blocks, const IcpState* st_in, IcpState* st_out, const KabschPlan& plan, pcr_cloud* c,
                           const pcr_cloud* seed_tgt = nullptr);
int comm_allreduce_f64_device(pcr_ctx* ctx, double* dev_buf, int n);   // RCCL on the ctx stream, no host round trip
int launch_plane_count(pcr_ctx* ctx, const pcr_cloud* pts, const double* planes4_host, size_t n_planes,
                       double thr, unsigned long long* counts_out);
int launch_plane_mask(pcr_ctx* ctx, const pcr_cloud* pts, const double plane4[4], double thr,
                      uint8_t* mask_dev, unsigned long long* count_dev);
int launch_knn_f64(pcr_ctx* ctx, const double* db_soa, size_t n, size_t n_cap, const double* q_soa, size_t m,
                   int k, int32_t* idx_dev, double* dist_dev, bool squared);
int launch_radius_count(pcr_ctx* ctx, const double* db_soa, size_t n, size_t n_cap, const double* q_soa,
                        size_t m, double r, unsigned long long* counts_dev, uint32_t slices);
int launch_radius_fill(pcr_ctx* ctx, const double* db_soa, size_t n, size_t n_cap, const double* q_soa,
                       size_t m, double r, const long long* row_ptr_dev, int32_t* idx_dev, double* dist_dev, uint32_t slices);

// ---- host numerics ------------------------------------------------------------------------------------
void svd3(const double A[9], double U[9], double S[3], double V[9]);
int kabsch_solve(const double sums[16], float R[9], float t[3]);
void mat4_mul_f32(const float A[16], const float B[16], float out[16]);

// ---- collectives ----------------------------------------------------------------------------------------
int comm_allreduce_f64(pcr_ctx* ctx, double* host_buf, double* dev_buf, int n);
// the slot of the globally last kept pair in a summed reduce row (the largest order key), or -1: the same rule on host and device
__host__ __device__ inline int icp_last_slot(const double* buf, int nranks)
{
    int best = -1;
    double key = 0.5;
    for (int r = 0; r < nranks; r++)
        if (buf[56 + 2 * r] > key) { key = buf[56 + 2 * r]; best = r; }
    return best;
}

}  // namespace pcr

// api.cpp — context, cloud management and the thin extern "C" entry points of libpcr_hip.so.
#include "pcr_internal.hpp"
#include "numerics.hpp"

#include <cmath>
#include <limits>
#include <new>

namespace pcr {

int fail(pcr_ctx* ctx, int code, const char* what, hipError_t e)
{
    if (ctx) {
        ctx->err = what ? what : "";
        if (e != hipSuccess) {
            ctx->err += ": ";
            ctx->err += hipGetErrorString(e);
        }
    }
    return code;
}

int ensure_keys(pcr_ctx* ctx, size_t n)
{
    if (n <= ctx->keys_cap) return PCR_OK;
    if (ctx->keys) PCR_HIP(ctx, hipFree(ctx->keys));
    ctx->keys = nullptr;
    ctx->keys_cap = 0;
    size_t cap = padded(n);
    PCR_HIP(ctx, hipMalloc((void**)&ctx->keys, cap * sizeof(unsigned long long)));
    ctx->keys_cap = cap;
    return PCR_OK;
}

int ensure_scratch(pcr_ctx* ctx, size_t bytes)
{
    if (bytes <= ctx->scratch_cap) return PCR_OK;
    if (ctx->scratch) PCR_HIP(ctx, hipFree(ctx->scratch));
    ctx->scratch = nullptr;
    ctx->scratch_cap = 0;
    size_t cap = (bytes + (1 << 20)) & ~((size_t)(1 << 20) - 1);
    PCR_HIP(ctx, hipMalloc(&ctx->scratch, cap));
    ctx->scratch_cap = cap;
    return PCR_OK;
}

int ensure_aux(pcr_ctx* ctx, size_t bytes)
{
    if (bytes <= ctx->aux_cap) return PCR_OK;
    if (ctx->aux) { PCR_HIP(ctx, hipStreamSynchronize(ctx->stream)); PCR_HIP(ctx, hipFree(ctx->aux)); }
    ctx->aux = nullptr;
    ctx->aux_cap = 0;
    const size_t cap = (bytes + (1 << 20)) & ~((size_t)(1 << 20) - 1);
    PCR_HIP(ctx, hipMalloc(&ctx->aux, cap));
    ctx->aux_cap = cap;
    return PCR_OK;
}

int ensure_stage(pcr_ctx* ctx, size_t bytes)
{
    if (bytes <= ctx->host_stage_cap) return PCR_OK;
    if (ctx->host_stage) PCR_HIP(ctx, hipHostFree(ctx->host_stage));
    ctx->host_stage = nullptr;
    ctx->host_stage_cap = 0;
    size_t cap = (bytes + (1 << 20)) & ~((size_t)(1 << 20) - 1);
    PCR_HIP(ctx, hipHostMalloc(&ctx->host_stage, cap, hipHostMallocDefault));
    ctx->host_stage_cap = cap;
    return PCR_OK;
}

int64_t tune_get(const pcr_ctx* ctx, const char* key, int64_t dflt)
{
    auto it = ctx->tune.find(key);
    return (it == ctx->tune.end() || it->second == 0) ? dflt : it->second;
}

ProfScope::ProfScope(pcr_ctx* c, const char* n, int level) : ctx(c), name(n)
{
    if (ctx->prof_level < level) return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
    hipEventRecord(a, ctx->stream);
}

ProfScope::~ProfScope()
{
    if (!a || !b) return;
    hipEventRecord(b, ctx->stream);
    ctx->prof[name].pending.emplace_back(a, b);
}

void prof_flush(pcr_ctx* ctx)
{
    for (auto& kv : ctx->prof) {
        for (auto& ev : kv.second.pending) {
            float ms = 0.f;
            if (hipEventSynchronize(ev.second) == hipSuccess && hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
                kv.second.launches++;
                kv.second.total_ms += ms;
                if (kv.second.each_ms.size() < 4096) kv.second.each_ms.push_back(ms);
            }
            hipEventDestroy(ev.first);
            hipEventDestroy(ev.second);
        }
        kv.second.pending.clear();
    }
}

void cloud_modified(pcr_cloud* c)
{
    if (c && c->grid) { grid_free(c->grid); c->grid = nullptr; }
    if (c && c->bt) { bt_free(c->bt); c->bt = nullptr; }
    if (c && c->knn_grid) { grid_free(c->knn_grid); c->knn_grid = nullptr; c->knn_grid_factor = 0.0; }
    if (c && c->rad_grid) { grid_free(c->rad_grid); c->rad_grid = nullptr; c->rad_grid_r = 0.0; }
    if (c) { c->absmax = -1.f; c->brute_searches = 0; }
}

// nn_method: 0 auto, 1 brute force, 2 uniform grid.  Auto:
// * inside an iterated loop (ICP: the index is built once, every later search starts from the previous correspondence) the exact
//   grid for targets >= 2048 points, and already from 128 points on when at least five searches may follow or the index exists —
//   measured, whole ICP calls on FRESH targets of 256 / 1 000 / 2 000 points: the grid is ahead from the 4th-5th iteration (5
//   iterations 284 / 306 / 350 us against 328 / 342 / 355; 20 iterations 577 / 637 / 666 against 1 091 / 1 167 / 1 220), per iteration
//   20-24 us against 51-58 us; at 64 points the two meet only at ~20 iterations;
// * a one-shot search takes the grid when the target's index exists already; on a target without one the exhaustive kernels are
//   ahead while queries x targets stays below ~2e9 — first search of a fresh n x n pair, wall: n = 2 048 76 against 222 us, 4 096 76 /
//   236, 8 192 209 / 264, 16 384 248 / 281, 32 768 326 / 383, then 65 536 576 / 462, 120 000 1 030 / 706 (profiles/r02_kabsch_probes.txt).
bool nn1_auto_grid(const pcr_ctx* ctx, const pcr_cloud* tgt, bool in_loop, size_t ns)
{
    const int64_t method = tune_get(ctx, "nn_method", 0);
    if (method == 2) return true;
    if (method == 1) return false;
    if (in_loop) return tgt->n >= 2048 || (tgt->n >= 128 && (tgt->grid != nullptr || ctx->loop_iters_hint >= 5));
    if (tgt->n < 2048) return false;
    if (tgt->grid != nullptr) return true;
    return tgt->n >= 65536 || (double)ns * (double)tgt->n > 2.0e9;
}

int launch_nn1(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* src, bool reuse_perm, float cap2)
{
    return nn1_auto_grid(ctx, tgt, reuse_perm, src->n) ? launch_nn1_grid(ctx, tgt, src, reuse_perm, cap2) : launch_nn1_brute(ctx, tgt, src, reuse_perm);
}

static void spare_trim(pcr_ctx* ctx)
{
    for (int k = 0; k < 2; k++)
        if (ctx->spare_base[k]) { hipFree(ctx->spare_base[k]); ctx->spare_base[k] = nullptr; ctx->spare_cap[k] = 0; }
}

int cloud_alloc(pcr_ctx* ctx, size_t n, pcr_cloud** out)
{
    pcr_cloud* c = new (std::nothrow) pcr_cloud();
    if (!c) return fail(ctx, PCR_ERR_NOMEM, "cloud alloc");
    c->n = n;
    c->cap = padded(n);
    for (int k = 0; k < 2; k++)
        if (ctx->spare_base[k] && ctx->spare_cap[k] == c->cap) {          // a buffer an ICP loop's working copy gave back (cloud_release)
            c->base = ctx->spare_base[k];
            ctx->spare_base[k] = nullptr; ctx->spare_cap[k] = 0;
            *out = c;
            return PCR_OK;
        }
    // a miss: what the slots hold belongs to a working size that is no longer the caller's — freed here, so that parked buffers never
    // outlive the loops that use them by more than one allocation (hipFree waits for the device: the hipMalloc below does too)
    spare_trim(ctx);
    hipError_t e = hipMalloc((void**)&c->base, 3 * c->cap * sizeof(float));
    if (e != hipSuccess) { delete c; return fail(ctx, PCR_ERR_HIP, "hipMalloc(cloud)", e); }
    *out = c;
    return PCR_OK;
}

static void cloud_forget(pcr_ctx* ctx, pcr_cloud* c)
{
    if (ctx && ctx->qperm_src == c) ctx->qperm_src = nullptr;
    if (ctx && ctx->keys_src == c) ctx->keys_src = nullptr;
    if (ctx && ctx->keys_tgt == c) { ctx->keys_tgt = nullptr; ctx->keys_warm = false; ctx->wpos_valid = false; }
    if (ctx && ctx->work_orig_src == c) ctx->work_orig_src = nullptr;
    if (ctx && (ctx->keys_seed_src == c || ctx->keys_seed_tgt == c)) { ctx->keys_seeded = false; ctx->keys_seed_src = ctx->keys_seed_tgt = nullptr; }
    cloud_modified(c);
}

// INTERNAL working copies only (the clone an ICP loop moves, the sorted copy that replaces it): the buffer is parked in one of two
// per-context slots for the next clone of the same size — no synchronisation, work in flight may still read it; whoever gets it next
// writes it on the same stream.  Clouds the CALLER owns never come here: pcr_cloud_destroy frees (below).
void cloud_release(pcr_ctx* ctx, pcr_cloud* c)
{
    if (!c) return;
    cloud_forget(ctx, c);
    if (c->gidx) { hipFree(c->gidx); c->gidx = nullptr; }
    if (c->base) {
        // (buffers beyond 2 GB are not kept: a spare slot is a convenience, not a cache of the caller's memory)
        int slot = -1;
        if (ctx && 3 * c->cap * sizeof(float) <= ((size_t)2 << 30))
            for (int k = 0; k < 2 && slot < 0; k++) if (!ctx->spare_base[k]) slot = k;
        if (slot >= 0) { ctx->spare_base[slot] = c->base; ctx->spare_cap[slot] = c->cap; }
        else hipFree(c->base);                                             // (synchronises the device: nothing in flight can still read it)
    }
    delete c;
}

}  // namespace pcr

using namespace pcr;

extern "C" {

const char* pcr_version(void) { return "pcr-mi355x 0.1 (gfx950)"; }

int pcr_device_count(int* count)
{
    if (!count) return PCR_ERR_ARG;
    *count = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return PCR_ERR_HIP;
    *count = n;
    return PCR_OK;
}

int pcr_ctx_create(int device, pcr_ctx** out)
{
    if (!out) return PCR_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return PCR_ERR_HIP;
    pcr_ctx* ctx = new (std::nothrow) pcr_ctx();
    if (!ctx) return PCR_ERR_NOMEM;
    ctx->device = device;
#define CK(call)                                          \
    do {                                                  \
        hipError_t e__ = (call);                          \
        if (e__ != hipSuccess) {                          \
            fprintf(stderr, "pcr_ctx_create: %s: %s\n", #call, hipGetErrorString(e__)); \
            delete ctx;                                   \
            return PCR_ERR_HIP;                           \
        }                                                 \
    } while (0)
    CK(hipSetDevice(device));
    CK(hipGetDeviceProperties(&ctx->prop, device));
    CK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    CK(hipMalloc((void**)&ctx->partials, 8192 * 58 * sizeof(double)));   // KB_MAX_BLOCKS x KB_ROW (kabsch.hip)
    ctx->partials_cap = 8192 * 58;
    CK(hipMalloc((void**)&ctx->dev_out, 128 * sizeof(double)));           // >= icp_nred(PCR_MAX_RANKS) = 104
    CK(hipHostMalloc((void**)&ctx->host_out, 128 * sizeof(double), hipHostMallocDefault));
#undef CK
    *out = ctx;
    return PCR_OK;
}

int pcr_ctx_destroy(pcr_ctx* ctx)
{
    if (!ctx) return PCR_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    prof_flush(ctx);
    pcr_comm_destroy(ctx);
    spare_trim(ctx);
    if (ctx->keys) hipFree(ctx->keys);
    if (ctx->far_list) hipFree(ctx->far_list);
    if (ctx->icp_state_dev) hipFree(ctx->icp_state_dev);
    if (ctx->icp_state_host) hipHostFree(ctx->icp_state_host);
    for (hipEvent_t ev : ctx->icp_events) if (ev) hipEventDestroy(ev);
    if (ctx->qperm) hipFree(ctx->qperm);
    if (ctx->wpos) hipFree(ctx->wpos);
    if (ctx->work_orig) hipFree(ctx->work_orig);
    if (ctx->grid_stats_dev) hipFree(ctx->grid_stats_dev);
    if (ctx->pin_words) { hipHostFree(ctx->pin_words); for (int k = 0; k < 2; k++) if (ctx->pin_ev[k]) hipEventDestroy(ctx->pin_ev[k]); }
    if (ctx->partials) hipFree(ctx->partials);
    if (ctx->dev_out) hipFree(ctx->dev_out);
    if (ctx->host_out) hipHostFree(ctx->host_out);
    if (ctx->scratch) hipFree(ctx->scratch);
    if (ctx->aux) hipFree(ctx->aux);
    if (ctx->host_stage) hipHostFree(ctx->host_stage);
    if (ctx->coop_host) hipHostFree(ctx->coop_host);
    if (ctx->coop_ticket) hipFree(ctx->coop_ticket);
    if (ctx->plane_ws) hipFree(ctx->plane_ws);
    hipStreamDestroy(ctx->stream);
    delete ctx;
    return PCR_OK;
}

int pcr_ctx_sync(pcr_ctx* ctx)
{
    if (!ctx) return PCR_ERR_ARG;
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PCR_OK;
}

const char* pcr_ctx_last_error(const pcr_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int pcr_ctx_device_info(const pcr_ctx* ctx, char* arch, size_t arch_cap, int* n_cu, uint64_t* hbm_bytes)
{
    if (!ctx) return PCR_ERR_ARG;
    if (arch && arch_cap) {
        strncpy(arch, ctx->prop.gcnArchName, arch_cap - 1);
        arch[arch_cap - 1] = 0;
    }
    if (n_cu) *n_cu = ctx->prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = ctx->prop.totalGlobalMem;
    return PCR_OK;
}

// ---------------------------------------------------------------------------------------------- clouds
int pcr_cloud_create(pcr_ctx* ctx, const float* host_xyz, size_t n, int layout, pcr_cloud** out)
{
    if (!ctx || !out || (n && !host_xyz) || !(layout == PCR_SOA || layout == PCR_AOS3 || layout == PCR_AOS4 || layout == PCR_AOS6))
        return fail(ctx, PCR_ERR_ARG, "pcr_cloud_create");
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    pcr_cloud* c = nullptr;
    int rc = cloud_alloc(ctx, n, &c);
    if (rc) return rc;
    // stage as padded SoA: x padding = +inf, y/z padding = 0
    rc = ensure_stage(ctx, 3 * c->cap * sizeof(float));
    if (rc) { pcr_cloud_destroy(ctx, c); return rc; }
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the staging buffer may still be in flight
    float* st = (float*)ctx->host_stage;
    float* sx = st, *sy = st + c->cap, *sz = st + 2 * c->cap;
    if (layout == PCR_SOA) {
        memcpy(sx, host_xyz, n * sizeof(float));
        memcpy(sy, host_xyz + n, n * sizeof(float));
        memcpy(sz, host_xyz + 2 * n, n * sizeof(float));
    } else {
        const size_t stride = layout == PCR_AOS3 ? 3 : (layout == PCR_AOS4 ? 4 : 6);
        for (size_t i = 0; i < n; i++) {
            sx[i] = host_xyz[i * stride];
            sy[i] = host_xyz[i * stride + 1];
            sz[i] = host_xyz[i * stride + 2];
        }
    }
    for (size_t i = n; i < c->cap; i++) { sx[i] = std::numeric_limits<float>::infinity(); sy[i] = 0.f; sz[i] = 0.f; }
    hipError_t e = hipMemcpyAsync(c->base, st, 3 * c->cap * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { pcr_cloud_destroy(ctx, c); return fail(ctx, PCR_ERR_HIP, "cloud upload", e); }
    *out = c;
    return PCR_OK;
}

int pcr_cloud_clone(pcr_ctx* ctx, const pcr_cloud* src, pcr_cloud** out)
{
    if (!ctx || !src || !out) return fail(ctx, PCR_ERR_ARG, "pcr_cloud_clone");
    pcr_cloud* c = nullptr;
    int rc = cloud_alloc(ctx, src->n, &c);
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(c->base, src->base, 3 * c->cap * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream);
    if (e != hipSuccess) { pcr_cloud_destroy(ctx, c); return fail(ctx, PCR_ERR_HIP, "cloud clone", e); }
    *out = c;
    return PCR_OK;
}

int pcr_cloud_assign(pcr_ctx* ctx, pcr_cloud* dst, const pcr_cloud* src)
{
    if (!ctx || !dst || !src || dst->n != src->n) return fail(ctx, PCR_ERR_ARG, "pcr_cloud_assign");
    cloud_modified(dst);
    PCR_HIP(ctx, hipMemcpyAsync(dst->base, src->base, 3 * src->cap * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    return PCR_OK;
}

int pcr_cloud_read(pcr_ctx* ctx, const pcr_cloud* c, float* host_xyz, int layout)
{
    if (!ctx || !c || (c->n && !host_xyz) || !(layout == PCR_SOA || layout == PCR_AOS3 || layout == PCR_AOS4 || layout == PCR_AOS6))
        return fail(ctx, PCR_ERR_ARG, "pcr_cloud_read");
    const size_t n = c->n;
    if (n == 0) return PCR_OK;
    int rc = ensure_stage(ctx, 3 * c->cap * sizeof(float));
    if (rc) return rc;
    float* st = (float*)ctx->host_stage;
    PCR_HIP(ctx, hipMemcpyAsync(st, c->base, 3 * c->cap * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const float* sx = st, *sy = st + c->cap, *sz = st + 2 * c->cap;
    if (layout == PCR_SOA) {
        memcpy(host_xyz, sx, n * sizeof(float));
        memcpy(host_xyz + n, sy, n * sizeof(float));
        memcpy(host_xyz + 2 * n, sz, n * sizeof(float));
    } else {
        const size_t stride = layout == PCR_AOS3 ? 3 : (layout == PCR_AOS4 ? 4 : 6);
        for (size_t i = 0; i < n; i++) {
            host_xyz[i * stride] = sx[i];
            host_xyz[i * stride + 1] = sy[i];
            host_xyz[i * stride + 2] = sz[i];
        }
    }
    return PCR_OK;
}

size_t pcr_cloud_size(const pcr_cloud* c) { return c ? c->n : 0; }

int pcr_cloud_destroy(pcr_ctx* ctx, pcr_cloud* c)
{
    if (!c) return PCR_OK;
    if (ctx) hipStreamSynchronize(ctx->stream);
    cloud_forget(ctx, c);
    if (c->gidx) hipFree(c->gidx);
    if (c->base) hipFree(c->base);        // the caller's memory goes back to the device at once, whatever context the handle is destroyed through
    delete c;
    return PCR_OK;
}

int pcr_ctx_parked_bytes(const pcr_ctx* ctx, uint64_t* bytes)
{
    if (!ctx || !bytes) return PCR_ERR_ARG;
    *bytes = 0;
    for (int k = 0; k < 2; k++) if (ctx->spare_base[k]) *bytes += 3 * (uint64_t)ctx->spare_cap[k] * sizeof(float);
    return PCR_OK;
}

int pcr_ctx_trim(pcr_ctx* ctx)
{
    if (!ctx) return PCR_ERR_ARG;
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    spare_trim(ctx);
    return PCR_OK;
}

// ---------------------------------------------------------------------------------------------- 1-NN
int pcr_nn1_f32_async(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* src)
{
    if (!ctx || !tgt || !src) return fail(ctx, PCR_ERR_ARG, "pcr_nn1_f32_async");
    // tune "nn1_async_in_loop" = 1: the caller iterates (its own ICP-style loop on the same pair) — the search may then be seeded by the
    // correspondences of its previous call exactly as the searches inside pcr_icp_p2p_f32 are (same results, bit for bit)
    return launch_nn1(ctx, tgt, src, tune_get(ctx, "nn1_async_in_loop", 0) > 0);
}

int pcr_nn1_f32_loop(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* src, float max_corr)
{
    if (!ctx || !tgt || !src || !(max_corr > 0.0f)) return fail(ctx, PCR_ERR_ARG, "pcr_nn1_f32_loop");
    const float gate = tune_get(ctx, "icp_bounded_search", 1) == 1 ? max_corr : __builtin_inff();
    const LoopHint hint(ctx, 1000);       // (the dispatcher's rule for loops, as in pcr_cloud_sort_for_target)
    return launch_nn1(ctx, tgt, src, true, gate);
}

int pcr_cloud_sort_for_target(pcr_ctx* ctx, const pcr_cloud* tgt, pcr_cloud* cloud, uint32_t* orig_index)
{
    if (!ctx || !tgt || !cloud || cloud == tgt) return fail(ctx, PCR_ERR_ARG, "pcr_cloud_sort_for_target");
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = cloud->n;
    pcr_cloud* w = cloud;
    const LoopHint hint(ctx, 1000);       // (a caller that sorts its cloud iterates: the dispatcher's rule for loops)
    int rc = nn1_auto_grid(ctx, tgt, true, n) ? grid_sort_working_cloud(ctx, tgt, &w, true) : bt_sort_working_cloud(ctx, tgt, &w, true);
    if (rc) return rc;
    const bool sorted = ctx->work_orig_src == cloud && ctx->work_orig_n == n;
    if (orig_index && n) {
        if (sorted) {
            PCR_HIP(ctx, hipMemcpyAsync(orig_index, ctx->work_orig, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        } else {
            for (size_t i = 0; i < n; i++) orig_index[i] = (uint32_t)i;      // (the library left the order alone: small clouds, switched off)
        }
    }
    return PCR_OK;
}

int pcr_cloud_shard_spatial(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* full, int nranks, int rank, int chunks_per_rank, pcr_cloud** out)
{
    if (!ctx || !tgt || !full || !out || nranks < 1 || rank < 0 || rank >= nranks || chunks_per_rank < 0) return fail(ctx, PCR_ERR_ARG, "pcr_cloud_shard_spatial");
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    return cloud_shard_spatial(ctx, tgt, full, nranks, rank, chunks_per_rank ? chunks_per_rank : 64, out, cloud_alloc);
}

int pcr_cloud_global_index(pcr_ctx* ctx, const pcr_cloud* c, uint32_t* index)
{
    if (!ctx || !c || (c->n && !index)) return fail(ctx, PCR_ERR_ARG, "pcr_cloud_global_index");
    if (!c->gidx) return fail(ctx, PCR_ERR_STATE, "pcr_cloud_global_index: not a shard");
    if (c->n == 0) return PCR_OK;
    PCR_HIP(ctx, hipMemcpyAsync(index, c->gidx, c->n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PCR_OK;
}

int pcr_nn1_fetch(pcr_ctx* ctx, size_t n, uint32_t* idx, float* d2)
{
    if (!ctx || n != ctx->keys_n || (n && (!idx || !d2))) return fail(ctx, PCR_ERR_ARG, "pcr_nn1_fetch");
    if (n == 0) return PCR_OK;
    int rc = ensure_scratch(ctx, n * 8);
    if (rc) return rc;
    uint32_t* idx_dev = (uint32_t*)ctx->scratch;
    float* d2_dev = (float*)((char*)ctx->scratch + n * 4);
    rc = nn1_unpack(ctx, n, idx_dev, d2_dev);
    if (rc) return rc;
    PCR_HIP(ctx, hipMemcpyAsync(idx, idx_dev, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipMemcpyAsync(d2, d2_dev, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PCR_OK;
}

int pcr_nn1_f32(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* src, uint32_t* idx, float* d2)
{
    int rc = pcr_nn1_f32_async(ctx, tgt, src);
    if (rc) return rc;
    return pcr_nn1_fetch(ctx, src->n, idx, d2);
}

// ---------------------------------------------------------------------------------------------- A8 / A7
int pcr_transform_f32(pcr_ctx* ctx, pcr_cloud* cloud, const float T[16])
{
    if (!ctx || !cloud || !T) return fail(ctx, PCR_ERR_ARG, "pcr_transform_f32");
    const float R[9] = { T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10] };
    const float t[3] = { T[3], T[7], T[11] };
    cloud_modified(cloud);
    return launch_transform(ctx, cloud, R, t);
}

int pcr_kabsch_sums(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* src, float max_corr, double sums[16],
                    int64_t* last_kept, float* last_d2)
{
    if (!ctx || !tgt || !src || !sums) return fail(ctx, PCR_ERR_ARG, "pcr_kabsch_sums");
    if (src->n == 0) {
        for (int k = 0; k < 16; k++) sums[k] = 0.0;
        if (last_kept) *last_kept = -1;
        if (last_d2) *last_d2 = 0.f;
        return PCR_OK;
    }
    KabschPlan plan;
    int rc = kabsch_plan(ctx, tgt, max_corr, &plan);
    if (rc) return rc;
    rc = launch_kabsch_sums(ctx, tgt, src, max_corr, plan);
    if (rc) return rc;
    PCR_HIP(ctx, hipMemcpyAsync(ctx->host_out, ctx->dev_out, 19 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->host_out[18] != 0.0) return fail(ctx, PCR_ERR_STATE, "pcr_kabsch_sums: a kept source point lies more than 2^20 target extents away from the target");
    memcpy(sums, ctx->host_out, 16 * sizeof(double));
    if (last_kept) *last_kept = (int64_t)ctx->host_out[16];
    if (last_d2) *last_d2 = (float)ctx->host_out[17];
    return PCR_OK;
}

int pcr_kabsch_grid_exponent(float target_absmax, float max_corr) { return kabsch_grid_exponent(target_absmax, max_corr); }

int pcr_kabsch_limbs_to_sums(double row[55], int e, double sums[16])
{
    if (!row || !sums) return PCR_ERR_ARG;
    num::limbs_normalize_row(row);
    num::limbs_to_sums(row, e, sums);
    return PCR_OK;
}

int pcr_kabsch_solve(const double sums[16], float R[9], float t[3])
{
    if (!sums || !R || !t) return PCR_ERR_ARG;
    return kabsch_solve(sums, R, t);
}

// ---------------------------------------------------------------------------------------------- profiling / tuning
int pcr_prof_reset(pcr_ctx* ctx)
{
    if (!ctx) return PCR_ERR_ARG;
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    prof_flush(ctx);
    ctx->prof.clear();
    return PCR_OK;
}

int pcr_prof_get(pcr_ctx* ctx, const char* kernel, uint64_t* launches, double* total_ms)
{
    if (!ctx || !kernel) return PCR_ERR_ARG;
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    prof_flush(ctx);
    auto it = ctx->prof.find(kernel);
    if (launches) *launches = it == ctx->prof.end() ? 0 : it->second.launches;
    if (total_ms) *total_ms = it == ctx->prof.end() ? 0.0 : it->second.total_ms;
    return PCR_OK;
}

int pcr_prof_get_each(pcr_ctx* ctx, const char* kernel, double* ms, size_t cap, size_t* n)
{
    if (!ctx || !kernel || !n || (cap && !ms)) return PCR_ERR_ARG;
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    prof_flush(ctx);
    auto it = ctx->prof.find(kernel);
    *n = it == ctx->prof.end() ? 0 : it->second.each_ms.size();
    for (size_t k = 0; k < *n && k < cap; k++) ms[k] = (double)it->second.each_ms[k];
    return PCR_OK;
}

// (the two-value entry points keep the contract round 2 shipped — a caller built against that header passes a 2-double array;
// the four-value forms are the _v2 symbols)
int pcr_selftest_mfma_bf16_v2(pcr_ctx* ctx, int trials, double worst[4])
{
    if (!ctx || !worst || trials < 0 || trials > 4096) return PCR_ERR_ARG;
    return pcr::bt_mfma_selftest(ctx, trials, worst);
}

int pcr_selftest_mfma_f16_v2(pcr_ctx* ctx, int trials, double worst[4])
{
    if (!ctx || !worst || trials < 0 || trials > 4096) return PCR_ERR_ARG;
    return pcr::ht_mfma_selftest(ctx, trials, worst);
}

int pcr_selftest_mfma_bf16(pcr_ctx* ctx, int trials, double worst[2])
{
    double w4[4];
    if (!worst) return PCR_ERR_ARG;
    const int rc = pcr_selftest_mfma_bf16_v2(ctx, trials, w4);
    if (rc == PCR_OK) { worst[0] = w4[0]; worst[1] = w4[1]; }
    return rc;
}

int pcr_selftest_mfma_f16(pcr_ctx* ctx, int trials, double worst[2])
{
    double w4[4];
    if (!worst) return PCR_ERR_ARG;
    const int rc = pcr_selftest_mfma_f16_v2(ctx, trials, w4);
    if (rc == PCR_OK) { worst[0] = w4[0]; worst[1] = w4[1]; }
    return rc;
}

int pcr_selftest_sign_f16(pcr_ctx* ctx, int trials, uint64_t out[4])
{
    if (!ctx || !out || trials < 0 || trials > 4096) return PCR_ERR_ARG;
    unsigned long long r[4];
    const int rc = pcr::st_sign_selftest(ctx, trials, r);
    for (int k = 0; k < 4; k++) out[k] = r[k];
    return rc;
}

int pcr_selftest_sphere_f16(pcr_ctx* ctx, int trials, uint64_t out[4])
{
    if (!ctx || !out || trials < 0 || trials > 4096) return PCR_ERR_ARG;
    unsigned long long r[4];
    const int rc = pcr::st_sphere_selftest(ctx, trials, r);
    for (int k = 0; k < 4; k++) out[k] = r[k];
    return rc;
}

int pcr_ctx_mfma_check(pcr_ctx* ctx, int run_now, pcr_mfma_check* out)
{
    if (!ctx || !out) return PCR_ERR_ARG;
    if (run_now) { (void)pcr::mfma_verdict(ctx, true); (void)pcr::mfma_verdict(ctx, false); }
    const int64_t force = tune_get(ctx, "mfma_force_fail", 0);
    out->f16_ok = (force > 0 && (force & 1)) ? 0 : ctx->mfma_ok16;
    out->bf16_ok = (force > 0 && (force & 2)) ? 0 : ctx->mfma_okbf;
    for (int k = 0; k < 4; k++) { out->f16_worst[k] = ctx->mfma_worst16[k]; out->bf16_worst[k] = ctx->mfma_worstbf[k]; }
    out->check_ms = ctx->mfma_check_ms;
    strncpy(out->last_nn1_kernel, ctx->last_nn1_kernel, sizeof(out->last_nn1_kernel) - 1);
    out->last_nn1_kernel[sizeof(out->last_nn1_kernel) - 1] = 0;
    return PCR_OK;
}

int pcr_grid_stats(pcr_ctx* ctx, uint64_t out[4])
{
    if (!ctx || !out) return PCR_ERR_ARG;
    for (int k = 0; k < 4; k++) out[k] = 0;
    if (!ctx->grid_stats_dev) return PCR_OK;
    unsigned long long h[PCR_NSTATS];
    PCR_HIP(ctx, hipMemcpyAsync(h, ctx->grid_stats_dev, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 4; k++) out[k] = h[k];
    return PCR_OK;
}

int pcr_nn1_stats(pcr_ctx* ctx, uint64_t out[16])
{
    if (!ctx || !out) return PCR_ERR_ARG;
    for (int k = 0; k < PCR_NSTATS; k++) out[k] = 0;
    if (!ctx->grid_stats_dev) return PCR_OK;
    unsigned long long h[PCR_NSTATS];
    PCR_HIP(ctx, hipMemcpyAsync(h, ctx->grid_stats_dev, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < PCR_NSTATS; k++) out[k] = h[k];
    return PCR_OK;
}

int pcr_tune_set(pcr_ctx* ctx, const char* key, int64_t value)
{
    if (!ctx || !key) return PCR_ERR_ARG;
    if (!strcmp(key, "prof")) { ctx->prof_level = (int)value; return PCR_OK; }   // 0 off (default), 1 nn kernels, 2 all
    ctx->tune[key] = value;
    return PCR_OK;
}

void pcr_shard_range(size_t n, int nranks, int rank, size_t* begin, size_t* end)
{
    if (nranks < 1) nranks = 1;
    if (rank < 0) rank = 0;
    if (rank >= nranks) rank = nranks - 1;
    const size_t base = n / (size_t)nranks, rem = n % (size_t)nranks;
    const size_t b = (size_t)rank * base + ((size_t)rank < rem ? (size_t)rank : rem);
    const size_t e = b + base + ((size_t)rank < rem ? 1 : 0);
    if (begin) *begin = b;
    if (end) *end = e;
}

}  // extern "C"

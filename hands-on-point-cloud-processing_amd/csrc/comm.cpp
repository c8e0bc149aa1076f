// comm.cpp — the one collective of the path: all-reduce(sum) of 56 + 2 * nranks <= 128 f64 per ICP iteration.
// Transport 1: RCCL over xGMI (one process per GPU).  librccl is bound with dlopen at first use so that a
// single-GPU process never loads it and so that, inside a PyTorch process, the already loaded
// librccl.so.1 (same SONAME) is shared instead of a second copy.
// Transport 2: a host callback (caller's own process group; used by the gloo CPU tests).
// The message is 128-192 B: pure latency (one ring/tree hop chain), nowhere near the 7 x ~153 GB/s xGMI
// link bound — see DESIGN.md §multi-GPU.
#include "pcr_internal.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>   // declarations only: the entry points are bound with dlopen below, their types come from here

#include <mutex>
#include <string>
#include <vector>

namespace pcr {

namespace {

// the ABI this file binds by name is the one rccl.h declares: the signatures are taken from the header (decltype), the
// by-value id is the header's struct, and the id crosses the C ABI of pcr.h as PCR_COMM_ID_BYTES opaque bytes
static_assert(sizeof(ncclUniqueId) == PCR_COMM_ID_BYTES, "pcr.h: PCR_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
static_assert(NCCL_UNIQUE_ID_BYTES == PCR_COMM_ID_BYTES, "pcr.h: PCR_COMM_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
    std::string path;     // what dlopen was given
};

void rccl_load(Rccl& r);

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;                     // several contexts (threads) may attach communicators concurrently
    std::call_once(once, [] { rccl_load(r); });
    return r;
}

void rccl_load(Rccl& r)
{
    // The communicator must live in the SAME HIP runtime as this library's streams and buffers.  A process can hold two
    // (PyTorch ships its own libamdhip64 + librccl under torch/lib; which one a bare SONAME resolves to depends on what was
    // loaded first), so the RCCL next to the libamdhip64 that THIS library is bound to is tried first, by path.
    std::string beside;
    Dl_info info;
    if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {
        beside = info.dli_fname;
        const size_t slash = beside.rfind('/');
        beside = slash == std::string::npos ? std::string() : beside.substr(0, slash + 1);
    }
    std::vector<std::string> names;
    if (!beside.empty()) { names.push_back(beside + "librccl.so.1"); names.push_back(beside + "librccl.so"); }
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    names.push_back("/opt/rocm/lib/librccl.so.1");
    for (const std::string& n : names) {
        r.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) { r.path = n; break; }
    }
    if (!r.handle) { r.err = std::string("dlopen(librccl): ") + dlerror(); return; }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) {
        r.err = "librccl: missing symbols";
        r.handle = nullptr;
    }
}

constexpr ncclDataType_t kNcclFloat64 = ncclFloat64;
constexpr ncclRedOp_t kNcclSum = ncclSum;

}  // namespace

int comm_allreduce_f64(pcr_ctx* ctx, double* host_buf, double* dev_buf, int n)
{
    Comm& c = ctx->comm;
    if (c.nranks <= 1) return PCR_OK;
    if (c.cb) {
        int rc = c.cb(c.cb_user, host_buf, n);
        if (rc != 0) return fail(ctx, PCR_ERR_COMM, "allreduce callback failed");
        return PCR_OK;
    }
    if (c.rccl) {
        Rccl& r = rccl();
        PCR_HIP(ctx, hipMemcpyAsync(dev_buf, host_buf, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        ncclResult_t rc = r.AllReduce(dev_buf, dev_buf, (size_t)n, kNcclFloat64, kNcclSum, (ncclComm_t)c.rccl, ctx->stream);
        if (rc != ncclSuccess) {
            ctx->err = std::string("ncclAllReduce: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error");
            return PCR_ERR_COMM;
        }
        PCR_HIP(ctx, hipMemcpyAsync(host_buf, dev_buf, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return PCR_OK;
    }
    return fail(ctx, PCR_ERR_STATE, "nranks > 1 but no transport attached");
}

int comm_allreduce_f64_device(pcr_ctx* ctx, double* dev_buf, int n)
{
    Comm& c = ctx->comm;
    // a one-rank job needs no collective; with a communicator attached and the test knob "icp_force_slots" set, the
    // call is made anyway (sum over one rank = identity) so that the path can be exercised on a single GPU
    if (c.nranks <= 1 && !(c.rccl && tune_get(ctx, "icp_force_slots", 0) > 0)) return PCR_OK;
    if (!c.rccl) return fail(ctx, PCR_ERR_STATE, "device all-reduce needs the RCCL transport");
    Rccl& r = rccl();
    ncclResult_t rc = r.AllReduce(dev_buf, dev_buf, (size_t)n, kNcclFloat64, kNcclSum, (ncclComm_t)c.rccl, ctx->stream);
    if (rc != ncclSuccess) {
        ctx->err = std::string("ncclAllReduce: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error");
        return PCR_ERR_COMM;
    }
    return PCR_OK;
}

}  // namespace pcr

using namespace pcr;

extern "C" {

int pcr_comm_unique_id(char id[PCR_COMM_ID_BYTES])
{
    if (!id) return PCR_ERR_ARG;
    Rccl& r = rccl();
    if (!r.handle) { fprintf(stderr, "pcr_comm_unique_id: %s\n", r.err.c_str()); return PCR_ERR_COMM; }
    ncclUniqueId u;
    memset(&u, 0, sizeof u);
    ncclResult_t rc = r.GetUniqueId(&u);
    if (rc != ncclSuccess) return PCR_ERR_COMM;
    memcpy(id, u.internal, PCR_COMM_ID_BYTES);
    return PCR_OK;
}

static_assert(56 + 2 * PCR_MAX_RANKS <= 128, "dev_out / host_out hold 128 doubles (api.cpp); icp_nred(nranks) = 56 + 2 * nranks");

int pcr_comm_init_rccl(pcr_ctx* ctx, int nranks, int rank, const char id[PCR_COMM_ID_BYTES])
{
    if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, PCR_ERR_ARG, "pcr_comm_init_rccl");
    if (nranks > PCR_MAX_RANKS) return fail(ctx, PCR_ERR_ARG, "pcr_comm_init_rccl: more than PCR_MAX_RANKS ranks (the reduce buffer holds 56 + 2 * nranks <= 128 f64)");
    pcr_comm_destroy(ctx);
    Rccl& r = rccl();
    if (!r.handle) return fail(ctx, PCR_ERR_COMM, r.err.c_str());
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(u.internal, id, PCR_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    ncclResult_t rc = r.CommInitRank(&comm, nranks, u, rank);
    if (rc != ncclSuccess) {
        ctx->err = std::string("ncclCommInitRank: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error");
        return PCR_ERR_COMM;
    }
    ctx->comm.nranks = nranks;
    ctx->comm.rank = rank;
    ctx->comm.rccl = comm;
    return PCR_OK;
}

int pcr_comm_init_callback(pcr_ctx* ctx, int nranks, int rank, pcr_allreduce_fn fn, void* user)
{
    if (!ctx || !fn || nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, PCR_ERR_ARG, "pcr_comm_init_callback");
    if (nranks > PCR_MAX_RANKS) return fail(ctx, PCR_ERR_ARG, "pcr_comm_init_callback: more than PCR_MAX_RANKS ranks (the reduce buffer holds 56 + 2 * nranks <= 128 f64)");
    pcr_comm_destroy(ctx);
    ctx->comm.nranks = nranks;
    ctx->comm.rank = rank;
    ctx->comm.cb = fn;
    ctx->comm.cb_user = user;
    return PCR_OK;
}

// Runs one real ncclAllReduce(sum, f64) of 8 values on the attached RCCL communicator (also with nranks == 1, where
// the loop itself skips the collective) and checks the result: exercises the dlopen'ed ABI end to end.
int pcr_comm_selftest(pcr_ctx* ctx)
{
    if (!ctx) return PCR_ERR_ARG;
    Comm& c = ctx->comm;
    if (!c.rccl) return fail(ctx, PCR_ERR_STATE, "pcr_comm_selftest: no RCCL communicator attached");
    Rccl& r = rccl();
    double h[8];
    for (int k = 0; k < 8; k++) h[k] = (double)(k + 1) * (c.rank + 1);
    PCR_HIP(ctx, hipMemcpyAsync(ctx->dev_out, h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
    ncclResult_t rc = r.AllReduce(ctx->dev_out, ctx->dev_out, 8, kNcclFloat64, kNcclSum, (ncclComm_t)c.rccl, ctx->stream);
    if (rc != ncclSuccess) {
        ctx->err = std::string("ncclAllReduce: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error");
        return PCR_ERR_COMM;
    }
    PCR_HIP(ctx, hipMemcpyAsync(h, ctx->dev_out, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const double tri = (double)c.nranks * (c.nranks + 1) / 2.0;   // sum over ranks of (rank + 1)
    for (int k = 0; k < 8; k++)
        if (h[k] != (double)(k + 1) * tri) return fail(ctx, PCR_ERR_COMM, "pcr_comm_selftest: wrong all-reduce result");
    return PCR_OK;
}

int pcr_comm_destroy(pcr_ctx* ctx)
{
    if (!ctx) return PCR_ERR_ARG;
    if (ctx->comm.rccl) {
        Rccl& r = rccl();
        if (r.handle) r.CommDestroy((ncclComm_t)ctx->comm.rccl);
    }
    ctx->comm = Comm();
    return PCR_OK;
}

}  // extern "C"

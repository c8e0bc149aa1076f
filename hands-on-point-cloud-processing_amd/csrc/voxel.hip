// voxel.hip — voxel-grid down-sampling (next row N3 of SURVEY.md §8f): Homework1 voxel_filter.py:17-52, centroid mode,
// reproduced bit for bit on the GPU so that the front of the registration pipeline no longer needs the CPU.
//   h(p) = hx + hy*Dx + hz*Dx*Dy,  h* = floor((p* - min*) / leaf)   (f32 subtraction, f64 division: the numpy-1.x
//   promotion the author ran), stable sort by h, one centroid per voxel = sequential f32 sum of its points in ascending
//   index order / count — and the reference's quirk: a voxel is emitted when the NEXT one starts (:43-50), so the last
//   voxel of the sorted order is dropped.
// Pipeline: bounds (reduction) -> keys -> stable radix sort of (h, index) [rocPRIM, sort.hip] -> segment heads + scan -> one
// lane per voxel walks its points in order (f32 sequential sum is order dependent: no tree reduction here).
// Algorithmic traffic: 12 B/pt read for the bounds, 12 B/pt for the keys, 12 B/pt sorted, 12 B/pt gathered + 12 B per
// output point: an HBM-bound pass sequence.
#include "pcr_internal.hpp"

#include "sort.hpp"   // stable device radix sort (rocPRIM, sort.hip; the rest is hand-written)

#include <cfloat>
#include <cmath>

#pragma clang fp contract(off)

namespace pcr {

constexpr int VX_BLOCK = 256;

// out[block][5] = {min x, min y, min z, max x, max y}  (Python min()/max() over the f32 columns, :22-26)
__global__ __launch_bounds__(VX_BLOCK) void vx_bounds_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ z, uint32_t n, float* __restrict__ out)
{
    float v[5] = { FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (uint32_t i = blockIdx.x * VX_BLOCK + threadIdx.x; i < n; i += gridDim.x * VX_BLOCK) {
        v[0] = fminf(v[0], x[i]); v[1] = fminf(v[1], y[i]); v[2] = fminf(v[2], z[i]);
        v[3] = fmaxf(v[3], x[i]); v[4] = fmaxf(v[4], y[i]);
    }
    __shared__ float red[VX_BLOCK / 64][5];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < 5; c++) {
        float a = v[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a = c < 3 ? fminf(a, __shfl_down(a, o, 64)) : fmaxf(a, __shfl_down(a, o, 64));
        if (lane == 0) red[wave][c] = a;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        const int c = threadIdx.x;
        float a = red[0][c];
        for (int w = 1; w < VX_BLOCK / 64; w++) a = c < 3 ? fminf(a, red[w][c]) : fmaxf(a, red[w][c]);
        out[blockIdx.x * 5 + c] = a;
    }
}

struct VxParams {
    float x_min, y_min, z_min;
    double leaf;
    long long Dx, Dy;
};

__global__ __launch_bounds__(VX_BLOCK) void vx_keys_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ z, uint32_t n, VxParams p,
                                                           unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const uint32_t i = blockIdx.x * VX_BLOCK + threadIdx.x;
    if (i >= n) return;
    // :33-36 — (f32 - f32) in f32, then / leaf in f64, floor
    const long long hx = (long long)floor((double)(x[i] - p.x_min) / p.leaf);
    const long long hy = (long long)floor((double)(y[i] - p.y_min) / p.leaf);
    const long long hz = (long long)floor((double)(z[i] - p.z_min) / p.leaf);
    keys[i] = (unsigned long long)(hx + hy * p.Dx + hz * p.Dx * p.Dy);
    vals[i] = i;
}

__global__ __launch_bounds__(VX_BLOCK) void vx_heads_kernel(const unsigned long long* __restrict__ keys, uint32_t n,
                                                            uint32_t* __restrict__ flags)
{
    const uint32_t i = blockIdx.x * VX_BLOCK + threadIdx.x;
    if (i >= n) return;
    flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

// starts[group] = first sorted position of the group; group id = exclusive scan of the head flags at that position
__global__ __launch_bounds__(VX_BLOCK) void vx_starts_kernel(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ gid,
                                                             uint32_t n, uint32_t* __restrict__ starts)
{
    const uint32_t i = blockIdx.x * VX_BLOCK + threadIdx.x;
    if (i >= n) return;
    if (flags[i]) starts[gid[i]] = i;
}

__global__ __launch_bounds__(VX_BLOCK) void vx_centroid_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                               const float* __restrict__ z, const uint32_t* __restrict__ order,
                                                               const uint32_t* __restrict__ starts, uint32_t n_out,
                                                               float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz)
{
    const uint32_t g = blockIdx.x * VX_BLOCK + threadIdx.x;
    if (g >= n_out) return;
    const uint32_t b = starts[g], e = starts[g + 1];     // group g + 1 exists: the last group is never emitted (:41-50)
    float sx = 0.0f, sy = 0.0f, sz = 0.0f;
    // ascending point index (the sort is stable): np.sum order.  The additions stay one sequential chain per voxel (that order IS
    // the result); the gathers do not have to wait for it: eight points are fetched at a time
    constexpr int U = 8;
    uint32_t t = b;
    for (; t + U <= e; t += U) {
        uint32_t i[U];
        float vx[U], vy[U], vz[U];
#pragma unroll
        for (int u = 0; u < U; u++) i[u] = order[t + u];
#pragma unroll
        for (int u = 0; u < U; u++) { vx[u] = x[i[u]]; vy[u] = y[i[u]]; vz[u] = z[i[u]]; }
#pragma unroll
        for (int u = 0; u < U; u++) { sx += vx[u]; sy += vy[u]; sz += vz[u]; }
    }
    for (; t < e; t++) {
        const uint32_t i = order[t];
        sx += x[i]; sy += y[i]; sz += z[i];
    }
    const float cnt = (float)(e - b);
    ox[g] = sx / cnt; oy[g] = sy / cnt; oz[g] = sz / cnt;
}

__global__ void vx_pad_kernel(float* __restrict__ x, float* __restrict__ y, float* __restrict__ z, uint32_t n, uint32_t cap)
{
    const uint32_t i = n + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) { x[i] = __builtin_inff(); y[i] = 0.0f; z[i] = 0.0f; }
}

}  // namespace pcr

using namespace pcr;

extern "C" int pcr_voxel_filter_f32(pcr_ctx* ctx, const pcr_cloud* in, double leaf_size, pcr_cloud** out)
{
    if (!ctx || !in || !out || !(leaf_size > 0.0)) return fail(ctx, PCR_ERR_ARG, "pcr_voxel_filter_f32");
    *out = nullptr;
    PCR_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = in->n;
    if (n > 0x7FFFFFF0ull) return fail(ctx, PCR_ERR_ARG, "pcr_voxel_filter_f32: cloud too large");
    pcr_cloud* res = nullptr;
    if (n == 0) {
        float dummy = 0.f;
        return pcr_cloud_create(ctx, &dummy, 0, PCR_SOA, out);
    }
    // 1. bounds
    const uint32_t bb = (uint32_t)std::max<size_t>(1, std::min<size_t>(256, (n + VX_BLOCK - 1) / VX_BLOCK));
    int rc = ensure_scratch(ctx, bb * 5 * sizeof(float));
    if (rc) return rc;
    hipLaunchKernelGGL(vx_bounds_kernel, dim3(bb), dim3(VX_BLOCK), 0, ctx->stream, in->x(), in->y(), in->z(), (uint32_t)n, (float*)ctx->scratch);
    std::vector<float> hb(bb * 5);
    PCR_HIP(ctx, hipMemcpyAsync(hb.data(), ctx->scratch, hb.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float b5[5] = { FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (uint32_t k = 0; k < bb; k++)
        for (int c = 0; c < 5; c++) b5[c] = c < 3 ? std::min(b5[c], hb[k * 5 + c]) : std::max(b5[c], hb[k * 5 + c]);
    VxParams p;
    p.x_min = b5[0]; p.y_min = b5[1]; p.z_min = b5[2];
    p.leaf = leaf_size;
    p.Dx = (long long)std::ceil((double)(float)(b5[3] - b5[0]) / leaf_size);      // :28
    p.Dy = (long long)std::ceil((double)(float)(b5[4] - b5[1]) / leaf_size);      // :29
    // 2. keys + stable sort
    size_t temp_bytes = 0;
    sort_pairs_u64_u32(nullptr, temp_bytes, nullptr, nullptr, nullptr, nullptr, n, 0, 64, ctx->stream);
    const size_t kb = ((n * 8 + 255) & ~(size_t)255), vb = ((n * 4 + 255) & ~(size_t)255);
    const size_t nb = (n + 1 + SCAN_TILE - 1) / SCAN_TILE;
    const size_t off_kout = kb, off_vin = 2 * kb, off_vout = 2 * kb + vb, off_flags = 2 * kb + 2 * vb, off_gid = off_flags + vb,
                 off_starts = off_gid + vb, off_tot = off_starts + vb + 256, off_temp = off_tot + ((nb + 2) * 4 + 255 & ~(size_t)255);
    rc = ensure_scratch(ctx, off_temp + temp_bytes + 256);
    if (rc) return rc;
    char* s = (char*)ctx->scratch;
    unsigned long long* k_in = (unsigned long long*)s;
    unsigned long long* k_out = (unsigned long long*)(s + off_kout);
    uint32_t* v_in = (uint32_t*)(s + off_vin);
    uint32_t* v_out = (uint32_t*)(s + off_vout);
    uint32_t* flags = (uint32_t*)(s + off_flags);
    uint32_t* gid = (uint32_t*)(s + off_gid);
    uint32_t* starts = (uint32_t*)(s + off_starts);
    uint32_t* totals = (uint32_t*)(s + off_tot);
    const dim3 gridn((unsigned)((n + VX_BLOCK - 1) / VX_BLOCK));
    hipLaunchKernelGGL(vx_keys_kernel, gridn, dim3(VX_BLOCK), 0, ctx->stream, in->x(), in->y(), in->z(), (uint32_t)n, p, k_in, v_in);
    PCR_HIP(ctx, sort_pairs_u64_u32(s + off_temp, temp_bytes, k_in, k_out, v_in, v_out, n, 0, 64, ctx->stream));
    // 3. segments
    hipLaunchKernelGGL(vx_heads_kernel, gridn, dim3(VX_BLOCK), 0, ctx->stream, k_out, (uint32_t)n, flags);
    rc = exclusive_scan_u32(ctx, flags, gid, n, totals, totals + nb);
    if (rc) return rc;
    uint32_t groups = 0;
    PCR_HIP(ctx, hipMemcpyAsync(&groups, totals + nb, 4, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    hipLaunchKernelGGL(vx_starts_kernel, gridn, dim3(VX_BLOCK), 0, ctx->stream, flags, gid, (uint32_t)n, starts);
    const size_t n_out = groups ? groups - 1 : 0;        // the last voxel is never emitted
    // 4. output cloud
    float dummy = 0.f;
    rc = pcr_cloud_create(ctx, &dummy, 0, PCR_SOA, &res);          // allocates an empty padded cloud ...
    if (rc) return rc;
    if (n_out) {
        pcr_cloud_destroy(ctx, res);                                 // ... replaced by one of the right size
        res = new (std::nothrow) pcr_cloud();
        if (!res) return fail(ctx, PCR_ERR_NOMEM, "voxel out");
        res->n = n_out;
        res->cap = padded(n_out);
        hipError_t e = hipMalloc((void**)&res->base, 3 * res->cap * sizeof(float));
        if (e != hipSuccess) { delete res; return fail(ctx, PCR_ERR_HIP, "hipMalloc(voxel out)", e); }
        {
            ProfScope pr(ctx, "voxel_centroid");
            hipLaunchKernelGGL(vx_centroid_kernel, dim3((unsigned)((n_out + VX_BLOCK - 1) / VX_BLOCK)), dim3(VX_BLOCK), 0, ctx->stream,
                               in->x(), in->y(), in->z(), v_out, starts, (uint32_t)n_out, res->x(), res->y(), res->z());
        }
        const uint32_t padn = (uint32_t)(res->cap - n_out);
        hipLaunchKernelGGL(vx_pad_kernel, dim3((padn + 255) / 256), dim3(256), 0, ctx->stream, res->x(), res->y(), res->z(), (uint32_t)n_out, (uint32_t)res->cap);
        hipError_t e2 = hipGetLastError();
        if (e2 == hipSuccess) e2 = hipStreamSynchronize(ctx->stream);
        if (e2 != hipSuccess) { pcr_cloud_destroy(ctx, res); return fail(ctx, PCR_ERR_HIP, "voxel filter", e2); }
    }
    *out = res;
    return PCR_OK;
}

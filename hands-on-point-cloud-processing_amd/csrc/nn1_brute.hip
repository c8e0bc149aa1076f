// nn1_brute.hip — brute-force 1-NN correspondence for gfx950 (MI355X), LDS-tiled.
//
// Replaces the per-query kd-tree descent of the reference's ICP inner loop
// (Homework9/hw9/src/registration.cpp:925-934 -> nanoflann.hpp:1222,1347) by an exhaustive scan that
// returns the same argmin: A1 arithmetic (nanoflann.hpp:403-406: ((dx*dx + dy*dy) + dz*dz), every op
// rounded to f32, NO fma) and the canonical tie rule "min d2, then lowest index".
//
// Mapping to CDNA4:
//  * one query per lane, QPL queries per lane held in VGPRs for the whole kernel;
//  * targets stream HBM -> registers (coalesced float4 of the SoA arrays) -> LDS tile of TILE points;
//    the next tile's global loads are issued before the current tile is consumed (software pipeline);
//  * every lane reads the same LDS address (ds_read_b128 broadcast: 4 targets per coordinate per read),
//    so LDS traffic is 12 B per 64*QPL pair evaluations and never bank-conflicts;
//  * inner loop per 8 targets and query: 64 VALU ops of distance arithmetic + 4 v_min3_u32 on the raw bit
//    patterns (d2 >= 0, so IEEE bits are order preserving; NaN bits sort above +inf and never win) + one
//    compare; the index is only resolved inside a rarely taken branch;
//  * the target set is cut into slices (gridDim.y) so that >> 256 workgroups exist even for one scan;
//    slices merge through one 64-bit atomicMin per query on key = d2_bits << 32 | idx, which implements
//    "min d2, then lowest index" exactly and independently of arrival order.
// The kernel is VALU-bound: 9 algorithmic lane-ops per (query, target) pair.
#include "pcr_internal.hpp"

#include <cfloat>

#pragma clang fp contract(off)

namespace pcr {

constexpr int NN_BLOCK = 256;
constexpr int NN_TILE = 1024;   // targets per LDS tile: 3 * 4 KiB

__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c)
{
    return min(min(a, b), c);   // -> v_min3_u32
}

template <int QPL>
__global__ __launch_bounds__(NN_BLOCK) void nn1_brute_kernel(
    const float* __restrict__ tx, const float* __restrict__ ty, const float* __restrict__ tz,
    const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
    uint32_t ns, uint32_t n_tiles, uint32_t tiles_per_slice,
    unsigned long long* __restrict__ keys, int merge_atomic)
{
    __shared__ float4 lx[NN_TILE / 4];
    __shared__ float4 ly[NN_TILE / 4];
    __shared__ float4 lz[NN_TILE / 4];

    const uint32_t tid = threadIdx.x;
    const uint32_t qbase = blockIdx.x * (NN_BLOCK * QPL);

    float qx[QPL], qy[QPL], qz[QPL];
    uint32_t best[QPL], bidx[QPL];
#pragma unroll
    for (int k = 0; k < QPL; k++) {
        uint32_t i = min(qbase + k * NN_BLOCK + tid, ns - 1);
        qx[k] = sx[i]; qy[k] = sy[i]; qz[k] = sz[i];
        best[k] = 0x7F7FFFFFu;          // FLT_MAX: nanoflann.hpp:163; accept only d2 < worst (:1360)
        bidx[k] = 0xFFFFFFFFu;
    }

    const uint32_t tile0 = blockIdx.y * tiles_per_slice;
    const uint32_t tile1 = min(tile0 + tiles_per_slice, n_tiles);
    if (tile0 < tile1) {
        const float4* gx = reinterpret_cast<const float4*>(tx) + (size_t)tile0 * (NN_TILE / 4);
        const float4* gy = reinterpret_cast<const float4*>(ty) + (size_t)tile0 * (NN_TILE / 4);
        const float4* gz = reinterpret_cast<const float4*>(tz) + (size_t)tile0 * (NN_TILE / 4);
        float4 rx = gx[tid], ry = gy[tid], rz = gz[tid];
        for (uint32_t tile = tile0; tile < tile1; tile++) {
            lx[tid] = rx; ly[tid] = ry; lz[tid] = rz;
            __syncthreads();
            if (tile + 1 < tile1) {     // prefetch the next tile while this one is consumed
                gx += NN_TILE / 4; gy += NN_TILE / 4; gz += NN_TILE / 4;
                rx = gx[tid]; ry = gy[tid]; rz = gz[tid];
            }
            const uint32_t jbase = tile * NN_TILE;
#pragma unroll 2
            for (int c = 0; c < NN_TILE / 8; c++) {
                const float4 xa = lx[2 * c], xb = lx[2 * c + 1];
                const float4 ya = ly[2 * c], yb = ly[2 * c + 1];
                const float4 za = lz[2 * c], zb = lz[2 * c + 1];
                const float X[8] = { xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w };
                const float Y[8] = { ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w };
                const float Z[8] = { za.x, za.y, za.z, za.w, zb.x, zb.y, zb.z, zb.w };
#pragma unroll
                for (int k = 0; k < QPL; k++) {
                    uint32_t d[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const float dx = qx[k] - X[j];
                        const float dy = qy[k] - Y[j];
                        const float dz = qz[k] - Z[j];
                        const float s = (dx * dx + dy * dy) + dz * dz;      // A1, unfused
                        d[j] = __float_as_uint(s);
                    }
                    uint32_t m = umin3(d[0], d[1], d[2]);
                    m = umin3(m, d[3], d[4]);
                    m = umin3(m, d[5], d[6]);
                    m = min(m, d[7]);
                    if (__builtin_expect(m < best[k], 0)) {
                        // rare: ascending scan with strict < keeps the lowest index among equal minima
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            if (d[j] < best[k]) { best[k] = d[j]; bidx[k] = jbase + 8 * c + j; }
                        }
                    }
                }
            }
            __syncthreads();
        }
    }

#pragma unroll
    for (int k = 0; k < QPL; k++) {
        const uint32_t i = qbase + k * NN_BLOCK + tid;
        if (i < ns) {
            const uint32_t bits = (bidx[k] == 0xFFFFFFFFu) ? 0x7F800000u : best[k];   // nothing accepted: +inf
            const unsigned long long key = ((unsigned long long)bits << 32) | bidx[k];
            if (merge_atomic) atomicMin(&keys[i], key);
            else keys[i] = key;
        }
    }
}

// keys -> (idx, d2) split for the host-facing API
__global__ void nn1_unpack_kernel(const unsigned long long* __restrict__ keys, uint32_t n,
                                  uint32_t* __restrict__ idx, float* __restrict__ d2)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        unsigned long long k = keys[i];
        idx[i] = (uint32_t)(k & 0xFFFFFFFFull);
        d2[i] = __uint_as_float((uint32_t)(k >> 32));
    }
}

int launch_nn1_brute(pcr_ctx* ctx, const pcr_cloud* tgt, const pcr_cloud* src)
{
    const size_t ns = src->n;
    if (ns == 0) { ctx->keys_n = 0; return PCR_OK; }
    if (ns > 0xFFFFFFF0ull || tgt->n > 0xFFFFFFF0ull) return fail(ctx, PCR_ERR_ARG, "cloud too large for u32 indices");
    int rc = ensure_keys(ctx, ns);
    if (rc) return rc;
    ctx->keys_n = ns;

    const int qpl = (int)tune_get(ctx, "nn1_qpl", 2);
    const uint32_t n_tiles = (uint32_t)((tgt->n + NN_TILE - 1) / NN_TILE);
    const uint32_t qblocks = (uint32_t)((ns + (size_t)NN_BLOCK * qpl - 1) / ((size_t)NN_BLOCK * qpl));
    // enough workgroups to balance 256 CUs x 8 resident blocks over several rounds
    int64_t tps = tune_get(ctx, "nn1_tiles_per_slice", 0);
    if (tps <= 0) {
        const int64_t want_blocks = tune_get(ctx, "nn1_target_blocks", 8192);
        int64_t slices = (want_blocks + qblocks - 1) / qblocks;
        if (slices < 1) slices = 1;
        tps = n_tiles ? (n_tiles + slices - 1) / slices : 1;
        if (tps < 1) tps = 1;
    }
    uint32_t slices = n_tiles ? (uint32_t)((n_tiles + tps - 1) / tps) : 1;
    if (slices > 65535) { slices = 65535; tps = (n_tiles + slices - 1) / slices; slices = (uint32_t)((n_tiles + tps - 1) / tps); }
    const int merge_atomic = slices > 1;
    if (merge_atomic) PCR_HIP(ctx, hipMemsetAsync(ctx->keys, 0xFF, ns * sizeof(unsigned long long), ctx->stream));

    dim3 grid(qblocks, slices), block(NN_BLOCK);
    {
        ProfScope p(ctx, "nn1_brute");
        switch (qpl) {
        case 1:
            hipLaunchKernelGGL(nn1_brute_kernel<1>, grid, block, 0, ctx->stream, tgt->x(), tgt->y(), tgt->z(),
                               src->x(), src->y(), src->z(), (uint32_t)ns, n_tiles, (uint32_t)tps, ctx->keys, merge_atomic);
            break;
        case 4:
            hipLaunchKernelGGL(nn1_brute_kernel<4>, grid, block, 0, ctx->stream, tgt->x(), tgt->y(), tgt->z(),
                               src->x(), src->y(), src->z(), (uint32_t)ns, n_tiles, (uint32_t)tps, ctx->keys, merge_atomic);
            break;
        default:
            hipLaunchKernelGGL(nn1_brute_kernel<2>, grid, block, 0, ctx->stream, tgt->x(), tgt->y(), tgt->z(),
                               src->x(), src->y(), src->z(), (uint32_t)ns, n_tiles, (uint32_t)tps, ctx->keys, merge_atomic);
            break;
        }
    }
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

int nn1_unpack(pcr_ctx* ctx, size_t n, uint32_t* idx_dev, float* d2_dev)
{
    if (n == 0) return PCR_OK;
    hipLaunchKernelGGL(nn1_unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       ctx->keys, (uint32_t)n, idx_dev, d2_dev);
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

}  // namespace pcr

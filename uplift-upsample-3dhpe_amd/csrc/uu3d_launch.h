// uu3d_launch.h -- host-side launch helpers shared by the ops ABI and the training step.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <algorithm>
#include "../../include/uu3d.h"
#include "uu3d_gemm.h"
#include "uu3d_gemm_h3.h"
#include "uu3d_bwd.h"

namespace uu3d {

constexpr size_t kOpScratchFloats = (size_t)1536 * 4096;

inline int ru(int v, int m) { return (v + m - 1) / m * m; }
inline hipError_t& last_launch_error() { static thread_local hipError_t e = hipSuccess; return e; }      // what hip_status() last consumed (for the error text)
inline int hip_status() { const hipError_t e = hipGetLastError(); if (e != hipSuccess) last_launch_error() = e; return e == hipSuccess ? UU3D_OK : UU3D_ERR_HIP; }

// C = A-op x Bt^T with the forward GEMM kernel (64x64 tiles, deterministic split-K when few tiles).
// Bh / Bl != nullptr: the operand's f16 hi / lo planes (same [Np][Kp] layout) -> f16x3 kernel (uu3d_gemm_h3.h).
template <class AL, class EP>
int launch_gemm(const AL& al, const float* Bt, int M, int N, int K, const EP& ep, float* slab, size_t slab_floats, hipStream_t stream,
                const _Float16* Bh = nullptr, const _Float16* Bl = nullptr) {
    const int Kp = ru(K, 32), KT = Kp / 32;
    const int tiles = ((M + 63) / 64) * ((N + 63) / 64);
    int slices = 1;
    if (tiles < 384 && KT >= 8) slices = std::max(1, std::min(KT / 4, (768 + tiles / 2) / tiles));     // (training step: 384 / 1536 target workgroups measure +0.5 %, no split +6 %)
    int kps = (KT + slices - 1) / slices;
    slices = (KT + kps - 1) / kps;
    const int ldslab = ru(N, 4);
    if (slices > 1 && (slab == nullptr || (size_t)slices * M * ldslab > slab_floats)) { slices = 1; kps = KT; }
    const int mt = (M + 63) / 64, nt = (N + 63) / 64;
    const int grid = ru(mt, 8) * nt;
    if (Bh != nullptr && Bl != nullptr) {
        constexpr size_t ldsh = gemm_h3_lds_bytes(64, 64);
        if (slices == 1) {
            if (gemm_h3_deep(grid)) hipLaunchKernelGGL((gemm_h3_kernel<1, 1, AL, EP, 1>), dim3(grid, 1), dim3(256), ldsh, stream, al, Bh, Bl, M, N, Kp, mt, nt, KT, ep);
            else hipLaunchKernelGGL((gemm_h3_kernel<1, 1, AL, EP>), dim3(grid, 1), dim3(256), ldsh, stream, al, Bh, Bl, M, N, Kp, mt, nt, KT, ep);
        } else {
            EpSlab es{slab, ldslab, (size_t)M * ldslab};
            if (gemm_h3_deep(grid * slices)) hipLaunchKernelGGL((gemm_h3_kernel<1, 1, AL, EpSlab, 1>), dim3(grid, slices), dim3(256), ldsh, stream, al, Bh, Bl, M, N, Kp, mt, nt, kps, es);
            else hipLaunchKernelGGL((gemm_h3_kernel<1, 1, AL, EpSlab>), dim3(grid, slices), dim3(256), ldsh, stream, al, Bh, Bl, M, N, Kp, mt, nt, kps, es);
            hipLaunchKernelGGL(splitk_reduce_kernel<EP>, dim3((M * N + 255) / 256), dim3(256), 0, stream, slab, slices,
                               (size_t)M * ldslab, M, N, ldslab, ep);
        }
        return hip_status();
    }
    constexpr size_t lds = gemm_lds_bytes(64, 64);
    if (slices == 1) {
        hipLaunchKernelGGL((gemm_f32_kernel<64, 64, AL, EP>), dim3(grid, 1), dim3(256), lds, stream, al, Bt, M, N, Kp, mt, nt, KT, ep);
    } else {
        EpSlab es{slab, ldslab, (size_t)M * ldslab};
        hipLaunchKernelGGL((gemm_f32_kernel<64, 64, AL, EpSlab>), dim3(grid, slices), dim3(256), lds, stream, al, Bt, M, N, Kp, mt, nt, kps, es);
        hipLaunchKernelGGL(splitk_reduce_kernel<EP>, dim3((M * N + 255) / 256), dim3(256), 0, stream, slab, slices,
                           (size_t)M * ldslab, M, N, ldslab, ep);
    }
    return hip_status();
}

inline int tnh_target_wgs() { static const int v = getenv("UU3D_TNH_WGS") ? atoi(getenv("UU3D_TNH_WGS")) : 192; return v; }     // these GEMMs share the chip with the activation-gradient chain: fewer, longer workgroups and half the combine traffic (384: 3.74 ms per step, 192: 3.60, 128: 3.61, 96: 3.65)
// C[P][Q] = A^T B over R rows, split over R into slabs, combined in order.
template <class AL, class EP>
int launch_gemm_tn(const AL& al, const float* B, int ldb, int R, int P, int Q, const EP& ep, float* slab, size_t slab_floats, hipStream_t stream,
                   bool h3 = false) {
    if (h3 && P >= 128 && Q >= 128) {        // f16x3 on 128 x 128 tiles (gemm_tn_h3_kernel); narrow results stay on the exact-f32 kernel
        const int pt = (P + 127) / 128, qt = (Q + 127) / 128, tiles = pt * qt;
        const int KT = (R + 31) / 32;
        const int ldslab = ru(Q, 4);
        int slices = std::max(1, std::min(KT / 2, (tnh_target_wgs() + tiles - 1) / tiles));
        while (slices > 1 && (size_t)slices * P * ldslab > slab_floats) --slices;
        const int kps = (KT + slices - 1) / slices;
        slices = (KT + kps - 1) / kps;
        auto k1 = gemm_tn_h3_kernel<AL, EP>; auto ks = gemm_tn_h3_kernel<AL, EpSlab>;
        static const bool attr_set = [&] {            // once per instantiation (one device per process)
            (void)hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TNH_LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)ks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TNH_LDS_BYTES);
            return true; }();
        (void)attr_set;
        if (slices == 1) {
            hipLaunchKernelGGL(k1, dim3(tiles, 1), dim3(256), TNH_LDS_BYTES, stream, al, B, ldb, R, P, Q, pt, qt, KT, ep);
        } else {
            EpSlab es{slab, ldslab, (size_t)P * ldslab};
            hipLaunchKernelGGL(ks, dim3(tiles, slices), dim3(256), TNH_LDS_BYTES, stream, al, B, ldb, R, P, Q, pt, qt, kps, es);
            hipLaunchKernelGGL(splitk_reduce_kernel<EP>, dim3((P * Q + 255) / 256), dim3(256), 0, stream, slab, slices,
                               (size_t)P * ldslab, P, Q, ldslab, ep);
        }
        return hip_status();
    }
    const int pt = (P + 63) / 64, qt = (Q + 63) / 64, tiles = pt * qt;
    const int KT = (R + 31) / 32;
    // <= 48 slabs when the combine is one thread per element; tall-skinny results (<= 4 tiles, e.g. the spatial stack's
    // 32 x 32 weight gradients over 77k rows) take up to 256 slabs and the 16-lane combine instead (512 slabs: +1.4 % per training step)
    const bool skinny = tiles <= 4 && KT >= 256;
    int slices = skinny ? std::max(1, std::min(KT / 4, 256 / tiles))
                        : std::max(1, std::min(std::min(std::max(KT / 4, 1), 48), (1024 + tiles / 2) / tiles));
    const int ldslab = ru(Q, 4);
    while (slices > 1 && (size_t)slices * P * ldslab > slab_floats) --slices;
    int kps = (KT + slices - 1) / slices;
    slices = (KT + kps - 1) / kps;
    if (slices == 1) {
        hipLaunchKernelGGL((gemm_tn_kernel<AL, EP>), dim3(tiles, 1), dim3(256), 0, stream, al, B, ldb, R, P, Q, pt, qt, KT, ep);
    } else {
        EpSlab es{slab, ldslab, (size_t)P * ldslab};
        hipLaunchKernelGGL((gemm_tn_kernel<AL, EpSlab>), dim3(tiles, slices), dim3(256), 0, stream, al, B, ldb, R, P, Q, pt, qt, kps, es);
        if (skinny)
            hipLaunchKernelGGL(splitk_reduce16_kernel<EP>, dim3((P * Q + 15) / 16), dim3(256), 0, stream, slab, slices,
                               (size_t)P * ldslab, P, Q, ldslab, ep);
        else
            hipLaunchKernelGGL(splitk_reduce_kernel<EP>, dim3((P * Q + 255) / 256), dim3(256), 0, stream, slab, slices,
                               (size_t)P * ldslab, P, Q, ldslab, ep);
    }
    return hip_status();
}

inline int launch_colsum(const float* x, int ldx, int R, int C, int period, const uint8_t* mask, int want, const ReduceOut out,
                         int accumulate, float* scratch, size_t scratch_floats, hipStream_t stream) {
    const int P = period > 0 ? period : 1;
    int slices = std::max(1, std::min(256, R / std::max(128, P)));      // >= 128 rows (and one period) per workgroup
    while (slices > 1 && (size_t)slices * P * C > scratch_floats) --slices;
    if ((size_t)slices * P * C > scratch_floats) return UU3D_ERR_WORKSPACE;
    if (period > 0 && mask == nullptr && (C % 4) == 0 && (ldx % 4) == 0 && R % period == 0) {
        const int nb = R / period, xb = (period * (C / 4) + 255) / 256;
        slices = std::max(1, std::min(std::max(1, 512 / xb), nb / 4));         // ~512 workgroups, >= 4 samples per thread
        while (slices > 1 && (size_t)slices * P * C > scratch_floats) --slices;
        if ((size_t)slices * P * C > scratch_floats) return UU3D_ERR_WORKSPACE;
        hipLaunchKernelGGL(colsum_period4_kernel, dim3(xb, slices), dim3(256), 0, stream, x, ldx, nb, period, C, scratch, slices);
    } else
    if (period <= 0 && mask == nullptr && (C % 4) == 0 && (ldx % 4) == 0) {
        const int cgs = std::max(1, std::min(64, (C + 3) / 4));
        const int xb = (C + 4 * cgs - 1) / (4 * cgs);
        slices = std::max(1, std::min(std::max(1, 320 / xb), R / 64));    // ~one workgroup per CU, >= 64 rows each; more slabs only move the time into the combine
        while (slices > 1 && (size_t)slices * C > scratch_floats) --slices;
        hipLaunchKernelGGL(colsum4_kernel, dim3(xb, slices), dim3(256), 0, stream, x, ldx, R, C, scratch, slices, cgs);
    } else
    hipLaunchKernelGGL(colsum_kernel, dim3((C + 63) / 64, slices), dim3(256), 0, stream, x, ldx, R, C, period, mask, want, scratch, slices);
    launch_reduce_partials(scratch, P * C, (size_t)P * C, slices, out, accumulate, stream);
    return hip_status();
}
inline int launch_colsum(const float* x, int ldx, int R, int C, int period, const uint8_t* mask, int want, float* out,
                         int accumulate, float* scratch, size_t scratch_floats, hipStream_t stream) {
    const int n = (period > 0 ? period : 1) * C;
    return launch_colsum(x, ldx, R, C, period, mask, want, ReduceOut{out, nullptr, nullptr, n > 0 ? n : 1}, accumulate, scratch, scratch_floats, stream);
}

// LayerNorm backward in two launches: the row kernel (dx, per-workgroup partial dgamma / dbeta in scratch; returns the number of
// partial slices) and the combine of the partials, which may run on another stream behind the first.
// accumulate: dx = res + d x (res == nullptr: dx itself, in place).
inline int launch_ln_bwd_rows(const float* x, const float* dy, const float2* stats, const float* gamma, int ld, int D, int M, float* dx,
                              int accumulate, const float* res, float* scratch, size_t scratch_floats, hipStream_t stream,
                              const LnBwdGated gated = LnBwdGated{nullptr, 1.f, 1, nullptr}) {
    if (!res) res = dx;
    // wide rows (D = 384): >= 500 workgroups at M = 4544 (8 rows per wave left 114 of 256 CUs idle), <= 2048 partial rows;
    // the spatial stack's D = 32 rows are cheap and many (77 k): fewer, longer workgroups keep the combine short
    if ((D == 32 || D == 64) && ld % 4 == 0) {          // narrow rows: D / 4 lanes per row, 256 * 4 / D rows per workgroup at once
        const int RG = 1024 / D;
        int rpg = std::max(4, (M + RG * 512 - 1) / (RG * 512));
        int wgs = (M + RG * rpg - 1) / (RG * rpg);
        while ((size_t)wgs * 2 * D > scratch_floats) { rpg *= 2; wgs = (M + RG * rpg - 1) / (RG * rpg); }
        if (D == 32) hipLaunchKernelGGL(ln_bwd_narrow_kernel<8>, dim3(wgs), dim3(256), 0, stream, x, dy, stats, gamma, ld, M, rpg, dx, res, accumulate, scratch, gated);
        else hipLaunchKernelGGL(ln_bwd_narrow_kernel<16>, dim3(wgs), dim3(256), 0, stream, x, dy, stats, gamma, ld, M, rpg, dx, res, accumulate, scratch, gated);
        return wgs;
    }
    int rpw = (D > 64) ? std::max(2, (M + 4 * 2048 - 1) / (4 * 2048)) : std::max(8, (M + 4 * 512 - 1) / (4 * 512));
    int wgs = (M + 4 * rpw - 1) / (4 * rpw);
    while ((size_t)wgs * 2 * D > scratch_floats) { rpw *= 2; wgs = (M + 4 * rpw - 1) / (4 * rpw); }
    if (D <= 512) hipLaunchKernelGGL(ln_bwd_kernel<2>, dim3(wgs), dim3(256), 0, stream, x, dy, stats, gamma, ld, D, M, rpw, dx, res, accumulate, scratch, gated);
    else hipLaunchKernelGGL(ln_bwd_kernel<4>, dim3(wgs), dim3(256), 0, stream, x, dy, stats, gamma, ld, D, M, rpw, dx, res, accumulate, scratch, gated);
    return wgs;
}

// partial layout [slice][2][D]; dgamma/dbeta: written (acc_params == 0) or accumulated
inline void launch_ln_bwd_combine(const float* scratch, int D, int slices, float* dgamma, float* dbeta, int acc_params, hipStream_t stream) {
    if (dbeta == dgamma + D) {      // gamma and beta are neighbours in the flat gradient buffer: one combine over [dgamma | dbeta]
        launch_reduce_partials(scratch, 2 * D, (size_t)2 * D, slices, dgamma, acc_params, stream);
    } else {
        launch_reduce_partials(scratch, D, (size_t)2 * D, slices, dgamma, acc_params, stream);
        launch_reduce_partials(scratch + D, D, (size_t)2 * D, slices, dbeta, acc_params, stream);
    }
}

inline int launch_ln_bwd(const float* x, const float* dy, const float2* stats, const float* gamma, int ld, int D, int M, float* dx,
                         int accumulate, float* dgamma, float* dbeta, int acc_params, float* scratch, size_t scratch_floats, hipStream_t stream) {
    const int slices = launch_ln_bwd_rows(x, dy, stats, gamma, ld, D, M, dx, accumulate, nullptr, scratch, scratch_floats, stream);
    launch_ln_bwd_combine(scratch, D, slices, dgamma, dbeta, acc_params, stream);
    return hip_status();
}

// head dims the generic forward kernel is instantiated for (launch_attn_generic)
inline bool attn_generic_head_dim_ok(int dh) { return dh == 2 || dh == 4 || dh == 8 || dh == 12 || dh == 16 || dh == 24 || dh == 32 || dh == 48 || dh == 64; }

// drop: Dropout on the attention weights (training with ATTENTION_DROP_RATE > 0) -- only the generic kernels implement it, so the
// unrolled / MFMA backward kernels are not taken then
inline int launch_attn_generic(bool backward, const float* qkv, const float* dO, int ld, int D, int B, int L, int H, int dh,
                               const uint8_t* mask, float* out, int ldo, hipStream_t stream, const DropCfg drop = DropCfg{}) {
    const int total = B * H;
    const dim3 block(128);
    if (dh != 4 && dh != 48) {
        // other head dims (generic-dims models, uu3d_create): thread = query row, one (sequence, head) per workgroup
        if (L > 128) return UU3D_ERR_UNSUPPORTED;
        const dim3 grid(total);
#define UU3D_ATTNG_CASE(d) case d: { static bool done##d = false; \
            if (!done##d) { (void)hipFuncSetAttribute((const void*)attn_generic_fwd_kernel<d>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                            (void)hipFuncSetAttribute((const void*)attn_generic_bwd_kernel<d>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); done##d = true; } \
            if (attn_generic_lds_bytes<d>(L, backward) > (size_t)160 * 1024) return UU3D_ERR_UNSUPPORTED; \
            if (backward) hipLaunchKernelGGL(attn_generic_bwd_kernel<d>, grid, block, attn_generic_lds_bytes<d>(L, true), stream, qkv, dO, ld, D, L, H, mask, out, ldo, 1, total, drop); \
            else hipLaunchKernelGGL(attn_generic_fwd_kernel<d>, grid, block, attn_generic_lds_bytes<d>(L, false), stream, qkv, ld, D, L, H, mask, out, ldo, 1, total, drop); } break;
        switch (dh) {
            UU3D_ATTNG_CASE(2) UU3D_ATTNG_CASE(8) UU3D_ATTNG_CASE(12) UU3D_ATTNG_CASE(16) UU3D_ATTNG_CASE(24) UU3D_ATTNG_CASE(32) UU3D_ATTNG_CASE(64)
            default: return UU3D_ERR_UNSUPPORTED;
        }
#undef UU3D_ATTNG_CASE
        return hip_status();
    }
    if (dh == 4) {
        const int pack = std::max(1, 128 / L);                         // (sequence, head) pairs per workgroup
        const dim3 grid((total + pack - 1) / pack);
        const size_t lds = attn_generic_lds_bytes<4>(L, backward) * pack;
        if (backward && L == 17 && mask == nullptr && !drop.on() && !getenv("UU3D_ATTN_BWD_GENERIC"))     // the spatial stack's shape: unrolled, rows in registers
            hipLaunchKernelGGL(attn_small_bwd_kernel<17>, grid, block, attn_small_bwd_lds_bytes<17>() * pack, stream, qkv, dO, ld, D, H, out, ldo, pack, total);
        else if (backward) hipLaunchKernelGGL(attn_generic_bwd_kernel<4>, grid, block, lds, stream, qkv, dO, ld, D, L, H, mask, out, ldo, pack, total, drop);
        else hipLaunchKernelGGL(attn_generic_fwd_kernel<4>, grid, block, lds, stream, qkv, ld, D, L, H, mask, out, ldo, pack, total, drop);
    } else {
        const dim3 grid(total);
        const size_t lds = attn_generic_lds_bytes<48>(L, backward);
        if (backward && L <= 128 && !drop.on() && !getenv("UU3D_ATTN_BWD_GENERIC")) {
            // MFMA backward, tiles in registers (attn_bwd_mfma_kernel); dqkv has the layout (and leading dimension) of qkv
            const int NT = (L + 15) / 16;
            const size_t l2 = attn_bwd_mfma_lds_bytes<48>(NT);
#define UU3D_ATTNB_CASE(nt) case nt: { static bool d2 = false; if (!d2) { (void)hipFuncSetAttribute((const void*)attn_bwd_mfma_kernel<nt, 48>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_bwd_mfma_lds_bytes<48>(nt)); d2 = true; } \
            hipLaunchKernelGGL((attn_bwd_mfma_kernel<nt, 48>), grid, dim3(64 * nt), l2, stream, qkv, dO, ld, D, L, H, mask, out, ldo); } break;
            switch (NT) {
                UU3D_ATTNB_CASE(1) UU3D_ATTNB_CASE(2) UU3D_ATTNB_CASE(3) UU3D_ATTNB_CASE(4)
                UU3D_ATTNB_CASE(5) UU3D_ATTNB_CASE(6) UU3D_ATTNB_CASE(7) UU3D_ATTNB_CASE(8)
            }
#undef UU3D_ATTNB_CASE
        } else if (backward) {
            static bool done = false;
            if (!done) { (void)hipFuncSetAttribute((const void*)attn_generic_bwd_kernel<48>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); done = true; }
            hipLaunchKernelGGL(attn_generic_bwd_kernel<48>, grid, block, lds, stream, qkv, dO, ld, D, L, H, mask, out, ldo, 1, total, drop);
        } else {
            static bool donef = false;
            if (!donef) { (void)hipFuncSetAttribute((const void*)attn_generic_fwd_kernel<48>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); donef = true; }
            hipLaunchKernelGGL(attn_generic_fwd_kernel<48>, grid, block, lds, stream, qkv, ld, D, L, H, mask, out, ldo, 1, total, drop);
        }
    }
    return hip_status();
}

}  // namespace uu3d

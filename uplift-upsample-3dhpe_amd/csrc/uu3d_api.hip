// uu3d_api.hip -- C ABI (include/uu3d.h) over the gfx950 kernels: model object, weight
// inventory / repacking, workspace carving and the forward launch schedule.
//
// Forward schedule (reference: UpliftUpsampleTransformer.call, u_u_t.py:388-421); f16x3 product path at >= 1024 token rows
// (fewer rows / precision f32: the tiled kernels with row_stats in front, same order):
//   compact_frames (mask given) ; spatial_stack  kp2d -> S as f16 planes (B*N, J*d_s)      [u_u_t.py:313-330]
//   gemm  S x W_s2t  (+token blend +PE) -> X (B*N, d_t)                       [:332,344-352]
//   temporal block i (x temporal_depth)                                       [vit.py:176-195]
//     LN1 -> A fragments (ln_split_frag; blocks > 0: ln_res_split_frag, which first adds the previous MLP's three partial slabs
//     + bias into X); row-panel GEMM LN1(X) x Wqkv -> q | k | v planes; attn_h3 -> O planes; gemm O x Wp (+res) -> X;
//     LN2 -> A fragments; mlp_fused (fc1, ReLU, fc2 per hidden slice) -> slabs
//       (the combine after the last block also writes XA = X + strided_pe_1)
//   gemm X x W_head1 -> full_out                                              [:400-404]
//   strided block i (x len(STRIDES)) on XA (B*L_i, d_t)                       [:122-160]
//     LN1 / QKV / attention / projection as above (few rows: gemm_h3_wt_kernel, exact-f32 attention); LN2 x W1 (+relu) -> planes;
//     gemm conv3(Hb) x Wc (+identity gather +bias +strided_pe_{i+1}) -> XB ; swap
//   gemm XA x W_head2 -> central_out                                          [:414-416]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/uu3d.h"
#include "uu3d_gemm.h"
#include "uu3d_gemm_h3.h"
#include "uu3d_gemm_panel.h"
#include <mutex>
#include "uu3d_gemm_panel8.h"
#include "uu3d_gemm_wt.h"
#include "uu3d_mlp_fused.h"
#include "uu3d_tchain16.h"
#include "uu3d_attn.h"
#include "uu3d_attn_h3.h"
#include "uu3d_spatial.h"
#include "uu3d_spatial_h3.h"
#include "uu3d_misc.h"
#include "uu3d_train.h"
#include "uu3d_bwd.h"
#include "uu3d_launch.h"
#include "uu3d_train_kernels.h"

using namespace uu3d;

namespace {

constexpr int kJ = 17, kDS = 32, kHS = 64, kHeads = 8, kDH = 48;
using SLY = SpatialBlockLayout<kDS, kHS>;
using SLY2 = SpatialBlockLayoutV2<kDS, kHS>;
constexpr int kFR = 3;   // frames per wave in the MFMA spatial kernel (3 * 17 = 51 rows)
#ifndef UU3D_SPATIAL_MT
#define UU3D_SPATIAL_MT 1
#endif
constexpr int kSpatialMT = UU3D_SPATIAL_MT;   // token tiles per wave of the f16x3 spatial kernel: 1 = two waves per 3 frames (see uu3d_spatial_h3.h)

std::string g_create_error;

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct WeightRec {
    std::string name;
    std::vector<int64_t> dims;
    int64_t numel = 0;
    std::vector<float> host;
    bool set = false;
};

// Device views of one packed transformer block (temporal or strided).
struct BlockDev {
    const float *ln1_g, *ln1_b, *wqkv_t, *bqkv, *wp_t, *bp, *ln2_g, *ln2_b, *w1_t, *b1, *w2_t, *b2;
    const float* pe;   // strided blocks: (L_i, d_t)
    // fragment-ordered f16 planes of wqkv / w1 for the row-panel GEMM (uu3d_gemm_panel.h); offsets in harena, 0 = none
    size_t wqkv_pf = 0, w1_pf = 0;
    size_t w2_mf = 0;                  // fc2 fragments of the fused MLP kernel (uu3d_mlp_fused.h, temporal blocks), offset in harena, 0 = none
    size_t wp_pf = 0;                  // fragment-ordered projection operand (row-panel GEMM), 0 = none
};

struct ProfRec {
    std::string name, kernel;
    double flops, bytes;
    hipEvent_t e0, e1;
};

}  // namespace

struct uu3d_train_state;
struct uu3d_model {
    uu3d_config cfg;
    uu3d_train_state* ts = nullptr;   // training packs (uu3d_train_init)
    int device = 0;
    std::vector<WeightRec> weights;
    std::map<std::string, int> index;
    std::vector<int> L;            // strided lengths L_0 .. L_ns
    bool committed = false;
    // Dims other than the compiled ones (J = 17, d_s = 32, h_s = 64, 8 heads, d_t = 384): the forward runs on the GENERIC kernels of the
    // training-mode chain (uu3d_train_step.inc: tiled GEMMs with LayerNorm / GELU loaders, attn_generic_fwd_kernel, one launch per layer)
    // from a master buffer the model owns -- correct and an order of magnitude slower than the specialised path; no backward pass.
    bool generic = false;
    float* gparams = nullptr;      // generic: master parameter buffer (uu3d_train_init layout)
    float* arena = nullptr;        // packed device weights
    size_t arena_floats = 0;
    bool no_mlpf = false;          // UU3D_NO_MLPF=1: fc1 and fc2 of the temporal blocks as two GEMM launches (A/B measurements, tests)
    bool no_wt = false;            // UU3D_NO_WT=1: few-row GEMMs stay on the tiled split-K kernels (A/B measurements, tests)
    bool attn_f32 = false;         // UU3D_ATTN_F32=1: sequences of 49-128 tokens stay on the exact-f32 attention kernels (A/B measurements, tests)
    bool attn_wg = false;          // UU3D_ATTN_WG=1: attention with one workgroup per (sequence, head) (attn_f32_kernel) instead of one wave per item (A/B measurements, tests)
    bool no_panel = false;         // UU3D_NO_PANEL=1: LayerNorm-fed GEMMs stay on the tiled kernels (A/B measurements, tests)
    bool no_panel_proj = false;    // UU3D_NO_PANEL_PROJ=1: the attention projection stays on the tiled LDS-DMA kernel
    bool throughput = false;       // uu3d_set_schedule: launches shaped for CU-microseconds (several forwards share the chip) instead of latency
    // Temporal chain (uu3d_tchain16.h; throughput schedule): one launch per temporal block for its row-local stages.  Launch 0 = LayerNorm 1 + QKV of
    // block 1; launch i (1 .. T) = projection + MLP of block i (+ LayerNorm 1 + QKV of the block behind it: temporal block i + 1, or the first strided
    // block with its positional encoding); launch T + 1 = projection + LayerNorm 2 + fc1 of the first strided block.  Empty = not available.
    struct TcLaunch { int flags; size_t w_off /* halfs, harena */; size_t p_off /* floats, arena */; };
    std::vector<TcLaunch> tchain;
    // When the chain runs: under the throughput schedule, whenever the shapes allow it (>= 1024 token rows, attention on attn_h3_kernel, no attention
    // maps asked for).  UU3D_TCHAIN=0 keeps the round-4 launches (A/B measurements, tests), UU3D_TCHAIN_MIN_TILES=n asks for at least n row tiles.
    // History: round 5 (128-row tiles, residual adds as float atomics, profiles/r05_tchain_ab.txt) 179-187 k sequences/s against 171-175 k of the round-4
    // launches at batch 128; round 6 (64-row tiles, everything on chip, profiles/r06_ab_tchain16.txt) 197-212 k on the same box as 179-201 k.
    int tchain_mode = -1;          // -1 by size (tchain_min_tiles), 0 never, 1 always
    int tchain_min_tiles = 8;      // (= the 1024 rows the panel kernels ask for as well)
    bool tchain_short = true;      // the chain (and attn_h3_kernel on its fragment-ordered q | k | v, which cost no split epilogue) also below 49 tokens: h36m_81 (41 tokens),
                                   // batch 256: 337 k -> 361 k sequences/s; UU3D_TCHAIN_SHORT=0: only where attn_h3_kernel is the attention kernel anyway
    int num_cus = 256;
    std::recursive_mutex train_mu; // the training-mode chain keeps per-call options in the handle's training state (uu3d_train_step.inc): one call at a time
    bool in_commit = false;        // uu3d_commit_weights is calling uu3d_train_init (generic dims): the training step's skip flag is not its to clear
    int* d_range = nullptr;        // sticky device word of the range guard (include/uu3d.h: uu3d_range_status)
    bool no_planes = false;        // UU3D_NO_PLANES=1: keep the on-the-fly split GEMMs in f16x3 mode (A/B measurements, tests)
    _Float16* harena = nullptr;    // f16 hi/lo planes of every GEMM operand (f16x3 mode)
    size_t harena_halfs = 0;
    std::map<size_t, std::pair<size_t, size_t>> hplanes;   // Bt float offset -> (hi offset, lo offset) in harena
    std::map<size_t, size_t> panel_off, mlpf_off;          // mlpf_off: Bt float offset of fc2 -> mlpf_pack_w2 fragments in harena                    // Bt float offset -> fragment-ordered planes in harena (uu3d_gemm_panel.h)
    // packed views
    SpatialParams sp{};
    const float* sp_blocks_v1 = nullptr;   // VALU kernel layout (kept for A/B runs: UU3D_SPATIAL=valu)
    const float* sp_blocks_v2 = nullptr;   // MFMA kernel layout
    bool spatial_valu = false;
    bool spatial_f32 = false;      // UU3D_SPATIAL=f32: exact-f32 MFMA spatial stack even in f16x3 mode
    bool spatial_h3_always = false;   // UU3D_SPATIAL=h3: f16x3 spatial stack for every launch size
    size_t sp_frag_off = 0;        // offset (halfs) of the spatial f16 fragment planes in harena
    const float *s2t_wt = nullptr, *s2t_b = nullptr, *token = nullptr, *pe_t = nullptr;
    std::vector<BlockDev> tblocks, sblocks;
    const float *h1_wt = nullptr, *h1_b = nullptr, *h2_wt = nullptr, *h2_b = nullptr;
    // profiling
    bool profiling = false;
    std::vector<ProfRec> prof;
    size_t prof_used = 0;
    std::string err;
};

namespace {

// (a property of the configuration, not of the commit state: uu3d_workspace_bytes may be asked before the weights are committed)
inline bool tchain_possible(const uu3d_config& c) {
    return c.precision == UU3D_PREC_F16X3 && c.d_temporal == 384 && c.h_temporal == 768 && c.num_heads == 8 && c.temporal_depth >= 1;
}

int fail(uu3d_model* m, int code, const std::string& msg) {
    if (m) m->err = msg; else g_create_error = msg;
    return code;
}

#define HIPCHK(m, expr)                                                                         \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return fail(m, UU3D_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));   \
    } while (0)

void add_weight(uu3d_model* m, const std::string& name, std::vector<int64_t> dims) {
    WeightRec r;
    r.name = name;
    r.dims = dims;
    r.numel = 1;
    for (auto d : dims) r.numel *= d;
    m->index[name] = (int)m->weights.size();
    m->weights.push_back(std::move(r));
}

void add_block(uu3d_model* m, const std::string& p, int d, int h, bool strided, bool qkv_bias) {
    add_weight(m, p + "/norm1/gamma", {d});
    add_weight(m, p + "/norm1/beta", {d});
    for (const char* nm : {"wq", "wk", "wv"}) {
        add_weight(m, p + "/attn/" + nm + "/kernel", {d, d});
        if (qkv_bias) add_weight(m, p + "/attn/" + nm + "/bias", {d});
    }
    add_weight(m, p + "/attn/projection/kernel", {d, d});
    add_weight(m, p + "/attn/projection/bias", {d});
    add_weight(m, p + "/norm2/gamma", {d});
    add_weight(m, p + "/norm2/beta", {d});
    if (strided) {
        add_weight(m, p + "/mlp/fc1/kernel", {1, d, h});
        add_weight(m, p + "/mlp/fc1/bias", {h});
        add_weight(m, p + "/mlp/strided_conv/kernel", {3, h, d});
        add_weight(m, p + "/mlp/strided_conv/bias", {d});
    } else {
        add_weight(m, p + "/mlp/fc1/kernel", {d, h});
        add_weight(m, p + "/mlp/fc1/bias", {h});
        add_weight(m, p + "/mlp/fc2/kernel", {h, d});
        add_weight(m, p + "/mlp/fc2/bias", {d});
    }
}

// Creation order of the reference's __init__ (u_u_t.py:196-285) = model.weights order.
void build_inventory(uu3d_model* m) {
    const uu3d_config& c = m->cfg;
    const int J = c.num_keypoints, N = c.num_frames, ds = c.d_spatial, dt = c.d_temporal;
    if (c.spatial_depth > 0) {
        add_weight(m, "keypoint_embedding/kernel", {2, ds});
        add_weight(m, "keypoint_embedding/bias", {ds});
        add_weight(m, "spatial_pe/positional_encoding_weights", {J, ds});
    }
    add_weight(m, "temporal_pe/positional_encoding_weights", {N, dt});
    for (int i = 0; i < c.num_strided; ++i)
        add_weight(m, "strided_temporal_pe_" + std::to_string(i + 1) + "/positional_encoding_weights", {m->L[i], dt});
    if (c.learnable_masked_token) add_weight(m, "learnable_masked_token_layer/learnable_masked_token", {dt});      // (u_u_t.py:219-220, in front of the strided-input token)
    if (c.has_strided_input) add_weight(m, "strided_input_token_layer/learnable_masked_token", {dt});
    if (c.spatial_depth > 0) {
        for (int i = 0; i < c.spatial_depth; ++i)
            add_block(m, "spatial_block_" + std::to_string(i + 1), ds, c.h_spatial, false, c.qkv_bias != 0);
        add_weight(m, "spatial_norm/gamma", {ds});
        add_weight(m, "spatial_norm/beta", {ds});
    }
    add_weight(m, "spatial_to_temporal_fc/kernel", {(int64_t)J * ds, dt});
    add_weight(m, "spatial_to_temporal_fc/bias", {dt});
    for (int i = 0; i < c.temporal_depth; ++i)
        add_block(m, "temporal_block_" + std::to_string(i + 1), dt, c.h_temporal, false, c.qkv_bias != 0);
    for (int i = 0; i < c.num_strided; ++i)
        add_block(m, "strided_temporal_block_" + std::to_string(i + 1), dt, c.h_temporal, true, c.qkv_bias != 0);
    // Keras' model.weights = trainable weights in creation order, then the non-trainable ones (the BatchNorm moving statistics)
    if (c.full_output && c.temporal_depth > 0) {
        if (c.output_bn) { add_weight(m, "temporal_norm/gamma", {dt}); add_weight(m, "temporal_norm/beta", {dt}); }
        add_weight(m, "temporal_fc/kernel", {dt, 3 * J});
        add_weight(m, "temporal_fc/bias", {3 * J});
    }
    if (c.output_bn) { add_weight(m, "strided_temporal_norm/gamma", {dt}); add_weight(m, "strided_temporal_norm/beta", {dt}); }
    add_weight(m, "strided_temporal_fc/kernel", {dt, 3 * J});
    add_weight(m, "strided_temporal_fc/bias", {3 * J});
    if (c.output_bn) {
        if (c.full_output && c.temporal_depth > 0) { add_weight(m, "temporal_norm/moving_mean", {dt}); add_weight(m, "temporal_norm/moving_variance", {dt}); }
        add_weight(m, "strided_temporal_norm/moving_mean", {dt}); add_weight(m, "strided_temporal_norm/moving_variance", {dt});
    }
}

// ---- host-side packing ------------------------------------------------------------------
struct Packer {
    std::vector<float> buf;
    std::vector<std::pair<size_t, size_t>> dense;  // (offset, floats) of every GEMM operand Bt[Np][Kp]
    size_t alloc(size_t n) {                       // 256-byte aligned segments
        size_t off = align_up(buf.size(), 64);
        buf.resize(off + n, 0.f);
        return off;
    }
    size_t alloc_dense(size_t n) { const size_t off = alloc(n); dense.emplace_back(off, n); return off; }
};

const float* W(const uu3d_model* m, const std::string& name) {
    auto it = m->index.find(name);
    return it == m->index.end() ? nullptr : m->weights[it->second].host.data();
}

// Keras Dense kernel (K, N) -> Bt[Np][Kp], Np = round_up(N,128), Kp = round_up(K,32); row offset n0.
void pack_dense_t(std::vector<float>& buf, size_t off, const float* w, int K, int N, int Kp, int n0) {
    for (int k = 0; k < K; ++k)
        for (int n = 0; n < N; ++n) buf[off + (size_t)(n0 + n) * Kp + k] = w[(size_t)k * N + n];
}

}  // namespace

// =========================================================================================
// Definitions below take C linkage from their extern "C" declarations in include/uu3d.h.

#ifdef UU3D_TIMING_BUILD
const char* uu3d_version(void) { return "uu3d 0.3.0 gfx950 f16x3+f32-mfma timing-build (UU3D_SKIP / UU3D_TIMING_PARTS honoured: results can be wrong)"; }
#else
const char* uu3d_version(void) { return "uu3d 0.3.0 gfx950 f16x3+f32-mfma"; }
#endif

const char* uu3d_status_string(int s) {
    switch (s) {
        case UU3D_OK: return "ok";
        case UU3D_ERR_INVALID_ARGUMENT: return "invalid argument";
        case UU3D_ERR_UNSUPPORTED: return "configuration not supported by the compiled kernels";
        case UU3D_ERR_SHAPE: return "shape mismatch";
        case UU3D_ERR_NOT_READY: return "weights not set/committed";
        case UU3D_ERR_WORKSPACE: return "workspace too small or misaligned";
        case UU3D_ERR_HIP: return "HIP runtime error";
        case UU3D_ERR_NO_DEVICE: return "no usable device";
        case UU3D_ERR_RANGE: return "values beyond the f16 range of the f16x3 products (or non-finite inputs)";
        default: return "unknown status";
    }
}

const char* uu3d_last_error(const uu3d_model* model) {
    return model ? model->err.c_str() : g_create_error.c_str();
}

int uu3d_create(const uu3d_config* c, int device, uu3d_model** out) {
    if (!c || !out) return fail(nullptr, UU3D_ERR_INVALID_ARGUMENT, "null config/out pointer");
    *out = nullptr;
    if (c->num_frames < 1 || c->num_keypoints < 1 || c->num_strided < 0 || c->num_strided > UU3D_MAX_STRIDED)
        return fail(nullptr, UU3D_ERR_INVALID_ARGUMENT, "bad num_frames/num_keypoints/num_strided");
    if (c->precision != UU3D_PREC_F32 && c->precision != UU3D_PREC_F16X3) return fail(nullptr, UU3D_ERR_UNSUPPORTED, "unknown precision");
    if (c->spatial_depth < 1) return fail(nullptr, UU3D_ERR_UNSUPPORTED, "spatial_depth must be >= 1");
    if (c->temporal_depth < 0) return fail(nullptr, UU3D_ERR_INVALID_ARGUMENT, "temporal_depth < 0");
    // the specialised kernels (all shipped configs): J = 17, d_s = 32, h_s = 64, 8 heads, d_t = 384.  Anything else the reference's constructor
    // accepts (SPATIAL_EMBED_DIM / TEMPORAL_EMBED_DIM / NUM_HEADS / MLP ratios, u_u_t_constructor.py:26-32) runs on the generic forward.
    const bool generic = c->num_keypoints != kJ || c->d_spatial != kDS || c->h_spatial != kHS || c->num_heads != kHeads || c->d_temporal != kDH * kHeads;
    if (generic) {
        if (c->num_heads < 1 || c->d_spatial < 1 || c->d_temporal < 1 || c->h_spatial < 1 || c->h_temporal < 1)
            return fail(nullptr, UU3D_ERR_INVALID_ARGUMENT, "dims and head count must be positive");
        if (c->d_spatial % c->num_heads != 0 || c->d_temporal % c->num_heads != 0)
            return fail(nullptr, UU3D_ERR_INVALID_ARGUMENT, "embed dims must be divisible by NUM_HEADS (vision_transformer.py:79)");
        if (!attn_generic_head_dim_ok(c->d_spatial / c->num_heads) || !attn_generic_head_dim_ok(c->d_temporal / c->num_heads))
            return fail(nullptr, UU3D_ERR_UNSUPPORTED, "generic forward: head dims (embed dim / NUM_HEADS) must be one of 2, 4, 8, 12, 16, 24, 32, 48, 64");
        if (c->d_spatial % 4 != 0 || c->d_temporal % 4 != 0 || c->h_spatial % 4 != 0)
            return fail(nullptr, UU3D_ERR_UNSUPPORTED, "generic forward: embed dims and MLP widths must be multiples of 4 (16-byte row pieces in every loader)");
        if (c->num_keypoints > 128 || c->num_frames > 128)
            return fail(nullptr, UU3D_ERR_UNSUPPORTED, "generic forward: at most 128 keypoints and 128 frames");
        if (c->temporal_depth < 1 || c->num_strided < 1 || !c->full_output)
            return fail(nullptr, UU3D_ERR_UNSUPPORTED, "generic forward: needs at least one temporal block, one strided block and the full-sequence head");
    }
    if (c->h_temporal % 4 != 0 || c->h_temporal < 4)
        return fail(nullptr, UU3D_ERR_UNSUPPORTED, "h_temporal must be a positive multiple of 4");
    // temporal_depth == 0 (u_u_t.py:356,372-380): the strided blocks follow the token blend directly and the FIRST one takes the key
    // mask; the reference hands the same (B, 1, 1, N) mask to every strided block below FIRST_STRIDED_TOKEN_ATTENTION_LAYER, which only
    // has the right shape for the first (N keys)
    if (c->temporal_depth == 0 && c->has_strided_input && c->num_strided > 0 && c->first_strided_token_attention_layer > 1)
        return fail(nullptr, UU3D_ERR_INVALID_ARGUMENT, "temporal_depth == 0 with FIRST_STRIDED_TOKEN_ATTENTION_LAYER > 1: the reference's key mask has N entries, strided block 2 has fewer keys");
    if (c->d_temporal > 64 * 4 * 4)
        return fail(nullptr, UU3D_ERR_UNSUPPORTED, "d_temporal > 1024");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return fail(nullptr, UU3D_ERR_NO_DEVICE, "HIP device not available");

    auto* m = new uu3d_model();
    m->cfg = *c;
    m->device = device;
    m->generic = generic;
    m->L.push_back(c->num_frames);
    for (int i = 0; i < c->num_strided; ++i) {
        if (c->strides[i] < 1 || c->pad_left[i] < 0 || c->pad_right[i] < 0) {
            delete m;
            return fail(nullptr, UU3D_ERR_INVALID_ARGUMENT, "bad stride/padding");
        }
        const int Lin = m->L.back();
        const int Lout = (Lin + c->pad_left[i] + c->pad_right[i] - 3) / c->strides[i] + 1;
        if (Lin + c->pad_left[i] + c->pad_right[i] < 3 || Lout < 1) {
            delete m;
            return fail(nullptr, UU3D_ERR_INVALID_ARGUMENT, "strided block reduces the sequence below one token");
        }
        // exact-f32 attention holds a query tile's logits in registers (<= 128 keys); the f16x3 kernel tiles the keys
        if (Lin > (c->precision == UU3D_PREC_F16X3 ? ATTN_H3_MAX_L : 128)) {
            delete m;
            return fail(nullptr, UU3D_ERR_UNSUPPORTED, c->precision == UU3D_PREC_F16X3 ? "sequence length > 416 tokens" : "sequence length > 128 tokens with precision f32");
        }
        m->L.push_back(Lout);
    }
    if (c->num_strided > 0 && m->L.back() != 1) {
        delete m;   // einops "b n (p c) -> (b n) p c", n=1 fails in the reference (u_u_t.py:416)
        return fail(nullptr, UU3D_ERR_INVALID_ARGUMENT, "STRIDES/PADDINGS must reduce the sequence to one token");
    }
    build_inventory(m);
    { const char* e = getenv("UU3D_SPATIAL"); m->spatial_valu = (e != nullptr && std::string(e) == "valu");
      m->spatial_f32 = (e != nullptr && std::string(e) == "f32"); m->spatial_h3_always = (e != nullptr && std::string(e) == "h3"); }
    { const char* e = getenv("UU3D_NO_PLANES"); m->no_planes = (e != nullptr && e[0] == '1'); }
    { const char* e = getenv("UU3D_ATTN_WG"); m->attn_wg = (e != nullptr && e[0] == '1'); }
    { const char* e = getenv("UU3D_NO_MLPF"); m->no_mlpf = (e != nullptr && e[0] == '1'); }
    { const char* e = getenv("UU3D_NO_WT"); m->no_wt = (e != nullptr && e[0] == '1'); }
    { const char* e = getenv("UU3D_ATTN_F32"); m->attn_f32 = (e != nullptr && e[0] == '1'); }
    { const char* e = getenv("UU3D_NO_PANEL"); m->no_panel = (e != nullptr && e[0] == '1'); }
    { const char* e = getenv("UU3D_NO_PANEL_PROJ"); m->no_panel_proj = (e != nullptr && e[0] == '1'); }
    { const char* e = getenv("UU3D_TCHAIN"); if (e != nullptr && (e[0] == '0' || e[0] == '1')) m->tchain_mode = e[0] - '0'; }
    { const char* e = getenv("UU3D_TCHAIN_MIN_TILES"); if (e != nullptr && atoi(e) > 0) m->tchain_min_tiles = atoi(e); }
    { const char* e = getenv("UU3D_TCHAIN_SHORT"); if (e != nullptr) m->tchain_short = atoi(e) != 0; }
    { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device) == hipSuccess && pr.multiProcessorCount > 0) m->num_cus = pr.multiProcessorCount; }
    // (the handle's device, not the caller's current one: every later call of the library sets it as well)
    if (hipSetDevice(device) != hipSuccess) {
        delete m;
        return fail(nullptr, UU3D_ERR_HIP, "hipSetDevice failed");
    }
    if (hipMalloc((void**)&m->d_range, 2 * sizeof(int)) != hipSuccess || hipMemset(m->d_range, 0, 2 * sizeof(int)) != hipSuccess) {      // [0] the sticky word, [1] what uu3d_range_status took
        delete m;
        return fail(nullptr, UU3D_ERR_HIP, "hipMalloc of the range-guard word failed");
    }
    *out = m;
    return UU3D_OK;
}

static void train_free(uu3d_model* m);
static int generic_forward(uu3d_model* m, const float* kp2d, const uint8_t* mask, int32_t B, float* full_out, float* central_out,
                           float* const* attn_out, void* workspace, size_t workspace_bytes, void* stream);      // uu3d_train_step.inc
void uu3d_destroy(uu3d_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    train_free(m);
    if (m->gparams) (void)hipFree(m->gparams);
    if (m->arena) (void)hipFree(m->arena);
    if (m->harena) (void)hipFree(m->harena);
    if (m->d_range) (void)hipFree(m->d_range);
    for (auto& r : m->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    delete m;
}

int uu3d_num_weights(const uu3d_model* m) { return m ? (int)m->weights.size() : 0; }

int uu3d_weight_info(const uu3d_model* m, int i, const char** name, int32_t* ndim, int64_t dims[4]) {
    if (!m || i < 0 || i >= (int)m->weights.size()) return UU3D_ERR_INVALID_ARGUMENT;
    const WeightRec& r = m->weights[i];
    if (name) *name = r.name.c_str();
    if (ndim) *ndim = (int32_t)r.dims.size();
    if (dims) for (size_t k = 0; k < 4; ++k) dims[k] = k < r.dims.size() ? r.dims[k] : 1;
    return UU3D_OK;
}

int uu3d_set_weight(uu3d_model* m, const char* name, const float* data, int64_t numel) {
    if (!m || !name || !data) return fail(m, UU3D_ERR_INVALID_ARGUMENT, "null argument to uu3d_set_weight");
    auto it = m->index.find(name);
    if (it == m->index.end()) return fail(m, UU3D_ERR_INVALID_ARGUMENT, std::string("unknown weight: ") + name);
    WeightRec& r = m->weights[it->second];
    if (numel != r.numel)
        return fail(m, UU3D_ERR_SHAPE, std::string("element count mismatch for ") + name + ": got " +
                                           std::to_string(numel) + ", expected " + std::to_string(r.numel));
    r.host.assign(data, data + numel);
    r.set = true;
    m->committed = false;
    return UU3D_OK;
}

int uu3d_get_weight(const uu3d_model* mc, const char* name, float* out, int64_t numel) {
    auto* m = const_cast<uu3d_model*>(mc);
    if (!m || !name || !out) return fail(m, UU3D_ERR_INVALID_ARGUMENT, "null argument to uu3d_get_weight");
    auto it = m->index.find(name);
    if (it == m->index.end()) return fail(m, UU3D_ERR_INVALID_ARGUMENT, std::string("unknown weight: ") + name);
    const WeightRec& r = m->weights[it->second];
    if (!r.set) return fail(m, UU3D_ERR_NOT_READY, std::string("weight never set: ") + name);
    if (numel != r.numel) return fail(m, UU3D_ERR_SHAPE, std::string("element count mismatch for ") + name);
    std::memcpy(out, r.host.data(), sizeof(float) * (size_t)numel);
    return UU3D_OK;
}

int uu3d_commit_weights(uu3d_model* m, void* stream_) {
    if (!m) return UU3D_ERR_INVALID_ARGUMENT;
    for (auto& r : m->weights)
        if (!r.set) return fail(m, UU3D_ERR_NOT_READY, "weight never set: " + r.name);
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(m, hipSetDevice(m->device));
    if (m->generic) {
        // generic dims: the forward reads the master buffer and the operand packs of the training-mode chain (uu3d_train_init; called again it
        // re-uploads the weights and repacks)
        if (m->gparams == nullptr) {
            long long n = 0;
            for (auto& r : m->weights) n += r.numel;
            HIPCHK(m, hipMalloc((void**)&m->gparams, (size_t)n * sizeof(float)));
        }
        m->in_commit = true;
        const int r = uu3d_train_init(m, m->gparams, stream_);
        m->in_commit = false;
        if (r != UU3D_OK) return r;
        HIPCHK(m, hipStreamSynchronize(stream));
        m->committed = true;
        return UU3D_OK;
    }
    const uu3d_config& c = m->cfg;
    const int J = c.num_keypoints, N = c.num_frames, ds = c.d_spatial, dt = c.d_temporal, ht = c.h_temporal;
    const int Kdt = round_up(dt, 32), Kht = round_up(ht, 32), Ks2t = round_up(J * ds, 32);
    Packer P;

    // ---- spatial ----
    const size_t o_ew = P.alloc(2 * ds), o_eb = P.alloc(ds), o_spe = P.alloc((size_t)J * ds);
    std::copy_n(W(m, "keypoint_embedding/kernel"), 2 * ds, P.buf.begin() + o_ew);
    std::copy_n(W(m, "keypoint_embedding/bias"), ds, P.buf.begin() + o_eb);
    std::copy_n(W(m, "spatial_pe/positional_encoding_weights"), J * ds, P.buf.begin() + o_spe);
    const size_t o_sblk = P.alloc((size_t)c.spatial_depth * SLY::size);
    for (int i = 0; i < c.spatial_depth; ++i) {
        const std::string p = "spatial_block_" + std::to_string(i + 1);
        float* d = P.buf.data() + o_sblk + (size_t)i * SLY::size;
        auto vec = [&](int off, const std::string& nm, int n) {
            const float* s = W(m, p + nm);
            if (s) std::copy_n(s, n, d + off);   // absent (qkv_bias false) stays zero
        };
        auto tr = [&](int off, const std::string& nm, int K, int Nn) {   // (K,N) -> [N][K]
            const float* s = W(m, p + nm);
            for (int k = 0; k < K; ++k) for (int n = 0; n < Nn; ++n) d[off + n * K + k] = s[k * Nn + n];
        };
        vec(SLY::ln1_g, "/norm1/gamma", ds); vec(SLY::ln1_b, "/norm1/beta", ds);
        tr(SLY::wq_t, "/attn/wq/kernel", ds, ds); vec(SLY::bq, "/attn/wq/bias", ds);
        tr(SLY::wk_t, "/attn/wk/kernel", ds, ds); vec(SLY::bk, "/attn/wk/bias", ds);
        tr(SLY::wv_t, "/attn/wv/kernel", ds, ds); vec(SLY::bv, "/attn/wv/bias", ds);
        vec(SLY::wp, "/attn/projection/kernel", ds * ds); vec(SLY::bp, "/attn/projection/bias", ds);
        vec(SLY::ln2_g, "/norm2/gamma", ds); vec(SLY::ln2_b, "/norm2/beta", ds);
        tr(SLY::w1_t, "/mlp/fc1/kernel", ds, kHS); vec(SLY::b1, "/mlp/fc1/bias", kHS);
        vec(SLY::w2, "/mlp/fc2/kernel", kHS * ds); vec(SLY::b2, "/mlp/fc2/bias", ds);
    }
    const size_t o_sblk2 = P.alloc((size_t)c.spatial_depth * SLY2::size);
    for (int i = 0; i < c.spatial_depth; ++i) {
        const std::string p = "spatial_block_" + std::to_string(i + 1);
        float* d = P.buf.data() + o_sblk2 + (size_t)i * SLY2::size;
        auto vec = [&](int off, const std::string& nm, int n) {
            const float* s = W(m, p + nm);
            if (s) std::copy_n(s, n, d + off);
        };
        // fragment order of the 32x32x2 MFMA B operand: [n-tile][kk][lane][s] = W[8kk + 4(lane>>5) + s][32nt + (lane&31)]
        auto frag = [&](int off, const std::string& nm, int K, int Nn) {
            const float* s = W(m, p + nm);
            for (int nt = 0; nt < Nn / 32; ++nt)
                for (int kk = 0; kk < K / 8; ++kk)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e)
                            d[off + ((nt * (K / 8) + kk) * 64 + lane) * 4 + e] =
                                s[(size_t)(8 * kk + 4 * (lane >> 5) + e) * Nn + 32 * nt + (lane & 31)];
        };
        vec(SLY2::ln1_g, "/norm1/gamma", ds); vec(SLY2::ln1_b, "/norm1/beta", ds);
        vec(SLY2::ln2_g, "/norm2/gamma", ds); vec(SLY2::ln2_b, "/norm2/beta", ds);
        vec(SLY2::bq, "/attn/wq/bias", ds); vec(SLY2::bk, "/attn/wk/bias", ds); vec(SLY2::bv, "/attn/wv/bias", ds);
        vec(SLY2::bp, "/attn/projection/bias", ds); vec(SLY2::b1, "/mlp/fc1/bias", kHS); vec(SLY2::b2, "/mlp/fc2/bias", ds);
        frag(SLY2::fq, "/attn/wq/kernel", ds, ds); frag(SLY2::fk, "/attn/wk/kernel", ds, ds);
        frag(SLY2::fv, "/attn/wv/kernel", ds, ds); frag(SLY2::fp, "/attn/projection/kernel", ds, ds);
        frag(SLY2::f1, "/mlp/fc1/kernel", ds, kHS); frag(SLY2::f2, "/mlp/fc2/kernel", kHS, ds);
    }
    const size_t o_sng = P.alloc(ds), o_snb = P.alloc(ds);
    std::copy_n(W(m, "spatial_norm/gamma"), ds, P.buf.begin() + o_sng);
    std::copy_n(W(m, "spatial_norm/beta"), ds, P.buf.begin() + o_snb);

    // ---- spatial_to_temporal_fc, token, temporal PE ----
    const int Npdt = round_up(dt, 128);
    const size_t o_s2t = P.alloc_dense((size_t)Npdt * Ks2t), o_s2tb = P.alloc(Npdt);
    pack_dense_t(P.buf, o_s2t, W(m, "spatial_to_temporal_fc/kernel"), J * ds, dt, Ks2t, 0);
    std::copy_n(W(m, "spatial_to_temporal_fc/bias"), dt, P.buf.begin() + o_s2tb);
    const size_t o_tok = P.alloc(dt);
    if (c.has_strided_input)
        std::copy_n(W(m, "strided_input_token_layer/learnable_masked_token"), dt, P.buf.begin() + o_tok);
    const size_t o_pet = P.alloc((size_t)N * dt);
    std::copy_n(W(m, "temporal_pe/positional_encoding_weights"), (size_t)N * dt, P.buf.begin() + o_pet);

    // ---- transformer blocks ----
    struct BlockOff { size_t ln1_g, ln1_b, wqkv, bqkv, wp, bp, ln2_g, ln2_b, w1, b1, w2, b2, pe; };
    auto pack_block = [&](const std::string& p, bool strided, int peL, const std::string& pe_name) {
        BlockOff o{};
        o.ln1_g = P.alloc(dt); o.ln1_b = P.alloc(dt);
        std::copy_n(W(m, p + "/norm1/gamma"), dt, P.buf.begin() + o.ln1_g);
        std::copy_n(W(m, p + "/norm1/beta"), dt, P.buf.begin() + o.ln1_b);
        const int Npq = round_up(3 * dt, 128);
        o.wqkv = P.alloc_dense((size_t)Npq * Kdt); o.bqkv = P.alloc(Npq);
        int part = 0;
        for (const char* nm : {"wq", "wk", "wv"}) {
            pack_dense_t(P.buf, o.wqkv, W(m, p + "/attn/" + nm + "/kernel"), dt, dt, Kdt, part * dt);
            const float* b = W(m, p + "/attn/" + nm + "/bias");
            if (b) std::copy_n(b, dt, P.buf.begin() + o.bqkv + part * dt);
            ++part;
        }
        o.wp = P.alloc_dense((size_t)Npdt * Kdt); o.bp = P.alloc(Npdt);
        pack_dense_t(P.buf, o.wp, W(m, p + "/attn/projection/kernel"), dt, dt, Kdt, 0);
        std::copy_n(W(m, p + "/attn/projection/bias"), dt, P.buf.begin() + o.bp);
        o.ln2_g = P.alloc(dt); o.ln2_b = P.alloc(dt);
        std::copy_n(W(m, p + "/norm2/gamma"), dt, P.buf.begin() + o.ln2_g);
        std::copy_n(W(m, p + "/norm2/beta"), dt, P.buf.begin() + o.ln2_b);
        const int Nph = round_up(ht, 128);
        o.w1 = P.alloc_dense((size_t)Nph * Kdt); o.b1 = P.alloc(Nph);
        pack_dense_t(P.buf, o.w1, W(m, p + "/mlp/fc1/kernel"), dt, ht, Kdt, 0);   // Conv1D k=1 (1,dt,ht) has the same flat layout
        std::copy_n(W(m, p + "/mlp/fc1/bias"), ht, P.buf.begin() + o.b1);
        if (strided) {
            const int Kc = round_up(3 * ht, 32);
            o.w2 = P.alloc_dense((size_t)Npdt * Kc); o.b2 = P.alloc(Npdt);
            // Conv1D kernel (3, ht, dt): flat (j*ht + c, n) is exactly a Dense kernel of K = 3*ht
            pack_dense_t(P.buf, o.w2, W(m, p + "/mlp/strided_conv/kernel"), 3 * ht, dt, Kc, 0);
            std::copy_n(W(m, p + "/mlp/strided_conv/bias"), dt, P.buf.begin() + o.b2);
            o.pe = P.alloc((size_t)peL * dt);
            std::copy_n(W(m, pe_name), (size_t)peL * dt, P.buf.begin() + o.pe);
        } else {
            o.w2 = P.alloc_dense((size_t)Npdt * Kht); o.b2 = P.alloc(Npdt);
            pack_dense_t(P.buf, o.w2, W(m, p + "/mlp/fc2/kernel"), ht, dt, Kht, 0);
            std::copy_n(W(m, p + "/mlp/fc2/bias"), dt, P.buf.begin() + o.b2);
        }
        return o;
    };
    std::vector<BlockOff> toff, soff;
    for (int i = 0; i < c.temporal_depth; ++i)
        toff.push_back(pack_block("temporal_block_" + std::to_string(i + 1), false, 0, ""));
    for (int i = 0; i < c.num_strided; ++i)
        soff.push_back(pack_block("strided_temporal_block_" + std::to_string(i + 1), true, m->L[i],
                                  "strided_temporal_pe_" + std::to_string(i + 1) + "/positional_encoding_weights"));

    // ---- heads ----
    const int Nph = round_up(3 * J, 128);
    size_t o_h1 = 0, o_h1b = 0;
    const bool has_h1 = c.full_output && c.temporal_depth > 0;
    // OUTPUT_BN at inference (u_u_t.py:275-285, Keras BatchNormalization with training=False): y = gamma (x - mean) / sqrt(var + eps) + beta
    // is a per-channel affine in front of the Dense head, so it is folded into the head's operands here:
    //   W'[k][n] = s[k] W[k][n],  b'[n] = b[n] + sum_k (beta[k] - mean[k] s[k]) W[k][n],  s = gamma / sqrt(var + 1e-5)
    auto pack_head = [&](size_t o_w, size_t o_b, const std::string& fc, const std::string& bn) {
        const float* Wk = W(m, fc + "/kernel"); const float* bk = W(m, fc + "/bias");
        if (!c.output_bn) {
            pack_dense_t(P.buf, o_w, Wk, dt, 3 * J, Kdt, 0);
            std::copy_n(bk, 3 * J, P.buf.begin() + o_b);
            return;
        }
        const float *g = W(m, bn + "/gamma"), *be = W(m, bn + "/beta"), *mu = W(m, bn + "/moving_mean"), *var = W(m, bn + "/moving_variance");
        std::vector<float> Wf((size_t)dt * 3 * J);
        std::vector<double> bf(3 * J);
        for (int n = 0; n < 3 * J; ++n) bf[n] = bk[n];
        for (int k = 0; k < dt; ++k) {
            const float s = g[k] / std::sqrt(var[k] + 1e-5f);
            const double sh = (double)be[k] - (double)mu[k] * s;
            for (int n = 0; n < 3 * J; ++n) { Wf[(size_t)k * 3 * J + n] = s * Wk[(size_t)k * 3 * J + n]; bf[n] += sh * Wk[(size_t)k * 3 * J + n]; }
        }
        pack_dense_t(P.buf, o_w, Wf.data(), dt, 3 * J, Kdt, 0);
        for (int n = 0; n < 3 * J; ++n) P.buf[o_b + n] = (float)bf[n];
    };
    if (has_h1) {
        o_h1 = P.alloc_dense((size_t)Nph * Kdt); o_h1b = P.alloc(Nph);
        pack_head(o_h1, o_h1b, "temporal_fc", "temporal_norm");
    }
    const size_t o_h2 = P.alloc_dense((size_t)Nph * Kdt), o_h2b = P.alloc(Nph);
    pack_head(o_h2, o_h2b, "strided_temporal_fc", "strided_temporal_norm");

    // ---- temporal chain (uu3d_tchain16.h): per launch one parameter table (floats) and one weight stream (f16 planes, built below) ----
    // A LayerNorm's affine part is folded into the Dense layer behind it: W' = diag(gamma) W, b' = b + beta W (f64 sums).
    const float tc_qscale = 1.44269504088896341f / sqrtf((float)(dt / std::max(1, c.num_heads)));      // = Launcher::attn_qscale(): log2(e) / sqrt(d_h)
    struct TcStage { std::vector<float> Wk; int K, N, kofs; bool natural; };        // Keras layout [K][N]
    struct TcBuild { int flags; size_t p_off; std::vector<TcStage> stages; };
    std::vector<TcBuild> tcb;
    if (tchain_possible(c)) {
        auto block_name = [&](bool strided, int i) { return std::string(strided ? "strided_temporal_block_" : "temporal_block_") + std::to_string(i + 1); };
        auto folded = [&](const std::vector<float>& Wk, const std::vector<float>& b, const float* g, const float* be, int K, int Nn, std::vector<float>& bout) {
            TcStage st{std::vector<float>((size_t)K * Nn), K, Nn, 0, false};
            bout.assign(Nn, 0.f);
            for (int n = 0; n < Nn; ++n) { double acc = b[n]; for (int k = 0; k < K; ++k) acc += (double)be[k] * Wk[(size_t)k * Nn + n]; bout[n] = (float)acc; }
            for (int k = 0; k < K; ++k) for (int n = 0; n < Nn; ++n) st.Wk[(size_t)k * Nn + n] = g[k] * Wk[(size_t)k * Nn + n];
            return st;
        };
        auto qkv_of = [&](const std::string& p, std::vector<float>& Wk, std::vector<float>& b) {       // wq | wk | wv as one (dt, 3 dt) kernel
            Wk.assign((size_t)dt * 3 * dt, 0.f); b.assign(3 * dt, 0.f);
            int part = 0;
            for (const char* nm : {"wq", "wk", "wv"}) {
                const float* w = W(m, p + "/attn/" + nm + "/kernel"); const float* bb = W(m, p + "/attn/" + nm + "/bias");
                for (int k = 0; k < dt; ++k) for (int n = 0; n < dt; ++n) Wk[(size_t)k * 3 * dt + part * dt + n] = w[(size_t)k * dt + n];
                if (bb) std::copy_n(bb, dt, b.begin() + part * dt);
                ++part;
            }
        };
        auto add_qkv = [&](TcBuild& tb, const std::string& p) {
            std::vector<float> Wk, b, bf; qkv_of(p, Wk, b);
            tb.stages.push_back(folded(Wk, b, W(m, p + "/norm1/gamma"), W(m, p + "/norm1/beta"), dt, 3 * dt, bf));
            for (int n = 0; n < dt; ++n) bf[n] *= tc_qscale;          // (q's scale is folded into wq and bq)
            std::copy_n(bf.begin(), 3 * dt, P.buf.begin() + tb.p_off + TCP_BQKV);
        };
        auto add_proj = [&](TcBuild& tb, const std::string& p) {
            const float* w = W(m, p + "/attn/projection/kernel");
            tb.stages.push_back(TcStage{std::vector<float>(w, w + (size_t)dt * dt), dt, dt, 0, true});
            std::copy_n(W(m, p + "/attn/projection/bias"), dt, P.buf.begin() + tb.p_off + TCP_BP);
        };
        auto add_fc1 = [&](TcBuild& tb, const std::string& p) {
            const float* w = W(m, p + "/mlp/fc1/kernel"); const float* b = W(m, p + "/mlp/fc1/bias");
            std::vector<float> bf;
            tb.stages.push_back(folded(std::vector<float>(w, w + (size_t)dt * ht), std::vector<float>(b, b + ht), W(m, p + "/norm2/gamma"), W(m, p + "/norm2/beta"), dt, ht, bf));
            std::copy_n(bf.begin(), ht, P.buf.begin() + tb.p_off + TCP_B1);
        };
        auto add_fc2 = [&](TcBuild& tb, const std::string& p) {
            const float* w = W(m, p + "/mlp/fc2/kernel");
            for (int half = 0; half < 2; ++half) tb.stages.push_back(TcStage{std::vector<float>(w, w + (size_t)ht * dt), ht, dt, half * dt, false});
            std::copy_n(W(m, p + "/mlp/fc2/bias"), dt, P.buf.begin() + tb.p_off + TCP_B2);
        };
        { TcBuild tb{TC_QKV, P.alloc(TCP_FLOATS), {}}; add_qkv(tb, block_name(false, 0)); tcb.push_back(std::move(tb)); }
        for (int i = 0; i < c.temporal_depth; ++i) {
            const bool last = i + 1 == c.temporal_depth;
            TcBuild tb{TC_PROJ | TC_MLP | (last ? (c.num_strided > 0 ? TC_QKV | TC_PE : 0) : TC_QKV), P.alloc(TCP_FLOATS), {}};
            add_proj(tb, block_name(false, i)); add_fc1(tb, block_name(false, i)); add_fc2(tb, block_name(false, i));
            if (!last) add_qkv(tb, block_name(false, i + 1)); else if (c.num_strided > 0) add_qkv(tb, block_name(true, 0));
            tcb.push_back(std::move(tb));
        }
        if (c.num_strided > 0) {
            TcBuild tb{TC_PROJ | TC_FC1_PLANES, P.alloc(TCP_FLOATS), {}};
            add_proj(tb, block_name(true, 0)); add_fc1(tb, block_name(true, 0));
            tcb.push_back(std::move(tb));
        }
    }

    // ---- upload ----
    if (m->arena_floats < P.buf.size()) {
        if (m->arena) HIPCHK(m, hipFree(m->arena));
        m->arena = nullptr;
        HIPCHK(m, hipMalloc((void**)&m->arena, P.buf.size() * sizeof(float)));
        m->arena_floats = P.buf.size();
    }
    HIPCHK(m, hipMemcpyAsync(m->arena, P.buf.data(), P.buf.size() * sizeof(float), hipMemcpyHostToDevice, stream));
    HIPCHK(m, hipStreamSynchronize(stream));
    if (c.precision == UU3D_PREC_F16X3) {
        // split every GEMM operand into f16 hi / (lo * 2048) planes (uu3d_gemm_h3.h)
        std::vector<_Float16> hb(64, (_Float16)0.f);      // [0, 64): zeros, the source of out-of-range conv taps (GLoadConv3)
        m->hplanes.clear();
        for (auto& d : P.dense) {
            const size_t hi = align_up(hb.size(), 64); hb.resize(hi + d.second);
            const size_t lo = align_up(hb.size(), 64); hb.resize(lo + d.second);
            for (size_t i = 0; i < d.second; ++i) {
                const float x = P.buf[d.first + i];
                if (!(std::fabs(x) < 65504.0f))            // (include/uu3d.h, RANGE CONTRACT: the hi plane of such a weight is Inf)
                    return fail(m, UU3D_ERR_RANGE, "a Dense / Conv1D kernel holds a value of magnitude >= 65504 (or a non-finite one): not representable by the f16x3 operand planes; build the model with precision f32");
                const _Float16 h = h3_hi(x);
                hb[hi + i] = h; hb[lo + i] = (_Float16)((x - (float)h) * H3_SCALE);
            }
            m->hplanes[d.first] = {hi, lo};
        }
        {   // spatial stack: A-operand fragments of W^T (uu3d_spatial_h3.h), [n-tile][kk][plane][lane][8]
            using FL = SpatialFragLayoutH3;
            m->sp_frag_off = align_up(hb.size(), 64);
            hb.resize(m->sp_frag_off + (size_t)c.spatial_depth * FL::size);
            for (int i = 0; i < c.spatial_depth; ++i) {
                const std::string p = "spatial_block_" + std::to_string(i + 1);
                _Float16* d = hb.data() + m->sp_frag_off + (size_t)i * FL::size;
                auto frag = [&](int off, const std::string& nm, int K, int Nn) {
                    const float* s = W(m, p + nm);
                    for (int nt = 0; nt < Nn / 32; ++nt)
                        for (int kk = 0; kk < K / 16; ++kk)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int e = 0; e < 8; ++e) {
                                    const float x = s[(size_t)(16 * kk + 8 * (lane >> 5) + e) * Nn + 32 * nt + (lane & 31)];
                                    const _Float16 h = h3_hi(x);
                                    const size_t at = (size_t)off + (((size_t)(nt * (K / 16) + kk) * 2) * 64 + lane) * 8 + e;
                                    d[at] = h;
                                    d[at + 64 * 8] = (_Float16)((x - (float)h) * H3_SCALE);
                                }
                };
                frag(FL::fq, "/attn/wq/kernel", ds, ds); frag(FL::fk, "/attn/wk/kernel", ds, ds);
                frag(FL::fv, "/attn/wv/kernel", ds, ds); frag(FL::fp, "/attn/projection/kernel", ds, ds);
                frag(FL::f1, "/mlp/fc1/kernel", ds, kHS); frag(FL::f2, "/mlp/fc2/kernel", kHS, ds);
            }
        }
        // row-panel GEMM operands (uu3d_gemm_panel.h): wqkv and w1 of every temporal / strided block, fragment ordered
        m->panel_off.clear();
        if (dt % 192 == 0 && ht % 32 == 0) {
            auto add_panel = [&](size_t bt_off, int Nn, int K = 0, int Kp = 0) {
                if (K == 0) { K = dt; Kp = Kdt; }
                const auto it = m->hplanes.find(bt_off);
                if (it == m->hplanes.end() || K % 384 != 0) return;
                const size_t at = align_up(hb.size(), 64);
                hb.resize(at + panel_b_halfs(Nn, K));
                panel_pack_operand(hb.data() + it->second.first, hb.data() + it->second.second, Nn, K, Kp, hb.data() + at);
                m->panel_off[bt_off] = at;
            };
            for (auto& o : toff) { add_panel(o.wqkv, 3 * dt); add_panel(o.w1, ht); add_panel(o.wp, dt); }
            // fused MLP (uu3d_mlp_fused.h): fc2 fragments in the k order fc1's accumulator registers have
            m->mlpf_off.clear();
            if (dt == 32 * MLPF_OC && ht == 256 * MLPF_SLICES)
                for (auto& o : toff) {
                    const auto it = m->hplanes.find(o.w2);
                    if (it == m->hplanes.end()) continue;
                    const size_t at = align_up(hb.size(), 64);
                    hb.resize(at + mlpf_w2_halfs());
                    mlpf_pack_w2(hb.data() + it->second.first, hb.data() + it->second.second, Kht, hb.data() + at);
                    m->mlpf_off[o.w2] = at;
                }
            for (auto& o : soff) { add_panel(o.wqkv, 3 * dt); add_panel(o.w1, ht); add_panel(o.wp, dt); }
        }
        // temporal chain: the launches' weight streams (tchain16_pack_stage: one 48 KiB chunk per 32 output channels and stage)
        m->tchain.clear();
        for (auto& tb : tcb) {
            const size_t at = align_up(hb.size(), 128);
            hb.resize(at + (size_t)tchain_chunks(tb.flags) * TC_CHUNK_HALFS);
            size_t o = at;
            for (auto& st : tb.stages) {
                std::vector<_Float16> Bh((size_t)st.N * st.K), Bl((size_t)st.N * st.K);
                const bool qkv_stage = st.N == 3 * dt;                 // (q's scale lives in wq and bq)
                for (int n = 0; n < st.N; ++n)
                    for (int k = 0; k < st.K; ++k) {
                        const float x = st.Wk[(size_t)k * st.N + n] * (qkv_stage && n < dt ? tc_qscale : 1.0f);
                        if (!(std::fabs(x) < 65504.0f)) return fail(m, UU3D_ERR_RANGE, "a LayerNorm-folded kernel of the temporal chain leaves the f16 range; build the model with precision f32");
                        const _Float16 h = h3_hi(x);
                        Bh[(size_t)n * st.K + k] = h; Bl[(size_t)n * st.K + k] = (_Float16)((x - (float)h) * H3_SCALE);
                    }
                tchain16_pack_stage(Bh.data(), Bl.data(), st.N, st.K, st.kofs, st.natural, hb.data() + o);
                o += (size_t)(st.N / 32) * TC_CHUNK_HALFS;
            }
            if (tb.flags & TC_MLP) {                                // W1 (24 chunks) | W2 half 0 | W2 half 1  ->  W1[0..11] | W2 half 0 | W1[12..23] | W2 half 1
                _Float16* mlp = hb.data() + at + (size_t)((tb.flags & TC_PROJ) ? 12 : 0) * TC_CHUNK_HALFS;
                const std::vector<_Float16> tmp(mlp, mlp + (size_t)48 * TC_CHUNK_HALFS);
                tchain16_reorder_mlp(tmp.data(), mlp);
            }
            m->tchain.push_back({tb.flags, at, tb.p_off});
        }
        if (m->harena_halfs < hb.size()) {
            if (m->harena) HIPCHK(m, hipFree(m->harena));
            m->harena = nullptr;
            HIPCHK(m, hipMalloc((void**)&m->harena, hb.size() * sizeof(_Float16)));
            m->harena_halfs = hb.size();
        }
        HIPCHK(m, hipMemcpyAsync(m->harena, hb.data(), hb.size() * sizeof(_Float16), hipMemcpyHostToDevice, stream));
        HIPCHK(m, hipStreamSynchronize(stream));
    }

    const float* A = m->arena;
    m->sp.embed_w = A + o_ew; m->sp.embed_b = A + o_eb; m->sp.pe = A + o_spe; m->sp.blocks = A + o_sblk;
    m->sp_blocks_v1 = A + o_sblk; m->sp_blocks_v2 = A + o_sblk2;
    m->sp.norm_g = A + o_sng; m->sp.norm_b = A + o_snb; m->sp.depth = c.spatial_depth; m->sp.total_frames = 0;
    m->s2t_wt = A + o_s2t; m->s2t_b = A + o_s2tb; m->token = A + o_tok; m->pe_t = A + o_pet;
    auto view = [&](const BlockOff& o, bool strided) {
        BlockDev b{};
        b.ln1_g = A + o.ln1_g; b.ln1_b = A + o.ln1_b; b.wqkv_t = A + o.wqkv; b.bqkv = A + o.bqkv;
        b.wp_t = A + o.wp; b.bp = A + o.bp; b.ln2_g = A + o.ln2_g; b.ln2_b = A + o.ln2_b;
        b.w1_t = A + o.w1; b.b1 = A + o.b1; b.w2_t = A + o.w2; b.b2 = A + o.b2;
        b.pe = strided ? A + o.pe : nullptr;
        { const auto it = m->panel_off.find(o.wqkv); b.wqkv_pf = (it != m->panel_off.end()) ? it->second : 0; }
        { const auto it = m->panel_off.find(o.w1); b.w1_pf = (it != m->panel_off.end()) ? it->second : 0; }
        { const auto it = m->mlpf_off.find(o.w2); b.w2_mf = (!strided && it != m->mlpf_off.end()) ? it->second : 0; }
        { const auto it = m->panel_off.find(o.wp); b.wp_pf = (it != m->panel_off.end()) ? it->second : 0; }
        return b;
    };
    m->tblocks.clear(); m->sblocks.clear();
    for (auto& o : toff) m->tblocks.push_back(view(o, false));
    for (auto& o : soff) m->sblocks.push_back(view(o, true));
    m->h1_wt = has_h1 ? A + o_h1 : nullptr; m->h1_b = has_h1 ? A + o_h1b : nullptr;
    m->h2_wt = A + o_h2; m->h2_b = A + o_h2b;
    m->committed = true;
    return UU3D_OK;
}

// ---- workspace --------------------------------------------------------------------------
namespace {
struct Workspace {
    float *S, *X, *QKV, *O, *Hb, *XA, *XB, *slab, *mslab;
    unsigned char* tc_scratch;     // temporal chain: the residual tiles between its launches (lane-linear), trash page (tchain16_scratch_bytes)
    int* frame_list;
    float2* stats;
    size_t slab_floats;
    size_t bytes;
};
Workspace carve(const uu3d_model* m, int B, char* base) {
    const uu3d_config& c = m->cfg;
    const size_t rows = (size_t)B * c.num_frames;
    const size_t rows_s = (size_t)B * (c.num_strided > 1 ? std::max(m->L[1], 1) : 1);
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off = align_up(off + n, 256); return o; };
    Workspace w{};
    const size_t oS = take(rows * c.num_keypoints * c.d_spatial * 4);
    const size_t oX = take(rows * c.d_temporal * 4);
    const size_t oQ = take((rows + 128) * 3 * c.d_temporal * 4);   // (+ one 128-row tile: the temporal chain writes q | k | v in whole tiles, fragment ordered)
    const size_t oO = take((rows + 32) * c.d_temporal * 4);   // + one 32-row panel: the panel GEMM's A operand is allocated in whole panels
    const size_t oH = take((rows + 32) * c.h_temporal * 4);   // + one 32-row panel (fragment-ordered hidden planes)
    const size_t oA = take(rows * c.d_temporal * 4);
    const size_t oB = take(rows_s * c.d_temporal * 4);
    const size_t oT = take(rows * sizeof(float2));
    w.slab_floats = (size_t)1536 * 4096;            // >= slices * M * N of any split GEMM (slices * tiles <= ~1150)
    const size_t oSl = take(w.slab_floats * 4);
    const size_t oFl = take((rows + 1) * sizeof(int));
    const size_t oMs = take((size_t)MLPF_SLICES * rows * c.d_temporal * 4);      // fused MLP: fc2 partial sums of the three hidden slices
    const size_t oCh = !tchain_possible(c) ? 0 : take(tchain16_scratch_bytes((int)((rows + 63) / 64)));
    w.bytes = off;
    if (base) {
        w.S = (float*)(base + oS); w.X = (float*)(base + oX); w.QKV = (float*)(base + oQ);
        w.O = (float*)(base + oO); w.Hb = (float*)(base + oH); w.XA = (float*)(base + oA);
        w.XB = (float*)(base + oB); w.stats = (float2*)(base + oT); w.slab = (float*)(base + oSl); w.frame_list = (int*)(base + oFl); w.mslab = (float*)(base + oMs);
        w.tc_scratch = !tchain_possible(c) ? nullptr : (unsigned char*)(base + oCh);
    }
    return w;
}
}  // namespace

size_t uu3d_workspace_bytes(const uu3d_model* m, int32_t batch) {
    if (!m || batch < 1) return 0;
    if (m->generic) return uu3d_train_workspace_bytes(m, batch);     // the generic forward keeps the training chain's activations
    return carve(m, batch, nullptr).bytes;
}

// ---- launch helpers ---------------------------------------------------------------------
namespace {

// Both spatial kernels work on 3 frames per workgroup and are latency bound: a launch takes (rounds of resident workgroups)
// x (one workgroup's run time).  Measured on MI355X, h36m_351 at batch 128 (3030 workgroups): the f16x3 kernel (two waves per
// workgroup, 26 KB LDS: 6 workgroups per CU = 1536 per round) 135 us for its two rounds; the exact-f32 kernel (one wave, 7 per
// CU = 1792 per round) ~133 us per round.  The frame count of the launch decides (an upper bound: masked frames drop out on
// the device).
inline bool spatial_h3_pays(int frames) {
    const int wgs = (frames + kFR - 1) / kFR;
    const int t_h3 = ((wgs + 1535) / 1536) * 68, t_f32 = ((wgs + 1791) / 1792) * 133;
    return t_h3 <= t_f32;
}

// UU3D_SKIP=<bit mask> (TIMING EXPERIMENTS ONLY: the skipped launches leave garbage, results are wrong): which launch classes of the
// forward are left out -- 1 spatial stack, 2 LayerNorm-fed panel GEMMs (QKV, fc1), 4 projection, 8 fused MLP, 16 attention, 32 ln_split_frag,
// 64 ln_res_split_frag, 128 the temporal chain launches, 256 strided blocks 2.., 512 strided block 1.  tools/marginal_exp.sh prices what each class costs the pipelined step.
// The hooks exist only in TIMING BUILDS (-DUU3D_TIMING_BUILD: `python uplift-upsample-3dhpe_amd/build.py --timing` writes csrc/libuu3d_timing.so,
// whose uu3d_version() says so); the product library compiles them out: no environment variable can make it skip a launch.
#ifdef UU3D_TIMING_BUILD
inline int skip_mask() {
    static const int mask = [] { const char* e = getenv("UU3D_SKIP"); const int v = e ? atoi(e) : 0;
                                 if (v) fprintf(stderr, "[uu3d] UU3D_SKIP=%d: launches are being skipped, RESULTS ARE WRONG (timing experiment)\n", v); return v; }();
    return mask;
}
#else
inline constexpr int skip_mask() { return 0; }
#endif

struct Launcher {
    uu3d_model* m;
    hipStream_t stream;
    float* slab = nullptr;          // split-K partial sums
    size_t slab_floats = 0;
    bool throughput = false;        // this CALL's schedule (uu3d_forward_ex): launches shaped for CU-microseconds instead of latency
    int precision = UU3D_PREC_F16X3;   // this CALL's arithmetic: the handle's, or UU3D_PREC_F32 with UU3D_SCHEDULE_EXACT_F32
    int status = UU3D_OK;
    bool few_splits = false;        // this forward runs the temporal chain: its split-K GEMMs aim for splitk_target() workgroups

    void begin(const char* name, const char* kernel, double flops, double bytes) {
        if (!m->profiling) return;
        if (m->prof_used == m->prof.size()) {
            ProfRec r;
            if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) { status = UU3D_ERR_HIP; return; }
            m->prof.push_back(r);
        }
        ProfRec& r = m->prof[m->prof_used];
        r.name = name; r.kernel = kernel; r.flops = flops; r.bytes = bytes;
        (void)hipEventRecord(r.e0, stream);
    }
    void end() {
        if (m->profiling && m->prof_used < m->prof.size()) { (void)hipEventRecord(m->prof[m->prof_used].e1, stream); ++m->prof_used; }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess && status == UU3D_OK) { status = UU3D_ERR_HIP; m->err = std::string("kernel launch failed: ") + hipGetErrorString(e); }
    }

    template <int BM, int BN, class AL, class EP>
    void gemm_tile(const AL& al, const float* Bt, int M, int N, int Kp, int slices, int kt_per_split, const EP& ep) {
        auto kern = gemm_f32_kernel<BM, BN, AL, EP>;
        constexpr size_t lds = gemm_lds_bytes(BM, BN);
        static bool attr_done = false;   // one attribute call per instantiation
        if (!attr_done) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done = true; }
        const int mt = (M + BM - 1) / BM, nt = (N + BN - 1) / BN;
        const int grid = round_up(mt, 8) * nt;
        hipLaunchKernelGGL(kern, dim3(grid, slices), dim3(256), lds, stream, al, Bt, M, N, Kp, mt, nt, kt_per_split, ep);
    }

    template <int TM, int TN, class AL, class EP>
    void gemm_h3_tile(const AL& al, const _Float16* Bh, const _Float16* Bl, int M, int N, int Kp, int slices, int kt_per_split, const EP& ep) {
        auto kern = gemm_h3_kernel<TM, TN, AL, EP>;
        auto kern_deep = gemm_h3_kernel<TM, TN, AL, EP, 1>;     // few workgroups per CU: loads three k-tiles ahead (uu3d_gemm_h3.h)
        constexpr size_t lds = gemm_h3_lds_bytes(64 * TM, 64 * TN);
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void*)kern_deep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_done = true;
        }
        const int mt = (M + 64 * TM - 1) / (64 * TM), nt = (N + 64 * TN - 1) / (64 * TN);
        const int grid = round_up(mt, 8) * nt;
        if (gemm_h3_deep(grid * slices)) hipLaunchKernelGGL(kern_deep, dim3(grid, slices), dim3(256), lds, stream, al, Bh, Bl, M, N, Kp, mt, nt, kt_per_split, ep);
        else hipLaunchKernelGGL(kern, dim3(grid, slices), dim3(256), lds, stream, al, Bh, Bl, M, N, Kp, mt, nt, kt_per_split, ep);
    }

    template <int TM, int TN, class GL, class EP>
    void gemm_h3g_tile(const GL& gl, const _Float16* Bh, const _Float16* Bl, int M, int N, int Kp, int slices, int kt_per_split, const EP& ep) {
        auto kern = gemm_h3g_kernel<TM, TN, GL, EP>;
        constexpr size_t lds = gemm_h3g_lds_bytes(64 * TM, 64 * TN);
        static bool attr_done = false;
        if (!attr_done) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done = true; }
        const int mt = (M + 64 * TM - 1) / (64 * TM), nt = (N + 64 * TN - 1) / (64 * TN);
        const int grid = round_up(mt, 8) * nt;
        hipLaunchKernelGGL(kern, dim3(grid, slices), dim3(256), lds, stream, gl, Bh, Bl, M, N, Kp, mt, nt, kt_per_split, ep);
    }

    // Workgroups a split-K GEMM aims for: enough to fill the chip three times over when the launch has it to itself (latency); under the throughput
    // schedule the chip is shared and a launch costs its CU-microseconds -- hundreds of workgroups that each wait ten microseconds for a few
    // hundred kilobytes are expensive then (round 5: strided blocks 2 and 3, 7 GFLOP, cost 9 % of the pipelined step)
    int splitk_target() const {
        // (h36m_351, batch 128, eight slots, ms per step by target: 768: 0.678, 384: 0.665, 192: 0.664, 96: 0.666, 48: 0.664 -- profiles/r05_tchain_ab.txt)
        static const int thr = [] { const char* e = getenv("UU3D_THR_SPLITK_TARGET"); return e != nullptr && atoi(e) > 0 ? atoi(e) : 192; }();
        return (throughput && few_splits) ? thr : 768;      // (only beside the temporal chain: forwards without it stay bit-identical between the schedules)
    }
    // f16x3 GEMM whose A operand already is a pair of f16 planes (written by ln_split / attention / the ReLU
    // epilogue): LDS-DMA staged kernel, K % 32 == 0.  Same split-K policy as gemm().
    template <class GL, class EP>
    void gemm_g(const char* name, const GL& gl, const float* Bt, int M, int N, int K, const EP& ep, double extra_bytes = 0) {
        const int KT = K / 32;
        const int tiles = ((M + 63) / 64) * ((N + 63) / 64);
        int slices = 1;
        // (>= 200 tiles with a short contraction fill the chip unsplit: strided block 2's projection, 276 tiles x 12 k-tiles,
        // 13.3 us against 12.6 + 6.5 us split three ways + reduce; the heads and the K = 2304 convolutions measured faster split)
        if (tiles < 384 && KT >= 8 && !(tiles >= 200 && KT <= 16)) slices = std::max(1, std::min(KT / 4, (splitk_target() + tiles / 2) / tiles));
        int kps = (KT + slices - 1) / slices;
        slices = (KT + kps - 1) / kps;
        const int ldslab = round_up(N, 4);
        if (slices > 1 && (size_t)slices * M * ldslab > slab_floats) { slices = 1; kps = KT; }
        begin(name, "gemm_h3", 2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N) + extra_bytes);
        const auto it = m->hplanes.find((size_t)(Bt - m->arena));
        if ((K % 32) != 0 || it == m->hplanes.end()) { status = UU3D_ERR_INVALID_ARGUMENT; m->err = "gemm_g: operand without f16 planes or K % 32 != 0"; end(); return; }
        const _Float16* Bh = m->harena + it->second.first; const _Float16* Bl = m->harena + it->second.second;
        if (slices == 1) {
            // measured (tools/gemm_bench, M = 4544 / 1472 rows): 64x128 is the fastest LDS-DMA tile down to ~200 tiles
            if (N % 128 == 0 && tiles >= 256) gemm_h3g_tile<1, 2>(gl, Bh, Bl, M, N, K, 1, KT, ep);
            else gemm_h3g_tile<1, 1>(gl, Bh, Bl, M, N, K, 1, KT, ep);
        } else {
            EpSlab es{slab, ldslab, (size_t)M * ldslab};
            // split K with many tiles (strided block 1's convolution: 276 tiles x 3 slices of K = 768): 64 x 128 tiles move a
            // quarter less through L2 -> LDS than 64 x 64 (41.6 -> 36.5 us); with few tiles the 64 x 64 grid fills more CUs
            if (N % 128 == 0 && tiles >= 200) gemm_h3g_tile<1, 2>(gl, Bh, Bl, M, N, K, slices, kps, es);
            else gemm_h3g_tile<1, 1>(gl, Bh, Bl, M, N, K, slices, kps, es);
            hipLaunchKernelGGL(splitk_reduce_kernel<EP>, dim3((M * N + 255) / 256), dim3(256), 0, stream,
                               slab, slices, (size_t)M * ldslab, M, N, ldslab, ep);
        }
        end();
    }

    // Row-panel GEMM (uu3d_gemm_panel.h): C = A B + colv with A the fragment-ordered planes written by ln_split_frag and
    // B the fragment-ordered operand at harena + pf.  K = 384.  The split S of the N / 32 column chunks over workgroups
    // minimises rounds x (prologue + chunks per workgroup) for one workgroup per CU (measured order of the candidates
    // at M = 9088 / 4544, N = 1152 / 768: tools/gemm_panel_exp).
    static int panel_splits(int M, int N) {
        const int mt = (M + 127) / 128, chunks = N / 32;
        int best = 0; double best_cost = 1e30;
        for (int S = 1; S <= chunks; ++S) {
            if (chunks % S != 0 || chunks / S > PANEL_COLV_FLOATS / 32) continue;
            const int per_xcd = (mt * S + 7) / 8;                           // workgroups on the busiest XCD (32 CUs, one workgroup each)
            const double cost = (double)((per_xcd + 31) / 32) * (2.5 + (double)chunks / S);
            if (cost < best_cost) { best_cost = cost; best = S; }
        }
        return best;
    }
    bool panel_ok(int M, int N, int K, size_t pf) const {
        return !m->no_panel && pf != 0 && K == 384 && N % 32 == 0 && M >= 1024 && (double)M * N * 4.0 < 4.0e9;      // 32-bit byte offsets in the epilogue stores
    }
    // The 8-wave form of the row-panel GEMM (uu3d_gemm_panel8.h: contraction split over wave pairs, two waves per SIMD) for the chunk
    // counts it is instantiated for; UU3D_PANEL4=1 keeps every launch on the 4-wave kernel (A/B measurements).  Returns false when
    // the caller has to launch the 4-wave kernel.
    static bool panel8_ok(int cpw) {
        static const bool off = getenv("UU3D_PANEL4") != nullptr && atoi(getenv("UU3D_PANEL4")) != 0;
        return !off && (cpw == 4 || cpw == 6 || cpw == 8 || cpw == 9 || cpw == 12);      // (9: QKV at 284 row tiles = batch 512, four column ranges)
    }
    // the profile records' kernel name = the kernel SYMBOL's distinguishing part (bench.py picks the dominant kernel by it)
    template <class EP> static const char* panel_symbol(bool eight) {
        const char* e = std::is_same<EP, PanelEpBiasSplitQ>::value ? "BiasSplitQ" : std::is_same<EP, PanelEpBiasResidual>::value ? "BiasResidual"
                      : std::is_same<EP, PanelEpBiasResidualLn>::value ? "BiasResidualLn" : std::is_same<EP, PanelEpBiasReluSplit>::value ? "BiasReluSplit"
                      : std::is_same<EP, PanelEpBiasRelu>::value ? "BiasRelu" : "Bias";
        static thread_local char buf[32];
        std::snprintf(buf, sizeof buf, "%s<%s>", eight ? "gemm_panel8" : "gemm_panel", e);
        return buf;
    }
    template <class EP>
    bool launch_panel8(const _Float16* Af, const _Float16* Bf, const float* colv, int M, int mt, int S, int cpw, const EP& ep) {
        if (!panel8_ok(cpw)) return false;
        const dim3 grid(8 * S, ((mt * S + 7) / 8 + S - 1) / S);
#define UU3D_P8_LAUNCH(CPW) { auto kern = gemm_h3_panel8_kernel<EP, CPW, 3>; \
            static const bool once = (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P8_LDS_TOTAL) == hipSuccess); (void)once; \
            hipLaunchKernelGGL(kern, grid, dim3(512), P8_LDS_TOTAL, stream, Af, Bf, colv, M, mt, S, ep); return true; }
        switch (cpw) {
            case 4: UU3D_P8_LAUNCH(4)
            case 6: UU3D_P8_LAUNCH(6)
            case 8: UU3D_P8_LAUNCH(8)
            case 9: UU3D_P8_LAUNCH(9)
            case 12: UU3D_P8_LAUNCH(12)
            default: return false;
        }
#undef UU3D_P8_LAUNCH
    }
    template <class EP>
    void gemm_panel(const char* name, const _Float16* Af, size_t pf, const float* colv, int M, int N, const EP& ep) {
        const int K = 384, mt = (M + 127) / 128;
        int S = panel_splits(M, N);
        { static const char* e = getenv("UU3D_PANEL_S"); if (e != nullptr && e[0] >= '1' && e[0] <= '3' && (N / 32) % (e[0] - '0') == 0 && (N / 32) / (e[0] - '0') <= PANEL_COLV_FLOATS / 32) S = e[0] - '0'; }   // (A/B measurements)
        if (skip_mask() & 2) return;
        begin(name, panel_symbol<EP>(panel8_ok((N / 32) / S)), 2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
        if (!launch_panel8(Af, m->harena + pf, colv, M, mt, S, (N / 32) / S, ep)) {
            auto kern = gemm_h3_panel_kernel<24, EP>;
            static bool attr_done = false;
            if (!attr_done) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PANEL_LDS_TOTAL); attr_done = true; }
            hipLaunchKernelGGL(kern, dim3(8 * S, ((mt * S + 7) / 8 + S - 1) / S), dim3(256), PANEL_LDS_TOTAL, stream, Af, m->harena + pf, colv, M, mt, S, (N / 32) / S, ep);
        }
        end();
    }
    // x[M][384] += A B + colv in place (the attention projection on the residual stream): N = K = 384, the kernel's chunk loop unrolled
    // (CPW = 4 / 6 / 12 chunks per workgroup for 3 / 2 / 1 column ranges per row tile -- uu3d_gemm_panel.h says why)
    // ln_g / ln_b / ln_out (optional): LayerNorm 2 (eps 1e-5) of the finished rows as the next panel GEMM's A fragments, by the same launch
    // when it owns whole rows (throughput schedule, 8-wave kernel) -- returns true when it did, false when the caller still has to
    // launch ln_split_frag
    bool gemm_panel_residual(const char* name, const _Float16* Af, size_t pf, const float* colv, int M, float* x,
                             const float* ln_g = nullptr, const float* ln_b = nullptr, _Float16* ln_out = nullptr) {
        const int K = 384, N = 384, mt = (M + 127) / 128;
        if (skip_mask() & 4) return false;
        int S = 1; double best = 1e30;
        for (int s : {1, 2, 3}) {                                  // same cost model as panel_splits
            const int per_xcd = (mt * s + 7) / 8;
            const double cost = (double)((per_xcd + 31) / 32) * (2.5 + 12.0 / s);
            if (cost < best) { best = cost; S = s; }
        }
        // several forwards in flight: the fewest CU-microseconds win, not the shortest launch (h36m_351 batch 128, 9088 rows: S = 3 / 2 / 1
        // = 213 / 142 / 71 workgroups, 16.3 / 18.9 / 27.5 us per launch; one batch at a time 127.7 / 126.2 / 121.2 k sequences/s, four in
        // flight 167.3 / 169.2 / 171.3 k on the same box)
        // (... for launches that fill a good part of the chip.  With fewer than 64 row tiles -- strided block 2: 23 -- the launch holds few CUs either way, and what it
        // costs the pipelined step is its DURATION on its forward's queue: 34.6 us as 23 workgroups x 12 chunks with LayerNorm 2 inside against 12.2 + 7.2 us)
        if (throughput && mt >= 64) S = 1;
        { static const char* e = getenv("UU3D_PANEL_PROJ_S"); if (e != nullptr && e[0] >= '1' && e[0] <= '3') S = e[0] - '0'; }   // (A/B measurements)
        static const bool no_ln_tail = getenv("UU3D_NO_LN_TAIL") != nullptr;      // (A/B measurements)
        const bool ln_tail = S == 1 && ln_out != nullptr && !no_ln_tail && panel8_ok(12);
        begin(name, ln_tail ? panel_symbol<PanelEpBiasResidualLn>(true) : panel_symbol<PanelEpBiasResidual>(panel8_ok(12 / S)),
              2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)N * K + 2.0 * (double)M * N));
        const PanelEpBiasResidual ep{x, N};
        const dim3 grid(8 * S, ((mt * S + 7) / 8 + S - 1) / S);
        if (ln_tail) {
            const PanelEpBiasResidualLn epl{{x, N}, ln_g, ln_b, 1e-5f, ln_out};
            if (launch_panel8(Af, m->harena + pf, colv, M, mt, 1, 12, epl)) { end(); return true; }
        }
#define UU3D_PANEL_RES(CPW) { auto kern = gemm_h3_panel_kernel<24, PanelEpBiasResidual, CPW>; \
            static const bool once = (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PANEL_LDS_TOTAL) == hipSuccess); (void)once; \
            hipLaunchKernelGGL(kern, grid, dim3(256), PANEL_LDS_TOTAL, stream, Af, m->harena + pf, colv, M, mt, S, CPW, ep); }
        if (launch_panel8(Af, m->harena + pf, colv, M, mt, S, 12 / S, ep)) { end(); return false; }
        if (S == 3) UU3D_PANEL_RES(4) else if (S == 2) UU3D_PANEL_RES(6) else UU3D_PANEL_RES(12)
#undef UU3D_PANEL_RES
        end();
        return false;
    }
    // the same with the operand given by address (training: the packs regenerated from the master buffer) -- K = 384, N % 32 == 0
    template <class EP>
    void gemm_panel_at(const char* name, const _Float16* Af, const _Float16* Bf, const float* colv, int M, int N, const EP& ep) {
        const int K = 384, S = panel_splits(M, N), mt = (M + 127) / 128;
        begin(name, panel_symbol<EP>(panel8_ok((N / 32) / S)), 2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
        if (!launch_panel8(Af, Bf, colv, M, mt, S, (N / 32) / S, ep)) {
            auto kern = gemm_h3_panel_kernel<24, EP>;
            static bool attr_done = false;
            if (!attr_done) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PANEL_LDS_TOTAL); attr_done = true; }
            hipLaunchKernelGGL(kern, dim3(8 * S, ((mt * S + 7) / 8 + S - 1) / S), dim3(256), PANEL_LDS_TOTAL, stream, Af, Bf, colv, M, mt, S, (N / 32) / S, ep);
        }
        end();
    }
    void ln_split_frag_stats(const char* name, const float* x, int M, const float* g, const float* b, _Float16* Af, float2* stats) {
        begin(name, "ln_split_frag", 0.0, 8.0 * (double)M * 384);
        hipLaunchKernelGGL((ln_split_frag_stats_kernel<24, 8>), dim3((M + 7) / 8), dim3(128), 0, stream, x, 384, M, 1e-5f, g, b, Af, stats);
        end();
    }
    // LayerNorm (eps 1e-5) of M rows of 384 floats, written as the fragment-ordered planes of the panel GEMM's A operand
    void ln_split_frag(const char* name, const float* x, int M, const float* g, const float* b, _Float16* Af) {
        if (skip_mask() & 32) return;
        begin(name, "ln_split_frag", 0.0, 8.0 * (double)M * 384);
        hipLaunchKernelGGL((ln_split_frag_kernel<24, 8>), dim3((M + 7) / 8), dim3(128), 0, stream, x, 384, M, 1e-5f, g, b, Af);   // 8 rows per workgroup: 6.6 us at 9088 rows (16: 7.0, 32: 7.9, 4: 6.6)
        end();
    }

    // C[M][N] = A[M][K] * W.  64x64 tiles (4 workgroups per CU) measured fastest on every shape
    // of this model (tools/gemm_bench.hip).  Problems with too few tiles to fill the chip are
    // split along K into slabs and combined deterministically (splitk_reduce_kernel).
    template <class AL, class EP>
    void gemm(const char* name, const AL& al, const float* Bt, int M, int N, int K, const EP& ep, double extra_bytes = 0) {
        const int Kp = round_up(K, 32), KT = Kp / 32;
        const int tiles = ((M + 63) / 64) * ((N + 63) / 64);
        int slices = 1;
        // (>= 200 tiles with a short contraction fill the chip unsplit: strided block 2's projection, 276 tiles x 12 k-tiles,
        // 13.3 us against 12.6 + 6.5 us split three ways + reduce; the heads and the K = 2304 convolutions measured faster split)
        if (tiles < 384 && KT >= 8 && !(tiles >= 200 && KT <= 16)) slices = std::max(1, std::min(KT / 4, (splitk_target() + tiles / 2) / tiles));
        int kps = (KT + slices - 1) / slices;
        slices = (KT + kps - 1) / kps;
        const int ldslab = round_up(N, 4);
        if (slices > 1 && (size_t)slices * M * ldslab > slab_floats) { slices = 1; kps = KT; }
        const bool h3 = (precision == UU3D_PREC_F16X3);
        begin(name, h3 ? "gemm_h3" : "gemm_f32", 2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N) + extra_bytes);
        if (h3) {
            const auto it = m->hplanes.find((size_t)(Bt - m->arena));
            if (it == m->hplanes.end()) { status = UU3D_ERR_INVALID_ARGUMENT; m->err = "operand without f16 planes"; end(); return; }
            const _Float16* Bh = m->harena + it->second.first; const _Float16* Bl = m->harena + it->second.second;
            if (slices == 1) {
                // measured (tools/gemm_bench): 64x128 is the fastest f16x3 tile on every shape of the model
                if (N % 128 == 0 && tiles >= 512) gemm_h3_tile<1, 2>(al, Bh, Bl, M, N, Kp, 1, KT, ep);
                else gemm_h3_tile<1, 1>(al, Bh, Bl, M, N, Kp, 1, KT, ep);
            } else {
                EpSlab es{slab, ldslab, (size_t)M * ldslab};
                gemm_h3_tile<1, 1>(al, Bh, Bl, M, N, Kp, slices, kps, es);
                hipLaunchKernelGGL(splitk_reduce_kernel<EP>, dim3((M * N + 255) / 256), dim3(256), 0, stream,
                                   slab, slices, (size_t)M * ldslab, M, N, ldslab, ep);
            }
        } else if (slices == 1) {
            // measured (tools/gemm_bench): 64x128 beats 64x64 by ~8 % on the N = 384 GEMMs, loses elsewhere
            if (N % 128 == 0 && N <= 512 && tiles >= 512) gemm_tile<64, 128>(al, Bt, M, N, Kp, 1, KT, ep);
            else gemm_tile<64, 64>(al, Bt, M, N, Kp, 1, KT, ep);
        } else {
            EpSlab es{slab, ldslab, (size_t)M * ldslab};
            gemm_tile<64, 64>(al, Bt, M, N, Kp, slices, kps, es);
            hipLaunchKernelGGL(splitk_reduce_kernel<EP>, dim3((M * N + 255) / 256), dim3(256), 0, stream,
                               slab, slices, (size_t)M * ldslab, M, N, ldslab, ep);
        }
        end();
    }

    // vit.MLP of a temporal block in one launch (uu3d_mlp_fused.h): partial fc2 sums of the 3 hidden slices -> mslab
    bool mlpf_ok(int M, const BlockDev& b) const {
        return !m->no_mlpf && b.w2_mf != 0 && panel_ok(M, m->cfg.h_temporal, m->cfg.d_temporal, b.w1_pf);
    }
    void mlp_fused(const char* name, const _Float16* Af, const BlockDev& b, int M, float* mslab) {
        if (skip_mask() & 8) return;
        const int mt = (M + 127) / 128, S = MLPF_SLICES;
        begin(name, "mlp_fused", 4.0 * M * (double)m->cfg.d_temporal * m->cfg.h_temporal, 4.0 * ((double)M * 384 + 2.0 * 384 * 768 + 3.0 * M * 384));
        static bool attr_done = false;
        if (!attr_done) { (void)hipFuncSetAttribute((const void*)mlp_fused_h3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PANEL_LDS_TOTAL); attr_done = true; }
        hipLaunchKernelGGL(mlp_fused_h3_kernel, dim3(8 * S, ((mt * S + 7) / 8 + S - 1) / S), dim3(256), PANEL_LDS_TOTAL, stream,
                           Af, m->harena + b.w1_pf, m->harena + b.w2_mf, b.b1, mslab, M, mt);
        end();
    }
    // One launch of the temporal chain (uu3d_tchain16.h): the row-local stages of a block for every 64-row tile
    void tchain(const char* name, const uu3d_model::TcLaunch& t, int M, const _Float16* Of, float* X, float* XA, const float* pe, int period,
                _Float16* Q, _Float16* H, unsigned char* scratch) {
        if (skip_mask() & 128) return;
        const int mt = (M + 63) / 64;
        TChainArgs a{};
        a.M = M; a.m_tiles = mt; a.period = period; a.qscale = attn_qscale();
        a.Of = Of; a.X = X; a.XA = XA; a.pe = pe; a.W = m->harena + t.w_off; a.P = m->arena + t.p_off; a.Q = Q; a.H = H; a.scratch = scratch;
        const double cols = ((t.flags & TC_PROJ) ? 384.0 : 0.0) + ((t.flags & TC_MLP) ? 1536.0 : 0.0) + ((t.flags & TC_FC1_PLANES) ? 768.0 : 0.0) + ((t.flags & TC_QKV) ? 1152.0 : 0.0);
        begin(name, "tchain", 2.0 * M * 384.0 * cols, 4.0 * (384.0 * cols + 2.0 * M * 384.0 + ((t.flags & TC_QKV) ? M * 1152.0 : 0.0) + ((t.flags & TC_FC1_PLANES) ? M * 768.0 : 0.0)));
#define UU3D_T16_LAUNCH(F) case F: { auto kern = tchain16_kernel<F>; \
            static const bool once = (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)T16_LDS_TOTAL) == hipSuccess); (void)once; \
            hipLaunchKernelGGL(kern, dim3(mt), dim3(512), T16_LDS_TOTAL, stream, a); } break;
        switch (t.flags) {
            UU3D_T16_LAUNCH(TC_QKV)
            UU3D_T16_LAUNCH(TC_PROJ | TC_MLP | TC_QKV)
            UU3D_T16_LAUNCH(TC_PROJ | TC_MLP | TC_QKV | TC_PE)
            UU3D_T16_LAUNCH(TC_PROJ | TC_MLP)
            UU3D_T16_LAUNCH(TC_PROJ | TC_FC1_PLANES)
            default: status = UU3D_ERR_UNSUPPORTED; m->err = "temporal chain: unknown stage set"; break;
        }
#undef UU3D_T16_LAUNCH
        end();
    }
    // the fused MLP's combine (x += b2 + slabs; optionally xa = x + pe) + LayerNorm + split into A fragments
    void ln_res_split_frag(const char* name, float* x, int M, const float* bias2, const float* mslab, float* xa, const float* pe, int period,
                           const float* g, const float* b, _Float16* Af) {
        if (skip_mask() & 64) return;
        begin(name, "ln_split_frag", 0.0, 4.0 * (double)M * 384 * (xa ? 8 : 7));
        hipLaunchKernelGGL((ln_res_split_frag_kernel<24, 8>), dim3((M + 7) / 8), dim3(128), 0, stream, x, 384, M, 1e-5f, bias2, mslab, xa, pe, period, g, b, Af);
        end();
    }

    // Few rows (strided blocks 2-3, the heads): one workgroup per 32 x 32 tile, split-K over its waves, LayerNorm in the loader
    // (uu3d_gemm_wt.h) -- one launch where the tiled path needs row_stats + split-K GEMM + splitk_reduce.
    bool wt_ok(const float* Bt, int K) const {
        return precision == UU3D_PREC_F16X3 && !m->no_wt && (K % 16) == 0 && m->hplanes.count((size_t)(Bt - m->arena)) != 0;
    }
    template <class AL, class EP>
    void gemm_wt(const char* name, const AL& al, const float* Bt, int M, int N, int K, const EP& ep, double extra_bytes = 0) {
        const int Kp = round_up(K, 32), slices = Kp / 16;
        const auto it = m->hplanes.find((size_t)(Bt - m->arena));
        const _Float16* Bh = m->harena + it->second.first; const _Float16* Bl = m->harena + it->second.second;
        const int mt = (M + 31) / 32, nt = (N + 31) / 32;
        begin(name, "gemm_wt", 2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N) + extra_bytes);
        if (slices <= 6 * WT_MAX_WAVES) {                    // K <= 768: one batch of loads per wave
            const int kw = (slices + 5) / 6;
            hipLaunchKernelGGL((gemm_h3_wt_kernel<AL, EP, 6>), dim3(mt * nt), dim3(64 * kw), gemm_wt_lds_bytes(kw), stream, al, Bh, Bl, M, N, Kp, nt, ep);
        } else { status = UU3D_ERR_UNSUPPORTED; m->err = "gemm_wt: contraction longer than 768"; }
        end();
    }

    void row_stats(const char* name, const float* x, int D, int M, float2* stats) {
        begin(name, "row_stats", 0.0, 4.0 * (double)M * D + 8.0 * M);
        hipLaunchKernelGGL(row_stats_kernel<4>, dim3((M + 3) / 4), dim3(256), 0, stream, x, D, D, M, 1e-5f, stats);
        end();
    }

    // sequences served by attn_h3_kernel (f16x3 products, online softmax; q / k / v as f16 planes from the QKV epilogue): everything
    // the exact-f32 kernels cannot hold (> 128 tokens) and, measured faster, 49-128 tokens as well
    // (shorter sequences, h36m_81's 41 tokens: the split epilogue of the QKV projection costs more than the attention gains -- 242.1 k
    // sequences/s with the exact-f32 kernels there against 240.2 k)
    bool attn_is_h3(int L, bool planes_out) const { return planes_out && L <= ATTN_H3_MAX_L && (L > 128 || (L > 48 && !m->attn_f32)); }
    bool attn_h3_any(int L) const { return precision == UU3D_PREC_F16X3 && L <= ATTN_H3_MAX_L && !m->attn_f32; }
    float attn_qscale() const { return 1.44269504088896341f / sqrtf((float)kDH); }
    // split_lo_off != 0: the context rows go out as f16 planes (hi at out, lo split_lo_off halfs further)
    // frag: the context rows in the row-panel GEMM's A-fragment order instead of row-major planes (split_lo_off != 0 only)
    // qfrag: q | k | v arrive in the temporal chain's fragment order (uu3d_tchain16.h, tchain_qf_index) -- attn_h3_kernel only
    void attn(const char* name, const float* qkv, int B, int L, const uint8_t* mask, float* out, size_t split_lo_off = 0, bool frag = false, bool qfrag = false) {
        if (skip_mask() & 16) return;
        const int D = m->cfg.d_temporal, H = m->cfg.num_heads;
        const int NT = (L + 15) / 16;
        const bool h3a = attn_is_h3(L, split_lo_off != 0) || (qfrag && attn_h3_any(L));          // (the chain's fragment-ordered planes cost no split epilogue: attn_h3_kernel below 49 tokens too)
        begin(name, h3a ? "attn_h3" : "attn_f32", 4.0 * B * (double)H * L * L * kDH, 4.0 * 4.0 * B * (double)L * D);
        const int items = B * H;
        const dim3 grid(items);
        if (qfrag && !h3a) { status = UU3D_ERR_UNSUPPORTED; m->err = "fragment-ordered q | k | v need attn_h3_kernel"; end(); return; }
        if (h3a) {                                                 // qkv = hi plane [B L][3 D] halfs, lo plane behind it
            const int nt = (L + 31) / 32;
            const size_t lds = attn_h3_lds_bytes(L, kDH);
            _Float16* oh = reinterpret_cast<_Float16*>(out);
            const _Float16* qh = reinterpret_cast<const _Float16*>(qkv); const _Float16* ql = qh + (size_t)B * L * 3 * D;
#define UU3D_ATTN_H3(MW, WPE, MASKED, PIPE, waves) { auto k = attn_h3_kernel<kDH, MW, WPE, MASKED, PIPE>; \
                static const bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_h3_lds_bytes(ATTN_H3_MAX_L, kDH)) == hipSuccess); (void)once; \
                hipLaunchKernelGGL(k, grid, dim3(64 * (waves)), lds, stream, qh, ql, 3 * D, D, L, H, mask, oh, frag ? (size_t)512 : split_lo_off, D, frag ? 1 : 0, qfrag ? 1 : 0); }
            // 4 .. 12 key tiles (dense-351: 11): one wave per query tile, three per SIMD.  UU3D_ATTN_PIPE=1 (round 4, measured 4 % SLOWER: 35.6 against
            // 34.3 us at 351 tokens, batch 32): the PIPE form (operand reads by name ahead of their use, uu3d_attn_h3.h) needs ~190 registers =
            // two waves per SIMD, so 8 waves walk the query tiles -- the exposed LDS round trips of the round-3 kernel are not what it waits for
            static const bool no_pipe = getenv("UU3D_ATTN_PIPE") == nullptr;
            if (nt <= 3) { if (mask) UU3D_ATTN_H3(3, 3, true, false, nt) else UU3D_ATTN_H3(3, 3, false, false, nt) }
            else if (nt <= 12 && no_pipe) { if (mask) UU3D_ATTN_H3(12, 3, true, false, nt) else UU3D_ATTN_H3(12, 3, false, false, nt) }     // one wave per query tile, three per SIMD (dense-351: 35.4 vs 36.8 us with 8 waves x 2 tiles)
            else if (nt <= 12) { if (mask) UU3D_ATTN_H3(8, 2, true, true, std::min(nt, 8)) else UU3D_ATTN_H3(8, 2, false, true, std::min(nt, 8)) }
            else { if (mask) UU3D_ATTN_H3(8, 2, true, false, std::min(nt, 8)) else UU3D_ATTN_H3(8, 2, false, false, std::min(nt, 8)) }
#undef UU3D_ATTN_H3
            end();
            return;
        }
        // one wave per (sequence, head), four heads per workgroup (attn_head_wave_kernel): whenever the heads come in fours and the
        // K / V tiles of four heads fit the LDS; UU3D_ATTN_WG=1 keeps the workgroup-per-item kernel
        // NT <= 3: more, smaller workgroups hide latency better (measured)
        if (!m->attn_wg && H % 4 == 0 && NT >= 4 && NT <= 5) {
            const dim3 hgrid((items + 3) / 4);
#define UU3D_ATTN_HW(nt) case nt: { \
            constexpr size_t lds = attn_head_wave_lds_bytes<nt, kDH>(); \
            if (split_lo_off) { auto k = attn_head_wave_kernel<nt, kDH, true>; static const bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess); (void)once; \
                hipLaunchKernelGGL(k, hgrid, dim3(256), lds, stream, qkv, 3 * D, D, L, H, mask, out, D, frag ? (size_t)512 : split_lo_off, items); } \
            else { auto k = attn_head_wave_kernel<nt, kDH, false>; static const bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess); (void)once; \
                hipLaunchKernelGGL(k, hgrid, dim3(256), lds, stream, qkv, 3 * D, D, L, H, mask, out, D, (size_t)0, items); } \
            } break;
            switch (NT) { UU3D_ATTN_HW(4) UU3D_ATTN_HW(5) default: break; }
#undef UU3D_ATTN_HW
            end();
            return;
        }
#define UU3D_ATTN_CASE(nt) case nt: \
        if (split_lo_off) hipLaunchKernelGGL((attn_f32_kernel<nt, kDH, true>), grid, dim3(64 * nt), 0, stream, qkv, 3 * D, D, L, H, mask, out, D, frag ? (size_t)512 : split_lo_off); \
        else hipLaunchKernelGGL((attn_f32_kernel<nt, kDH, false>), grid, dim3(64 * nt), 0, stream, qkv, 3 * D, D, L, H, mask, out, D, (size_t)0); \
        break;
        switch (NT) {
            UU3D_ATTN_CASE(1) UU3D_ATTN_CASE(2) UU3D_ATTN_CASE(3) UU3D_ATTN_CASE(4)
            UU3D_ATTN_CASE(5) UU3D_ATTN_CASE(6) UU3D_ATTN_CASE(7) UU3D_ATTN_CASE(8)
            default: status = UU3D_ERR_UNSUPPORTED; m->err = "attention over more than 128 tokens needs the f16x3 path (precision f16x3, d_t and h_t multiples of 32)"; break;
        }
#undef UU3D_ATTN_CASE
        end();
    }
};

}  // namespace

int uu3d_forward(uu3d_model* m, const float* kp2d, const uint8_t* mask, int32_t B, float* full_out,
                 float* central_out, void* workspace, size_t workspace_bytes, void* stream_) {
    return uu3d_forward_attention(m, kp2d, mask, B, full_out, central_out, nullptr, workspace, workspace_bytes, stream_);
}

int uu3d_forward_attention(uu3d_model* m, const float* kp2d, const uint8_t* mask, int32_t B, float* full_out,
                           float* central_out, float* const* attn_out, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!m) return UU3D_ERR_INVALID_ARGUMENT;
    return uu3d_forward_ex(m, kp2d, mask, B, full_out, central_out, attn_out, workspace, workspace_bytes,
                           m->throughput ? UU3D_SCHEDULE_THROUGHPUT : UU3D_SCHEDULE_LATENCY, stream_);
}

int uu3d_forward_ex(uu3d_model* m, const float* kp2d, const uint8_t* mask, int32_t B, float* full_out,
                    float* central_out, float* const* attn_out, void* workspace, size_t workspace_bytes, int32_t schedule, void* stream_) {
    if (!m) return UU3D_ERR_INVALID_ARGUMENT;
    const bool exact_f32 = (schedule & UU3D_SCHEDULE_EXACT_F32) != 0;
    schedule &= ~UU3D_SCHEDULE_EXACT_F32;
    // TIMING EXPERIMENT (tools/tail_branch_exp.py; results wrong): 0x200 = only the launches up to the first strided block, 0x400 = only the ones behind it
    // (only with UU3D_TIMING_PARTS=1 in the environment: otherwise the bits are an invalid schedule like any other unknown value)
#ifdef UU3D_TIMING_BUILD
    static const bool parts_ok = getenv("UU3D_TIMING_PARTS") != nullptr && atoi(getenv("UU3D_TIMING_PARTS")) != 0;
#else
    constexpr bool parts_ok = false;
#endif
    const bool part_body = parts_ok && (schedule & 0x200) != 0, part_tail = parts_ok && (schedule & 0x400) != 0;
    if (parts_ok) schedule &= ~0x600;
    if (schedule != UU3D_SCHEDULE_LATENCY && schedule != UU3D_SCHEDULE_THROUGHPUT) return fail(m, UU3D_ERR_INVALID_ARGUMENT, "schedule must be UU3D_SCHEDULE_LATENCY or UU3D_SCHEDULE_THROUGHPUT");
    if (!m->committed) return fail(m, UU3D_ERR_NOT_READY, "uu3d_commit_weights has not been called");
    if (!kp2d || !central_out || !workspace || B < 1) return fail(m, UU3D_ERR_INVALID_ARGUMENT, "null buffer or batch < 1");
    uu3d_config c = m->cfg;                                // (a copy: UU3D_SCHEDULE_EXACT_F32 changes THIS call's arithmetic, not the handle's)
    if (exact_f32) {
        if (m->generic) return fail(m, UU3D_ERR_UNSUPPORTED, "UU3D_SCHEDULE_EXACT_F32: handles with generic dims have no exact-f32 forward");
        int Lmax = c.num_frames;
        if (Lmax > 128) return fail(m, UU3D_ERR_UNSUPPORTED, "UU3D_SCHEDULE_EXACT_F32: the exact-f32 attention holds sequences of <= 128 tokens");
        c.precision = UU3D_PREC_F32;
    }
    if ((c.has_strided_input != 0) != (mask != nullptr))
        return fail(m, UU3D_ERR_INVALID_ARGUMENT, "stride_mask must be given iff the model has strided input");
    const bool has_h1 = c.full_output && c.temporal_depth > 0;
    if (has_h1 && !full_out) return fail(m, UU3D_ERR_INVALID_ARGUMENT, "full_out_dev is required for this model");
    if (((uintptr_t)workspace & 255) != 0) return fail(m, UU3D_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    if (m->generic) {
        // dims other than the compiled ones: the training-mode chain, forward only, with every stochastic layer off (no DropPath draws, no token
        // mask, Dropout rates 0): vit / u_u_t in inference mode
        const int r = generic_forward(m, kp2d, mask, B, full_out, central_out, attn_out, workspace, workspace_bytes, stream_);
        if (r == UU3D_OK && c.precision == UU3D_PREC_F16X3) {
            const bool has_full = c.full_output && c.temporal_depth > 0 && full_out != nullptr;
            hipLaunchKernelGGL(range_check_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream_, has_full ? full_out : central_out,
                               has_full ? (long)B * c.num_frames * c.num_keypoints * 3 : 0L, central_out, (long)B * c.num_keypoints * 3, m->d_range);
        }
        return r;
    }
    Workspace w = carve(m, B, (char*)workspace);
    if (workspace_bytes < w.bytes) return fail(m, UU3D_ERR_WORKSPACE, "workspace smaller than uu3d_workspace_bytes(batch)");
    if ((long)B * c.num_frames * c.num_keypoints > (1L << 30)) return fail(m, UU3D_ERR_INVALID_ARGUMENT, "batch too large");

    HIPCHK(m, hipSetDevice(m->device));
    Launcher Lh{m, (hipStream_t)stream_, w.slab, w.slab_floats, schedule == UU3D_SCHEDULE_THROUGHPUT, c.precision};
    m->prof_used = 0;
    const int N = c.num_frames, J = c.num_keypoints, ds = c.d_spatial, dt = c.d_temporal, ht = c.h_temporal;
    const int M = B * N;
    char nm[48];

    // 1. spatial stack
    // the f16x3 spatial kernel writes its output as the two f16 planes spatial_to_temporal_fc reads (LDS-DMA GEMM, no split in
    // the GEMM's loader): 32.6 -> 29.6 us for the GEMM, the spatial kernel unchanged (round 1's kernel paid 1-2 us for the split stores)
    const bool s2t_planes = !m->no_planes && c.precision == UU3D_PREC_F16X3 && !m->spatial_valu && !m->spatial_f32 && (m->spatial_h3_always || spatial_h3_pays(B * c.num_frames)) && ((c.num_keypoints * c.d_spatial) % 32 == 0);
    if (!part_tail) {
        SpatialParams sp = m->sp;
        sp.total_frames = M;
        sp.frame_list = nullptr;
        if (mask != nullptr && !m->spatial_valu) {
            Lh.begin("compact_frames", "compact_frames", 0.0, (double)M * 5.0);
            hipLaunchKernelGGL(compact_frames_kernel, dim3(1), dim3(1024), 0, Lh.stream, mask, M, w.frame_list);
            Lh.end();
            sp.frame_list = w.frame_list;
        }
        const double fl = (double)M * (2.0 * J * 2 * ds + c.spatial_depth * (4.0 * 2 * J * ds * ds + 8.0 * 4 * J * J * (ds / 8) + 2.0 * 2 * J * ds * kHS));
        if (m->spatial_valu) {
            constexpr int FPW = 256 / kJ;
            sp.blocks = m->sp_blocks_v1;
            auto kern = spatial_stack_kernel<kJ, kDS, kHS, kHeads>;
            static bool attr_done = false;
            if (!attr_done) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)spatial_lds_bytes(kDS)); attr_done = true; }
            Lh.begin("spatial_stack", "spatial_valu", fl, 4.0 * M * J * (2.0 + ds));
            hipLaunchKernelGGL(kern, dim3((M + FPW - 1) / FPW), dim3(256), spatial_lds_bytes(kDS), Lh.stream, kp2d, sp, w.S);
            Lh.end();
        } else if (c.precision == UU3D_PREC_F16X3 && !m->spatial_f32 && (m->spatial_h3_always || spatial_h3_pays(M))) {
            sp.blocks = m->sp_blocks_v2;             // LayerNorm parameters and biases
            auto kern = spatial_stack_h3_kernel<kJ, kFR, kSpatialMT>;
            Lh.begin("spatial_stack", "spatial_h3", fl, 4.0 * M * J * (2.0 + ds));
            if (!(skip_mask() & 1))
            hipLaunchKernelGGL(kern, dim3((M + kFR - 1) / kFR), dim3(64 * (2 / kSpatialMT)), sh3::lds_bytes(), Lh.stream, kp2d, sp,
                               m->harena + m->sp_frag_off, w.S, s2t_planes ? reinterpret_cast<_Float16*>(w.S) : (_Float16*)nullptr,
                               s2t_planes ? reinterpret_cast<_Float16*>(w.S) + (size_t)M * J * ds : (_Float16*)nullptr, SpatialTrainIO{});
            Lh.end();
        } else {
            sp.blocks = m->sp_blocks_v2;
            auto kern = spatial_stack_mfma_kernel<kJ, kFR>;
            Lh.begin("spatial_stack", "spatial_mfma", fl, 4.0 * M * J * (2.0 + ds));
            hipLaunchKernelGGL(kern, dim3((M + kFR - 1) / kFR), dim3(64), spatial_v2_lds_bytes(), Lh.stream, kp2d, sp, w.S);
            Lh.end();
        }
    }
    // 2. spatial_to_temporal_fc + token blend + temporal PE
    if (!part_tail) {
        EpSpatialToTemporal ep{w.X, m->s2t_b, dt, mask, m->token, m->pe_t, N};
        if (s2t_planes) {
            Lh.gemm_g("s2t", GLoadPlain{reinterpret_cast<const _Float16*>(w.S), reinterpret_cast<const _Float16*>(w.S) + (size_t)M * J * ds, J * ds, M}, m->s2t_wt, M, dt, J * ds, ep);
        } else {
            ALoadPlain al{w.S, J * ds, M, J * ds};
            Lh.gemm("s2t", al, m->s2t_wt, M, dt, J * ds, ep);
        }
    }
    // f16x3 with K % 32 == 0 everywhere: activations that feed a GEMM travel as f16 hi/lo planes (same bytes as the
    // f32 tensors they replace: O and Hb are reused) and the GEMMs are the LDS-DMA kernel gemm_h3g_kernel
    const bool planes = (c.precision == UU3D_PREC_F16X3) && (dt % 32 == 0) && (ht % 32 == 0) && !m->no_planes;
    constexpr int kFewRows = 512;                                 // up to here a GEMM runs on gemm_h3_wt_kernel (measured: see DESIGN.md)
    _Float16* const Ph = reinterpret_cast<_Float16*>(w.O);       // LayerNorm output (fragment order), then attention output (planes)
    _Float16* const Hh = reinterpret_cast<_Float16*>(w.Hb);      // relu(fc1)
    const _Float16* const hzero = m->harena;                     // 64 zero halfs (uu3d_commit_weights)

    // The first five launches of a transformer block (temporal: vit.py:176-188, strided: u_u_t.py:126-135), on the residual
    // stream x of Mr = B * L rows:  x += proj(MHA(LN1(x)));  Hb = relu(fc1(LN2(x)))  -- Hb as f16 planes when `planes`.
    //   LayerNorm-fed Dense layers: ln_split_frag + the row-panel GEMM (>= 1024 rows), else row_stats + the tiled GEMM
    //   with LayerNorm in its loader.
    // `pend` != nullptr: the previous temporal block ended in the fused MLP -- its fc2 result still sits in w.mslab; this
    // block's first LayerNorm launch adds it to the residual stream (for the first strided block: to w.X, and x = w.XA = X + pe).
    // `fuse_mlp`: stop after the second LayerNorm (its fragments in Ph feed mlp_fused).
    // returns the fragment-ordered LayerNorm-2 output when `fuse_mlp` stopped the block in front of the MLP (else nullptr)
    auto block_head = [&](const char* tag, int i, const BlockDev& b, float* x, int L, const uint8_t* kmask, const BlockDev* pend, bool fuse_mlp) -> const _Float16* {
        const int Mr = B * L;
        auto name = [&](const char* what) { snprintf(nm, sizeof nm, "%s%d.%s", tag, i + 1, what); return nm; };
        _Float16* const Pl = Ph + (size_t)Mr * dt; _Float16* const Hl = Hh + (size_t)Mr * ht;
        // Few rows (strided block 3: 384): the LayerNorm-fed Dense layers as ONE launch each on gemm_h3_wt_kernel (LayerNorm in
        // the loader, split-K inside the workgroup) instead of row_stats + split-K GEMM + splitk_reduce.  Measured per layer
        // (h36m_351, batch 128, HIP events): 16.6 vs 14.2 + 6.3 us (QKV), 16.8 vs 12.7 + 6.4 us (fc1).  The other few-row GEMMs
        // (projection, strided convolution, heads) are SLOWER there -- 32 x 32 tiles re-read both operands too often
        // (conv: 29 vs 20 us, head1: 19 vs 12 us) -- and stay on the tiled kernels.
        // q / k / v for attn_h3_kernel: f16 planes (hi at QKV, lo Mr * 3 d_t halfs further), q pre-multiplied by log2(e) / sqrt(d_h)
        const bool qsplit = Lh.attn_is_h3(L, planes);
        _Float16* const Qh = reinterpret_cast<_Float16*>(w.QKV); _Float16* const Ql = Qh + (size_t)Mr * 3 * dt;
        const EpBiasSplitQ ep_qs{Qh, Ql, b.bqkv, 3 * dt, dt, Lh.attn_qscale()};
        const bool few = planes && Mr <= kFewRows && Lh.wt_ok(b.wqkv_t, dt) && Lh.wt_ok(b.w1_t, dt);
        auto maps = [&]() {
            if (attn_out != nullptr && tag[0] == 't' && attn_out[i] != nullptr) {        // return_attention=True: the block's attention maps, recomputed from q | k
                Lh.begin(name("attn_maps"), "attn_probs", 2.0 * B * (double)c.num_heads * L * L * kDH, 4.0 * B * (double)c.num_heads * L * L);
                const size_t lds = (size_t)L * (kDH + 1) * sizeof(float);
                static const bool once = (hipFuncSetAttribute((const void*)attn_probs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) == hipSuccess); (void)once;
                if (qsplit) hipLaunchKernelGGL(attn_probs_kernel, dim3(B * c.num_heads), dim3(256), lds, Lh.stream, (const void*)Qh, (const _Float16*)Ql, 3 * dt, dt, L, c.num_heads, kDH, kmask, 1.0f, 1, attn_out[i]);
                else hipLaunchKernelGGL(attn_probs_kernel, dim3(B * c.num_heads), dim3(256), lds, Lh.stream, (const void*)w.QKV, (const _Float16*)nullptr, 3 * dt, dt, L, c.num_heads, kDH, kmask, 1.0f / sqrtf((float)kDH), 0, attn_out[i]);
                Lh.end();
            }
        };
        if (few) {
            WtLoadF32 l1{x, dt, Mr, dt, b.ln1_g, b.ln1_b, 1e-5f, 1};
            if (qsplit) Lh.gemm_wt(name("ln_qkv"), l1, b.wqkv_t, Mr, 3 * dt, dt, ep_qs);
            else Lh.gemm_wt(name("ln_qkv"), l1, b.wqkv_t, Mr, 3 * dt, dt, EpBias{w.QKV, b.bqkv, 3 * dt});
            maps();
            Lh.attn(name("attn"), w.QKV, B, L, kmask, w.O, (size_t)Mr * dt);
            { GLoadPlain gl{Ph, Pl, dt, Mr}; Lh.gemm_g(name("proj_res"), gl, b.wp_t, Mr, dt, dt, EpBiasResidual{x, b.bp, dt, nullptr, nullptr, 1}, 4.0 * Mr * dt); }
            WtLoadF32 l2{x, dt, Mr, dt, b.ln2_g, b.ln2_b, 1e-5f, 1};
            Lh.gemm_wt(name("ln_fc1"), l2, b.w1_t, Mr, ht, dt, EpBiasReluSplit{Hh, Hl, b.b1, ht});
            return nullptr;
        }
        if (pend != nullptr) {                                     // (same row count as the block that left it: the panel path holds)
            if (x == w.X) Lh.ln_res_split_frag(name("ln1_split"), w.X, Mr, pend->b2, w.mslab, nullptr, nullptr, 1, b.ln1_g, b.ln1_b, Ph);
            else Lh.ln_res_split_frag(name("ln1_split"), w.X, Mr, pend->b2, w.mslab, x, b.pe, L, b.ln1_g, b.ln1_b, Ph);
            if (qsplit) Lh.gemm_panel(name("ln_qkv"), Ph, b.wqkv_pf, b.bqkv, Mr, 3 * dt, PanelEpBiasSplitQ{Qh, Ql, 3 * dt, dt, Lh.attn_qscale()});
            else Lh.gemm_panel(name("ln_qkv"), Ph, b.wqkv_pf, b.bqkv, Mr, 3 * dt, PanelEpBias{w.QKV, 3 * dt});
        } else if (planes && Lh.panel_ok(Mr, 3 * dt, dt, b.wqkv_pf)) {
            Lh.ln_split_frag(name("ln1_split"), x, Mr, b.ln1_g, b.ln1_b, Ph);
            if (qsplit) Lh.gemm_panel(name("ln_qkv"), Ph, b.wqkv_pf, b.bqkv, Mr, 3 * dt, PanelEpBiasSplitQ{Qh, Ql, 3 * dt, dt, Lh.attn_qscale()});
            else Lh.gemm_panel(name("ln_qkv"), Ph, b.wqkv_pf, b.bqkv, Mr, 3 * dt, PanelEpBias{w.QKV, 3 * dt});
        } else {
            Lh.row_stats(name("stats1"), x, dt, Mr, w.stats);
            ALoadLayerNorm al{x, w.stats, b.ln1_g, b.ln1_b, dt, Mr, dt};
            if (qsplit) Lh.gemm(name("ln_qkv"), al, b.wqkv_t, Mr, 3 * dt, dt, ep_qs);
            else { EpBias ep{w.QKV, b.bqkv, 3 * dt}; Lh.gemm(name("ln_qkv"), al, b.wqkv_t, Mr, 3 * dt, dt, ep); }
        }
        maps();
        // projection on the row-panel GEMM (round 3): attn_h3_kernel writes the context rows in A-fragment order, the residual is
        // added in the epilogue from values requested a chunk earlier (UU3D_NO_PANEL_PROJ=1: the tiled LDS-DMA kernel)
        const bool pproj = planes && !m->no_panel_proj && Lh.panel_ok(Mr, dt, dt, b.wp_pf);      // (every attention kernel writes either layout)
        Lh.attn(name("attn"), w.QKV, B, L, kmask, w.O, planes ? (size_t)Mr * dt : 0, pproj);
        const bool mlp_panel = planes && Lh.panel_ok(Mr, ht, dt, b.w1_pf);
        bool ln2_done = false;
        if (pproj) ln2_done = mlp_panel ? Lh.gemm_panel_residual(name("proj_res"), Ph, b.wp_pf, b.bp, Mr, x, b.ln2_g, b.ln2_b, Ph)
                                        : Lh.gemm_panel_residual(name("proj_res"), Ph, b.wp_pf, b.bp, Mr, x);
        else {
            EpBiasResidual ep{x, b.bp, dt, nullptr, nullptr, 1};
            if (planes) { GLoadPlain gl{Ph, Pl, dt, Mr}; Lh.gemm_g(name("proj_res"), gl, b.wp_t, Mr, dt, dt, ep, 4.0 * Mr * dt); }
            else { ALoadPlain al{w.O, dt, Mr, dt}; Lh.gemm(name("proj_res"), al, b.wp_t, Mr, dt, dt, ep, 4.0 * Mr * dt); }
        }
        if (mlp_panel) {
            if (!ln2_done) Lh.ln_split_frag(name("ln2_split"), x, Mr, b.ln2_g, b.ln2_b, Ph);      // (throughput schedule: LayerNorm 2 rode in the projection's launch)
            if (fuse_mlp) return Ph;
            Lh.gemm_panel(name("ln_fc1"), Ph, b.w1_pf, b.b1, Mr, ht, PanelEpBiasReluSplit{Hh, Hl, ht});
        } else {
            Lh.row_stats(name("stats2"), x, dt, Mr, w.stats);
            ALoadLayerNorm al{x, w.stats, b.ln2_g, b.ln2_b, dt, Mr, dt};
            if (planes) { EpBiasReluSplit ep{Hh, Hl, b.b1, ht}; Lh.gemm(name("ln_fc1"), al, b.w1_t, Mr, ht, dt, ep); }
            else { EpBiasRelu ep{w.Hb, b.b1, ht}; Lh.gemm(name("ln_fc1"), al, b.w1_t, Mr, ht, dt, ep); }
        }
        return nullptr;
    };

    // 3. temporal blocks.
    // Throughput schedule (several forwards share the chip): the TEMPORAL CHAIN (uu3d_tchain16.h) -- per block one attention launch and one
    // launch for everything row-local (projection + residual, LayerNorm 2, fc1, ReLU, fc2 + residual, the next block's LayerNorm 1 + QKV) by
    // workgroups that own 64 token rows: 2 T + 3 launches for T temporal blocks and the head of the first strided block instead of 5 T + 5,
    // no partial-sum slabs, no LayerNorm passes.
    const bool chain = Lh.throughput && planes && m->tchain_mode != 0 && (m->tchain_mode == 1 || (M + 127) / 128 >= m->tchain_min_tiles) && !m->tchain.empty() && M >= 1024 && (Lh.attn_is_h3(N, true) || (m->tchain_short && Lh.attn_h3_any(N))) &&
                       (c.num_strided == 0 || m->L[0] == N) && (double)M * 1152 * 4.0 < 4.0e9 &&
                       attn_out == nullptr;      // (return_attention=True: the maps kernel reads row-major q | k planes, the chain writes fragment order)
    Lh.few_splits = chain;
    if (chain && !part_tail) {
        _Float16* const Q = reinterpret_cast<_Float16*>(w.QKV);
        Lh.tchain("t1.ln_qkv", m->tchain[0], M, nullptr, w.X, nullptr, nullptr, 1, Q, nullptr, w.tc_scratch);
        for (int i = 0; i < c.temporal_depth; ++i) {
            const bool masked = c.has_strided_input && i < c.first_strided_token_attention_layer;
            snprintf(nm, sizeof nm, "t%d.attn", i + 1);
            Lh.attn(nm, w.QKV, B, N, masked ? mask : nullptr, w.O, (size_t)M * dt, true, true);
            const bool to_strided = i + 1 == c.temporal_depth && c.num_strided > 0;
            snprintf(nm, sizeof nm, "t%d.chain", i + 1);
            Lh.tchain(nm, m->tchain[i + 1], M, Ph, w.X, to_strided ? w.XA : nullptr, to_strided ? m->sblocks[0].pe : nullptr, N, Q, nullptr, w.tc_scratch);
        }
    }
    // Otherwise: with >= 1024 token rows the MLP is one launch (uu3d_mlp_fused.h): its three partial fc2 sums are
    // added to the residual stream by the NEXT block's first LayerNorm launch (`pend`).
    const BlockDev* pend = nullptr;
    for (int i = 0; i < (chain || part_tail ? 0 : c.temporal_depth); ++i) {
        const BlockDev& b = m->tblocks[i];
        const bool masked = c.has_strided_input && i < c.first_strided_token_attention_layer;
        const bool last = (i + 1 == c.temporal_depth);
        // (the fused MLP leaves its result for the NEXT block's first LayerNorm launch: without strided blocks the last temporal block has none)
        const bool fuse = planes && Lh.mlpf_ok(M, b) && (last ? (c.num_strided > 0 && Lh.panel_ok(M, 3 * dt, dt, m->sblocks[0].wqkv_pf))
                                                              : Lh.panel_ok(M, 3 * dt, dt, m->tblocks[i + 1].wqkv_pf));
        const _Float16* const a2 = block_head("t", i, b, w.X, N, masked ? mask : nullptr, pend, fuse);
        pend = nullptr;
        if (fuse) {
            snprintf(nm, sizeof nm, "t%d.mlp", i + 1);
            Lh.mlp_fused(nm, a2, b, M, w.mslab);
            pend = &b;
            continue;
        }
        const bool to_strided = last && c.num_strided > 0;
        EpBiasResidual ep_fc2{w.X, b.b2, dt, to_strided ? w.XA : nullptr, to_strided ? m->sblocks[0].pe : nullptr, N};
        snprintf(nm, sizeof nm, "t%d.fc2_res", i + 1);
        if (planes) { GLoadPlain gl{Hh, Hh + (size_t)M * ht, ht, M}; Lh.gemm_g(nm, gl, b.w2_t, M, dt, ht, ep_fc2, 4.0 * M * dt); }
        else { ALoadPlain al{w.Hb, ht, M, ht}; Lh.gemm(nm, al, b.w2_t, M, dt, ht, ep_fc2, 4.0 * M * dt); }
    }
    if (c.num_strided == 0 && has_h1 && !part_tail) {      // 4. head1 without strided blocks (otherwise launched inside strided block 1, below)
        ALoadPlain al{w.X, dt, M, dt}; EpBias ep{full_out, m->h1_b, 3 * J};
        Lh.gemm("head1", al, m->h1_wt, M, 3 * J, dt, ep);
    }
    // 5. strided blocks
    float* xa = w.XA; float* xb = w.XB;
    if (c.temporal_depth == 0 && c.num_strided > 0) {      // no temporal block whose epilogue adds the first strided PE (u_u_t.py:382-383)
        Lh.begin("s1.add_pe", "add_pe", 0.0, 12.0 * M * dt);
        hipLaunchKernelGGL(add_period_kernel, dim3((unsigned)(((size_t)M * dt / 4 + 255) / 256)), dim3(256), 0, Lh.stream, w.X, m->sblocks[0].pe, M, dt, N, w.XA);
        Lh.end();
    }
    for (int i = 0; i < c.num_strided; ++i) {
        const BlockDev& b = m->sblocks[i];
        const int Li = m->L[i], Lo = m->L[i + 1], Mi = B * Li, Mo = B * Lo;
        if (((skip_mask() & 256) || part_body) && i >= 1) continue;       // (timing experiments: the strided blocks behind the first / 512: the first)
        if (((skip_mask() & 512) || part_tail) && i == 0) { std::swap(xa, xb); xb = w.XA; continue; }
        // MaxPool1D(pool 1, stride s) on the trimmed sequence; stride 1 keeps x untrimmed (u_u_t.py:138-154)
        const int lo = (c.strides[i] > 1 && c.pad_left[i] == 0) ? 1 : 0;
        const EpConvResidual ep_conv{xb, b.b2, dt, xa, Li, Lo, c.strides[i], lo,
                                     (i + 1 < c.num_strided) ? m->sblocks[i + 1].pe : nullptr};
        // (without temporal blocks the first strided block is the one that must not attend to the upsampling tokens, u_u_t.py:372-376)
        const bool smask = c.temporal_depth == 0 && c.has_strided_input && i < c.first_strided_token_attention_layer;
        if (chain && i == 0) {
            // the chain's last launch left q | k | v of this block (LayerNorm 1 of xa = x + pe); its projection, LayerNorm 2 and fc1 are the next one
            Lh.attn("s1.attn", w.QKV, B, Li, nullptr, w.O, (size_t)Mi * dt, true, true);
            Lh.tchain("s1.chain", m->tchain[c.temporal_depth + 1], Mi, Ph, nullptr, xa, nullptr, 1, nullptr, Hh, w.tc_scratch);      // (its stream: xa, lane-linear in the scratch since the last temporal launch; row-major xa written here for the convolution's residual rows)
        } else
        block_head("s", i, b, xa, Li, smask ? mask : nullptr, i == 0 ? pend : nullptr, false);
        if (i == 0 && has_h1) {
            // 4. head1 (after strided block 1's first LayerNorm launch, which completes w.X when the last MLP was fused).  Nothing
            // downstream reads it, but a side stream next to the strided blocks measured SLOWER (1.085 vs 1.040 ms per forward
            // replayed from a hipGraph: the cross-stream edges cost more than the 12 us they hide).
            ALoadPlain al{w.X, dt, M, dt}; EpBias ep{full_out, m->h1_b, 3 * J};
            Lh.gemm("head1", al, m->h1_wt, M, 3 * J, dt, ep);
        }
        snprintf(nm, sizeof nm, "s%d.conv_res", i + 1);
        if (planes) { GLoadConv3 gl{Hh, Hh + (size_t)Mi * ht, hzero, ht, Li, Lo, c.strides[i], c.pad_left[i], Mo};
                      Lh.gemm_g(nm, gl, b.w2_t, Mo, dt, 3 * ht, ep_conv, 4.0 * Mo * dt); }
        else { ALoadConv3 al{w.Hb, ht, Li, Lo, c.strides[i], c.pad_left[i], Mo, 3 * ht};
               Lh.gemm(nm, al, b.w2_t, Mo, dt, 3 * ht, ep_conv, 4.0 * Mo * dt); }
        std::swap(xa, xb);
        if (i == 0) xb = w.XA;   // XA (B*N rows) is free again; XB only needs B*L_1 rows
    }
    // 6. head2
    if (!part_body) {
        // no strided blocks: the central token x[:, N // 2] (u_u_t.py:411-413) = row N / 2 of every sequence, leading dimension N d_t
        ALoadPlain al{c.num_strided > 0 ? xa : w.X + (size_t)(N / 2) * dt, c.num_strided > 0 ? dt : N * dt, B, dt};
        EpBias ep{central_out, m->h2_b, 3 * J};
        Lh.gemm("head2", al, m->h2_wt, B, 3 * J, dt, ep);
    }
    // range guard (include/uu3d.h): non-finite outputs of an f16x3 forward set the model's sticky word
    if (c.precision == UU3D_PREC_F16X3 && !part_body) {
        Lh.begin("range_check", "range_check", 0.0, 4.0 * ((has_h1 ? (double)M * 3 * J : 0.0) + (double)B * 3 * J));
        hipLaunchKernelGGL(range_check_kernel, dim3(256), dim3(256), 0, Lh.stream, has_h1 ? full_out : central_out, has_h1 ? (long)M * 3 * J : 0L,
                           central_out, (long)B * 3 * J, m->d_range);
        Lh.end();
    }
    if (Lh.status != UU3D_OK) return Lh.status;
    return UU3D_OK;
}

int uu3d_range_status(uu3d_model* m, void* stream, int32_t* out_flag) {
    if (!m) return UU3D_ERR_INVALID_ARGUMENT;
    HIPCHK(m, hipSetDevice(m->device));
    int h = 0;
    hipLaunchKernelGGL(range_take_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, m->d_range, m->d_range + 1);      // read and clear: ONE atomic exchange
    HIPCHK(m, hipMemcpyAsync(&h, m->d_range + 1, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(m, hipStreamSynchronize((hipStream_t)stream));
    if (out_flag) *out_flag = h != 0;
    if (h != 0) return fail(m, UU3D_ERR_RANGE, "non-finite values in a forward's outputs: activations beyond the f16 range (65504) of the f16x3 products, or non-finite inputs "
                                               "(repeat the batch with UU3D_SCHEDULE_EXACT_F32 or build the model with precision f32)");
    return UU3D_OK;
}

int uu3d_mpjpe(const float* pred, const float* gt, int32_t B, int32_t J, int32_t root, double* out, void* stream) {
    if (!pred || !gt || !out || B < 1 || J < 1 || root < 0 || root >= J) return UU3D_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(mpjpe_kernel, dim3((B * J + 255) / 256), dim3(256), 0, (hipStream_t)stream, pred, gt, B, J, root, out);
    return hipGetLastError() == hipSuccess ? UU3D_OK : UU3D_ERR_HIP;
}

int uu3d_gather_windows(const float* poses, const int64_t* video_start, const int32_t* video_len, const uu3d_window* windows,
                        const int32_t* flip_order, int32_t B, int32_t N, int32_t J, int32_t Cc, int32_t pad_edge,
                        int32_t zero_masked, float* out, uint8_t* stride_mask, uint8_t* pad_mask, void* stream) {
    if (!poses || !video_start || !video_len || !windows || !out || !stride_mask || B < 1 || N < 1 || J < 1 || Cc < 1 || Cc > 4)
        return UU3D_ERR_INVALID_ARGUMENT;
    static_assert(sizeof(uu3d_window) == sizeof(WindowDesc), "descriptor layouts must agree");
    const long total = (long)B * N * J;
    hipLaunchKernelGGL(gather_windows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       poses, video_start, video_len, reinterpret_cast<const WindowDesc*>(windows), flip_order,
                       B, N, J, Cc, pad_edge, zero_masked, out, stride_mask, pad_mask);
    return hipGetLastError() == hipSuccess ? UU3D_OK : UU3D_ERR_HIP;
}

int uu3d_world_to_cam_2d(const float* world, const float* cams, int32_t B, int32_t N, int32_t J, float* cam3d, float* kp2d, void* stream) {
    if (!world || !cams || B < 1 || N < 1 || J < 1 || (!cam3d && !kp2d)) return UU3D_ERR_INVALID_ARGUMENT;
    const long per = (long)N * J, total = per * B;
    hipLaunchKernelGGL(world_to_cam_2d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, world, cams, per, total, cam3d, kp2d);
    return hipGetLastError() == hipSuccess ? UU3D_OK : UU3D_ERR_HIP;
}

int uu3d_set_schedule(uu3d_model* m, int32_t schedule) {
    if (!m) return UU3D_ERR_INVALID_ARGUMENT;
    if (schedule != UU3D_SCHEDULE_LATENCY && schedule != UU3D_SCHEDULE_THROUGHPUT) return fail(m, UU3D_ERR_INVALID_ARGUMENT, "unknown schedule");
    m->throughput = (schedule == UU3D_SCHEDULE_THROUGHPUT);
    return UU3D_OK;
}

int uu3d_set_profiling(uu3d_model* m, int32_t enabled) {
    if (!m) return UU3D_ERR_INVALID_ARGUMENT;
    m->profiling = enabled != 0;
    m->prof_used = 0;
    return UU3D_OK;
}

int uu3d_profile_read(uu3d_model* m, uu3d_profile_entry* out, int32_t capacity, int32_t* count) {
    if (!m || !count) return UU3D_ERR_INVALID_ARGUMENT;
    *count = (int32_t)m->prof_used;
    if (!out) return UU3D_OK;
    for (size_t i = 0; i < m->prof_used && (int32_t)i < capacity; ++i) {
        ProfRec& r = m->prof[i];
        HIPCHK(m, hipEventSynchronize(r.e1));
        float ms = 0.f;
        HIPCHK(m, hipEventElapsedTime(&ms, r.e0, r.e1));
        uu3d_profile_entry& e = out[i];
        std::memset(&e, 0, sizeof e);
        std::snprintf(e.name, sizeof e.name, "%s", r.name.c_str());
        std::snprintf(e.kernel, sizeof e.kernel, "%s", r.kernel.c_str());
        e.ms = ms; e.flops = r.flops; e.bytes = r.bytes;
    }
    return UU3D_OK;
}


// ---- training-step arithmetic without back-propagation (T1, T3, T4) -----------------------------
int uu3d_mpjpe_loss(const float* pred_full, const float* pred_central, const float* gt3d, int32_t B, int32_t N,
                    int32_t J, int32_t root, float w_center, float w_seq, int32_t batch_size_norm, float* loss_out,
                    float* grad_full, float* grad_central, float* scratch, void* stream_) {
    if (!pred_central || !gt3d || !loss_out || !scratch || B < 1 || N < 1 || J < 1 || root < 0 || root >= J ||
        batch_size_norm < 1)
        return UU3D_ERR_INVALID_ARGUMENT;
    if ((long)B * N * J > (1L << 30)) return UU3D_ERR_INVALID_ARGUMENT;
    hipStream_t stream = (hipStream_t)stream_;
    const float norm_cen = (float)batch_size_norm * (float)J;
    const float norm_seq = (float)batch_size_norm * (float)N * (float)J;
    const bool has_seq = pred_full != nullptr;
    // d loss / d dist: w / norm (fallback without sequence loss: (w_c + w_s) / norm_cen)
    const float gs_cen = (has_seq ? w_center : (w_center + w_seq)) / norm_cen;
    const float gs_seq = w_seq / norm_seq;
    hipLaunchKernelGGL(mpjpe_loss_stage1, dim3(kLossGrid), dim3(256), 0, stream, pred_full, pred_central, gt3d, B, N, J,
                       root, gs_seq, gs_cen, has_seq ? grad_full : nullptr, grad_central, scratch);
    hipLaunchKernelGGL(mpjpe_loss_stage2, dim3(1), dim3(64), 0, stream, scratch, norm_seq, norm_cen, w_center, w_seq,
                       has_seq ? 1 : 0, loss_out);
    return hipGetLastError() == hipSuccess ? UU3D_OK : UU3D_ERR_HIP;
}

int uu3d_adamw_update(float* var, float* m, float* v, float* vhat, const float* grad, int64_t n, float lr, float wd, float beta1,
                      float beta2, float epsilon, int64_t step, void* stream) {
    return uu3d_adamw_update_guarded(var, m, v, vhat, grad, n, lr, wd, beta1, beta2, epsilon, step, nullptr, stream);
}

int uu3d_adamw_update_guarded(float* var, float* m, float* v, float* vhat, const float* grad, int64_t n, float lr, float wd, float beta1,
                              float beta2, float epsilon, int64_t step, const uint32_t* skip, void* stream) {
    if (!var || !m || !v || !grad || n < 1 || step < 1) return UU3D_ERR_INVALID_ARGUMENT;
    if ((((uintptr_t)var | (uintptr_t)m | (uintptr_t)v | (uintptr_t)vhat | (uintptr_t)grad) & 15) != 0) return UU3D_ERR_INVALID_ARGUMENT;
    // float32 like the TF kernel: alpha = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
    // beta^t rounded once from double (libm powf and numpy's float32 power disagree by an ulp at some t, e.g. 0.9^4)
    const float b1p = (float)pow((double)beta1, (double)step), b2p = (float)pow((double)beta2, (double)step);
    const float alpha = lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
    const long long n4 = n >> 2;
    const int grid = (int)std::min<long long>(std::max<long long>((n4 + 255) / 256, 1), 256 * 32);
    if (vhat) hipLaunchKernelGGL(adamw_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, var, m, v, vhat, grad, (long long)n, wd, alpha,
                                 1.0f - beta1, 1.0f - beta2, epsilon, skip);
    else hipLaunchKernelGGL(adamw_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, var, m, v, vhat, grad, (long long)n, wd, alpha,
                            1.0f - beta1, 1.0f - beta2, epsilon, skip);
    return hipGetLastError() == hipSuccess ? UU3D_OK : UU3D_ERR_HIP;
}

int uu3d_ema_update(float* ema, const float* w, int64_t n, float decay, void* stream) {
    if (!ema || !w || n < 1) return UU3D_ERR_INVALID_ARGUMENT;
    const int grid = (int)std::min<long long>((n + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(ema_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, ema, w, (long long)n, 1.0f - decay);
    return hipGetLastError() == hipSuccess ? UU3D_OK : UU3D_ERR_HIP;
}

#include "uu3d_train_step.inc"

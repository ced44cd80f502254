// Round 6: the temporal chain on 64-row tiles, eight waves on 16-token panels (csrc/uu3d_tchain16.h), against float64, and its time.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops -I uplift-upsample-3dhpe_amd/csrc -o tools/tchain16_exp tools/tchain16_exp.hip
//   tools/tchain16_exp [M] [iters] [warm-up launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include <algorithm>
#include <chrono>
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_gemm.h"
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_gemm_h3.h"
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_gemm_panel.h"
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_gemm_panel8.h"
#include "../uplift-upsample-3dhpe_amd/csrc/uu3d_tchain16.h"
using namespace uu3d;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <class T> T* dev(const std::vector<T>& v) { T* p; CK(hipMalloc(&p, v.size() * sizeof(T))); CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return p; }
template <class T> T* devz(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); CK(hipMemset(p, 0, n * sizeof(T))); return p; }

struct Dense {                     // Keras layout W[k][n], bias[n]
    int K, N; std::vector<float> W, b;
    Dense(int K_, int N_, std::mt19937& rng) : K(K_), N(N_), W((size_t)K_ * N_), b(N_) {
        std::normal_distribution<float> nd(0.f, 1.f);
        for (auto& v : W) v = nd(rng) / sqrtf((float)K_);
        for (auto& v : b) v = 0.1f * nd(rng);
    }
    // stage chunks appended to `out`
    void pack(std::vector<_Float16>& out, int k0, int klen, bool natural) const {
        std::vector<_Float16> Bh((size_t)N * K), Bl((size_t)N * K);
        for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) { const float x = W[(size_t)k * N + n]; const _Float16 h = h3_hi(x); Bh[(size_t)n * K + k] = h; Bl[(size_t)n * K + k] = (_Float16)((x - (float)h) * H3_SCALE); }
        (void)klen;
        const size_t at = out.size(); out.resize(at + (size_t)(N / 32) * TC_CHUNK_HALFS);
        tchain16_pack_stage(Bh.data(), Bl.data(), N, K, k0, natural, out.data() + at);
    }
};

static void layer_norm(const std::vector<double>& x, const std::vector<float>& g, const std::vector<float>& b, std::vector<double>& y) {
    const int D = (int)x.size(); double s = 0, v = 0;
    for (double e : x) s += e; const double mean = s / D;
    for (double e : x) v += (e - mean) * (e - mean); const double rstd = 1.0 / sqrt(v / D + 1e-5);
    y.resize(D); for (int k = 0; k < D; ++k) y[k] = (x[k] - mean) * rstd * g[k] + b[k];
}
static void dense(const std::vector<double>& x, const Dense& d, std::vector<double>& y) {
    y.assign(d.N, 0.0);
    for (int n = 0; n < d.N; ++n) y[n] = d.b[n];
    for (int k = 0; k < d.K; ++k) { const double xv = x[k]; const float* w = &d.W[(size_t)k * d.N]; for (int n = 0; n < d.N; ++n) y[n] += xv * w[n]; }
}

static int warm = 3;
template <int FLAGS> void run(const char* tag, int M, int iters) {
    printf("=== %s (flags %d, %d chunks) M = %d\n", tag, FLAGS, tchain_chunks(FLAGS), M);
    std::mt19937 rng(7 + FLAGS); std::normal_distribution<float> nd(0.f, 1.f);
    const int D = 384, Hd = 768, period = 71;
    Dense wp(D, D, rng), w1(D, Hd, rng), w2(Hd, D, rng), wqkv(D, 3 * D, rng);
    std::vector<float> g2(D), be2(D), g1(D), be1(D), pe((size_t)period * D);
    for (int k = 0; k < D; ++k) { g2[k] = 1.f + 0.1f * nd(rng); be2[k] = 0.1f * nd(rng); g1[k] = 1.f + 0.1f * nd(rng); be1[k] = 0.1f * nd(rng); }
    for (auto& v : pe) v = 0.02f * nd(rng);
    std::vector<float> X((size_t)M * D), O((size_t)M * D);
    for (int r = 0; r < M; ++r) { const float off = 0.3f * nd(rng), sc = 0.5f + fabsf(nd(rng)); for (int k = 0; k < D; ++k) { X[(size_t)r * D + k] = off + sc * nd(rng); O[(size_t)r * D + k] = nd(rng); } }
    const int mt = (M + 63) / 64;
    std::vector<_Float16> Of(panel_a_halfs(mt * 64, D), (_Float16)0.f);
    // (the attention output arrives with the channels of a 16-slice in LANE order -- the order v had in the fragment-ordered q | k | v: tchain_qf_index)
    for (int r = 0; r < M; ++r) for (int k = 0; k < D; ++k) { const float x = O[(size_t)r * D + k]; const _Float16 h = h3_hi(x); const int w16 = k & 15; const size_t i = panel_a_index(r, (k & ~15) + 8 * ((w16 >> 2) & 1) + (((w16 >> 3) << 2) | (w16 & 3)), D); Of[i] = h; Of[i + 512] = (_Float16)((x - (float)h) * H3_SCALE); }
    // LayerNorm's affine part folded into the Dense layer behind it: W' = diag(gamma) W, b' = b + beta W
    auto fold = [&](const Dense& d, const std::vector<float>& g, const std::vector<float>& be) {
        Dense f = d;
        for (int n = 0; n < d.N; ++n) { double acc = d.b[n]; for (int k = 0; k < d.K; ++k) acc += (double)be[k] * d.W[(size_t)k * d.N + n]; f.b[n] = (float)acc; }
        for (int k = 0; k < d.K; ++k) for (int n = 0; n < d.N; ++n) f.W[(size_t)k * d.N + n] = g[k] * d.W[(size_t)k * d.N + n];
        return f;
    };
    Dense w1f = fold(w1, g2, be2), wqkvf = fold(wqkv, g1, be1);
    const float qscale = 1.44269504088896341f / sqrtf(48.f);
    for (int k = 0; k < D; ++k) for (int n = 0; n < D; ++n) wqkvf.W[(size_t)k * 3 * D + n] *= qscale;     // (the 64-row kernel: q's scale folded into wq, bq)
    for (int n = 0; n < D; ++n) wqkvf.b[n] *= qscale;
    std::vector<_Float16> W;
    if (FLAGS & TC_PROJ) wp.pack(W, 0, D, true);
    if (FLAGS & (TC_MLP | TC_FC1_PLANES)) w1f.pack(W, 0, D, false);
    if (FLAGS & TC_MLP) {
        w2.pack(W, 0, D, false); w2.pack(W, D, D, false);
        const size_t at = W.size() - (size_t)48 * TC_CHUNK_HALFS;          // W1 (24) | W2 half 0 | W2 half 1  ->  W1[0..11] | W2 half 0 | W1[12..23] | W2 half 1
        std::vector<_Float16> tmp(W.begin() + at, W.end());
        tchain16_reorder_mlp(tmp.data(), W.data() + at);
    }
    if (FLAGS & TC_QKV) wqkvf.pack(W, 0, D, false);
    if (W.size() != (size_t)tchain_chunks(FLAGS) * TC_CHUNK_HALFS) { printf("stream size mismatch\n"); exit(1); }

    std::vector<float> P(TCP_FLOATS);
    std::copy(wp.b.begin(), wp.b.end(), P.begin() + TCP_BP); std::copy(w1f.b.begin(), w1f.b.end(), P.begin() + TCP_B1);
    std::copy(w2.b.begin(), w2.b.end(), P.begin() + TCP_B2); std::copy(wqkvf.b.begin(), wqkvf.b.end(), P.begin() + TCP_BQKV);
    TChainArgs a{};
    a.M = M; a.m_tiles = mt; a.period = period; a.qscale = 1.44269504088896341f / sqrtf(48.f);
    a.Of = dev(Of); a.X = dev(X); a.XA = devz<float>((size_t)M * D); a.pe = dev(pe);
    a.W = dev(W); a.P = dev(P);
    a.Q = devz<_Float16>((size_t)mt * 2 * 72 * 2 * 512);
    a.H = devz<_Float16>((size_t)M * Hd * 2);
    // scratch = hidden fragments | xs | xas | trash; the launches that add into the residual stream find it there in lane-linear order
    std::vector<unsigned char> scr(tchain16_scratch_bytes(mt), 0);
    float* xs_h = reinterpret_cast<float*>(scr.data());
    float* xas_h = xs_h + (size_t)mt * T16_X_FLOATS_PER_TILE;
    for (int r = 0; r < M; ++r) for (int k = 0; k < D; ++k) { xs_h[tchain16_xs_index(r, k)] = X[(size_t)r * D + k]; xas_h[tchain16_xs_index(r, k)] = X[(size_t)r * D + k]; }
    a.scratch = dev(scr);
    unsigned char* scratch0 = dev(scr);
    auto kern = tchain16_kernel<FLAGS>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)T16_LDS_TOTAL));
    float* Xin = dev(X);
    auto launch = [&]() { hipLaunchKernelGGL(kern, dim3(mt), dim3(512), T16_LDS_TOTAL, 0, a); };
    launch(); CK(hipDeviceSynchronize());
    auto fetch_linear = [&](std::vector<float>& out, bool strided1) {        // the lane-linear tile copy -> row-major
        std::vector<unsigned char> sc(scr.size());
        CK(hipMemcpy(sc.data(), a.scratch, sc.size(), hipMemcpyDeviceToHost));
        const float* base = reinterpret_cast<const float*>(sc.data()) + (strided1 ? (size_t)mt * T16_X_FLOATS_PER_TILE : 0);
        for (int r = 0; r < M; ++r) for (int k = 0; k < D; ++k) out[(size_t)r * D + k] = base[tchain16_xs_index(r, k)];
    };

    std::vector<float> Xo((size_t)M * D), XAo((size_t)M * D); std::vector<_Float16> Q((size_t)mt * 2 * 72 * 2 * 512), Hp((size_t)M * Hd * 2);
    CK(hipMemcpy(Xo.data(), a.X, Xo.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(XAo.data(), a.XA, XAo.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(Q.data(), a.Q, Q.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(Hp.data(), a.H, Hp.size() * 2, hipMemcpyDeviceToHost));
    // where the launch leaves the residual stream: row-major x only when it ends the temporal stack (no QKV, or + pe); else the lane-linear tile;
    // the first strided block's launch: row-major xa
    if (FLAGS & TC_FC1_PLANES) Xo = XAo;
    else if ((FLAGS & TC_QKV) && !(FLAGS & TC_PE)) fetch_linear(Xo, false);
    if (FLAGS & TC_PE) fetch_linear(XAo, true);
    double ex = 0, exa = 0, eq = 0, eh = 0, sx = 0, sq = 0, sh = 0; size_t nan = 0; int rows = 0;
    for (int r = 0; r < M; r += (r < 160 || r > M - 160) ? 1 : 53) {
        ++rows;
        std::vector<double> x(D), o(D), y, n, hd, z;
        for (int k = 0; k < D; ++k) { x[k] = X[(size_t)r * D + k]; o[k] = O[(size_t)r * D + k]; }
        if (FLAGS & TC_PROJ) { dense(o, wp, y); for (int k = 0; k < D; ++k) x[k] += y[k]; }
        if (FLAGS & (TC_MLP | TC_FC1_PLANES)) {
            layer_norm(x, g2, be2, n); dense(n, w1, hd); for (auto& v : hd) v = std::max(v, 0.0);
            if (FLAGS & TC_FC1_PLANES) for (int k = 0; k < Hd; ++k) { const double got = (double)Hp[(size_t)r * Hd + k] + (double)Hp[(size_t)M * Hd + (size_t)r * Hd + k] / 2048.0; const double e = fabs(got - hd[k]); if (e != e) ++nan; eh = std::max(eh, e); sh = std::max(sh, fabs(hd[k])); }
            else { dense(hd, w2, z); for (int k = 0; k < D; ++k) x[k] += z[k]; }
        }
        for (int k = 0; k < D; ++k) { const double e = fabs(x[k] - Xo[(size_t)r * D + k]); if (e != e) ++nan; ex = std::max(ex, e); sx = std::max(sx, fabs(x[k]));
            if (e > 2e-5 && getenv("DEBUG_X")) { static int shown = 0; if (shown++ < 40) printf("      x[%d][%d] got %.6f want %.6f (err %.2e)\n", r, k, Xo[(size_t)r * D + k], x[k], e); } }
        if (FLAGS & TC_QKV) {
            if (FLAGS & TC_PE) for (int k = 0; k < D; ++k) { x[k] += pe[(size_t)(r % period) * D + k]; const double e = fabs(x[k] - XAo[(size_t)r * D + k]); if (e != e) ++nan; exa = std::max(exa, e); }
            layer_norm(x, g1, be1, n); dense(n, wqkv, z);
            for (int k = 0; k < 3 * D; ++k) { const double want = k < D ? z[k] * a.qscale : z[k]; const double got = (double)Q[tchain_qf_index(r, k, 0)] + (double)Q[tchain_qf_index(r, k, 1)] / 2048.0;
                const double e = fabs(got - want); if (e != e) ++nan; eq = std::max(eq, e); sq = std::max(sq, fabs(want)); }
        }
    }
    printf("    %d rows vs float64: x %.3e (scale %.1f)  xa %.3e  qkv %.3e (scale %.1f)  fc1 planes %.3e (scale %.1f)  NaN %zu\n", rows, ex, sx, exa, eq, sq, eh, sh, nan);

    // determinism + time (x is updated in place: restore it in front of every launch that is checked, not in the timed ones)
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemcpy(a.X, Xin, Xo.size() * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(a.scratch, scratch0, scr.size(), hipMemcpyDeviceToDevice)); launch(); CK(hipDeviceSynchronize());
    std::vector<float> X2((size_t)M * D); std::vector<_Float16> Q2(Q.size());
    CK(hipMemcpy(X2.data(), (FLAGS & TC_FC1_PLANES) ? a.XA : a.X, X2.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(Q2.data(), a.Q, Q2.size() * 2, hipMemcpyDeviceToHost));
    if (!(FLAGS & TC_FC1_PLANES) && (FLAGS & TC_QKV) && !(FLAGS & TC_PE)) fetch_linear(X2, false);
    size_t diff = 0; for (size_t i = 0; i < X2.size(); ++i) diff += (X2[i] != Xo[i]); for (size_t i = 0; i < Q2.size(); ++i) diff += ((float)Q2[i] != (float)Q[i]);
    printf("    second run: %zu differing values\n", diff);
    for (int i = 0; i < warm; ++i) launch();
    CK(hipEventRecord(e0)); for (int i = 0; i < iters; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = 2.0 * M * (double)D * (((FLAGS & TC_PROJ) ? D : 0) + ((FLAGS & TC_MLP) ? 2 * Hd : 0) + ((FLAGS & TC_FC1_PLANES) ? Hd : 0) + ((FLAGS & TC_QKV) ? 3 * D : 0));
    printf("    %.2f us per launch (%d workgroups), %.1f TFLOP/s algorithmic\n", ms / iters * 1e3, mt, flop / (ms / iters * 1e-3) * 1e-12);
    if (const char* mixs = getenv("MIX")) {      // MIX=n: n launches side by side on n streams, each with its OWN copy of the weight stream and buffers (what a pipeline of forwards in different blocks does to L2)
        const int nmix = atoi(mixs);
        std::vector<TChainArgs> as(nmix, a); std::vector<hipStream_t> st(nmix);
        for (int i = 0; i < nmix; ++i) {
            CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
            if (i == 0) continue;
            as[i].W = dev(W); as[i].scratch = dev(scr); as[i].X = dev(X); as[i].Of = dev(Of);
            _Float16* q; CK(hipMalloc(&q, Q.size() * 2)); as[i].Q = q;
        }
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int it = 0; it < iters; ++it) for (int i = 0; i < nmix; ++i) hipLaunchKernelGGL(kern, dim3(mt), dim3(512), T16_LDS_TOTAL, st[i], as[i]);
        CK(hipDeviceSynchronize());
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("    MIX %d streams x %d launches of %d workgroups, own weights each: %.2f us per launch-equivalent (%.2f us per %d launches side by side)\n", nmix, iters, mt, us / (iters * nmix), us / iters, nmix);
    }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 9088, iters = argc > 2 ? atoi(argv[2]) : 20;
    if (argc > 3) warm = atoi(argv[3]);
    if (getenv("ONLY_MID")) { run<TC_PROJ | TC_MLP | TC_QKV>("mid: proj + MLP + LN1 + QKV", M, iters); return 0; }      // (tools/power_tchain16.sh: one stage set, long enough to sample power)
    run<TC_QKV>("first: LN1 + QKV", M, iters);
    run<TC_PROJ | TC_MLP | TC_QKV>("mid: proj + MLP + LN1 + QKV", M, iters);
    run<TC_PROJ | TC_MLP | TC_QKV | TC_PE>("mid -> strided 1 (pe)", M, iters);
    run<TC_PROJ | TC_FC1_PLANES>("last: proj + fc1 planes", M, iters);
    run<TC_PROJ | TC_MLP>("end: proj + MLP", M, iters);
    if (M == 9088) { run<TC_PROJ | TC_MLP | TC_QKV>("mid, ragged", 1000, 5); run<TC_QKV>("first, ragged", 71 * 3, 5); }
    return 0;
}

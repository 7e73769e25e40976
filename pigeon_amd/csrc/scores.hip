// Per-image preference / alignment scores of the scoring stage (see include/pigeon_hip.h: pgn_clip_scores).
// Replaces, for one batch of CLIP image features, the arithmetic of Evaluation/eval_utils.py:121-129
// (CLIPScore.calculate_clip_score: L2-normalise both sides, 100 * cosine similarity) and :540-548
// (evaluate_personalization_given_data_sim against cached history features) together with the mean over the
// 5 history images of Evaluation/cal_scores.py:178-179.  One wavefront per image: every feature row is read once,
// all dot products and norms are fp32 wave reductions, two floats are written.  HBM / launch-latency bound.
#include "pgn_common.h"

namespace {

constexpr int SC_THREADS = 256;
constexpr int SC_MAX_HIST = 16;

__device__ __forceinline__ float sc_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ float sc_ld(const void* p, int dt, long long i) {
    if (dt == PGN_F32) return reinterpret_cast<const float*>(p)[i];
    const unsigned short h = reinterpret_cast<const unsigned short*>(p)[i];
    return dt == PGN_BF16 ? Cvt<PGN_BF16>::to_f32(h) : Cvt<PGN_F16>::to_f32(h);
}

// F.cosine_similarity(a, b) = a.b / max(|a| |b|, eps) with eps = 1e-8; the explicit normalisations in front of it
// (eval_utils.py:120-123) change nothing beyond fp32 rounding.
__device__ __forceinline__ float sc_cos100(float dot, float na2, float nb2) {
    return 100.0f * dot / fmaxf(sqrtf(na2) * sqrtf(nb2), 1e-8f);
}

__global__ __launch_bounds__(SC_THREADS) void pgn_clip_scores_kernel(
    const void* __restrict__ img, int img_dt, long long img_ld, const void* __restrict__ txt, int txt_dt,
    long long txt_ld, const float* __restrict__ hist, long long hist_ld, int n_hist, float* __restrict__ tgt_score,
    float* __restrict__ hist_score, long long M, int D) {
    const int lane = threadIdx.x & 63;
    const long long m = (long long)blockIdx.x * (SC_THREADS / 64) + (threadIdx.x >> 6);
    if (m >= M) return;   // whole wavefronts leave together; no barrier below
    float ii = 0.f, it = 0.f, tt = 0.f;
    float ih[SC_MAX_HIST], hh[SC_MAX_HIST];
#pragma unroll
    for (int k = 0; k < SC_MAX_HIST; ++k) ih[k] = hh[k] = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float a = sc_ld(img, img_dt, m * img_ld + d);
        ii += a * a;
        if (txt != nullptr) {
            const float t = sc_ld(txt, txt_dt, m * txt_ld + d);
            it += a * t;
            tt += t * t;
        }
#pragma unroll
        for (int k = 0; k < SC_MAX_HIST; ++k)
            if (k < n_hist) {
                const float h = hist[m * hist_ld + (long long)k * D + d];
                ih[k] += a * h;
                hh[k] += h * h;
            }
    }
    ii = sc_wave_sum(ii);
    if (txt != nullptr) {
        it = sc_wave_sum(it);
        tt = sc_wave_sum(tt);
        if (lane == 0 && tgt_score != nullptr) tgt_score[m] = sc_cos100(it, ii, tt);
    }
    if (n_hist > 0) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < SC_MAX_HIST; ++k)
            if (k < n_hist) acc += sc_cos100(sc_wave_sum(ih[k]), ii, sc_wave_sum(hh[k]));
        if (lane == 0 && hist_score != nullptr) hist_score[m] = acc / (float)n_hist;
    }
}

}  // namespace

extern "C" int pgn_clip_scores(const void* img_feat, int img_dtype, int64_t img_ld, const void* txt_feat, int txt_dtype,
                               int64_t txt_ld, const float* hist_feat, int64_t hist_ld, int n_hist, float* tgt_score,
                               float* hist_score, int64_t M, int D, void* stream) {
    if (img_feat == nullptr || M <= 0 || D <= 0) return PGN_ERR_INVALID;
    if (pgn_dtype_size(img_dtype) == 0 || img_ld < D) return PGN_ERR_INVALID;
    if (txt_feat != nullptr && (pgn_dtype_size(txt_dtype) == 0 || txt_ld < D || tgt_score == nullptr)) return PGN_ERR_INVALID;
    if (n_hist < 0 || (n_hist > 0 && (hist_feat == nullptr || hist_score == nullptr || hist_ld < (int64_t)n_hist * D)))
        return PGN_ERR_INVALID;
    if (txt_feat == nullptr && n_hist == 0) return PGN_ERR_INVALID;   // nothing to compute
    if (n_hist > SC_MAX_HIST) return PGN_ERR_UNSUPPORTED;
    const long long blocks = (M + SC_THREADS / 64 - 1) / (SC_THREADS / 64);
    if (blocks > 0x7fffffffLL) return PGN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pgn_clip_scores_kernel, dim3((unsigned)blocks), dim3(SC_THREADS), 0, (hipStream_t)stream, img_feat,
                       img_dtype, (long long)img_ld, txt_feat, txt_dtype, (long long)txt_ld, hist_feat, (long long)hist_ld,
                       n_hist, tgt_score, hist_score, (long long)M, D);
    PGN_LAUNCH_CHECK();
    return PGN_OK;
}

// Fused CFG combine + scheduler step kernels (see include/pigeon_hip.h).
// Reference: Pigeon/models/lavit_for_pigeon.py:747-748,760-764 and diffusers 0.27.2
// DDIMScheduler.step / EulerDiscreteScheduler.step (third-party, restated in oracle/pipeline_ref.py: DDIMRef / EulerRef, and oracle/ops_ref.py: cfg_ddim_step_ref / cfg_euler_step_ref).
//
// HBM-bound elementwise: one read of eps_u, eps_t, x and one write of x' per element (12 B with bf16
// eps + fp32 latents).  4 elements per lane per access (8-byte bf16x4 / 16-byte float4), grid-stride.
//
// This file is compiled with -ffp-contract=off: when PGN_STEP_FUSED_FP32 is clear every operation is a
// separately rounded IEEE fp32 op plus explicit roundings to the eps dtype at the points where torch
// materialises a 16-bit tensor, so the result is BIT-EXACT with the reference arithmetic.
#include "pgn_common.h"

#pragma clang fp contract(off)

namespace {

struct DdimParams {
    float g, sa_t, sb_t, sa_prev, dir;
    float clip;
    int pred;
    unsigned flags;
    int has_uncond;
};

struct EulerParams {
    float g, sigma, dt, c_out, s2p1;
    int pred;
    unsigned flags;
    int has_uncond;
};

// Round an fp32 value to DT and back.  The empty asm makes the fp32 value opaque first: otherwise hipcc folds
// `(half)(a * b)` into v_fma_mixlo_f16, which rounds ONCE to f16, whereas torch materialises the fp32 product
// (opmath) and then casts - a double rounding that differs on exact-midpoint products.
template <int DT>
__device__ __forceinline__ float rnd(float x) {
    if constexpr (DT == PGN_F32) return x;
    asm volatile("" : "+v"(x));
    return Cvt<DT>::to_f32(Cvt<DT>::from_f32(x));
}

template <int DT>
__device__ __forceinline__ void load4(const void* p, int64_t i, float (&v)[4]) {
    if constexpr (DT == PGN_F32) {
        const float4 t = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        const u16x4_t t = *reinterpret_cast<const u16x4_t*>(reinterpret_cast<const unsigned short*>(p) + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = Cvt<DT>::to_f32(t[j]);
    }
}
template <int DT>
__device__ __forceinline__ void store4v(void* p, int64_t i, const float (&v)[4]) {
    if constexpr (DT == PGN_F32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p) + i) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        u16x4_t t;
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = Cvt<DT>::from_f32(v[j]);
        *reinterpret_cast<u16x4_t*>(reinterpret_cast<unsigned short*>(p) + i) = t;
    }
}
template <int DT>
__device__ __forceinline__ float load1(const void* p, int64_t i) {
    return Cvt<DT>::to_f32(reinterpret_cast<const typename Cvt<DT>::storage*>(p)[i]);
}
template <int DT>
__device__ __forceinline__ void store1(void* p, int64_t i, float v) {
    reinterpret_cast<typename Cvt<DT>::storage*>(p)[i] = Cvt<DT>::from_f32(v);
}

// CFG combine with torch's rounding points: each of (et - eu), g*(.), eu + (.) is a tensor in the eps dtype.
template <int EDT, bool FUSED>
__device__ __forceinline__ float cfg_combine(float eu, float et, float g, int has_uncond) {
    if (!has_uncond) return et;
    if constexpr (FUSED) {
        return eu + g * (et - eu);
    } else {
        const float d = rnd<EDT>(et - eu);
        const float gd = rnd<EDT>(g * d);
        return rnd<EDT>(eu + gd);
    }
}

// One DDIM element.  EDT: eps dtype, PDT: promoted dtype (= sample dtype).
template <int EDT, int PDT, bool FUSED>
__device__ __forceinline__ void ddim_elem(float eu, float et, float x, const DdimParams& p, float& prev, float& x0o) {
    const float e = cfg_combine<EDT, FUSED>(eu, et, p.g, p.has_uncond);
    float x0, pe;
    if constexpr (FUSED) {
        if (p.pred == PGN_PRED_EPSILON) {
            x0 = (x - p.sb_t * e) / p.sa_t;
            pe = e;
        } else if (p.pred == PGN_PRED_SAMPLE) {
            x0 = e;
            pe = (x - p.sa_t * x0) / p.sb_t;
        } else {
            x0 = p.sa_t * x - p.sb_t * e;
            pe = p.sa_t * e + p.sb_t * x;
        }
        if (p.flags & PGN_STEP_CLIP_SAMPLE) {
            x0 = fminf(fmaxf(x0, -p.clip), p.clip);
            if (p.flags & PGN_STEP_USE_CLIPPED_OUTPUT) pe = (x - p.sa_t * x0) / p.sb_t;
        }
        prev = p.sa_prev * x0 + p.dir * pe;
        x0o = x0;
    } else {
        // dtype of pred_epsilon / pred_original_sample as torch would produce them
        bool pe_is_e = false, x0_is_e = false;
        if (p.pred == PGN_PRED_EPSILON) {
            const float t1 = rnd<EDT>(p.sb_t * e);
            x0 = rnd<PDT>(rnd<PDT>(x - t1) / p.sa_t);
            pe = e;
            pe_is_e = true;
        } else if (p.pred == PGN_PRED_SAMPLE) {
            x0 = e;
            x0_is_e = true;
            const float a = rnd<EDT>(p.sa_t * x0);
            pe = rnd<PDT>(rnd<PDT>(x - a) / p.sb_t);
        } else {
            x0 = rnd<PDT>(rnd<PDT>(p.sa_t * x) - rnd<EDT>(p.sb_t * e));
            pe = rnd<PDT>(rnd<EDT>(p.sa_t * e) + rnd<PDT>(p.sb_t * x));
        }
        if (p.flags & PGN_STEP_CLIP_SAMPLE) {
            x0 = fminf(fmaxf(x0, -p.clip), p.clip);
            if (p.flags & PGN_STEP_USE_CLIPPED_OUTPUT) {
                const float a = x0_is_e ? rnd<EDT>(p.sa_t * x0) : rnd<PDT>(p.sa_t * x0);
                pe = rnd<PDT>(rnd<PDT>(x - a) / p.sb_t);
                pe_is_e = false;
            }
        }
        const float dirv = pe_is_e ? rnd<EDT>(p.dir * pe) : rnd<PDT>(p.dir * pe);
        const float ax0 = x0_is_e ? rnd<EDT>(p.sa_prev * x0) : rnd<PDT>(p.sa_prev * x0);
        prev = rnd<PDT>(ax0 + dirv);
        x0o = x0;
    }
}

template <int EDT, int PDT, bool FUSED>
__global__ __launch_bounds__(256) void pgn_ddim_kernel(const void* __restrict__ eu_p, const void* __restrict__ et_p,
                                                       const void* __restrict__ x_p, void* __restrict__ prev_p,
                                                       void* __restrict__ x0_p, int64_t numel, DdimParams p) {
    const int64_t nvec = numel >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        const int64_t i = v << 2;
        float eu[4] = {0.f, 0.f, 0.f, 0.f}, et[4], x[4], prev[4], x0[4];
        if (p.has_uncond) load4<EDT>(eu_p, i, eu);
        load4<EDT>(et_p, i, et);
        load4<PDT>(x_p, i, x);
#pragma unroll
        for (int j = 0; j < 4; ++j) ddim_elem<EDT, PDT, FUSED>(eu[j], et[j], x[j], p, prev[j], x0[j]);
        store4v<PDT>(prev_p, i, prev);
        if (x0_p != nullptr) store4v<PDT>(x0_p, i, x0);
    }
    // scalar tail (numel % 4)
    const int64_t tail0 = nvec << 2;
    const int64_t t = tail0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < numel) {
        const float eu = p.has_uncond ? load1<EDT>(eu_p, t) : 0.f;
        float prev, x0;
        ddim_elem<EDT, PDT, FUSED>(eu, load1<EDT>(et_p, t), load1<PDT>(x_p, t), p, prev, x0);
        store1<PDT>(prev_p, t, prev);
        if (x0_p != nullptr) store1<PDT>(x0_p, t, x0);
    }
}

// One Euler element: sample is upcast to fp32 (diffusers 0.27.2), outputs rounded to ODT by the store.
template <int EDT, bool FUSED>
__device__ __forceinline__ void euler_elem(float eu, float et, float x, const EulerParams& p, float& prev, float& x0o) {
    const float e = cfg_combine<EDT, FUSED>(eu, et, p.g, p.has_uncond);
    float x0;
    if constexpr (FUSED) {
        if (p.pred == PGN_PRED_EPSILON)
            x0 = x - p.sigma * e;
        else if (p.pred == PGN_PRED_SAMPLE)
            x0 = e;
        else
            x0 = e * p.c_out + x / p.s2p1;
    } else {
        if (p.pred == PGN_PRED_EPSILON)
            x0 = x - rnd<EDT>(p.sigma * e);
        else if (p.pred == PGN_PRED_SAMPLE)
            x0 = e;
        else
            x0 = rnd<EDT>(e * p.c_out) + x / p.s2p1;
    }
    const float d = (x - x0) / p.sigma;
    prev = x + d * p.dt;
    x0o = x0;
}

template <int EDT, int SDT, int ODT, bool FUSED>
__global__ __launch_bounds__(256) void pgn_euler_kernel(const void* __restrict__ eu_p, const void* __restrict__ et_p,
                                                        const void* __restrict__ x_p, void* __restrict__ prev_p,
                                                        void* __restrict__ x0_p, int64_t numel, EulerParams p) {
    const int64_t nvec = numel >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        const int64_t i = v << 2;
        float eu[4] = {0.f, 0.f, 0.f, 0.f}, et[4], x[4], prev[4], x0[4];
        if (p.has_uncond) load4<EDT>(eu_p, i, eu);
        load4<EDT>(et_p, i, et);
        load4<SDT>(x_p, i, x);
#pragma unroll
        for (int j = 0; j < 4; ++j) euler_elem<EDT, FUSED>(eu[j], et[j], x[j], p, prev[j], x0[j]);
        store4v<ODT>(prev_p, i, prev);
        if (x0_p != nullptr) store4v<ODT>(x0_p, i, x0);
    }
    const int64_t tail0 = nvec << 2;
    const int64_t t = tail0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < numel) {
        const float eu = p.has_uncond ? load1<EDT>(eu_p, t) : 0.f;
        float prev, x0;
        euler_elem<EDT, FUSED>(eu, load1<EDT>(et_p, t), load1<SDT>(x_p, t), p, prev, x0);
        store1<ODT>(prev_p, t, prev);
        if (x0_p != nullptr) store1<ODT>(x0_p, t, x0);
    }
}

template <int IDT, int ODT>
__global__ __launch_bounds__(256) void pgn_expand_kernel(const void* __restrict__ in_p, void* __restrict__ out_p,
                                                         int64_t numel, float divisor) {
    const int64_t nvec = numel >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        const int64_t i = v << 2;
        float x[4];
        load4<IDT>(in_p, i, x);
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = rnd<IDT>(x[j] / divisor);
        store4v<ODT>(out_p, i, x);
        store4v<ODT>(out_p, i + numel, x);
    }
    const int64_t t = (nvec << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < numel) {
        const float x = rnd<IDT>(load1<IDT>(in_p, t) / divisor);
        store1<ODT>(out_p, t, x);
        store1<ODT>(out_p, t + numel, x);
    }
}

inline int step_grid(int64_t numel) {
    int64_t nvec = (numel + 3) >> 2;
    int64_t blocks = (nvec + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;  // 256 CUs x 8 blocks, grid-stride the rest
    return (int)blocks;
}

inline bool ok16(int dt) { return dt == PGN_BF16 || dt == PGN_F16; }

}  // namespace

extern "C" int pgn_cfg_ddim_step(const void* eps_uncond, const void* eps_text, int eps_dtype, const void* sample,
                                 void* prev_sample, void* pred_x0, int sample_dtype, int64_t numel, float guidance,
                                 float sqrt_alpha_t, float sqrt_beta_t, float sqrt_alpha_prev, float dir_coef,
                                 int prediction_type, unsigned flags, float clip_range, void* stream) {
    if (numel == 0) return PGN_OK;
    if (eps_text == nullptr || sample == nullptr || prev_sample == nullptr || numel < 0) return PGN_ERR_INVALID;
    if (pgn_dtype_size(eps_dtype) == 0 || pgn_dtype_size(sample_dtype) == 0) return PGN_ERR_INVALID;
    if (prediction_type < PGN_PRED_EPSILON || prediction_type > PGN_PRED_V) return PGN_ERR_INVALID;
    if (!(sample_dtype == PGN_F32 || sample_dtype == eps_dtype)) return PGN_ERR_UNSUPPORTED;
    const size_t ea = eps_dtype == PGN_F32 ? 16 : 8, sa = sample_dtype == PGN_F32 ? 16 : 8;
    if (!pgn_aligned(eps_uncond, ea) || !pgn_aligned(eps_text, ea) || !pgn_aligned(sample, sa) ||
        !pgn_aligned(prev_sample, sa) || !pgn_aligned(pred_x0, sa))
        return PGN_ERR_INVALID;
    DdimParams p;
    p.g = guidance; p.sa_t = sqrt_alpha_t; p.sb_t = sqrt_beta_t; p.sa_prev = sqrt_alpha_prev; p.dir = dir_coef;
    p.clip = clip_range; p.pred = prediction_type; p.flags = flags; p.has_uncond = eps_uncond != nullptr;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(step_grid(numel)), block(256);
    const bool fused = (flags & PGN_STEP_FUSED_FP32) != 0;
#define PGN_DDIM_LAUNCH(E, P)                                                                                     \
    do {                                                                                                          \
        if (fused)                                                                                                \
            hipLaunchKernelGGL((pgn_ddim_kernel<E, P, true>), grid, block, 0, s, eps_uncond, eps_text, sample,     \
                               prev_sample, pred_x0, numel, p);                                                   \
        else                                                                                                      \
            hipLaunchKernelGGL((pgn_ddim_kernel<E, P, false>), grid, block, 0, s, eps_uncond, eps_text, sample,    \
                               prev_sample, pred_x0, numel, p);                                                   \
    } while (0)
    if (eps_dtype == PGN_BF16 && sample_dtype == PGN_F32) PGN_DDIM_LAUNCH(PGN_BF16, PGN_F32);
    else if (eps_dtype == PGN_F16 && sample_dtype == PGN_F32) PGN_DDIM_LAUNCH(PGN_F16, PGN_F32);
    else if (eps_dtype == PGN_F32 && sample_dtype == PGN_F32) PGN_DDIM_LAUNCH(PGN_F32, PGN_F32);
    else if (eps_dtype == PGN_BF16 && sample_dtype == PGN_BF16) PGN_DDIM_LAUNCH(PGN_BF16, PGN_BF16);
    else if (eps_dtype == PGN_F16 && sample_dtype == PGN_F16) PGN_DDIM_LAUNCH(PGN_F16, PGN_F16);
    else return PGN_ERR_UNSUPPORTED;
#undef PGN_DDIM_LAUNCH
    PGN_LAUNCH_CHECK();
    return PGN_OK;
}

extern "C" int pgn_cfg_euler_step(const void* eps_uncond, const void* eps_text, int eps_dtype, const void* sample,
                                  int sample_dtype, void* prev_sample, void* pred_x0, int out_dtype, int64_t numel,
                                  float guidance, float sigma, float sigma_next, int prediction_type, unsigned flags,
                                  void* stream) {
    if (numel == 0) return PGN_OK;
    if (eps_text == nullptr || sample == nullptr || prev_sample == nullptr || numel < 0) return PGN_ERR_INVALID;
    if (pgn_dtype_size(eps_dtype) == 0 || pgn_dtype_size(sample_dtype) == 0 || pgn_dtype_size(out_dtype) == 0)
        return PGN_ERR_INVALID;
    if (prediction_type < PGN_PRED_EPSILON || prediction_type > PGN_PRED_V) return PGN_ERR_INVALID;
    if (!(sigma > 0.f)) return PGN_ERR_INVALID;
    if (ok16(eps_dtype) && ok16(sample_dtype) && eps_dtype != sample_dtype) return PGN_ERR_UNSUPPORTED;
    if (ok16(out_dtype) && ok16(eps_dtype) && eps_dtype != out_dtype) return PGN_ERR_UNSUPPORTED;
    const size_t ea = eps_dtype == PGN_F32 ? 16 : 8, sa = sample_dtype == PGN_F32 ? 16 : 8,
                 oa = out_dtype == PGN_F32 ? 16 : 8;
    if (!pgn_aligned(eps_uncond, ea) || !pgn_aligned(eps_text, ea) || !pgn_aligned(sample, sa) ||
        !pgn_aligned(prev_sample, oa) || !pgn_aligned(pred_x0, oa))
        return PGN_ERR_INVALID;
    EulerParams p;
    p.g = guidance; p.sigma = sigma; p.dt = sigma_next - sigma;
    p.s2p1 = sigma * sigma + 1.0f;
    p.c_out = -sigma / sqrtf(p.s2p1);
    p.pred = prediction_type; p.flags = flags; p.has_uncond = eps_uncond != nullptr;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(step_grid(numel)), block(256);
    const bool fused = (flags & PGN_STEP_FUSED_FP32) != 0;
#define PGN_EULER_LAUNCH(E, S, O)                                                                                   \
    do {                                                                                                            \
        if (fused)                                                                                                  \
            hipLaunchKernelGGL((pgn_euler_kernel<E, S, O, true>), grid, block, 0, s, eps_uncond, eps_text, sample,   \
                               prev_sample, pred_x0, numel, p);                                                     \
        else                                                                                                        \
            hipLaunchKernelGGL((pgn_euler_kernel<E, S, O, false>), grid, block, 0, s, eps_uncond, eps_text, sample,  \
                               prev_sample, pred_x0, numel, p);                                                     \
    } while (0)
#define PGN_EULER_SO(E)                                                                       \
    do {                                                                                      \
        if (sample_dtype == PGN_F32 && out_dtype == PGN_F32) PGN_EULER_LAUNCH(E, PGN_F32, PGN_F32); \
        else if (sample_dtype == PGN_F32 && out_dtype == E) PGN_EULER_LAUNCH(E, PGN_F32, E);  \
        else if (sample_dtype == E && out_dtype == E) PGN_EULER_LAUNCH(E, E, E);              \
        else if (sample_dtype == E && out_dtype == PGN_F32) PGN_EULER_LAUNCH(E, E, PGN_F32);  \
        else return PGN_ERR_UNSUPPORTED;                                                      \
    } while (0)
    if (eps_dtype == PGN_BF16) PGN_EULER_SO(PGN_BF16);
    else if (eps_dtype == PGN_F16) PGN_EULER_SO(PGN_F16);
    else {
        if (sample_dtype == PGN_F32 && out_dtype == PGN_F32) PGN_EULER_LAUNCH(PGN_F32, PGN_F32, PGN_F32);
        else return PGN_ERR_UNSUPPORTED;
    }
#undef PGN_EULER_SO
#undef PGN_EULER_LAUNCH
    PGN_LAUNCH_CHECK();
    return PGN_OK;
}

extern "C" int pgn_cfg_expand_scale(const void* latents, int in_dtype, void* out, int out_dtype, int64_t numel,
                                    float divisor, void* stream) {
    if (numel == 0) return PGN_OK;
    if (latents == nullptr || out == nullptr || numel < 0) return PGN_ERR_INVALID;
    if (pgn_dtype_size(in_dtype) == 0 || pgn_dtype_size(out_dtype) == 0) return PGN_ERR_INVALID;
    if (!(divisor > 0.f)) return PGN_ERR_INVALID;
    if ((numel & 3) != 0) {
        // second half starts at out + numel: keep its vector stores aligned
        return PGN_ERR_UNSUPPORTED;
    }
    if (!pgn_aligned(latents, in_dtype == PGN_F32 ? 16 : 8) || !pgn_aligned(out, out_dtype == PGN_F32 ? 16 : 8))
        return PGN_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(step_grid(numel)), block(256);
#define PGN_EXP_LAUNCH(I, O) \
    hipLaunchKernelGGL((pgn_expand_kernel<I, O>), grid, block, 0, s, latents, out, numel, divisor)
    if (in_dtype == PGN_F32 && out_dtype == PGN_F32) PGN_EXP_LAUNCH(PGN_F32, PGN_F32);
    else if (in_dtype == PGN_F32 && out_dtype == PGN_BF16) PGN_EXP_LAUNCH(PGN_F32, PGN_BF16);
    else if (in_dtype == PGN_F32 && out_dtype == PGN_F16) PGN_EXP_LAUNCH(PGN_F32, PGN_F16);
    else if (in_dtype == PGN_BF16 && out_dtype == PGN_BF16) PGN_EXP_LAUNCH(PGN_BF16, PGN_BF16);
    else if (in_dtype == PGN_F16 && out_dtype == PGN_F16) PGN_EXP_LAUNCH(PGN_F16, PGN_F16);
    else return PGN_ERR_UNSUPPORTED;
#undef PGN_EXP_LAUNCH
    PGN_LAUNCH_CHECK();
    return PGN_OK;
}

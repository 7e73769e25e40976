// GroupNorm (+SiLU), channels-last: x[n, p, c] with p = h*W + w (see include/pigeon_hip.h: pgn_groupnorm_silu_nhwc).
// Same op as groupnorm_silu.hip for activations kept in the layout the MIOpen / CK convolution kernels compute in
// (torch memory_format channels_last): no NCHW<->NHWC transposes around every convolution, and the [B, HW, C] view the
// transformer blocks want is free.
//
// A (sample, group) is now a strided set (C/G channels out of every C), so a workgroup streams whole rows instead:
//   thread -> (row lane r, 16-byte channel chunk cc); a wavefront reads 1 KiB of consecutive bytes per access.
//   kernel 1 (stats)    per (sample, row chunk): per-channel pivot-shifted sum / sum of squares, reduced over the row
//                       lanes through LDS, written to the workspace  [N][S][C] float2
//   kernel 2 (finalize) per (sample, group): fixed-order reduction over chunks and channels -> mean, rstd  [N][G] float2
//                       (per-CHANNEL pivots p_c = median of x[n, {0, HW/2, HW-1}, c]:
//                        sum_x (x-mu)^2 = sum_c q_c + 2 (p_c-mu) s_c + HW (p_c-mu)^2)
//   kernel 3 (apply)    same mapping as kernel 1: (x - mean) * rstd * gamma + beta, SiLU, store
// Two reads + one write of the tensor, deterministic, no atomics.  Measured fabric traffic inside the bench is 1.5x the
// algorithmic 1R + 1W (142 MB statistics pass + 282 MB apply pass per call at the launch-weighted shape): the apply
// pass re-reads the whole tensor through the fabric; TCC fetch counters include Infinity-Cache hits, so how much of
// that second read HBM itself serves is not measurable here.
#include "pgn_common.h"

#include <cstdint>
#include <cstdlib>
#include <mutex>

namespace {

constexpr int NH_MAX_THREADS = 1024;

struct NhArgs {
    const void* x;   // channels [0, C1)
    const void* x2;  // channels [C1, C) of the same rows (channel concatenation read in place), or nullptr: C1 == C
    int C1;
    void* y;
    const void* gamma;
    const void* beta;
    const void* addnc;
    int adt, add_dt;
    int C, HW, G, cpg;
    int C8;          // 16-byte chunks per row
    int R;           // row lanes per workgroup
    int S;           // row chunks per sample
    int rows;        // rows per chunk
    float eps;
    int act;
    int reverse;     // apply pass: walk the workgroups from the last row chunk of the last sample to the first
    float2* part;    // [N][S][C]
    float2* stats;   // [N][G]
    // what the finalize launch reads: `part` as [N][partS][partC] for channels [0, partC); channels [partC, C) from
    // `part2` as [N][part2S][C - partC] when the two halves of a concatenation bring their own partials
    // (pgn_groupnorm_silu_nhwc_pre), else part2 == nullptr, partS == S, partC == C
    const float2* part2;
    int partS, partC, part2S;
    // one-pass form only
    unsigned int* arrived;           // [N] workgroups of the sample that have published (zeroed by the launch function)
    unsigned long long* gpart;       // [N][S][G] {sum, sum of squares} about the group pivot, one 8-byte word each
    unsigned int timeout;            // bound of the wait, in s_memrealtime ticks (100 MHz)
#ifdef PGN_TUNING_HOOKS
    unsigned long long* dbg;         // measurement build: [workgroups][8] s_memrealtime stamps, or nullptr
#endif
};

__device__ __forceinline__ float nh_param(const void* p, int dt, size_t i, float dflt) {
    if (p == nullptr) return dflt;
    if (dt == PGN_F32) return reinterpret_cast<const float*>(p)[i];
    const unsigned short h = reinterpret_cast<const unsigned short*>(p)[i];
    return dt == PGN_BF16 ? Cvt<PGN_BF16>::to_f32(h) : Cvt<PGN_F16>::to_f32(h);
}

// eight consecutive parameters starting at an element index that is a multiple of 8 (16-byte aligned rows: C % 8 == 0,
// 16-byte aligned base): one or two 16-byte loads instead of eight scalar ones in every workgroup's prologue
__device__ __forceinline__ void nh_param8(const void* p, int dt, size_t i, float dflt, float (&out)[8]) {
    if (p == nullptr) {
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = dflt;
    } else if (dt == PGN_F32) {
        const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(p) + i);
        const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(p) + i + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { out[j] = lo[j]; out[4 + j] = hi[j]; }
    } else {
        const u16x8_t v = *reinterpret_cast<const u16x8_t*>(reinterpret_cast<const unsigned short*>(p) + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = dt == PGN_BF16 ? Cvt<PGN_BF16>::to_f32(v[j]) : Cvt<PGN_F16>::to_f32(v[j]);
    }
}

// a lane's 8 channels live in one of the two sources (C1 % 8 == 0): start of its column in sample n, and the row pitch
__device__ __forceinline__ const unsigned short* nh_lane_src(const NhArgs& a, int n, int c, int* pitch) {
    if (c < a.C1) {
        *pitch = a.C1;
        return reinterpret_cast<const unsigned short*>(a.x) + (size_t)n * a.HW * a.C1 + c;
    }
    *pitch = a.C - a.C1;
    return reinterpret_cast<const unsigned short*>(a.x2) + (size_t)n * a.HW * (a.C - a.C1) + (c - a.C1);
}

template <int DT>
__device__ __forceinline__ void nh_load8(const unsigned short* p, const float (&t)[8], bool has_add, float (&f)[8]) {
    const u16x8_t v = *reinterpret_cast<const u16x8_t*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = Cvt<DT>::to_f32(v[j]);
    if (has_add) {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = Cvt<DT>::to_f32(Cvt<DT>::from_f32(f[j] + t[j]));   // the reference's bf16 add
    }
}

__device__ __forceinline__ float nh_silu(float v) {
    const float e = __builtin_amdgcn_exp2f(v * -1.4426950408889634f);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}

// HAS_ADD (the per-(sample, channel) addend of ResnetBlock2D's norm2) and, in the apply kernel, ACT are compile-time:
// as run-time flags they were if-converted into ~110 v_cndmask per loop body.  The per-element arithmetic is written on
// float pairs so that it compiles to packed fp32 instructions (v_pk_add / v_pk_mul / v_pk_fma: two elements per issue
// slot); the operations and their order are unchanged, results are bit-identical to the scalar form.
// one 16-byte vector of a lane's column into its pivot-shifted sums: shared by the statistics launch and by the
// producer-side form (pgn_bias_res_stats_kernel), whose words must be the statistics launch's bit for bit
template <int DT, bool HAS_ADD>
__device__ __forceinline__ void nh_accum8(const u16x8_t& v, const float (&tadd)[8], const float (&piv)[8], f32x2_t (&s2)[4],
                                          f32x2_t (&q2)[4]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        f32x2_t f = {Cvt<DT>::to_f32(v[2 * p]), Cvt<DT>::to_f32(v[2 * p + 1])};
        if constexpr (HAS_ADD) {
            f += f32x2_t{tadd[2 * p], tadd[2 * p + 1]};
            f = f32x2_t{Cvt<DT>::to_f32(Cvt<DT>::from_f32(f[0])), Cvt<DT>::to_f32(Cvt<DT>::from_f32(f[1]))};
        }
        const f32x2_t d = f - f32x2_t{piv[2 * p], piv[2 * p + 1]};
        s2[p] += d;
        q2[p] += d * d;
    }
}

template <int DT, bool HAS_ADD>
__global__ __launch_bounds__(NH_MAX_THREADS) void pgn_gn_nhwc_stats_kernel(NhArgs a) {
    extern __shared__ float2 sh[];   // [R][C]
    const int n = blockIdx.x / a.S, sc = blockIdx.x % a.S;
    const int t = threadIdx.x;
    const int r = t / a.C8, cc = t - r * a.C8;
    const bool active = r < a.R;
    constexpr bool has_add = HAS_ADD;
    float tadd[8], piv[8];
    f32x2_t s2[4], q2[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) { tadd[j] = 0.f; piv[j] = 0.f; }
#pragma unroll
    for (int p = 0; p < 4; ++p) { s2[p] = f32x2_t{0.f, 0.f}; q2[p] = f32x2_t{0.f, 0.f}; }
    if (active) {
        if (has_add) nh_param8(a.addnc, a.add_dt, (size_t)n * a.C + cc * 8, 0.f, tadd);
        int pitch;
        const unsigned short* xp = nh_lane_src(a, n, cc * 8, &pitch);
        // one pivot per channel: the MEDIAN of the channel's values in rows 0, HW / 2 and HW - 1 of the sample.  A
        // single element (the value in row 0, until round 3) is a bad pivot exactly when it is an outlier: every
        // deviation is then large and the finalize kernel's q + 2 (p - mu) s + HW (p - mu)^2 cancels; one outlier
        // among three rows cannot be their median (test_outlier_at_the_pivot_position).
        {
            float pa[8], pb[8];
            nh_load8<DT>(xp, tadd, has_add, piv);
            nh_load8<DT>(xp + (size_t)(a.HW / 2) * pitch, tadd, has_add, pa);
            nh_load8<DT>(xp + (size_t)(a.HW - 1) * pitch, tadd, has_add, pb);
#pragma unroll
            for (int j = 0; j < 8; ++j) piv[j] = __builtin_amdgcn_fmed3f(piv[j], pa[j], pb[j]);
        }
        const int row1 = min(a.HW, (sc + 1) * a.rows);
        auto accum = [&](const u16x8_t& v) { nh_accum8<DT, HAS_ADD>(v, tadd, piv, s2, q2); };
        int row = sc * a.rows + r;
        // four independent 16-byte loads in flight per lane (a lane's rows are R rows apart)
        for (; row + 3 * a.R < row1; row += 4 * a.R) {
            const u16x8_t v0 = *reinterpret_cast<const u16x8_t*>(xp + (size_t)row * pitch);
            const u16x8_t v1 = *reinterpret_cast<const u16x8_t*>(xp + (size_t)(row + a.R) * pitch);
            const u16x8_t v2 = *reinterpret_cast<const u16x8_t*>(xp + (size_t)(row + 2 * a.R) * pitch);
            const u16x8_t v3 = *reinterpret_cast<const u16x8_t*>(xp + (size_t)(row + 3 * a.R) * pitch);
            accum(v0); accum(v1); accum(v2); accum(v3);
        }
        for (; row < row1; row += a.R) accum(*reinterpret_cast<const u16x8_t*>(xp + (size_t)row * pitch));
#pragma unroll
        for (int j = 0; j < 8; ++j) sh[(size_t)r * a.C + cc * 8 + j] = make_float2(s2[j >> 1][j & 1], q2[j >> 1][j & 1]);
    }
    __syncthreads();
    for (int c = t; c < a.C; c += blockDim.x) {
        float ss = 0.f, qq = 0.f;
        for (int rr = 0; rr < a.R; ++rr) {
            const float2 v = sh[(size_t)rr * a.C + c];
            ss += v.x;
            qq += v.y;
        }
        a.part[((size_t)n * a.S + sc) * a.C + c] = make_float2(ss, qq);
    }
}

// Producer-side statistics (pgn_bias_residual_add_stats): y = round(round(x + bias[c]) + res), the arithmetic of
// pgn_bias_res_cl_kernel, on the statistics launch's workgroup geometry, and the per-(sample, row chunk, channel)
// partials of y in the same pass - the words pgn_gn_nhwc_stats_kernel<DT, false> would write after reading y back
// (same pivots: median of y's rows 0, HW / 2, HW - 1, recomputed here from the inputs; same order of accumulation).
// The GroupNorm that consumes y then starts at its finalize launch: its first read of the tensor is gone.  x (and res)
// must not overlap y: every workgroup of a sample reads the three pivot rows of the INPUTS.
// U rows in flight per lane, WPS waves per SIMD the register budget is cut for (measured: see the launch function)
template <int DT, bool HAS_RES, int U, int WPS>
__global__ __launch_bounds__(NH_MAX_THREADS, WPS) void pgn_bias_res_stats_kernel(NhArgs a, const void* __restrict__ res,
                                                                               const void* __restrict__ bias, int bdt) {
    extern __shared__ float2 sh[];   // [R][C]
    const int n = blockIdx.x / a.S, sc = blockIdx.x % a.S;
    const int t = threadIdx.x;
    const int r = t / a.C8, cc = t - r * a.C8;
    const bool active = r < a.R;
    if (active) {
        float b[8], piv[8], none[8];
        f32x2_t s2[4], q2[4];
        nh_param8(bias, bdt, (size_t)cc * 8, 0.f, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) none[j] = 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p) { s2[p] = f32x2_t{0.f, 0.f}; q2[p] = f32x2_t{0.f, 0.f}; }
        const size_t col = (size_t)n * a.HW * a.C + (size_t)cc * 8;
        const unsigned short* xp = reinterpret_cast<const unsigned short*>(a.x) + col;
        const unsigned short* rp = reinterpret_cast<const unsigned short*>(res) + col;
        unsigned short* yp = reinterpret_cast<unsigned short*>(a.y) + col;
        auto produce = [&](const u16x8_t& vx, const u16x8_t& vr) {
            u16x8_t o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f = Cvt<DT>::to_f32(Cvt<DT>::from_f32(Cvt<DT>::to_f32(vx[j]) + b[j]));
                if constexpr (HAS_RES) f += Cvt<DT>::to_f32(vr[j]);
                o[j] = Cvt<DT>::from_f32(f);
            }
            return o;
        };
        auto ldx = [&](size_t row) { return *reinterpret_cast<const u16x8_t*>(xp + row * a.C); };
        auto ldr = [&](size_t row) {
            if constexpr (HAS_RES) return *reinterpret_cast<const u16x8_t*>(rp + row * a.C);
            else return u16x8_t{};
        };
        {
            const size_t rows3[3] = {0, (size_t)(a.HW / 2), (size_t)(a.HW - 1)};
            float pv[3][8];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const u16x8_t o = produce(ldx(rows3[k]), ldr(rows3[k]));
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[k][j] = Cvt<DT>::to_f32(o[j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) piv[j] = __builtin_amdgcn_fmed3f(pv[0][j], pv[1][j], pv[2][j]);
        }
        const int row1 = min(a.HW, (sc + 1) * a.rows);
        auto one = [&](const u16x8_t& vx, const u16x8_t& vr, size_t row) {
            const u16x8_t o = produce(vx, vr);
            *reinterpret_cast<u16x8_t*>(yp + row * a.C) = o;
            nh_accum8<DT, false>(o, none, piv, s2, q2);
        };
        int row = sc * a.rows + r;
        // two rows (four 16-byte loads with a residual) in flight per lane: at <= 64 VGPRs two 1024-thread workgroups
        // share a CU, so one streams while the other is in its LDS reduction (four rows: 123 VGPRs, one workgroup)
        for (; row + (U - 1) * a.R < row1; row += U * a.R) {
            u16x8_t xv[U], rv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) xv[u] = ldx(row + u * a.R);
#pragma unroll
            for (int u = 0; u < U; ++u) rv[u] = ldr(row + u * a.R);
#pragma unroll
            for (int u = 0; u < U; ++u) one(xv[u], rv[u], row + u * a.R);
        }
        for (; row < row1; row += a.R) one(ldx(row), ldr(row), row);
#pragma unroll
        for (int j = 0; j < 8; ++j) sh[(size_t)r * a.C + cc * 8 + j] = make_float2(s2[j >> 1][j & 1], q2[j >> 1][j & 1]);
    }
    __syncthreads();
    for (int c = t; c < a.C; c += blockDim.x) {
        float ss = 0.f, qq = 0.f;
        for (int rr = 0; rr < a.R; ++rr) {
            const float2 v = sh[(size_t)rr * a.C + c];
            ss += v.x;
            qq += v.y;
        }
        a.part[((size_t)n * a.S + sc) * a.C + c] = make_float2(ss, qq);
    }
}

// one 256-thread workgroup per (sample, group): thread -> (chunk lane, channel of the group); the chunk lanes split the S
// row chunks, LDS brings them together, threads [0, cpg) finish their channel in a fixed order
constexpr int NH_FIN_THREADS = 256;

__device__ __forceinline__ float nh_fin_block_sum(float v, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();                                   // red may still be read from the previous sum
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int w = 0; w < NH_FIN_THREADS / 64; ++w) r += red[w];
    return r;
}

template <int DT>
__global__ __launch_bounds__(NH_FIN_THREADS) void pgn_gn_nhwc_finalize_kernel(NhArgs a) {
    __shared__ float2 sh[NH_FIN_THREADS];
    __shared__ float red[NH_FIN_THREADS / 64];
    const int n = blockIdx.x / a.G, g = blockIdx.x % a.G;
    const int t = threadIdx.x;
    const float hw = (float)a.HW;
    const int L = NH_FIN_THREADS / a.cpg;              // cpg <= 128: at least two chunk lanes
    const int ls = t / a.cpg, cl = t - ls * a.cpg;
    float s = 0.f, q = 0.f;
    if (ls < L) {
        const int c = g * a.cpg + cl;
        const bool second = a.part2 != nullptr && c >= a.partC;
        const int pitch = second ? a.C - a.partC : a.partC, Sc = second ? a.part2S : a.partS;
        const float2* p = second ? a.part2 + (size_t)n * Sc * pitch + (c - a.partC) : a.part + (size_t)n * Sc * pitch + c;
        for (int sc = ls; sc < Sc; sc += L) {
            const float2 v = p[(size_t)sc * pitch];
            s += v.x;
            q += v.y;
        }
    }
    sh[t] = make_float2(s, q);
    __syncthreads();
    const bool v0 = t < a.cpg;
    float S0 = 0.f, Q0 = 0.f, p0 = 0.f;
    if (v0) {
        for (int l = 0; l < L; ++l) {
            const float2 v = sh[l * a.cpg + t];
            S0 += v.x;
            Q0 += v.y;
        }
        const int c = g * a.cpg + t;
        int pitch;
        // per-channel pivot, as the statistics kernel took it: median of rows 0, HW / 2, HW - 1
        const unsigned short* xc = nh_lane_src(a, n, c, &pitch);
        float pr[3] = {Cvt<DT>::to_f32(xc[0]), Cvt<DT>::to_f32(xc[(size_t)(a.HW / 2) * pitch]),
                       Cvt<DT>::to_f32(xc[(size_t)(a.HW - 1) * pitch])};
        if (a.addnc != nullptr) {
            const float ad = nh_param(a.addnc, a.add_dt, (size_t)n * a.C + c, 0.f);
#pragma unroll
            for (int k = 0; k < 3; ++k) pr[k] = Cvt<DT>::to_f32(Cvt<DT>::from_f32(pr[k] + ad));
        }
        p0 = __builtin_amdgcn_fmed3f(pr[0], pr[1], pr[2]);
    }
    const float mean = nh_fin_block_sum(v0 ? p0 + S0 / hw : 0.f, red) / (float)a.cpg;
    float var = 0.f;
    if (v0) { const float d = p0 - mean; var = Q0 + 2.f * d * S0 + hw * d * d; }
    var = fmaxf(nh_fin_block_sum(var, red) / ((float)a.cpg * hw), 0.f);
    if (t == 0) a.stats[(size_t)n * a.G + g] = make_float2(mean, 1.0f / sqrtf(var + a.eps));
}

// U rows in flight per lane, WPS waves per SIMD the register budget is cut for (swept: scripts/gn_apply_sweep.py)
template <int DT, bool HAS_ADD, bool ACT, int U = 4, int WPS = 4>
__global__ __launch_bounds__(NH_MAX_THREADS, WPS) void pgn_gn_nhwc_apply_kernel(NhArgs a) {
    const unsigned bid = a.reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const int n = bid / a.S, sc = bid % a.S;
    const int t = threadIdx.x;
    const int r = t / a.C8, cc = t - r * a.C8;
    if (r >= a.R) return;
    const size_t img = (size_t)n * a.HW * a.C;
    float tadd[8], mean[8], ga[8], be[8];
    nh_param8(HAS_ADD ? a.addnc : nullptr, a.add_dt, (size_t)n * a.C + cc * 8, 0.f, tadd);
    nh_param8(a.gamma, a.adt, (size_t)cc * 8, 1.f, ga);
    nh_param8(a.beta, a.adt, (size_t)cc * 8, 0.f, be);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float2 st = a.stats[(size_t)n * a.G + (cc * 8 + j) / a.cpg];
        mean[j] = st.x;
        ga[j] *= st.y;
    }
    const int row1 = min(a.HW, (sc + 1) * a.rows);
    unsigned short* yp = reinterpret_cast<unsigned short*>(a.y);
    int pitch;
    const unsigned short* xp = nh_lane_src(a, n, cc * 8, &pitch);
    auto apply = [&](const u16x8_t& v, size_t e) {
        u16x8_t o;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            f32x2_t f = {Cvt<DT>::to_f32(v[2 * p]), Cvt<DT>::to_f32(v[2 * p + 1])};
            if constexpr (HAS_ADD) {
                f += f32x2_t{tadd[2 * p], tadd[2 * p + 1]};
                f = f32x2_t{Cvt<DT>::to_f32(Cvt<DT>::from_f32(f[0])), Cvt<DT>::to_f32(Cvt<DT>::from_f32(f[1]))};
            }
            f32x2_t w = (f - f32x2_t{mean[2 * p], mean[2 * p + 1]}) * f32x2_t{ga[2 * p], ga[2 * p + 1]} +
                        f32x2_t{be[2 * p], be[2 * p + 1]};
            if constexpr (ACT) {   // SiLU: w / (1 + exp(-w)), exp2 domain
                const f32x2_t x2 = w * f32x2_t{-1.4426950408889634f, -1.4426950408889634f};
                f32x2_t d = f32x2_t{__builtin_amdgcn_exp2f(x2[0]), __builtin_amdgcn_exp2f(x2[1])} + f32x2_t{1.0f, 1.0f};
                w = w * f32x2_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
            }
            o[2 * p] = Cvt<DT>::from_f32(w[0]);
            o[2 * p + 1] = Cvt<DT>::from_f32(w[1]);
        }
        *reinterpret_cast<u16x8_t*>(yp + e) = o;
    };
    int row = sc * a.rows + r;
    if constexpr (U == 4) {      // the shipped form, written out (the generic loop below costs it a 20-byte spill)
        for (; row + 3 * a.R < row1; row += 4 * a.R) {
            const size_t e0 = img + (size_t)row * a.C + cc * 8, st = (size_t)a.R * a.C;
            const size_t i0 = (size_t)row * pitch, si = (size_t)a.R * pitch;
            const u16x8_t v0 = *reinterpret_cast<const u16x8_t*>(xp + i0);
            const u16x8_t v1 = *reinterpret_cast<const u16x8_t*>(xp + i0 + si);
            const u16x8_t v2 = *reinterpret_cast<const u16x8_t*>(xp + i0 + 2 * si);
            const u16x8_t v3 = *reinterpret_cast<const u16x8_t*>(xp + i0 + 3 * si);
            apply(v0, e0); apply(v1, e0 + st); apply(v2, e0 + 2 * st); apply(v3, e0 + 3 * st);
        }
    } else {
        for (; row + (U - 1) * a.R < row1; row += U * a.R) {
            const size_t e0 = img + (size_t)row * a.C + cc * 8, st = (size_t)a.R * a.C;
            const size_t i0 = (size_t)row * pitch, si = (size_t)a.R * pitch;
            u16x8_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const u16x8_t*>(xp + i0 + u * si);
#pragma unroll
            for (int u = 0; u < U; ++u) apply(v[u], e0 + u * st);
        }
    }
    for (; row < row1; row += a.R) {
        const size_t e = img + (size_t)row * a.C + cc * 8;
        apply(*reinterpret_cast<const u16x8_t*>(xp + (size_t)row * pitch), e);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// One-pass form: 1 read + 1 write of the tensor.  A workgroup keeps its row chunk in registers (NH_OP_ROWS 16-byte
// loads per lane), publishes its per-group partial sums, waits until every workgroup of its sample has published,
// reduces the S x G partials in a fixed order itself and applies from registers.  A workgroup's contribution is G
// words, not C: per group the (mean, sum of squared deviations about that mean) of its own chunk, taken in two passes
// over the registers; the chunks are merged with the parallel-variance update (no pivot, no E[d^2] - E[d]^2).
//
// Visibility between workgroups (other CUs, other XCDs) follows the guide's counter hand-off: the partials are 8-byte
// agent-scope atomic stores (write-through), every storing wave drains vmcnt, a workgroup barrier, then one lane's
// agent-scope add on the sample's counter; the consumer polls that counter relaxed from one lane, runs ONE agent-scope
// acquire, drains, barrier, and only then loads the partials (as agent-scope atomic loads as well).  x, gamma, beta and
// add_nc are written by nobody in the launch; y is written and never read.
//
// Progress does not depend on dispatch order or residency: the wait is bounded (a.timeout), and a workgroup whose wait
// runs out computes the missing chunks' partials itself from global memory with the same arithmetic in the same order
// (bit-identical words; it also stores them, which is a rewrite of identical values).  The launch function only picks
// this form when a sample's S workgroups fit the chip several times over (and, by default, only for tensors of up to
// NH_OP_MAX_BYTES: see nh_run), so the bounded path is a safety net, and it needs x and y not to overlap (the pivot row and, in the safety net, other chunks are read after peers may have
// written theirs).
typedef __attribute__((address_space(1))) unsigned long long nh_gu64;
typedef __attribute__((address_space(1))) unsigned int nh_gu32;
constexpr int NH_OP_ROWS = 16;
constexpr int NH_OP_MAX_G = 64;
constexpr size_t NH_OP_MAX_BYTES = (size_t)32 << 20;   // tensors up to this size take the one-pass form by default

template <int DT, bool HAS_ADD, bool ACT>
__global__ __launch_bounds__(NH_MAX_THREADS) void pgn_gn_nhwc_onepass_kernel(NhArgs a) {
    extern __shared__ __attribute__((aligned(16))) float2 sh[];
    float2* shc = sh;                                             // [R][C] lane sums, then row 0 = per-channel sums
    float2* shL = shc + (size_t)a.R * a.C;                        // [L][G] chunk-lane sums of the S-reduction
    float2* shS = shL + blockDim.x;                               // [G] mean, rstd
    float* shP = reinterpret_cast<float*>(shS + NH_OP_MAX_G);     // [G] pivots
    int* shFlag = reinterpret_cast<int*>(shP + NH_OP_MAX_G);
    const int n = blockIdx.x / a.S, sc = blockIdx.x % a.S;
    const int t = threadIdx.x;
#ifdef PGN_TUNING_HOOKS
#define NH_STAMP(k)                                                                                                  \
    do {                                                                                                             \
        if (a.dbg != nullptr && t == 0) a.dbg[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime();      \
    } while (0)
#else
#define NH_STAMP(k) do { } while (0)
#endif
    NH_STAMP(0);
    const int r = t / a.C8, cc = t - r * a.C8;
    const bool active = r < a.R;
    int pitch = 0;
    const unsigned short* xp = nullptr;
    float tadd[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) tadd[j] = 0.f;
    // own chunk: every load issued before anything else needs its data
    u16x8_t v[NH_OP_ROWS];
    int nv = 0;
    const int row0 = sc * a.rows, row1 = min(a.HW, row0 + a.rows);
    if (active) {
        xp = nh_lane_src(a, n, cc * 8, &pitch);
        if (HAS_ADD) nh_param8(a.addnc, a.add_dt, (size_t)n * a.C + cc * 8, 0.f, tadd);
        const int first = row0 + r;
        nv = first < row1 ? (row1 - first + a.R - 1) / a.R : 0;
#pragma unroll
        for (int i = 0; i < NH_OP_ROWS; ++i)
            if (i < nv) v[i] = *reinterpret_cast<const u16x8_t*>(xp + (size_t)(first + i * a.R) * pitch);
    }
    // Statistics of a chunk: TWO passes over the chunk's rows (registers for the workgroup's own chunk) - per-group sum ->
    // the chunk's own mean m, then the sum of squared deviations about m.  No pivot: a pivot element far from the group
    // (an outlier pixel at the pivot position) made every deviation large and E[d^2] - E[d]^2 cancel (round 3 test
    // test_outlier_at_the_pivot_position: 1.4 absolute at 300 sigma).  The S chunks of a sample are merged with the
    // parallel-variance update (Chan et al.) in a fixed order, so nothing ever subtracts two large numbers.
    float* shf = reinterpret_cast<float*>(shc);                  // [R][C] lane sums (one float per element)
    auto value = [&](const u16x8_t& w, int j, const float (&ta)[8]) -> float {
        float f = Cvt<DT>::to_f32(w[j]);
        if constexpr (HAS_ADD) f = Cvt<DT>::to_f32(Cvt<DT>::from_f32(f + ta[j]));
        return f;
    };
    // lanes' per-channel partials -> per-channel -> per-group totals in shf[0 .. G); ends behind a barrier
    auto fold = [&](const float (&acc)[8]) {
        if (active) {
#pragma unroll
            for (int j = 0; j < 8; ++j) shf[(size_t)r * a.C + cc * 8 + j] = acc[j];
        }
        __syncthreads();
        float tot = 0.f;
        for (int c = t; c < a.C; c += blockDim.x) {               // column c belongs to this thread alone
            float ss = 0.f;
            for (int rr = 0; rr < a.R; ++rr) ss += shf[(size_t)rr * a.C + c];
            shf[c] = ss;
        }
        __syncthreads();
        if (t < a.G) {
            for (int cl = 0; cl < a.cpg; ++cl) tot += shf[t * a.cpg + cl];
        }
        __syncthreads();                                          // row 0 has been read: shf may be rewritten
        return tot;                                               // valid in threads t < G
    };
    // statistics of chunk `chunk` from `rows_of(i)` (i-th row of this lane), published as (mean, M2) of the chunk
    auto chunk_stats = [&](int chunk, int cnt_rows, auto&& row_at, int nrows, const float (&ta)[8]) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int i = 0; i < nrows; ++i) {
            const u16x8_t w = row_at(i);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += value(w, j, ta);
        }
        const float tot = fold(acc);
        const float cntf = (float)cnt_rows * (float)a.cpg;
        if (t < a.G) shP[t] = tot / cntf;                         // the chunk's own mean per group
        __syncthreads();
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            m[j] = active ? shP[(cc * 8 + j) / a.cpg] : 0.f;
            acc[j] = 0.f;
        }
        for (int i = 0; i < nrows; ++i) {
            const u16x8_t w = row_at(i);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = value(w, j, ta) - m[j];
                acc[j] += d * d;
            }
        }
        const float m2 = fold(acc);
        if (t < a.G) {
            const unsigned long long word = ((unsigned long long)__float_as_uint(m2) << 32) | __float_as_uint(shP[t]);
            __hip_atomic_store(((nh_gu64*)a.gpart) + ((size_t)n * a.S + chunk) * a.G + t, word, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    {   // own chunk, from registers (static indices: the loops are unrolled over NH_OP_ROWS with an `i < nv` guard).
        // With add_nc the rounded sums round(x + add) REPLACE the rows in their registers first: every later pass
        // (sum, squared deviations, apply) reads the value the reference's tensor add would have written, the add is
        // done once instead of three times and its eight addends are dead during the passes (128-VGPR budget).
        if constexpr (HAS_ADD) {
#pragma unroll
            for (int i = 0; i < NH_OP_ROWS; ++i)
                if (i < nv) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[i][j] = Cvt<DT>::from_f32(Cvt<DT>::to_f32(v[i][j]) + tadd[j]);
                }
        }
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int i = 0; i < NH_OP_ROWS; ++i)
            if (i < nv) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += Cvt<DT>::to_f32(v[i][j]);
            }
        const float tot = fold(acc);
        if (t < a.G) shP[t] = tot / ((float)(row1 - row0) * (float)a.cpg);
        __syncthreads();
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            m[j] = active ? shP[(cc * 8 + j) / a.cpg] : 0.f;
            acc[j] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < NH_OP_ROWS; ++i)
            if (i < nv) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = Cvt<DT>::to_f32(v[i][j]) - m[j];
                    acc[j] += d * d;
                }
            }
        NH_STAMP(1);
        const float m2 = fold(acc);
        if (t < a.G) {
            const unsigned long long word = ((unsigned long long)__float_as_uint(m2) << 32) | __float_as_uint(shP[t]);
            __hip_atomic_store(((nh_gu64*)a.gpart) + ((size_t)n * a.S + sc) * a.G + t, word, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    NH_STAMP(2);
    nh_gu32* arrived = ((nh_gu32*)a.arrived) + n;
    if (t == 0) {
        __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)a.S) {
            if (__builtin_amdgcn_s_memrealtime() - t0 >= (unsigned long long)a.timeout) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(4);
        }
        *shFlag = ok;
    }
    __syncthreads();
    NH_STAMP(3);
    if (*shFlag == 0) {
        // safety net: the peers have not all published within the bound; compute their chunks' words here, with the
        // same arithmetic in the same order (a lane walks its rows in the order the owner holds them in registers)
        float ta[8];                                              // the addends again: not kept live across the kernel
#pragma unroll
        for (int j = 0; j < 8; ++j) ta[j] = 0.f;
        if (HAS_ADD && active) nh_param8(a.addnc, a.add_dt, (size_t)n * a.C + cc * 8, 0.f, ta);
        for (int chunk = 0; chunk < a.S; ++chunk) {
            if (chunk == sc) continue;
            const int rb = min(a.HW, (chunk + 1) * a.rows), ra = chunk * a.rows;
            const int first = ra + r;
            const int nrows = (active && first < rb) ? (rb - first + a.R - 1) / a.R : 0;
            chunk_stats(chunk, rb - ra,
                        [&](int i) { return *reinterpret_cast<const u16x8_t*>(xp + (size_t)(first + i * a.R) * pitch); },
                        nrows, ta);
        }
    }
    if (t == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {   // S x G words -> G statistics in a fixed order: chunk lane ls merges chunks ls, ls + L, ... one after the other,
        // then the L lanes are merged in order.  Merge of (n_a, m_a, M2_a) with (n_b, m_b, M2_b):
        //   d = m_b - m_a;  n = n_a + n_b;  m = m_a + d n_b / n;  M2 = M2_a + M2_b + d^2 n_a n_b / n
        const int L = blockDim.x / a.G;
        const int ls = t / a.G, g = t - ls * a.G;
        auto chunk_cnt = [&](int c2) { return (float)(min(a.HW, (c2 + 1) * a.rows) - c2 * a.rows) * (float)a.cpg; };
        if (ls < L) {
            float cn = 0.f, cm = 0.f, cM2 = 0.f;
            const nh_gu64* p = ((const nh_gu64*)a.gpart) + (size_t)n * a.S * a.G + g;
            for (int c2 = ls; c2 < a.S; c2 += 8 * L) {            // eight loads in flight, merged in chunk order
                unsigned long long word[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    word[u] = c2 + u * L < a.S ? __hip_atomic_load(p + (size_t)(c2 + u * L) * a.G, __ATOMIC_RELAXED,
                                                                  __HIP_MEMORY_SCOPE_AGENT)
                                               : 0ull;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (c2 + u * L < a.S) {
                        const float nb = chunk_cnt(c2 + u * L);
                        const float mb = __uint_as_float((unsigned)word[u]), M2b = __uint_as_float((unsigned)(word[u] >> 32));
                        const float nn = cn + nb, d = mb - cm, f = nb / nn;
                        cm += d * f;
                        cM2 += M2b + d * d * cn * f;
                        cn = nn;
                    }
                }
            }
            shL[ls * a.G + g] = make_float2(cm, cM2);
        }
        __syncthreads();
        if (t < a.G) {
            float cn = 0.f, cm = 0.f, cM2 = 0.f;
            for (int l = 0; l < L && l < a.S; ++l) {
                // elements merged by chunk lane l: chunks l, l + L, ... (all full but possibly the sample's last one)
                const int k = (a.S - l + L - 1) / L;
                float nb = (float)k * (float)a.rows;
                if ((a.S - 1) % L == l) nb -= (float)(a.S * a.rows - a.HW);
                nb *= (float)a.cpg;
                const float2 w = shL[l * a.G + t];
                const float nn = cn + nb, d = w.x - cm, f = nb / nn;
                cm += d * f;
                cM2 += w.y + d * d * cn * f;
                cn = nn;
            }
            const float var = fmaxf(cM2 / cn, 0.f);
            shS[t] = make_float2(cm, 1.0f / sqrtf(var + a.eps));
        }
        __syncthreads();
    }
    NH_STAMP(4);
    if (!active) return;
    float mean[8], ga[8], be[8];
    nh_param8(a.gamma, a.adt, (size_t)cc * 8, 1.f, ga);
    nh_param8(a.beta, a.adt, (size_t)cc * 8, 0.f, be);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float2 st = shS[(cc * 8 + j) / a.cpg];
        mean[j] = st.x;
        ga[j] *= st.y;
    }
    unsigned short* yp = reinterpret_cast<unsigned short*>(a.y) + (size_t)n * a.HW * a.C + (size_t)(row0 + r) * a.C + cc * 8;
#pragma unroll
    for (int i = 0; i < NH_OP_ROWS; ++i) {
        if (i < nv) {
            u16x8_t o;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const f32x2_t f = {Cvt<DT>::to_f32(v[i][2 * p]), Cvt<DT>::to_f32(v[i][2 * p + 1])};   // add_nc: already in
                f32x2_t w = (f - f32x2_t{mean[2 * p], mean[2 * p + 1]}) * f32x2_t{ga[2 * p], ga[2 * p + 1]} +
                            f32x2_t{be[2 * p], be[2 * p + 1]};
                if constexpr (ACT) {
                    const f32x2_t x2 = w * f32x2_t{-1.4426950408889634f, -1.4426950408889634f};
                    f32x2_t d = f32x2_t{__builtin_amdgcn_exp2f(x2[0]), __builtin_amdgcn_exp2f(x2[1])} + f32x2_t{1.0f, 1.0f};
                    w = w * f32x2_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
                }
                o[2 * p] = Cvt<DT>::from_f32(w[0]);
                o[2 * p + 1] = Cvt<DT>::from_f32(w[1]);
            }
            *reinterpret_cast<u16x8_t*>(yp + (size_t)i * a.R * a.C) = o;
        }
    }
#ifdef PGN_TUNING_HOOKS
    if (a.dbg != nullptr && t == 0) {
        a.dbg[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        a.dbg[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime();
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.dbg[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
#undef NH_STAMP
}

#ifdef PGN_TUNING_HOOKS
// ---------------------------------------------------------------------------------------------------------------------
// Pipelined one-pass form for LARGE tensors (round 3; 1 read + 1 write like the form above).
// MEASURED NEGATIVE, measurement build only (PGN_GN_PIPELINE_ALWAYS in `act` selects it there): correct (its tests pass:
// fp64 bar, bit-identical safety net, concatenation, short chunks), traffic 1.0x, but 0.35-0.49x the speed of the three
// launches at UNet batch 32 (1.2-1.7 against 3.4 TB/s algorithmic; profiles/r03_groupnorm_pipeline_ab.txt) and 0.2-0.3x
// with four smaller workgroups per CU.  A step chains five or six global round trips that nothing overlaps at two
// workgroups per CU (publish + drain, fetch_add with its result, first poll, acquire, statistics load, then the exposed
// row loads), ~40 us per 61 KB item; the three launches stream with none.  The text below is the design as built.
// Why the form above loses on a saturated chip (batch 32: 0.70-0.87x of the three passes): between publishing its
// partial statistics and applying them a workgroup WAITS for the slowest workgroup of its sample (median 21 us of a
// 47 us life, scripts/gn_wg_timeline.py) with its rows parked in registers and nothing in flight.  Here the wait is
// taken out of the critical path by software pipelining inside PERSISTENT workgroups (two of <= 256 threads per CU): a
// workgroup walks items (sample, row chunk) w, w + P, w + 2 P, ... and keeps two items in registers (2 x NH_PP_ROWS
// 16-byte rows per lane = 128 VGPRs).  Per step: statistics of item k -> publish -> arrive; THEN finish item k - 1 (its sample's statistics were
// published a whole step ago by everybody, so the wait is normally over before it starts) -> apply from registers ->
// store; then the loads of item k + 1 go into the registers just freed.  Items of one sample are consecutive, so its
// chunks sit on different workgroups at the same (or the neighbouring) step.
// The S x G words of a sample are merged ONCE, by the workgroup that arrives last (fetch_add returns S - 1): it
// publishes G (mean, rstd) words and bumps the counter to S + 1, which is what finish() waits for; every other
// workgroup reads 8 G bytes instead of 8 S G.  Same chunk statistics, same fixed-order merge as the form above: the
// result does not depend on who merges.  Visibility: the counter hand-off described above, twice.
// Progress never depends on residency or dispatch order: publishing always precedes waiting within a step, every
// wait is bounded, and a workgroup whose wait runs out computes the sample's statistics itself (all chunks from
// global memory, same arithmetic, same order).  x and y must not overlap.
constexpr int NH_PP_ROWS = 16;
constexpr long long NH_PP_WGS_PER_CU = 2;

template <int DT, bool HAS_ADD, bool ACT>
__global__ __launch_bounds__(256, 2) void pgn_gn_nhwc_pipe_kernel(NhArgs a, int n_items) {
    // No FMA contraction in this kernel: a chunk's word is computed by two differently shaped copies of the same
    // arithmetic (rows in registers; rows from global memory in the safety net and for a short last chunk), and the
    // merge is inlined twice; the words must not depend on which copy the compiler chose to contract.
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float2 sh[];
    float* shf = reinterpret_cast<float*>(sh);                            // [R][C] lane sums
    float2* shL = reinterpret_cast<float2*>(shf + (size_t)a.R * a.C);     // [threads] chunk-lane partial merges
    float2* shS = shL + blockDim.x;                                       // [G] mean, rstd of the item being finished
    float* shP = reinterpret_cast<float*>(shS + NH_OP_MAX_G);             // [G] a chunk's own means
    int* shFlag = reinterpret_cast<int*>(shP + NH_OP_MAX_G);              // [2]
    const int P = gridDim.x, w = blockIdx.x;
    // The lane's coordinates are RE-DERIVED from an opaque copy of the thread id at the top of every step: as loop
    // invariants of the persistent loop, everything computed from them (eight group indices, LDS and parameter
    // addresses, source pointers) is hoisted to kernel entry and kept live across the loop.  A few integer divisions
    // per ~60 KB item are cheaper.
    int t = threadIdx.x;
    int r = t / a.C8, cc = t - r * a.C8;
    bool active = r < a.R;
    const int nst = (n_items - w + P - 1) / P;
    nh_gu64* gstat = (nh_gu64*)a.stats;                                   // [N][G] {mean, rstd} words

    // An item lives in registers only when its chunk is FULL (every active lane holds exactly NH_PP_ROWS rows: no
    // per-row conditions anywhere).  The one short chunk a sample may end with (HW % rows != 0: < 1 % of the rows)
    // takes the global-memory path the safety net uses: its statistics read it twice, its apply a third time.
    struct Item { int n, sc, row0, row1; bool full; };
    auto item = [&](int k) {
        Item it;
        const int idx = w + k * P;
        it.n = idx / a.S;
        it.sc = idx - it.n * a.S;
        it.row0 = it.sc * a.rows;
        it.row1 = min(a.HW, it.row0 + a.rows);
        it.full = it.row1 - it.row0 == a.rows;
        return it;
    };
    // one running 32-bit byte offset per access stream (a sample is < 4 GiB: checked on the host)
    auto load = [&](u16x8_t (&v)[NH_PP_ROWS], const Item& it) {
        if (!active || !it.full) return;
        int pitch;
        const char* xp = reinterpret_cast<const char*>(nh_lane_src(a, it.n, cc * 8, &pitch));
        unsigned off = (unsigned)(it.row0 + r) * (unsigned)pitch * 2u;
        const unsigned step = (unsigned)a.R * (unsigned)pitch * 2u;
#pragma unroll
        for (int i = 0; i < NH_PP_ROWS; ++i) {
            v[i] = *reinterpret_cast<const u16x8_t*>(xp + off);
            off += step;
        }
    };
    // lanes' per-channel partials -> per-channel -> per-group totals (valid in threads t < G); ends behind a barrier
    auto fold = [&](const float (&acc)[8]) {
        if (active) {
#pragma unroll
            for (int j = 0; j < 8; ++j) shf[(size_t)r * a.C + cc * 8 + j] = acc[j];
        }
        __syncthreads();
        float tot = 0.f;
        for (int c = t; c < a.C; c += blockDim.x) {
            float ss = 0.f;
            for (int rr = 0; rr < a.R; ++rr) ss += shf[(size_t)rr * a.C + c];
            shf[c] = ss;
        }
        __syncthreads();
        if (t < a.G) {
            for (int cl = 0; cl < a.cpg; ++cl) tot += shf[t * a.cpg + cl];
        }
        __syncthreads();
        return tot;
    };
    // store the chunk's word (its own mean per group is in shP, its M2 in m2 of threads t < G); drained + barrier
    auto publish_word = [&](int n, int chunk, float m2) {
        if (t < a.G) {
            const unsigned long long word = ((unsigned long long)__float_as_uint(m2) << 32) | __float_as_uint(shP[t]);
            __hip_atomic_store(((nh_gu64*)a.gpart) + ((size_t)n * a.S + chunk) * a.G + t, word, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    // one row of a lane from global memory, holding round(x + add) like the rows kept in registers
    auto row_from_memory = [&](const char* xp, unsigned off, const float (&ta)[8]) {
        u16x8_t wv = *reinterpret_cast<const u16x8_t*>(xp + off);
        if constexpr (HAS_ADD) {
#pragma unroll
            for (int j = 0; j < 8; ++j) wv[j] = Cvt<DT>::from_f32(Cvt<DT>::to_f32(wv[j]) + ta[j]);
        }
        return wv;
    };
    // (mean, M2) of chunk `chunk` of sample n straight from global memory (two reads), published; same arithmetic in
    // the same order as the register path: a lane walks its rows first, first + R, ...
    auto publish_chunk_from_memory = [&](int n, int chunk) {
        const int ra = chunk * a.rows, rb = min(a.HW, ra + a.rows);
        int pitch = 0;
        const char* xp = active ? reinterpret_cast<const char*>(nh_lane_src(a, n, cc * 8, &pitch)) : nullptr;
        float ta[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) ta[j] = 0.f;
        if (HAS_ADD && active) nh_param8(a.addnc, a.add_dt, (size_t)n * a.C + cc * 8, 0.f, ta);
        const unsigned step = (unsigned)a.R * (unsigned)pitch * 2u;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        if (active) {
            unsigned off = (unsigned)(ra + r) * (unsigned)pitch * 2u;
            for (int row = ra + r; row < rb; row += a.R, off += step) {
                const u16x8_t wv = row_from_memory(xp, off, ta);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += Cvt<DT>::to_f32(wv[j]);
            }
        }
        const float tot = fold(acc);
        if (t < a.G) shP[t] = tot / ((float)(rb - ra) * (float)a.cpg);
        __syncthreads();
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            m[j] = active ? shP[(cc * 8 + j) / a.cpg] : 0.f;
            acc[j] = 0.f;
        }
        if (active) {
            unsigned off = (unsigned)(ra + r) * (unsigned)pitch * 2u;
            for (int row = ra + r; row < rb; row += a.R, off += step) {
                const u16x8_t wv = row_from_memory(xp, off, ta);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = Cvt<DT>::to_f32(wv[j]) - m[j];
                    acc[j] += d * d;
                }
            }
        }
        publish_word(n, chunk, fold(acc));
    };
    // S x G words of sample n -> shS[g] = (mean, rstd); fixed order (see pgn_gn_nhwc_onepass_kernel).  The caller has
    // made the words visible (acquire + drain + barrier).
    auto merge = [&](int n) {
        const int L = blockDim.x / a.G;
        const int ls = t / a.G, g = t - ls * a.G;
        auto chunk_cnt = [&](int c2) { return (float)(min(a.HW, (c2 + 1) * a.rows) - c2 * a.rows) * (float)a.cpg; };
        if (ls < L) {
            float cn = 0.f, cm = 0.f, cM2 = 0.f;
            const nh_gu64* p = ((const nh_gu64*)a.gpart) + (size_t)n * a.S * a.G + g;
            for (int c2 = ls; c2 < a.S; c2 += 4 * L) {
                unsigned long long word[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    word[u] = c2 + u * L < a.S ? __hip_atomic_load(p + (size_t)(c2 + u * L) * a.G, __ATOMIC_RELAXED,
                                                                  __HIP_MEMORY_SCOPE_AGENT)
                                               : 0ull;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (c2 + u * L < a.S) {
                        const float nb = chunk_cnt(c2 + u * L);
                        const float mb = __uint_as_float((unsigned)word[u]), M2b = __uint_as_float((unsigned)(word[u] >> 32));
                        const float nn = cn + nb, d = mb - cm, f = nb / nn;
                        cm += d * f;
                        cM2 += M2b + d * d * cn * f;
                        cn = nn;
                    }
                }
            }
            shL[ls * a.G + g] = make_float2(cm, cM2);
        }
        __syncthreads();
        if (t < a.G) {
            float cn = 0.f, cm = 0.f, cM2 = 0.f;
            for (int l = 0; l < L && l < a.S; ++l) {
                const int k = (a.S - l + L - 1) / L;
                float nb = (float)k * (float)a.rows;
                if ((a.S - 1) % L == l) nb -= (float)(a.S * a.rows - a.HW);
                nb *= (float)a.cpg;
                const float2 wv = shL[l * a.G + t];
                const float nn = cn + nb, d = wv.x - cm, f = nb / nn;
                cm += d * f;
                cM2 += wv.y + d * d * cn * f;
                cn = nn;
            }
            const float var = fmaxf(cM2 / cn, 0.f);
            shS[t] = make_float2(cm, 1.0f / sqrtf(var + a.eps));
        }
        __syncthreads();
    };
    // statistics of an item: pre-add, publish, arrive; the last arriver merges for the sample
    auto stats = [&](u16x8_t (&v)[NH_PP_ROWS], const Item& it) {
        if (it.full) {
            if constexpr (HAS_ADD) {
                if (active) {
                    float ta[8];
                    nh_param8(a.addnc, a.add_dt, (size_t)it.n * a.C + cc * 8, 0.f, ta);
#pragma unroll
                    for (int i = 0; i < NH_PP_ROWS; ++i)
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[i][j] = Cvt<DT>::from_f32(Cvt<DT>::to_f32(v[i][j]) + ta[j]);
                }
            }
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
            for (int i = 0; i < NH_PP_ROWS; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += Cvt<DT>::to_f32(v[i][j]);
            const float tot = fold(acc);
            if (t < a.G) shP[t] = tot / ((float)a.rows * (float)a.cpg);
            __syncthreads();
            float m[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                m[j] = active ? shP[(cc * 8 + j) / a.cpg] : 0.f;
                acc[j] = 0.f;
            }
            // the rows stay PACKED between the passes: without this the compiler keeps the 128 widened fp32 values of
            // the first pass for the second one and for the apply (another 128 VGPRs, i.e. spills)
#pragma unroll
            for (int i = 0; i < NH_PP_ROWS; ++i) asm volatile("" : "+v"(v[i]));
#pragma unroll
            for (int i = 0; i < NH_PP_ROWS; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = Cvt<DT>::to_f32(v[i][j]) - m[j];
                    acc[j] += d * d;
                }
            publish_word(it.n, it.sc, fold(acc));
        } else {
            publish_chunk_from_memory(it.n, it.sc);
        }
        nh_gu32* arrived = ((nh_gu32*)a.arrived) + it.n;
        if (t == 0) shFlag[0] = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(a.S - 1);
        __syncthreads();
        if (shFlag[0]) {                                          // every chunk of the sample has been published
            if (t == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            merge(it.n);
            if (t < a.G) {
                const float2 st = shS[t];
                const unsigned long long word = ((unsigned long long)__float_as_uint(st.y) << 32) | __float_as_uint(st.x);
                __hip_atomic_store(gstat + (size_t)it.n * a.G + t, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // S + 1
        }
    };
    auto apply_row = [&](const u16x8_t& wv_in, const float (&mean)[8], const float (&ga)[8], const float (&be)[8]) {
        u16x8_t o;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const f32x2_t f = {Cvt<DT>::to_f32(wv_in[2 * p]), Cvt<DT>::to_f32(wv_in[2 * p + 1])};
            f32x2_t wv = (f - f32x2_t{mean[2 * p], mean[2 * p + 1]}) * f32x2_t{ga[2 * p], ga[2 * p + 1]} +
                         f32x2_t{be[2 * p], be[2 * p + 1]};
            if constexpr (ACT) {
                const f32x2_t x2 = wv * f32x2_t{-1.4426950408889634f, -1.4426950408889634f};
                f32x2_t d = f32x2_t{__builtin_amdgcn_exp2f(x2[0]), __builtin_amdgcn_exp2f(x2[1])} + f32x2_t{1.0f, 1.0f};
                wv = wv * f32x2_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
            }
            o[2 * p] = Cvt<DT>::from_f32(wv[0]);
            o[2 * p + 1] = Cvt<DT>::from_f32(wv[1]);
        }
        return o;
    };
    // wait for the sample's statistics, apply them to the item's rows, store
    auto finish = [&](u16x8_t (&v)[NH_PP_ROWS], const Item& it) {
        nh_gu32* arrived = ((nh_gu32*)a.arrived) + it.n;
        if (t == 0) {
            int ok = 1;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(a.S + 1)) {
                if (__builtin_amdgcn_s_memrealtime() - t0 >= (unsigned long long)a.timeout) { ok = 0; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            shFlag[1] = ok;
        }
        __syncthreads();
        if (shFlag[1]) {
            if (t == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t < a.G) {
                const unsigned long long word = __hip_atomic_load(gstat + (size_t)it.n * a.G + t, __ATOMIC_RELAXED,
                                                                  __HIP_MEMORY_SCOPE_AGENT);
                shS[t] = make_float2(__uint_as_float((unsigned)word), __uint_as_float((unsigned)(word >> 32)));
            }
            __syncthreads();
        } else {
            // safety net: the sample's statistics did not appear within the bound - compute them here from global memory
            // (every chunk but the own one, whose word this workgroup has published; identical words are rewritten)
            for (int chunk = 0; chunk < a.S; ++chunk) {
                if (chunk == it.sc) continue;
                publish_chunk_from_memory(it.n, chunk);
            }
            if (t == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            merge(it.n);
        }
        if (active) {
            // y = (x - mean) * (gamma * rstd) + beta, as the other forms compute it
            float mean[8], ga[8], be[8];
            nh_param8(a.gamma, a.adt, (size_t)cc * 8, 1.f, ga);
            nh_param8(a.beta, a.adt, (size_t)cc * 8, 0.f, be);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float2 st = shS[(cc * 8 + j) / a.cpg];
                mean[j] = st.x;
                ga[j] *= st.y;
            }
            char* yp = reinterpret_cast<char*>(a.y) + ((size_t)it.n * a.HW * a.C + cc * 8) * 2;
            unsigned off = (unsigned)(it.row0 + r) * (unsigned)a.C * 2u;
            const unsigned step = (unsigned)a.R * (unsigned)a.C * 2u;
            if (it.full) {
#pragma unroll
                for (int i = 0; i < NH_PP_ROWS; ++i) asm volatile("" : "+v"(v[i]));
#pragma unroll
                for (int i = 0; i < NH_PP_ROWS; ++i) {
                    *reinterpret_cast<u16x8_t*>(yp + off) = apply_row(v[i], mean, ga, be);
                    off += step;
                }
            } else {                                              // short last chunk: rows from global memory again
                int pitch;
                const char* xp = reinterpret_cast<const char*>(nh_lane_src(a, it.n, cc * 8, &pitch));
                float ta[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) ta[j] = 0.f;
                if (HAS_ADD) nh_param8(a.addnc, a.add_dt, (size_t)it.n * a.C + cc * 8, 0.f, ta);
                unsigned xoff = (unsigned)(it.row0 + r) * (unsigned)pitch * 2u;
                const unsigned xstep = (unsigned)a.R * (unsigned)pitch * 2u;
                for (int row = it.row0 + r; row < it.row1; row += a.R, off += step, xoff += xstep)
                    *reinterpret_cast<u16x8_t*>(yp + off) = apply_row(row_from_memory(xp, xoff, ta), mean, ga, be);
            }
        }
        __syncthreads();                                          // shS is rewritten by the next finish / merge
    };

    // ONE copy of each phase in the code: the item whose statistics were just published moves to the `prev` registers
    // (64 v_mov per item of ~60 KB) instead of the loop being unrolled over two register sets.
    u16x8_t cur[NH_PP_ROWS], prev[NH_PP_ROWS];
    Item ic = item(0), ip = ic;
    load(cur, ic);
    for (int k = 0; k <= nst; ++k) {
        t = threadIdx.x;
        asm volatile("" : "+v"(t));
        r = t / a.C8;
        cc = t - r * a.C8;
        active = r < a.R;
        if (k < nst) stats(cur, ic);
        if (k > 0) finish(prev, ip);
        if (k < nst) {
#pragma unroll
            for (int i = 0; i < NH_PP_ROWS; ++i) {
                asm volatile("" : "+v"(cur[i]));
                prev[i] = cur[i];
            }
            ip = ic;
            if (k + 1 < nst) {
                ic = item(k + 1);
                load(cur, ic);
            }
        }
    }
}

#endif  // PGN_TUNING_HOOKS (pgn_gn_nhwc_pipe_kernel)

static size_t nh_onepass_counter_bytes_fwd(int N) { return (((size_t)N * sizeof(unsigned int)) + 255) & ~(size_t)255; }

// workgroups aimed at per launch, minimum rows per lane: swept on MI355X (scripts/gn_exp.py): fewer, longer workgroups
#ifdef PGN_TUNING_HOOKS
unsigned long long* g_nh_dbg = nullptr;
int g_nh_reverse = 1;
int g_apply_variant = 0; // rows in flight / register budget of pgn_gn_nhwc_apply_kernel (scripts/gn_apply_sweep.py)
int g_brs_variant = 0;   // rows in flight / register budget of pgn_bias_res_stats_kernel (scripts/producer_stats_exp.py)
int g_nh_target_wgs = 768, g_nh_min_rows_per_lane = 24;
#else
constexpr int g_nh_target_wgs = 768, g_nh_min_rows_per_lane = 24;
constexpr int g_nh_reverse = 1;
#endif

// row-chunk geometry of the statistics / apply launches: a function of (N, C, HW) alone, so the launch that PRODUCES
// a tensor can emit the partial statistics in it without knowing the consuming GroupNorm's group count
static bool nh_chunking(int N, int C, int HW, NhArgs* a) {
    if (N <= 0 || C <= 0 || HW <= 0 || (C & 7) != 0 || C / 8 > NH_MAX_THREADS) return false;
    a->C = C; a->HW = HW; a->C8 = C / 8;
    int R = NH_MAX_THREADS / a->C8;
    if (R > HW) R = HW;
    if (R > 64) R = 64;
    a->R = R;
    long long target = g_nh_target_wgs / N;      // ~8 workgroups per CU over the whole launch
    if (target < 1) target = 1;
    if (target > 1024) target = 1024;
    long long rows = (HW + target - 1) / target;
    if (rows < (long long)g_nh_min_rows_per_lane * R) rows = (long long)g_nh_min_rows_per_lane * R;   // amortise the LDS reduction
    if (rows > HW) rows = HW;
    a->rows = (int)rows;
    a->S = (int)((HW + rows - 1) / rows);
    return true;
}

// geometry shared by the workspace query and the launch
static bool nh_geometry(int N, int C, int HW, int G, NhArgs* a) {
    if (N <= 0 || C <= 0 || HW <= 0 || G <= 0 || (C % G) != 0) return false;
    if ((C & 7) != 0 || C / 8 > NH_MAX_THREADS || C / G > 128) return false;
    if (!nh_chunking(N, C, HW, a)) return false;
    a->G = G; a->cpg = C / G;
    return true;
}

// one-pass form: R row lanes x NH_OP_ROWS rows per lane per workgroup, at most 512 threads (two workgroups per CU at
// <= 128 VGPRs).  False: the shape is outside the form (the three-pass form runs).
static bool nh_geometry_onepass(int N, int C, int HW, int G, NhArgs* a, int* threads) {
    if (N <= 0 || C <= 0 || HW <= 0 || G <= 0 || (C % G) != 0) return false;
    if ((C & 7) != 0 || C / 8 > 512 || G > NH_OP_MAX_G) return false;
    a->C = C; a->HW = HW; a->G = G; a->cpg = C / G; a->C8 = C / 8;
    int R = 512 / a->C8;
    if (R > HW) R = HW;
    if (R > 64) R = 64;
    a->R = R;
    a->rows = NH_OP_ROWS * R;
    a->S = (HW + a->rows - 1) / a->rows;
    *threads = ((R * a->C8 + 63) / 64) * 64;
    if (*threads < 64) *threads = 64;
    if (*threads < G) return false;
    return (long long)N * a->S <= 0x7fffffffLL;
}

#ifdef PGN_TUNING_HOOKS
// pipelined form: workgroups of at most 256 threads (two per CU at 256 VGPRs: two items of NH_PP_ROWS rows per lane in
// registers and room for the phases' working set - at 512 threads / 128 VGPRs the persistent loop spilled a whole item)
static bool nh_geometry_pipe(int N, int C, int HW, int G, NhArgs* a, int* threads) {
    if (N <= 0 || C <= 0 || HW <= 0 || G <= 0 || (C % G) != 0) return false;
    if ((C & 7) != 0 || C / 8 > 256 || G > NH_OP_MAX_G) return false;
    if ((long long)HW * C * 2 >= (1LL << 32)) return false;          // 32-bit byte offsets inside a sample
    a->C = C; a->HW = HW; a->G = G; a->cpg = C / G; a->C8 = C / 8;
    int R = 256 / a->C8;
    if (R > HW) R = HW;
    if (R > 64) R = 64;
    a->R = R;
    a->rows = NH_PP_ROWS * R;
    a->S = (HW + a->rows - 1) / a->rows;
    *threads = ((R * a->C8 + 63) / 64) * 64;
    if (*threads < 64) *threads = 64;
    if (*threads < G) return false;
    return (long long)N * a->S <= 0x7fffffffLL;
}
static size_t nh_pipe_lds(const NhArgs& a, int threads) {
    return (size_t)a.R * a.C * sizeof(float) + ((size_t)threads + NH_OP_MAX_G) * sizeof(float2) + NH_OP_MAX_G * sizeof(float) + 16;
}
// workspace of the pipelined form: [N] arrival counters (zeroed per call), [N][G] statistics words, [N][S][G] chunk words
static size_t nh_pipe_workspace(int N, int G, int S) {
    return nh_onepass_counter_bytes_fwd(N) + (size_t)N * G * 8 + (size_t)N * S * G * 8;
}
#endif

static size_t nh_onepass_lds(const NhArgs& a, int threads) {
    return ((size_t)a.R * a.C + (size_t)threads + NH_OP_MAX_G) * sizeof(float2) + NH_OP_MAX_G * sizeof(float) + 16;
}

// workspace of the one-pass form: [N] arrival counters in a block of their own (padded to 256 bytes: the per-call memset
// zeroes exactly that block), then the [N][S][G] words
static size_t nh_onepass_counter_bytes(int N) { return (((size_t)N * sizeof(unsigned int)) + 255) & ~(size_t)255; }

static size_t nh_workspace_need(int N, int C, int HW, int G) {
    NhArgs a;
    if (!nh_geometry(N, C, HW, G, &a)) return 0;
    size_t need = ((size_t)N * a.S * C + (size_t)N * G) * sizeof(float2);
    NhArgs b;
    int threads;
    if (nh_geometry_onepass(N, C, HW, G, &b, &threads)) {
        const size_t n1 = nh_onepass_counter_bytes(N) + (size_t)N * b.S * G * sizeof(unsigned long long);
        if (n1 > need) need = n1;
    }
#ifdef PGN_TUNING_HOOKS
    if (nh_geometry_pipe(N, C, HW, G, &b, &threads)) {
        const size_t n2 = nh_pipe_workspace(N, G, b.S);
        if (n2 > need) need = n2;
    }
#endif
    return need;
}

// compute units of the current device (the one-pass form is only picked when a sample's workgroups fit the chip)
static int nh_device_cus() {
    static int cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cus[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        cus[dev] = v;
    }
    return cus[dev];
}

static bool nh_ranges_overlap(const void* p, size_t pb, const void* q, size_t qb) {
    const uintptr_t a0 = (uintptr_t)p, b0 = (uintptr_t)q;
    return a0 < b0 + qb && b0 < a0 + pb;
}

}  // namespace

#ifdef PGN_TUNING_HOOKS
// measurement build only (make tuning): geometry sweep hook of scripts/gn_exp.py; absent from the shipped library
extern "C" void pgn_groupnorm_nhwc_debug_buffer(void* p) { g_nh_dbg = (unsigned long long*)p; }
extern "C" void pgn_groupnorm_nhwc_reverse(int on) { g_nh_reverse = on; }
extern "C" void pgn_bias_res_stats_variant(int v) { g_brs_variant = v; }
extern "C" void pgn_gn_apply_variant(int v) { g_apply_variant = v; }
extern "C" void pgn_groupnorm_nhwc_tuning(int target_wgs, int min_rows_per_lane) {
    g_nh_target_wgs = target_wgs;
    g_nh_min_rows_per_lane = min_rows_per_lane;
}
#endif

extern "C" size_t pgn_groupnorm_nhwc_workspace_bytes(int N, int C, int HW, int G) {
    return nh_workspace_need(N, C, HW, G);
}

// the apply launch (grid, block, s in scope); the measurement build can pick rows in flight / register budget
#ifdef PGN_TUNING_HOOKS
#define PGN_NH_APPLY(DT, ADD, ACT, ARGS)                                                                             \
    do {                                                                                                             \
        if (g_apply_variant == 1) hipLaunchKernelGGL((pgn_gn_nhwc_apply_kernel<DT, ADD, ACT, 8, 4>), grid, block, 0, s, ARGS);      \
        else if (g_apply_variant == 2) hipLaunchKernelGGL((pgn_gn_nhwc_apply_kernel<DT, ADD, ACT, 2, 8>), grid, block, 0, s, ARGS); \
        else if (g_apply_variant == 3) hipLaunchKernelGGL((pgn_gn_nhwc_apply_kernel<DT, ADD, ACT, 2, 4>), grid, block, 0, s, ARGS); \
        else if (g_apply_variant == 4) hipLaunchKernelGGL((pgn_gn_nhwc_apply_kernel<DT, ADD, ACT, 4, 8>), grid, block, 0, s, ARGS); \
        else hipLaunchKernelGGL((pgn_gn_nhwc_apply_kernel<DT, ADD, ACT, 4, 4>), grid, block, 0, s, ARGS);            \
    } while (0)
#else
#define PGN_NH_APPLY(DT, ADD, ACT, ARGS)                                                                             \
    hipLaunchKernelGGL((pgn_gn_nhwc_apply_kernel<DT, ADD, ACT, 4, 4>), grid, block, 0, s, ARGS)
#endif

// pre1 / pre2: partial statistics the producers of x / x2 already wrote (pgn_bias_residual_add_stats), or nullptr: the
// statistics launch runs here
static int nh_run(const void* x, int C1, const void* x2, void* y, int dtype, const void* gamma, const void* beta,
                  int affine_dtype, const void* add_nc, int add_dtype, int N, int C, int HW, int G,
                  float eps, int act, void* workspace, size_t workspace_bytes, void* stream,
                  const void* pre1 = nullptr, const void* pre2 = nullptr) {
    if (x == nullptr || y == nullptr || workspace == nullptr) return PGN_ERR_INVALID;
    if (C1 <= 0 || C1 > C || (C1 < C) != (x2 != nullptr)) return PGN_ERR_INVALID;
    if ((C1 & 7) != 0) return PGN_ERR_UNSUPPORTED;
    if (x2 != nullptr && !pgn_aligned(x2, 16)) return PGN_ERR_INVALID;
    if (N <= 0 || C <= 0 || HW <= 0 || G <= 0 || (C % G) != 0) return PGN_ERR_INVALID;
    if ((gamma == nullptr) != (beta == nullptr)) return PGN_ERR_INVALID;
    if (gamma != nullptr && pgn_dtype_size(affine_dtype) == 0) return PGN_ERR_INVALID;
    if (add_nc != nullptr && pgn_dtype_size(add_dtype) == 0) return PGN_ERR_INVALID;
    // `act`: bit 0 = SiLU, bits 1-3 = the caller's choice of form (PGN_GN_ONEPASS_*); NEVER and ALWAYS exclude each other
    if ((act & ~(PGN_GN_ACT_SILU | PGN_GN_ONEPASS_NEVER | PGN_GN_ONEPASS_ALWAYS | PGN_GN_ONEPASS_NO_WAIT |
                 PGN_GN_PIPELINE_NEVER | PGN_GN_PIPELINE_ALWAYS)) != 0 ||
        ((act & PGN_GN_ONEPASS_NEVER) && (act & PGN_GN_ONEPASS_ALWAYS)) ||
        ((act & PGN_GN_PIPELINE_NEVER) && (act & PGN_GN_PIPELINE_ALWAYS)) || !(eps >= 0.f))
        return PGN_ERR_INVALID;
    const int form = act;
    act &= PGN_GN_ACT_SILU;
    if (dtype != PGN_BF16 && dtype != PGN_F16) return PGN_ERR_UNSUPPORTED;
    NhArgs a;
    if (!nh_geometry(N, C, HW, G, &a)) return PGN_ERR_UNSUPPORTED;
    if ((long long)N * a.S > 0x7fffffffLL || (long long)N * HW * C > 0x7fffffffffffLL) return PGN_ERR_UNSUPPORTED;
    if (!pgn_aligned(x, 16) || !pgn_aligned(y, 16) || !pgn_aligned(workspace, 8)) return PGN_ERR_INVALID;
    // gamma / beta / add_nc rows are read as 16-byte vectors
    if ((gamma != nullptr && (!pgn_aligned(gamma, 16) || !pgn_aligned(beta, 16))) ||
        (add_nc != nullptr && !pgn_aligned(add_nc, 16)))
        return PGN_ERR_INVALID;
    const bool pre = pre1 != nullptr;
    const size_t need = pre ? (size_t)N * G * sizeof(float2) : nh_workspace_need(N, C, HW, G);
    if (workspace_bytes < need) return PGN_ERR_INVALID;
    NhArgs h1, h2;   // the producers' chunk geometry (pre only)
    if (pre) {
        if (add_nc != nullptr || (x2 != nullptr) != (pre2 != nullptr)) return PGN_ERR_INVALID;
        if (!pgn_aligned(pre1, 8) || (pre2 != nullptr && !pgn_aligned(pre2, 8))) return PGN_ERR_INVALID;
        if (!nh_chunking(N, C1, HW, &h1) || (x2 != nullptr && !nh_chunking(N, C - C1, HW, &h2))) return PGN_ERR_UNSUPPORTED;
    }
    a.x = x; a.x2 = x2; a.C1 = C1; a.y = y; a.gamma = gamma; a.beta = beta; a.addnc = add_nc; a.adt = affine_dtype; a.add_dt = add_dtype;
    a.eps = eps; a.act = act;
    // Three-pass form: the apply pass re-reads what the statistics pass read.  Tensors of up to ~170 MB are still in
    // the 256 MiB Infinity Cache then (3.6 TB/s algorithmic); of a larger one only what was read LAST is, so its apply
    // pass walks the tensor backwards (336-503 MB tensors: +5..8 %, scripts/gn_reverse_ab.py; smaller ones lose 0-2 %
    // that way and keep the forward walk).  Same arithmetic per workgroup: bit-identical.  Processing the tensor in
    // sample ranges that fit the cache, statistics and apply back to back, was measured too and is 10-24 % SLOWER
    // (smaller launches, and the range's own output evicts the rows it has not re-read yet).
    a.reverse = (g_nh_reverse && (size_t)N * HW * C * pgn_dtype_size(dtype) > ((size_t)200 << 20)) ? 1 : 0;
    a.arrived = nullptr; a.gpart = nullptr; a.timeout = 0;
    a.part = nullptr; a.stats = nullptr; a.part2 = nullptr; a.partS = 0; a.partC = 0; a.part2S = 0;
#ifdef PGN_TUNING_HOOKS
    a.dbg = g_nh_dbg;
#endif
    hipStream_t s = (hipStream_t)stream;
    const bool has_add = add_nc != nullptr;
    if (pre) {
        // the producers' partials stand in for the statistics launch: finalize + apply, one read + one write
        a.part = nullptr; a.partS = h1.S; a.partC = C1; a.part2 = (const float2*)pre2; a.part2S = x2 != nullptr ? h2.S : 0;
        a.stats = (float2*)workspace;
        NhArgs f = a;
        f.part = (float2*)const_cast<void*>(pre1);
        const int threads = ((a.R * a.C8 + 63) / 64) * 64;
        const dim3 grid((unsigned)(N * a.S)), block(threads);
#define PGN_NH_PRE(DT)                                                                                               \
    do {                                                                                                             \
        hipLaunchKernelGGL((pgn_gn_nhwc_finalize_kernel<DT>), dim3(N * G), dim3(NH_FIN_THREADS), 0, s, f);           \
        if (act) PGN_NH_APPLY(DT, false, true, f);                                                                   \
        else PGN_NH_APPLY(DT, false, false, f);                                                                      \
    } while (0)
        if (dtype == PGN_BF16) PGN_NH_PRE(PGN_BF16);
        else PGN_NH_PRE(PGN_F16);
#undef PGN_NH_PRE
        PGN_LAUNCH_CHECK();
        return PGN_OK;
    }
#ifdef PGN_TUNING_HOOKS
    {
        // Pipelined one-pass form (1R + 1W, persistent workgroups): measurement build only, and only on request
        // (PGN_GN_PIPELINE_ALWAYS) - it measured 0.35-0.49x of the three launches: see pgn_gn_nhwc_pipe_kernel.
        NhArgs b = a;
        int threads1 = 0;
        const int cus = nh_device_cus();
        const size_t es = pgn_dtype_size(dtype);
        const size_t yb = (size_t)N * HW * C * es, x1b = (size_t)N * HW * C1 * es, x2b = (size_t)N * HW * (C - C1) * es;
        bool pipe = !(form & PGN_GN_PIPELINE_NEVER) && nh_geometry_pipe(N, C, HW, G, &b, &threads1) && cus > 0 &&
                    (form & PGN_GN_PIPELINE_ALWAYS) &&
                    nh_pipe_lds(b, threads1) <= 64 * 1024 && !nh_ranges_overlap(x, x1b, y, yb) &&
                    !(x2 != nullptr && nh_ranges_overlap(x2, x2b, y, yb));
        // a workgroup must never hold two chunks of one sample more than one step apart (it would wait for a word it
        // publishes itself two steps later): S <= P
        const long long resident = NH_PP_WGS_PER_CU * cus;
        if (pipe && (long long)b.S > ((long long)N * b.S < resident ? (long long)N * b.S : resident)) pipe = false;
        if (pipe) {
            const long long items = (long long)N * b.S;
            const int P = (int)(items < resident ? items : resident);
            b.timeout = (form & PGN_GN_ONEPASS_NO_WAIT) ? 0u : 200000u;
            b.arrived = (unsigned int*)workspace;
            b.stats = (float2*)((char*)workspace + nh_onepass_counter_bytes(N));
            b.gpart = (unsigned long long*)((char*)b.stats + (size_t)N * G * 8);
            const hipError_t me = hipMemsetAsync(workspace, 0, nh_onepass_counter_bytes(N), s);
            if (me != hipSuccess) return (int)me;
            const dim3 gridp((unsigned)P), blockp(threads1);
            const size_t ldsp = nh_pipe_lds(b, threads1);
            const int n_items = (int)items;
#define PGN_NH_PIPE(DT)                                                                                               \
    do {                                                                                                              \
        if (has_add && act) hipLaunchKernelGGL((pgn_gn_nhwc_pipe_kernel<DT, true, true>), gridp, blockp, ldsp, s, b, n_items);   \
        else if (has_add) hipLaunchKernelGGL((pgn_gn_nhwc_pipe_kernel<DT, true, false>), gridp, blockp, ldsp, s, b, n_items);    \
        else if (act) hipLaunchKernelGGL((pgn_gn_nhwc_pipe_kernel<DT, false, true>), gridp, blockp, ldsp, s, b, n_items);        \
        else hipLaunchKernelGGL((pgn_gn_nhwc_pipe_kernel<DT, false, false>), gridp, blockp, ldsp, s, b, n_items);                \
    } while (0)
            if (dtype == PGN_BF16) PGN_NH_PIPE(PGN_BF16);
            else PGN_NH_PIPE(PGN_F16);
#undef PGN_NH_PIPE
            PGN_LAUNCH_CHECK();
            return PGN_OK;
        }
    }
#endif
    {
        // One-pass form (1R + 1W, one launch) for tensors of up to NH_OP_MAX_BYTES: there it beats the three launches
        // (18-25 us against 28-31 us at the UNet's shapes for batches 1-4, scripts/gn_onepass_ab.py).  On larger
        // tensors every CU is streaming, the exchange between a sample's workgroups takes 20-30 us under that load and
        // the 368 KB a CU holds in registers cannot cover it: measured 0.70-0.87x of the three-pass form at batch 32
        // (profiles/r02_groupnorm_onepass_ab.txt), so those keep the three passes.  Also required: a sample's S
        // workgroups fit the chip at least twice over, x and y do not overlap.  The caller may override the size rule
        // through `act`: PGN_GN_ONEPASS_NEVER, PGN_GN_ONEPASS_ALWAYS (whenever the shape allows: tests, measurements),
        // PGN_GN_ONEPASS_NO_WAIT (wait bound 0: every workgroup takes the safety net; tests).  No environment is read
        // in here: the library has no process-wide switches (pigeon_amd/ops.py maps its own onto these bits).
        NhArgs b = a;
        int threads1 = 0;
        bool onepass = !(form & PGN_GN_ONEPASS_NEVER) && nh_geometry_onepass(N, C, HW, G, &b, &threads1);
        if (onepass) {
            const int cus = nh_device_cus();
            const long long slots = 2LL * cus;
            const size_t es = pgn_dtype_size(dtype);
            const size_t yb = (size_t)N * HW * C * es, x1b = (size_t)N * HW * C1 * es, x2b = (size_t)N * HW * (C - C1) * es;
            const bool forced = (form & PGN_GN_ONEPASS_ALWAYS) != 0;
            onepass = cus > 0 && 2LL * b.S <= slots && nh_onepass_lds(b, threads1) <= 64 * 1024 &&
                      (forced || yb <= NH_OP_MAX_BYTES) &&
                      !nh_ranges_overlap(x, x1b, y, yb) && !(x2 != nullptr && nh_ranges_overlap(x2, x2b, y, yb));
        }
        if (onepass) {
            b.timeout = (form & PGN_GN_ONEPASS_NO_WAIT) ? 0u : 200000u;   // 2 ms of 100 MHz ticks
            b.arrived = (unsigned int*)workspace;
            b.gpart = (unsigned long long*)((char*)workspace + nh_onepass_counter_bytes(N));
            const hipError_t me = hipMemsetAsync(workspace, 0, nh_onepass_counter_bytes(N), s);
            if (me != hipSuccess) return (int)me;
            const dim3 grid1((unsigned)(N * b.S)), block1(threads1);
            const size_t lds1 = nh_onepass_lds(b, threads1);
#define PGN_NH_ONEPASS(DT)                                                                                            \
    do {                                                                                                              \
        if (has_add && act) hipLaunchKernelGGL((pgn_gn_nhwc_onepass_kernel<DT, true, true>), grid1, block1, lds1, s, b);       \
        else if (has_add) hipLaunchKernelGGL((pgn_gn_nhwc_onepass_kernel<DT, true, false>), grid1, block1, lds1, s, b);        \
        else if (act) hipLaunchKernelGGL((pgn_gn_nhwc_onepass_kernel<DT, false, true>), grid1, block1, lds1, s, b);            \
        else hipLaunchKernelGGL((pgn_gn_nhwc_onepass_kernel<DT, false, false>), grid1, block1, lds1, s, b);                    \
    } while (0)
            if (dtype == PGN_BF16) PGN_NH_ONEPASS(PGN_BF16);
            else PGN_NH_ONEPASS(PGN_F16);
#undef PGN_NH_ONEPASS
            PGN_LAUNCH_CHECK();
            return PGN_OK;
        }
    }
    a.part = (float2*)workspace;
    a.partS = a.S; a.partC = C; a.part2 = nullptr; a.part2S = 0;
    a.stats = a.part + (size_t)N * a.S * C;
    const int threads = ((a.R * a.C8 + 63) / 64) * 64;
    const size_t lds = (size_t)a.R * C * sizeof(float2);
    const dim3 grid((unsigned)(N * a.S)), block(threads);
#define PGN_NH_LAUNCH(DT)                                                                                            \
    do {                                                                                                             \
        if (has_add) hipLaunchKernelGGL((pgn_gn_nhwc_stats_kernel<DT, true>), grid, block, lds, s, a);               \
        else hipLaunchKernelGGL((pgn_gn_nhwc_stats_kernel<DT, false>), grid, block, lds, s, a);                      \
        hipLaunchKernelGGL((pgn_gn_nhwc_finalize_kernel<DT>), dim3(N * G), dim3(NH_FIN_THREADS), 0, s, a);           \
        if (has_add && act) PGN_NH_APPLY(DT, true, true, a);                                                         \
        else if (has_add) PGN_NH_APPLY(DT, true, false, a);                                                          \
        else if (act) PGN_NH_APPLY(DT, false, true, a);                                                              \
        else PGN_NH_APPLY(DT, false, false, a);                                                                      \
    } while (0)
    if (dtype == PGN_BF16) PGN_NH_LAUNCH(PGN_BF16);
    else PGN_NH_LAUNCH(PGN_F16);
#undef PGN_NH_LAUNCH
    PGN_LAUNCH_CHECK();
    return PGN_OK;
}

extern "C" int pgn_groupnorm_silu_nhwc(const void* x, void* y, int dtype, const void* gamma, const void* beta,
                                       int affine_dtype, const void* add_nc, int add_dtype, int N, int C, int HW, int G,
                                       float eps, int act, void* workspace, size_t workspace_bytes, void* stream) {
    return nh_run(x, C, nullptr, y, dtype, gamma, beta, affine_dtype, add_nc, add_dtype, N, C, HW, G, eps, act,
                  workspace, workspace_bytes, stream);
}

extern "C" int pgn_groupnorm_silu_nhwc_cat(const void* x1, int C1, const void* x2, int C2, void* y, int dtype,
                                           const void* gamma, const void* beta, int affine_dtype, const void* add_nc,
                                           int add_dtype, int N, int HW, int G, float eps, int act, void* workspace,
                                           size_t workspace_bytes, void* stream) {
    if (x2 == nullptr || C1 <= 0 || C2 <= 0) return PGN_ERR_INVALID;
    if ((C2 & 7) != 0) return PGN_ERR_UNSUPPORTED;
    return nh_run(x1, C1, x2, y, dtype, gamma, beta, affine_dtype, add_nc, add_dtype, N, C1 + C2, HW, G, eps, act,
                  workspace, workspace_bytes, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// Producer-side statistics (include/pigeon_hip.h: pgn_bias_residual_add_stats, pgn_groupnorm_silu_nhwc_pre)
extern "C" size_t pgn_bias_residual_add_stats_bytes(int N, int C, int HW) {
    NhArgs a;
    if (!nh_chunking(N, C, HW, &a) || (long long)N * a.S > 0x7fffffffLL) return 0;
    return (size_t)N * a.S * C * sizeof(float2);
}

extern "C" int pgn_bias_residual_add_stats(const void* x, const void* residual, const void* bias, int bias_dtype,
                                           void* y, int dtype, int N, int C, int HW, void* part, size_t part_bytes,
                                           void* stream) {
    if (x == nullptr || bias == nullptr || y == nullptr || part == nullptr) return PGN_ERR_INVALID;
    if (N <= 0 || C <= 0 || HW <= 0 || pgn_dtype_size(bias_dtype) == 0) return PGN_ERR_INVALID;
    if (dtype != PGN_BF16 && dtype != PGN_F16) return PGN_ERR_UNSUPPORTED;
    NhArgs a;
    if (!nh_chunking(N, C, HW, &a)) return PGN_ERR_UNSUPPORTED;
    if ((long long)N * a.S > 0x7fffffffLL || (long long)N * HW * C > 0x7fffffffffffLL) return PGN_ERR_UNSUPPORTED;
    if (!pgn_aligned(x, 16) || !pgn_aligned(y, 16) || !pgn_aligned(bias, 16) || !pgn_aligned(part, 8) ||
        (residual != nullptr && !pgn_aligned(residual, 16)))
        return PGN_ERR_INVALID;
    if (part_bytes < (size_t)N * a.S * C * sizeof(float2)) return PGN_ERR_INVALID;
    // every workgroup of a sample reads the pivot rows of the INPUTS: y may alias neither
    const size_t tb = (size_t)N * HW * C * pgn_dtype_size(dtype);
    if (nh_ranges_overlap(x, tb, y, tb) || (residual != nullptr && nh_ranges_overlap(residual, tb, y, tb)))
        return PGN_ERR_INVALID;
    a.x = x; a.x2 = nullptr; a.C1 = C; a.y = y; a.gamma = nullptr; a.beta = nullptr; a.addnc = nullptr;
    a.adt = PGN_F32; a.add_dt = PGN_F32; a.G = 0; a.cpg = 0; a.eps = 0.f; a.act = 0; a.reverse = 0;
    a.part = (float2*)part; a.stats = nullptr; a.part2 = nullptr; a.partS = a.S; a.partC = C; a.part2S = 0;
    a.arrived = nullptr; a.gpart = nullptr; a.timeout = 0;
#ifdef PGN_TUNING_HOOKS
    a.dbg = nullptr;
#endif
    hipStream_t s = (hipStream_t)stream;
    const int threads = ((a.R * a.C8 + 63) / 64) * 64;
    const size_t lds = (size_t)a.R * C * sizeof(float2);
    const dim3 grid((unsigned)(N * a.S)), block(threads);
#define PGN_BRS_V(DT, U, W)                                                                                          \
    do {                                                                                                             \
        if (residual) hipLaunchKernelGGL((pgn_bias_res_stats_kernel<DT, true, U, W>), grid, block, lds, s, a, residual, bias, bias_dtype);  \
        else hipLaunchKernelGGL((pgn_bias_res_stats_kernel<DT, false, U, W>), grid, block, lds, s, a, residual, bias, bias_dtype);          \
    } while (0)
#ifdef PGN_TUNING_HOOKS
#define PGN_BRS(DT)                                                                                                  \
    do {                                                                                                             \
        if (g_brs_variant == 1) PGN_BRS_V(DT, 4, 4);                                                                 \
        else if (g_brs_variant == 2) PGN_BRS_V(DT, 2, 8);                                                            \
        else if (g_brs_variant == 3) PGN_BRS_V(DT, 1, 8);                                                            \
        else PGN_BRS_V(DT, 2, 4);                                                                                    \
    } while (0)
#else
#define PGN_BRS(DT) PGN_BRS_V(DT, 2, 4)
#endif
    if (dtype == PGN_BF16) PGN_BRS(PGN_BF16);
    else PGN_BRS(PGN_F16);
#undef PGN_BRS
#undef PGN_BRS_V
    PGN_LAUNCH_CHECK();
    return PGN_OK;
}

extern "C" int pgn_groupnorm_silu_nhwc_pre(const void* x1, int C1, const void* part1, const void* x2, int C2,
                                           const void* part2, void* y, int dtype, const void* gamma, const void* beta,
                                           int affine_dtype, int N, int HW, int G, float eps, int act, void* workspace,
                                           size_t workspace_bytes, void* stream) {
    if (part1 == nullptr || C1 <= 0 || C2 < 0 || (C2 > 0) != (x2 != nullptr)) return PGN_ERR_INVALID;
    if ((C2 & 7) != 0) return PGN_ERR_UNSUPPORTED;
    if (act != 0 && act != PGN_GN_ACT_SILU) return PGN_ERR_INVALID;
    return nh_run(x1, C1, x2, y, dtype, gamma, beta, affine_dtype, nullptr, PGN_F32, N, C1 + C2, HW, G, eps, act,
                  workspace, workspace_bytes, stream, part1, part2);
}

// ---------------------------------------------------------------------------------------------------------------------
// Allocation-free variant (include/pigeon_hip.h: pgn_groupnorm_silu_nhwc_static).  The partial-statistics workspace the
// three kernels exchange lives in a file-scope device array instead of caller memory, so the call path contains no
// hipMalloc / hipFree / allocating library routine.  Every GPU the library runs on holds its own copy of the array at
// its own address until the process ends; `scratch` below is only the host-side handle of the symbol, never a device
// address: the per-device address is resolved once with hipGetSymbolAddress and only that pointer is handed to kernels.
static __device__ __align__(256) unsigned char scratch[PGN_GN_STATIC_SCRATCH_BYTES];

namespace {
constexpr int NH_MAX_DEVICES = 64;
void* g_scratch_dev_ptr[NH_MAX_DEVICES];   // device ordinal -> address of that device's copy of `scratch`
std::mutex g_scratch_mu;

// address of the current device's scratch array (resolved on the first call on that device, with it current)
int nh_scratch_ptr(void** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= NH_MAX_DEVICES) return PGN_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_dev_ptr[dev] == nullptr) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return (int)e;
        if (p == nullptr) return (int)hipErrorInvalidSymbol;
        g_scratch_dev_ptr[dev] = p;
    }
    *out = g_scratch_dev_ptr[dev];
    return PGN_OK;
}
}  // namespace

extern "C" int pgn_groupnorm_silu_nhwc_static(const void* x, void* y, int dtype, const void* gamma, const void* beta,
                                              int affine_dtype, const void* add_nc, int add_dtype, int N, int C, int HW,
                                              int G, float eps, int act, void* stream) {
    if (x == nullptr || y == nullptr) return PGN_ERR_INVALID;
    if (N <= 0 || C <= 0 || HW <= 0 || G <= 0 || (C % G) != 0) return PGN_ERR_INVALID;
    NhArgs a;
    if (!nh_geometry(N, C, HW, G, &a)) return PGN_ERR_UNSUPPORTED;
    const size_t need = nh_workspace_need(N, C, HW, G);
    if (need > (size_t)PGN_GN_STATIC_SCRATCH_BYTES) return PGN_ERR_SCRATCH;
    void* ws = nullptr;
    const int rc = nh_scratch_ptr(&ws);
    if (rc != PGN_OK) return rc;
    // the array was resolved for the CURRENT device: a stream of another GPU would hand its kernels the wrong copy
    if (stream != nullptr) {
        int cur = -1, sdev = -1;
        if (hipGetDevice(&cur) != hipSuccess) return PGN_ERR_INVALID;
        if (hipStreamGetDevice((hipStream_t)stream, &sdev) == hipSuccess && sdev != cur) return PGN_ERR_INVALID;
    }
    // three-launch form: every workspace word the finalize / apply launches read is written by the launch before it
    // on the same stream; one-launch form: nh_run zeroes the arrival counters with hipMemsetAsync on the same stream
    // and every partial-sum word is published before it is read.  So nothing depends on what the previous call left.
    return nh_run(x, C, nullptr, y, dtype, gamma, beta, affine_dtype, add_nc, add_dtype, N, C, HW, G, eps, act, ws,
                  (size_t)PGN_GN_STATIC_SCRATCH_BYTES, stream);
}

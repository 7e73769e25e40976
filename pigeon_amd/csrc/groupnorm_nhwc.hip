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
//                       (per-CHANNEL pivots p_c = x[n, 0, c]: sum_x (x-mu)^2 = sum_c q_c + 2 (p_c-mu) s_c + HW (p_c-mu)^2)
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
        nh_load8<DT>(xp, tadd, has_add, piv);   // row 0 of the sample: one pivot per channel
        const int row1 = min(a.HW, (sc + 1) * a.rows);
        auto accum = [&](const u16x8_t& v) {
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
        };
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
        const float2* p = a.part + (size_t)n * a.S * a.C + (size_t)g * a.cpg + cl;
        for (int sc = ls; sc < a.S; sc += L) {
            const float2 v = p[(size_t)sc * a.C];
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
        p0 = Cvt<DT>::to_f32(*nh_lane_src(a, n, c, &pitch));      // per-channel pivot: row 0 of the sample
        if (a.addnc != nullptr)
            p0 = Cvt<DT>::to_f32(Cvt<DT>::from_f32(p0 + nh_param(a.addnc, a.add_dt, (size_t)n * a.C + c, 0.f)));
    }
    const float mean = nh_fin_block_sum(v0 ? p0 + S0 / hw : 0.f, red) / (float)a.cpg;
    float var = 0.f;
    if (v0) { const float d = p0 - mean; var = Q0 + 2.f * d * S0 + hw * d * d; }
    var = fmaxf(nh_fin_block_sum(var, red) / ((float)a.cpg * hw), 0.f);
    if (t == 0) a.stats[(size_t)n * a.G + g] = make_float2(mean, 1.0f / sqrtf(var + a.eps));
}

template <int DT, bool HAS_ADD, bool ACT>
__global__ __launch_bounds__(NH_MAX_THREADS) void pgn_gn_nhwc_apply_kernel(NhArgs a) {
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
    for (; row + 3 * a.R < row1; row += 4 * a.R) {
        const size_t e0 = img + (size_t)row * a.C + cc * 8, st = (size_t)a.R * a.C;
        const size_t i0 = (size_t)row * pitch, si = (size_t)a.R * pitch;
        const u16x8_t v0 = *reinterpret_cast<const u16x8_t*>(xp + i0);
        const u16x8_t v1 = *reinterpret_cast<const u16x8_t*>(xp + i0 + si);
        const u16x8_t v2 = *reinterpret_cast<const u16x8_t*>(xp + i0 + 2 * si);
        const u16x8_t v3 = *reinterpret_cast<const u16x8_t*>(xp + i0 + 3 * si);
        apply(v0, e0); apply(v1, e0 + st); apply(v2, e0 + 2 * st); apply(v3, e0 + 3 * st);
    }
    for (; row < row1; row += a.R) {
        const size_t e = img + (size_t)row * a.C + cc * 8;
        apply(*reinterpret_cast<const u16x8_t*>(xp + (size_t)row * pitch), e);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// One-pass form: 1 read + 1 write of the tensor.  A workgroup keeps its row chunk in registers (NH_OP_ROWS 16-byte
// loads per lane), publishes its per-group partial sums, waits until every workgroup of its sample has published,
// reduces the S x G partials in a fixed order itself and applies from registers.  The statistics are pivot-shifted sums
// about ONE pivot per (sample, group) - the group's first channel in row 0, which every workgroup can read for itself -
// so that a workgroup's contribution is G words, not C.
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
    for (int g = t; g < a.G; g += blockDim.x) {                   // one pivot per (sample, group)
        const int c0 = g * a.cpg;
        int pp;
        float p = Cvt<DT>::to_f32(*nh_lane_src(a, n, c0, &pp));
        if (HAS_ADD) p = Cvt<DT>::to_f32(Cvt<DT>::from_f32(p + nh_param(a.addnc, a.add_dt, (size_t)n * a.C + c0, 0.f)));
        shP[g] = p;
    }
    __syncthreads();
    float piv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) piv[j] = active ? shP[(cc * 8 + j) / a.cpg] : 0.f;

    f32x2_t s2[4], q2[4];
    auto clear = [&]() {
#pragma unroll
        for (int p = 0; p < 4; ++p) { s2[p] = f32x2_t{0.f, 0.f}; q2[p] = f32x2_t{0.f, 0.f}; }
    };
    auto accum = [&](const u16x8_t& w) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            f32x2_t f = {Cvt<DT>::to_f32(w[2 * p]), Cvt<DT>::to_f32(w[2 * p + 1])};
            if constexpr (HAS_ADD) {
                f += f32x2_t{tadd[2 * p], tadd[2 * p + 1]};
                f = f32x2_t{Cvt<DT>::to_f32(Cvt<DT>::from_f32(f[0])), Cvt<DT>::to_f32(Cvt<DT>::from_f32(f[1]))};
            }
            const f32x2_t d = f - f32x2_t{piv[2 * p], piv[2 * p + 1]};
            s2[p] += d;
            q2[p] += d * d;
        }
    };
    // lanes' sums -> per-channel -> per-group, published as chunk `chunk` of the sample; ends behind a barrier with
    // every storing wave drained
    auto fold_publish = [&](int chunk) {
        if (active) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                shc[(size_t)r * a.C + cc * 8 + j] = make_float2(s2[j >> 1][j & 1], q2[j >> 1][j & 1]);
        }
        __syncthreads();
        for (int c = t; c < a.C; c += blockDim.x) {               // column c belongs to this thread alone
            float ss = 0.f, qq = 0.f;
            for (int rr = 0; rr < a.R; ++rr) {
                const float2 w = shc[(size_t)rr * a.C + c];
                ss += w.x;
                qq += w.y;
            }
            shc[c] = make_float2(ss, qq);
        }
        __syncthreads();
        for (int g = t; g < a.G; g += blockDim.x) {
            float ss = 0.f, qq = 0.f;
            for (int cl = 0; cl < a.cpg; ++cl) {
                const float2 w = shc[g * a.cpg + cl];
                ss += w.x;
                qq += w.y;
            }
            const unsigned long long word = ((unsigned long long)__float_as_uint(qq) << 32) | __float_as_uint(ss);
            __hip_atomic_store(((nh_gu64*)a.gpart) + ((size_t)n * a.S + chunk) * a.G + g, word,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    clear();
#pragma unroll
    for (int i = 0; i < NH_OP_ROWS; ++i)
        if (i < nv) accum(v[i]);
    NH_STAMP(1);
    fold_publish(sc);
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
        // safety net: the peers have not all published within the bound; compute their chunks' words here
        for (int chunk = 0; chunk < a.S; ++chunk) {
            if (chunk == sc) continue;
            clear();
            if (active) {
                const int rb = min(a.HW, (chunk + 1) * a.rows);
                for (int row = chunk * a.rows + r; row < rb; row += a.R)
                    accum(*reinterpret_cast<const u16x8_t*>(xp + (size_t)row * pitch));
            }
            fold_publish(chunk);
        }
    }
    if (t == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {   // S x G words -> G statistics, fixed order: chunk lanes stride over the chunks, then the lanes in order
        const int L = blockDim.x / a.G;
        const int ls = t / a.G, g = t - ls * a.G;
        if (ls < L) {
            float ss = 0.f, qq = 0.f;
            const nh_gu64* p = ((const nh_gu64*)a.gpart) + (size_t)n * a.S * a.G + g;
            for (int c2 = ls; c2 < a.S; c2 += 8 * L) {            // eight loads in flight, summed in chunk order
                unsigned long long word[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    word[u] = c2 + u * L < a.S ? __hip_atomic_load(p + (size_t)(c2 + u * L) * a.G, __ATOMIC_RELAXED,
                                                                  __HIP_MEMORY_SCOPE_AGENT)
                                               : 0ull;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (c2 + u * L < a.S) {
                        ss += __uint_as_float((unsigned)word[u]);
                        qq += __uint_as_float((unsigned)(word[u] >> 32));
                    }
                }
            }
            shL[ls * a.G + g] = make_float2(ss, qq);
        }
        __syncthreads();
        if (t < a.G) {
            float ss = 0.f, qq = 0.f;
            for (int l = 0; l < L; ++l) {
                const float2 w = shL[l * a.G + t];
                ss += w.x;
                qq += w.y;
            }
            const float cnt = (float)a.cpg * (float)a.HW;
            const float md = ss / cnt;
            const float var = fmaxf(qq / cnt - md * md, 0.f);
            shS[t] = make_float2(shP[t] + md, 1.0f / sqrtf(var + a.eps));
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
                f32x2_t f = {Cvt<DT>::to_f32(v[i][2 * p]), Cvt<DT>::to_f32(v[i][2 * p + 1])};
                if constexpr (HAS_ADD) {
                    f += f32x2_t{tadd[2 * p], tadd[2 * p + 1]};
                    f = f32x2_t{Cvt<DT>::to_f32(Cvt<DT>::from_f32(f[0])), Cvt<DT>::to_f32(Cvt<DT>::from_f32(f[1]))};
                }
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

// workgroups aimed at per launch, minimum rows per lane: swept on MI355X (scripts/gn_exp.py): fewer, longer workgroups
#ifdef PGN_TUNING_HOOKS
unsigned long long* g_nh_dbg = nullptr;
int g_nh_reverse = 1;
int g_nh_target_wgs = 768, g_nh_min_rows_per_lane = 24;
#else
constexpr int g_nh_target_wgs = 768, g_nh_min_rows_per_lane = 24;
constexpr int g_nh_reverse = 1;
#endif

// geometry shared by the workspace query and the launch
static bool nh_geometry(int N, int C, int HW, int G, NhArgs* a) {
    if (N <= 0 || C <= 0 || HW <= 0 || G <= 0 || (C % G) != 0) return false;
    if ((C & 7) != 0 || C / 8 > NH_MAX_THREADS || C / G > 128) return false;
    a->C = C; a->HW = HW; a->G = G; a->cpg = C / G; a->C8 = C / 8;
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
extern "C" void pgn_groupnorm_nhwc_tuning(int target_wgs, int min_rows_per_lane) {
    g_nh_target_wgs = target_wgs;
    g_nh_min_rows_per_lane = min_rows_per_lane;
}
#endif

extern "C" size_t pgn_groupnorm_nhwc_workspace_bytes(int N, int C, int HW, int G) {
    return nh_workspace_need(N, C, HW, G);
}

static int nh_run(const void* x, int C1, const void* x2, void* y, int dtype, const void* gamma, const void* beta,
                  int affine_dtype, const void* add_nc, int add_dtype, int N, int C, int HW, int G,
                  float eps, int act, void* workspace, size_t workspace_bytes, void* stream) {
    if (x == nullptr || y == nullptr || workspace == nullptr) return PGN_ERR_INVALID;
    if (C1 <= 0 || C1 > C || (C1 < C) != (x2 != nullptr)) return PGN_ERR_INVALID;
    if ((C1 & 7) != 0) return PGN_ERR_UNSUPPORTED;
    if (x2 != nullptr && !pgn_aligned(x2, 16)) return PGN_ERR_INVALID;
    if (N <= 0 || C <= 0 || HW <= 0 || G <= 0 || (C % G) != 0) return PGN_ERR_INVALID;
    if ((gamma == nullptr) != (beta == nullptr)) return PGN_ERR_INVALID;
    if (gamma != nullptr && pgn_dtype_size(affine_dtype) == 0) return PGN_ERR_INVALID;
    if (add_nc != nullptr && pgn_dtype_size(add_dtype) == 0) return PGN_ERR_INVALID;
    // `act`: bit 0 = SiLU, bits 1-3 = the caller's choice of form (PGN_GN_ONEPASS_*); NEVER and ALWAYS exclude each other
    if ((act & ~(PGN_GN_ACT_SILU | PGN_GN_ONEPASS_NEVER | PGN_GN_ONEPASS_ALWAYS | PGN_GN_ONEPASS_NO_WAIT)) != 0 ||
        ((act & PGN_GN_ONEPASS_NEVER) && (act & PGN_GN_ONEPASS_ALWAYS)) || !(eps >= 0.f))
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
    const size_t need = nh_workspace_need(N, C, HW, G);
    if (workspace_bytes < need) return PGN_ERR_INVALID;
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
#ifdef PGN_TUNING_HOOKS
    a.dbg = g_nh_dbg;
#endif
    hipStream_t s = (hipStream_t)stream;
    const bool has_add = add_nc != nullptr;
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
    a.stats = a.part + (size_t)N * a.S * C;
    const int threads = ((a.R * a.C8 + 63) / 64) * 64;
    const size_t lds = (size_t)a.R * C * sizeof(float2);
    const dim3 grid((unsigned)(N * a.S)), block(threads);
#define PGN_NH_LAUNCH(DT)                                                                                            \
    do {                                                                                                             \
        if (has_add) hipLaunchKernelGGL((pgn_gn_nhwc_stats_kernel<DT, true>), grid, block, lds, s, a);               \
        else hipLaunchKernelGGL((pgn_gn_nhwc_stats_kernel<DT, false>), grid, block, lds, s, a);                      \
        hipLaunchKernelGGL((pgn_gn_nhwc_finalize_kernel<DT>), dim3(N * G), dim3(NH_FIN_THREADS), 0, s, a);           \
        if (has_add && act) hipLaunchKernelGGL((pgn_gn_nhwc_apply_kernel<DT, true, true>), grid, block, 0, s, a);    \
        else if (has_add) hipLaunchKernelGGL((pgn_gn_nhwc_apply_kernel<DT, true, false>), grid, block, 0, s, a);     \
        else if (act) hipLaunchKernelGGL((pgn_gn_nhwc_apply_kernel<DT, false, true>), grid, block, 0, s, a);         \
        else hipLaunchKernelGGL((pgn_gn_nhwc_apply_kernel<DT, false, false>), grid, block, 0, s, a);                 \
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

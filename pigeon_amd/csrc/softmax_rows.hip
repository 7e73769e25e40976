// Row softmax of materialised attention scores (see include/pigeon_hip.h: pgn_softmax_rows): the middle step of the
// wide-head attention of the VAE decoder (one 512-wide head over N = H*W latent pixels; diffusers AutoencoderKL
// mid-block Attention, reached through `self.vae.decode(...)` at Pigeon/models/lavit_for_pigeon.py:774), whose head dim
// exceeds what the fused MFMA kernel holds in registers.  Scores arrive in fp32 from a bf16 GEMM (fp32 accumulate,
// fp32 out), probabilities leave in the 16-bit type the second GEMM consumes — the rounding points of
// xformers.memory_efficient_attention (fp32 softmax, 16-bit P).
// One 256-thread workgroup per row.  Rows of up to 16 Ki scores stay in registers (one read, one write); longer rows
// are walked three times (max, sum, write) and come from L2 after the first walk.  exp2 domain, fp32 statistics.
#include "pgn_common.h"

namespace {

constexpr int SM_THREADS = 256;

__device__ __forceinline__ float sm_wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float sm_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

template <int DT>
__global__ __launch_bounds__(SM_THREADS) void pgn_softmax_rows_kernel(const float* __restrict__ s, long long ld_s,
                                                                      unsigned short* __restrict__ p, long long ld_p,
                                                                      long long N, float c) {
    __shared__ float red[SM_THREADS / 64];
    const float* row = s + (long long)blockIdx.x * ld_s;
    unsigned short* out = p + (long long)blockIdx.x * ld_p;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const long long n4 = N >> 2;
    // ---- max
    float m = -INFINITY;
    for (long long i = t; i < n4; i += SM_THREADS) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(row + 4 * i);
        m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
    }
    for (long long i = 4 * n4 + t; i < N; i += SM_THREADS) m = fmaxf(m, row[i]);
    m = sm_wave_max(m);
    if (lane == 0) red[wid] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    const float mc = m * c;   // c = scale * log2(e) > 0: max of the scaled scores
    // ---- sum of exp
    float l = 0.f;
    for (long long i = t; i < n4; i += SM_THREADS) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(row + 4 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) l += __builtin_amdgcn_exp2f(v[j] * c - mc);
    }
    for (long long i = 4 * n4 + t; i < N; i += SM_THREADS) l += __builtin_amdgcn_exp2f(row[i] * c - mc);
    l = sm_wave_sum(l);
    if (lane == 0) red[wid] = l;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    // ---- probabilities
    for (long long i = t; i < n4; i += SM_THREADS) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(row + 4 * i);
        u16x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = Cvt<DT>::from_f32(__builtin_amdgcn_exp2f(v[j] * c - mc) * inv);
        *reinterpret_cast<u16x4_t*>(out + 4 * i) = o;
    }
    for (long long i = 4 * n4 + t; i < N; i += SM_THREADS)
        out[i] = Cvt<DT>::from_f32(__builtin_amdgcn_exp2f(row[i] * c - mc) * inv);
}

// Rows of up to 16 Ki scores (the VAE at 1024^2: N = 16384) are held in registers between the three steps: one read.
template <int DT>
__global__ __launch_bounds__(SM_THREADS) void pgn_softmax_rows_resident_kernel(const float* __restrict__ s, long long ld_s,
                                                                               unsigned short* __restrict__ p,
                                                                               long long ld_p, int n4, float c) {
    constexpr int R = 16;   // float4 per lane: 256 threads x 16 x 4 = 16384 scores
    __shared__ float red[SM_THREADS / 64];
    const float* row = s + (long long)blockIdx.x * ld_s;
    unsigned short* out = p + (long long)blockIdx.x * ld_p;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    f32x4_t v[R];
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = t + r * SM_THREADS;
        if (i < n4) {
            v[r] = *reinterpret_cast<const f32x4_t*>(row + 4 * (long long)i);
            m = fmaxf(fmaxf(m, fmaxf(v[r][0], v[r][1])), fmaxf(v[r][2], v[r][3]));
        }
    }
    m = sm_wave_max(m);
    if (lane == 0) red[wid] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    const float mc = m * c;
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (t + r * SM_THREADS < n4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[r][j] = __builtin_amdgcn_exp2f(v[r][j] * c - mc);
                l += v[r][j];
            }
        }
    }
    l = sm_wave_sum(l);
    if (lane == 0) red[wid] = l;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = t + r * SM_THREADS;
        if (i < n4) {
            u16x4_t o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = Cvt<DT>::from_f32(v[r][j] * inv);
            *reinterpret_cast<u16x4_t*>(out + 4 * (long long)i) = o;
        }
    }
}

}  // namespace

extern "C" int pgn_softmax_rows(const float* scores, int64_t ld_scores, void* probs, int probs_dtype, int64_t ld_probs,
                                int64_t M, int64_t N, float scale, void* stream) {
    if (scores == nullptr || probs == nullptr || M <= 0 || N <= 0) return PGN_ERR_INVALID;
    if (ld_scores < N || ld_probs < N || !(scale > 0.f)) return PGN_ERR_INVALID;
    if (probs_dtype != PGN_BF16 && probs_dtype != PGN_F16) return PGN_ERR_UNSUPPORTED;
    // 16-byte score loads, 8-byte probability stores
    if (!pgn_aligned(scores, 16) || !pgn_aligned(probs, 8) || (ld_scores & 3) != 0 || (ld_probs & 3) != 0) return PGN_ERR_INVALID;
    if (M > 0x7fffffffLL) return PGN_ERR_UNSUPPORTED;
    const float c = scale * 1.4426950408889634f;
    hipStream_t s = (hipStream_t)stream;
    if ((N & 3) == 0 && N <= 16384) {
        if (probs_dtype == PGN_BF16)
            hipLaunchKernelGGL((pgn_softmax_rows_resident_kernel<PGN_BF16>), dim3((unsigned)M), dim3(SM_THREADS), 0, s,
                               scores, (long long)ld_scores, (unsigned short*)probs, (long long)ld_probs, (int)(N >> 2), c);
        else
            hipLaunchKernelGGL((pgn_softmax_rows_resident_kernel<PGN_F16>), dim3((unsigned)M), dim3(SM_THREADS), 0, s,
                               scores, (long long)ld_scores, (unsigned short*)probs, (long long)ld_probs, (int)(N >> 2), c);
    } else if (probs_dtype == PGN_BF16)
        hipLaunchKernelGGL((pgn_softmax_rows_kernel<PGN_BF16>), dim3((unsigned)M), dim3(SM_THREADS), 0, s, scores,
                           (long long)ld_scores, (unsigned short*)probs, (long long)ld_probs, (long long)N, c);
    else
        hipLaunchKernelGGL((pgn_softmax_rows_kernel<PGN_F16>), dim3((unsigned)M), dim3(SM_THREADS), 0, s, scores,
                           (long long)ld_scores, (unsigned short*)probs, (long long)ld_probs, (long long)N, c);
    PGN_LAUNCH_CHECK();
    return PGN_OK;
}

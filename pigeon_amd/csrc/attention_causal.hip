// Causal self-attention over SHORT sequences (see include/pigeon_hip.h: pgn_attention_causal_fwd).
// The CLIP text tower of the scoring stage (Evaluation/eval_utils.py:115,126: open_clip ViT-H-14 `encode_text`, 77
// tokens, 16 heads x 64, 24 layers, additive upper-triangular -inf mask): 0.4 GFLOP per layer at batch 16, i.e. launch
// latency, not matrix throughput - so no MFMA tiling here, and the flash kernel's hot loop is not touched for it.
//
// One 256-thread workgroup per (batch, head): K and V rows of the head are staged once in LDS as 16-bit values (K at a
// pitch of D + 2 elements, so lanes that walk DIFFERENT key rows at the same column hit different banks); each
// wavefront then takes query rows i = wave, wave + 4, ...:
//   scores   lane j (and j + 64) owns key j: fp32 dot of q_i (broadcast from LDS) with K_j over D, keys j > i masked
//   softmax  wave max / sum in fp32, exp2 domain
//   output   lane d (and d + 64) owns head dim d: sum over keys j <= i of p_j * V[j][d], p broadcast from LDS
// P stays in fp32 (it never feeds a matrix instruction), the output is rounded once to the 16-bit type.
#include "pgn_common.h"

#define PGN_CAUSAL_MAX_N 128
#define PGN_CAUSAL_MAX_D 128
#define PGN_CAUSAL_WAVES 4

namespace {

template <int DT>
__global__ __launch_bounds__(PGN_CAUSAL_WAVES * 64) void pgn_attn_causal_small_kernel(
    const unsigned short* __restrict__ q, const unsigned short* __restrict__ k, const unsigned short* __restrict__ v,
    unsigned short* __restrict__ o, int N, int D, int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn,
    int64_t k_sh, int64_t v_sb, int64_t v_sn, int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh, float c) {
    extern __shared__ unsigned char smem[];
    const int kp = D + 2;                                                   // K row pitch in elements
    unsigned short* ks = reinterpret_cast<unsigned short*>(smem);           // [N][D + 2]
    unsigned short* vs = ks + (size_t)PGN_CAUSAL_MAX_N * kp;                // [N][D]
    float* qs = reinterpret_cast<float*>(vs + (size_t)PGN_CAUSAL_MAX_N * D);  // [waves][D]
    float* ps = qs + PGN_CAUSAL_WAVES * PGN_CAUSAL_MAX_D;                   // [waves][MAX_N]

    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const unsigned short* kb = k + b * k_sb + h * k_sh;
    const unsigned short* vb = v + b * v_sb + h * v_sh;
    const int half = D >> 1;                                                // 32-bit words per row
    for (int e = tid; e < N * half; e += PGN_CAUSAL_WAVES * 64) {
        const int r = e / half, w = e - r * half;
        const unsigned kw = *reinterpret_cast<const unsigned*>(kb + r * k_sn + 2 * w);
        const unsigned vw = *reinterpret_cast<const unsigned*>(vb + r * v_sn + 2 * w);
        *reinterpret_cast<unsigned*>(ks + (size_t)r * kp + 2 * w) = kw;
        *reinterpret_cast<unsigned*>(vs + (size_t)r * D + 2 * w) = vw;
    }
    __syncthreads();

    float* qw = qs + wid * PGN_CAUSAL_MAX_D;
    float* pw = ps + wid * PGN_CAUSAL_MAX_N;
    for (int i = wid; i < N; i += PGN_CAUSAL_WAVES) {                       // wave-uniform loop
        const unsigned short* qr = q + b * q_sb + i * q_sn + h * q_sh;
        for (int d = lane; d < D; d += 64) qw[d] = Cvt<DT>::to_f32(qr[d]);
        __builtin_amdgcn_wave_barrier();
        float s[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = lane + 64 * u;
            float acc = -INFINITY;
            if (j <= i) {                                                   // causal: keys up to and including i
                acc = 0.f;
                const unsigned short* kr = ks + (size_t)j * kp;
                for (int d = 0; d < D; d += 2) {
                    const unsigned w = *reinterpret_cast<const unsigned*>(kr + d);
                    acc = fmaf(qw[d], Cvt<DT>::to_f32((unsigned short)(w & 0xffffu)), acc);
                    acc = fmaf(qw[d + 1], Cvt<DT>::to_f32((unsigned short)(w >> 16)), acc);
                }
                acc *= c;                                                   // exp2 units
            }
            s[u] = acc;
        }
        float m = fmaxf(s[0], s[1]);
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        const float p0 = __builtin_amdgcn_exp2f(s[0] - m);                  // exp2(-inf) = 0 for masked keys; key 0 is
        const float p1 = __builtin_amdgcn_exp2f(s[1] - m);                  // never masked, so m is finite
        float l = p0 + p1;
        for (int off = 32; off > 0; off >>= 1) l += __shfl_xor(l, off);
        pw[lane] = p0;
        pw[lane + 64] = p1;
        __builtin_amdgcn_wave_barrier();
        const float inv = 1.0f / l;
        unsigned short* orow = o + b * o_sb + i * o_sn + h * o_sh;
        for (int d = lane; d < D; d += 64) {
            float acc = 0.f;
            for (int j = 0; j <= i; ++j) acc = fmaf(pw[j], Cvt<DT>::to_f32(vs[(size_t)j * D + d]), acc);
            orow[d] = Cvt<DT>::from_f32(acc * inv);
        }
        __builtin_amdgcn_wave_barrier();                                    // qw / pw are rewritten by the next row
    }
}

}  // namespace

extern "C" int pgn_attention_causal_fwd(const void* q, const void* k, const void* v, void* out, int dtype, int B, int H,
                                        int N, int D, int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb,
                                        int64_t k_sn, int64_t k_sh, int64_t v_sb, int64_t v_sn, int64_t v_sh,
                                        int64_t o_sb, int64_t o_sn, int64_t o_sh, float scale, void* stream) {
    if (q == nullptr || k == nullptr || v == nullptr || out == nullptr) return PGN_ERR_INVALID;
    if (B <= 0 || H <= 0 || N <= 0 || D <= 0) return PGN_ERR_INVALID;
    if (dtype != PGN_BF16 && dtype != PGN_F16) return PGN_ERR_UNSUPPORTED;
    if (N > PGN_CAUSAL_MAX_N || D > PGN_CAUSAL_MAX_D || (D & 1) != 0) return PGN_ERR_UNSUPPORTED;
    if (B > 65535) return PGN_ERR_UNSUPPORTED;
    const int64_t st[12] = {q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh, o_sb, o_sn, o_sh};
    for (int i = 0; i < 12; ++i)
        if (st[i] < 0 || (st[i] & 1) != 0) return PGN_ERR_INVALID;          // 4-byte loads of k / v row pairs
    if (!pgn_aligned(k, 4) || !pgn_aligned(v, 4)) return PGN_ERR_INVALID;
    const size_t lds = (size_t)PGN_CAUSAL_MAX_N * (D + 2) * 2 + (size_t)PGN_CAUSAL_MAX_N * D * 2 +
                       (size_t)PGN_CAUSAL_WAVES * (PGN_CAUSAL_MAX_D + PGN_CAUSAL_MAX_N) * 4;   // <= 70 KiB
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(H, B), block(PGN_CAUSAL_WAVES * 64);
    const float c = scale * 1.4426950408889634f;
#define PGN_CAUSAL_LAUNCH(DT)                                                                                        \
    do {                                                                                                             \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pgn_attn_causal_small_kernel<DT>),          \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);                   \
        if (e != hipSuccess) return (int)e;                                                                          \
        hipLaunchKernelGGL((pgn_attn_causal_small_kernel<DT>), grid, block, lds, s, (const unsigned short*)q,        \
                           (const unsigned short*)k, (const unsigned short*)v, (unsigned short*)out, N, D, q_sb,     \
                           q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh, o_sb, o_sn, o_sh, c);                      \
    } while (0)
    if (dtype == PGN_BF16) PGN_CAUSAL_LAUNCH(PGN_BF16);
    else PGN_CAUSAL_LAUNCH(PGN_F16);
#undef PGN_CAUSAL_LAUNCH
    PGN_LAUNCH_CHECK();
    return PGN_OK;
}

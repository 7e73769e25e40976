// Row-wise HBM-bound kernels of the transformer blocks inside the UNet (see include/pigeon_hip.h:
// pgn_geglu, pgn_add_layernorm).  Both replace chains of torch elementwise / norm launches that diffusers'
// BasicTransformerBlock issues between its GEMMs (reached from Pigeon/models/lavit_for_pigeon.py:752-757):
//   GEGLU           proj(x).chunk(2) -> hidden * F.gelu(gate)      : strided gelu launch + mul launch  -> one pass
//   add + LayerNorm attn_out + hidden_states ; LayerNorm(...)      : add launch + norm launch          -> one pass
// One 16-byte vector per lane per access, rows are contiguous, statistics in fp32 with wave64 shuffles only
// (one wavefront owns one row: no LDS, no barrier).
#include "pgn_common.h"

namespace {

constexpr int EW_THREADS = 256;

template <int DT>
__device__ __forceinline__ void unpack8(const u16x8_t& v, float (&f)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = Cvt<DT>::to_f32(v[j]);
}
template <int DT>
__device__ __forceinline__ u16x8_t pack8(const float (&f)[8]) {
    u16x8_t v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = Cvt<DT>::from_f32(f[j]);
    return v;
}

// ---- GEGLU -------------------------------------------------------------------------------------
// torch semantics, op by op (both ops run in fp32 "opmath" and round to the io type):
//   g = round(0.5 * gate * (1 + erf(gate / sqrt(2))))        F.gelu(gate), approximate='none'
//   y = round(hidden * g)
template <int DT>
__global__ __launch_bounds__(EW_THREADS) void pgn_geglu_kernel(const unsigned short* __restrict__ h,
                                                               unsigned short* __restrict__ y, long long M, int D8,
                                                               long long ld_h, long long ld_y) {
    const long long total = M * D8;
    const long long step = (long long)gridDim.x * EW_THREADS;
    for (long long i = (long long)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += step) {
        const long long m = i / D8;
        const int c = (int)(i - m * D8) * 8;
        const unsigned short* row = h + m * ld_h;
        const u16x8_t xv = *reinterpret_cast<const u16x8_t*>(row + c);
        const u16x8_t gv = *reinterpret_cast<const u16x8_t*>(row + (long long)D8 * 8 + c);
        float x[8], g[8];
        unpack8<DT>(xv, x);
        unpack8<DT>(gv, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float ge = 0.5f * g[j] * (1.0f + erff(g[j] * 0.70710678118654752440f));
            const float gr = Cvt<DT>::to_f32(Cvt<DT>::from_f32(ge));
            x[j] = x[j] * gr;
        }
        *reinterpret_cast<u16x8_t*>(y + m * ld_y + c) = pack8<DT>(x);
    }
}

// ---- (residual add +) LayerNorm ------------------------------------------------------------------
__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ float ld_param(const void* p, int adt, int c) {
    if (adt == PGN_F32) return reinterpret_cast<const float*>(p)[c];
    const unsigned short hh = reinterpret_cast<const unsigned short*>(p)[c];
    return adt == PGN_BF16 ? Cvt<PGN_BF16>::to_f32(hh) : Cvt<PGN_F16>::to_f32(hh);
}

// ---- convolution bias (+ residual) add -----------------------------------------------------------
// y = round(round(x + bias[c]) + res): the two roundings torch makes when a convolution's bias is a separate
// broadcast add (MIOpen convolutions carry no bias epilogue) followed by ResnetBlock2D's `input_tensor + hidden`.
// Memory is [rows][C][inner]: inner = 1 for channels-last activations, inner = H*W for NCHW.
__device__ __forceinline__ void ld_param8(const void* p, int adt, int c, float (&b)[8]) {
    // eight consecutive parameters starting at a multiple of 8: one or two 16-byte loads
    if (adt == PGN_F32) {
        const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(p) + c);
        const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(p) + c + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { b[j] = lo[j]; b[4 + j] = hi[j]; }
    } else {
        const u16x8_t v = *reinterpret_cast<const u16x8_t*>(reinterpret_cast<const unsigned short*>(p) + c);
        if (adt == PGN_BF16) unpack8<PGN_BF16>(v, b);
        else unpack8<PGN_F16>(v, b);
    }
}

template <int DT, bool HAS_RES>
__device__ __forceinline__ void bias_res_8(const unsigned short* __restrict__ x, const unsigned short* __restrict__ res,
                                           unsigned short* __restrict__ y, long long i8, const float (&b)[8]) {
    float f[8];
    unpack8<DT>(*reinterpret_cast<const u16x8_t*>(x + i8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = Cvt<DT>::to_f32(Cvt<DT>::from_f32(f[j] + b[j]));
    if constexpr (HAS_RES) {
        float r[8];
        unpack8<DT>(*reinterpret_cast<const u16x8_t*>(res + i8), r);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] += r[j];
    }
    *reinterpret_cast<u16x8_t*>(y + i8) = pack8<DT>(f);
}

// channels-last memory [rows][C]: a lane's 8 elements are 8 consecutive channels; the channel index advances by a
// constant per grid-stride step (no division in the loop)
template <int DT, bool HAS_RES>
__global__ __launch_bounds__(EW_THREADS) void pgn_bias_res_cl_kernel(const unsigned short* __restrict__ x,
                                                                     const unsigned short* __restrict__ res,
                                                                     const void* __restrict__ bias, int bdt,
                                                                     unsigned short* __restrict__ y, long long total8,
                                                                     int C) {
    const long long step = (long long)gridDim.x * EW_THREADS;
    long long i = (long long)blockIdx.x * EW_THREADS + threadIdx.x;
    int c = (int)((8 * i) % C);
    const int cstep = (int)((8 * step) % C);
    for (; i < total8; i += step) {
        float b[8];
        ld_param8(bias, bdt, c, b);
        bias_res_8<DT, HAS_RES>(x, res, y, 8 * i, b);
        c += cstep;
        if (c >= C) c -= C;
    }
}

// NCHW memory [rows*C planes][inner]: blockIdx.y walks the planes, one bias value per plane
template <int DT, bool HAS_RES>
__global__ __launch_bounds__(EW_THREADS) void pgn_bias_res_nchw_kernel(const unsigned short* __restrict__ x,
                                                                       const unsigned short* __restrict__ res,
                                                                       const void* __restrict__ bias, int bdt,
                                                                       unsigned short* __restrict__ y, long long planes,
                                                                       int C, long long inner8) {
    for (long long pl = blockIdx.y; pl < planes; pl += gridDim.y) {
        const float bb = ld_param(bias, bdt, (int)(pl % C));
        const float b[8] = {bb, bb, bb, bb, bb, bb, bb, bb};
        for (long long j = (long long)blockIdx.x * EW_THREADS + threadIdx.x; j < inner8; j += (long long)gridDim.x * EW_THREADS)
            bias_res_8<DT, HAS_RES>(x, res, y, 8 * (pl * inner8 + j), b);
    }
}

// One wavefront per NR consecutive rows, R vectors of 8 elements per lane and row held in registers (C <= 512 * R).
// NR = 2 doubles the bytes a wavefront has in flight: narrow rows (C = 640: 80 of 128 lane slots) are latency-bound
// with one row per wavefront.  Rows are independent: the arithmetic per row does not depend on NR.
template <int DT, int R, bool HAS_RES, int NR>
__global__ __launch_bounds__(EW_THREADS) void pgn_add_layernorm_kernel(
    const unsigned short* __restrict__ x, const unsigned short* __restrict__ res, unsigned short* __restrict__ sum_out,
    unsigned short* __restrict__ y, const void* __restrict__ gamma, const void* __restrict__ beta, int adt, long long M,
    int C, float eps, const float* __restrict__ cb) {
    const int lane = threadIdx.x & 63;
    const long long row0 = ((long long)blockIdx.x * (EW_THREADS / 64) + (threadIdx.x >> 6)) * NR;
    const int C8 = C >> 3;
    // The pending column bias is the same C floats for every row: read per row and lane from global memory it tripled
    // the kernel's load requests, all of them to the same twenty-odd cache lines (C = 640: 58.9 -> 75.7 us).  The
    // workgroup stages it once in LDS instead, low and high halves of a lane's 8 values in separate arrays so that the
    // 16-byte LDS reads of consecutive lanes are consecutive.
    // gamma / beta stored like the activations (the usual case) are staged the same way, behind them in the array.
    extern __shared__ __attribute__((aligned(16))) float ln_cb[];   // [C/2] low halves, [C/2] high halves (or empty)
    const bool packed_affine = gamma != nullptr && adt == DT;
    u16x8_t* ln_gb = reinterpret_cast<u16x8_t*>(ln_cb + (cb != nullptr ? C : 0));   // [C8] gamma, [C8] beta
    if (cb != nullptr || packed_affine) {                           // workgroup-uniform
        if (cb != nullptr) {
            for (int i = threadIdx.x; i < 2 * C8; i += EW_THREADS)
                *reinterpret_cast<f32x4_t*>(ln_cb + ((i & 1) ? 4 * C8 : 0) + 4 * (i >> 1)) =
                    *reinterpret_cast<const f32x4_t*>(cb + 4 * i);
        }
        if (packed_affine) {
            for (int i = threadIdx.x; i < C8; i += EW_THREADS) {
                ln_gb[i] = *reinterpret_cast<const u16x8_t*>(reinterpret_cast<const unsigned short*>(gamma) + 8 * i);
                ln_gb[C8 + i] = *reinterpret_cast<const u16x8_t*>(reinterpret_cast<const unsigned short*>(beta) + 8 * i);
            }
        }
        __syncthreads();
    }
    if (row0 >= M) return;   // whole wavefront leaves together, behind the only barrier
    u16x8_t raw[NR][R], rraw[NR][R];
#pragma unroll
    for (int n = 0; n < NR; ++n) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int v = lane + 64 * r;
            if (v < C8 && row0 + n < M) {
                raw[n][r] = *reinterpret_cast<const u16x8_t*>(x + (row0 + n) * C + 8 * v);
                if constexpr (HAS_RES) rraw[n][r] = *reinterpret_cast<const u16x8_t*>(res + (row0 + n) * C + 8 * v);
            }
        }
    }
    float f[NR][R][8];
    float s[NR];
#pragma unroll
    for (int n = 0; n < NR; ++n) {
        s[n] = 0.f;
        const long long row = row0 + n;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int v = lane + 64 * r;
            if (v < C8 && row < M) {
                unpack8<DT>(raw[n][r], f[n][r]);
                if (cb != nullptr) {   // pending per-channel bias of bias-free GEMMs accumulated into x (uniform branch)
                    const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(ln_cb + 4 * v);
                    const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(ln_cb + 4 * C8 + 4 * v);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { f[n][r][j] += lo[j]; f[n][r][4 + j] += hi[j]; }
                }
                if constexpr (HAS_RES) {
                    float g[8];
                    unpack8<DT>(rraw[n][r], g);
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[n][r][j] += g[j];   // one fp32 add, rounded below: torch's bf16 add
                    const u16x8_t hv = pack8<DT>(f[n][r]);
                    *reinterpret_cast<u16x8_t*>(sum_out + row * C + 8 * v) = hv;
                    unpack8<DT>(hv, f[n][r]);                          // the norm sees the rounded sum, as in the reference
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) s[n] += f[n][r][j];
            }
        }
    }
    const float inv_c = 1.0f / (float)C;
    float mean[NR], rstd[NR];
#pragma unroll
    for (int n = 0; n < NR; ++n) mean[n] = wave_sum64(s[n]) * inv_c;
#pragma unroll
    for (int n = 0; n < NR; ++n) {
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (lane + 64 * r < C8 && row0 + n < M) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = f[n][r][j] - mean[n];
                    q += d * d;
                }
            }
        }
        rstd[n] = 1.0f / sqrtf(wave_sum64(q) * inv_c + eps);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int v = lane + 64 * r;
        if (v < C8) {
            float ga[8], be[8];
            if (gamma == nullptr) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { ga[j] = 1.0f; be[j] = 0.0f; }
            } else if (packed_affine) {   // the usual case: parameters stored like the activations, one 16-byte LDS read each
                unpack8<DT>(ln_gb[v], ga);
                unpack8<DT>(ln_gb[C8 + v], be);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    ga[j] = ld_param(gamma, adt, 8 * v + j);
                    be[j] = ld_param(beta, adt, 8 * v + j);
                }
            }
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                if (row0 + n < M) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[n][r][j] = (f[n][r][j] - mean[n]) * rstd[n] * ga[j] + be[j];
                    *reinterpret_cast<u16x8_t*>(y + (row0 + n) * C + 8 * v) = pack8<DT>(f[n][r]);
                }
            }
        }
    }
}


// ---- VAE output -> uint8 NHWC ---------------------------------------------------------------------
// lavit_for_pigeon.py:780-782 + numpy_to_pil :929: image.float() / 2 + 0.5, clamp(0, 1), NCHW -> NHWC,
// (x * 255).round() (half to even), uint8.  Every fp32 operation is rounded separately (this file is built with
// -ffp-contract=off), so the bytes equal the reference's.
template <int DT, int PX>
__global__ __launch_bounds__(EW_THREADS) void pgn_image_u8_kernel(const void* __restrict__ x, unsigned char* __restrict__ y,
                                                                  int C, long long HW, long long total_px) {
    typedef typename Cvt<DT>::storage XS;
    const XS* xp = reinterpret_cast<const XS*>(x);
    const long long step = (long long)gridDim.x * EW_THREADS * PX;
    for (long long p0 = ((long long)blockIdx.x * EW_THREADS + threadIdx.x) * PX; p0 < total_px; p0 += step) {
        const long long n = p0 / HW, hw = p0 - n * HW;     // PX consecutive pixels never straddle images (HW % PX == 0)
        unsigned char out[PX * 4];
        for (int c = 0; c < C; ++c) {
            const XS* src = xp + (n * C + c) * HW + hw;
#pragma unroll
            for (int i = 0; i < PX; ++i) {
                float v = Cvt<DT>::to_f32(src[i]);
                v = v / 2.0f + 0.5f;
                v = fminf(fmaxf(v, 0.0f), 1.0f);
                if (v != v) v = 0.0f;   // NaN: numpy's cast of NaN to uint8 is undefined; write 0
                out[i * C + c] = (unsigned char)__builtin_rintf(v * 255.0f);
            }
        }
        unsigned char* dst = y + p0 * C;
        for (int i = 0; i < PX * C; ++i) dst[i] = out[i];
    }
}

// ---- nearest-neighbour 2x upsampling, channels-last ---------------------------------------------------------------
// y[n, 2h+dy, 2w+dx, :] = x[n, h, w, :].  One workgroup per (input row, segment): a lane reads 16 bytes once and
// stores them four times; the (pixel, chunk) pair is advanced without a division.  Pure data movement: any 16-bit type.
__global__ __launch_bounds__(EW_THREADS) void pgn_upsample2x_cl_kernel(const unsigned short* __restrict__ x,
                                                                        unsigned short* __restrict__ y, int H, int W,
                                                                        int C8, int per_seg) {
    const long long row = blockIdx.x;                 // n * H + h
    const int n = (int)(row / H), h = (int)(row - (long long)n * H);
    const int chunks = W * C8;
    const int i0 = blockIdx.y * per_seg, i1 = min(chunks, i0 + per_seg);
    const size_t C = (size_t)C8 * 8;
    const unsigned short* xr = x + (size_t)row * W * C;
    unsigned short* y0 = y + ((size_t)n * 2 * H + 2 * h) * (2 * (size_t)W) * C;
    unsigned short* y1 = y0 + 2 * (size_t)W * C;
    int i = i0 + (int)threadIdx.x;
    int w = i / C8, c = i - w * C8;
    const int dw = EW_THREADS / C8, dc = EW_THREADS - dw * C8;
    for (; i < i1; i += EW_THREADS) {
        const u16x8_t v = *reinterpret_cast<const u16x8_t*>(xr + (size_t)i * 8);
        const size_t o = (size_t)w * 2 * C + (size_t)c * 8;
        *reinterpret_cast<u16x8_t*>(y0 + o) = v;
        *reinterpret_cast<u16x8_t*>(y0 + o + C) = v;
        *reinterpret_cast<u16x8_t*>(y1 + o) = v;
        *reinterpret_cast<u16x8_t*>(y1 + o + C) = v;
        w += dw;
        c += dc;
        if (c >= C8) { c -= C8; ++w; }
    }
}

}  // namespace

extern "C" int pgn_geglu(const void* h, void* y, int dtype, int64_t M, int D, int64_t ld_h, int64_t ld_y,
                         void* stream) {
    if (h == nullptr || y == nullptr) return PGN_ERR_INVALID;
    if (M <= 0 || D <= 0 || ld_h < 2LL * D || ld_y < D) return PGN_ERR_INVALID;
    if (dtype != PGN_BF16 && dtype != PGN_F16) return PGN_ERR_UNSUPPORTED;
    if ((D & 7) != 0 || (ld_h & 7) != 0 || (ld_y & 7) != 0 || !pgn_aligned(h, 16) || !pgn_aligned(y, 16))
        return PGN_ERR_UNSUPPORTED;
    const long long total = (long long)M * (D / 8);
    long long blocks = (total + EW_THREADS - 1) / EW_THREADS;
    if (blocks > 256LL * 64) blocks = 256LL * 64;   // grid-stride beyond 64 workgroups per CU
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)blocks), block(EW_THREADS);
    if (dtype == PGN_BF16)
        hipLaunchKernelGGL((pgn_geglu_kernel<PGN_BF16>), grid, block, 0, s, (const unsigned short*)h, (unsigned short*)y,
                           (long long)M, D / 8, (long long)ld_h, (long long)ld_y);
    else
        hipLaunchKernelGGL((pgn_geglu_kernel<PGN_F16>), grid, block, 0, s, (const unsigned short*)h, (unsigned short*)y,
                           (long long)M, D / 8, (long long)ld_h, (long long)ld_y);
    PGN_LAUNCH_CHECK();
    return PGN_OK;
}

#ifdef PGN_TUNING_HOOKS
int g_ln_rows2_max_r = 2;
extern "C" void pgn_layernorm_tuning(int rows2_max_r) { g_ln_rows2_max_r = rows2_max_r; }   // measurement build only
#else
constexpr int g_ln_rows2_max_r = 2;
#endif
// measured: C = 640 95 -> 78 us, C = 320 216 -> 162 us at 131 k / 524 k rows; C = 1280 (3 vectors per lane) loses 7 %

extern "C" int pgn_add_layernorm(const void* x, const void* residual, void* sum_out, void* y, const void* gamma,
                                 const void* beta, int dtype, int affine_dtype, int64_t M, int C, float eps,
                                 void* stream) {
    return pgn_add_layernorm_cb(x, residual, sum_out, y, gamma, beta, dtype, affine_dtype, nullptr, M, C, eps, stream);
}

extern "C" int pgn_add_layernorm_cb(const void* x, const void* residual, void* sum_out, void* y, const void* gamma,
                                    const void* beta, int dtype, int affine_dtype, const float* col_bias, int64_t M,
                                    int C, float eps, void* stream) {
    if (x == nullptr || y == nullptr) return PGN_ERR_INVALID;
    if (col_bias != nullptr && !pgn_aligned(col_bias, 16)) return PGN_ERR_INVALID;
    if ((residual == nullptr) != (sum_out == nullptr)) return PGN_ERR_INVALID;
    if ((gamma == nullptr) != (beta == nullptr)) return PGN_ERR_INVALID;
    if (M <= 0 || C <= 0 || !(eps >= 0.f)) return PGN_ERR_INVALID;
    if (gamma != nullptr && pgn_dtype_size(affine_dtype) == 0) return PGN_ERR_INVALID;
    if (dtype != PGN_BF16 && dtype != PGN_F16) return PGN_ERR_UNSUPPORTED;
    if ((C & 7) != 0 || C > 4096) return PGN_ERR_UNSUPPORTED;
    if (!pgn_aligned(x, 16) || !pgn_aligned(y, 16) || !pgn_aligned(residual, 16) || !pgn_aligned(sum_out, 16))
        return PGN_ERR_UNSUPPORTED;
    const int r = (C / 8 + 63) / 64;   // vectors per lane: 1..8
    // two rows per wavefront where rows are narrow and plentiful (g_ln_rows2_max_r: tuning hook, not part of the ABI)
    const int nr = (r <= g_ln_rows2_max_r && M >= 16384) ? 2 : 1;
    const long long rows_per_block = (long long)(EW_THREADS / 64) * nr;
    const long long blocks = (M + rows_per_block - 1) / rows_per_block;
    if (blocks > 0x7fffffffLL) return PGN_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)blocks), block(EW_THREADS);
    const size_t lds = (col_bias != nullptr ? (size_t)C * sizeof(float) : 0) +
                       (gamma != nullptr && affine_dtype == dtype ? (size_t)C * 4 : 0);
#define PGN_LN_LAUNCH(DT, R, RES, NR)                                                                         \
    hipLaunchKernelGGL((pgn_add_layernorm_kernel<DT, R, RES, NR>), grid, block, lds, s, (const unsigned short*)x, \
                       (const unsigned short*)residual, (unsigned short*)sum_out, (unsigned short*)y, gamma,  \
                       beta, affine_dtype, (long long)M, C, eps, col_bias)
#define PGN_LN_R(DT, RES)                       \
    do {                                        \
        if (r <= 1) { if (nr == 2) PGN_LN_LAUNCH(DT, 1, RES, 2); else PGN_LN_LAUNCH(DT, 1, RES, 1); }      \
        else if (r <= 2) { if (nr == 2) PGN_LN_LAUNCH(DT, 2, RES, 2); else PGN_LN_LAUNCH(DT, 2, RES, 1); } \
        else if (r <= 3) { if (nr == 2) PGN_LN_LAUNCH(DT, 3, RES, 2); else PGN_LN_LAUNCH(DT, 3, RES, 1); } \
        else if (r <= 4) PGN_LN_LAUNCH(DT, 4, RES, 1); \
        else if (r <= 6) PGN_LN_LAUNCH(DT, 6, RES, 1); \
        else PGN_LN_LAUNCH(DT, 8, RES, 1);         \
    } while (0)
    if (dtype == PGN_BF16) {
        if (residual) PGN_LN_R(PGN_BF16, true);
        else PGN_LN_R(PGN_BF16, false);
    } else {
        if (residual) PGN_LN_R(PGN_F16, true);
        else PGN_LN_R(PGN_F16, false);
    }
#undef PGN_LN_R
#undef PGN_LN_LAUNCH
    PGN_LAUNCH_CHECK();
    return PGN_OK;
}

extern "C" int pgn_image_to_uint8(const void* x, int x_dtype, void* y, int N, int C, int H, int W, void* stream) {
    if (x == nullptr || y == nullptr) return PGN_ERR_INVALID;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || pgn_dtype_size(x_dtype) == 0) return PGN_ERR_INVALID;
    if (C > 4) return PGN_ERR_UNSUPPORTED;
    const long long HW = (long long)H * W, total = HW * N;
    hipStream_t s = (hipStream_t)stream;
    const bool quad = (HW % 4) == 0;
    const long long work = quad ? total / 4 : total;
    long long blocks = (work + EW_THREADS - 1) / EW_THREADS;
    if (blocks > 256LL * 32) blocks = 256LL * 32;
    const dim3 grid((unsigned)blocks), block(EW_THREADS);
#define PGN_U8(DT)                                                                                          \
    do {                                                                                                    \
        if (quad) hipLaunchKernelGGL((pgn_image_u8_kernel<DT, 4>), grid, block, 0, s, x, (unsigned char*)y, C, HW, total); \
        else hipLaunchKernelGGL((pgn_image_u8_kernel<DT, 1>), grid, block, 0, s, x, (unsigned char*)y, C, HW, total);      \
    } while (0)
    if (x_dtype == PGN_BF16) PGN_U8(PGN_BF16);
    else if (x_dtype == PGN_F16) PGN_U8(PGN_F16);
    else PGN_U8(PGN_F32);
#undef PGN_U8
    PGN_LAUNCH_CHECK();
    return PGN_OK;
}

extern "C" int pgn_bias_residual_add(const void* x, const void* residual, const void* bias, int bias_dtype, void* y,
                                     int dtype, int64_t rows, int C, int64_t inner, void* stream) {
    if (x == nullptr || bias == nullptr || y == nullptr) return PGN_ERR_INVALID;
    if (rows <= 0 || C <= 0 || inner <= 0 || pgn_dtype_size(bias_dtype) == 0) return PGN_ERR_INVALID;
    if (dtype != PGN_BF16 && dtype != PGN_F16) return PGN_ERR_UNSUPPORTED;
    // eight consecutive elements per lane: all inside one channel row (inner == 1) or one channel (inner % 8 == 0)
    if (inner == 1 ? (C % 8) != 0 : (inner % 8) != 0) return PGN_ERR_UNSUPPORTED;
    if (!pgn_aligned(x, 16) || !pgn_aligned(y, 16) || (residual != nullptr && !pgn_aligned(residual, 16))) return PGN_ERR_INVALID;
    if (inner == 1 && !pgn_aligned(bias, 16)) return PGN_ERR_INVALID;   // 16-byte bias vector loads
    hipStream_t s = (hipStream_t)stream;
    const dim3 block(EW_THREADS);
    dim3 grid;
    const long long total8 = (long long)rows * C * inner / 8, planes = (long long)rows * C, inner8 = inner / 8;
    if (inner == 1) {
        long long blocks = (total8 + EW_THREADS - 1) / EW_THREADS;
        grid = dim3((unsigned)(blocks > 256LL * 32 ? 256LL * 32 : blocks));
    } else {
        long long bx = (inner8 + EW_THREADS - 1) / EW_THREADS;
        if (bx > 64) bx = 64;
        grid = dim3((unsigned)bx, (unsigned)(planes > 65535 ? 65535 : planes));
    }
#define PGN_BR_D(DT)                                                                                                  \
    do {                                                                                                              \
        const unsigned short* xs = (const unsigned short*)x;                                                          \
        const unsigned short* rs = (const unsigned short*)residual;                                                   \
        unsigned short* ys = (unsigned short*)y;                                                                      \
        if (inner == 1) {                                                                                             \
            if (residual) hipLaunchKernelGGL((pgn_bias_res_cl_kernel<DT, true>), grid, block, 0, s, xs, rs, bias, bias_dtype, ys, total8, C);   \
            else hipLaunchKernelGGL((pgn_bias_res_cl_kernel<DT, false>), grid, block, 0, s, xs, rs, bias, bias_dtype, ys, total8, C);           \
        } else {                                                                                                      \
            if (residual) hipLaunchKernelGGL((pgn_bias_res_nchw_kernel<DT, true>), grid, block, 0, s, xs, rs, bias, bias_dtype, ys, planes, C, inner8); \
            else hipLaunchKernelGGL((pgn_bias_res_nchw_kernel<DT, false>), grid, block, 0, s, xs, rs, bias, bias_dtype, ys, planes, C, inner8);        \
        }                                                                                                             \
    } while (0)
    if (dtype == PGN_BF16) PGN_BR_D(PGN_BF16);
    else PGN_BR_D(PGN_F16);
#undef PGN_BR_D
    PGN_LAUNCH_CHECK();
    return PGN_OK;
}

extern "C" int pgn_upsample_nearest2x_nhwc(const void* x, void* y, int dtype, int N, int H, int W, int C, void* stream) {
    if (x == nullptr || y == nullptr || x == y) return PGN_ERR_INVALID;
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return PGN_ERR_INVALID;
    if (dtype != PGN_BF16 && dtype != PGN_F16) return PGN_ERR_UNSUPPORTED;
    if ((C % 8) != 0) return PGN_ERR_UNSUPPORTED;
    if (!pgn_aligned(x, 16) || !pgn_aligned(y, 16)) return PGN_ERR_INVALID;
    const long long rows = (long long)N * H, chunks = (long long)W * (C / 8);
    if (rows > 0x7fffffffLL || chunks > 0x7fffffffLL) return PGN_ERR_UNSUPPORTED;
    // ~2048 workgroups per launch, each at least one pass of the block over its segment
    long long segs = (2048 + rows - 1) / rows;
    const long long max_segs = (chunks + EW_THREADS - 1) / EW_THREADS;
    if (segs > max_segs) segs = max_segs;
    if (segs > 65535) segs = 65535;
    if (segs < 1) segs = 1;
    const int per_seg = (int)((chunks + segs - 1) / segs);
    hipLaunchKernelGGL(pgn_upsample2x_cl_kernel, dim3((unsigned)rows, (unsigned)segs), dim3(EW_THREADS), 0,
                       (hipStream_t)stream, (const unsigned short*)x, (unsigned short*)y, H, W, C / 8, per_seg);
    PGN_LAUNCH_CHECK();
    return PGN_OK;
}

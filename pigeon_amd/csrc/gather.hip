// Fused token -> codebook-embedding gather (see include/pigeon_hip.h: pgn_token_gather).
// Follows Pigeon/models/lavit_for_pigeon.py:903-918 and Pigeon/models/modeling_decoder.py:310-317.
//
// HBM-bound but tiny (<= 32 KiB per sample), so launch latency dominates: two launches total for the
// whole batch instead of the reference's per-sample Python loop with host syncs.
//   1. count:  one wave per sample, ballot/popcount of (id - id_offset >= 0), n = min(count, max_tokens)
//   2. gather: one workgroup per sample; prefix of token_nums gives the packed row base, the surviving
//      ids are compacted in order into LDS, then rows move as float4 loads -> 8-byte bf16x4 stores.
// pgn_token_embed carries the same gather on into the token decoder's first layer (modeling_decoder.py:310-319:
// in_proj over the gathered rows, padding rows zeroed, + pos_embed) and emits the additive key bias of :100-101:
//   2'. embed: one workgroup per (sample, 16 token rows): the same ordered compaction, the 16 codebook rows rounded to
//       the 16-bit type in LDS, every thread owns output columns e = tid, tid + 256, ... with that column's in_proj
//       row in registers: fp32 dot over `dim`, + bias, rounded (the GEMM's output), x mask, + pos_embed, rounded.
#include "pgn_common.h"

#define PGN_GATHER_MAX_TOKENS 1024
#define PGN_GATHER_THREADS 256

__global__ __launch_bounds__(PGN_WAVE) void pgn_gather_count_kernel(const int64_t* __restrict__ ids,
                                                                  const int32_t* __restrict__ offsets,
                                                                  int64_t id_offset, int max_tokens,
                                                                  int num_codes, int64_t* __restrict__ token_nums,
                                                                  int32_t* __restrict__ status) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const int o0 = offsets[b], o1 = offsets[b + 1];
    int kept = 0;
    bool bad = false;
    for (int base = o0; base < o1 && kept < max_tokens; base += PGN_WAVE) {
        const int i = base + lane;
        int64_t code = -1;
        if (i < o1) code = ids[i] - id_offset;
        const bool valid = code >= 0;
        const unsigned long long m = __ballot(valid);
        const int rank = kept + __popcll(m & ((1ull << lane) - 1ull));
        if (valid && rank < max_tokens && code >= (int64_t)num_codes) bad = true;
        kept += __popcll(m);
    }
    if (kept > max_tokens) kept = max_tokens;
    if (lane == 0) token_nums[b] = (int64_t)kept;
    if (status != nullptr && __any(bad) && lane == 0) atomicExch(status, 1);
}

template <int DT>
__device__ __forceinline__ void store4(void* dst, size_t elem_index, float4 v) {
    typedef typename Cvt<DT>::storage S;
    if constexpr (DT == PGN_F32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(dst) + elem_index) = v;
    } else {
        u16x4_t o;
        o[0] = Cvt<DT>::from_f32(v.x);
        o[1] = Cvt<DT>::from_f32(v.y);
        o[2] = Cvt<DT>::from_f32(v.z);
        o[3] = Cvt<DT>::from_f32(v.w);
        *reinterpret_cast<u16x4_t*>(reinterpret_cast<S*>(dst) + elem_index) = o;
    }
}

template <int DT>
__global__ __launch_bounds__(PGN_GATHER_THREADS) void pgn_gather_rows_kernel(
    const int64_t* __restrict__ ids, const int32_t* __restrict__ offsets, const float* __restrict__ codebook,
    int num_codes, int dim, int64_t id_offset, int max_tokens, const int64_t* __restrict__ token_nums,
    void* __restrict__ out_packed, void* __restrict__ out_padded, void* __restrict__ out_mask, int pad_len) {
    __shared__ int src[PGN_GATHER_MAX_TOKENS];
    __shared__ long long red[PGN_GATHER_THREADS / PGN_WAVE];
    __shared__ int wave_cnt[PGN_GATHER_THREADS / PGN_WAVE];

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    constexpr int NW = PGN_GATHER_THREADS / PGN_WAVE;

    // packed row base = sum of token_nums[0..b)
    long long part = 0;
    for (int i = tid; i < b; i += PGN_GATHER_THREADS) part += token_nums[i];
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0) red[wid] = part;
    __syncthreads();
    long long row_base = 0;
    for (int w = 0; w < NW; ++w) row_base += red[w];

    const int n = (int)token_nums[b];
    const int o0 = offsets[b], o1 = offsets[b + 1];

    // ordered compaction of surviving ids into LDS
    int kept = 0;
    for (int base = o0; base < o1 && kept < max_tokens; base += PGN_GATHER_THREADS) {
        const int i = base + tid;
        int64_t code = -1;
        if (i < o1) code = ids[i] - id_offset;
        const bool valid = code >= 0;
        const unsigned long long m = __ballot(valid);
        __syncthreads();  // wave_cnt reuse
        if (lane == 0) wave_cnt[wid] = __popcll(m);
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < NW; ++w) {
            const int c = wave_cnt[w];
            if (w < wid) before += c;
            total += c;
        }
        const int rank = kept + before + __popcll(m & ((1ull << lane) - 1ull));
        if (valid && rank < max_tokens) src[rank] = (code < (int64_t)num_codes) ? (int)code : -1;
        kept += total;
    }
    __syncthreads();

    const int vpr = dim >> 2;  // float4 per row
    for (int e = tid; e < n * vpr; e += PGN_GATHER_THREADS) {
        const int row = e / vpr, c = e - row * vpr;
        const int code = src[row];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (code >= 0) v = *reinterpret_cast<const float4*>(codebook + (size_t)code * dim + 4 * c);
        if (out_packed != nullptr) store4<DT>(out_packed, (size_t)(row_base + row) * dim + 4 * c, v);
        if (out_padded != nullptr) store4<DT>(out_padded, ((size_t)b * pad_len + row) * dim + 4 * c, v);
    }
    if (out_padded != nullptr) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = n * vpr + tid; e < pad_len * vpr; e += PGN_GATHER_THREADS)
            store4<DT>(out_padded, (size_t)b * pad_len * dim + (size_t)e * 4, z);
    }
    if (out_mask != nullptr) {
        typedef typename Cvt<DT>::storage S;
        S* mk = reinterpret_cast<S*>(out_mask) + (size_t)b * pad_len;
        for (int j = tid; j < pad_len; j += PGN_GATHER_THREADS) mk[j] = Cvt<DT>::from_f32(j < n ? 1.f : 0.f);
    }
}

#define PGN_EMBED_ROWS 16
#define PGN_EMBED_MAX_DIM 64

template <int DT>
__global__ __launch_bounds__(PGN_GATHER_THREADS) void pgn_token_embed_kernel(
    const int64_t* __restrict__ ids, const int32_t* __restrict__ offsets, const float* __restrict__ codebook,
    int num_codes, int dim, int64_t id_offset, int max_tokens, const int64_t* __restrict__ token_nums,
    const void* __restrict__ w_, const void* __restrict__ bias_, const void* __restrict__ pos_, int E,
    void* __restrict__ out_, float* __restrict__ key_bias, int pad_len) {
    typedef typename Cvt<DT>::storage S;
    __shared__ int src[PGN_GATHER_MAX_TOKENS];
    __shared__ int wave_cnt[PGN_GATHER_THREADS / PGN_WAVE];
    __shared__ float rows[PGN_EMBED_ROWS][PGN_EMBED_MAX_DIM];

    const int b = blockIdx.y;
    const int row0 = blockIdx.x * PGN_EMBED_ROWS;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    constexpr int NW = PGN_GATHER_THREADS / PGN_WAVE;
    const int n = (int)token_nums[b];
    const int o0 = offsets[b], o1 = offsets[b + 1];
    const S* w = reinterpret_cast<const S*>(w_);
    const S* bias = reinterpret_cast<const S*>(bias_);
    const S* pos = reinterpret_cast<const S*>(pos_);
    S* out = reinterpret_cast<S*>(out_);

    if (row0 < n) {   // block-uniform: a chunk of padding rows needs no ids
        int kept = 0;
        for (int base = o0; base < o1 && kept < max_tokens && kept < row0 + PGN_EMBED_ROWS; base += PGN_GATHER_THREADS) {
            const int i = base + tid;
            int64_t code = -1;
            if (i < o1) code = ids[i] - id_offset;
            const bool valid = code >= 0;
            const unsigned long long m = __ballot(valid);
            __syncthreads();
            if (lane == 0) wave_cnt[wid] = __popcll(m);
            __syncthreads();
            int before = 0, total = 0;
            for (int wv = 0; wv < NW; ++wv) {
                const int c = wave_cnt[wv];
                if (wv < wid) before += c;
                total += c;
            }
            const int rank = kept + before + __popcll(m & ((1ull << lane) - 1ull));
            if (valid && rank < max_tokens) src[rank] = (code < (int64_t)num_codes) ? (int)code : -1;
            kept += total;
        }
        __syncthreads();
        // the rows as the reference's GEMM sees them: codebook values rounded to the 16-bit type (:913, :310)
        // (all PGN_EMBED_MAX_DIM columns are written: the dot product below runs over the full width against zero
        // weights, and 0 x LDS garbage would be NaN whenever the garbage is)
        for (int e = tid; e < PGN_EMBED_ROWS * PGN_EMBED_MAX_DIM; e += PGN_GATHER_THREADS) {
            const int r = e / PGN_EMBED_MAX_DIM, c = e - r * PGN_EMBED_MAX_DIM;
            float v = 0.f;
            if (c < dim && row0 + r < n) {
                const int code = src[row0 + r];
                if (code >= 0) v = Cvt<DT>::to_f32(Cvt<DT>::from_f32(codebook[(size_t)code * dim + c]));
            }
            rows[r][c] = v;
        }
        __syncthreads();
    }
    if (key_bias != nullptr && tid < PGN_EMBED_ROWS && row0 + tid < pad_len)
        key_bias[(size_t)b * pad_len + row0 + tid] = (row0 + tid < n) ? 0.f : -10000.f;

    for (int e = tid; e < E; e += PGN_GATHER_THREADS) {
        float wr[PGN_EMBED_MAX_DIM];
#pragma unroll
        for (int c = 0; c < PGN_EMBED_MAX_DIM; ++c) wr[c] = c < dim ? Cvt<DT>::to_f32(w[(size_t)e * dim + c]) : 0.f;
        const float be = bias != nullptr ? Cvt<DT>::to_f32(bias[e]) : 0.f;
#pragma unroll 1
        for (int r = 0; r < PGN_EMBED_ROWS; ++r) {
            const int row = row0 + r;
            if (row >= pad_len) break;
            const float pe = Cvt<DT>::to_f32(pos[(size_t)row * E + e]);
            float y = 0.f;
            if (row < n) {
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < PGN_EMBED_MAX_DIM; ++c) acc = fmaf(rows[r][c], wr[c], acc);
                y = Cvt<DT>::to_f32(Cvt<DT>::from_f32(acc + be));      // in_proj's 16-bit output; x mask (1) is exact
            }
            out[((size_t)b * pad_len + row) * E + e] = Cvt<DT>::from_f32(y + pe);
        }
    }
}

extern "C" int pgn_token_gather(const int64_t* ids, const int32_t* offsets, int B, const float* codebook,
                                int num_codes, int dim, int64_t id_offset, int max_tokens, int out_dtype,
                                int64_t* token_nums, void* out_packed, void* out_padded, void* out_mask,
                                int pad_len, int32_t* status, void* stream) {
    if (B == 0) return PGN_OK;
    if (ids == nullptr || offsets == nullptr || codebook == nullptr || token_nums == nullptr) return PGN_ERR_INVALID;
    if (B < 0 || num_codes <= 0 || dim <= 0 || (dim & 3) != 0) return PGN_ERR_INVALID;
    if (max_tokens <= 0 || max_tokens > PGN_GATHER_MAX_TOKENS) return PGN_ERR_INVALID;
    if (pgn_dtype_size(out_dtype) == 0) return PGN_ERR_INVALID;
    if ((out_padded != nullptr || out_mask != nullptr) && pad_len < max_tokens) return PGN_ERR_INVALID;
    if (!pgn_aligned(codebook, 16) || !pgn_aligned(out_packed, 8) || !pgn_aligned(out_padded, 8)) return PGN_ERR_INVALID;
    if (out_dtype == PGN_F32 && (!pgn_aligned(out_packed, 16) || !pgn_aligned(out_padded, 16))) return PGN_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(pgn_gather_count_kernel, dim3(B), dim3(PGN_WAVE), 0, s, ids, offsets, id_offset, max_tokens,
                       num_codes, token_nums, status);
    PGN_LAUNCH_CHECK();
#define PGN_GATHER_LAUNCH(DT)                                                                                       \
    hipLaunchKernelGGL(pgn_gather_rows_kernel<DT>, dim3(B), dim3(PGN_GATHER_THREADS), 0, s, ids, offsets, codebook, \
                       num_codes, dim, id_offset, max_tokens, token_nums, out_packed, out_padded, out_mask, pad_len)
    if (out_dtype == PGN_BF16)
        PGN_GATHER_LAUNCH(PGN_BF16);
    else if (out_dtype == PGN_F16)
        PGN_GATHER_LAUNCH(PGN_F16);
    else
        PGN_GATHER_LAUNCH(PGN_F32);
#undef PGN_GATHER_LAUNCH
    PGN_LAUNCH_CHECK();
    return PGN_OK;
}

extern "C" int pgn_token_embed(const int64_t* ids, const int32_t* offsets, int B, const float* codebook, int num_codes,
                               int dim, int64_t id_offset, int max_tokens, int dtype, const void* in_proj_weight,
                               const void* in_proj_bias, const void* pos_embed, int E, int64_t* token_nums, void* out,
                               float* key_bias, int pad_len, int32_t* status, void* stream) {
    if (B == 0) return PGN_OK;
    if (ids == nullptr || offsets == nullptr || codebook == nullptr || token_nums == nullptr) return PGN_ERR_INVALID;
    if (in_proj_weight == nullptr || pos_embed == nullptr || out == nullptr) return PGN_ERR_INVALID;
    if (B < 0 || B > 65535 || num_codes <= 0 || E <= 0) return PGN_ERR_INVALID;
    if (dim <= 0 || dim > PGN_EMBED_MAX_DIM) return PGN_ERR_UNSUPPORTED;
    if (max_tokens <= 0 || max_tokens > PGN_GATHER_MAX_TOKENS || pad_len < max_tokens) return PGN_ERR_INVALID;
    if (dtype != PGN_BF16 && dtype != PGN_F16) return PGN_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(pgn_gather_count_kernel, dim3(B), dim3(PGN_WAVE), 0, s, ids, offsets, id_offset, max_tokens,
                       num_codes, token_nums, status);
    PGN_LAUNCH_CHECK();
    const dim3 grid((pad_len + PGN_EMBED_ROWS - 1) / PGN_EMBED_ROWS, B);
#define PGN_EMBED_LAUNCH(DT)                                                                                          \
    hipLaunchKernelGGL(pgn_token_embed_kernel<DT>, grid, dim3(PGN_GATHER_THREADS), 0, s, ids, offsets, codebook,      \
                       num_codes, dim, id_offset, max_tokens, token_nums, in_proj_weight, in_proj_bias, pos_embed, E,  \
                       out, key_bias, pad_len)
    if (dtype == PGN_BF16)
        PGN_EMBED_LAUNCH(PGN_BF16);
    else
        PGN_EMBED_LAUNCH(PGN_F16);
#undef PGN_EMBED_LAUNCH
    PGN_LAUNCH_CHECK();
    return PGN_OK;
}

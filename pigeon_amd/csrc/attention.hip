// Flash-style multi-head attention forward on gfx950 MFMA (see include/pigeon_hip.h: pgn_attention_fwd).
// Replaces xformers.ops.memory_efficient_attention as reached from diffusers XFormersAttnProcessor
// (enabled at Pigeon/models/lavit_for_pigeon.py:202): softmax(q k^T * scale [+ key bias]) v, head_dim 64.
//
// Two kernels ship from this file (pgn_attention_fwd picks one; measured A/B per shape: scripts/attn_ab.py):
//   pgn_attn_w64r_kernel  head dim 64, no key bias, full key tiles, >= 1024 workgroups - every head-dim-64 launch of the
//                         SDXL UNet at batch >= 4: 4 waves x 64 query rows, K / V by LDS-DMA through a ring of three
//                         buffers with counted waits; softmax in PLAIN fp32 instructions (they issue underneath a
//                         running MFMA on gfx950, packed fp32 ones hold the matrix pipe: scripts/ubench_overlap.hip),
//                         no row maximum on the common path, one query block's PV MFMAs under the other's exponentials
//                         (a ragged key count: the same kernel with a masked last tile);
//                         its two-buffer predecessor pgn_attn_w64_kernel is compiled into the measurement build only
//   pgn_attn_fwd_kernel   everything else (key bias, head dims 8..256, small launches): 4 waves x 32 rows,
//                         register-staged double buffer
// Common structure (MI355X-first, not a warp-shaped port):
//   * workgroup ids are remapped so that all query tiles of a head run on one XCD and its K/V stay in that XCD's L2;
//   * K/V stream through LDS in 64-key tiles, the next tile(s) in flight while the current one is computed, one
//     barrier per tile;
//   * S^T = K Q^T with v_mfma_f32_32x32x16_bf16 (keys on the accumulator rows, the query on the lane), so
//     every lane owns one query row: row max / sum are lane-local plus ONE permlane32 swap, and the
//     accumulator registers are already the B operand of O^T = V^T P^T (no LDS round trip for P);
//   * K tile is XOR-swizzled for conflict-free ds_read_b128; V tile is laid out in 8-key x 32-dim
//     sub-tiles and read with ds_read_b64_tr_b16 (hardware transpose) as the A operand;
//   * q/k/v/out are addressed through (batch, sequence, head) element strides: the [B, N, H*64] output
//     of to_q/to_k/to_v is consumed in place, no head_to_batch_dim copy.
#include <type_traits>

#include "pgn_common.h"

namespace {

constexpr int ATT_BN = 64;       // keys per tile
// The softmax reference of a 32-row block is raised (and O, l rescaled) only when a row maximum outgrows it by more
// than 2^8: softmax is invariant to the reference, which merely has to keep exp2 in range (P <= 256: exact in fp32,
// inside the bf16 / fp16 range; the accumulators are fp32).  With 32 rows per block SOME row sets a new maximum in most
// tiles of iid scores, so an exact running maximum pays the O-wide rescale at almost every tile; deferred, it runs a
// few times per row (+3...6 % on every SDXL shape, A/B in one process: scripts/attn_ab.py).
constexpr float ATT_DEFER_LOG2 = 8.0f;
// Head dims: the kernel is instantiated for a PADDED head dim DP of 64, 128 or 256 (256: one wavefront per SIMD, for
// SD-1.5's 160-wide heads only); a head dim D < DP (D % 8 == 0: the token decoder's 88, SD-1.5's 40 / 80 / 160) is zero-filled up to DP on the way into registers / LDS (zero q and k columns
// add nothing to the scores, zero v columns are never stored), so callers pass their tensors as they are.
// pgn_attn_fwd_kernel: a workgroup = NWAVES wave64 of 32 query rows each; NWAVES = 4 ships (128 rows, several
// workgroups per CU), the 8-wave geometry exists in the measurement build only


typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

struct AttnArgs {
    const unsigned short* q;
    const unsigned short* k;
    const unsigned short* v;
    unsigned short* o;
    const float* bias;
    int H, Nq, Nk, nqt, D;
#ifdef PGN_TUNING_HOOKS
    unsigned long long* dbg;   // measurement build: per-workgroup timestamps of the ring kernel (s_memrealtime, 100 MHz)
#endif
    long long q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh, o_sb, o_sn, o_sh, bias_sb;
    float c;  // scale * log2(e)
};

template <int DT>
__device__ __forceinline__ f32x16_t mfma32(u16x8_t a, u16x8_t b, f32x16_t c) {
    if constexpr (DT == PGN_BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                       c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c,
                                                      0, 0, 0);
}

__device__ __forceinline__ float xhalf_max(float x) {
    // max over the two 32-lane halves (lane l <-> lane l^32)
    const unsigned u = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float x) {
    const unsigned u = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// LDS byte offsets of 16-byte chunk `ch` (0..DP/8-1) of key row `row` (0..63).  K rows are XOR-swizzled so that the 16
// lanes ds_read_b128 serves together (16 different rows, same logical chunk) hit 16 different 16-byte columns of the
// 256-byte bank line: two 128-byte rows per line at DP = 64, one 256-byte row per line at DP = 128.
template <int DP>
__device__ __forceinline__ int k_lds_off(int row, int ch) {
    if constexpr (DP == 64) return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4);
    else return row * (2 * DP) + ((ch ^ (row & 15)) << 4);   // DP 128 / 256: the XOR stays inside a 256-byte line
}
// V: 8-key x 32-dim sub-tiles of 512 bytes (what ds_read_b64_tr_b16 transposes), DP/32 of them per 8-key group
template <int DP>
__device__ __forceinline__ int v_lds_off(int row, int ch) {
    return (16 * DP) * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * (ch & 3);
}

// Epilogue of one 32-row query block at head dim 64: the lane-per-row accumulators (lane = row l31, half hh: four
// consecutive dims per register group) are normalised, rounded and written to a [32 rows][144 bytes] LDS image of the
// wavefront, then stored as WHOLE 128-byte rows, 16 bytes per lane (8 rows per store instruction).  Per-lane stores at a
// row stride touch 32 cache lines with 8 bytes each and made the epilogue store-issue-bound (short key loops: the
// epilogue was 15-20 % of a workgroup's life).  `lds` is private to the wavefront; the caller has passed the barrier
// that ends the last tile, so the K / V images are dead.
constexpr int ATT_O_PITCH = 144;   // 128-byte rows + 16: 16-byte aligned for ds_read_b128, rows 8 apart share a bank pair
template <int DT>
__device__ __forceinline__ void att_store_block_d64(unsigned char* lds, const f32x16_t (&oacc)[2], float inv,
                                                    unsigned short* orow0, long long o_sn, int rows_valid, int lane) {
    const int l31 = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            u16x4_t w;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = Cvt<DT>::from_f32(oacc[d][4 * g4 + j] * inv);
            *reinterpret_cast<u16x4_t*>(lds + l31 * ATT_O_PITCH + (d * 32 + 8 * g4 + 4 * hh) * 2) = w;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the rows are read back by other lanes of this wavefront
    const int r8 = lane >> 3, ch = lane & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = i * 8 + r8;
        const u16x8_t v = *reinterpret_cast<const u16x8_t*>(lds + row * ATT_O_PITCH + ch * 16);
        if (row < rows_valid) *reinterpret_cast<u16x8_t*>(orow0 + (long long)row * o_sn + ch * 8) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // before the image is overwritten by the next block
}

template <int DT, bool HAS_BIAS, int NWAVES, int MINW, int DP, bool PAD>
__global__ __launch_bounds__(NWAVES * 64, MINW) void pgn_attn_fwd_kernel(AttnArgs a) {
    constexpr int ATT_THREADS = NWAVES * 64;
    constexpr int ATT_BM = NWAVES * 32;
    constexpr int ATT_TILE_ELEMS = ATT_BN * DP;         // 8 KiB (DP 64) / 16 KiB (DP 128) of 16-bit elements
    constexpr int NCH = DP / 8;                          // 16-byte chunks per key row
    constexpr int SROWS = ATT_THREADS / NCH;             // key rows staged per pass
    constexpr int NSTAGE = ATT_BN / SROWS;               // (row, chunk) pairs of each tile per thread
    constexpr int NKS = DP / 16;                         // MFMA k-steps of the score product
    constexpr int NOD = DP / 32;                         // 32-dim blocks of the output accumulator
    // [buf][K|V][8 KiB] + [buf][64 floats of key bias]; one object so the base stays 16-byte aligned
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * ATT_TILE_ELEMS * 2 + 2 * ATT_BN * 4];
    float* bias_lds = reinterpret_cast<float*>(smem + 2 * 2 * ATT_TILE_ELEMS * 2);

    // XCD-aware bijective remap: consecutive logical ids (same head) share blockIdx % 8, i.e. one XCD
    const unsigned nwg = gridDim.x, orig = blockIdx.x;
    const unsigned xcd = orig & 7u, qd = nwg >> 3, rm = nwg & 7u;
    const unsigned wgid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (orig >> 3);
    const int bh = (int)(wgid / (unsigned)a.nqt), qt = (int)(wgid % (unsigned)a.nqt);
    const int b = bh / a.H, h = bh % a.H;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;

    // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[row][16 s + 8 hh + j]
    const int qrow = qt * ATT_BM + wid * 32 + l31;
    const int qrow_c = qrow < a.Nq ? qrow : a.Nq - 1;
    const unsigned short* qp = a.q + b * a.q_sb + (long long)qrow_c * a.q_sn + h * a.q_sh + 8 * hh;
    const u16x8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    u16x8_t qf[NKS];
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
        if (!PAD || 16 * s + 8 * hh < a.D) qf[s] = *reinterpret_cast<const u16x8_t*>(qp + 16 * s);
        else qf[s] = zero8;
    }

    // ---- staging assignment: thread -> (key row, 16-byte chunk) of the K and V tiles
    const int srow = tid / NCH, sch = tid % NCH;   // + i * SROWS rows for the i-th staged chunk
    const bool sch_ok = !PAD || sch * 8 < a.D;     // chunks past the real head dim are staged as zeros
    // K / V addressing: a wave-uniform 64-bit base per (batch, head) in SGPRs plus 32-bit per-thread byte offsets
    // (the host refuses key slices of 4 GiB and more), so a staged row costs one v_add_u32 per tile.
    const char* kbase = reinterpret_cast<const char*>(a.k + b * a.k_sb + h * a.k_sh);
    const char* vbase = reinterpret_cast<const char*>(a.v + b * a.v_sb + h * a.v_sh);
    const unsigned k_rowb = (unsigned)a.k_sn * 2u, v_rowb = (unsigned)a.v_sn * 2u;   // bytes per key row
    int k_woff[NSTAGE], v_woff[NSTAGE];
#pragma unroll
    for (int i = 0; i < NSTAGE; ++i) {
        k_woff[i] = k_lds_off<DP>(srow + i * SROWS, sch);
        v_woff[i] = v_lds_off<DP>(srow + i * SROWS, sch);
    }
    const float* bp = HAS_BIAS ? a.bias + b * a.bias_sb : nullptr;

    const int nt = (a.Nk + ATT_BN - 1) / ATT_BN;
    const bool ragged = (a.Nk & (ATT_BN - 1)) != 0;
    const int nfull = ragged ? nt - 1 : nt;

    u16x8_t kreg[NSTAGE], vreg[NSTAGE];
    float breg = 0.f;
    auto stage_load = [&](int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NSTAGE; ++i) {
            const int key = t * ATT_BN + srow + i * SROWS;
            if (key < a.Nk && sch_ok) {
                kreg[i] = *reinterpret_cast<const u16x8_t*>(kbase + ((unsigned)key * k_rowb + (unsigned)sch * 16u));
                vreg[i] = *reinterpret_cast<const u16x8_t*>(vbase + ((unsigned)key * v_rowb + (unsigned)sch * 16u));
            } else {
                kreg[i] = zero8;
                vreg[i] = zero8;
            }
        }
        if constexpr (HAS_BIAS) {
            if (tid < ATT_BN) {
                const int kk = t * ATT_BN + tid;
                breg = kk < a.Nk ? bp[kk] : -INFINITY;
            }
        }
    };
    // Full tiles of the unpadded head dims: no bounds checks; one running byte offset per tensor, the NSTAGE staged
    // rows of a thread sit at uniform distances (SROWS rows) that fold into the scalar base.
    unsigned koff = (unsigned)(ATT_BN + srow) * k_rowb + (unsigned)sch * 16u;   // tile 1: tile 0 goes through stage_load
    unsigned voff = (unsigned)(ATT_BN + srow) * v_rowb + (unsigned)sch * 16u;
    const unsigned k_step = (unsigned)ATT_BN * k_rowb, v_step = (unsigned)ATT_BN * v_rowb;
    auto stage_load_full = [&](int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NSTAGE; ++i) {
            kreg[i] = *reinterpret_cast<const u16x8_t*>(kbase + (size_t)(i * SROWS) * k_rowb + koff);
            vreg[i] = *reinterpret_cast<const u16x8_t*>(vbase + (size_t)(i * SROWS) * v_rowb + voff);
        }
        koff += k_step;
        voff += v_step;
        if constexpr (HAS_BIAS) {
            if (tid < ATT_BN) breg = bp[t * ATT_BN + tid];
        }
    };
    auto stage_write = [&](int buf) __attribute__((always_inline)) {
        unsigned char* base = smem + buf * (2 * ATT_TILE_ELEMS * 2);
#pragma unroll
        for (int i = 0; i < NSTAGE; ++i) {
            *reinterpret_cast<u16x8_t*>(base + k_woff[i]) = kreg[i];
            *reinterpret_cast<u16x8_t*>(base + ATT_TILE_ELEMS * 2 + v_woff[i]) = vreg[i];
        }
        if constexpr (HAS_BIAS) {
            if (tid < ATT_BN) bias_lds[buf * ATT_BN + tid] = breg * 1.4426950408889634f;   // exp2 units
        }
    };

    // ---- per-lane read offsets
    // K (A operand of S^T): row = kb*32 + l31, chunk = 2 s + hh
    int k_roff[2][NKS];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s = 0; s < NKS; ++s) k_roff[kb][s] = k_lds_off<DP>(kb * 32 + l31, 2 * s + hh);
    // V (A operand of O^T = V^T P^T) through ds_read_b64_tr_b16: lane i of a 16-lane group supplies
    // row (i>>2), columns 4*(i&3).. of the group's 4-key x 16-dim block
    const int i16 = lane & 15, g16 = (lane >> 4) & 1;
    const int v_lane_off = 64 * (i16 >> 2) + 32 * g16 + 8 * (i16 & 3);

    f32x16_t oacc[NOD];
#pragma unroll
    for (int d = 0; d < NOD; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    float m_run = -1.0e30f;  // softmax reference (raw score units without bias, exp2 units with bias)
    float l_run = 0.f;       // this lane's half of the running denominator

    stage_load(0);
    stage_write(0);
    __syncthreads();

    // One key tile.  MASKED is a compile-time tag so that the -inf masking of a ragged last tile costs nothing in the
    // other iterations (the runtime form compiled to 31 v_cndmask per tile).
    // BUF (the LDS buffer this tile reads) is a compile-time tag too: the tile loop is unrolled by two so that every
    // ds_read / ds_write address is a per-lane register plus an immediate.
    auto tile = [&](int t, auto masked_tag, auto buf_tag) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        constexpr int BUFC = decltype(buf_tag)::value;   // -1: run-time parity (the wide head dims, register-bound)
        const int buf = BUFC >= 0 ? BUFC : (t & 1);
        if constexpr (!PAD) {
            if (t + 1 < nfull) stage_load_full(t + 1);
            else if (t + 1 < nt) stage_load(t + 1);
        } else {
            if (t + 1 < nt) stage_load(t + 1);
        }

        const unsigned char* kt = smem + buf * (2 * ATT_TILE_ELEMS * 2);
        const unsigned char* vt = kt + ATT_TILE_ELEMS * 2;

        // ---- S^T = K Q^T : sacc[kb][r] = S[q = l31][key = kb*32 + (r&3) + 8*(r>>2) + 4*hh]
        f32x16_t sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                const u16x8_t kf = *reinterpret_cast<const u16x8_t*>(kt + k_roff[kb][s]);
                sacc[kb] = mfma32<DT>(kf, qf[s], sacc[kb]);
            }
        }

        // ---- scores in exp2 units / masking
        if constexpr (HAS_BIAS) {
            const float* bl = bias_lds + buf * ATT_BN;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    sacc[kb][r] = fmaf(sacc[kb][r], a.c, bl[kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh]);
        } else if constexpr (MASKED) {
            const int rem = a.Nk - t * ATT_BN;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= rem) sacc[kb][r] = -INFINITY;
        }

        // ---- online softmax (one query row per lane; the two lane halves hold disjoint keys)
        float tmax = sacc[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[kb][r]);
        tmax = xhalf_max(tmax);
        const float cm = HAS_BIAS ? 1.0f : a.c;
        if (__any((tmax - m_run) * cm > ATT_DEFER_LOG2)) {
            const float m_new = fmaxf(m_run, tmax);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * cm);
            l_run *= alpha;
            const f32x2_t alpha2 = {alpha, alpha};
#pragma unroll
            for (int d = 0; d < NOD; ++d)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f32x2_t o2 = {oacc[d][r], oacc[d][r + 1]};
                    o2 *= alpha2;   // v_pk_mul_f32
                    oacc[d][r] = o2[0];
                    oacc[d][r + 1] = o2[1];
                }
            m_run = m_new;
        }
        const float mc = m_run * cm;
        // packed fp32 math (v_pk_fma_f32 / v_pk_add_f32): two scores per VALU instruction.  Packed instructions do not
        // issue underneath an MFMA (see pgn_attn_w64r_kernel); with this kernel's three waves per SIMD the plain form
        // measured the same (+-0.5 %, profiles/r02_attention_32row_plain_valu_ab.txt) and costs a few spilled registers
        // at the 168-VGPR cap, so the packed form stays here
        const f32x2_t c2 = {HAS_BIAS ? 1.0f : a.c, HAS_BIAS ? 1.0f : a.c};
        const f32x2_t nmc2 = {-mc, -mc};
        f32x2_t lsum2[2] = {{0.f, 0.f}, {0.f, 0.f}};   // two chains: no dependent back-to-back packed adds
        u16x8_t pf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2_t x = {sacc[kb][r], sacc[kb][r + 1]};
                x = x * c2 + nmc2;
                f32x2_t p = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
                lsum2[(r >> 1) & 1] += p;
                pf[kb][r >> 3][r & 7] = Cvt<DT>::from_f32(p[0]);
                pf[kb][r >> 3][(r & 7) + 1] = Cvt<DT>::from_f32(p[1]);
            }
        lsum2[0] += lsum2[1];
        l_run += lsum2[0][0] + lsum2[0][1];

        // ---- O^T += V^T P^T : A operand element j of lane half hh is key 16 s2 + 8 (j>>2) + 4 hh + (j&3)
#pragma unroll
        for (int d = 0; d < NOD; ++d) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int key0 = kb * 32 + 16 * s2 + 4 * hh;  // multiple of 4; +8 for the upper 4 elements
                    const int off0 = (16 * DP) * (key0 >> 3) + 64 * (key0 & 7) + 512 * d + v_lane_off;
                    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4_t*)(vt + off0));
                    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4_t*)(vt + off0 + 16 * DP));
                    u16x8_t vf;
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                    oacc[d] = mfma32<DT>(vf, pf[kb][s2], oacc[d]);
                }
            }
        }

        if (t + 1 < nt) stage_write(buf ^ 1);
        __syncthreads();
    };

    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    using BR = std::integral_constant<int, -1>;
    if constexpr (DP == 64) {
        int t = 0;
        for (; t + 1 < nfull; t += 2) {
            tile(t, std::false_type{}, B0{});
            tile(t + 1, std::false_type{}, B1{});
        }
        if (t < nfull) {
            tile(t, std::false_type{}, B0{});
            if (ragged) tile(nt - 1, std::true_type{}, B1{});
        } else if (ragged) {
            tile(nt - 1, std::true_type{}, B0{});
        }
    } else {
        for (int t = 0; t < nfull; ++t) tile(t, std::false_type{}, BR{});
        if (ragged) tile(nt - 1, std::true_type{}, BR{});
    }

    // ---- epilogue: normalise and store (lane owns row qrow; 4 consecutive dims per 8-byte store)
    const float l_tot = xhalf_sum(l_run);
    const float inv = 1.0f / l_tot;
    if constexpr (DP == 64 && !PAD) {
        const int row0 = qt * ATT_BM + wid * 32;                      // wave-uniform
        att_store_block_d64<DT>(smem + wid * (32 * ATT_O_PITCH), oacc, inv,
                                a.o + b * a.o_sb + (long long)row0 * a.o_sn + h * a.o_sh, a.o_sn, a.Nq - row0, lane);
    } else if (qrow < a.Nq) {
        unsigned short* op = a.o + b * a.o_sb + (long long)qrow * a.o_sn + h * a.o_sh;
#pragma unroll
        for (int d = 0; d < NOD; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                u16x4_t w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = Cvt<DT>::from_f32(oacc[d][4 * g4 + j] * inv);
                if (!PAD || d * 32 + 8 * g4 + 4 * hh < a.D)   // D % 4 == 0: a 4-dim group is inside or outside
                    *reinterpret_cast<u16x4_t*>(op + d * 32 + 8 * g4 + 4 * hh) = w;
            }
    }
}

#ifdef PGN_TUNING_HOOKS   // the ring kernel below replaced this one everywhere; kept for A/B runs (variant 3)
// =====================================================================================================================
// The UNet's shape class (head dim 64, no key bias): 4 wavefronts x 64 query rows, K / V tiles by LDS-DMA.
//   * a wavefront owns two 32-row query blocks: every K / V fragment read from LDS feeds two MFMAs (half the LDS read
//     bytes per flop of the kernel above), a workgroup covers 256 rows (half the global -> LDS traffic per flop);
//   * K / V tiles go global -> LDS with global_load_lds_dwordx4 (no staging registers, no ds_write): each wave
//     instruction fills 1 KiB of consecutive LDS; the swizzled K image and the sub-tiled V image of the kernel above
//     are produced by permuting the per-lane SOURCE addresses; tile t+1 is in flight while tile t is computed, one
//     barrier per tile;
//   * two workgroups per CU (~230 VGPRs): the two query blocks of a wave are independent MFMA / VALU chains that the
//     scheduler interleaves inside one wave.
template <int DT>
__global__ __launch_bounds__(256, 2) void pgn_attn_w64_kernel(AttnArgs a) {
    constexpr int DP = 64, QB = 2;
    constexpr int TILE_BYTES = ATT_BN * DP * 2;          // 8 KiB
    constexpr int NKS = DP / 16, NOD = DP / 32;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * TILE_BYTES];   // [buf][K | V]

    const unsigned nwg = gridDim.x, orig = blockIdx.x;
    const unsigned xcd = orig & 7u, qd = nwg >> 3, rm = nwg & 7u;
    const unsigned wgid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (orig >> 3);
    const int bh = (int)(wgid / (unsigned)a.nqt), qt = (int)(wgid % (unsigned)a.nqt);
    const int b = bh / a.H, h = bh % a.H;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;

    int qrow[QB];
    u16x8_t qf[QB][NKS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        qrow[qb] = qt * 256 + wid * 64 + qb * 32 + l31;
        const int qrow_c = qrow[qb] < a.Nq ? qrow[qb] : a.Nq - 1;
        const unsigned short* qp = a.q + b * a.q_sb + (long long)qrow_c * a.q_sn + h * a.q_sh + 8 * hh;
#pragma unroll
        for (int s = 0; s < NKS; ++s) qf[qb][s] = *reinterpret_cast<const u16x8_t*>(qp + 16 * s);
    }

    const char* kbase = reinterpret_cast<const char*>(a.k + b * a.k_sb + h * a.k_sh);
    const char* vbase = reinterpret_cast<const char*>(a.v + b * a.v_sb + h * a.v_sh);
    const unsigned k_rowb = (unsigned)a.k_sn * 2u, v_rowb = (unsigned)a.v_sn * 2u;
    const int nt = (a.Nk + ATT_BN - 1) / ATT_BN;
    const int rem = a.Nk - (nt - 1) * ATT_BN;            // keys of the last tile (1..64)

    // LDS-DMA: wave w, instruction i fills the 16-byte slots [(2 w + i) 64, +64) of a tile image; a lane supplies the
    // (row, chunk) whose image position its slot is.  Rows past the end of a ragged last tile re-read its last row
    // (their scores are masked to -inf, so their probabilities are exactly 0 and V's content does not matter).
    unsigned ksrc[2], vsrc[2], ksrc_last[2], vsrc_last[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int slot = (wid * 2 + i) * 64 + lane;
        const int krow = slot >> 3, kch = (slot & 7) ^ ((krow >> 1) & 7);                 // inverse of k_lds_off<64>
        const int vrow = ((slot >> 6) << 3) | ((slot >> 2) & 7), vch = (((slot >> 5) & 1) << 2) | (slot & 3);  // of v_lds_off
        ksrc[i] = (unsigned)krow * k_rowb + (unsigned)kch * 16u;
        vsrc[i] = (unsigned)vrow * v_rowb + (unsigned)vch * 16u;
        ksrc_last[i] = (unsigned)min(krow, rem - 1) * k_rowb + (unsigned)kch * 16u;
        vsrc_last[i] = (unsigned)min(vrow, rem - 1) * v_rowb + (unsigned)vch * 16u;
    }
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto load_tile = [&](int t, int buf) __attribute__((always_inline)) {
        const char* gk = kbase + (size_t)t * ATT_BN * k_rowb;
        const char* gv = vbase + (size_t)t * ATT_BN * v_rowb;
        unsigned char* l = smem + buf * (2 * TILE_BYTES) + wid * 2048;
        const bool last = t == nt - 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)(gk + (last ? ksrc_last[i] : ksrc[i])), (lptr_t)(l + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(gv + (last ? vsrc_last[i] : vsrc[i])),
                                             (lptr_t)(l + TILE_BYTES + i * 1024), 16, 0, 0);
        }
    };

    int k_roff[NKS];
#pragma unroll
    for (int s = 0; s < NKS; ++s) k_roff[s] = k_lds_off<DP>(l31, 2 * s + hh);    // + 4096 for key block 1
    const int i16 = lane & 15, g16 = (lane >> 4) & 1;
    const int v_lane_off = 64 * (i16 >> 2) + 32 * g16 + 8 * (i16 & 3);

    f32x16_t oacc[QB][NOD];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
        for (int d = 0; d < NOD; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qb][d][r] = 0.f;
        m_run[qb] = -1.0e30f;
        l_run[qb] = 0.f;
    }
    const f32x2_t c2 = {a.c, a.c};

    load_tile(0, 0);
    __syncthreads();

    auto tile = [&](int t, auto masked_tag, auto buf_tag) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        constexpr int BUF = decltype(buf_tag)::value;
        if (t + 1 < nt) load_tile(t + 1, BUF ^ 1);        // its last readers passed the barrier of tile t-1
        const unsigned char* kt = smem + BUF * (2 * TILE_BYTES);
        const unsigned char* vt = kt + TILE_BYTES;

        f32x16_t sacc[QB][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[qb][kb][r] = 0.f;
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                const u16x8_t kf = *reinterpret_cast<const u16x8_t*>(kt + kb * 4096 + k_roff[s]);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) sacc[qb][kb] = mfma32<DT>(kf, qf[qb][s], sacc[qb][kb]);
            }
        }

        u16x8_t pf[QB][2][2];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            if constexpr (MASKED) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= rem) sacc[qb][kb][r] = -INFINITY;
            }
            // online softmax with a deferred reference (see pgn_attn_fwd_kernel)
            float tmax = sacc[qb][0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[qb][kb][r]);
            tmax = xhalf_max(tmax);
            if (__any((tmax - m_run[qb]) * a.c > ATT_DEFER_LOG2)) {
                const float m_new = fmaxf(m_run[qb], tmax);
                const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * a.c);
                l_run[qb] *= alpha;
                const f32x2_t alpha2 = {alpha, alpha};
#pragma unroll
                for (int d = 0; d < NOD; ++d)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        f32x2_t o2 = {oacc[qb][d][r], oacc[qb][d][r + 1]};
                        o2 *= alpha2;
                        oacc[qb][d][r] = o2[0];
                        oacc[qb][d][r + 1] = o2[1];
                    }
                m_run[qb] = m_new;
            }
            const float mc = m_run[qb] * a.c;
            const f32x2_t nmc2 = {-mc, -mc};
            f32x2_t lsum2[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f32x2_t x = {sacc[qb][kb][r], sacc[qb][kb][r + 1]};
                    x = x * c2 + nmc2;
                    f32x2_t p = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
                    lsum2[(r >> 1) & 1] += p;
                    pf[qb][kb][r >> 3][r & 7] = Cvt<DT>::from_f32(p[0]);
                    pf[qb][kb][r >> 3][(r & 7) + 1] = Cvt<DT>::from_f32(p[1]);
                }
            lsum2[0] += lsum2[1];
            l_run[qb] += lsum2[0][0] + lsum2[0][1];
        }

#pragma unroll
        for (int d = 0; d < NOD; ++d)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int key0 = kb * 32 + 16 * s2 + 4 * hh;
                    const int off0 = (16 * DP) * (key0 >> 3) + 64 * (key0 & 7) + 512 * d + v_lane_off;
                    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4_t*)(vt + off0));
                    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4_t*)(vt + off0 + 16 * DP));
                    u16x8_t vf;
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) oacc[qb][d] = mfma32<DT>(vf, pf[qb][kb][s2], oacc[qb][d]);
                }
        __syncthreads();                                   // tile t+1 has landed (vmcnt) and tile t is free to refill
    };

    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    const bool ragged = rem < ATT_BN;
    const int nfull = ragged ? nt - 1 : nt;
    int t = 0;
    for (; t + 1 < nfull; t += 2) {
        tile(t, std::false_type{}, B0{});
        tile(t + 1, std::false_type{}, B1{});
    }
    if (t < nfull) {
        tile(t, std::false_type{}, B0{});
        if (ragged) tile(nt - 1, std::true_type{}, B1{});
    } else if (ragged) {
        tile(nt - 1, std::true_type{}, B0{});
    }

#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float l_tot = xhalf_sum(l_run[qb]);
        const float inv = 1.0f / l_tot;
        const int row0 = qt * 256 + wid * 64 + qb * 32;               // wave-uniform
        att_store_block_d64<DT>(smem + wid * (32 * ATT_O_PITCH), oacc[qb], inv,
                                a.o + b * a.o_sb + (long long)row0 * a.o_sn + h * a.o_sh, a.o_sn, a.Nq - row0, lane);
    }
}


#endif  // PGN_TUNING_HOOKS

// =====================================================================================================================
// The 64-row kernel with a RING of three K / V buffers and counted waits (RAGGED: a masked last tile for Nk % 64 != 0).
// Why: per-tile SQ counters at N = 4096 against N = 1024 (profiles/r02_attention_sq_counters_L1_L2.txt) show a fixed
// cost of ~4.8 tile times per (batch, head, query block) workgroup, more than half of it waiting - and a workgroup's
// share of K / V tiles that miss L2 is fixed too (a head's K / V is fetched once and shared by its Nq / 256 workgroups,
// which run in lockstep and miss together: 4 tile loads per workgroup at either size).  In the kernel above the tile
// prefetched at the start of tile t has to land by the MIDDLE of tile t: the compiler answers the first
// ds_read_b64_tr_b16 it sees after an LDS-DMA with s_waitcnt vmcnt(0) (it cannot tell the LDS regions apart), and
// __syncthreads() adds another vmcnt(0).  ~1 us is less than an HBM round trip under load, so every missing tile stalls.
// Here tile t+2 is issued at the start of tile t and waited for with vmcnt(4) - "all but the newest tile" - at the end of
// tile t+... i.e. two tile times later, before a raw s_barrier; the V fragment reads are inline asm (invisible to the
// wait-count pass), issued early (block d = 0 between the two query blocks' softmax, d = 1 after it) and fenced by
// counted lgkmcnt waits that take the fragment registers as in/out operands, so that no use can move above its wait.
// Template parameters: the shipped instantiations are <DT, SUMCHK = true, PACKED = false, PIPE = 2, RAGGED = false / true>;
//   SUMCHK  no row maximum on the common path (the lane's sum of its 32 probabilities is the overflow check)
//   PACKED  softmax arithmetic in packed fp32 instructions (measurement build only: they do not issue under an MFMA)
//   PIPE    0: per-block order + first-tile branch, 1: no first-tile branch, 2: both V fragment blocks read before the
//           second query block's softmax and the first block's PV MFMAs placed between its exponentials and its check
// (the earlier forms stay selectable in the measurement build so that every A/B in profiles/ can be repeated).
template <int DT, bool SUMCHK, bool PACKED = false, int PIPE = 0, bool RAGGED = false>
__global__ __launch_bounds__(256, 2) void pgn_attn_w64r_kernel(AttnArgs a) {
    constexpr int DP = 64, QB = 2;
    constexpr int TILE_BYTES = ATT_BN * DP * 2;          // 8 KiB
    constexpr int NKS = DP / 16, NOD = DP / 32;
    static_assert(NOD == 2, "two asm blocks of V reads");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[6 * TILE_BYTES];   // [buf 0..2][K | V]

    const unsigned nwg = gridDim.x, orig = blockIdx.x;
    const unsigned xcd = orig & 7u, qd = nwg >> 3, rm = nwg & 7u;
    const unsigned wgid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (orig >> 3);
    const int bh = (int)(wgid / (unsigned)a.nqt), qt = (int)(wgid % (unsigned)a.nqt);
    const int b = bh / a.H, h = bh % a.H;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
#ifdef PGN_TUNING_HOOKS
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
    if (a.dbg) ts0 = __builtin_amdgcn_s_memrealtime();
#endif

    u16x8_t qf[QB][NKS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qrow = qt * 256 + wid * 64 + qb * 32 + l31;
        const int qrow_c = qrow < a.Nq ? qrow : a.Nq - 1;
        const unsigned short* qp = a.q + b * a.q_sb + (long long)qrow_c * a.q_sn + h * a.q_sh + 8 * hh;
#pragma unroll
        for (int s = 0; s < NKS; ++s) qf[qb][s] = *reinterpret_cast<const u16x8_t*>(qp + 16 * s);
    }

    const char* kbase = reinterpret_cast<const char*>(a.k + b * a.k_sb + h * a.k_sh);
    const char* vbase = reinterpret_cast<const char*>(a.v + b * a.v_sb + h * a.v_sh);
    const unsigned k_rowb = (unsigned)a.k_sn * 2u, v_rowb = (unsigned)a.v_sn * 2u;
    // RAGGED: the last tile holds rem < 64 keys (its missing rows re-read row rem - 1, their scores are masked to -inf)
    const int nt = RAGGED ? (a.Nk + ATT_BN - 1) / ATT_BN : a.Nk / ATT_BN;
    const int rem = a.Nk - (nt - 1) * ATT_BN;

    unsigned ksrc[2], vsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int slot = (wid * 2 + i) * 64 + lane;
        const int krow = slot >> 3, kch = (slot & 7) ^ ((krow >> 1) & 7);                 // inverse of k_lds_off<64>
        const int vrow = ((slot >> 6) << 3) | ((slot >> 2) & 7), vch = (((slot >> 5) & 1) << 2) | (slot & 3);  // of v_lds_off
        ksrc[i] = (unsigned)krow * k_rowb + (unsigned)kch * 16u;
        vsrc[i] = (unsigned)vrow * v_rowb + (unsigned)vch * 16u;
    }
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    // four DMA instructions per wave and tile: the vmcnt arithmetic below counts on exactly that
    auto load_tile = [&](int t, int buf) __attribute__((always_inline)) {
        const char* gk = kbase + (size_t)t * ATT_BN * k_rowb;
        const char* gv = vbase + (size_t)t * ATT_BN * v_rowb;
        unsigned char* l = smem + buf * (2 * TILE_BYTES) + wid * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // keep the 32 -> 64-bit extension of the lane offsets next to their use: hoisted out of the loop they
            // become register PAIRS, the DMA loses its scalar-base + 32-bit-offset form and the pairs spill
            unsigned ko = ksrc[i], vo = vsrc[i];
            if constexpr (RAGGED) {
                if (t == nt - 1) {                               // once per workgroup: clamp the rows past the end
                    const int slot = (wid * 2 + i) * 64 + lane;
                    const int krow = slot >> 3, kch = (slot & 7) ^ ((krow >> 1) & 7);
                    const int vrow = ((slot >> 6) << 3) | ((slot >> 2) & 7), vch = (((slot >> 5) & 1) << 2) | (slot & 3);
                    ko = (unsigned)min(krow, rem - 1) * k_rowb + (unsigned)kch * 16u;
                    vo = (unsigned)min(vrow, rem - 1) * v_rowb + (unsigned)vch * 16u;
                }
            }
            asm volatile("" : "+v"(ko), "+v"(vo));
            __builtin_amdgcn_global_load_lds((gptr_t)(gk + ko), (lptr_t)(l + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(gv + vo), (lptr_t)(l + TILE_BYTES + i * 1024), 16, 0, 0);
        }
    };

    int k_roff[NKS];
#pragma unroll
    for (int s = 0; s < NKS; ++s) k_roff[s] = k_lds_off<DP>(l31, 2 * s + hh);    // + 4096 for key block 1
    const int i16 = lane & 15, g16 = (lane >> 4) & 1;
    // per-lane part of a V fragment address: 64 * (4 hh) for the lane half's keys + the transposing read's lane map
    const unsigned v_lane = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)smem +
                            (unsigned)(TILE_BYTES + 256 * hh + 64 * (i16 >> 2) + 32 * g16 + 8 * (i16 & 3));

    f32x16_t oacc[QB][NOD];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
        for (int d = 0; d < NOD; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qb][d][r] = 0.f;
        m_run[qb] = -1.0e30f;
        l_run[qb] = 0.f;
    }
    const f32x2_t c2 = {a.c, a.c};

    // prologue: the Q loads above are older than the DMAs, so "all but the newest tile" covers them and tile 0
    load_tile(0, 0);
    if (nt > 1) {
        load_tile(1, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#ifdef PGN_TUNING_HOOKS
    if (a.dbg) ts1 = __builtin_amdgcn_s_memrealtime();
#endif

#define PGN_TRV(i, OFF) "ds_read_b64_tr_b16 %" #i ", %8 offset:" #OFF "\n"
    auto tile = [&](int t, auto buf_tag, auto maybe_last_tag) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr bool MAYBE_LAST = decltype(maybe_last_tag)::value;
        constexpr int NEXT2 = (BUF + 2) % 3;
        if (t + 2 < nt) load_tile(t + 2, NEXT2);          // buffer of tile t-1: its readers passed the last barrier
        const unsigned char* kt = smem + BUF * (2 * TILE_BYTES);
        const unsigned vaddr = v_lane + (unsigned)(BUF * (2 * TILE_BYTES));

        f32x16_t sacc[QB][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[qb][kb][r] = 0.f;
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                const u16x8_t kf = *reinterpret_cast<const u16x8_t*>(kt + kb * 4096 + k_roff[s]);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) sacc[qb][kb] = mfma32<DT>(kf, qf[qb][s], sacc[qb][kb]);
            }
        }

        if constexpr (RAGGED && MAYBE_LAST) {
            if (t == nt - 1) {                                   // wave-uniform; key of sacc[.][kb][r] = kb*32 + (r&3) + 8*(r>>2) + 4*hh
                int lim = rem - 4 * hh;
                asm volatile("" : "+v"(lim));                   // the 32 compare masks are loop invariants: keep them in here
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (kb * 32 + (r & 3) + 8 * (r >> 2) >= lim) sacc[qb][kb][r] = -INFINITY;
            }
        }
        u16x8_t pf[QB][2][2];
        s16x4_t f0, f1, f2, f3, f4, f5, f6, f7;            // V fragments of output dims 0..31: keys (2 j) | (2 j + 1) halves
        s16x4_t g0, g1, g2, g3, g4, g5, g6, g7;            // output dims 32..63
        // PV of ONE query block against the fragments of output-dim block d
        auto pv_q = [&](int qb, int d, const s16x4_t(&fr)[8]) __attribute__((always_inline)) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const s16x4_t lo = fr[2 * (2 * kb + s2)], hi = fr[2 * (2 * kb + s2) + 1];
                    u16x8_t vf;
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                    oacc[qb][d] = mfma32<DT>(vf, pf[qb][kb][s2], oacc[qb][d]);
                }
        };
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            // raise the reference to the exact running maximum (O, l rescaled): the rare path of both forms
            auto raise_reference = [&]() __attribute__((always_inline)) {
                float tmax = sacc[qb][0][0];
                // the maximum chain starts from an opaque value: it must stay inside the rare branch (the scores
                // are loop invariants of the sum-check loop below and would be hoisted onto the common path)
                asm volatile("" : "+v"(tmax));
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[qb][kb][r]);
                tmax = xhalf_max(tmax);
                const float m_new = fmaxf(m_run[qb], tmax);
                const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * a.c);
                l_run[qb] *= alpha;
                const f32x2_t alpha2 = {alpha, alpha};
#pragma unroll
                for (int d = 0; d < NOD; ++d)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        f32x2_t o2 = {oacc[qb][d][r], oacc[qb][d][r + 1]};
                        o2 *= alpha2;
                        oacc[qb][d][r] = o2[0];
                        oacc[qb][d][r + 1] = o2[1];
                    }
                m_run[qb] = m_new;
            };
            f32x2_t lsum2[2];
            auto probabilities = [&]() __attribute__((always_inline)) {
                const float mc = m_run[qb] * a.c;
                if constexpr (PACKED) {
                    const f32x2_t nmc2 = {-mc, -mc};
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; r += 2) {
                            f32x2_t x = {sacc[qb][kb][r], sacc[qb][kb][r + 1]};
                            x = __builtin_elementwise_fma(x, c2, nmc2);
                            f32x2_t p = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
                            if (kb == 0 && r < 4) lsum2[r >> 1] = p;     // two chains, no zero-initialised adds
                            else lsum2[(r >> 1) & 1] += p;
                            pf[qb][kb][r >> 3][r & 7] = Cvt<DT>::from_f32(p[0]);
                            pf[qb][kb][r >> 3][(r & 7) + 1] = Cvt<DT>::from_f32(p[1]);
                        }
                    lsum2[0] += lsum2[1];
                } else {
                    // PLAIN fp32 instructions on purpose: on gfx950 v_fma_f32 / v_add_f32 / v_cvt_pk issue underneath a
                    // running MFMA at no cost, while the packed forms (v_pk_fma_f32, v_pk_add_f32) hold the matrix
                    // pipe for 4.6-6.5 cycles each (scripts/ubench_overlap.hip).  This file is compiled with
                    // -fno-slp-vectorize so that the scalar statements below stay scalar.
                    const float nmc = -mc;
                    float ls[4];
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[qb][kb][r], a.c, nmc));
                            if (kb == 0 && r < 4) ls[r] = p;             // four chains
                            else ls[r & 3] += p;
                            pf[qb][kb][r >> 3][r & 7] = Cvt<DT>::from_f32(p);
                        }
                    lsum2[0] = f32x2_t{ls[0] + ls[2], ls[1] + ls[3]};
                }
            };
            if constexpr (SUMCHK) {
                // No row maximum on the common path: the probabilities are taken against the current reference and
                // the lane's own sum of its 32 of them is the overflow check - it bounds every one of them, so
                // "sum <= 2^8" keeps P inside the same range as the deferred-maximum form (and is false for the inf
                // and NaN of scores that overflowed).  Only then is the exact maximum taken and the tile's
                // probabilities computed again (<= 1 each).  The first tile has no reference yet and takes it first.
                if constexpr (PIPE == 0) {
                    if (t == 0) raise_reference();
                }
                probabilities();
                if constexpr (PIPE == 2) {
                    if (qb == 1) {
                        // the first block's PV MFMAs sit between the second block's exponentials and its check: they
                        // are independent of both, so the scheduler can issue them underneath the plain VALU work
                        asm volatile("s_waitcnt lgkmcnt(8)"
                                     : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)
                                     :
                                     : "memory");
                        {
                            const s16x4_t fr[8] = {f0, f1, f2, f3, f4, f5, f6, f7};
                            pv_q(0, 0, fr);
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)"
                                     : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3), "+v"(g4), "+v"(g5), "+v"(g6), "+v"(g7)
                                     :
                                     : "memory");
                        {
                            const s16x4_t fr[8] = {g0, g1, g2, g3, g4, g5, g6, g7};
                            pv_q(0, 1, fr);
                        }
                    }
                }
                if (__any(!(lsum2[0][0] + lsum2[0][1] <= 256.0f))) {
                    raise_reference();
                    probabilities();
                }
            } else {
                float tmax = sacc[qb][0][0];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[qb][kb][r]);
                tmax = xhalf_max(tmax);
                if (__any((tmax - m_run[qb]) * a.c > ATT_DEFER_LOG2)) raise_reference();
                probabilities();
            }
            l_run[qb] += lsum2[0][0] + lsum2[0][1];
            if (PIPE == 2 && qb == 0) {
                asm volatile(PGN_TRV(0, 0) PGN_TRV(1, 1024) PGN_TRV(2, 2048) PGN_TRV(3, 3072) PGN_TRV(4, 4096)
                             PGN_TRV(5, 5120) PGN_TRV(6, 6144) PGN_TRV(7, 7168)
                             : "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3), "=&v"(f4), "=&v"(f5), "=&v"(f6), "=&v"(f7)
                             : "v"(vaddr)
                             : "memory");
                asm volatile(PGN_TRV(0, 512) PGN_TRV(1, 1536) PGN_TRV(2, 2560) PGN_TRV(3, 3584) PGN_TRV(4, 4608)
                             PGN_TRV(5, 5632) PGN_TRV(6, 6656) PGN_TRV(7, 7680)
                             : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3), "=&v"(g4), "=&v"(g5), "=&v"(g6), "=&v"(g7)
                             : "v"(vaddr)
                             : "memory");
            } else if (qb == 0) {
                // the first query block's scores are dead: their registers take the d = 0 fragments, whose LDS latency
                // then hides under the second block's softmax
                asm volatile(PGN_TRV(0, 0) PGN_TRV(1, 1024) PGN_TRV(2, 2048) PGN_TRV(3, 3072) PGN_TRV(4, 4096)
                             PGN_TRV(5, 5120) PGN_TRV(6, 6144) PGN_TRV(7, 7168)
                             : "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3), "=&v"(f4), "=&v"(f5), "=&v"(f6), "=&v"(f7)
                             : "v"(vaddr)
                             : "memory");
            }
        }
        if constexpr (PIPE == 2) {
            {
                const s16x4_t fr[8] = {f0, f1, f2, f3, f4, f5, f6, f7};
                pv_q(1, 0, fr);
            }
            {
                const s16x4_t fr[8] = {g0, g1, g2, g3, g4, g5, g6, g7};
                pv_q(1, 1, fr);
            }
        } else {
        asm volatile(PGN_TRV(0, 512) PGN_TRV(1, 1536) PGN_TRV(2, 2560) PGN_TRV(3, 3584) PGN_TRV(4, 4608) PGN_TRV(5, 5632)
                     PGN_TRV(6, 6656) PGN_TRV(7, 7680)
                     : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3), "=&v"(g4), "=&v"(g5), "=&v"(g6), "=&v"(g7)
                     : "v"(vaddr)
                     : "memory");
        // d = 0 fragments have landed when at most the 8 newer reads are outstanding; the in/out operands pin every
        // use of f* below this wait (hipcc moves register-only instructions across an asm s_waitcnt otherwise)
        asm volatile("s_waitcnt lgkmcnt(8)"
                     : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)
                     :
                     : "memory");
        auto pv = [&](int d, const s16x4_t(&fr)[8]) __attribute__((always_inline)) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const s16x4_t lo = fr[2 * (2 * kb + s2)], hi = fr[2 * (2 * kb + s2) + 1];
                    u16x8_t vf;
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) oacc[qb][d] = mfma32<DT>(vf, pf[qb][kb][s2], oacc[qb][d]);
                }
        };
        {
            const s16x4_t fr[8] = {f0, f1, f2, f3, f4, f5, f6, f7};
            pv(0, fr);
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3), "+v"(g4), "+v"(g5), "+v"(g6), "+v"(g7)
                     :
                     : "memory");
        {
            const s16x4_t fr[8] = {g0, g1, g2, g3, g4, g5, g6, g7};
            pv(1, fr);
        }
        }
        // tile t+1 (issued one tile ago) must have landed in every wave's share before anyone reads it: all but the
        // newest tile's four DMAs retired, then the barrier.  The last tiles have nothing newer in flight.
        if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
#undef PGN_TRV

    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    using B2 = std::integral_constant<int, 2>;
    // RAGGED: the last tile is the third of the last full round (nt % 3 == 0) or one of the two remainder tiles; those
    // three tile bodies carry the masking code behind a wave-uniform "is this the last tile" branch
    using ML = std::integral_constant<bool, RAGGED>;
    int t = 0;
    for (; t + 2 < nt; t += 3) {
        tile(t, B0{}, std::false_type{});
        tile(t + 1, B1{}, std::false_type{});
        tile(t + 2, B2{}, ML{});
    }
    if (t < nt) tile(t, B0{}, ML{});
    if (t + 1 < nt) tile(t + 1, B1{}, ML{});
#ifdef PGN_TUNING_HOOKS
    if (a.dbg) ts2 = __builtin_amdgcn_s_memrealtime();
#endif

#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float l_tot = xhalf_sum(l_run[qb]);
        const float inv = 1.0f / l_tot;
        const int row0 = qt * 256 + wid * 64 + qb * 32;               // wave-uniform
        att_store_block_d64<DT>(smem + wid * (32 * ATT_O_PITCH), oacc[qb], inv,
                                a.o + b * a.o_sb + (long long)row0 * a.o_sn + h * a.o_sh, a.o_sn, a.Nq - row0, lane);
    }
#ifdef PGN_TUNING_HOOKS
    if (a.dbg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the output stores have left
        const unsigned long long ts3 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* d = a.dbg + (size_t)blockIdx.x * 6;
            d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = ts3; d[4] = hwid; d[5] = xcc;
        }
    }
#endif
}


#ifdef PGN_TUNING_HOOKS   // measured slower than the 32x32x16 ring kernel (see the comment): measurement build only, variant 20
// =====================================================================================================================
// Round 3: the ring kernel's products on v_mfma_f32_16x16x32_bf16 (MI355X_MICROARCH.md, DVFS
// give-back item 7: the chip holds a higher clock on this shape - bare MFMA loops on random data +13 % wall at equal
// cycles, +4 % under this kernel's softmax VALU mix at 12 % more cycles: profiles/r03_ubench_mfma_shape.txt).
// RESULT (profiles/r03_attention_x16_ab.txt, same process, interleaved rounds): correct to the same 0.703 of the bar,
// and 7 % SLOWER at N = 4096, 12 % at N = 1024, 9-10 % on the cross-attention shapes (998 / 807 / 641 / 585 against
// 1071 / 915 / 713 / 641 TFLOP/s), no scratch, 255 VGPRs.  Twice the MFMA instructions hold the SIMD's vector issue for
// twice as long (8 cycles each, whatever the shape), and this loop is bound by vector issue, not by the matrix pipe; the
// higher clock does not pay that back.  Kept in the measurement build (variant 20) so that the A/B can be repeated.
// Same workgroup (4 waves x 64 query rows), same three-buffer LDS-DMA ring and counted waits, same softmax (plain fp32
// instructions, deferred reference, the lane's own sum as the overflow check); what changes is the tiling inside a wave:
//   * S^T = K Q^T in 16-key x 16-query blocks (K = 32 head dims per MFMA): a lane holds FOUR keys of one query column
//     per block, and the four lanes i, i+16, i+32, i+48 share a query.  The row sum stays a lane-local partial that is
//     reduced over the four lanes once, in the epilogue; the reference is raised for all four together (two swaps);
//   * the S^T accumulators are the B operand of O^T = V^T P^T as they are: k-slot 8 g + j of a PV MFMA takes register
//     j & 3 of key block 2 ks + (j >> 2) - the contraction index may be permuted freely as long as V's fragment
//     (ds_read_b64_tr_b16, one read per key block) uses the same permutation.  No cross-lane movement for P;
//   * both LDS images have plain 128-byte key rows with the 16-byte chunk index XOR-swizzled, each with the function
//     that makes its own read conflict-free for the lane groups the hardware serves together (ds_read_b128: four
//     NON-contiguous 16-lane groups; ds_read_b64_tr_b16: the two 32-lane halves; MI355X_MICROARCH.md, LDS table).
template <int DT>
__device__ __forceinline__ f32x4_t mfma16(u16x8_t a, u16x8_t b, f32x4_t c) {
    if constexpr (DT == PGN_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                       c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c,
                                                      0, 0, 0);
}
// max / sum over the four lanes l, l ^ 16, l ^ 32, l ^ 48 (the 16-lane groups that share a query column)
__device__ __forceinline__ float xquad_max(float x) {
    const unsigned u = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return xhalf_max(fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])));
}
__device__ __forceinline__ float xquad_sum(float x) {
    const unsigned u = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return xhalf_sum(__uint_as_float(r[0]) + __uint_as_float(r[1]));
}

template <int DT, bool RAGGED>
__global__ __launch_bounds__(256, 2) void pgn_attn_x16_kernel(AttnArgs a) {
    constexpr int DP = 64;
    constexpr int NQ = 4;                                 // 16-row query blocks per wave
    constexpr int NKB = 4;                                // 16-key blocks per tile
    constexpr int NDB = 4;                                // 16-dim blocks of the output
    constexpr int TILE_BYTES = ATT_BN * DP * 2;          // 8 KiB
    __shared__ __attribute__((aligned(1024))) unsigned char smem[6 * TILE_BYTES];   // [buf 0..2][K | V]

    const unsigned nwg = gridDim.x, orig = blockIdx.x;
    const unsigned xcd = orig & 7u, qd = nwg >> 3, rm = nwg & 7u;
    const unsigned wgid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (orig >> 3);
    const int bh = (int)(wgid / (unsigned)a.nqt), qt = (int)(wgid % (unsigned)a.nqt);
    const int b = bh / a.H, h = bh % a.H;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;

    // Q fragments (B operand of S^T = K Q^T): lane holds Q[16 qb + i16][32 s + 8 g + j]
    u16x8_t qf[NQ][2];
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) {
        const int qrow = qt * 256 + wid * 64 + qb * 16 + i16;
        const int qrow_c = qrow < a.Nq ? qrow : a.Nq - 1;
        const unsigned short* qp = a.q + b * a.q_sb + (long long)qrow_c * a.q_sn + h * a.q_sh + 8 * g;
#pragma unroll
        for (int s = 0; s < 2; ++s) qf[qb][s] = *reinterpret_cast<const u16x8_t*>(qp + 32 * s);
    }

    const char* kbase = reinterpret_cast<const char*>(a.k + b * a.k_sb + h * a.k_sh);
    const char* vbase = reinterpret_cast<const char*>(a.v + b * a.v_sb + h * a.v_sh);
    const unsigned k_rowb = (unsigned)a.k_sn * 2u, v_rowb = (unsigned)a.v_sn * 2u;
    const int nt = RAGGED ? (a.Nk + ATT_BN - 1) / ATT_BN : a.Nk / ATT_BN;
    const int rem = a.Nk - (nt - 1) * ATT_BN;

    // LDS-DMA: both images have 128-byte key rows with the 16-byte chunk index XOR-swizzled: K by (row >> 1) & 7
    // (k_lds_off<64>), V by (row & 6) | ((row >> 3) & 1) - each makes ITS read conflict-free (see below).
    // The per-lane source offsets are recomputed for every tile from an opaque copy of the lane id (a dozen VALU
    // instructions per tile): as loop invariants they are the first registers the allocator spills, and a spill's
    // reload comes with a vmcnt(0) that serialises the ring (and breaks its counted waits).
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    // four DMA instructions per wave and tile: the vmcnt arithmetic below counts on exactly that
    auto load_tile = [&](int t, int buf) __attribute__((always_inline)) {
        const char* gk = kbase + (size_t)t * ATT_BN * k_rowb;
        const char* gv = vbase + (size_t)t * ATT_BN * v_rowb;
        unsigned char* l = smem + buf * (2 * TILE_BYTES) + wid * 2048;
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int slot = (wid * 2 + i) * 64 + ln;
            int row = slot >> 3;
            const int kch = (slot & 7) ^ ((row >> 1) & 7);                       // K image: k_lds_off<64>
            const int vch = (slot & 7) ^ ((row & 6) | ((row >> 3) & 1));         // V image: x16_v_swz
            if constexpr (RAGGED) {
                if (t == nt - 1) row = min(row, rem - 1);          // rows past the end re-read the last row
            }
            unsigned ko = (unsigned)row * k_rowb + (unsigned)kch * 16u, vo = (unsigned)row * v_rowb + (unsigned)vch * 16u;
            asm volatile("" : "+v"(ko), "+v"(vo));   // see pgn_attn_w64r_kernel: keep the zero extension next to its use
            __builtin_amdgcn_global_load_lds((gptr_t)(gk + ko), (lptr_t)(l + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(gv + vo), (lptr_t)(l + TILE_BYTES + i * 1024), 16, 0, 0);
        }
    };

    // K fragment (A operand of S^T): accumulator row i16 of key block kb is key 16 kb + i16, head dims 32 s + 8 g ..
    // ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32): 8 rows with chunk
    // 4 s + g and 8 rows with chunk 4 s + (g ^ 1) - with the XOR by (row >> 1) & 7 the 16 lanes of a group hit 16
    // different 16-byte columns of the 256-byte bank line.
    int k_roff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) k_roff[s] = i16 * 128 + (((4 * s + g) ^ ((i16 >> 1) & 7)) << 4);   // + 2048 per key block
    // V fragment (A operand of O^T = V^T P^T), one ds_read_b64_tr_b16 per (dim block, key block): lane 4 q + p of a
    // 16-lane group addresses key 4 g + q of the block, dims 16 db + 4 p ..  The read is served per 32-lane half (two
    // groups: 8 keys x 2 chunks, 8 bytes per lane); with the chunk XOR (row & 6) | ((row >> 3) & 1) the 16 chunks of a
    // half lie in 16 different columns.  Key block and ring buffer are immediates.
    const int vq = i16 >> 2, vp = i16 & 3;
    const int vkin = 4 * g + vq;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)smem;
    // dim block db: chunk (2 db) ^ ... -> the address is vb0 ^ (32 db) (smem is 1024-byte aligned, bits 4-6 of the rest
    // are zero); the three XORs are redone per tile behind an opaque copy so that they do not become loop invariants
    const unsigned vb0 = lds0 + (unsigned)(vkin * 128 + (((vp >> 1) ^ ((vkin & 6) | ((vkin >> 3) & 1))) << 4) + 8 * (vp & 1));

    f32x4_t oacc[NQ][NDB];
    float m_run[NQ], l_run[NQ];
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) {
#pragma unroll
        for (int db = 0; db < NDB; ++db) oacc[qb][db] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        m_run[qb] = -1.0e30f;
        l_run[qb] = 0.f;
    }

    load_tile(0, 0);
    if (nt > 1) {
        load_tile(1, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    auto tile = [&](int t, auto buf_tag, auto maybe_last_tag) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr bool MAYBE_LAST = decltype(maybe_last_tag)::value;
        constexpr int NEXT2 = (BUF + 2) % 3;
        constexpr int VOFF = BUF * (2 * TILE_BYTES) + TILE_BYTES;
        if (t + 2 < nt) load_tile(t + 2, NEXT2);
        const unsigned char* kt = smem + BUF * (2 * TILE_BYTES);
        unsigned vb[NDB];
        vb[0] = vb0;
        asm volatile("" : "+v"(vb[0]));
#pragma unroll
        for (int db = 1; db < NDB; ++db) vb[db] = vb[0] ^ (unsigned)(32 * db);

        // ---- S^T blocks: sacc[qb][kb][r] = S[query 16 qb + i16][key 16 kb + 4 g + r]
        f32x4_t sacc[NQ][NKB];
        auto scores = [&](auto q0_tag, auto q1_tag) __attribute__((always_inline)) {
            constexpr int Q0 = decltype(q0_tag)::value, Q1 = decltype(q1_tag)::value;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
                for (int qb = Q0; qb < Q1; ++qb) sacc[qb][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const u16x8_t kf = *reinterpret_cast<const u16x8_t*>(kt + kb * 2048 + k_roff[s]);
#pragma unroll
                    for (int qb = Q0; qb < Q1; ++qb) sacc[qb][kb] = mfma16<DT>(kf, qf[qb][s], sacc[qb][kb]);
                }
            }
            if constexpr (RAGGED && MAYBE_LAST) {
                if (t == nt - 1) {                                   // wave-uniform
                    int lim = rem - 4 * g;
                    asm volatile("" : "+v"(lim));                   // keep the compare masks inside this branch
#pragma unroll
                    for (int qb = Q0; qb < Q1; ++qb)
#pragma unroll
                        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (16 * kb + r >= lim) sacc[qb][kb][r] = -INFINITY;
                }
            }
        };
        using Q_0 = std::integral_constant<int, 0>;
        using Q_2 = std::integral_constant<int, 2>;
        using Q_4 = std::integral_constant<int, 4>;
        scores(Q_0{}, Q_4{});

        u16x8_t pf[NQ][2];                                  // P^T fragments: [query block][k-step of 32 keys]
        s16x4_t f[NDB][NKB];                                // V fragments: [dim block][key block]
        float lsum[NQ];
        auto raise_reference = [&](int qb) __attribute__((always_inline)) {
            float tmax = sacc[qb][0][0];
            asm volatile("" : "+v"(tmax));                  // the chain stays inside the rare branch
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, sacc[qb][kb][r]);
            tmax = xquad_max(tmax);
            const float m_new = fmaxf(m_run[qb], tmax);
            const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * a.c);
            l_run[qb] *= alpha;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int r = 0; r < 4; ++r) oacc[qb][db][r] *= alpha;
            m_run[qb] = m_new;
        };
        auto probabilities = [&](int qb) __attribute__((always_inline)) {
            const float nmc = -(m_run[qb] * a.c);
            float ls[2];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[qb][kb][r], a.c, nmc));
                    if (kb == 0 && r < 2) ls[r] = p;
                    else ls[r & 1] += p;
                    pf[qb][kb >> 1][4 * (kb & 1) + r] = Cvt<DT>::from_f32(p);
                }
            lsum[qb] = ls[0] + ls[1];
        };
        // PV of a 32-row half (query blocks 2 sb, 2 sb + 1) against the fragments of dim blocks [D0, D1)
        auto pv = [&](int sb, int d0, int d1) __attribute__((always_inline)) {
#pragma unroll
            for (int db = d0; db < d1; ++db)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const s16x4_t lo = f[db][2 * ks], hi = f[db][2 * ks + 1];
                    u16x8_t vf;
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2)
                        oacc[2 * sb + q2][db] = mfma16<DT>(vf, pf[2 * sb + q2][ks], oacc[2 * sb + q2][db]);
                }
        };
#define PGN_READ_V(db)                                                                                    \
    asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%5\n"                                                  \
                 "ds_read_b64_tr_b16 %1, %4 offset:%6\n"                                                  \
                 "ds_read_b64_tr_b16 %2, %4 offset:%7\n"                                                  \
                 "ds_read_b64_tr_b16 %3, %4 offset:%8\n"                                                  \
                 : "=&v"(f[db][0]), "=&v"(f[db][1]), "=&v"(f[db][2]), "=&v"(f[db][3])                     \
                 : "v"(vb[db]), "n"(VOFF), "n"(VOFF + 2048), "n"(VOFF + 4096), "n"(VOFF + 6144)           \
                 : "memory")

        // A 32-row half of the wave shares one overflow check, as in the 32x32x16 kernel.  First half: scores still live.
        probabilities(0);
        probabilities(1);
        if (__any(!(lsum[0] + lsum[1] <= 256.0f))) {
            raise_reference(0);
            raise_reference(1);
            probabilities(0);
            probabilities(1);
        }
        l_run[0] += lsum[0];
        l_run[1] += lsum[1];
        // the first half's scores are dead: their registers take the V fragments of dim blocks 0 and 1, whose LDS
        // latency hides under the second half's softmax; dim blocks 2 and 3 follow when the second half's scores die
        PGN_READ_V(0);
        PGN_READ_V(1);
        probabilities(2);
        probabilities(3);
        PGN_READ_V(2);
        PGN_READ_V(3);
        // the first half's PV MFMAs between the second half's exponentials and its check (plain VALU work issues
        // underneath a running MFMA); the fragments are pinned below their waits by in/out operands
        asm volatile("s_waitcnt lgkmcnt(8)"
                     : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[0][2]), "+v"(f[0][3]), "+v"(f[1][0]), "+v"(f[1][1]),
                       "+v"(f[1][2]), "+v"(f[1][3])
                     :
                     : "memory");
        pv(0, 0, 2);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(f[2][0]), "+v"(f[2][1]), "+v"(f[2][2]), "+v"(f[2][3]), "+v"(f[3][0]), "+v"(f[3][1]),
                       "+v"(f[3][2]), "+v"(f[3][3])
                     :
                     : "memory");
        pv(0, 2, 4);
        // Second half: its scores are NOT kept for the rare path (32 registers at the kernel's pressure peak): a failed
        // check computes them again from the K tile, which stays in LDS until this tile's barrier
        if (__any(!(lsum[2] + lsum[3] <= 256.0f))) {
            scores(Q_2{}, Q_4{});
            raise_reference(2);
            raise_reference(3);
            probabilities(2);
            probabilities(3);
        }
        l_run[2] += lsum[2];
        l_run[3] += lsum[3];
        pv(1, 0, 4);
#undef PGN_READ_V
        if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    using B2 = std::integral_constant<int, 2>;
    using ML = std::integral_constant<bool, RAGGED>;
    int t = 0;
    for (; t + 2 < nt; t += 3) {
        tile(t, B0{}, std::false_type{});
        tile(t + 1, B1{}, std::false_type{});
        tile(t + 2, B2{}, ML{});
    }
    if (t < nt) tile(t, B0{}, ML{});
    if (t + 1 < nt) tile(t + 1, B1{}, ML{});

    // ---- epilogue: lane holds O[query 16 qb + i16][dims 16 db + 4 g .. + 3]; two query blocks (32 rows) per round
    // through the wave's [32 rows][144 bytes] LDS image, then whole 128-byte rows (see att_store_block_d64)
    unsigned char* ol = smem + wid * (32 * ATT_O_PITCH);
    // the lane id again (v_mbcnt): a copy kept across the key loop is one more register under its pressure peak
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int i16e = lane_e & 15, ge = lane_e >> 4;
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
            const int qb = 2 * sb + q2;
            const float inv = 1.0f / xquad_sum(l_run[qb]);
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                u16x4_t w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = Cvt<DT>::from_f32(oacc[qb][db][j] * inv);
                *reinterpret_cast<u16x4_t*>(ol + (16 * q2 + i16e) * ATT_O_PITCH + (16 * db + 4 * ge) * 2) = w;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int row0 = qt * 256 + wid * 64 + sb * 32;               // wave-uniform
        unsigned short* orow0 = a.o + b * a.o_sb + (long long)row0 * a.o_sn + h * a.o_sh;
        const int rows_valid = a.Nq - row0;
        const int r8 = lane_e >> 3, ch = lane_e & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 8 + r8;
            const u16x8_t v = *reinterpret_cast<const u16x8_t*>(ol + row * ATT_O_PITCH + ch * 16);
            if (row < rows_valid) *reinterpret_cast<u16x8_t*>(orow0 + (long long)row * a.o_sn + ch * 8) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}
#endif  // PGN_TUNING_HOOKS (pgn_attn_x16_kernel)

#ifdef PGN_TUNING_HOOKS
int g_attn_variant = 0;  // 20 = the ring kernel on 16x16x32 MFMAs, 0 = shipped heuristic, 1 = 4 waves x 32 rows, 3 = 4 waves x 64 rows (two LDS-DMA buffers), 6 = 64 rows with the three-buffer ring and a row maximum per tile, 7 = the shipped ring kernel forced, 8 / 9 / 10 = its earlier forms (packed fp32 softmax arithmetic; plain arithmetic in the old order; the same without the first-tile branch), 16 = 8 waves x 32 rows
#else
constexpr int g_attn_variant = 0;
#endif

}  // namespace

#ifdef PGN_TUNING_HOOKS
// Only in the measurement build (make tuning -> libpigeon_hip_tuning.so, loaded by scripts/bench_kernels.py): geometry
// override for A/B runs in one process.  The shipped library has no such export and no mutable global state.
extern "C" void pgn_attention_tuning(int variant) { g_attn_variant = variant; }
static unsigned long long* g_attn_dbg = nullptr;
// device buffer of 6 x uint64 per workgroup of the next ring-kernel launches (nullptr: off)
extern "C" void pgn_attention_debug_buffer(void* p) { g_attn_dbg = (unsigned long long*)p; }
#endif

// Which kernel class pgn_attention_fwd launches for a shape (host arithmetic only): the rule lives here once.
static bool att_takes_ring_kernel(int B, int H, int Nq, int Nk, int D, bool has_bias) {
    const long long wg256 = (long long)B * H * ((Nq + 255) / 256), wg128 = (long long)B * H * ((Nq + 127) / 128);
    auto fill = [](long long n, long long slots) { return (double)n / (double)(((n + slots - 1) / slots) * slots); };
    const bool ring_fills = 1.18 * fill(wg256, 512) > fill(wg128, 768);
    return D == 64 && !has_bias && Nq >= 512 && (wg256 >= 1024 || ring_fills) && Nk >= 256;
}

extern "C" int pgn_attention_kernel_class(int B, int H, int Nq, int Nk, int D, int has_key_bias) {
    if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0 || D <= 0) return PGN_ERR_INVALID;
    if (D > 256 || (D & 7) != 0) return PGN_ERR_UNSUPPORTED;
    return att_takes_ring_kernel(B, H, Nq, Nk, D, has_key_bias != 0) ? PGN_ATT_KERNEL_RING64 : PGN_ATT_KERNEL_ROWS32;
}

extern "C" int pgn_attention_fwd(const void* q, const void* k, const void* v, void* out, int dtype, int B, int H,
                                 int Nq, int Nk, int D, int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb,
                                 int64_t k_sn, int64_t k_sh, int64_t v_sb, int64_t v_sn, int64_t v_sh, int64_t o_sb,
                                 int64_t o_sn, int64_t o_sh, const float* key_bias, int64_t bias_sb, float scale,
                                 void* stream) {
    if (q == nullptr || k == nullptr || v == nullptr || out == nullptr) return PGN_ERR_INVALID;
    if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0 || D <= 0) return PGN_ERR_INVALID;
    if (dtype != PGN_BF16 && dtype != PGN_F16) return PGN_ERR_UNSUPPORTED;
    if (D > 256 || (D & 7) != 0) return PGN_ERR_UNSUPPORTED;   // head dims 8..256 in steps of 8 (16-byte row chunks)
    // 16-byte vector loads of q/k/v rows, 8-byte stores of out rows
    const int64_t st[12] = {q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh, o_sb, o_sn, o_sh};
    for (int i = 0; i < 12; ++i)
        if (st[i] < 0 || (st[i] & 7) != 0) return PGN_ERR_INVALID;
    if (!pgn_aligned(q, 16) || !pgn_aligned(k, 16) || !pgn_aligned(v, 16) || !pgn_aligned(out, 16)) return PGN_ERR_INVALID;
    if (key_bias != nullptr && bias_sb < 0) return PGN_ERR_INVALID;
    // the kernel addresses a (batch, head) key / value slice with 32-bit byte offsets
    if (((int64_t)Nk + 2 * ATT_BN) * (k_sn > v_sn ? k_sn : v_sn) * 2 >= (int64_t)1 << 32) return PGN_ERR_UNSUPPORTED;
    // Head dim 64 without a key bias and with at least 16 key tiles (the UNet's self-attention: N = 4096 and 1024, and
    // everything at higher resolution): pgn_attn_w64_kernel, 4 waves x 64 rows with LDS-DMA staging - 1003 vs 922 TFLOP/s
    // at N = 4096, 852 vs 812 at N = 1024; on the 4-tile key loop of cross-attention (Nk = 256) its 256-row workgroups
    // have too little to amortise their prologue over (592-653 vs 600-660) and the 32-row kernel stays.
    // Otherwise: 4 waves x 32 rows, three workgroups per CU at head dim 64 (two at 128, one at 256): the waves of a
    // SIMD come from DIFFERENT workgroups, so one workgroup's barrier / prologue / epilogue hides behind the others'
    // main loops.  Measured on MI355X at every
    // SDXL shape (scripts/bench_kernels.py): 4 waves beat the 8-wave 256-row geometry by 4 % (self, N 4096) to
    // 30 % (cross, Nk 256); the 8-wave form is compiled into the measurement build only (PGN_TUNING_HOOKS).
    int nw = 4;
    // ... and only with at least 1024 of its 256-row workgroups (two rounds of the 512 resident ones): below that the
    // 128-row workgroups of the 32-row kernel fill the chip better (batch 1 / 2 / 4 runs, N = 4096: 716 vs 616
    // TFLOP/s at 320 workgroups, 835 vs 778 at 640, 871 vs 953 at 1280; profiles/r02_attention_small_batches.txt).
    // With full key tiles the 64-row kernel runs in its ring form, which also wins on the 4-tile key loop of
    // cross-attention (Nk = 256: 738 / 648 against 700 / 627 TFLOP/s, profiles/r02_attention_ring_ab.txt).
    // Below 1024 workgroups the ring form (full key tiles) still wins where ITS rounds fill the chip better than the
    // 32-row kernel's: efficiency of a launch = workgroups / (rounds x resident slots), 512 slots of 256 rows against
    // 768 of 128 rows, and the ring kernel is ~1.18x faster per flop once the chip is full (end of round 2: N = 4096
    // at 960 workgroups 985 vs 835 TFLOP/s, N = 1024 at 480 workgroups 771 vs 709, but at 640 workgroups 759 vs 776
    // and at 320 558 vs 627; profiles/r02_attention_small_batches_final.txt).
    bool w64 = att_takes_ring_kernel(B, H, Nq, Nk, D, key_bias != nullptr);
    if (g_attn_variant == 16 && D == 64) nw = 8;
    if (g_attn_variant == 1 || nw == 8) w64 = false;
    if (g_attn_variant == 3 || (g_attn_variant >= 6 && g_attn_variant <= 10) || g_attn_variant == 20)
        w64 = D == 64 && key_bias == nullptr;
#ifdef PGN_TUNING_HOOKS
    const bool x16 = g_attn_variant == 20;   // the ring kernel on 16x16x32 MFMAs (pgn_attn_x16_kernel)
#endif
    // full key tiles: the three-buffer ring with counted waits; a ragged last tile keeps the two-buffer kernel
    const bool ring = w64 && g_attn_variant != 3;                        // every shape the 64-row class takes
    const bool ragged = (Nk % ATT_BN) != 0;                              // ... a ragged key count in its masked-last-tile form
#ifdef PGN_TUNING_HOOKS
    const bool sumchk = g_attn_variant != 6;   // 6: the ring kernel with the row maximum on every tile
#endif
    const int bm = w64 ? 256 : nw * 32;
    const long long nqt = (Nq + bm - 1) / bm;
    const long long blocks = (long long)B * H * nqt;
    if (blocks > 0x7fffffffLL) return PGN_ERR_UNSUPPORTED;

    AttnArgs a;
    a.q = (const unsigned short*)q; a.k = (const unsigned short*)k; a.v = (const unsigned short*)v;
    a.o = (unsigned short*)out; a.bias = key_bias;
    a.H = H; a.Nq = Nq; a.Nk = Nk; a.nqt = (int)nqt; a.D = D;
    a.q_sb = q_sb; a.q_sn = q_sn; a.q_sh = q_sh; a.k_sb = k_sb; a.k_sn = k_sn; a.k_sh = k_sh;
    a.v_sb = v_sb; a.v_sn = v_sn; a.v_sh = v_sh; a.o_sb = o_sb; a.o_sn = o_sn; a.o_sh = o_sh; a.bias_sb = bias_sb;
    a.c = scale * 1.4426950408889634f;
#ifdef PGN_TUNING_HOOKS
    a.dbg = g_attn_dbg;
#endif
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)blocks), block(nw * 64);
#ifdef PGN_TUNING_HOOKS
#define PGN_ATT_LAUNCH_8W(DT, BIAS) \
    if (nw == 8) hipLaunchKernelGGL((pgn_attn_fwd_kernel<DT, BIAS, 8, 2, 64, false>), grid, block, 0, s, a); else
#define PGN_ATT_LAUNCH_W64_2BUF(DT) \
    else if (w64) hipLaunchKernelGGL((pgn_attn_w64_kernel<DT>), grid, block, 0, s, a);
#define PGN_ATT_LAUNCH_RING_MAX(DT) /* the earlier forms take full key tiles only */ \
    if (!ragged && x16) hipLaunchKernelGGL((pgn_attn_x16_kernel<DT, false>), grid, block, 0, s, a); else \
    if (!ragged && !sumchk) hipLaunchKernelGGL((pgn_attn_w64r_kernel<DT, false, true, 0>), grid, block, 0, s, a); else \
    if (!ragged && g_attn_variant == 8) hipLaunchKernelGGL((pgn_attn_w64r_kernel<DT, true, true, 0>), grid, block, 0, s, a); else \
    if (!ragged && g_attn_variant == 9) hipLaunchKernelGGL((pgn_attn_w64r_kernel<DT, true, false, 0>), grid, block, 0, s, a); else \
    if (!ragged && g_attn_variant == 10) hipLaunchKernelGGL((pgn_attn_w64r_kernel<DT, true, false, 1>), grid, block, 0, s, a); else
#else
#define PGN_ATT_LAUNCH_8W(DT, BIAS)
#define PGN_ATT_LAUNCH_RING_MAX(DT)
#define PGN_ATT_LAUNCH_W64_2BUF(DT)
#endif
#define PGN_ATT_LAUNCH(DT, BIAS)                                                                                  \
    do {                                                                                                          \
        if (ring) {                                                                                               \
            PGN_ATT_LAUNCH_RING_MAX(DT)                                                                           \
            if (ragged) hipLaunchKernelGGL((pgn_attn_w64r_kernel<DT, true, false, 2, true>), grid, block, 0, s, a); \
            else hipLaunchKernelGGL((pgn_attn_w64r_kernel<DT, true, false, 2>), grid, block, 0, s, a);            \
        }                                                                                                         \
        PGN_ATT_LAUNCH_W64_2BUF(DT)                                                                               \
        else if (D == 64) {                                                                                       \
            PGN_ATT_LAUNCH_8W(DT, BIAS)                                                                           \
            hipLaunchKernelGGL((pgn_attn_fwd_kernel<DT, BIAS, 4, 3, 64, false>), grid, block, 0, s, a);             \
        } else if (D < 64) hipLaunchKernelGGL((pgn_attn_fwd_kernel<DT, BIAS, 4, 3, 64, true>), grid, block, 0, s, a); \
        else if (D == 128) hipLaunchKernelGGL((pgn_attn_fwd_kernel<DT, BIAS, 4, 2, 128, false>), grid, block, 0, s, a); \
        else if (D < 128) hipLaunchKernelGGL((pgn_attn_fwd_kernel<DT, BIAS, 4, 2, 128, true>), grid, block, 0, s, a); \
        else if (D == 256) hipLaunchKernelGGL((pgn_attn_fwd_kernel<DT, BIAS, 4, 1, 256, false>), grid, block, 0, s, a); \
        else hipLaunchKernelGGL((pgn_attn_fwd_kernel<DT, BIAS, 4, 1, 256, true>), grid, block, 0, s, a);           \
    } while (0)
    if (dtype == PGN_BF16) {
        if (key_bias != nullptr) PGN_ATT_LAUNCH(PGN_BF16, true);
        else PGN_ATT_LAUNCH(PGN_BF16, false);
    } else {
        if (key_bias != nullptr) PGN_ATT_LAUNCH(PGN_F16, true);
        else PGN_ATT_LAUNCH(PGN_F16, false);
    }
#undef PGN_ATT_LAUNCH
#undef PGN_ATT_LAUNCH_8W
#undef PGN_ATT_LAUNCH_RING_MAX
#undef PGN_ATT_LAUNCH_W64_2BUF
    PGN_LAUNCH_CHECK();
    return PGN_OK;
}

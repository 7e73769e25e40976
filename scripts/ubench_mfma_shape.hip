// v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16 under the attention ring kernel's VALU mix, on RANDOM
// operands (the clock the chip holds depends on the data and on the MFMA shape: MI355X_MICROARCH.md, DVFS give-back
// items 1 and 7; a constant-operand loop ranks the shapes by cycles only).
// One loop iteration models one 64-key tile of one wave of pgn_attn_w64r_kernel: 1024 matrix-pipe cycles of MFMA work
// (32 x 32x32x16 on four f32x16 accumulators, or 64 x 16x16x32 on sixteen f32x4 accumulators: the same 64 accumulator
// registers, the same FLOP) and the shipped softmax arithmetic of 64 scores per lane in plain fp32 instructions:
// 64 v_fma_f32 + 64 v_exp_f32 + 48 v_add_f32 + 32 v_cvt_pk_bf16_f32 = 208 VALU instructions.
//   VMODE 0  no VALU work            (the MFMA roof of each shape at the clock it holds)
//   VMODE 1  VALU spread evenly between the MFMAs (13 per 2 / per 4 MFMAs)
//   VMODE 2  all MFMAs, then all VALU (what one wave's dependencies give by themselves)
//   VMODE 3  half: first quarter of the MFMAs bare, the VALU work under the middle half, last quarter bare
// hipcc -O3 --offload-arch=gfx950 scripts/ubench_mfma_shape.hip -o scripts/ubench_mfma_shape && ./scripts/ubench_mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define MFMA32(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B));
#define MFMA16(ACC, A, B) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B));

// 13 VALU instructions = the softmax of four scores: x = s * c - m; p = exp2(x); l += p; u = pack(p, p')
#define VG_A(j)                                                                                  \
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[0]) : "v"(sc[(j) & 7]), "v"(c), "v"(nm));   \
    asm volatile("v_exp_f32 %0, %1" : "=v"(e[0]) : "v"(x[0]));                                   \
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(l[0]) : "v"(e[2]));
#define VG_B(j)                                                                                  \
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[1]) : "v"(sc[((j) + 1) & 7]), "v"(c), "v"(nm)); \
    asm volatile("v_exp_f32 %0, %1" : "=v"(e[1]) : "v"(x[1]));                                   \
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[0]) : "v"(e[2]), "v"(e[3]));
#define VG_C(j)                                                                                  \
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[2]) : "v"(sc[((j) + 2) & 7]), "v"(c), "v"(nm)); \
    asm volatile("v_exp_f32 %0, %1" : "=v"(e[2]) : "v"(x[2]));                                   \
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(l[1]) : "v"(e[0]));
#define VG_D(j)                                                                                  \
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[3]) : "v"(sc[((j) + 3) & 7]), "v"(c), "v"(nm)); \
    asm volatile("v_exp_f32 %0, %1" : "=v"(e[3]) : "v"(x[3]));                                   \
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(l[2]) : "v"(e[1]));                               \
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[1]) : "v"(e[0]), "v"(e[1]));
#define VG13(j) VG_A(j) VG_B(j) VG_C(j) VG_D(j)

__device__ __forceinline__ unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }

template <int SHAPE, int VMODE>
__global__ __launch_bounds__(256, 2) void k(float* out, long long* cyc, int iters, float c, float nm) {
    unsigned seed = 0x9e3779b9u * (threadIdx.x + 1) + 0x85ebca6bu * (blockIdx.x + 1);
    bf16x8_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a[i][j] = (__bf16)(((int)(lcg(seed) >> 8) & 0xffff) * (2.0f / 65536.0f) - 1.0f);
            b[i][j] = (__bf16)(((int)(lcg(seed) >> 8) & 0xffff) * (2.0f / 65536.0f) - 1.0f);
        }
    float sc[8], x[4], e[4] = {0.f, 0.f, 0.f, 0.f}, l[3] = {0.f, 0.f, 0.f};
    unsigned u[2] = {0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) sc[i] = ((int)(lcg(seed) >> 8) & 0xffff) * (8.0f / 65536.0f) - 4.0f;
    f32x16_t w0, w1, w2, w3;
    f32x4_t n[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { w0[i] = 0.f; w1[i] = 0.f; w2[i] = 0.f; w3[i] = 0.f; n[i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (SHAPE == 0) {
            // 8 groups of 4 MFMAs; VALU: 16 groups of 13
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const bool under = VMODE == 1 || (VMODE == 3 && g >= 2 && g < 6);
                MFMA32(w0, a[g & 3], b[0]) if (under && VMODE == 1) { VG_A(g) VG_B(g) } if (under && VMODE == 3) { VG13(g) }
                MFMA32(w1, a[g & 3], b[1]) if (under && VMODE == 1) { VG_C(g) VG_D(g) } if (under && VMODE == 3) { VG13(g + 1) }
                MFMA32(w2, a[g & 3], b[2]) if (under && VMODE == 1) { VG_A(g + 4) VG_B(g + 4) } if (under && VMODE == 3) { VG13(g + 2) }
                MFMA32(w3, a[g & 3], b[3]) if (under && VMODE == 1) { VG_C(g + 4) VG_D(g + 4) } if (under && VMODE == 3) { VG13(g + 3) }
            }
        } else {
            // 16 groups of 4 MFMAs
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const bool under = VMODE == 1 || (VMODE == 3 && g >= 4 && g < 12);
                MFMA16(n[(4 * g) & 15], a[g & 3], b[0]) if (under && VMODE == 1) { VG_A(g) } if (under && VMODE == 3) { VG_A(g) VG_B(g) }
                MFMA16(n[(4 * g + 1) & 15], a[g & 3], b[1]) if (under && VMODE == 1) { VG_B(g) } if (under && VMODE == 3) { VG_C(g) VG_D(g) }
                MFMA16(n[(4 * g + 2) & 15], a[g & 3], b[2]) if (under && VMODE == 1) { VG_C(g) } if (under && VMODE == 3) { VG_A(g + 4) VG_B(g + 4) }
                MFMA16(n[(4 * g + 3) & 15], a[g & 3], b[3]) if (under && VMODE == 1) { VG_D(g) } if (under && VMODE == 3) { VG_C(g + 4) VG_D(g + 4) }
            }
        }
        if constexpr (VMODE == 2) {
#pragma unroll
            for (int g = 0; g < 16; ++g) { VG13(g) }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = l[0] + l[1] + l[2] + __uint_as_float(u[0]) + __uint_as_float(u[1]);
#pragma unroll
    for (int i = 0; i < 16; ++i) s += w0[i] + w1[i] + w2[i] + w3[i] + n[i][0] + n[i][1] + n[i][2] + n[i][3];
    if (s == 12345.678f) out[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

struct Res { double ticks, ms, tf; };

template <int SHAPE, int VMODE>
Res run_once(int blocks_per_cu, float* out, long long* cyc) {
    const int iters = 4000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, VMODE>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 0.18f, -0.7f);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    Res r;
    r.ticks = (double)h[h.size() / 2] / iters;
    r.ms = ms;
    r.tf = 32.0 * 32768.0 * iters * blocks * 4 / (ms * 1e-3) / 1e12;
    return r;
}

template <int VMODE>
void ab(const char* name, int w, float* out, long long* cyc) {
    // interleaved rounds in one process (guide rule 24); medians
    std::vector<Res> r0, r1;
    run_once<0, VMODE>(w, out, cyc); run_once<1, VMODE>(w, out, cyc);
    for (int rnd = 0; rnd < 5; ++rnd) { r0.push_back(run_once<0, VMODE>(w, out, cyc)); r1.push_back(run_once<1, VMODE>(w, out, cyc)); }
    auto med = [](std::vector<Res>& v, auto f) { std::vector<double> x; for (auto& r : v) x.push_back(f(r)); std::sort(x.begin(), x.end()); return x[x.size() / 2]; };
    const double t0 = med(r0, [](Res& r) { return r.ticks; }), t1 = med(r1, [](Res& r) { return r.ticks; });
    const double m0 = med(r0, [](Res& r) { return r.ms; }), m1 = med(r1, [](Res& r) { return r.ms; });
    const double f0 = med(r0, [](Res& r) { return r.tf; }), f1 = med(r1, [](Res& r) { return r.tf; });
    // clock = ticks * iters / wall  (s_memtime ticks are shader cycles)
    printf("%-12s waves/SIMD %d | 32x32x16: %7.1f cyc/tile %7.3f ms %7.1f TF (%.2f GHz) | 16x16x32: %7.1f cyc/tile %7.3f ms %7.1f TF (%.2f GHz) | wall ratio %.3f, cycle ratio %.3f\n",
           name, w, t0, m0, f0, t0 * 4000 / (m0 * 1e-3) / 1e9, t1, m1, f1, t1 * 4000 / (m1 * 1e-3) / 1e9, f1 / f0, t0 / t1);
    fflush(stdout);
}

int main() {
    float* out;
    long long* cyc;
    (void)hipMalloc(&out, 4);
    (void)hipMalloc(&cyc, sizeof(long long) * 512 * 4);
    for (int w = 1; w <= 2; ++w) {
        ab<0>("no VALU", w, out, cyc);
        ab<1>("interleaved", w, out, cyc);
        ab<3>("half", w, out, cyc);
        ab<2>("phased", w, out, cyc);
    }
    (void)hipFree(out); (void)hipFree(cyc);
    return 0;
}

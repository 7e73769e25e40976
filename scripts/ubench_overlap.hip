// Does interleaving one wave's MFMAs with its own VALU work buy anything on gfx950 when two waves share a SIMD?
// One loop iteration models one key tile of the 64-row attention kernel per wave: 32 v_mfma_f32_32x32x16_bf16 on four
// independent accumulators and 192 VALU instructions in the kernel's mix (per MFMA: 2 v_exp_f32, 1 v_pk_fma_f32,
// 1 v_pk_add_f32, 1 v_cvt_pk_bf16_f32, 1 v_max3_f32).  Three orders of the SAME instructions:
//   phased       32 MFMA, then 192 VALU                      (what the kernel's dependencies give by themselves)
//   half         8 MFMA | 16 x (MFMA + 12 VALU) | 8 MFMA     (softmax of one key half under the MFMAs of the other)
//   interleaved  32 x (MFMA + 6 VALU)                        (a full software pipeline)
// hipcc -O3 --offload-arch=gfx950 scripts/ubench_overlap.hip -o scripts/ubench_overlap && ./scripts/ubench_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

#define MFMA(ACC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b));
// six VALU instructions of the attention mix on rotating registers (no instruction depends on the one before it)
#define VALU6(i0, i1, i2, i3)                                                                       \
    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i0]) : "v"(p[i1]), "v"(p[i2]));           \
    asm volatile("v_exp_f32 %0, %0" : "+v"(r[2 * i3]));                                             \
    asm volatile("v_exp_f32 %0, %0" : "+v"(r[2 * i3 + 1]));                                         \
    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i1]) : "v"(p[i2]));                             \
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i0]) : "v"(r[2 * i2]), "v"(r[2 * i2 + 1])); \
    asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m[i0]) : "v"(r[2 * i1]), "v"(r[2 * i1 + 1]));

// one VALU instruction of type T on rotating registers
template <int T>
__device__ __forceinline__ void one(f32x2_t (&p)[4], float* r, unsigned (&u)[4], float (&m)[4], int i) {
    const int i0 = i & 3, i1 = (i + 1) & 3, i2 = (i + 2) & 3;
    if constexpr (T == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i & 7]));
    else if constexpr (T == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i0]) : "v"(p[i1]), "v"(p[i2]));
    else if constexpr (T == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i0]) : "v"(p[i1]));
    else if constexpr (T == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i0]) : "v"(r[i & 7]), "v"(r[(i + 1) & 7]));
    else if constexpr (T == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m[i0]) : "v"(r[i & 7]), "v"(r[(i + 1) & 7]));
    else if constexpr (T == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i & 7]) : "v"(r[(i + 1) & 7]), "v"(r[(i + 2) & 7]));
    else if constexpr (T == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i & 7]) : "v"(r[(i + 1) & 7]));
    else if constexpr (T == 7) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[i & 7]) : "v"(r[(i + 1) & 7]));
    else if constexpr (T == 8) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i0]) : "v"(r[i & 7]));
    else if constexpr (T == 9) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i0]) : "v"(p[i1]));
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, long long* cyc, int iters) {
    f32x2_t p[4];
    float* r = reinterpret_cast<float*>(p);   // the eight floats of p[0..3]
    unsigned u[4];
    float m[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { p[i] = f32x2_t{0.5f + 0.001f * threadIdx.x, 0.25f}; m[i] = 0.f; u[i] = 0; }
    f32x16_t acc0, acc1, acc2, acc3;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; acc2[i] = 0.f; acc3[i] = 0.f; }
    bf16x8_t a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (threadIdx.x & 7)); b[i] = (__bf16)(0.02f * i); }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int g = 0; g < 8; ++g) { MFMA(acc0) MFMA(acc1) MFMA(acc2) MFMA(acc3) }
#pragma unroll
            for (int g = 0; g < 8; ++g) { VALU6(0, 1, 2, 3) VALU6(1, 2, 3, 0) VALU6(2, 3, 0, 1) VALU6(3, 0, 1, 2) }
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int g = 0; g < 2; ++g) { MFMA(acc0) MFMA(acc1) MFMA(acc2) MFMA(acc3) }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                MFMA(acc0) VALU6(0, 1, 2, 3) VALU6(1, 2, 3, 0)
                MFMA(acc1) VALU6(2, 3, 0, 1) VALU6(3, 0, 1, 2)
                MFMA(acc2) VALU6(0, 1, 2, 3) VALU6(1, 2, 3, 0)
                MFMA(acc3) VALU6(2, 3, 0, 1) VALU6(3, 0, 1, 2)
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) { MFMA(acc0) MFMA(acc1) MFMA(acc2) MFMA(acc3) }
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                MFMA(acc0) VALU6(0, 1, 2, 3)
                MFMA(acc1) VALU6(1, 2, 3, 0)
                MFMA(acc2) VALU6(2, 3, 0, 1)
                MFMA(acc3) VALU6(3, 0, 1, 2)
            }
        } else if constexpr (MODE >= 300) {   // 32 x (MFMA + 6 instructions of one type)
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                MFMA(acc0)
#pragma unroll
                for (int j = 0; j < 6; ++j) one<MODE - 300>(p, r, u, m, 24 * g + j);
                MFMA(acc1)
#pragma unroll
                for (int j = 0; j < 6; ++j) one<MODE - 300>(p, r, u, m, 24 * g + 6 + j);
                MFMA(acc2)
#pragma unroll
                for (int j = 0; j < 6; ++j) one<MODE - 300>(p, r, u, m, 24 * g + 12 + j);
                MFMA(acc3)
#pragma unroll
                for (int j = 0; j < 6; ++j) one<MODE - 300>(p, r, u, m, 24 * g + 18 + j);
            }
        } else if constexpr (MODE >= 200) {   // 32 MFMA, then 192 instructions of one type
#pragma unroll
            for (int g = 0; g < 8; ++g) { MFMA(acc0) MFMA(acc1) MFMA(acc2) MFMA(acc3) }
#pragma unroll
            for (int j = 0; j < 192; ++j) one<MODE - 200>(p, r, u, m, j);
        } else {   // MODE = 100 + G: phased, 32 MFMA then G groups of 24 VALU (how the time scales with the VALU count)
#pragma unroll
            for (int g = 0; g < 8; ++g) { MFMA(acc0) MFMA(acc1) MFMA(acc2) MFMA(acc3) }
#pragma unroll
            for (int g = 0; g < MODE - 100; ++g) { VALU6(0, 1, 2, 3) VALU6(1, 2, 3, 0) VALU6(2, 3, 0, 1) VALU6(3, 0, 1, 2) }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += p[i][0] + p[i][1] + m[i] + __uint_as_float(u[i]);
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    if (s == 12345.678f) out[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char* name, int blocks_per_cu) {
    const int iters = 2000, blocks = 256 * blocks_per_cu;
    float* out;
    long long* cyc;
    (void)hipMalloc(&out, 4);
    (void)hipMalloc(&cyc, sizeof(long long) * blocks * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2] / iters;
    // 32 MFMAs of 32768 FLOP per wave and iteration
    const double tf = 32.0 * 32768.0 * iters * blocks * 4 / (ms * 1e-3) / 1e12;
    printf("%-12s waves/SIMD %d: %8.1f clock ticks per tile and wave (s_memtime), %7.3f ms wall, %7.1f TFLOP/s of MFMA work\n",
           name, blocks_per_cu, med, ms, tf);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run<0>("phased", w);
        run<1>("half", w);
        run<2>("interleaved", w);
        run<100>("0 VALU", w);
        run<104>("96 VALU", w);
        run<106>("144 VALU", w);
        run<108>("192 VALU", w);
        run<110>("240 VALU", w);
        run<112>("288 VALU", w);
        run<200>("exp ph", w);      run<300>("exp il", w);
        run<201>("pk_fma ph", w);   run<301>("pk_fma il", w);
        run<202>("pk_add ph", w);   run<302>("pk_add il", w);
        run<209>("pk_mul ph", w);   run<309>("pk_mul il", w);
        run<203>("cvt_pk ph", w);   run<303>("cvt_pk il", w);
        run<204>("max3 ph", w);     run<304>("max3 il", w);
        run<205>("fma ph", w);      run<305>("fma il", w);
        run<206>("add ph", w);      run<306>("add il", w);
        run<207>("max ph", w);      run<307>("max il", w);
        run<208>("mov ph", w);      run<308>("mov il", w);
    }
    return 0;
}

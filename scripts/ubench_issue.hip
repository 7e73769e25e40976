// Issue-rate micro-benchmarks for gfx950 (MI355X): how many cycles a SIMD spends per VALU / transcendental / packed /
// MFMA instruction when 1, 2 or 3 wavefronts share it, and whether one wave's MFMAs overlap another wave's VALU work.
// Numbers feed the attention kernel's design notes.
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench_issue.hip -o scripts/ubench_issue && ./scripts/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

enum { OP_FMA = 0, OP_EXP = 1, OP_PKFMA = 2, OP_CVT = 3, OP_MAX3 = 4, OP_MFMA = 5, OP_MIX_SPLIT = 6, OP_MIX_SAME = 7, OP_PKMUL = 8, OP_ADD = 9 };

#define REP8(x) x x x x x x x x

template <int OP>
__device__ __forceinline__ void body(float (&r)[8], f32x16_t& acc, bf16x8_t a, bf16x8_t b, int role) {
    if constexpr (OP == OP_FMA) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 1) & 7]), "v"(r[(i + 2) & 7]));
    } else if constexpr (OP == OP_ADD) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
    } else if constexpr (OP == OP_EXP) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
    } else if constexpr (OP == OP_PKFMA) {
        f32x2_t* p = reinterpret_cast<f32x2_t*>(r);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 3]), "v"(p[(i + 2) & 3]));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 3]), "v"(p[(i + 3) & 3]));
        }
    } else if constexpr (OP == OP_PKMUL) {
        f32x2_t* p = reinterpret_cast<f32x2_t*>(r);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 3]));
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 2) & 3]));
        }
    } else if constexpr (OP == OP_CVT) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            unsigned u;
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u) : "v"(r[i]), "v"(r[(i + 1) & 7]));
            r[i] = __uint_as_float(u | 0x3f800000u);
        }
    } else if constexpr (OP == OP_MAX3) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 1) & 7]), "v"(r[(i + 2) & 7]));
    } else if constexpr (OP == OP_MFMA) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    } else if constexpr (OP == OP_MIX_SPLIT) {
        // role 0: MFMA only; role 1: exp only (different wavefronts of one SIMD)
        if (role == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
        }
    } else if constexpr (OP == OP_MIX_SAME) {
        // every wave: 1 MFMA + 4 exp interleaved in its own stream
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            asm volatile("v_exp_f32 %0, %0" : "+v"(r[0]));
            asm volatile("v_exp_f32 %0, %0" : "+v"(r[1]));
            asm volatile("v_exp_f32 %0, %0" : "+v"(r[2]));
            asm volatile("v_exp_f32 %0, %0" : "+v"(r[3]));
        }
    }
}

// one block of 256 * W threads per CU (W waves per SIMD; 100 KiB of LDS keeps a second block off the CU)
template <int OP>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters, int roles) {
    extern __shared__ float lds[];
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = 0.5f + 0.001f * (threadIdx.x + i);
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    bf16x8_t a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (threadIdx.x & 7)); b[i] = (__bf16)(0.02f * i); }
    const int role = roles > 1 ? (int)((threadIdx.x >> 8) % roles) : 0;   // waves 0-3 role 0, 4-7 role 1, ...
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) body<OP>(r, acc, a, b, role);
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += r[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 12345.678f) out[0] = s + lds[0];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(const char* name, int waves_per_simd, int instr_per_body, int roles = 1) {
    const int iters = 4000;
    const int blocks = 256;
    const size_t lds = 100 * 1024;
    float* out;
    long long* cyc;
    hipMalloc(&out, 4);
    hipMalloc(&cyc, sizeof(long long) * blocks * 16);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256 * waves_per_simd), lds, 0, out, cyc, iters, roles);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 16);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * 16, hipMemcpyDeviceToHost);
    for (int role = 0; role < roles; ++role) {
        std::vector<long long> v;
        for (int b = 0; b < blocks; ++b)
            for (int w = 0; w < 4 * waves_per_simd; ++w)
                if (roles == 1 || (w / 4) % roles == role) v.push_back(h[b * 16 + w]);
        std::sort(v.begin(), v.end());
        const double med = (double)v[v.size() / 2];
        printf("%-34s waves/SIMD %d role %d: %8.2f cycles per body (%d instr) = %6.2f per instr per wave, %6.2f SIMD cycles per instr\n",
               name, waves_per_simd, role, med / iters, instr_per_body, med / iters / instr_per_body,
               med / iters / instr_per_body / waves_per_simd * (roles > 1 ? roles : 1));
    }
    hipFree(out);
    hipFree(cyc);
}

int main() {
    for (int w = 1; w <= 4; ++w) {
        run<OP_FMA>("v_fma_f32", w, 8);
        run<OP_ADD>("v_add_f32", w, 8);
        run<OP_MAX3>("v_max3_f32", w, 8);
        run<OP_EXP>("v_exp_f32", w, 8);
        run<OP_PKFMA>("v_pk_fma_f32", w, 8);
        run<OP_PKMUL>("v_pk_mul_f32", w, 8);
        run<OP_CVT>("v_cvt_pk_bf16_f32 (+v_or)", w, 16);
        run<OP_MFMA>("v_mfma_f32_32x32x16_bf16", w, 8);
        run<OP_MIX_SAME>("1 mfma + 4 exp, same wave", w, 8);
    }
    run<OP_MIX_SPLIT>("8 mfma | 32 exp, split waves", 2, 8, 2);
    run<OP_MIX_SPLIT>("8 mfma | 32 exp, split waves", 4, 8, 2);
    return 0;
}

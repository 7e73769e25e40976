// Shared device/host helpers for libpigeon_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pigeon_hip.h"

#define PGN_WAVE 64

#define PGN_LAUNCH_CHECK()                          \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// 16-bit storage <-> fp32.  Casts through __bf16 / _Float16 lower to v_cvt_* (RNE, NaN-preserving).
template <int DT>
struct Cvt;
template <>
struct Cvt<PGN_BF16> {
    typedef unsigned short storage;
    static __device__ __forceinline__ float to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
    static __device__ __forceinline__ unsigned short from_f32(float f) {
        __bf16 b = (__bf16)f;
        return __builtin_bit_cast(unsigned short, b);
    }
};
template <>
struct Cvt<PGN_F16> {
    typedef unsigned short storage;
    static __device__ __forceinline__ float to_f32(unsigned short h) {
        return (float)__builtin_bit_cast(_Float16, h);
    }
    static __device__ __forceinline__ unsigned short from_f32(float f) {
        _Float16 b = (_Float16)f;
        return __builtin_bit_cast(unsigned short, b);
    }
};
template <>
struct Cvt<PGN_F32> {
    typedef float storage;
    static __device__ __forceinline__ float to_f32(float h) { return h; }
    static __device__ __forceinline__ float from_f32(float f) { return f; }
};

static inline bool pgn_aligned(const void* p, size_t a) { return (((uintptr_t)p) & (a - 1)) == 0; }
static inline int pgn_dtype_size(int dt) { return dt == PGN_F32 ? 4 : ((dt == PGN_BF16 || dt == PGN_F16) ? 2 : 0); }

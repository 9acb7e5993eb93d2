// elem_types.h — element types and wave64 reductions shared by the gfx950 kernels of librnnt_mi355.so.
#pragma once
#include <hip/hip_runtime.h>

namespace rnnt_mi355_detail {

constexpr int kWave = 64;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------
// element types: a "chunk" is 16 bytes of a logit row = EPC elements
// ---------------------------------------------------------------------------------------------
struct F32 {
    typedef float elem;
    static constexpr int EPC = 4;
    static __device__ __forceinline__ void unpack(const u32x4 &raw, float (&x)[EPC]) {
        x[0] = __uint_as_float(raw.x);
        x[1] = __uint_as_float(raw.y);
        x[2] = __uint_as_float(raw.z);
        x[3] = __uint_as_float(raw.w);
    }
    static __device__ __forceinline__ u32x4 pack(const float (&x)[EPC]) {
        u32x4 r;
        r.x = __float_as_uint(x[0]);
        r.y = __float_as_uint(x[1]);
        r.z = __float_as_uint(x[2]);
        r.w = __float_as_uint(x[3]);
        return r;
    }
    static __device__ __forceinline__ float to_float(elem v) { return v; }
    static __device__ __forceinline__ elem from_float(float v) { return v; }
};

struct BF16 {
    typedef __bf16 elem;
    static constexpr int EPC = 8;
    static __device__ __forceinline__ void unpack(const u32x4 &raw, float (&x)[EPC]) {
        const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x[2 * i] = __uint_as_float(w[i] << 16);
            x[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ unsigned pack2(float lo, float hi) {
        // one v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN stays NaN) for the pair
        const f32x2 v = {lo, hi};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    }
    static __device__ __forceinline__ u32x4 pack(const float (&x)[EPC]) {
        u32x4 r;
        r.x = pack2(x[0], x[1]);
        r.y = pack2(x[2], x[3]);
        r.z = pack2(x[4], x[5]);
        r.w = pack2(x[6], x[7]);
        return r;
    }
    static __device__ __forceinline__ float to_float(elem v) { return (float)v; }
    static __device__ __forceinline__ elem from_float(float v) { return (__bf16)v; }
};

struct F16 {
    typedef _Float16 elem;
    static constexpr int EPC = 8;
    static __device__ __forceinline__ void unpack(const u32x4 &raw, float (&x)[EPC]) {
        const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x[2 * i] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w[i] & 0xffffu));
            x[2 * i + 1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w[i] >> 16));
        }
    }
    static __device__ __forceinline__ unsigned pack2(float lo, float hi) {
        const f32x2 v = {lo, hi};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
    }
    static __device__ __forceinline__ u32x4 pack(const float (&x)[EPC]) {
        u32x4 r;
        r.x = pack2(x[0], x[1]);
        r.y = pack2(x[2], x[3]);
        r.z = pack2(x[4], x[5]);
        r.w = pack2(x[6], x[7]);
        return r;
    }
    static __device__ __forceinline__ float to_float(elem v) { return (float)v; }
    static __device__ __forceinline__ elem from_float(float v) { return (_Float16)v; }
};

// ---------------------------------------------------------------------------------------------
// wave64 reductions (butterfly; every lane ends with the result)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, kWave));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

}  // namespace rnnt_mi355_detail

// mma.h — matrix-core and LDS-DMA primitives shared by the MFMA kernels of librnnt_mi355.so
// (fused_mi355.hip: projection + K1; fused_bwd_mi355.hip: K3 + the projection's two backward products).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "elem_types.h"

namespace rnnt_mi355_detail {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// v_mfma_f32_32x32x16_{bf16,f16}: lane l = (r = l & 31, h = l >> 5) holds A[row r][k = 8h + j] and
// B[k = 8h + j][col r] in element j of its fragment; D: column l & 31, row (i & 3) + 8 (i >> 2) + 4 h in register i.
template <typename DT>
struct Mma;
template <>
struct Mma<BF16> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <>
struct Mma<F16> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

// LDS-DMA through inline asm: the compiler's wait-count pass, seeing an LDS-DMA of its own in flight,
// puts `s_waitcnt vmcnt(0)` in front of the next LDS read (it cannot tell the buffers apart), which
// would wait for the tile being prefetched at the top of every tile.  These DMAs are invisible to it; the
// kernels order them themselves with `s_waitcnt vmcnt(N)` + `s_barrier`.
// M0 carries the LDS byte address of lane 0's 16 bytes; the hardware adds lane * 16.  M0 is reserved by the
// compiler and not preserved around a statement: it is written in the statement that reads it and restored.
__device__ __forceinline__ void lds_dma16(unsigned voff, const void *sbase, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}
// the same for bytes that this CU reads once and no other does: streaming ("nt") policy, they do not displace
// what the other CUs of the XCD keep re-reading from its L2
__device__ __forceinline__ void lds_dma16_nt(unsigned voff, const void *sbase, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}

// Two ds_read_b64_tr_b16 = one 32x32x16 fragment whose k index is the ROW index of a row-major LDS image:
// within a group of 16 lanes, lane 4q + p passes the address of row q, columns 4p .. 4p+3 of a 4 x 16 block and
// lane i receives column i of the four rows.  `addr` addresses rows k0 + 8h + q of the image, `step4` is the
// byte distance of four rows: elements 0-3 are rows 8h .. 8h+3, elements 4-7 rows 8h+4 .. 8h+7.
// EXEC must be all ones (the gather crosses lanes).
template <typename FRAG>
__device__ __forceinline__ FRAG lds_read_tr_frag(unsigned addr, unsigned step4) {
    typedef __attribute__((address_space(3))) s16x4 *lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)addr);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)(addr + step4));
    return __builtin_bit_cast(FRAG, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

}  // namespace rnnt_mi355_detail

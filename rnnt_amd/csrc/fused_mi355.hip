// fused_mi355.hip — the joint network's output projection fused with K1 of the loss.
//
// Replaces, for 16-bit operands, the pair
//     logits = Linear(H -> V)(hidden)                 /root/reference/src/rnnt/model.py:17,59
//     K1 of RNNTLoss(logits, ...)                     /root/reference/src/rnnt/model.py:116
// by ONE kernel: an MFMA GEMM whose epilogue is K1.  The logits are still written once (K3 reads them
// in backward) but never read back by a log-softmax pass: max and sum-exp of every lattice node's row
// are carried in registers across the vocabulary tiles.
//
// Shape of the kernel (gfx950, wave64):
//   * a workgroup = 8 waves owns 256 lattice nodes and walks the WHOLE vocabulary in tiles of 64;
//     a wave owns 32 nodes: their hidden vectors (32 x H 16-bit values) stay in 128 VGPRs as the MFMA
//     B operand for the lifetime of the workgroup.
//   * the weight tile [64 vocab x H] is the A operand: staged by global_load_lds (LDS-DMA, 16 bytes per
//     lane) into one of two LDS buffers while the previous tile is being multiplied, read with
//     ds_read_b128; 16 bytes of padding per KiB keep those reads off each other's banks (TileBuf).
//     W is [V][H] row-major, exactly nn.Linear's weight: both operands are K-contiguous.
//   * D = W_tile . hidden^T ("swapped" orientation): v_mfma_f32_32x32x16 leaves lane l with 16 vocabulary
//     entries of ONE node (column l & 31), so the online {max, sum-exp} is lane-local; the two lanes
//     l, l+32 that share a node are merged once at the end.
//   * the bias initialises the accumulators (no add in the epilogue); logits are rounded to the 16-bit
//     type, the softmax statistics are taken from the ROUNDED values (K3 recomputes the softmax from what
//     is stored, so rows of the gradient sum to zero exactly as in the unfused path), and the wave's
//     32 x 32 tile is transposed through a private LDS buffer so that four consecutive lanes store the 64
//     contiguous bytes of one row.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/rnnt_mi355.h"
#include "elem_types.h"
#include "lattice.h"

namespace {

using namespace rnnt_mi355_detail;

constexpr float kLog2e = 1.4426950408889634f;

#ifndef FUSED_MMA_PRIO
#define FUSED_MMA_PRIO 1
#endif
#ifndef FUSED_EPI_PRIO
#define FUSED_EPI_PRIO 0
#endif
#ifndef FUSED_PF
#define FUSED_PF 4
#endif
#ifndef FUSED_NO_VSPLIT
#define FUSED_NO_VSPLIT 0  // 1: developer A/B build without the shared surplus row tiles (tools/build_fused_variants.sh)
#endif
#ifndef FUSED_STORE_AUX
#define FUSED_STORE_AUX 0  // cache policy of the logits stores: 16 = sc1, 2 = nt, 0 = default
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

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

#ifdef FUSED_ABLATE_STAMP  // developer build: per-phase s_memtime stamps of workgroup 0 into a buffer set from the host
__device__ unsigned long long *g_stamp_buf;
#define STAMP(slot)                                                                                     \
    do {                                                                                                \
        if (blockIdx.x == gridDim.x / 2 && (threadIdx.x & 63) == 0 && n < 64)                            \
            g_stamp_buf[((n * 8 + (threadIdx.x >> 6)) * 4) + (slot)] = __builtin_amdgcn_s_memtime();       \
    } while (0)
#else
#define STAMP(slot)
#endif

constexpr int kWaves = 8;             // waves per workgroup
constexpr int kRowsPerWave = 32;      // lattice nodes per wave = MFMA columns
constexpr int kBM = kWaves * kRowsPerWave;  // nodes per workgroup
constexpr int kBN = 64;               // vocabulary entries per tile = two 32-row MFMA tiles
constexpr int kThreads = kWaves * kWave;

__device__ __forceinline__ float finite_or_zero(float m) {  // as K1: avoids inf - inf in the exponent
    return (m == -INFINITY || m == INFINITY) ? 0.0f : m;
}

// One tile's LDS buffer: the weight tile image followed by the bias slice.  The image is a sequence of 1-KiB
// "pieces" (what one wave-instruction of the DMA writes: 64 lanes x 16 bytes, lane-linear, i.e. 1024 / 2H
// consecutive weight rows), each piece starting kPiecePitch = 1040 bytes after the previous one.  The 16
// bytes of padding rotate consecutive rows by one 16-byte bank slot, so the 16 lanes that a ds_read_b128
// serves together (16 different rows, same k) hit 16 different slots for H = 512 (2 rows share a slot for
// H = 256, 4 for H = 128), and every fragment address is lane base + compile-time offset.
template <int H>
struct TileBuf {
    static constexpr int kCPR = H / 8;                    // 16-byte chunks per weight row
    static constexpr int kRPP = kWave / kCPR;             // weight rows per piece
    static constexpr int kPieces = kBN / kRPP;            // pieces per tile
    static constexpr int kPiecePitch = kWave * 16 + 16;   // 1040
    static constexpr int kWBytes = kPieces * kPiecePitch;
    static constexpr int kBiasBytes = kBN * 4;            // fp32 bias slice
    static constexpr int kBytes = kWBytes + kBiasBytes;
    static_assert(kCPR <= kWave && kWave % kCPR == 0 && kBN % kRPP == 0, "tile shape");
};

// LDS-DMA through inline asm: the compiler's wait-count pass, seeing an LDS-DMA of its own in flight,
// puts `s_waitcnt vmcnt(0)` in front of the next LDS read (it cannot tell the two buffers apart), which
// would wait for the tile being prefetched at the top of every tile.  These DMAs are invisible to it; the
// kernel orders them itself with one counted `s_waitcnt vmcnt(N)` + `s_barrier` per tile.
// M0 carries the LDS byte address of lane 0's 16 bytes; the hardware adds lane * size.  M0 is reserved by the
// compiler and not preserved around a statement: it is written in the statement that reads it and restored.
__device__ __forceinline__ void dma16(unsigned voff, const void *sbase, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}
__device__ __forceinline__ void dma4_ptr(const void *src, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_addr)
                 : "memory");
}

constexpr int kLoaders = kWaves / 2;  // waves 4-7 issue the DMAs, in the phase where they run an epilogue

// Weight tile with first row c0 -> LDS buffer at byte address `lds`, by the kLoaders loader waves
// (`part` = 0 .. kLoaders-1): piece q = j * kLoaders + part is 1 KiB of consecutive weight memory.
template <typename DT, int H>
__device__ __forceinline__ void stage_tile(const typename DT::elem *__restrict__ weight,
                                           const float *__restrict__ bias, int c0, unsigned lds, int part, int lane) {
    typedef TileBuf<H> TB;
    static_assert(TB::kPieces % kLoaders == 0, "pieces must divide over the loader waves");
    const char *tile = reinterpret_cast<const char *>(weight) + (size_t)c0 * H * 2;  // first weight row of the tile
#pragma unroll
    for (int j = 0; j < TB::kPieces / kLoaders; ++j) {
        const int q = j * kLoaders + part;
        dma16(lane * 16, tile + (size_t)q * (kWave * 16), lds + q * TB::kPiecePitch);
    }
    if (part == 0) dma4_ptr(bias + c0 + lane, lds + TB::kWBytes);
}

// 64 MFMAs of one tile: acc[t] (t = 0, 1: vocabulary entries t*32 .. t*32+31 of the tile) over all of H.
// `abase` = LDS byte address of this lane's weight row (l & 31) of the tile buffer + 16 h; the fragment of
// k-step s and tile t is at a compile-time offset from it.  The A fragments are read kPF k-steps ahead of the
// MFMAs that use them; the order is pinned with sched_group_barrier (left alone, hipcc reads each fragment
// right in front of its MFMA and waits for it).
template <typename DT, int H>
__device__ __forceinline__ void tile_mma(unsigned abase, unsigned bias_lds, const typename Mma<DT>::frag (&hid)[H / 16],
                                         f32x16 (&acc)[2]) {
    typedef TileBuf<H> TB;
    typedef typename Mma<DT>::frag frag;
    typedef const __attribute__((address_space(3))) frag *lds_frag_ptr;
    typedef const __attribute__((address_space(3))) f32x4 *lds_f4_ptr;
    constexpr int KS = H / 16;
    constexpr int kPF = FUSED_PF;
    // accumulators start at the bias: register i of tile t is vocabulary entry t*32 + 8 (i>>2) + 4 h + (i&3)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b4 = *(lds_f4_ptr)(uintptr_t)(bias_lds + (t * 32 + 8 * g) * 4);
            acc[t][4 * g + 0] = b4.x;
            acc[t][4 * g + 1] = b4.y;
            acc[t][4 * g + 2] = b4.z;
            acc[t][4 * g + 3] = b4.w;
        }
#ifdef FUSED_ABLATE_LDSREAD  // timing only: operands from registers, no LDS reads in the matrix phase
    auto lda = [&](int s, int t) { return hid[(s + t) % KS]; };
#else
    auto lda = [&](int s, int t) {
        return *(lds_frag_ptr)(uintptr_t)(abase + t * (32 / TB::kRPP) * TB::kPiecePitch + 32 * s);
    };
#endif
    frag a[KS][2];
    __builtin_amdgcn_s_setprio(FUSED_MMA_PRIO);  // the partner wave on this SIMD is in its (vector ALU) epilogue
#pragma unroll
    for (int s = 0; s < kPF && s < KS; ++s) {
        a[s][0] = lda(s, 0);
        a[s][1] = lda(s, 1);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 8 + 2 * kPF, 0);  // bias reads + the first kPF k-steps' fragments
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        // reads first: behind the second MFMA they would wait 32 cycles for the matrix pipe before issuing
        if (s + kPF < KS) {
            a[s + kPF][0] = lda(s + kPF, 0);
            a[s + kPF][1] = lda(s + kPF, 1);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // the fragments of k-step s + kPF
        }
#ifndef FUSED_ABLATE_MMA
        acc[0] = Mma<DT>::mma(a[s][0], hid[s], acc[0]);
        acc[1] = Mma<DT>::mma(a[s][1], hid[s], acc[1]);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // 2 MFMAs of k-step s
#else
        asm volatile("" ::"v"(a[s][0]), "v"(a[s][1]));
#endif
    }
    __builtin_amdgcn_s_setprio(0);
}

// Running {max, sum 2^((x - max) log2e)} of this lane's share of a node's row.
struct RowStat {
    float m, s;
};

// Epilogue of one tile (vocabulary entries c0 .. c0+63): round to the 16-bit type, fold the ROUNDED values into
// the row statistics, store.  Rounding is monotonic, so the maximum of the rounded values is the rounded
// maximum: one pass over the raw accumulators finds it, the second pass rounds, exponentiates and stores group
// by group (few live registers: the kernel sits at the 256-VGPR limit of two waves per SIMD).
// MASKED: the last tile of a vocabulary that is no multiple of 64 starts at V - 64 and overlaps its
// predecessor; entries below `vmin` were counted there and only take part in the stores (same values again).
// Rows of a vocabulary that is no multiple of 8 are not 16-byte aligned: the 16-byte stores then go to 2-byte
// aligned addresses, which the memory pipeline accepts (unaligned access mode); measured against element stores
// from the accumulator layout: V = 4234 3.24 -> 2.81 ms, V = 5455 4.14 -> 3.50 ms.
struct StoreLane {   // where this lane stores after the transpose through LDS
    unsigned xw;     // LDS byte address this lane WRITES its groups to: xbuf + (l & 31) * kXPitch + 8 h
    unsigned xr;     // LDS byte address it READS a 16-byte piece from: xbuf + (l >> 2) * kXPitch + (l & 3) * 16
    unsigned voff;   // byte offset in the workgroup's 256 logits rows: (32 wave + (l >> 2)) * 2V + (l & 3) * 16
    char *base; unsigned nbytes;
    int ldv;         // elements between consecutive rows of the logits tensor
    __amdgpu_buffer_rsrc_t rows;  // the workgroup's rows of the logits tensor (rows past the end are out of range:
                                  // the hardware drops those stores, every wave still issues the same number)
};
constexpr int kXPitch = 80;  // bytes per row of the transpose buffer: 64 of data, 16-byte aligned, 2-way banks at worst

template <typename DT, bool MASKED>
__device__ __forceinline__ void tile_epilogue_impl(const f32x16 (&acc)[2], RowStat &st,
                                                   typename DT::elem *out_row, const StoreLane &sl, int c0,
                                                   int vmin, int V, int h) {
    typedef typename DT::elem E;
    auto widen = [](unsigned w, float &lo, float &hi) {
        if constexpr (__is_same(E, __bf16)) {
            lo = __uint_as_float(w << 16);
            hi = __uint_as_float(w & 0xffff0000u);
        } else {
            lo = (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu));
            hi = (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
        }
    };
    auto counted = [&](int t, int i) { return !MASKED || c0 + t * 32 + 8 * (i >> 2) + 4 * h + (i & 3) >= vmin; };
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, counted(t, i) ? acc[t][i] : -INFINITY);  // (drops a NaN; the
    // sum below brings it back, as in K1)
    {
        float lo, hi;
        widen(DT::pack2(mx, mx), lo, hi);
        mx = lo;
    }
    __builtin_amdgcn_s_setprio(FUSED_EPI_PRIO);
    const float m_new = fmaxf(st.m, mx);
    const float mm = finite_or_zero(m_new);
    const float carried = (st.s == 0.0f) ? 0.0f : st.s * __builtin_amdgcn_exp2f((finite_or_zero(st.m) - mm) * kLog2e);
    st.m = m_new;
    float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // four independent chains (summed in a fixed order below)
    const float nmm2 = -mm * kLog2e;

#pragma unroll
    for (int t = 0; t < 2; ++t) {
        unsigned pk[8];  // pk[2g], pk[2g+1]: entries 8g+4h .. 8g+4h+3 of tile t as 16-bit pairs
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            pk[q] = DT::pack2(acc[t][2 * q], acc[t][2 * q + 1]);
#ifndef FUSED_ABLATE_EXP
            float lo, hi;
            widen(pk[q], lo, hi);
            // 2^(x log2e - max log2e) in one fused multiply-add: its rounding (2^-24 of |x log2e|) is far below
            // what the 16-bit logit itself resolves, so nothing is lost against subtracting first as K1 does
            part[q & 3] += counted(t, 2 * q) ? __builtin_amdgcn_exp2f(__builtin_fmaf(lo, kLog2e, nmm2)) : 0.0f;
            part[q & 3] += counted(t, 2 * q + 1) ? __builtin_amdgcn_exp2f(__builtin_fmaf(hi, kLog2e, nmm2)) : 0.0f;
#endif
        }
        {
            // A lane holds 16 entries of ONE node in four groups of 8 bytes; stored from here, a wave instruction
            // would scatter 64 pieces of 16 bytes over 32 rows.  The wave's 32 x 32 tile goes through its
            // private LDS buffer instead (DS instructions of one wave execute in order: no wait, no barrier)
            // and comes back with four consecutive lanes on the 64 contiguous bytes of one row.
            // (may_alias: the 8-byte writes and the 16-byte reads go to the same bytes; with distinct plain types
            // the compiler is free to move the next tile's writes above this tile's reads)
            typedef unsigned long long __attribute__((may_alias)) u64_alias;
            typedef u32x4 __attribute__((may_alias)) u32x4_alias;
            typedef __attribute__((address_space(3))) u64_alias *lds_u64;
            typedef const __attribute__((address_space(3))) u32x4_alias *lds_u128;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(lds_u64)(uintptr_t)(sl.xw + 16 * g) = (unsigned long long)pk[2 * g] | ((unsigned long long)pk[2 * g + 1] << 32);
            u32x4 o2[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) o2[k] = *(lds_u128)(uintptr_t)(sl.xr + k * 16 * kXPitch);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const u32x4 o = o2[k];
#if !defined(FUSED_ABLATE_STORE)
                // sc1: written through and not kept in the XCD's L2, which the weight tiles need (4.3 MB of
                // weights per 4 MiB L2, re-read by every workgroup, against 4.4 GB of logits passing through)
#ifdef FUSED_ABLATE_GSTORE
                if ((size_t)sl.voff + (size_t)k * 16 * sl.ldv * 2 < (size_t)sl.nbytes)
                    *reinterpret_cast<u32x4 *>(sl.base + sl.voff + (size_t)(k * 16 * sl.ldv + c0 + t * 32) * 2) = o;
#else
                // (whole offset in the VGPR operand: with a scalar offset register hipcc lets a vector
                // instruction overwrite the data registers in the cycle after the store, and on gfx950 the
                // store then picks up the new values: 1 in 10^5 elements came out wrong)
                __builtin_amdgcn_raw_buffer_store_b128(o, sl.rows, sl.voff + (unsigned)((k * 16 * sl.ldv + c0 + t * 32) * 2), 0,
                                                       FUSED_STORE_AUX);
#endif
#ifdef FUSED_ABLATE_DBGWAIT
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#else
                asm volatile("" ::"v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w));
#endif
            }
        }
    }
    st.s = carried + ((part[0] + part[1]) + (part[2] + part[3]));
    __builtin_amdgcn_s_setprio(0);
}

// Tile n covers vocabulary entries c0 .. c0+63 with c0 = min(64 n, V - 64).
template <typename DT>
__device__ __forceinline__ void tile_epilogue(const f32x16 (&acc)[2], RowStat &st, typename DT::elem *out_row,
                                              const StoreLane &sl, int n, int V, int h) {
    const int vmin = n * kBN;
    if (vmin + kBN <= V)  // workgroup-uniform
        tile_epilogue_impl<DT, false>(acc, st, out_row, sl, vmin, vmin, V, h);
    else
        tile_epilogue_impl<DT, true>(acc, st, out_row, sl, V - kBN, vmin, V, h);
}

// Two waves share a SIMD (waves w and w + 4).  Waves 0-3 multiply tile n while waves 4-7 run the epilogue of
// tile n-1, then the roles swap: the matrix pipe and the vector ALU of every SIMD are busy in both halves
// of a step instead of both waves queueing for the same unit.
template <typename DT, int H>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
void rnnt_fused_proj_k1(const typename DT::elem *__restrict__ hidden, const typename DT::elem *__restrict__ weight,
                        const float *__restrict__ bias, typename DT::elem *__restrict__ logits,
                        const int *__restrict__ labels, const int *__restrict__ act_lens,
                        const int *__restrict__ label_lens, Lattice L, int maxT, int U1, int V, int ldv, int blank,
                        long long row_cap, long long row_tile0, int vsplit, float2 *__restrict__ partial, Compact cm) {
    typedef TileBuf<H> TB;
    typedef typename Mma<DT>::frag frag;
    typedef typename DT::elem E;
    constexpr int KS = H / 16;  // MFMA k-steps
    constexpr int kXBytes = kRowsPerWave * kXPitch;  // per-wave transpose buffer of the epilogue
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * TB::kBytes + kWaves * kXBytes];

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const bool lead = wave < kWaves / 2;  // waves 0-3 multiply first
    const int r = lane & 31, h = lane >> 5;
    // vsplit = 1: workgroup b owns row tile row_tile0 + b and walks the whole vocabulary.  vsplit = S > 1 (the
    // surplus row tiles of a launch that does not fill a whole round of CUs): S workgroups share a row tile, each
    // walks 1/S of the vocabulary tiles and leaves its {max, sum} per node in `partial` for rnnt_fused_merge.
    const long long row_tile = row_tile0 + blockIdx.x / vsplit;
    const int split = blockIdx.x % vsplit;
    // Rows are lattice nodes: every row of the [B, maxT, U1] tensor, or (compacted order, lattice.h) the nodes inside
    // the lattices only; the grid is cut for `row_cap` rows and the workgroups past the last node leave at once.
    const long long total_rows = compact_total(cm, row_cap);
    if (row_tile * kBM >= total_rows) return;  // workgroup-uniform, before the first barrier
    const long long my_row = row_tile * kBM + wave * kRowsPerWave + r;
    const long long ld_row = my_row < total_rows ? my_row : total_rows - 1;
    const int ntiles_all = (V + kBN - 1) / kBN;
    const int n0 = (int)((long long)split * ntiles_all / vsplit), ntiles = (int)((long long)(split + 1) * ntiles_all / vsplit);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;

    // this wave's 32 hidden vectors: B operand, k-step s = elements [16 s + 8 h, +8) of row l & 31
    frag hid[KS];
    {
        const E *hp = hidden + (size_t)compact_row(cm, ld_row) * H + 8 * h;  // (hidden stays [B, maxT, U1, H])
#pragma unroll
        for (int s = 0; s < KS; ++s) hid[s] = *reinterpret_cast<const frag *>(hp + 16 * s);
    }
    const int part = wave - kLoaders;  // loader index of waves 4-7
    // tile_mma: this lane's weight row (l & 31) inside a tile buffer, + 16 h for its half of a k-step
    const unsigned arow = (unsigned)((r / TB::kRPP) * TB::kPiecePitch + (r % TB::kRPP) * (H * 2) + 16 * h);

    if (!lead) stage_tile<DT, H>(weight, bias, min(n0 * kBN, V - kBN), lds0, part, lane);

    RowStat st = {-INFINITY, 0.0f};
    E *out_row = logits + (size_t)ld_row * (size_t)ldv;  // ldv: elements from one node's row to the next (>= V)
    StoreLane sl;
    {
        const unsigned xbuf = lds0 + 2 * TB::kBytes + wave * kXBytes;
        sl.xw = xbuf + r * kXPitch + 8 * h;
        sl.xr = xbuf + (lane >> 2) * kXPitch + (lane & 3) * 16;
        sl.voff = (unsigned)((wave * kRowsPerWave + (lane >> 2)) * ldv * 2 + (lane & 3) * 16);
        const long long wg_row0 = row_tile * kBM;
        const long long wg_rows = total_rows - wg_row0 < kBM ? total_rows - wg_row0 : kBM;
        sl.base = reinterpret_cast<char *>(logits) + (size_t)wg_row0 * ldv * 2;
        sl.nbytes = (unsigned)(wg_rows * ldv * 2);
        sl.ldv = ldv;
        int store_bytes = (int)(wg_rows * ldv * 2);
#ifdef FUSED_ABLATE_STORE_LEAD  // timing only: the hardware drops the stores of waves 0-3 (out of range)
        if (lead) store_bytes = 0;
#endif
#ifdef FUSED_ABLATE_STORE_LOADER
        if (!lead) store_bytes = 0;
#endif
        sl.rows = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(logits) + (size_t)wg_row0 * ldv * 2, (short)0,
                                                    store_bytes, 0x00020000);
    }

    f32x16 acc[2];

    // (the builtin, so that the compiler knows the hidden-vector loads have landed and does not count them
    // down with vmcnt(N) waits inside the loop; 0x0f70 = vmcnt(0), other counters untouched)
    __builtin_amdgcn_s_waitcnt(0x0f70);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    // Step n has two halves separated by a barrier.  First half: waves 0-3 multiply tile n; waves 4-7 issue the
    // DMAs of tile n+1 into the other buffer (its readers finished with it before the barrier that ended step
    // n-1) and run the epilogue of tile n-1.  Second half: waves 0-3 run the epilogue of tile n, waves 4-7
    // multiply tile n.  The DMAs have the whole step to land; their issuers wait for them before the barrier
    // that ends the step.  Stores need no wait at all.
    for (int n = n0; n < ntiles; ++n) {
        const unsigned buf = lds0 + ((n - n0) & 1) * TB::kBytes;
        STAMP(0);
        if (lead) {
            tile_mma<DT, H>(buf + arow, buf + TB::kWBytes + 16 * h, hid, acc);
        } else {
#ifndef FUSED_ABLATE_DMA
            if (n + 1 < ntiles)
                stage_tile<DT, H>(weight, bias, min((n + 1) * kBN, V - kBN), lds0 + ((n + 1 - n0) & 1) * TB::kBytes, part,
                                  lane);
#endif
            if (n > n0) tile_epilogue<DT>(acc, st, out_row, sl, n - 1, V, h);
        }
        STAMP(1);
        asm volatile("s_barrier" ::: "memory");
        STAMP(2);
        if (lead) {
            tile_epilogue<DT>(acc, st, out_row, sl, n, V, h);
            STAMP(3);
            asm volatile("s_barrier" ::: "memory");
        } else {
            tile_mma<DT, H>(buf + arow, buf + TB::kWBytes + 16 * h, hid, acc);
            // The DMAs of this step are older than its epilogue's stores and the counter retires in order:
            // allow exactly those stores to stay in flight (their write acknowledgements take longer than
            // a step under load; waiting for them stalled every step).
            STAMP(3);
            // In the first step there was no epilogue: the youngest four operations ARE DMAs and must land too
            // (seen as stale pieces of the walk's second tile, once in ~150 small launches, when many workgroups
            // start cold at once and a DMA takes longer than the 64 MFMAs between its issue and this wait).
            if (n == n0)
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");  // 4 16-byte stores per epilogue
        }
    }
    if (!lead) tile_epilogue<DT>(acc, st, out_row, sl, ntiles - 1, V, h);
    const float m = st.m, ssum = st.s;

    // ---- merge the two lanes of a node, gather the blank and label logits, write the lattice planes ----
    const float m_o = __shfl_xor(m, 32, kWave), s_o = __shfl_xor(ssum, 32, kWave);
    const float M = fmaxf(m, m_o);
    const float MM = finite_or_zero(M);
    float S = (ssum == 0.0f) ? 0.0f : ssum * __builtin_amdgcn_exp2f((finite_or_zero(m) - MM) * kLog2e);
    S += (s_o == 0.0f) ? 0.0f : s_o * __builtin_amdgcn_exp2f((finite_or_zero(m_o) - MM) * kLog2e);
    // (lane l adds its own share first, lane l+32 the other's first: only lanes 0-31 write, so the order is fixed)

    const bool row_ok = my_row < total_rows;
    if (vsplit > 1) {  // workgroup-uniform: a share of the vocabulary only; rnnt_fused_merge finishes the node
        if (row_ok && h == 0) partial[(size_t)(my_row - row_tile0 * kBM) * vsplit + split] = make_float2(M, S);
        return;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's own stores of its 32 rows are complete
    const RowCoord c = decode_row(compact_row(cm, ld_row), maxT, U1);  // (looked up again: not kept across the walk)
    const int Tb = act_lens[c.b];
    const int Ub = label_lens[c.b];
    const bool valid = row_ok && lens_ok(Tb, Ub, maxT, U1) && c.t < Tb && c.u <= Ub;
    const bool emit = valid && c.u < Ub;
    const int y = emit ? labels[(size_t)c.b * (U1 - 1) + c.u] : blank;
    const bool y_ok = y >= 0 && y < V;
    // lanes 0-31 fetch the blank logit, lanes 32-63 the label logit of the same node, with AGENT-SCOPE loads
    // (global_load_ushort ... sc1): they are served by the XCD's L2, where this wave's own stores have arrived (the
    // s_waitcnt vmcnt(0) above: the vector L1 is write-through), never by a line of the CU's L1 that a neighbouring
    // wave pulled in before those stores landed.  (Picking the two values out of the accumulators / the transpose
    // buffer in the epilogue of the tile that holds them was built in round 3: the label and the picked pair stay
    // live across the vocabulary walk, and at the 256-register limit of this kernel that cost 23-30 spilled VGPRs
    // with scratch reloads inside the walk; a load per node at the end costs nothing there.)
    const int pick = h == 0 ? blank : (y_ok ? y : 0);
    const unsigned short xbits = __hip_atomic_load(reinterpret_cast<const unsigned short *>(out_row) + pick,
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float xg = DT::to_float(__builtin_bit_cast(E, xbits));
    const float x_other = __shfl_xor(xg, 32, kWave);
    if (valid && h == 0) {
        const float logs = row_log_sum(S, M);
        const float xb = xg;
        const float xy = y_ok ? x_other : __builtin_nanf("");
        const size_t o = skew_index(c.b, c.t, c.u, L.ND, L.US);
        L.ml[o] = make_float2(MM, logs);
        L.lp[o] = make_float2((xb - MM) - logs, emit ? (xy - MM) - logs : -INFINITY);
    }
}

// Finishes the nodes whose vocabulary was shared by `vsplit` workgroups: combines their {max, sum} pairs (one per
// lane, butterfly reductions: a fixed order), gathers the blank and label logits from the stored row and writes
// the lattice planes, as the tail of rnnt_fused_proj_k1 does for a node one workgroup walked alone.  One wave per node.
template <typename DT>
__global__ __launch_bounds__(256) void rnnt_fused_merge(const float2 *__restrict__ partial, int vsplit, long long row0,
                                                        long long row_cap,
                                                        const typename DT::elem *__restrict__ logits,
                                                        const int *__restrict__ labels,
                                                        const int *__restrict__ act_lens,
                                                        const int *__restrict__ label_lens, Lattice L, int maxT, int U1,
                                                        int V, int ldv, int blank, Compact cm) {
    const int lane = threadIdx.x & (kWave - 1);
    const long long row = row0 + (long long)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (row >= compact_total(cm, row_cap)) return;  // wave-uniform
    const float2 *p = partial + (size_t)(row - row0) * vsplit;
    const RowCoord c = decode_row(compact_row(cm, row), maxT, U1);
    const int Tb = act_lens[c.b];
    const int Ub = label_lens[c.b];
    if (!(lens_ok(Tb, Ub, maxT, U1) && c.t < Tb && c.u <= Ub)) return;
    const bool emit = c.u < Ub;
    const int y = emit ? labels[(size_t)c.b * (U1 - 1) + c.u] : blank;
    const bool y_ok = y >= 0 && y < V;
    const typename DT::elem *out_row = logits + (size_t)row * (size_t)ldv;
    const float xb = DT::to_float(out_row[blank]);
    const float xy = y_ok ? DT::to_float(out_row[y]) : __builtin_nanf("");

    float M = -INFINITY;
    for (int i = lane; i < vsplit; i += kWave) M = fmaxf(M, p[i].x);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) M = fmaxf(M, __shfl_xor(M, d, kWave));
    const float MM = finite_or_zero(M);
    float S = 0.0f;
    for (int i = lane; i < vsplit; i += kWave) {
        const float2 q = p[i];
        S += (q.y == 0.0f) ? 0.0f : q.y * __builtin_amdgcn_exp2f((finite_or_zero(q.x) - MM) * kLog2e);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) S += __shfl_xor(S, d, kWave);
    if (lane == 0) {
        const float logs = row_log_sum(S, M);
        const size_t o = skew_index(c.b, c.t, c.u, L.ND, L.US);
        L.ml[o] = make_float2(MM, logs);
        L.lp[o] = make_float2((xb - MM) - logs, emit ? (xy - MM) - logs : -INFINITY);
    }
}

// Compute units of the current device (the size of one round of workgroups: the kernel runs one per CU).
int compute_units() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

template <typename DT, int H>
rnntStatus_t launch_fused(const void *hidden, const void *weight, const float *bias, void *logits,
                          const int *labels, const int *act_lens, const int *label_lens, int V, int ldv, int B,
                          const Lattice &L, const rnntOptions &o, long long row_cap, int *rowmap) {
    typedef typename DT::elem E;
    // rows the grid is cut for: every row of the tensor, or (compacted order) what the logits buffer holds
    const long long tensor_rows = (long long)B * o.maxT * o.maxU;
    const long long rows = rowmap && row_cap < tensor_rows ? row_cap : tensor_rows;
    const long long grid = (rows + kBM - 1) / kBM;
    if (grid > 0x7fffffffLL) return RNNT_STATUS_INVALID_VALUE;
    hipStream_t st = static_cast<hipStream_t>(o.stream);
    const Compact cm = {rowmap, L.spans, B};
    if (rowmap) {
        rnntStatus_t s = launch_node_spans(act_lens, label_lens, B, L, o);
        if (s == RNNT_STATUS_SUCCESS) s = launch_compact_rows(rowmap, L, B, rows, o);  // (rows numbered with 32 bits)
        if (s != RNNT_STATUS_SUCCESS) return s;
    }
    // One workgroup per CU at a time: `grid` row tiles take ceil(grid / CUs) rounds, and a last round with few
    // tiles costs as much as a full one (config 2: 2050 tiles on 256 CUs = 8 rounds + 2 tiles, +11 %).  The
    // surplus tiles (up to half a round) are therefore launched apart, their vocabulary shared by as many
    // workgroups as fill the round (and as the scratch holds); the {max, sum} pairs go through the alpha and beta
    // planes, which K2 only writes later.
    const int cus = compute_units();
    const int ntiles = (V + kBN - 1) / kBN;
    long long full = grid / cus * cus, surplus = grid - full;
    int vsplit = 1;
    if (surplus > 0 && surplus <= cus / 2) {
        vsplit = (int)(cus / surplus);
        if (vsplit > ntiles) vsplit = ntiles;
        // scratch: the alpha and beta planes (adjacent in the workspace), 8 bytes per node and share
        const size_t scratch = 2 * ((size_t)B * (L.ND + 1) * L.US + 2 * L.US) * sizeof(float);
        const size_t fit = scratch / ((size_t)surplus * kBM * sizeof(float2));
        if ((size_t)vsplit > fit) vsplit = (int)fit;
    }
    if (FUSED_NO_VSPLIT || vsplit < 2) {
        full = grid;
        surplus = 0;
        vsplit = 1;
    }
    float2 *partial = reinterpret_cast<float2 *>(L.alpha);
#define FUSED_ARGS                                                                                            \
    static_cast<const E *>(hidden), static_cast<const E *>(weight), bias, static_cast<E *>(logits), labels,   \
        act_lens, label_lens, L, o.maxT, o.maxU, V, ldv, o.blank_label, rows
#define FUSED_LAUNCH(NWG, TILE0, SPLIT)                                                                      \
    hipLaunchKernelGGL((rnnt_fused_proj_k1<DT, H>), dim3((unsigned)(NWG)), dim3(kThreads), 0, st, FUSED_ARGS, \
                       (long long)(TILE0), (int)(SPLIT), partial, cm)
    if (full > 0) FUSED_LAUNCH(full, 0, 1);
    if (surplus > 0) {
        FUSED_LAUNCH(surplus * vsplit, full, vsplit);
        const long long row0 = full * kBM, n = rows - row0;
        hipLaunchKernelGGL((rnnt_fused_merge<DT>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, partial, vsplit,
                           row0, rows, static_cast<const E *>(logits), labels, act_lens, label_lens, L, o.maxT, o.maxU, V,
                           ldv, o.blank_label, cm);
    }
#undef FUSED_LAUNCH
#undef FUSED_ARGS
    return hipGetLastError() == hipSuccess ? RNNT_STATUS_SUCCESS : RNNT_STATUS_EXECUTION_FAILED;
}

template <typename DT>
rnntStatus_t launch_fused_h(int H, const void *hidden, const void *weight, const float *bias, void *logits,
                            const int *labels, const int *act_lens, const int *label_lens, int V, int ldv, int B,
                            const Lattice &L, const rnntOptions &o, long long row_cap, int *rowmap) {
    switch (H) {
        case 512: return launch_fused<DT, 512>(hidden, weight, bias, logits, labels, act_lens, label_lens, V, ldv, B, L, o, row_cap, rowmap);
        case 256: return launch_fused<DT, 256>(hidden, weight, bias, logits, labels, act_lens, label_lens, V, ldv, B, L, o, row_cap, rowmap);
        case 128: return launch_fused<DT, 128>(hidden, weight, bias, logits, labels, act_lens, label_lens, V, ldv, B, L, o, row_cap, rowmap);
        default: return RNNT_STATUS_INVALID_VALUE;
    }
}

}  // namespace

extern "C" {

#ifdef FUSED_ABLATE_STAMP
int rnnt_fused_debug_set_stamp_buffer(void *p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &p, sizeof(p));
}
#endif

rnntStatus_t rnnt_joint_loss_compact_workspace_size(int minibatch, int maxT, int maxU, size_t *size_bytes) {
    if (!size_bytes || minibatch <= 0 || maxT <= 0 || maxU <= 0) return RNNT_STATUS_INVALID_VALUE;
    const long long rows = (long long)minibatch * maxT * maxU;
    if (rows > 0x7fffffffLL) return RNNT_STATUS_INVALID_VALUE;
    *size_bytes = ((size_t)rows * sizeof(int) + 255) / 256 * 256;
    return RNNT_STATUS_SUCCESS;
}

static rnntStatus_t joint_loss_forward(const void *hidden, const void *weight, const float *bias, void *logits_out,
                                       int logits_row_stride, long long logits_rows, void *compact_workspace,
                                       const int *labels, const int *act_lens, const int *label_lens, int hidden_size,
                                       int alphabet_size, int minibatch, float *costs, float *loss_out, int reduction,
                                       void *workspace, rnntOptions options) {
    if (logits_row_stride < alphabet_size) return RNNT_STATUS_INVALID_VALUE;
    // (a workgroup's 256 rows are addressed with 32-bit byte offsets)
    if ((long long)logits_row_stride * 2 * kBM > 0x7fffffffLL) return RNNT_STATUS_INVALID_VALUE;
    if (!hidden || !weight || !bias || !logits_out || !act_lens || !label_lens || !workspace || !costs)
        return RNNT_STATUS_INVALID_VALUE;
    if (!labels && options.maxU > 1) return RNNT_STATUS_INVALID_VALUE;
    if (alphabet_size < 64) return RNNT_STATUS_INVALID_VALUE;  // a tile is 64 vocabulary entries wide
    if (alphabet_size <= 0 || minibatch <= 0 || options.maxT <= 0 || options.maxU <= 0 || options.maxU > RNNT_MAX_LATTICE_COLUMNS)
        return RNNT_STATUS_INVALID_VALUE;
    if (options.blank_label < 0 || options.blank_label >= alphabet_size) return RNNT_STATUS_INVALID_VALUE;
    if (options.dtype != RNNT_DTYPE_BF16 && options.dtype != RNNT_DTYPE_F16) return RNNT_STATUS_INVALID_VALUE;
    if (reduction < RNNT_REDUCTION_NONE || reduction > RNNT_REDUCTION_SUM) return RNNT_STATUS_INVALID_VALUE;
    if (reinterpret_cast<uintptr_t>(workspace) % 256 != 0 || reinterpret_cast<uintptr_t>(hidden) % 16 != 0 ||
        reinterpret_cast<uintptr_t>(weight) % 16 != 0 || reinterpret_cast<uintptr_t>(bias) % 4 != 0 ||
        reinterpret_cast<uintptr_t>(logits_out) % 2 != 0)
        return RNNT_STATUS_INVALID_VALUE;
    const Lattice w = carve_workspace(workspace, options.maxT, options.maxU, minibatch);
    if (options.stage_events && options.stage_events[0])
        (void)hipEventRecord(static_cast<hipEvent_t>(options.stage_events[0]), static_cast<hipStream_t>(options.stream));
    const rnntStatus_t s =
        options.dtype == RNNT_DTYPE_BF16
            ? launch_fused_h<BF16>(hidden_size, hidden, weight, bias, logits_out, labels, act_lens, label_lens,
                                   alphabet_size, logits_row_stride, minibatch, w, options, logits_rows,
                                   static_cast<int *>(compact_workspace))
            : launch_fused_h<F16>(hidden_size, hidden, weight, bias, logits_out, labels, act_lens, label_lens,
                                  alphabet_size, logits_row_stride, minibatch, w, options, logits_rows,
                                  static_cast<int *>(compact_workspace));
    if (s != RNNT_STATUS_SUCCESS) return s;
    return launch_lattice_and_costs(act_lens, label_lens, minibatch, costs, loss_out, reduction, w, options);
}

rnntStatus_t rnnt_joint_loss_forward_strided(const void *hidden, const void *weight, const float *bias,
                                             void *logits_out, int logits_row_stride, const int *labels,
                                             const int *act_lens, const int *label_lens, int hidden_size,
                                             int alphabet_size, int minibatch, float *costs, float *loss_out,
                                             int reduction, void *workspace, rnntOptions options) {
    return joint_loss_forward(hidden, weight, bias, logits_out, logits_row_stride, 0, nullptr, labels, act_lens,
                              label_lens, hidden_size, alphabet_size, minibatch, costs, loss_out, reduction, workspace,
                              options);
}

rnntStatus_t rnnt_joint_loss_forward_compact(const void *hidden, const void *weight, const float *bias,
                                             void *logits_out, int logits_row_stride, long long logits_rows,
                                             const int *labels, const int *act_lens, const int *label_lens,
                                             int hidden_size, int alphabet_size, int minibatch, float *costs,
                                             float *loss_out, int reduction, void *workspace, void *compact_workspace,
                                             rnntOptions options) {
    if (!compact_workspace || reinterpret_cast<uintptr_t>(compact_workspace) % 256 != 0 || logits_rows <= 0)
        return RNNT_STATUS_INVALID_VALUE;
    return joint_loss_forward(hidden, weight, bias, logits_out, logits_row_stride, logits_rows, compact_workspace, labels,
                              act_lens, label_lens, hidden_size, alphabet_size, minibatch, costs, loss_out, reduction,
                              workspace, options);
}


rnntStatus_t rnnt_joint_loss_forward(const void *hidden, const void *weight, const float *bias, void *logits_out,
                                     const int *labels, const int *act_lens, const int *label_lens,
                                     int hidden_size, int alphabet_size, int minibatch, float *costs,
                                     float *loss_out, int reduction, void *workspace, rnntOptions options) {
    return rnnt_joint_loss_forward_strided(hidden, weight, bias, logits_out, alphabet_size, labels, act_lens, label_lens,
                                           hidden_size, alphabet_size, minibatch, costs, loss_out, reduction, workspace,
                                           options);
}

}  // extern "C"

// fused_bwd_mi355.hip — backward of the joint network's output projection + RNN-T loss with K3 as the
// PROLOGUE of the two matrix products (backward half).
//
// Replaces, for 16-bit operands, the chain autograd runs behind
//     logits = Linear(H -> V)(hidden)                 /root/reference/src/rnnt/model.py:17,59
//     loss   = RNNTLoss(logits, ...) ; loss.backward()  /root/reference/src/rnnt/model.py:116, bin/train.py:43
// i.e. K3 (g = d loss / d logits, written as a [B,T,U+1,V] tensor) followed by the two GEMMs of a Linear
// layer's backward, dH = g W and dW = g^T hidden, each of which reads g back.  Here g is never written:
//   rnnt_bwd_node_coefs  one record per lattice node: what K3 derives from alpha/beta/lp/ll per node
//   rnnt_fused_bwd_dh    owns 128 lattice nodes, walks the vocabulary: reads the stored logits tile, forms the
//                        gradient tile in registers exactly as K3's grad_chunk does (same operations, same
//                        rounding to the 16-bit type), feeds it to the matrix cores against the weight tile
//   rnnt_fused_bwd_dw    owns 128 vocabulary entries, walks a range of lattice nodes: the same prologue, the
//                        gradient tile read TRANSPOSED (ds_read_b64_tr_b16) against the hidden tile; the bias
//                        gradient is the column sum of the same tiles
//   rnnt_fused_bwd_reduce  sums the per-row-range partials of dW / dbias in range order (no atomics anywhere:
//                        every output is bitwise reproducible)
// The logits are read twice (once per product) instead of g being written once and read twice.
//
// Shape of the two MFMA kernels (gfx950, wave64, 8 waves = 2 x 4, one workgroup per CU):
//   acc[128 x 512] += P[128 x 64] . Q[64 x 512] per step, 32 v_mfma_f32_32x32x16 per wave and step.
//   Q (weight rows / hidden rows, 1 KiB each) arrives by LDS-DMA, one row per wave instruction, two buffers;
//   its 64-byte windows are XOR-ed with (row & 3) on the SOURCE address so that the transposed reads of the
//   B operand (four rows x 64 bytes per 32-lane half) touch every bank once.
//   P is the gradient tile: computed by all 512 threads (16 logits each) from 16-byte loads issued two tiles
//   ahead, written to LDS as 16-bit values.  Waves 0-3 multiply tile j and then compute tile j+1, waves 4-7
//   (their SIMD partners) compute first and multiply second, so the matrix pipe and the vector ALU of every
//   SIMD have work in both halves of a step; one barrier per step.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rnnt_mi355.h"
#include "elem_types.h"
#include "lattice.h"
#include "mma.h"

namespace {

using namespace rnnt_mi355_detail;

constexpr float kLog2e = 1.4426950408889634f;

#ifndef FUSED_BWD_VALU_PER_MFMA
#define FUSED_BWD_VALU_PER_MFMA 3  // 0: developer A/B build with the gradient values computed after the MFMAs
#endif

// Developer timing builds with one part removed (tools/build_bwd_variants.sh; results are wrong by design):
//   FUSED_BWD_ABLATE_MMA    no MFMAs (fragments still read)      FUSED_BWD_ABLATE_VALU  no exp/fma on the logits
//   FUSED_BWD_ABLATE_DMA    no LDS-DMA of the Q tiles             FUSED_BWD_ABLATE_LOAD  no logits / coefficient loads
//   FUSED_BWD_ABLATE_READ   fragments from registers, no LDS reads

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * kWave;
constexpr int kBM = 128;  // output rows per workgroup: lattice nodes (dH) or vocabulary entries (dW)
constexpr int kBK = 64;   // contraction length per step: vocabulary entries (dH) or lattice nodes (dW)
constexpr int kH = 512;   // the projection's input size (the reference's inner_dim); N of both products
constexpr int kQRow = kH * 2;          // bytes of one Q row = one LDS-DMA piece
constexpr int kQBytes = kBK * kQRow;   // 64 KiB
constexpr int kGBytes = kBM * kBK * 2; // 16 KiB
constexpr int kLdsBytes = 2 * kQBytes + 2 * kGBytes;  // 160 KiB: all of a CU's LDS
constexpr unsigned kOob = 0x80000000u;  // buffer offset past every range used here: the load returns 0, no request

// What K3 derives per lattice node (NodeCoef of lattice.h) in the form the tile prologue uses.
struct __attribute__((aligned(32))) NodeG {
    float m, nl2, sa, syk;  // g_v = sa * 2^((x_v - m) log2e + nl2)  [- sbk at blank] [- syk at y]
    float sbk;
    int y;                  // label of the node's emit arc, -1 if none
    int valid;              // 0: padding (t >= T_b or u > U_b, or lengths that do not fit) -> g = 0 whatever the logits hold
    int pad;
};

__global__ __launch_bounds__(256) void rnnt_bwd_node_coefs(NodeG *__restrict__ out, const int *__restrict__ labels,
                                                           const int *__restrict__ act_lens,
                                                           const int *__restrict__ label_lens, Lattice L,
                                                           const float *__restrict__ grad_out, int grad_out_stride,
                                                           float grad_scale, int maxT, int U1, long long total_rows,
                                                           long long padded_rows) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= padded_rows) return;
    NodeG g = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, -1, 0, 0};
    if (r < total_rows) {
        const RowCoord c = decode_row(r, maxT, U1);
        const int Tb = act_lens[c.b];
        const int Ub = label_lens[c.b];
        if (lens_ok(Tb, Ub, maxT, U1) && c.t < Tb && c.u <= Ub) {
            const bool emit = c.u < Ub;
            const NodeIn in = node_in(L, c, emit, grad_out, grad_out_stride);
            const int y = emit ? labels[(size_t)c.b * (U1 - 1) + c.u] : -1;
            const NodeCoef n = node_coef_from(in, c.t, c.u, Tb, Ub, y, grad_scale);
            g.m = n.m;
            g.nl2 = -n.l * kLog2e;
            g.sa = n.sa;
            g.syk = n.syk;
            g.sbk = n.sbk;
            g.y = n.y;
            g.valid = 1;
        }
    }
    out[r] = g;
}

// Eight gradient values from eight stored logits: K3's grad_chunk (rnnt_mi355.hip) operation for operation —
// x - m, fused multiply-add with log2e and -l log2e, exp2, times sa, then the two sparse corrections in K3's
// order — so that the rounded tile equals what rnnt_backward would have written.  In two parts: the dense part
// (branch-free: it is scheduled between the MFMAs of the previous tile) and the corrections.  dy / db: position of
// the label / blank logit relative to this chunk's first element; the corrections are real branches (one lane in
// hundreds takes them); `keep_from`: elements below it belong to the previous, overlapping tile and give 0.
template <typename DT>
__device__ __forceinline__ void grad8_main(const u32x4 &raw, float m, float nl2, float sa, float (&x)[8]) {
    DT::unpack(raw, x);
#ifndef FUSED_BWD_ABLATE_VALU
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[j] - m, kLog2e, nl2)) * sa;
#endif
}
__device__ __forceinline__ void grad8_fix(float syk, float sbk, int dy, int db, bool masked, int keep_from, float (&x)[8]) {
    if ((unsigned)db < 8u) {
        asm volatile("; blank-logit correction" ::: "memory");  // keeps this a branch (see grad_chunk)
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] -= (j == db) ? sbk : 0.0f;
    }
    if ((unsigned)dy < 8u) {
        asm volatile("; label-logit correction" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] -= (j == dy) ? syk : 0.0f;
    }
    if (masked) {  // workgroup-uniform
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (j >= keep_from) ? x[j] : 0.0f;
    }
}

// Per-lane source offset of an LDS-DMA piece = one 1-KiB Q row: lane l lands on bytes [16 l, 16 l + 16) of the
// row's LDS image, which must hold the 64-byte window (l >> 2) ^ (row & 3) of the row.
__device__ __forceinline__ unsigned dma_voff(int lane, int row_and_3) {
    return (unsigned)((((lane >> 2) ^ row_and_3) << 6) | ((lane & 3) << 4));
}

typedef __attribute__((address_space(3))) u32x4 *lds_u128;
typedef const __attribute__((address_space(3))) u32x4 *lds_cu128;

// B fragments of one step for this wave's four 32-column tiles and the 32 MFMAs of the step.
//   bq[jn]: this lane's LDS address inside the Q image for column tile jn, k-step 0, rows 8h + q
//   AFRAG(s, i): A fragment of k-step s, row tile i (differs between the two kernels)
//   A_READS: LDS read instructions per A fragment (1: ds_read_b128, 2: transposed)
// The fragments of k-step s + 1 are read between the MFMAs of k-step s; the order is pinned with
// sched_group_barrier (left alone, hipcc hoists every read of the step to its top and spills the accumulators).
//   VALU: vector-ALU work that does not depend on this step (the dense part of the next tile's gradient values):
//         three of its instructions go between consecutive MFMAs, where the matrix pipe leaves the issue port idle
template <typename DT, int A_READS, typename AFRAG, typename VALU, typename KEEP>
__device__ __forceinline__ void mma_step(const unsigned (&bq)[4], f32x16 (&acc)[2][4], AFRAG afrag, VALU valu, KEEP keep) {
    typedef typename Mma<DT>::frag frag;
    constexpr int KS = kBK / 16;
    frag a[2][2], b[2][4];  // [k-step parity]
    auto read_frags = [&](int s) {
#ifndef FUSED_BWD_ABLATE_READ
#pragma unroll
        for (int i = 0; i < 2; ++i) a[s & 1][i] = afrag(s, i);
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
            b[s & 1][jn] = lds_read_tr_frag<frag>(bq[jn] + s * 16 * kQRow, 4 * kQRow);
#else
#pragma unroll
        for (int i = 0; i < 2; ++i) a[s & 1][i] = __builtin_bit_cast(frag, u32x4{bq[0], bq[1], bq[2], (unsigned)(s + i)});
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) b[s & 1][jn] = __builtin_bit_cast(frag, u32x4{bq[jn], bq[1], bq[2], (unsigned)s});
#endif
    };
    __builtin_amdgcn_sched_barrier(0);
    read_frags(0);
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * A_READS + 8, 0);
    valu();
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (s + 1 < KS) read_frags(s + 1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) {
#ifndef FUSED_BWD_ABLATE_MMA
                acc[i][jn] = Mma<DT>::mma(a[s & 1][i], b[s & 1][jn], acc[i][jn]);
#else
                asm volatile("" ::"v"(a[s & 1][i]), "v"(b[s & 1][jn]));
#endif
            }
        // 8 MFMAs with the next k-step's 2 A_READS + 8 reads spread between them: 2,1,x,1,2,1,x,1 (x = A_READS)
#define MMA1 __builtin_amdgcn_sched_group_barrier(0x008, 1, 0)
#define DSR(N) __builtin_amdgcn_sched_group_barrier(0x100, N, 0)
#define VAL __builtin_amdgcn_sched_group_barrier(0x002, FUSED_BWD_VALU_PER_MFMA, 0)
        if (s + 1 < KS) {
            MMA1; DSR(2); VAL; MMA1; DSR(1); VAL; MMA1; DSR(A_READS); VAL; MMA1; DSR(1); VAL;
            MMA1; DSR(2); VAL; MMA1; DSR(1); VAL; MMA1; DSR(A_READS); VAL; MMA1; DSR(1); VAL;
        } else {
            MMA1; VAL; MMA1; VAL; MMA1; VAL; MMA1; VAL; MMA1; VAL; MMA1; VAL; MMA1; VAL; MMA1; VAL;
        }
#undef MMA1
#undef DSR
#undef VAL
    }
    keep();  // (uses of the vector work's results inside this scheduling region: hipcc otherwise sinks the work into
             // the conditional block that consumes it, behind the MFMAs)
    __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------------------------------------------
// dH[node, :] = sum_v g[node, v] W[v, :]
// ---------------------------------------------------------------------------------------------------------
template <typename DT>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
void rnnt_fused_bwd_dh(const typename DT::elem *__restrict__ logits, const typename DT::elem *__restrict__ weight,
                       const NodeG *__restrict__ coefs, typename DT::elem *__restrict__ grad_hidden, int V, int blank,
                       long long total_rows) {
    typedef typename DT::elem E;
    typedef typename Mma<DT>::frag frag;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kLdsBytes];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const unsigned gbase = lds0 + 2 * kQBytes;

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 31, h = lane >> 5;
    const long long wg_row0 = (long long)blockIdx.x * kBM;
    const int wg_rows = (int)(total_rows - wg_row0 < kBM ? total_rows - wg_row0 : kBM);
    const int NT = (V + kBK - 1) / kBK;
    auto tile_c0 = [&](int j) { return min(j * kBK, V - kBK); };  // the last tile overlaps its predecessor

    // ---- the gradient tile's producer side: thread -> chunk gc of rows gr0 and gr0 + 64 ----
    const int gc = tid & 7, gr0 = tid >> 3;
    NodeG cf[2];
    unsigned row_off[2], gw[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int rl = gr0 + 64 * q;
        const long long row = wg_row0 + rl < total_rows ? wg_row0 + rl : total_rows - 1;
        const float4 c0 = *reinterpret_cast<const float4 *>(&coefs[row]);
        const int4 c1 = *(reinterpret_cast<const int4 *>(&coefs[row]) + 1);
        const bool ok = rl < wg_rows && c1.z != 0;
        cf[q].m = ok ? c0.x : 0.0f;
        cf[q].nl2 = ok ? c0.y : 0.0f;
        cf[q].sa = ok ? c0.z : 0.0f;
        cf[q].syk = ok ? c0.w : 0.0f;
        cf[q].sbk = ok ? __int_as_float(c1.x) : 0.0f;
        cf[q].y = ok ? c1.y : -1;
        row_off[q] = ok ? (unsigned)((rl * V + 8 * gc) * 2) : kOob;  // padding nodes load nothing and read zeros
        // LDS image [128 nodes][64 v]: 128-byte rows, 16-byte chunk c of row n at chunk c ^ ((n >> 1) & 7)
        gw[q] = (unsigned)(rl * 128 + ((gc ^ ((rl >> 1) & 7)) << 4));
    }
    const __amdgpu_buffer_rsrc_t rows_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<E *>(logits) + (size_t)wg_row0 * V, (short)0, (int)((size_t)wg_rows * V * 2), 0x00020000);

    // ---- consumer side ----
    unsigned aoff[4], boff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) aoff[s] = (unsigned)((64 * wm + r) * 128 + (((2 * s + h) ^ ((r >> 1) & 7)) << 4));
    {
        const int q4 = (lane & 15) >> 2, p = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
            boff[jn] = (unsigned)((8 * h + q4) * kQRow + (((4 * wn + jn) ^ q4) << 6) + 32 * g1 + 8 * p);
    }

    auto stage_q = [&](int j) {  // weight rows tile_c0(j) .. +63 -> Q buffer j & 1, eight rows per wave
        const char *src = reinterpret_cast<const char *>(weight) + (size_t)tile_c0(j) * kQRow;
        const unsigned dst = lds0 + (j & 1) * kQBytes;
#pragma unroll
        for (int i = 0; i < kBK / kWaves; ++i) {
            const int kr = wave * (kBK / kWaves) + i;
#ifndef FUSED_BWD_ABLATE_DMA
            lds_dma16(dma_voff(lane, i & 3), src + (size_t)kr * kQRow, dst + kr * kQRow);
#endif
        }
    };
    auto load_tile = [&](int j, u32x4 (&raw)[2]) {
        const unsigned off = j < NT ? (unsigned)(tile_c0(j) * 2) : kOob;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#ifndef FUSED_BWD_ABLATE_LOAD
            raw[q] = __builtin_amdgcn_raw_buffer_load_b128(rows_rsrc, row_off[q] + off, 0, 0);
#else
            raw[q] = u32x4{off, row_off[q], 0u, 0u};
#endif
        }
    };
    float gx[2][8];  // the gradient values of the next tile, before the corrections
    auto g_main = [&](const u32x4 (&raw)[2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) grad8_main<DT>(raw[q], cf[q].m, cf[q].nl2, cf[q].sa, gx[q]);
    };
    auto keep_gx = [&]() {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            asm volatile("" ::"v"(gx[q][0]), "v"(gx[q][1]), "v"(gx[q][2]), "v"(gx[q][3]), "v"(gx[q][4]), "v"(gx[q][5]),
                         "v"(gx[q][6]), "v"(gx[q][7]));
    };
    auto g_finish = [&](int j) {
        const int c0 = tile_c0(j);
        const int vcol = c0 + 8 * gc;
        const bool masked = c0 != j * kBK;
        const unsigned gbuf = gbase + (j & 1) * kGBytes;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            grad8_fix(cf[q].syk, cf[q].sbk, cf[q].y - vcol, blank - vcol, masked, j * kBK - vcol, gx[q]);
            *(lds_u128)(uintptr_t)(gbuf + gw[q]) = DT::pack(gx[q]);
        }
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.0f;
    // (the per-step base addresses go through an empty asm: hipcc otherwise re-associates base + lane offset +
    // constant into loop-invariant lane offset + constant registers and one vector add per LDS read)
    auto opaque = [](unsigned v) {
        asm volatile("" : "+v"(v));
        return v;
    };
    auto mma = [&](int j, const u32x4 (&raw)[2]) {  // tile j on the matrix cores, tile j + 1's gradient values beside it
        const unsigned gbuf = gbase + (j & 1) * kGBytes, qbuf = lds0 + (j & 1) * kQBytes;
        unsigned ga[4], bq[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ga[k] = opaque(gbuf + aoff[k]);
            bq[k] = opaque(qbuf + boff[k]);
        }
        mma_step<DT, 1>(
            bq, acc,
            [&](int s, int i) { return *(const __attribute__((address_space(3))) frag *)(uintptr_t)(ga[s] + i * 32 * 128); },
            [&]() { g_main(raw); }, keep_gx);
    };

    // Step j: the logits of tile j + 2 and the weight rows of tile j + 1 are requested at its top and have the whole
    // step to arrive; the wait in front of the barrier is the BUILTIN so that the compiler's own counter knows the
    // loads have landed (it does not see the DMAs and would otherwise wait for them at the first use of `raw`).
    u32x4 raw[2], raw_next[2];
    stage_q(0);
    load_tile(0, raw);
    g_main(raw);
    g_finish(0);
    load_tile(1, raw);
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
    asm volatile("s_barrier" ::: "memory");
    for (int j = 0; j < NT; ++j) {
        const bool more = j + 1 < NT;
        load_tile(j + 2, raw_next);  // (past the last tile: offsets out of range, nothing is fetched)
        if (more) stage_q(j + 1);    // its buffer was last read in step j - 1, which every wave has left
        mma(j, raw);                 // tile j on the matrix cores; the dense part of tile j + 1's gradient beside it
        if (more) g_finish(j + 1);
        __builtin_amdgcn_s_waitcnt(0x0070);
        asm volatile("s_barrier" ::: "memory");
#pragma unroll
        for (int q = 0; q < 2; ++q) raw[q] = raw_next[q];
    }

    // ---- dH tile -> 16-bit, through LDS (the tile buffers are free: every wave is past the last barrier) so
    // that rows leave as whole 16-byte pieces ----
    constexpr int SP = kQRow + 16;  // staging pitch
    typedef __attribute__((address_space(3))) E *lds_elem;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = 64 * wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int n = (4 * wn + jn) * 32 + r;
                *(lds_elem)(uintptr_t)(lds0 + row * SP + n * 2) = DT::from_float(acc[i][jn][e]);
            }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kBM * (kQRow / 16) / kThreads; ++k) {
        const int id = tid + kThreads * k;
        const int row = id / (kQRow / 16), ch = id % (kQRow / 16);
        const u32x4 v = *(lds_cu128)(uintptr_t)(lds0 + row * SP + ch * 16);
        if (row < wg_rows)
            *reinterpret_cast<u32x4 *>(reinterpret_cast<char *>(grad_hidden) + (size_t)(wg_row0 + row) * kQRow + ch * 16) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// partial dW[v, :] = sum over this workgroup's lattice nodes of g[node, v] hidden[node, :]
// ---------------------------------------------------------------------------------------------------------
template <typename DT>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
void rnnt_fused_bwd_dw(const typename DT::elem *__restrict__ logits, const typename DT::elem *__restrict__ hidden,
                       const NodeG *__restrict__ coefs, float *__restrict__ partial, float *__restrict__ pbias, int V,
                       int blank, long long total_rows, long long coef_rows, int nvb, int steps_per_chunk) {
    typedef typename DT::elem E;
    typedef typename Mma<DT>::frag frag;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kLdsBytes];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const unsigned gbase = lds0 + 2 * kQBytes;

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 31, h = lane >> 5;
    const int vb = blockIdx.x % nvb;  // consecutive workgroups share a node range: its hidden rows stay in L2 / MALL
    const int rc = blockIdx.x / nvb;
    const int vmin = vb * kBM;
    const int vb0 = min(vmin, V - kBM);  // the last vocabulary block overlaps its predecessor
    const bool masked = vb0 != vmin;
    const long long row_begin = (long long)rc * steps_per_chunk * kBK;
    const long long rows_here = total_rows - row_begin;
    const int nsteps = rows_here <= 0 ? 0 : (int)min((long long)steps_per_chunk, (rows_here + kBK - 1) / kBK);

    // ---- producer side: thread -> chunk gc (eight vocabulary entries) of nodes gr0 and gr0 + 32 of the step ----
    const int gc = tid & 15, gr0 = tid >> 4;
    const int vcol = vb0 + 8 * gc;
    const int db = blank - vcol, keep_from = vmin - vcol;
    unsigned gw[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int rl = gr0 + 32 * q;
        // LDS image [64 nodes][128 v]: 256-byte rows, 64-byte window w of row n at window w ^ (n & 3)
        gw[q] = (unsigned)(rl * 256 + ((((gc >> 2) ^ (rl & 3)) << 6) | ((gc & 3) << 4)));
    }
    struct Pre {  // what the prologue of one step needs, loaded a step ahead
        u32x4 raw[2];
        float4 c0[2];
        int4 c1[2];
    };
    auto load_step = [&](int j, Pre &p) {
        const long long row0 = row_begin + (long long)j * kBK;
        long long left = total_rows - row0;
        left = j < nsteps ? (left < kBK ? left : kBK) : 0;  // rows of this step that exist; none: every load returns 0
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<E *>(logits) + (size_t)(j < nsteps ? row0 : 0) * V, (short)0, (int)((size_t)left * V * 2),
            0x00020000);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int rl = gr0 + 32 * q;
            long long cr = row0 + rl;
            cr = cr < coef_rows ? cr : coef_rows - 1;
#ifndef FUSED_BWD_ABLATE_LOAD
            p.c0[q] = *reinterpret_cast<const float4 *>(&coefs[cr]);
            p.c1[q] = *(reinterpret_cast<const int4 *>(&coefs[cr]) + 1);
            p.raw[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)((rl * V + vcol) * 2), 0, 0);
#else
            p.c0[q] = make_float4(0.f, 0.f, 1.f, 0.f);
            p.c1[q] = make_int4(0, (int)cr, 1, 0);
            p.raw[q] = u32x4{(unsigned)cr, 0u, 0u, 0u};
            (void)rsrc;
#endif
        }
    };
    float bsum[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};  // this thread's share of the bias gradient
    float gx[2][8];
    float fsyk[2], fsbk[2];
    int fdy[2];
    auto g_main = [&](const Pre &p) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            // padding nodes: zero coefficients (rnnt_bwd_node_coefs) and zero logits, whatever the tensor holds there
            const bool ok = p.c1[q].z != 0;
            u32x4 raw = p.raw[q];
            raw.x = ok ? raw.x : 0u;
            raw.y = ok ? raw.y : 0u;
            raw.z = ok ? raw.z : 0u;
            raw.w = ok ? raw.w : 0u;
            grad8_main<DT>(raw, p.c0[q].x, p.c0[q].y, p.c0[q].z, gx[q]);
            fsyk[q] = p.c0[q].w;
            fsbk[q] = __int_as_float(p.c1[q].x);
            fdy[q] = p.c1[q].y - vcol;
        }
    };
    auto keep_gx = [&]() {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            asm volatile("" ::"v"(gx[q][0]), "v"(gx[q][1]), "v"(gx[q][2]), "v"(gx[q][3]), "v"(gx[q][4]), "v"(gx[q][5]),
                         "v"(gx[q][6]), "v"(gx[q][7]));
    };
    auto g_finish = [&](int j) {
        const unsigned gbuf = gbase + (j & 1) * kGBytes;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            grad8_fix(fsyk[q], fsbk[q], fdy[q], db, masked, keep_from, gx[q]);
#pragma unroll
            for (int e = 0; e < 8; ++e) bsum[e] += gx[q][e];
            *(lds_u128)(uintptr_t)(gbuf + gw[q]) = DT::pack(gx[q]);
        }
    };
    auto stage_q = [&](int j) {  // hidden rows of step j -> Q buffer j & 1 (rows past the end: the last row, times g = 0)
        const long long row0 = row_begin + (long long)j * kBK;
        const unsigned dst = lds0 + (j & 1) * kQBytes;
#pragma unroll
        for (int i = 0; i < kBK / kWaves; ++i) {
            const int kr = wave * (kBK / kWaves) + i;
            const long long row = row0 + kr < total_rows ? row0 + kr : total_rows - 1;
#ifndef FUSED_BWD_ABLATE_DMA
            lds_dma16(dma_voff(lane, i & 3), reinterpret_cast<const char *>(hidden) + (size_t)row * kQRow, dst + kr * kQRow);
#endif
        }
    };

    // ---- consumer side: A = g^T (vocabulary entry on the lane, nodes in the registers: transposed reads) ----
    unsigned aoff[2], boff[4];
    {
        const int q4 = (lane & 15) >> 2, p = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) aoff[i] = (unsigned)((8 * h + q4) * 256 + (((2 * wm + i) ^ q4) << 6) + 32 * g1 + 8 * p);
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
            boff[jn] = (unsigned)((8 * h + q4) * kQRow + (((4 * wn + jn) ^ q4) << 6) + 32 * g1 + 8 * p);
    }
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.0f;
    auto opaque = [](unsigned v) {  // (see rnnt_fused_bwd_dh)
        asm volatile("" : "+v"(v));
        return v;
    };
    auto mma = [&](int j, const Pre &p) {
        const unsigned gbuf = gbase + (j & 1) * kGBytes, qbuf = lds0 + (j & 1) * kQBytes;
        unsigned ga[2], bq[4];
#pragma unroll
        for (int k = 0; k < 2; ++k) ga[k] = opaque(gbuf + aoff[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) bq[k] = opaque(qbuf + boff[k]);
        mma_step<DT, 2>(
            bq, acc, [&](int s, int i) { return lds_read_tr_frag<frag>(ga[i] + s * 16 * 256, 4 * 256); },
            [&]() { g_main(p); }, keep_gx);
    };

    if (nsteps > 0) {  // workgroup-uniform; the pipeline of rnnt_fused_bwd_dh
        Pre pre, pre_next;
        stage_q(0);
        load_step(0, pre);
        g_main(pre);
        g_finish(0);
        load_step(1, pre);
        __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
        asm volatile("s_barrier" ::: "memory");
        for (int j = 0; j < nsteps; ++j) {
            const bool more = j + 1 < nsteps;
            load_step(j + 2, pre_next);
            if (more) stage_q(j + 1);
            mma(j, pre);
            if (more) g_finish(j + 1);
            __builtin_amdgcn_s_waitcnt(0x0070);
            asm volatile("s_barrier" ::: "memory");
            pre = pre_next;
        }
    }

    // ---- partial products out: fp32, [rc][vb][128 v][512]; register e of a tile is one row of 32 consecutive floats ----
    float *out = partial + (size_t)blockIdx.x * kBM * kH;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = 64 * wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int n = (4 * wn + jn) * 32 + r;
                out[(size_t)row * kH + n] = acc[i][jn][e];
            }
    // ---- bias partial: the 32 threads that share a chunk column, summed in thread order ----
    float *bl = reinterpret_cast<float *>(smem);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) bl[gr0 * kBM + 8 * gc + e] = bsum[e];
    __syncthreads();
    if (tid < kBM) {
        float s = 0.0f;
        for (int g = 0; g < kThreads / 16; ++g) s += bl[g * kBM + tid];
        pbias[(size_t)blockIdx.x * kBM + tid] = s;
    }
}

// dW[v, :] and dbias[v] = sum over the node ranges, in range order.  One thread per four columns.
template <typename DT>
__global__ __launch_bounds__(256) void rnnt_fused_bwd_reduce(const float *__restrict__ partial,
                                                             const float *__restrict__ pbias,
                                                             typename DT::elem *__restrict__ grad_weight,
                                                             float *__restrict__ grad_bias, int V, int nvb, int nchunks) {
    const int v = blockIdx.x * 2 + (threadIdx.x >> 7);
    if (v >= V) return;
    const int col = (threadIdx.x & 127) * 4;
    const int last0 = V - kBM;  // first entry of the last (overlapping) block; it owns entries from (nvb-1)*128 on
    const int vb = v >= (nvb - 1) * kBM ? nvb - 1 : v / kBM;
    const int local = vb == nvb - 1 ? v - last0 : v - vb * kBM;
    if (grad_weight) {
        f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int c = 0; c < nchunks; ++c)
            s += *reinterpret_cast<const f32x4 *>(partial + (((size_t)c * nvb + vb) * kBM + local) * kH + col);
        typename DT::elem o[4] = {DT::from_float(s.x), DT::from_float(s.y), DT::from_float(s.z), DT::from_float(s.w)};
        *reinterpret_cast<uint2 *>(grad_weight + (size_t)v * kH + col) = *reinterpret_cast<const uint2 *>(o);
    }
    if (grad_bias && col == 0) {
        float s = 0.0f;
        for (int c = 0; c < nchunks; ++c) s += pbias[((size_t)c * nvb + vb) * kBM + local];
        grad_bias[v] = s;
    }
}

// How the lattice nodes are cut into ranges for the dW kernel: about 512 workgroups (two rounds of a 256-CU chip,
// one workgroup per CU), at least eight steps each.  Depends on the shape only (the workspace size must not depend
// on the device).
struct DwPlan {
    int nvb;              // vocabulary blocks of 128
    int nchunks;          // node ranges
    int steps_per_chunk;  // steps of 64 nodes per range
    long long coef_rows;  // node records: the ranges' rows and the dH kernel's row tiles
};
DwPlan dw_plan(long long rows, int V) {
    DwPlan p;
    p.nvb = (V + kBM - 1) / kBM;
    const long long steps = (rows + kBK - 1) / kBK;
    long long n = 512 / p.nvb;
    if (n < 1) n = 1;
    if (n > (steps + 7) / 8) n = (steps + 7) / 8;
    p.steps_per_chunk = (int)((steps + n - 1) / n);
    p.nchunks = (int)((steps + p.steps_per_chunk - 1) / p.steps_per_chunk);
    const long long a = (long long)p.nchunks * p.steps_per_chunk * kBK, b = (rows + kBM - 1) / kBM * kBM;
    p.coef_rows = a > b ? a : b;
    return p;
}
inline size_t align256(size_t n) { return (n + 255) / 256 * 256; }

struct BwdWorkspace {
    NodeG *coefs;
    float *partial;
    float *pbias;
    size_t bytes;
};
BwdWorkspace carve_bwd(void *ws, const DwPlan &p) {
    BwdWorkspace w;
    char *c = static_cast<char *>(ws);
    size_t off = 0;
    w.coefs = reinterpret_cast<NodeG *>(c + off);
    off += align256((size_t)p.coef_rows * sizeof(NodeG));
    w.partial = reinterpret_cast<float *>(c + off);
    off += align256((size_t)p.nchunks * p.nvb * kBM * kH * sizeof(float));
    w.pbias = reinterpret_cast<float *>(c + off);
    off += align256((size_t)p.nchunks * p.nvb * kBM * sizeof(float));
    w.bytes = off;
    return w;
}

template <typename DT>
rnntStatus_t launch_bwd(const void *hidden, const void *weight, const void *logits, const int *labels,
                        const int *act_lens, const int *label_lens, int V, int B, const float *grad_out,
                        int grad_out_stride, float grad_scale, void *grad_hidden, void *grad_weight, float *grad_bias,
                        const Lattice &L, void *bwd_ws, const rnntOptions &o) {
    typedef typename DT::elem E;
    const long long rows = (long long)B * o.maxT * o.maxU;
    const DwPlan p = dw_plan(rows, V);
    const BwdWorkspace w = carve_bwd(bwd_ws, p);
    hipStream_t st = static_cast<hipStream_t>(o.stream);
    const long long row_tiles = (rows + kBM - 1) / kBM;
    if (row_tiles > 0x7fffffffLL || (p.coef_rows + 255) / 256 > 0x7fffffffLL) return RNNT_STATUS_INVALID_VALUE;
    hipLaunchKernelGGL(rnnt_bwd_node_coefs, dim3((unsigned)((p.coef_rows + 255) / 256)), dim3(256), 0, st, w.coefs, labels,
                       act_lens, label_lens, L, grad_out, grad_out_stride, grad_scale, o.maxT, o.maxU, rows, p.coef_rows);
    if (o.stage_events && o.stage_events[1]) (void)hipEventRecord(static_cast<hipEvent_t>(o.stage_events[1]), st);
    if (grad_hidden)
        hipLaunchKernelGGL((rnnt_fused_bwd_dh<DT>), dim3((unsigned)row_tiles), dim3(kThreads), 0, st,
                           static_cast<const E *>(logits), static_cast<const E *>(weight), w.coefs,
                           static_cast<E *>(grad_hidden), V, o.blank_label, rows);
    if (o.stage_events && o.stage_events[2]) (void)hipEventRecord(static_cast<hipEvent_t>(o.stage_events[2]), st);
    if (grad_weight || grad_bias) {
        hipLaunchKernelGGL((rnnt_fused_bwd_dw<DT>), dim3((unsigned)(p.nchunks * p.nvb)), dim3(kThreads), 0, st,
                           static_cast<const E *>(logits), static_cast<const E *>(hidden), w.coefs, w.partial, w.pbias, V,
                           o.blank_label, rows, p.coef_rows, p.nvb, p.steps_per_chunk);
        hipLaunchKernelGGL((rnnt_fused_bwd_reduce<DT>), dim3((unsigned)((V + 1) / 2)), dim3(256), 0, st, w.partial,
                           w.pbias, static_cast<E *>(grad_weight), grad_bias, V, p.nvb, p.nchunks);
    }
    if (o.stage_events && o.stage_events[3]) (void)hipEventRecord(static_cast<hipEvent_t>(o.stage_events[3]), st);
    return hipGetLastError() == hipSuccess ? RNNT_STATUS_SUCCESS : RNNT_STATUS_EXECUTION_FAILED;
}

}  // namespace

extern "C" {

rnntStatus_t rnnt_joint_loss_backward_workspace_size(int hidden_size, int alphabet_size, int minibatch, int maxT,
                                                     int maxU, size_t *size_bytes) {
    if (!size_bytes || hidden_size != kH || alphabet_size < kBM || minibatch <= 0 || maxT <= 0 || maxU <= 0)
        return RNNT_STATUS_INVALID_VALUE;
    const DwPlan p = dw_plan((long long)minibatch * maxT * maxU, alphabet_size);
    *size_bytes = carve_bwd(nullptr, p).bytes;
    return RNNT_STATUS_SUCCESS;
}

rnntStatus_t rnnt_joint_loss_backward(const void *hidden, const void *weight, const void *logits, const int *labels,
                                      const int *act_lens, const int *label_lens, int hidden_size, int alphabet_size,
                                      int minibatch, const float *grad_out, int grad_out_stride, float grad_scale,
                                      void *grad_hidden, void *grad_weight, float *grad_bias, const void *workspace,
                                      void *backward_workspace, rnntOptions options) {
    if (!hidden || !weight || !logits || !act_lens || !label_lens || !workspace || !backward_workspace)
        return RNNT_STATUS_INVALID_VALUE;
    if (!labels && options.maxU > 1) return RNNT_STATUS_INVALID_VALUE;
    if (hidden_size != kH || alphabet_size < kBM) return RNNT_STATUS_INVALID_VALUE;
    if (minibatch <= 0 || options.maxT <= 0 || options.maxU <= 0 || options.maxU > RNNT_MAX_LATTICE_COLUMNS)
        return RNNT_STATUS_INVALID_VALUE;
    if (options.blank_label < 0 || options.blank_label >= alphabet_size) return RNNT_STATUS_INVALID_VALUE;
    if (options.dtype != RNNT_DTYPE_BF16 && options.dtype != RNNT_DTYPE_F16) return RNNT_STATUS_INVALID_VALUE;
    if (reinterpret_cast<uintptr_t>(workspace) % 256 != 0 || reinterpret_cast<uintptr_t>(backward_workspace) % 256 != 0 ||
        reinterpret_cast<uintptr_t>(hidden) % 16 != 0 || reinterpret_cast<uintptr_t>(weight) % 16 != 0 ||
        reinterpret_cast<uintptr_t>(logits) % 2 != 0 || reinterpret_cast<uintptr_t>(grad_hidden) % 16 != 0 ||
        reinterpret_cast<uintptr_t>(grad_weight) % 8 != 0 || reinterpret_cast<uintptr_t>(grad_bias) % 4 != 0)
        return RNNT_STATUS_INVALID_VALUE;
    if ((size_t)kBM * alphabet_size * 2 > 0x7fffffffu) return RNNT_STATUS_INVALID_VALUE;  // one buffer range per row tile
    const Lattice w = carve_workspace(const_cast<void *>(workspace), options.maxT, options.maxU, minibatch);
    if (options.stage_events && options.stage_events[0])
        (void)hipEventRecord(static_cast<hipEvent_t>(options.stage_events[0]), static_cast<hipStream_t>(options.stream));
    return options.dtype == RNNT_DTYPE_BF16
               ? launch_bwd<BF16>(hidden, weight, logits, labels, act_lens, label_lens, alphabet_size, minibatch, grad_out,
                                  grad_out_stride, grad_scale, grad_hidden, grad_weight, grad_bias, w, backward_workspace,
                                  options)
               : launch_bwd<F16>(hidden, weight, logits, labels, act_lens, label_lens, alphabet_size, minibatch, grad_out,
                                 grad_out_stride, grad_scale, grad_hidden, grad_weight, grad_bias, w, backward_workspace,
                                 options);
}

}  // extern "C"

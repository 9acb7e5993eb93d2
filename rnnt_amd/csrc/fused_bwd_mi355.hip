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
constexpr int kBK = 32;   // contraction length per step: vocabulary entries (dH) or lattice nodes (dW)
constexpr int kH = 512;   // the projection's input size (the reference's inner_dim); N of both products
constexpr int kQRow = kH * 2;          // bytes of one Q row = one LDS-DMA piece
constexpr int kQBytes = kBK * kQRow;   // 32 KiB
constexpr int kNQ = 3;                 // Q ring: the tile being multiplied, the next one (landed), one in flight
constexpr int kGBytes = kBM * kBK * 2; // 8 KiB: one 1-KiB LDS-DMA piece per wave
constexpr int kNG = 5;                 // gradient-tile ring: multiplied / converted in place / landed / two in flight
constexpr int kCBytes = kWaves * 256;  // dW: node records of one step, 256 bytes per wave
constexpr int kNC = 7;                 // dW: ring of node records: they land three steps before their tile is converted,
                                       // one step before the hidden rows they name are fetched
constexpr int kQ0 = 0, kG0 = kNQ * kQBytes, kC0 = kG0 + kNG * kGBytes;
constexpr int kLdsBytesDh = kC0, kLdsBytesDw = kC0 + kNC * kCBytes;  // 136 KiB, 150 KiB
constexpr int kDmaPerStep = kBK / kWaves;  // Q rows per wave and step

// What K3 derives per lattice node (NodeCoef of lattice.h) in the form the tile prologue uses.
struct __attribute__((aligned(32))) NodeG {
    float m, nl2, sa, syk;  // g_v = sa * 2^((x_v - m) log2e + nl2)  [- sbk at blank] [- syk at y]
    float sbk;
    int y;                  // label of the node's emit arc, -1 if none, kPadding for a padding node (t >= T_b or
                            // u > U_b, or lengths that do not fit): g = 0 there whatever the logits hold
    int row;                // the node's row of the [B, maxT, U1] tensor: where its hidden vector is (the records are
                            // in the order the logits are stored in, lattice.h: Compact); 0 for a padding record
    int pad;
};
constexpr int kPadding = -2;

__global__ __launch_bounds__(256) void rnnt_bwd_node_coefs(NodeG *__restrict__ out, const int *__restrict__ labels,
                                                           const int *__restrict__ act_lens,
                                                           const int *__restrict__ label_lens, Lattice L,
                                                           const float *__restrict__ grad_out, int grad_out_stride,
                                                           float grad_scale, int maxT, int U1, long long row_cap,
                                                           long long padded_rows, Compact cmap) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= padded_rows) return;
    NodeG g = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, kPadding, 0, 0};
    if (r < compact_total(cmap, row_cap)) {  // (rows are lattice nodes in the order the logits are stored in: lattice.h)
        const long long row = compact_row(cmap, r);
        const RowCoord c = decode_row(row, maxT, U1);
        g.row = (int)row;
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
        }
    }
    out[r] = g;
}

// Eight gradient values from eight stored logits: K3's grad_chunk (rnnt_mi355.hip) operation for operation —
// x - m, fused multiply-add with log2e and -l log2e, exp2, times sa, then the two sparse corrections in K3's
// order — so that the rounded tile equals what rnnt_backward would have written.  In two parts: the dense part
// (branch-free: it is scheduled between the MFMAs of the previous tile) and the corrections.  dy / db: position of
// the label / blank logit relative to this chunk's first element; the corrections are real branches (one lane in
// hundreds takes them); of a `masked` chunk only elements keep_from <= j < keep_until count, the others give 0:
// those below belong to the previous, overlapping tile (rows without padding), those above lie past the vocabulary,
// in the padding of a row whose stride is a multiple of 128 (EdgeMode below).
template <typename DT>
__device__ __forceinline__ void grad8_main(const u32x4 &raw, float m, float nl2, float sa, float (&x)[8]) {
    DT::unpack(raw, x);
#ifndef FUSED_BWD_ABLATE_VALU
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[j] - m, kLog2e, nl2)) * sa;
#endif
}
__device__ __forceinline__ void grad8_fix(float syk, float sbk, int dy, int db, bool masked, int keep_from, int keep_until,
                                          float (&x)[8]) {
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
        for (int j = 0; j < 8; ++j) x[j] = (j >= keep_from && j < keep_until) ? x[j] : 0.0f;  // (a select: drops NaN)
    }
}

// Per-lane source offset of an LDS-DMA piece = one 1-KiB Q row: lane l lands on bytes [16 l, 16 l + 16) of the
// row's LDS image, which must hold the 64-byte window (l >> 2) ^ (row & 3) of the row.
__device__ __forceinline__ unsigned dma_voff(int lane, int row_and_3) {
    return (unsigned)((((lane >> 2) ^ row_and_3) << 6) | ((lane & 3) << 4));
}

typedef __attribute__((address_space(3))) u32x4 *lds_u128;
typedef const __attribute__((address_space(3))) u32x4 *lds_cu128;

// ---- pipeline of both MFMA kernels ----------------------------------------------------------------------------
// Every global read inside the step loops is an LDS-DMA (inline asm, mma.h): no vector register is ever the
// destination of a load in flight, so nothing the register allocator does (copies, live-range splits) can touch
// data that has not landed, and the only wait counter to manage is the DMAs' own.  They retire in issue order
// and every wave issues the same operations in the same order in every step, so "the N youngest may still be in
// flight" is a compile-time constant.  Step j (kBK contraction elements, 16 MFMAs per wave):
//   1. DMA of Q tile j + 2 into Q slot (j + 2) % 3 (last read in step j - 1, which every wave has left), then of the
//      logits of tile j + 4 into gradient slot (j + 4) % 5 (last read in step j - 1) [dW: and its node records];
//      issued one by one between the first MFMAs of 2.
//   2. the MFMAs of tile j; between them the dense part of the conversion of tile j + 1: each thread reads the 16
//      bytes of logits its own lane's DMA fetched (landed and behind a barrier since step j - 1) ...
//   3. ... applies the corrections, rounds and writes the gradient values back IN PLACE
//   4. s_waitcnt vmcnt(N): everything older than the N operations issued after the Q DMAs of step j - 1 has landed,
//      i.e. Q tile j + 1 and the logits of tile j + 2; LDS writes done; barrier
// N = L + (kDmaPerStep + L) with L = 1 (dH) or 2 (dW) DMAs after the Q pieces of a step.
template <int L>
struct Waits {
    static constexpr int kStep = L + kDmaPerStep + L;
};
__device__ __forceinline__ void lds_dma4(unsigned voff, const void *sbase, unsigned lds_addr) {  // 4 bytes per lane
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}

// The 16 MFMAs of one step for this wave's 2 x 4 tiles of 32 x 32.
//   bq[jn]: this lane's LDS address inside the Q tile for column tile jn, k-step 0, rows 8h + q
//   AFRAG(s, i): A fragment of k-step s, row tile i (differs between the two kernels)
//   A_READS: LDS read instructions per A fragment (1: ds_read_b128, 2: transposed)
// The fragments of k-step 1 are read between the MFMAs of k-step 0; the order is pinned with sched_group_barrier
// (left alone, hipcc hoists every read to the top and spills the accumulators).
//   VALU: vector-ALU work that does not depend on this step (the dense part of the next tile's gradient values):
//         a few of its instructions go between consecutive MFMAs, where the matrix pipe leaves the issue port idle
//   BETWEEN(k), k = 0 .. 5: called after the k-th fragment read of k-step 1, i.e. between the MFMAs of k-step 0: the
//         step's LDS-DMAs go there one by one (an asm statement with a "memory" clobber stays between the LDS reads
//         around it).  Issued back to back at the top of the step they cost the wave 100-180 cycles each with
//         nothing else of its own to overlap; beside its MFMAs the issue is hidden.
template <typename DT, int A_READS, typename AFRAG, typename VALU, typename KEEP, typename BETWEEN>
__device__ __forceinline__ void mma_step(const unsigned (&bq)[4], f32x16 (&acc)[2][4], AFRAG afrag, VALU valu, KEEP keep,
                                         BETWEEN between) {
    typedef typename Mma<DT>::frag frag;
    constexpr int KS = kBK / 16;
    frag a[2][2], b[2][4];  // [k-step parity]
    auto read_frags = [&](int s) {
#ifndef FUSED_BWD_ABLATE_READ
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) {
            b[s & 1][jn] = lds_read_tr_frag<frag>(bq[jn] + s * 16 * kQRow, 4 * kQRow);
            if (s == 1) between(jn);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[s & 1][i] = afrag(s, i);
            if (s == 1) between(4 + i);
        }
#else
#pragma unroll
        for (int i = 0; i < 2; ++i) a[s & 1][i] = __builtin_bit_cast(frag, u32x4{bq[0], bq[1], bq[2], (unsigned)(s + i)});
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) b[s & 1][jn] = __builtin_bit_cast(frag, u32x4{bq[jn], bq[1], bq[2], (unsigned)s});
        if (s == 1)
            for (int k = 0; k < 6; ++k) between(k);
#endif
    };
    __builtin_amdgcn_sched_barrier(0);
    read_frags(0);
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * A_READS + 8, 0);
    static_assert(kBK / 16 == 2, "the DMAs are placed in k-step 0");
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
        // 8 MFMAs with the next k-step's 8 + 2 A_READS reads spread between them, in program order (four B fragments
        // of two reads each, then two A fragments of A_READS each): 2,1,1,2,1,1,x,x
#define MMA1 __builtin_amdgcn_sched_group_barrier(0x008, 1, 0)
#define DSR(N) __builtin_amdgcn_sched_group_barrier(0x100, N, 0)
#define VAL __builtin_amdgcn_sched_group_barrier(0x002, FUSED_BWD_VALU_PER_MFMA, 0)
        if (s + 1 < KS) {
            MMA1; DSR(2); VAL; MMA1; DSR(1); VAL; MMA1; DSR(1); VAL; MMA1; DSR(2); VAL;
            MMA1; DSR(1); VAL; MMA1; DSR(1); VAL; MMA1; DSR(A_READS); VAL; MMA1; DSR(A_READS); VAL;
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

// (per-step LDS base addresses go through an empty asm: hipcc otherwise re-associates base + lane offset +
// constant into loop-invariant lane offset + constant registers and one vector add per LDS read)
__device__ __forceinline__ unsigned opaque(unsigned v) {
    asm volatile("" : "+v"(v));
    return v;
}

// ---------------------------------------------------------------------------------------------------------
// dH[node, :] = sum_v g[node, v] W[v, :]
// ---------------------------------------------------------------------------------------------------------
template <typename DT>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
void rnnt_fused_bwd_dh(const typename DT::elem *__restrict__ logits, const typename DT::elem *__restrict__ weight,
                       const NodeG *__restrict__ coefs, typename DT::elem *__restrict__ grad_hidden, int V, int ldv,
                       int padded, int blank, long long row_cap, long long row_tile0, int vsplit,
                       float *__restrict__ split_partial, Compact cmap) {
    typedef typename DT::elem E;
    typedef typename Mma<DT>::frag frag;
    typedef Waits<1> W;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kLdsBytesDh];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 31, h = lane >> 5;
    // vsplit = 1: workgroup b owns row tile row_tile0 + b and walks the whole vocabulary.  vsplit = S > 1 (the surplus
    // row tiles of a launch that does not fill a whole round of CUs, see launch_bwd): S workgroups share a row tile,
    // each walks 1/S of the vocabulary tiles and leaves its fp32 sums in `split_partial` for rnnt_fused_bwd_dh_merge.
    const long long wg_row0 = (row_tile0 + blockIdx.x / vsplit) * kBM;
    const int split = blockIdx.x % vsplit;
    // rows = lattice nodes in the order the logits are stored in (lattice.h: Compact); the grid is cut for row_cap rows
    // and the workgroups past the last node leave at once
    const long long total_rows = compact_total(cmap, row_cap);
    if (wg_row0 >= total_rows) return;  // workgroup-uniform, before the first barrier
    const int wg_rows = (int)(total_rows - wg_row0 < kBM ? total_rows - wg_row0 : kBM);
    const int NT_all = (V + kBK - 1) / kBK;
    const int jt0 = (int)((long long)split * NT_all / vsplit);                 // first vocabulary tile of this workgroup
    const int NT = (int)((long long)(split + 1) * NT_all / vsplit) - jt0;      // its number of tiles (>= 1: vsplit <= NT_all)
    // Tile j of this workgroup is vocabulary tile jt0 + j.  The last tile of a vocabulary that is no multiple of 32: in
    // padded rows it runs on into the padding (entries from V on are masked), otherwise it starts at V - 32 and
    // overlaps its predecessor (whose entries are masked)
    auto tile_c0 = [&](int j) { return padded ? (jt0 + j) * kBK : min((jt0 + j) * kBK, V - kBK); };

    // ---- producer side.  Gradient tile in LDS: [128 nodes][32 v], 64-byte rows, 16-byte chunk c of node n at
    // position c ^ ((n >> 2) & 3).  Thread tid owns the 16 bytes at tid * 16, in every step: its lane's DMA fetches
    // the logits there, the thread converts them in place.  That is node gr, chunk gc:
    const int gr = tid >> 2;
    const int gc = (tid & 3) ^ ((gr >> 2) & 3);
    float cm, cnl2, csa, csyk, csbk;
    int cy;
    bool ok;
    {
        const long long row = wg_row0 + gr < total_rows ? wg_row0 + gr : total_rows - 1;
        const float4 c0 = *reinterpret_cast<const float4 *>(&coefs[row]);
        const int2 c1 = *(reinterpret_cast<const int2 *>(&coefs[row]) + 2);
        ok = gr < wg_rows && c1.y != kPadding;
        cm = ok ? c0.x : 0.0f;
        cnl2 = ok ? c0.y : 0.0f;
        csa = ok ? c0.z : 0.0f;
        csyk = ok ? c0.w : 0.0f;
        csbk = ok ? __int_as_float(c1.x) : 0.0f;
        cy = ok ? c1.y : -1;
    }
    // (rows past the end of the tensor: the last row's logits are fetched and ignored)
    const unsigned l_voff = (unsigned)(((gr < wg_rows ? gr : wg_rows - 1) * ldv + 8 * gc) * 2);
    const char *l_base = reinterpret_cast<const char *>(logits + (size_t)wg_row0 * ldv);
    __builtin_amdgcn_s_waitcnt(0x0070);  // the node record has landed: no compiler-visible load is in flight below

    // ---- consumer side ----
    unsigned aoff[2], boff[4];
#pragma unroll
    for (int s = 0; s < 2; ++s) aoff[s] = (unsigned)((64 * wm + r) * 64 + (((2 * s + h) ^ ((r >> 2) & 3)) << 4));
    {
        const int q4 = (lane & 15) >> 2, p = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
            boff[jn] = (unsigned)((8 * h + q4) * kQRow + (((4 * wn + jn) ^ q4) << 6) + 32 * g1 + 8 * p);
    }

    // step j's DMAs, k = 0 .. 3: weight row k of this wave's four of Q tile j + 2; k = 4: logits of tile j + 4
    auto issue = [&](int j, int k) {
        if (k < kDmaPerStep) {
            const int t = j + 2 < NT ? j + 2 : NT - 1;  // (past the end: the last tile again; never read)
            const int kr = wave * kDmaPerStep + k;
            // (weight rows past the vocabulary, padded rows only: the last row again, times g = 0)
            const char *src = reinterpret_cast<const char *>(weight) + (size_t)min(tile_c0(t) + kr, V - 1) * kQRow;
#ifndef FUSED_BWD_ABLATE_DMA
            if (j + 2 >= 0) lds_dma16(dma_voff(lane, k & 3), src, lds0 + kQ0 + ((j + 2) % kNQ) * kQBytes + kr * kQRow);
#endif
        } else if (k == kDmaPerStep) {
            const int t = j + 4 < NT ? j + 4 : NT - 1;
#ifndef FUSED_BWD_ABLATE_LOAD
            lds_dma16(l_voff, l_base + (size_t)tile_c0(t) * 2, lds0 + kG0 + ((j + 4) % kNG) * kGBytes + wave * 1024);
#endif
        }
    };
    float gx[8];  // the gradient values of the next tile, before the corrections
    auto keep_gx = [&]() {
        asm volatile("" ::"v"(gx[0]), "v"(gx[1]), "v"(gx[2]), "v"(gx[3]), "v"(gx[4]), "v"(gx[5]), "v"(gx[6]), "v"(gx[7]));
    };
    auto conv_main = [&](int j) {
        u32x4 raw = *(lds_cu128)(uintptr_t)(lds0 + kG0 + (j % kNG) * kGBytes + tid * 16);
        raw.x = ok ? raw.x : 0u;  // padding nodes: g = 0 whatever the logits hold
        raw.y = ok ? raw.y : 0u;
        raw.z = ok ? raw.z : 0u;
        raw.w = ok ? raw.w : 0u;
        grad8_main<DT>(raw, cm, cnl2, csa, gx);
    };
    auto conv_finish = [&](int j) {
        const int c0 = tile_c0(j);
        const int vcol = c0 + 8 * gc;
        grad8_fix(csyk, csbk, cy - vcol, blank - vcol, padded ? c0 + kBK > V : c0 != (jt0 + j) * kBK,
                  padded ? 0 : (jt0 + j) * kBK - vcol, padded ? V - vcol : 8, gx);
        *(lds_u128)(uintptr_t)(lds0 + kG0 + (j % kNG) * kGBytes + tid * 16) = DT::pack(gx);
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.0f;
    auto mma = [&](int j) {  // tile j on the matrix cores, tile j + 1's gradient values beside it
        const unsigned gbuf = lds0 + kG0 + (j % kNG) * kGBytes, qbuf = lds0 + kQ0 + (j % kNQ) * kQBytes;
        unsigned ga[2], bq[4];
#pragma unroll
        for (int k = 0; k < 2; ++k) ga[k] = opaque(gbuf + aoff[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) bq[k] = opaque(qbuf + boff[k]);
        mma_step<DT, 1>(
            bq, acc,
            [&](int s, int i) { return *(const __attribute__((address_space(3))) frag *)(uintptr_t)(ga[s] + i * 32 * 64); },
            [&]() { conv_main(j + 1); }, keep_gx, [&](int k) { issue(j, k); });
    };

    // ---- prologue: the DMAs of steps -4 .. -1, everything landed, tile 0 converted ----
    for (int j = -4; j < 0; ++j)
#pragma unroll
        for (int k = 0; k <= kDmaPerStep; ++k) issue(j, k);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    conv_main(0);
    conv_finish(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    for (int j = 0; j < NT; ++j) {
        mma(j);  // (and the DMAs of step j)
        if (j + 1 < NT) conv_finish(j + 1);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(W::kStep) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // nothing may land in LDS once it is reused below

    if (vsplit > 1) {  // workgroup-uniform: fp32 sums of this share of the vocabulary, [workgroup][128 nodes][512]
        float *out = split_partial + (size_t)blockIdx.x * kBM * kH;
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
        return;
    }
    // ---- dH tile -> 16-bit, through LDS so that rows leave as whole 16-byte pieces ----
    constexpr int SP = kQRow + 16;  // staging pitch
    static_assert(kBM * SP <= kLdsBytesDh, "staging tile");
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
        if (row < wg_rows)  // (grad_hidden is [B, maxT, U1, H] whatever the order of the nodes)
            *reinterpret_cast<u32x4 *>(reinterpret_cast<char *>(grad_hidden) +
                                       (size_t)compact_row(cmap, wg_row0 + row) * kQRow + ch * 16) = v;
    }
}

// dH rows of the row tiles whose vocabulary `vsplit` workgroups shared: their fp32 sums added in share order (a fixed
// order: bitwise reproducible), rounded once.  One thread per four columns of a row.
template <typename DT>
__global__ __launch_bounds__(256) void rnnt_fused_bwd_dh_merge(const float *__restrict__ split_partial, int vsplit,
                                                               long long row0, long long row_cap,
                                                               typename DT::elem *__restrict__ grad_hidden, Compact cmap) {
    const long long row = row0 + (long long)blockIdx.x * 2 + (threadIdx.x >> 7);
    if (row >= compact_total(cmap, row_cap)) return;
    const int col = (threadIdx.x & 127) * 4;
    const long long local = row - row0;
    const float *p = split_partial + ((size_t)(local / kBM) * vsplit * kBM + (size_t)(local % kBM)) * kH + col;
    f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = 0; i < vsplit; ++i) s += *reinterpret_cast<const f32x4 *>(p + (size_t)i * kBM * kH);
    typename DT::elem o[4] = {DT::from_float(s.x), DT::from_float(s.y), DT::from_float(s.z), DT::from_float(s.w)};
    *reinterpret_cast<uint2 *>(grad_hidden + (size_t)compact_row(cmap, row) * kH + col) = *reinterpret_cast<const uint2 *>(o);
}

// Compacted order: the rows of grad_hidden [B, maxT, U1, H] that are no lattice node (padding, or a node past the
// capacity of the logits buffer) are written by no workgroup of rnnt_fused_bwd_dh: zeros from here.  One thread looks
// one row of the tensor up; the wave then writes the rows that need it, one KiB per instruction (16 bytes per lane),
// so a batch without padding costs the lookups only.
template <typename DT>
__global__ __launch_bounds__(256) void rnnt_fused_bwd_zero_padding(typename DT::elem *__restrict__ grad_hidden,
                                                                   const NodeSpan *__restrict__ spans, int maxT, int U1,
                                                                   long long rows, long long row_cap) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    bool zero = false;
    if (r < rows) {
        const RowCoord c = decode_row(r, maxT, U1);
        const NodeSpan s = spans[c.b];
        zero = !(c.t < s.Tb && c.u < s.Ub1 && s.first + (long long)c.t * s.Ub1 + c.u < row_cap);
    }
    const int lane = threadIdx.x & (kWave - 1);
    unsigned long long todo = __ballot(zero);
    char *wave_row0 = reinterpret_cast<char *>(grad_hidden) + (size_t)(r - lane) * kQRow;
    while (todo) {  // wave-uniform
        const int k = __builtin_ctzll(todo);
        todo &= todo - 1;
        *reinterpret_cast<u32x4 *>(wave_row0 + (size_t)k * kQRow + lane * 16) = u32x4{0u, 0u, 0u, 0u};
    }
}

// ---------------------------------------------------------------------------------------------------------
// partial dW[v, :] = sum over this workgroup's lattice nodes of g[node, v] hidden[node, :]
// ---------------------------------------------------------------------------------------------------------
template <typename DT>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
void rnnt_fused_bwd_dw(const typename DT::elem *__restrict__ logits, const typename DT::elem *__restrict__ hidden,
                       const NodeG *__restrict__ coefs, float *__restrict__ partial, float *__restrict__ pbias, int V,
                       int ldv, int padded, int blank, long long row_cap, long long coef_rows, int nvb,
                       int steps_per_chunk, Compact cmap) {
    typedef typename Mma<DT>::frag frag;
    typedef Waits<2> W;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kLdsBytesDw];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 31, h = lane >> 5;
    const int vb = blockIdx.x % nvb;  // consecutive workgroups share a node range: its hidden rows stay in L2 / MALL
    const int rc = blockIdx.x / nvb;
    const int vmin = vb * kBM;
    // the last vocabulary block: runs on into the padding of padded rows, else overlaps its predecessor (as the
    // last tile of rnnt_fused_bwd_dh)
    const int vb0 = padded ? vmin : min(vmin, V - kBM);
    const bool masked = padded ? vmin + kBM > V : vb0 != vmin;
    // rows = lattice nodes in the order the logits are stored in.  Compacted order (lattice.h): the host cut the
    // node ranges for row_cap rows without knowing how many nodes the lattices hold; the ranges are re-cut here from
    // the count on the device, so that every workgroup gets an equal share of the nodes that exist.
    const long long total_rows = compact_total(cmap, row_cap);
    if (cmap.rowmap) {
        const long long steps = (total_rows + kBK - 1) / kBK;
        const int nchunks = (int)(gridDim.x / nvb);
        steps_per_chunk = (int)((steps + nchunks - 1) / nchunks);  // (<= the host's: total_rows <= row_cap)
    }
    const long long row_begin = (long long)rc * steps_per_chunk * kBK;
    const long long rows_here = total_rows - row_begin;
    const int nsteps = rows_here <= 0 ? 0 : (int)min((long long)steps_per_chunk, (rows_here + kBK - 1) / kBK);

    // ---- producer side.  Gradient tile in LDS: [32 nodes][128 v], 256-byte rows, 64-byte window w of node n at
    // window w ^ (n & 3).  Thread tid owns the 16 bytes at tid * 16 (DMA destination of its lane, converted in
    // place): node gr of the step, chunk gc (eight vocabulary entries from vcol).
    const int gr = tid >> 4;
    const int gc = ((((tid >> 2) & 3) ^ (gr & 3)) << 2) | (tid & 3);
    const int vcol = vb0 + 8 * gc;
    const int db = blank - vcol, keep_from = padded ? 0 : vmin - vcol, keep_until = padded ? V - vcol : 8;

    // step j's DMAs, k = 0 .. 3: hidden row k of this wave's four of step j + 2; k = 4: logits of step j + 4; k = 5: node
    // records of step j + 6.  The hidden vector of a node is row NodeG::row of `hidden` (the nodes are walked in the order
    // the logits are stored in, which in the compacted order is not the tensor's): hrow[k] holds those rows for the Q
    // tile this step fetches, wave-uniform, read from the records in LDS one step earlier (read_rows / take_rows).
    int hrow[kDmaPerStep];
    auto issue = [&](int j, int k) {
        if (k < kDmaPerStep) {
            if (j + 2 >= 0) {
                const int kr = wave * kDmaPerStep + k;
#if defined(FUSED_BWD_ABLATE_LOAD) || defined(FUSED_BWD_ARITH_ROWS)  // (timing builds: rows by arithmetic, dense order only)
                const long long row0 = row_begin + (long long)(j + 2 < nsteps ? j + 2 : 0) * kBK;
                hrow[k] = (int)(row0 + kr < total_rows ? row0 + kr : total_rows - 1);
#endif
#ifndef FUSED_BWD_ABLATE_DMA
                lds_dma16(dma_voff(lane, k & 3), reinterpret_cast<const char *>(hidden) + (size_t)(unsigned)hrow[k] * kQRow,
                          lds0 + kQ0 + ((j + 2) % kNQ) * kQBytes + kr * kQRow);
#endif
            }
        } else if (k == kDmaPerStep) {
#ifndef FUSED_BWD_ABLATE_LOAD
            // logits: this wave's piece = nodes 4 wave .. 4 wave + 3 of the step (past the end: the last node, ignored)
            const long long row0 = row_begin + (long long)(j + 4 < nsteps ? j + 4 : 0) * kBK;
            long long lrow = row0 + gr;
            lrow = lrow < total_rows ? lrow : total_rows - 1;
            const long long wrow0 = row0 + 4 * wave < total_rows ? row0 + 4 * wave : total_rows - 1;  // wave-uniform
            const unsigned voff = (unsigned)(((lrow - wrow0) * ldv + vcol) * 2);  // 0 .. 3 rows on, this lane's chunk
            lds_dma16(voff, reinterpret_cast<const char *>(logits) + (size_t)wrow0 * ldv * 2,
                      lds0 + kG0 + (unsigned)((j + 4) % kNG) * kGBytes + wave * 1024);
#endif
        } else {
#ifndef FUSED_BWD_ABLATE_LOAD
            // node records of this wave's four nodes: 128 bytes, fetched by lanes 0-31 (and again by lanes 32-63); past the
            // end of the range: the range's first records again (valid rows, never converted)
            const long long row0 = row_begin + (long long)(j + 6 < nsteps ? j + 6 : 0) * kBK;
            long long crow0 = row0 + 4 * wave;
            crow0 = crow0 + 4 <= coef_rows ? crow0 : coef_rows - 4;
            lds_dma4((unsigned)((lane & 31) * 4), reinterpret_cast<const char *>(coefs + crow0),
                     lds0 + kC0 + (unsigned)((j + 6) % kNC) * kCBytes + wave * 256);
#endif
        }
    };
    // lane k < 4: NodeG::row of node 4 wave + k of tile t, from the records in LDS (landed and behind a barrier)
    auto read_rows = [&](int t) {
        return *(const __attribute__((address_space(3))) int *)(uintptr_t)(lds0 + kC0 + (unsigned)(t % kNC) * kCBytes +
                                                                           wave * 256 + (lane & 3) * 32 + 24);
    };
    auto take_rows = [&](int v) {
#ifndef FUSED_BWD_ARITH_ROWS
#pragma unroll
        for (int k = 0; k < kDmaPerStep; ++k) hrow[k] = __builtin_amdgcn_readlane(v, k);
#endif
    };
    float bsum[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};  // this thread's share of the bias gradient
    float gx[8];
    float fsyk, fsbk;
    int fdy;
    auto conv_main = [&](int j) {
        const unsigned slot = (unsigned)(j % kNG);
        const unsigned ca = lds0 + kC0 + (unsigned)(j % kNC) * kCBytes + wave * 256 + (lane >> 4) * 32;
        const u32x4 c0 = *(lds_cu128)(uintptr_t)ca;
        const u32x2 c1 = *(const __attribute__((address_space(3))) u32x2 *)(uintptr_t)(ca + 16);
        u32x4 raw = *(lds_cu128)(uintptr_t)(lds0 + kG0 + slot * kGBytes + tid * 16);
        // padding nodes: zero coefficients (rnnt_bwd_node_coefs) and zero logits, whatever the tensor holds there
        const bool ok = (int)c1.y != kPadding;
        raw.x = ok ? raw.x : 0u;
        raw.y = ok ? raw.y : 0u;
        raw.z = ok ? raw.z : 0u;
        raw.w = ok ? raw.w : 0u;
        grad8_main<DT>(raw, __uint_as_float(c0.x), __uint_as_float(c0.y), __uint_as_float(c0.z), gx);
        fsyk = __uint_as_float(c0.w);
        fsbk = __uint_as_float(c1.x);
        fdy = (int)c1.y - vcol;
    };
    auto keep_gx = [&]() {
        asm volatile("" ::"v"(gx[0]), "v"(gx[1]), "v"(gx[2]), "v"(gx[3]), "v"(gx[4]), "v"(gx[5]), "v"(gx[6]), "v"(gx[7]),
                     "v"(fsyk), "v"(fsbk), "v"(fdy));
    };
    auto conv_finish = [&](int j) {
        grad8_fix(fsyk, fsbk, fdy, db, masked, keep_from, keep_until, gx);
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[e] += gx[e];
        *(lds_u128)(uintptr_t)(lds0 + kG0 + (j % kNG) * kGBytes + tid * 16) = DT::pack(gx);
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
    auto mma = [&](int j) {
        const unsigned gbuf = lds0 + kG0 + (j % kNG) * kGBytes, qbuf = lds0 + kQ0 + (j % kNQ) * kQBytes;
        unsigned ga[2], bq[4];
#pragma unroll
        for (int k = 0; k < 2; ++k) ga[k] = opaque(gbuf + aoff[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) bq[k] = opaque(qbuf + boff[k]);
        mma_step<DT, 2>(
            bq, acc, [&](int s, int i) { return lds_read_tr_frag<frag>(ga[i] + s * 16 * 256, 4 * 256); },
            [&]() { conv_main(j + 1); }, keep_gx, [&](int k) { issue(j, k); });
    };

    if (nsteps > 0) {  // workgroup-uniform; the pipeline of rnnt_fused_bwd_dh, with the records two steps further ahead
        for (int j = -6; j < 0; ++j) issue(j, kDmaPerStep + 1);
        for (int j = -4; j < 0; ++j) issue(j, kDmaPerStep);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        for (int j = -2; j < 0; ++j) {  // Q tiles 0 and 1, their rows from the records that have just landed
            take_rows(read_rows(j + 2));
#pragma unroll
            for (int k = 0; k < kDmaPerStep; ++k) issue(j, k);
        }
        take_rows(read_rows(2));
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        conv_main(0);
        conv_finish(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int j = 0; j < nsteps; ++j) {
            const int next_rows = read_rows(j + 3);  // (records of tile j + 3: landed since the end of step j - 1)
            mma(j);  // (and the DMAs of step j)
            if (j + 1 < nsteps) conv_finish(j + 1);
            take_rows(next_rows);
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(W::kStep) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
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
    for (int e = 0; e < 8; ++e) bl[gr * kBM + 8 * gc + e] = bsum[e];
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
                                                             float *__restrict__ grad_bias, int V, int padded, int nvb,
                                                             int nchunks) {
    const int v = blockIdx.x * 2 + (threadIdx.x >> 7);
    if (v >= V) return;
    const int col = (threadIdx.x & 127) * 4;
    const int last0 = V - kBM;  // first entry of the last (overlapping) block; it owns entries from (nvb-1)*128 on
    const int vb = v >= (nvb - 1) * kBM ? nvb - 1 : v / kBM;
    const int local = (vb == nvb - 1 && !padded) ? v - last0 : v - vb * kBM;
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
    // dH: one workgroup per CU at a time, so `row_tiles` take ceil(row_tiles / CUs) rounds and a last round with few
    // tiles costs as much as a full one (config 2: 4100 tiles on 256 CUs = 16 rounds + 4 tiles, +6 %).  The surplus
    // tiles (up to half a round) are launched apart, their vocabulary shared by as many workgroups as fill the round.
    // Planned for the 256 CUs of an MI355X whatever the device (the workspace size must depend on the shape only).
    long long dh_full;    // row tiles that walk the whole vocabulary
    int dh_surplus;       // row tiles whose vocabulary is shared ...
    int dh_vsplit;        // ... by this many workgroups each (1: no surplus launch)
};
constexpr int kPlanCUs = 256;
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
    const long long row_tiles = (rows + kBM - 1) / kBM;
    const int nt = (V + kBK - 1) / kBK;
    p.dh_full = row_tiles / kPlanCUs * kPlanCUs;
    p.dh_surplus = (int)(row_tiles - p.dh_full);
    p.dh_vsplit = 1;
    if (p.dh_surplus > 0 && p.dh_surplus <= kPlanCUs / 2) {
        p.dh_vsplit = kPlanCUs / p.dh_surplus;
        if (p.dh_vsplit > nt / 2) p.dh_vsplit = nt / 2;  // at least two tiles per share
    }
    if (p.dh_vsplit < 2) {
        p.dh_full = row_tiles;
        p.dh_surplus = 0;
        p.dh_vsplit = 1;
    }
    return p;
}
inline size_t align256(size_t n) { return (n + 255) / 256 * 256; }

struct BwdWorkspace {
    NodeG *coefs;
    float *partial;
    float *pbias;
    float *dh_partial;  // [dh_surplus * dh_vsplit][128][512] fp32
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
    w.dh_partial = reinterpret_cast<float *>(c + off);
    off += align256((size_t)p.dh_surplus * p.dh_vsplit * kBM * kH * sizeof(float));
    w.bytes = off;
    return w;
}

template <typename DT>
rnntStatus_t launch_bwd(const void *hidden, const void *weight, const void *logits, const int *labels,
                        const int *act_lens, const int *label_lens, int V, int ldv, int B, const float *grad_out,
                        int grad_out_stride, float grad_scale, void *grad_hidden, void *grad_weight, float *grad_bias,
                        const Lattice &L, void *bwd_ws, const rnntOptions &o, long long row_cap, const int *rowmap,
                        bool zero_padding_rows) {
    typedef typename DT::elem E;
    const int padded = ldv >= (V + kBM - 1) / kBM * kBM;  // every tile of every row may be read in full
    // rows the grids are cut for: every row of the tensor, or (compacted order) what the logits buffer holds
    const long long tensor_rows = (long long)B * o.maxT * o.maxU;
    const long long rows = rowmap && row_cap < tensor_rows ? row_cap : tensor_rows;
    const Compact cmap = {rowmap, L.spans, B};
    const DwPlan p = dw_plan(rows, V);
    const BwdWorkspace w = carve_bwd(bwd_ws, p);
    hipStream_t st = static_cast<hipStream_t>(o.stream);
    const long long row_tiles = (rows + kBM - 1) / kBM;
    if (row_tiles > 0x7fffffffLL || (p.coef_rows + 255) / 256 > 0x7fffffffLL) return RNNT_STATUS_INVALID_VALUE;
    hipLaunchKernelGGL(rnnt_bwd_node_coefs, dim3((unsigned)((p.coef_rows + 255) / 256)), dim3(256), 0, st, w.coefs, labels,
                       act_lens, label_lens, L, grad_out, grad_out_stride, grad_scale, o.maxT, o.maxU, rows, p.coef_rows, cmap);
    if (o.stage_events && o.stage_events[1]) (void)hipEventRecord(static_cast<hipEvent_t>(o.stage_events[1]), st);
    if (grad_hidden) {
        if (rowmap && zero_padding_rows)
            hipLaunchKernelGGL((rnnt_fused_bwd_zero_padding<DT>), dim3((unsigned)((tensor_rows + 255) / 256)), dim3(256), 0, st,
                               static_cast<E *>(grad_hidden), L.spans, o.maxT, o.maxU, tensor_rows, rows);
#define DH_LAUNCH(NWG, TILE0, SPLIT)                                                                              \
    hipLaunchKernelGGL((rnnt_fused_bwd_dh<DT>), dim3((unsigned)(NWG)), dim3(kThreads), 0, st,                      \
                       static_cast<const E *>(logits), static_cast<const E *>(weight), w.coefs,                   \
                       static_cast<E *>(grad_hidden), V, ldv, padded, o.blank_label, rows, (long long)(TILE0),    \
                       (int)(SPLIT), w.dh_partial, cmap)
        if (p.dh_full > 0) DH_LAUNCH(p.dh_full, 0, 1);
        if (p.dh_surplus > 0) {
            DH_LAUNCH((long long)p.dh_surplus * p.dh_vsplit, p.dh_full, p.dh_vsplit);
            const long long row0 = p.dh_full * kBM, n = rows - row0;
            hipLaunchKernelGGL((rnnt_fused_bwd_dh_merge<DT>), dim3((unsigned)((n + 1) / 2)), dim3(256), 0, st, w.dh_partial,
                               p.dh_vsplit, row0, rows, static_cast<E *>(grad_hidden), cmap);
        }
#undef DH_LAUNCH
    }
    if (o.stage_events && o.stage_events[2]) (void)hipEventRecord(static_cast<hipEvent_t>(o.stage_events[2]), st);
    if (grad_weight || grad_bias) {
        hipLaunchKernelGGL((rnnt_fused_bwd_dw<DT>), dim3((unsigned)(p.nchunks * p.nvb)), dim3(kThreads), 0, st,
                           static_cast<const E *>(logits), static_cast<const E *>(hidden), w.coefs, w.partial, w.pbias, V,
                           ldv, padded, o.blank_label, rows, p.coef_rows, p.nvb, p.steps_per_chunk, cmap);
        hipLaunchKernelGGL((rnnt_fused_bwd_reduce<DT>), dim3((unsigned)((V + 1) / 2)), dim3(256), 0, st, w.partial,
                           w.pbias, static_cast<E *>(grad_weight), grad_bias, V, padded, p.nvb, p.nchunks);
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

rnntStatus_t rnnt_joint_loss_backward_compact_workspace_size(int hidden_size, int alphabet_size, long long logits_rows,
                                                             size_t *size_bytes) {
    if (!size_bytes || hidden_size != kH || alphabet_size < kBM || logits_rows <= 0) return RNNT_STATUS_INVALID_VALUE;
    *size_bytes = carve_bwd(nullptr, dw_plan(logits_rows, alphabet_size)).bytes;
    return RNNT_STATUS_SUCCESS;
}

static rnntStatus_t joint_loss_backward(const void *hidden, const void *weight, const void *logits,
                                        int logits_row_stride, long long logits_rows, const void *compact_workspace,
                                        const int *labels, const int *act_lens, const int *label_lens, int hidden_size,
                                        int alphabet_size, int minibatch, const float *grad_out, int grad_out_stride,
                                        float grad_scale, void *grad_hidden, void *grad_weight, float *grad_bias,
                                        const void *workspace, void *backward_workspace, rnntOptions options,
                                        bool zero_padding_rows = true) {
    if (!hidden || !weight || !logits || !act_lens || !label_lens || !workspace || !backward_workspace)
        return RNNT_STATUS_INVALID_VALUE;
    if (!labels && options.maxU > 1) return RNNT_STATUS_INVALID_VALUE;
    if (hidden_size != kH || alphabet_size < kBM) return RNNT_STATUS_INVALID_VALUE;
    // every row of the logits starts on 16 bytes (LDS-DMA); rows without padding up to a multiple of 128 entries are
    // read in overlapping tiles, whose starts must be 16-byte aligned too
    if (logits_row_stride < alphabet_size || logits_row_stride % 8 != 0) return RNNT_STATUS_INVALID_VALUE;
    if (logits_row_stride < (alphabet_size + kBM - 1) / kBM * kBM && alphabet_size % 8 != 0) return RNNT_STATUS_INVALID_VALUE;
    if (minibatch <= 0 || options.maxT <= 0 || options.maxU <= 0 || options.maxU > RNNT_MAX_LATTICE_COLUMNS)
        return RNNT_STATUS_INVALID_VALUE;
    if (options.blank_label < 0 || options.blank_label >= alphabet_size) return RNNT_STATUS_INVALID_VALUE;
    if (options.dtype != RNNT_DTYPE_BF16 && options.dtype != RNNT_DTYPE_F16) return RNNT_STATUS_INVALID_VALUE;
    if (reinterpret_cast<uintptr_t>(workspace) % 256 != 0 || reinterpret_cast<uintptr_t>(backward_workspace) % 256 != 0 ||
        reinterpret_cast<uintptr_t>(hidden) % 16 != 0 || reinterpret_cast<uintptr_t>(weight) % 16 != 0 ||
        reinterpret_cast<uintptr_t>(logits) % 16 != 0 || reinterpret_cast<uintptr_t>(grad_hidden) % 16 != 0 ||
        reinterpret_cast<uintptr_t>(grad_weight) % 8 != 0 || reinterpret_cast<uintptr_t>(grad_bias) % 4 != 0)
        return RNNT_STATUS_INVALID_VALUE;
    if ((size_t)kBM * logits_row_stride * 2 > 0x7fffffffu) return RNNT_STATUS_INVALID_VALUE;  // 32-bit offsets per row tile
    const Lattice w = carve_workspace(const_cast<void *>(workspace), options.maxT, options.maxU, minibatch);
    if (options.stage_events && options.stage_events[0])
        (void)hipEventRecord(static_cast<hipEvent_t>(options.stage_events[0]), static_cast<hipStream_t>(options.stream));
    return options.dtype == RNNT_DTYPE_BF16
               ? launch_bwd<BF16>(hidden, weight, logits, labels, act_lens, label_lens, alphabet_size, logits_row_stride, minibatch, grad_out,
                                  grad_out_stride, grad_scale, grad_hidden, grad_weight, grad_bias, w, backward_workspace,
                                  options, logits_rows, static_cast<const int *>(compact_workspace), zero_padding_rows)
               : launch_bwd<F16>(hidden, weight, logits, labels, act_lens, label_lens, alphabet_size, logits_row_stride, minibatch, grad_out,
                                 grad_out_stride, grad_scale, grad_hidden, grad_weight, grad_bias, w, backward_workspace,
                                 options, logits_rows, static_cast<const int *>(compact_workspace), zero_padding_rows);
}

rnntStatus_t rnnt_joint_loss_backward_strided(const void *hidden, const void *weight, const void *logits,
                                              int logits_row_stride, const int *labels, const int *act_lens,
                                              const int *label_lens, int hidden_size, int alphabet_size, int minibatch,
                                              const float *grad_out, int grad_out_stride, float grad_scale,
                                              void *grad_hidden, void *grad_weight, float *grad_bias,
                                              const void *workspace, void *backward_workspace, rnntOptions options) {
    return joint_loss_backward(hidden, weight, logits, logits_row_stride, 0, nullptr, labels, act_lens, label_lens,
                               hidden_size, alphabet_size, minibatch, grad_out, grad_out_stride, grad_scale, grad_hidden,
                               grad_weight, grad_bias, workspace, backward_workspace, options);
}

rnntStatus_t rnnt_joint_loss_backward_compact(const void *hidden, const void *weight, const void *logits,
                                              int logits_row_stride, long long logits_rows, const int *labels,
                                              const int *act_lens, const int *label_lens, int hidden_size,
                                              int alphabet_size, int minibatch, const float *grad_out, int grad_out_stride,
                                              float grad_scale, void *grad_hidden, int zero_padding_rows, void *grad_weight,
                                              float *grad_bias, const void *workspace, const void *compact_workspace,
                                              void *backward_workspace, rnntOptions options) {
    if (!compact_workspace || reinterpret_cast<uintptr_t>(compact_workspace) % 256 != 0 || logits_rows <= 0)
        return RNNT_STATUS_INVALID_VALUE;
    if ((long long)minibatch * options.maxT * options.maxU > 0x7fffffffLL) return RNNT_STATUS_INVALID_VALUE;
    return joint_loss_backward(hidden, weight, logits, logits_row_stride, logits_rows, compact_workspace, labels, act_lens,
                               label_lens, hidden_size, alphabet_size, minibatch, grad_out, grad_out_stride, grad_scale,
                               grad_hidden, grad_weight, grad_bias, workspace, backward_workspace, options,
                               zero_padding_rows != 0);
}


rnntStatus_t rnnt_joint_loss_backward(const void *hidden, const void *weight, const void *logits, const int *labels,
                                      const int *act_lens, const int *label_lens, int hidden_size, int alphabet_size,
                                      int minibatch, const float *grad_out, int grad_out_stride, float grad_scale,
                                      void *grad_hidden, void *grad_weight, float *grad_bias, const void *workspace,
                                      void *backward_workspace, rnntOptions options) {
    return rnnt_joint_loss_backward_strided(hidden, weight, logits, alphabet_size, labels, act_lens, label_lens,
                                            hidden_size, alphabet_size, minibatch, grad_out, grad_out_stride, grad_scale,
                                            grad_hidden, grad_weight, grad_bias, workspace, backward_workspace, options);
}

}  // extern "C"

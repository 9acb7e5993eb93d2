// rnnt_mi355.hip — MI355X (gfx950 / CDNA4) RNN-Transducer loss: kernels + C ABI.
//
// Replaces the native code under warprnnt_pytorch.RNNTLoss, the one call the
// reference makes at /root/reference/src/rnnt/model.py:116.
// Written for wave64 / gfx950 only; no CUDA, no portability layer.
//
// Three kernels, three passes over the [B,T,U+1,V] logits in total:
//   K1  rnnt_k1_lse      one wavefront per lattice node: the V logits of the node are held
//                        in registers (16-byte coalesced loads), max and sum-exp are reduced
//                        across the 64 lanes, and only lse / lp_blank / lp_label are kept.
//   K2  rnnt_k2_lattice  one workgroup per utterance: alpha waves walk the T+U anti-diagonals
//                        upwards, beta waves downwards; lane = u, the (t,u-1)/(t,u+1)
//                        neighbour arrives by a one-lane DPP wave shift, (t-1,u)/(t+1,u) is
//                        the lane's own previous value.  No global sync, no atomics.
//   K3  rnnt_k3_grad     one wavefront per node again: reads the logits a second time and
//                        writes d loss / d logits, scaled by grad_out and 1/B, zeros in padding.
//
// Lattice arrays live in a SKEWED layout: node (b,t,u) is element
//   (b*ND + (t+u))*U1 + u,   ND = maxT + U1 - 1, U1 = maxU
// so that one anti-diagonal is contiguous and K2's loads/stores are coalesced.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>

#include "../../include/rnnt_mi355.h"
#include "elem_types.h"
#include "lattice.h"

namespace {

using namespace rnnt_mi355_detail;

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr int kAbiVersion = 1;

// Word 3 of a raw (stride 0, byte-offset) buffer descriptor on gfx90a/gfx942/gfx950: 32-bit data
// format, offsets checked against num_records; and the `nt` bit of a buffer load's cache policy.
constexpr int kRawBufferFlags = 0x00020000;
constexpr int kBufferNontemporal = 2;

// Descriptor of `bytes` bytes at p; both must be the same in every lane of the wave (it lives in SGPRs).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_buffer(const void *p, int bytes) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uintptr_t)hi << 32) | lo), (short)0,
                                             __builtin_amdgcn_readfirstlane(bytes), kRawBufferFlags);
}

// 16-byte chunks between the previous 128-byte boundary and p
__device__ __forceinline__ int row_misalign(const void *p) {
    return (int)((reinterpret_cast<uintptr_t>(p) >> 4) & 7);
}

// Previous 16-byte boundary at or below p.  Pointer arithmetic, not an integer mask: a pointer
// rebuilt from a masked integer loses its address space and every access through it becomes a
// flat_load / flat_store the compiler waits on one at a time.
template <typename T>
__device__ __forceinline__ const u32x4 *align_down16(const T *p) {
    return reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(p) - (reinterpret_cast<uintptr_t>(p) & 15));
}
template <typename T>
__device__ __forceinline__ u32x4 *align_down16(T *p) {
    return reinterpret_cast<u32x4 *>(reinterpret_cast<char *>(p) - (reinterpret_cast<uintptr_t>(p) & 15));
}

// Rows that are not 16-byte aligned (V*elem % 16 != 0, e.g. the reference's V = 4234 and 5455):
// the wave still moves whole aligned 16-byte chunks; `s` elements of the previous row share the
// first chunk and some of the next row share the last one.  Only those two chunks need care.
template <typename DT>
__device__ __forceinline__ u32x4 mask_outside_row(const u32x4 &raw, int first, int V) {
    // elements whose row index first+j is outside [0,V) become -inf (max and sum-exp ignore them)
    float x[DT::EPC];
    DT::unpack(raw, x);
#pragma unroll
    for (int j = 0; j < DT::EPC; ++j)
        if ((unsigned)(first + j) >= (unsigned)V) x[j] = -INFINITY;
    return DT::pack(x);
}
template <typename DT>
__device__ __forceinline__ void store_inside_row(const u32x4 &v, u32x4 *p, int first, int V) {
    // element-wise stores of the part of a boundary chunk that belongs to this row (plain integer
    // moves: going through an array of the element type made hipcc spill hundreds of registers for fp16)
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    if constexpr (sizeof(typename DT::elem) == 4) {
        unsigned *o = reinterpret_cast<unsigned *>(p);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((unsigned)(first + j) < (unsigned)V) o[j] = w[j];
    } else {
        unsigned short *o = reinterpret_cast<unsigned short *>(p);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if ((unsigned)(first + j) < (unsigned)V) o[j] = (unsigned short)(w[j >> 1] >> (16 * (j & 1)));
    }
}

// ---------------------------------------------------------------------------------------------
// K1: log-softmax denominators; keep blank and next-label log-probs.  
// Bound: HBM read of N_valid elements.  WPR waves per node (1 for V*elem <= 17 KiB), each wave
// holds NCH 16-byte chunks per lane in registers; max and sum-exp are reduced with wave
// butterflies and, for WPR > 1, a two-word LDS exchange.
// ---------------------------------------------------------------------------------------------
// Waves per SIMD that stream fastest, by bytes a wave keeps in flight (NCH KiB).  Measured on
// config 2 (NCH = 17): K1 reads 6.91 TB/s at 3 waves/SIMD against 6.76 at 4 and 6.1 at 2; K3 runs
// 3.07 ms at 2 waves/SIMD (one 512-thread workgroup per CU) against 3.17 at 4.  Fewer, larger
// bursts per CU suit HBM better than maximum occupancy; smaller rows get more waves to keep
// roughly the same bytes in flight per CU.
// epc = elements per 16-byte chunk.  16-bit logits carry twice the VALU work per byte: K1 then wants
// one more wave per SIMD (config 5: 18.8 -> 17.9 ms); K3 still runs best at 2 (38.2 vs 40.0 at 3).
// The attribute below pins BOTH bounds (min = max = N), i.e. it also limits the kernel to 512/N
// VGPRs: N must not exceed what the instantiation reaches without spilling (a cap of 8 on a kernel
// that wants 91 registers costs 179 spills; fp16 converts need more registers than bf16 shifts).
// `cvt` = the element type unpacks through conversions (fp16).  The spill-free bounds below come from
// the -Rpass-analysis=kernel-resource-usage report of a build without the attribute
// (`make -C rnnt_amd/csrc resources`); raising N beyond them makes hipcc spill, lowering some of
// them (K3 NCH 6/8 to 4/3) flips its scheduler into a 256-VGPR schedule that runs 30x slower.
[[maybe_unused]] constexpr int k1_waves(int nch, int epc, bool mis, int wpr, bool cvt) {
    int w = nch >= 16 ? (epc > 4 ? 4 : 3) : nch >= 12 ? 4 : nch >= 8 ? 6 : 8;
    // fp16 rows with boundary masks, 16+ chunks per lane in one wave: 132-140 VGPRs, a few more than 4
    // waves allow (V=8201: K1 0.79 ms at 3 waves, 0.99 ms with 4-15 spilled registers at 4)
    if (mis && cvt && nch >= 16 && wpr == 1) w = 3;
    if (mis && cvt && nch == 10 && wpr == 1) w = 5;  // 13 spilled registers at 6
    // 16-bit aligned rows with 10 chunks per lane (bf16 V=4232): 3 / 4 / 5 / 6 / 7 waves measured 0.93 / 0.82 / 0.74 /
    // 0.70 / 0.69 ms (fp16 0.93 / 0.83 / 0.76 / 0.72 / 0.71; bf16 V=5000 1.00 / 0.88 / 0.82 / 0.81 / 0.82): stays at 6
    return w;
}
[[maybe_unused]] constexpr int k3_waves(int nch, int epc, bool mis) {
    // up to 4 chunks per lane every type fits 64 VGPRs since short rows are read through a buffer
    // descriptor (fp16 V=1000: 1.69 -> 1.46 ms at 8 waves); 6 chunks per lane measured 1-2 % better
    // at 5 waves than at 6
    // 16-bit aligned rows of 9-12 chunks per lane (bf16/fp16 V = 4232 ... 6000): 2 waves measured 1.595 -> 1.546 ms
    // (bf16 V=4232), 1.583 -> 1.513 (fp16), 1.845 -> 1.750 (fp16 V=5000); fp32 (V=2600, 3000) and rows that are
    // not 16-byte aligned (bf16 V=4234: 1.555 -> 1.623) keep 3; 4 and 5 waves spill (1.90 / 2.60 ms)
    if (nch == 12 && epc > 4 && !mis) return 2;
    return nch >= 16 ? 2 : nch >= 12 ? 3 : nch >= 8 ? 4 : nch >= 6 ? 5 : 8;
}
template <typename DT>
constexpr bool kCvtUnpack = sizeof(typename DT::elem) == 2 && !__is_same(typename DT::elem, __bf16);

// fp16 only: the conversions of the first pass otherwise get scheduled (or merged, when both run
// under the same lane mask) right behind each load, and every load is waited for before the next
// one is issued.  An empty asm that names all chunks keeps the loads together.
template <int N>
__device__ __forceinline__ void loads_issued(u32x4 (&raw)[N]) {
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("" : "+v"(raw[k].x), "+v"(raw[k].y), "+v"(raw[k].z), "+v"(raw[k].w));
}

// ---------------------------------------------------------------------------------------------
// Compacted node order for K1.  A ragged batch (config 3: T 100-800, U 10-60 in a tensor cut for 800 x 61)
// has two thirds of its rows in padding; dealt one node per wave in MEMORY order, two thirds of K1's
// workgroups found nothing to read and left, and the bytes in flight thinned out (5.9 TB/s against 6.9 on a
// uniform batch).  K1 therefore walks the nodes INSIDE the lattices only: one wave of rnnt_node_spans turns the
// length tensors into per-utterance prefix offsets on the device, and K1 maps its node counter through them,
// so consecutive workgroups stream consecutive valid rows and the surplus workgroups form an empty tail.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void rnnt_node_spans(const int *__restrict__ act_lens,
                                                      const int *__restrict__ label_lens, int B, int maxT,
                                                      int U1, NodeSpan *__restrict__ spans) {
    const int lane = threadIdx.x;
    long long carry = 0;
    for (int b0 = 0; b0 < B; b0 += kWave) {
        const int b = b0 + lane;
        int Tb = 0, Ub1 = 1;
        if (b < B) {
            const int t = act_lens[b], u = label_lens[b];
            if (lens_ok(t, u, maxT, U1)) {
                Tb = t;
                Ub1 = u + 1;
            }
        }
        const long long n = (long long)Tb * Ub1;
        long long inc = n;  // inclusive scan over the wave
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const long long o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (b < B) {
            NodeSpan e;
            e.first = carry + inc - n;
            e.Tb = Tb;
            e.Ub1 = Ub1;
            spans[b] = e;
        }
        carry += __shfl(inc, kWave - 1);
    }
    if (lane == 0) {
        NodeSpan e;
        e.first = carry;
        e.Tb = 0;
        e.Ub1 = 1;
        spans[B] = e;
    }
}

// rowmap[k] = row (b * maxT + t) * U1 + u of the [B, maxT, U1] tensor of compacted node k (lattice.h: Compact), from
// the spans above; one thread per row of the tensor.  A node past `cap` (the caller's buffer of per-node rows holds
// fewer rows than there are nodes) cannot be computed: its lattice entries are set to NaN (where the planes are given),
// so that its utterance's cost comes out NaN instead of being formed from whatever the workspace held.
__global__ __launch_bounds__(256) void rnnt_compact_rows_kernel(int *__restrict__ rowmap, const NodeSpan *__restrict__ spans,
                                                                float2 *__restrict__ ml, float2 *__restrict__ lp, int ND,
                                                                int US, int maxT, int U1, long long rows, long long cap) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const RowCoord c = decode_row(r, maxT, U1);
    const NodeSpan s = spans[c.b];
    if (c.t >= s.Tb || c.u >= s.Ub1) return;  // padding (an utterance whose lengths do not fit has Tb = 0)
    const long long k = s.first + (long long)c.t * s.Ub1 + c.u;
    if (k < cap) {
        if (rowmap) rowmap[k] = (int)r;
        return;
    }
    if (ml) {
        const size_t o = skew_index(c.b, c.t, c.u, ND, US);
        const float nan = __builtin_nanf("");
        ml[o] = make_float2(nan, nan);
        lp[o] = make_float2(nan, nan);
    }
}

struct Node {
    int b, t, u;  // position
    int Tb, Ub;   // lengths of its utterance
};
// Node number k < total (wave-uniform) of the compacted order.  The search starts at utterance k / (maxT*U1)
// — no utterance holds more nodes than that, so it is a lower bound, and the exact answer on a uniform batch
// — and looks at 64 spans per step, one per lane (one memory round trip, what the length lookup cost before).
// Steps overlap by one span: lane 0 always starts at or before k, so the ballot is never empty.
__device__ __forceinline__ Node find_node(const NodeSpan *__restrict__ spans, int B, long long k, long long per_b,
                                          int lane) {
    Node n = {0, 0, 0, 0, 0};
    for (int b0 = (int)(k / per_b); b0 < B; b0 += kWave - 1) {
        // one 16-byte load per lane (field by field hipcc fetched Ub1 in a second round trip after the ballot);
        // lanes past the end see the total (> k) again
        const u32x4 e = *reinterpret_cast<const u32x4 *>(spans + min(b0 + lane, B));
        const long long e_first = (long long)(((unsigned long long)e.y << 32) | e.x);
        const unsigned long long below = __ballot(e_first <= k);
        if (below == ~0ull) continue;
        const int src = __popcll(below) - 1;  // last span that starts at or before k: empty ones in front of it are skipped
        const long long first = ((long long)__builtin_amdgcn_readlane((int)e.y, src) << 32) |
                                (unsigned)__builtin_amdgcn_readlane((int)e.x, src);
        n.b = b0 + src;
        n.Tb = __builtin_amdgcn_readlane((int)e.z, src);
        const int Ub1 = __builtin_amdgcn_readlane((int)e.w, src);
        n.Ub = Ub1 - 1;
        const int rem = (int)(k - first);
        n.t = rem / Ub1;
        n.u = rem - n.t * Ub1;
        break;
    }
    return n;
}

template <typename DT, int NCH, int WPR, bool MIS = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(k1_waves(NCH, DT::EPC, MIS, WPR, kCvtUnpack<DT>), k1_waves(NCH, DT::EPC, MIS, WPR, kCvtUnpack<DT>))))
void rnnt_k1_lse(const typename DT::elem *__restrict__ acts,
                                                   const int *__restrict__ labels,
                                                   const int *__restrict__ act_lens,
                                                   const int *__restrict__ label_lens, Lattice L,
                                                   int maxT, int U1, int V, int blank,
                                                   long long total_rows, int B, long long compact_rows) {
    constexpr int EPC = DT::EPC;
    constexpr int WPB = 4;          // waves per workgroup (host launches 256 threads)
    constexpr int RPB = WPB / WPR;  // nodes per workgroup
    __shared__ float red[2][WPB];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;  // (left in a VGPR: forcing it scalar measured 1.6 % slower here)
    const int rib = wave / WPR;
    const int seg = wave - rib * WPR;
    // !MIS: host guarantees V % EPC == 0, 16-byte aligned rows and V/EPC <= WPR*NCH*64 - 7
    // nodes inside the lattices (a scalar load: the surplus workgroups of a ragged batch leave at once)
    // compact_rows > 0: `acts` holds one row per NODE, in this order (rnnt_forward_compact), that many rows
    const long long total = min(min(L.spans[B].first, total_rows), compact_rows ? compact_rows : total_rows);
    const long long ngroups = (total + RPB - 1) / RPB;
    const long long per_b = (long long)maxT * U1;

    for (long long g = blockIdx.x; g < ngroups; g += gridDim.x) {  // host launches one block per group of the FULL tensor
        const long long k = g * RPB + rib;
        const bool valid = k < total;  // padding is never visited
        if (WPR == 1 && !valid) break;
        const Node c = find_node(L.spans, B, valid ? k : 0, per_b, lane);
        const int Ub = c.Ub;
        const long long r = compact_rows ? (valid ? k : 0) : ((long long)c.b * maxT + c.t) * U1 + c.u;

        const typename DT::elem *row = acts + (size_t)r * (size_t)V;
        const uintptr_t addr = reinterpret_cast<uintptr_t>(row);
        const int shift = MIS ? (int)((addr & 15) / sizeof(typename DT::elem)) : 0;  // previous row's elements in chunk 0
        const u32x4 *row4 = MIS ? align_down16(row) : reinterpret_cast<const u32x4 *>(row);
        const int nchunks = MIS ? (shift + V + EPC - 1) / EPC : V / EPC;
        // Chunk index of lane 0 is shifted back so that every wave-instruction covers one
        // 128-byte-aligned KiB of memory whatever the row's own alignment (rows are only 16-byte
        // aligned: V*elem need not be a multiple of 128).
        const int c0 = seg * NCH * kWave + lane - row_misalign(row4);
        u32x4 raw[NCH];
        // Buffer loads bounded to the row: lanes before its first or past its last chunk get zeros
        // back without a memory request and without a branch.  (With each load under its own lane
        // mask hipcc merged it with the max pass below and waited for one load before issuing the
        // next in the fp16 and in every misaligned instantiation: 3.6 TB/s instead of 5.4.)
        // The descriptor has to sit in SGPRs; the row is the same for the whole wave.
        {
            const __amdgpu_buffer_rsrc_t rsrc = row_buffer(row4, valid ? nchunks * 16 : 0);
#pragma unroll
            for (int k = 0; k < NCH; ++k)
                raw[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (c0 + k * kWave) * 16, 0, kBufferNontemporal);
        }
        if (MIS) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int idx = c0 + k * kWave;
                if (valid && (idx == 0 || idx == nchunks - 1))  // at most two lanes of the wave
                    raw[k] = mask_outside_row<DT>(raw[k], idx * EPC - shift, V);
            }
        }
        // blank / label logits: tiny extra loads that hit the lines fetched above
        const int y = (valid && c.u < Ub) ? labels[(size_t)c.b * (U1 - 1) + c.u] : blank;
        const float xb = DT::to_float(row[blank]);
        const float xy = (y >= 0 && y < V) ? DT::to_float(row[y]) : __builtin_nanf("");

        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (valid && c0 + k * kWave >= 0 && c0 + k * kWave < nchunks) {
                float x[EPC];
                DT::unpack(raw[k], x);
#pragma unroll
                for (int j = 0; j < EPC; ++j) m = fmaxf(m, x[j]);
            }
        }
        m = wave_max(m);
        if (EPC > 4) {
            // 16-bit logits: make `raw` opaque here so the second pass re-unpacks it (two shifts)
            // instead of keeping 8 floats per chunk alive across the reduction (2x the registers).
#pragma unroll
            for (int k = 0; k < NCH; ++k)
                asm volatile("" : "+v"(raw[k].x), "+v"(raw[k].y), "+v"(raw[k].z), "+v"(raw[k].w));
        }
        if (WPR > 1) {
            red[0][wave] = m;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < WPR; ++i) m = fmaxf(m, red[0][rib * WPR + i]);
        }
        // all -inf row or +inf max: avoid inf-inf in the sum; torch.log_softmax returns NaN for the whole
        // row in both cases, and so does this (`logs` below), so the cost is NaN (bin/train.py:36-41 skips it)
        const float mm = (m == -INFINITY || m == INFINITY) ? 0.0f : m;
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (valid && c0 + k * kWave >= 0 && c0 + k * kWave < nchunks) {
                float x[EPC];
                DT::unpack(raw[k], x);
#pragma unroll
                for (int j = 0; j < EPC; ++j) s += __builtin_amdgcn_exp2f((x[j] - mm) * kLog2e);  // subtract first:
                // x*log2e - max*log2e would lose ulp(|x|*log2e) when the logits carry a large common offset
            }
        }
        s = wave_sum(s);
        if (WPR > 1) {
            red[1][wave] = s;
            __syncthreads();
            s = 0.0f;
#pragma unroll
            for (int i = 0; i < WPR; ++i) s += red[1][rib * WPR + i];  // fixed order: reproducible
        }
        if (valid && seg == 0 && lane == 0) {
            const float logs = row_log_sum(s, m);
            const size_t o = skew_index(c.b, c.t, c.u, L.ND, L.US);
            L.ml[o] = make_float2(mm, logs);
            // same association as torch.log_softmax on the upstream CPU path: (x - max) - log(sum)
            L.lp[o] = make_float2((xb - mm) - logs, (c.u < Ub) ? (xy - mm) - logs : -INFINITY);  // no emit arc at u = U
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Small rows (up to 128 chunks of 16 bytes, i.e. V <= 512 fp32 / 1024 16-bit; char and small
// word-piece vocabularies).  One wave per node is latency-bound there: a wave has one short row in flight and
// spends ~7k cycles on the dependent chain lengths -> row -> reductions -> store.  These variants
// give a wave G*NR consecutive nodes per iteration: LPR lanes per node (G = 64/LPR nodes side by
// side, NCH chunks per lane) times NR nodes per lane group, all loads issued before the first
// reduction.  Rows may start at any element offset of a 16-byte chunk (V = 29: 116-byte rows); boundary chunks are
// masked on the way in and stored by element on the way out, exactly as in the MIS variants above.
// ---------------------------------------------------------------------------------------------
struct SmallRow {
    int b, t, u;     // node
    int Tb, Ub;      // lengths of its utterance
    bool valid;      // inside the tensor AND inside the utterance's lattice
    bool exists;     // inside the tensor (padding nodes exist and get zero gradients)
};
// node of row (r0 + o) given the decoded node of r0 (wave-uniform) and a per-lane offset o < 2^16
// (x + 0.5) * (1/d) truncated is x / d exactly while x < 2^22 (the half keeps the product away from
// every integer by 0.5/d, far more than the fp32 rounding of it): maxT and maxU are far below that
__device__ __forceinline__ int small_div(int x, float inv_d) { return (int)(((float)x + 0.5f) * inv_d); }
// The NR nodes of one lane group.  All length loads are issued before the first is used (loads
// return in order: interleaved with their uses they cost one memory round trip per node).
template <int NR, int G>
__device__ __forceinline__ void small_rows(SmallRow (&s)[NR], const RowCoord &c0, int grp, long long r0,
                                           long long total_rows, const int *__restrict__ act_lens,
                                           const int *__restrict__ label_lens, int maxT, int U1, int B,
                                           float inv_T, float inv_U1) {
#pragma unroll
    for (int n = 0; n < NR; ++n) {
        const int o = n * G + grp;
        const int uu = c0.u + o;
        const int dt = small_div(uu, inv_U1);
        s[n].u = uu - dt * U1;
        const int tt = c0.t + dt;
        const int db = small_div(tt, inv_T);
        s[n].t = tt - db * maxT;
        s[n].b = c0.b + db;
        s[n].exists = r0 + o < total_rows;
    }
#pragma unroll
    for (int n = 0; n < NR; ++n) {
        const int bb = s[n].b < B ? s[n].b : B - 1;
        s[n].Tb = act_lens[bb];
        s[n].Ub = label_lens[bb];
    }
#pragma unroll
    for (int n = 0; n < NR; ++n)
        s[n].valid = s[n].exists && lens_ok(s[n].Tb, s[n].Ub, maxT, U1) && s[n].t < s[n].Tb && s[n].u <= s[n].Ub;
}

template <typename DT, int LPR, int NCH, int NR, bool MIS>
__global__ __launch_bounds__(256) void rnnt_k1_lse_small(
    const typename DT::elem *__restrict__ acts, const int *__restrict__ labels,
    const int *__restrict__ act_lens, const int *__restrict__ label_lens, Lattice L, int maxT, int U1,
    int V, int blank, long long total_rows, int B) {
    constexpr int EPC = DT::EPC;
    constexpr int G = kWave / LPR;
    constexpr int RPW = G * NR;  // nodes per wave iteration
    const int lane = threadIdx.x & (kWave - 1);
    const int grp = lane / LPR, sub = lane % LPR;
    const long long wave0 = (long long)blockIdx.x * (blockDim.x / kWave) + __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);  // wave-uniform: row arithmetic on the scalar unit
    const long long nwaves = (long long)gridDim.x * (blockDim.x / kWave);
    const long long niter = (total_rows + RPW - 1) / RPW;
    const float inv_T = 1.0f / (float)maxT, inv_U1 = 1.0f / (float)U1;

    for (long long it = wave0; it < niter; it += nwaves) {  // host launches one wave per iteration
        const long long r0 = it * RPW;
        const RowCoord c0 = decode_row(r0, maxT, U1);
        // Loads return in order, so each phase below issues its loads for all NR nodes before anything
        // waits for them: lengths -> labels and rows -> the two gathered logits (three round trips per
        // iteration; interleaved per node it was one round trip per node).
        SmallRow row[NR];
        small_rows<NR, G>(row, c0, grp, r0, total_rows, act_lens, label_lens, maxT, U1, B, inv_T, inv_U1);
        int y[NR];
        if (U1 > 1) {  // (labels may be an empty tensor otherwise)
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                const bool emit = row[n].valid && row[n].u < row[n].Ub;
                y[n] = labels[emit ? (size_t)row[n].b * (U1 - 1) + row[n].u : 0];
            }
        }
#pragma unroll
        for (int n = 0; n < NR; ++n) y[n] = (U1 > 1 && row[n].valid && row[n].u < row[n].Ub) ? y[n] : blank;
        u32x4 raw[NR][NCH];
        int first[NR];   // row index of element 0 of this lane's chunk 0 (negative in the boundary chunk)
        int nchunks[NR];
        size_t rowoff[NR];  // (an array of row pointers would come back as flat pointers)
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            const long long r = row[n].valid ? r0 + (n * G + grp) : 0;  // skipped nodes read (and ignore) row 0
            rowoff[n] = (size_t)r * (size_t)V;
            const typename DT::elem *rowp = acts + rowoff[n];
            const uintptr_t addr = reinterpret_cast<uintptr_t>(rowp);
            const int shift = MIS ? (int)((addr & 15) / sizeof(typename DT::elem)) : 0;
            const u32x4 *row4 = MIS ? align_down16(rowp) : reinterpret_cast<const u32x4 *>(rowp);
            nchunks[n] = MIS ? (shift + V + EPC - 1) / EPC : V / EPC;
            first[n] = sub * EPC - shift;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {  // unconditional loads: surplus lanes re-read the last chunk
                const int idx = sub + k * LPR;
                raw[n][k] = __builtin_nontemporal_load(row4 + (idx < nchunks[n] ? idx : nchunks[n] - 1));
            }
        }
        // the two gathered logits (same address in every lane of a group; a label outside [0,V) is
        // the caller's error and yields NaN, as in rnnt_k1_lse)
        float xb[NR], xy[NR];
#pragma unroll
        for (int n = 0; n < NR; ++n) xb[n] = xy[n] = 0.0f;
        if (sub == 0) {  // one branch around all of them: the leading lane of each group stores the result
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                const bool y_ok = y[n] >= 0 && y[n] < V;
                xb[n] = DT::to_float(acts[rowoff[n] + blank]);
                xy[n] = DT::to_float(acts[rowoff[n] + (y_ok ? y[n] : 0)]);
                xy[n] = y_ok ? xy[n] : __builtin_nanf("");
            }
        }
        float m[NR], s[NR];
        bool pinf[NR];
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            m[n] = -INFINITY;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                float x[EPC];
                DT::unpack(raw[n][k], x);
                const int e0 = first[n] + k * LPR * EPC;
#pragma unroll
                for (int j = 0; j < EPC; ++j) {
                    // aligned rows: a chunk is inside the row or outside it as a whole
                    const bool in = MIS ? (unsigned)(e0 + j) < (unsigned)V : e0 < V;
                    m[n] = fmaxf(m[n], in ? x[j] : -INFINITY);
                }
            }
        }
#pragma unroll
        for (int off = LPR / 2; off > 0; off >>= 1)
#pragma unroll
            for (int n = 0; n < NR; ++n) m[n] = fmaxf(m[n], __shfl_xor(m[n], off, kWave));
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            pinf[n] = m[n] == INFINITY;
            m[n] = (m[n] == -INFINITY || m[n] == INFINITY) ? 0.0f : m[n];  // as rnnt_k1_lse
            s[n] = 0.0f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                float x[EPC];
                DT::unpack(raw[n][k], x);
                const int e0 = first[n] + k * LPR * EPC;
#pragma unroll
                for (int j = 0; j < EPC; ++j) {
                    const bool in = MIS ? (unsigned)(e0 + j) < (unsigned)V : e0 < V;
                    s[n] += in ? __builtin_amdgcn_exp2f((x[j] - m[n]) * kLog2e) : 0.0f;
                }
            }
        }
#pragma unroll
        for (int off = LPR / 2; off > 0; off >>= 1)
#pragma unroll
            for (int n = 0; n < NR; ++n) s[n] += __shfl_xor(s[n], off, kWave);
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            if (sub == 0 && row[n].valid) {
                const float logs = row_log_sum(s[n], pinf[n] ? INFINITY : 0.0f);
                const size_t o = skew_index(row[n].b, row[n].t, row[n].u, L.ND, L.US);
                L.ml[o] = make_float2(m[n], logs);
                L.lp[o] = make_float2((xb[n] - m[n]) - logs,
                                      row[n].u < row[n].Ub ? (xy[n] - m[n]) - logs : -INFINITY);
            }
        }
    }
}

// Rows too long for four waves' registers (more than 4*17*64 chunks = 69 632 bytes, e.g. V = 32 768
// fp32): one workgroup per node streams the row ONCE, 8 chunks per thread at a time, with a running
// {max, sum}: the sum is rescaled when a batch raises the max (one extra exp2 per 32..64 logits).
template <typename DT>
__global__ __launch_bounds__(256) void rnnt_k1_lse_wide(
    const typename DT::elem *__restrict__ acts, const int *__restrict__ labels,
    const int *__restrict__ act_lens, const int *__restrict__ label_lens, Lattice L, int maxT, int U1,
    int V, int blank, long long total_rows, int B, long long compact_rows) {
    constexpr int EPC = DT::EPC;
    constexpr int UNR = 8;
    __shared__ float red_m[4], red_s[4];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);  // wave-uniform
    const int nchunks = V / EPC;  // host guarantees V % EPC == 0 and 16-byte aligned rows
    auto safe = [](float m) { return (m == -INFINITY || m == INFINITY) ? 0.0f : m; };  // as rnnt_k1_lse

    const long long per_b = (long long)maxT * U1;
    const long long total = min(min(L.spans[B].first, total_rows), compact_rows ? compact_rows : total_rows);
    for (long long k = blockIdx.x; k < total; k += gridDim.x) {  // host launches one workgroup per node of the full tensor
        const Node c = find_node(L.spans, B, k, per_b, lane);
        const int Ub = c.Ub;
        const long long r = compact_rows ? k : ((long long)c.b * maxT + c.t) * U1 + c.u;
        const typename DT::elem *row = acts + (size_t)r * (size_t)V;
        const u32x4 *row4 = reinterpret_cast<const u32x4 *>(row);

        float m = -INFINITY, s = 0.0f;
        for (int base = threadIdx.x; base < nchunks; base += 256 * UNR) {
            u32x4 raw[UNR];
#pragma unroll
            for (int k = 0; k < UNR; ++k) {  // unconditional loads: a surplus slot re-reads the last chunk
                const int idx = base + k * 256;
                raw[k] = __builtin_nontemporal_load(row4 + (idx < nchunks ? idx : nchunks - 1));
            }
            if (kCvtUnpack<DT>) loads_issued(raw);
            float bm = -INFINITY;
#pragma unroll
            for (int k = 0; k < UNR; ++k) {
                float x[EPC];
                DT::unpack(raw[k], x);
                const bool in = base + k * 256 < nchunks;
#pragma unroll
                for (int j = 0; j < EPC; ++j) bm = fmaxf(bm, in ? x[j] : -INFINITY);
            }
            const float m_new = fmaxf(m, bm);
            const float mm = safe(m_new);
            s = (s == 0.0f) ? 0.0f : s * __builtin_amdgcn_exp2f((safe(m) - mm) * kLog2e);
#pragma unroll
            for (int k = 0; k < UNR; ++k) {
                float x[EPC];
                DT::unpack(raw[k], x);
                const bool in = base + k * 256 < nchunks;
#pragma unroll
                for (int j = 0; j < EPC; ++j) s += in ? __builtin_amdgcn_exp2f((x[j] - mm) * kLog2e) : 0.0f;
            }
            m = m_new;
        }
        // combine the 256 running pairs: lanes, then waves (fixed order: reproducible)
        float M = wave_max(m);
        s = (s == 0.0f) ? 0.0f : s * __builtin_amdgcn_exp2f((safe(m) - safe(M)) * kLog2e);
        s = wave_sum(s);
        if (lane == 0) {
            red_m[wave] = M;
            red_s[wave] = s;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float MM = red_m[0];
#pragma unroll
            for (int i = 1; i < 4; ++i) MM = fmaxf(MM, red_m[i]);
            const float mm = safe(MM);
            float S = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                S += (red_s[i] == 0.0f) ? 0.0f : red_s[i] * __builtin_amdgcn_exp2f((safe(red_m[i]) - mm) * kLog2e);
            const int y = (c.u < Ub) ? labels[(size_t)c.b * (U1 - 1) + c.u] : blank;
            const float xy = (y >= 0 && y < V) ? DT::to_float(row[y]) : __builtin_nanf("");
            const float logs = row_log_sum(S, MM);
            const size_t o = skew_index(c.b, c.t, c.u, L.ND, L.US);
            L.ml[o] = make_float2(mm, logs);
            L.lp[o] = make_float2((DT::to_float(row[blank]) - mm) - logs, (c.u < Ub) ? (xy - mm) - logs : -INFINITY);
        }
        __syncthreads();  // red_* are reused by the next node of a grid-stride launch
    }
}

// Fallback for any V / alignment (element-granular loads, two passes over an L2-resident row).
template <typename DT>
__global__ __launch_bounds__(256) void rnnt_k1_lse_generic(
    const typename DT::elem *__restrict__ acts, const int *__restrict__ labels,
    const int *__restrict__ act_lens, const int *__restrict__ label_lens, Lattice L, int maxT, int U1,
    int V, int blank, long long total_rows, int B, long long compact_rows) {
    const int lane = threadIdx.x & (kWave - 1);
    const long long per_b = (long long)maxT * U1;
    const long long wave0 = (long long)blockIdx.x * (blockDim.x / kWave) + __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);  // wave-uniform: row arithmetic on the scalar unit
    const long long nwaves = (long long)gridDim.x * (blockDim.x / kWave);
    const long long total = min(min(L.spans[B].first, total_rows), compact_rows ? compact_rows : total_rows);
    for (long long k = wave0; k < total; k += nwaves) {
        const Node c = find_node(L.spans, B, k, per_b, lane);
        const int Ub = c.Ub;
        const long long r = compact_rows ? k : ((long long)c.b * maxT + c.t) * U1 + c.u;
        const typename DT::elem *row = acts + (size_t)r * (size_t)V;
        float m = -INFINITY;
        for (int v = lane; v < V; v += kWave) m = fmaxf(m, DT::to_float(row[v]));
        m = wave_max(m);
        const float mm = (m == -INFINITY || m == INFINITY) ? 0.0f : m;
        float s = 0.0f;
        for (int v = lane; v < V; v += kWave)
            s += __builtin_amdgcn_exp2f((DT::to_float(row[v]) - mm) * kLog2e);
        s = wave_sum(s);
        if (lane == 0) {
            const int y = (c.u < Ub) ? labels[(size_t)c.b * (U1 - 1) + c.u] : blank;
            const float logs = row_log_sum(s, m);
            const size_t o = skew_index(c.b, c.t, c.u, L.ND, L.US);
            const float xy = (y >= 0 && y < V) ? DT::to_float(row[y]) : __builtin_nanf("");
            L.ml[o] = make_float2(mm, logs);
            L.lp[o] = make_float2((DT::to_float(row[blank]) - mm) - logs, (c.u < Ub) ? (xy - mm) - logs : -INFINITY);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K2: alpha / beta over anti-diagonals.  
// Latency-bound: T+U dependent steps.  Block = 2*NW waves; waves [0,NW) alpha, [NW,2NW) beta.
// Lane = u; on diagonal d the lane owns node (d-u, u):
//   alpha(t,u) = logaddexp(alpha(t-1,u) + lp_blank(t-1,u), alpha(t,u-1) + lp_label(t,u-1))
//   beta(t,u)  = logaddexp(beta(t+1,u) + lp_blank(t,u),    beta(t,u+1) + lp_label(t,u))
// (the sums and their order are those of the upstream CPU helper, so the fp32 rounding follows
// it).  The own-column operand is the lane's previous value, the neighbour's arrives by a
// one-lane DPP whole-wave shift (through a two-slot LDS cell across a wave edge).  The loop is
// branch-free and needs no validity selects: a lane holds -inf until its column enters the
// lattice, every load is clamped to a valid node of the lane's own column (so no uninitialised
// slot is ever added to a -inf), lp_label is -inf at u = U, and whatever a lane computes after
// leaving the lattice only reaches lanes that have left it too.  kPF diagonals of
// {lp_blank, lp_label} per register set, two sets ping-pong behind counted s_waitcnt vmcnt.
// ---------------------------------------------------------------------------------------------
// DPP whole-wave shifts by one lane (GFX9 dpp_ctrl wave_shr:1 = 0x138, wave_shl:1 = 0x130).
// Lanes with no source keep `fill`.
__device__ __forceinline__ float wave_shift_up(float v, float fill) {  // lane i <- lane i-1
    return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(fill), __float_as_uint(v),
                                                       0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_shift_down(float v, float fill) {  // lane i <- lane i+1
    return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(fill), __float_as_uint(v),
                                                       0x130, 0xf, 0xf, false));
}

// log(e^a + e^b) = big + log(1 + e^(small-big)), the arrangement of the upstream CPU helper.
// Branch-free: a -inf operand gives exactly the other one; -inf with -inf gives -inf (the fminf
// drops the NaN of inf-inf).  v_exp_f32 / v_log_f32 directly: the argument of the log is in [1,2].
__device__ __forceinline__ float log_add_exp(float a, float b) {
    const float big = fmaxf(a, b);
    const float d = fminf(fminf(a, b) - big, 0.0f);
    const float r = big + kLn2 * __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(d * kLog2e));
    // v_max/v_min return the other operand for a NaN; the loss must surface a NaN logit as a NaN
    // cost (bin/train.py:36-41 skips the update on it), as the upstream helper does.
    return __builtin_isunordered(a, b) ? __builtin_nanf("") : r;
}

constexpr int kPF = 16;  // diagonals per register set; two sets ping-pong, so 16..32 rows are in flight

template <int NW>
__global__ __launch_bounds__(2 * NW * kWave) void rnnt_k2_lattice(Lattice L, float *__restrict__ costs,
                                                                  const int *__restrict__ act_lens,
                                                                  const int *__restrict__ label_lens,
                                                                  int maxT, int U1) {
    constexpr int US = NW * kWave;
    __shared__ float xch[2][2][NW];  // [direction][step parity][wave]: values crossing a wave edge
    const int b = blockIdx.x;
    const int T = act_lens[b];
    const int U = label_lens[b];  // labels; lattice columns 0..U
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);  // wave-uniform
    const int dir = wave / NW;
    const int w = wave - dir * NW;
    const int u = w * kWave + lane;
    if (!lens_ok(T, U, maxT, U1)) {  // block-uniform exit, before any barrier
        if (threadIdx.x == 0) {
            L.ll[b] = __builtin_nanf("");
            costs[b] = __builtin_nanf("");
        }
        return;
    }
    const int nd = T + U;                                           // diagonals 0 .. nd-1
    const int nsteps = (nd + 2 * kPF - 1) / (2 * kPF) * (2 * kPF);  // whole groups; surplus steps
    const int junk = L.ND;                                          // store into the utterance's junk row
    const size_t base = (size_t)b * (L.ND + 1) * US;
    // loads: this lane's column (clamped into the lattice), rows clamped to that column's nodes
    const int uc = min(u, U);
    const float2 *__restrict__ lpc = L.lp + base + uc;
    const int row_lo = uc, row_hi = uc + T - 1;
    auto row_of = [&](int d) { return (size_t)min(max(d, row_lo), row_hi) * US; };
    const float NEG = -INFINITY;

    if (dir == 0) {
        float *__restrict__ out = L.alpha + base + u;
        float up = (u == 0) ? 0.0f : NEG;  // alpha(t-1,u) + lp_blank(t-1,u); a virtual 0 enters (0,0)
        float right = NEG;                 // alpha(t,u) + lp_label(t,u), for lane u+1
        float lik = NEG;
        auto step = [&](int d, const float2 cur) {
            float left = wave_shift_up(right, NEG);
            if (NW > 1) {
                const float edge = xch[0][(d + 1) & 1][w > 0 ? w - 1 : 0];  // written at step d-1
                if (lane == 0 && w > 0 && d > 0) left = edge;
            }
            const float a = log_add_exp(left, up);
            out[(size_t)(d < nd ? d : junk) * US] = a;
            up = a + cur.x;
            right = a + cur.y;
            if (d == nd - 1) lik = up;  // wave-uniform; lane U: alpha(T-1,U) + lp_blank(T-1,U)
            if (NW > 1) {
                if (lane == kWave - 1) xch[0][d & 1][w] = right;
                __syncthreads();
            }
        };
        float2 ra[kPF], rb[kPF];
#pragma unroll
        for (int i = 0; i < kPF; ++i) ra[i] = lpc[row_of(i)];
        for (int d0 = 0; d0 < nsteps; d0 += 2 * kPF) {
#pragma unroll
            for (int i = 0; i < kPF; ++i) {  // consume set A, refill set B for steps d0+kPF ..
                rb[i] = lpc[row_of(d0 + kPF + i)];
                step(d0 + i, ra[i]);
            }
#pragma unroll
            for (int i = 0; i < kPF; ++i) {  // consume set B, refill set A for steps d0+2*kPF ..
                ra[i] = lpc[row_of(d0 + 2 * kPF + i)];
                step(d0 + kPF + i, rb[i]);
            }
        }
        if (u == U) {
            L.ll[b] = lik;
            costs[b] = -lik;
        }
    } else {
        float *__restrict__ out = L.beta + base + u;
        float down = (u == U) ? 0.0f : NEG;  // beta(t+1,u); a virtual 0 leaves (T-1,U) through blank
        auto step = [&](int s, const float2 cur) {
            const int d = nd - 1 - s;
            float right = wave_shift_down(down, NEG);  // beta(t,u+1): lane u+1's value of the previous step
            if (NW > 1) {
                const float edge = xch[1][(s + 1) & 1][w < NW - 1 ? w + 1 : 0];  // written at step s-1
                if (lane == kWave - 1 && w < NW - 1) right = edge;
            }
            if (s == 0) right = NEG;  // the virtual 0 in lane U belongs to lane U alone
            const float v = log_add_exp(right + cur.y, down + cur.x);
            out[(size_t)(d >= 0 ? d : junk) * US] = v;
            down = v;
            if (NW > 1) {
                if (lane == 0) xch[1][s & 1][w] = v;
                __syncthreads();
            }
        };
        float2 ra[kPF], rb[kPF];
#pragma unroll
        for (int i = 0; i < kPF; ++i) ra[i] = lpc[row_of(nd - 1 - i)];
        for (int s0 = 0; s0 < nsteps; s0 += 2 * kPF) {
#pragma unroll
            for (int i = 0; i < kPF; ++i) {
                rb[i] = lpc[row_of(nd - 1 - (s0 + kPF + i))];
                step(s0 + i, ra[i]);
            }
#pragma unroll
            for (int i = 0; i < kPF; ++i) {
                ra[i] = lpc[row_of(nd - 1 - (s0 + 2 * kPF + i))];
                step(s0 + kPF + i, rb[i]);
            }
        }
    }
}

// Lattices wider than 512 columns (more than 511 labels; upstream has no such limit, the reference's data
// never comes near it): one workgroup of 1024 threads per utterance, threads [0,512) walk alpha up the
// anti-diagonals and [512,1024) walk beta down, each thread taking columns j, j+512, ...  The previous
// diagonal lives in LDS (two ping-pong rows of US floats per direction), one barrier per diagonal; same
// operands in the same order as rnnt_k2_lattice, so a lattice that fits both gives the same bits.
// Latency-bound like the fast kernel, without its register prefetch: ~1-2 us per diagonal.
constexpr int kMaxLatticeColumns = RNNT_MAX_LATTICE_COLUMNS;  // 16 bytes of LDS per column: 64 KiB
static_assert(kMaxLatticeColumns % 64 == 0 && kMaxLatticeColumns * 16 <= 64 * 1024, "LDS of rnnt_k2_lattice_wide");
__global__ __launch_bounds__(1024) void rnnt_k2_lattice_wide(Lattice L, float *__restrict__ costs,
                                                             const int *__restrict__ act_lens,
                                                             const int *__restrict__ label_lens, int maxT,
                                                             int U1) {
    extern __shared__ float diag[];  // [direction][step parity][US]
    const int US = L.US;
    const int b = blockIdx.x;
    const int T = act_lens[b];
    const int U = label_lens[b];
    if (!lens_ok(T, U, maxT, U1)) {  // block-uniform exit, before any barrier
        if (threadIdx.x == 0) {
            L.ll[b] = __builtin_nanf("");
            costs[b] = __builtin_nanf("");
        }
        return;
    }
    const int dir = threadIdx.x >> 9;
    const int j = threadIdx.x & 511;
    const int nd = T + U;  // diagonals 0 .. nd-1
    const size_t base = (size_t)b * (L.ND + 1) * US;
    float *mine = diag + (size_t)dir * 2 * US;
    const float NEG = -INFINITY;
    for (int s = 0; s < nd; ++s) {
        const int d = dir == 0 ? s : nd - 1 - s;
        float *cur = mine + (size_t)(s & 1) * US;
        const float *prv = mine + (size_t)((s + 1) & 1) * US;
        const int u_lo = max(0, d - (T - 1)), u_hi = min(U, d);  // nodes (d-u, u) of this diagonal
        for (int u = u_lo + j; u <= u_hi; u += 512) {
            const int t = d - u;
            const size_t o = base + (size_t)d * US + u;
            if (dir == 0) {
                // alpha(t,u) = logaddexp(alpha(t,u-1) + lp_label(t,u-1), alpha(t-1,u) + lp_blank(t-1,u)); both
                // source nodes sit on diagonal d-1, columns u-1 and u
                float left = NEG, up = (d == 0) ? 0.0f : NEG;  // a virtual 0 enters (0,0)
                if (u > 0) left = prv[u - 1] + L.lp[o - US - 1].y;
                if (t > 0) up = prv[u] + L.lp[o - US].x;
                const float a = log_add_exp(left, up);
                cur[u] = a;
                L.alpha[o] = a;
                if (d == nd - 1) {  // node (T-1, U)
                    const float lik = a + L.lp[o].x;
                    L.ll[b] = lik;
                    costs[b] = -lik;
                }
            } else {
                // beta(t,u) = logaddexp(beta(t,u+1) + lp_label(t,u), beta(t+1,u) + lp_blank(t,u)); both source
                // nodes sit on diagonal d+1, columns u+1 and u
                const float2 lp = L.lp[o];
                float right = NEG, down = (s == 0) ? 0.0f : NEG;  // a virtual 0 leaves (T-1,U) through blank
                if (u < U) right = prv[u + 1];
                if (t < T - 1) down = prv[u];
                const float v = log_add_exp(right + lp.y, down + lp.x);
                cur[u] = v;
                L.beta[o] = v;
            }
        }
        __syncthreads();
    }
}

// costs -> loss_out, fixed summation order (bitwise reproducible, identical on every rank)
__global__ void rnnt_reduce_costs_kernel(const float *__restrict__ costs, float *__restrict__ loss_out,
                                         int B, int reduction) {
    if (reduction == RNNT_REDUCTION_NONE) {
        for (int i = threadIdx.x; i < B; i += blockDim.x) loss_out[i] = costs[i];
        return;
    }
    if (threadIdx.x == 0) {
        float s = 0.0f;
        for (int i = 0; i < B; ++i) s += costs[i];
        loss_out[0] = (reduction == RNNT_REDUCTION_MEAN) ? s / (float)B : s;
    }
}

// ---------------------------------------------------------------------------------------------
// K3: fused gradient.  (SURVEY.md §8a A5 + A6)
// Bound: HBM read N_valid + write N_full.  One wave per node.
//   g_v = scale * ( softmax_v * (Bk + Yk)  - [v=blank] Bk - [v=y_u] Yk )
// ---------------------------------------------------------------------------------------------
// Same association as the upstream CPU path: alpha + beta_next + lp - ll, and the dense softmax
// term weighted by (Bk + Yk) rather than by e^{alpha+beta-ll}.
__device__ __forceinline__ NodeCoef node_coef(const Lattice &L, const RowCoord &c, int Tb, int Ub,
                                              const int *__restrict__ labels, int U1,
                                              const float *__restrict__ grad_out, int grad_out_stride,
                                              float grad_scale) {
    const size_t o = skew_index(c.b, c.t, c.u, L.ND, L.US);
    const float llb = L.ll[c.b];
    const float a = L.alpha[o];
    const float2 lp = L.lp[o];
    const float scale = grad_scale * (grad_out ? grad_out[(size_t)c.b * grad_out_stride] : 1.0f);
    float bk = 0.0f, yk = 0.0f;
    NodeCoef n;
    n.y = -1;
    if (c.t < Tb - 1)
        bk = __expf(a + L.beta[o + L.US] + lp.x - llb);  // beta(t+1,u): next diagonal, same column
    else if (c.u == Ub)
        bk = __expf(a + lp.x - llb);
    if (c.u < Ub) {
        n.y = labels[(size_t)c.b * (U1 - 1) + c.u];
        yk = __expf(a + L.beta[o + L.US + 1] + lp.y - llb);  // beta(t,u+1): next diagonal, column+1
    }
    const float2 ml = L.ml[o];
    n.m = ml.x;
    n.l = ml.y;
    n.sa = scale * (bk + yk);
    n.sbk = scale * bk;
    n.syk = scale * yk;
    return n;
}

// One 16-byte chunk of a node's row.  The two sparse corrections (blank and label logit) touch one
// lane of one wave-instruction each; `hit_b` / `hit_y` are wave-uniform ("this instruction holds
// it"), so the element-wise part stays at sub, fma, exp2, mul per logit — K3 at two waves per SIMD
// turned out to be sensitive to VALU work (+2 ops per logit cost 7 %).
template <typename DT>
__device__ __forceinline__ u32x4 grad_chunk(const u32x4 &raw, const NodeCoef &n, float nl2, bool hit_b,
                                            int lane_b, bool hit_y, int lane_y, int lane, int jb, int jy) {
    constexpr int EPC = DT::EPC;
    float x[EPC];
    DT::unpack(raw, x);
    // two logits per VALU instruction (v_pk_add_f32 / v_pk_fma_f32 / v_pk_mul_f32); only exp2 is scalar
    const f32x2 m2 = {n.m, n.m}, k2 = {kLog2e, kLog2e}, n2 = {nl2, nl2}, sa2 = {n.sa, n.sa};
#pragma unroll
    for (int j = 0; j < EPC; j += 2) {
        f32x2 v = {x[j], x[j + 1]};
        v = __builtin_elementwise_fma(v - m2, k2, n2);
        v.x = __builtin_amdgcn_exp2f(v.x);
        v.y = __builtin_amdgcn_exp2f(v.y);
        v = v * sa2;
        x[j] = v.x;
        x[j + 1] = v.y;
    }
    if (hit_b) {
        asm volatile("; blank-logit correction" ::: "memory");  // keeps this a (scalar) branch: the
        // compiler otherwise if-converts it into 2 VALU ops per logit on every chunk
        const float sub = (lane == lane_b) ? n.sbk : 0.0f;
#pragma unroll
        for (int j = 0; j < EPC; ++j) x[j] -= (j == jb) ? sub : 0.0f;
    }
    if (hit_y) {
        asm volatile("; label-logit correction" ::: "memory");
        const float sub = (lane == lane_y) ? n.syk : 0.0f;
#pragma unroll
        for (int j = 0; j < EPC; ++j) x[j] -= (j == jy) ? sub : 0.0f;
    }
    return DT::pack(x);
}

// The whole row is in flight per wave (NCH chunks per lane): few, large read bursts followed by
// large write bursts per CU measured faster than many small ones at high occupancy.
template <bool NT, typename P>
__device__ __forceinline__ u32x4 ld16(const P *p) {
    return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT>
__device__ __forceinline__ void st16(const u32x4 &v, u32x4 *p) {
    if (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

template <typename DT, int NCH, bool MIS = false, bool NTL = true, bool NTS = true>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(k3_waves(NCH, DT::EPC, MIS), k3_waves(NCH, DT::EPC, MIS))))
void rnnt_k3_grad(
    const typename DT::elem *__restrict__ acts, typename DT::elem *__restrict__ grads,
    const int *__restrict__ labels, const int *__restrict__ act_lens,
    const int *__restrict__ label_lens, Lattice L, const float *__restrict__ grad_out,
    int grad_out_stride, float grad_scale, int maxT, int U1, int V, int blank, long long total_rows,
    int nseg, int B, long long compact_rows) {
    constexpr int EPC = DT::EPC;
    const int lane = threadIdx.x & (kWave - 1);
    const long long wave0 = (long long)blockIdx.x * (blockDim.x / kWave) + __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);  // wave-uniform: row arithmetic on the scalar unit
    const long long nwaves = (long long)gridDim.x * (blockDim.x / kWave);
    // compact_rows > 0 (rnnt_backward_compact): acts and grads hold one row per NODE, in the compacted order K1 walks
    // (rnnt_node_spans); rows are then nodes, found as K1 finds them, and nothing outside the lattices exists
    const long long nrows = compact_rows ? min(min(L.spans[B].first, total_rows), compact_rows) : total_rows;
    const long long nitems = nrows * nseg;  // a work item = NCH*64 consecutive chunks of one node
    const long long per_b = (long long)maxT * U1;

    for (long long wi = wave0; wi < nitems; wi += nwaves) {  // host launches one wave per item
        const long long r = wi / nseg;
        RowCoord c;
        int Tb, Ub;
        if (compact_rows) {  // (kernel-uniform)
            const Node nd = find_node(L.spans, B, r, per_b, lane);
            c.b = nd.b, c.t = nd.t, c.u = nd.u;
            Tb = nd.Tb, Ub = nd.Ub;
        } else {
            c = decode_row(r, maxT, U1);
            Tb = act_lens[c.b];
            Ub = label_lens[c.b];
        }
        // MIS: rows are not 16-byte aligned (host guarantees acts and grads share the misalignment);
        // `shift` elements of the previous row sit in chunk 0, see mask_outside_row
        typename DT::elem *orow = grads + (size_t)r * (size_t)V;
        const int shift = MIS ? (int)((reinterpret_cast<uintptr_t>(orow) & 15) / sizeof(typename DT::elem)) : 0;
        u32x4 *out4 = MIS ? align_down16(orow) : reinterpret_cast<u32x4 *>(orow);
        const int nchunks = MIS ? (shift + V + EPC - 1) / EPC : V / EPC;
        // lane 0 shifted back so each wave-instruction covers a 128-byte-aligned KiB (see K1)
        const int c0 = (int)(wi - r * nseg) * NCH * kWave + lane - row_misalign(out4);
        if (!lens_ok(Tb, Ub, maxT, U1) || c.t >= Tb || c.u > Ub) {
            const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int idx = c0 + k * kWave;
                if (idx >= 0 && idx < nchunks) {
                    if (MIS && (idx == 0 || idx == nchunks - 1))
                        store_inside_row<DT>(z, out4 + idx, idx * EPC - shift, V);
                    else
                        st16<NTS>(z, out4 + idx);
                }
            }
            continue;
        }
        // The node's lattice values and label before the row: loads return in order, so behind the
        // row they cost further round trips after it has arrived.  16-bit rows and short fp32 rows
        // gain 6-9 % (bf16 V = 4232: 1.69 -> 1.59 ms).  Long fp32 rows at 2-4 waves per SIMD lose
        // 2-3 % (V = 4232: 2.89 -> 2.95 ms) and keep the late loads.
        constexpr bool kEarly = EPC > 4 || NCH <= 6;
        const bool emit = c.u < Ub;
        NodeIn in;
        int y = -1;
        if (kEarly) {
            in = node_in(L, c, emit, grad_out, grad_out_stride);
            if (U1 > 1) y = labels[emit ? (size_t)c.b * (U1 - 1) + c.u : 0];
        }
        const typename DT::elem *irow = acts + (size_t)r * (size_t)V;
        const u32x4 *row4 = MIS ? align_down16(irow) : reinterpret_cast<const u32x4 *>(irow);
        u32x4 raw[NCH];
        if (NCH <= 6) {
            // short rows: bounded buffer loads as in K1 (fp16 with 4 chunks per lane otherwise waits
            // for each load before issuing the next: V = 1000, K3 2.46 -> 1.85 ms).  From 8 chunks
            // per lane up the lane-masked loads below measure the same or better (V = 8192: +4.5 %).
            const __amdgpu_buffer_rsrc_t rsrc = row_buffer(row4, nchunks * 16);
#pragma unroll
            for (int k = 0; k < NCH; ++k)
                raw[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (c0 + k * kWave) * 16, 0, NTL ? kBufferNontemporal : 0);
        } else {
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int idx = c0 + k * kWave;
                if (idx >= 0 && idx < nchunks) raw[k] = ld16<NTL>(row4 + idx);
            }
        }
        const NodeCoef n = kEarly ? node_coef_from(in, c.t, c.u, Tb, Ub, y, grad_scale)
                                  : node_coef(L, c, Tb, Ub, labels, U1, grad_out, grad_out_stride, grad_scale);
        const float nl2 = -n.l * kLog2e;  // softmax_v = 2^((x_v - m)*log2e - l*log2e)
        // which wave-instruction k / lane / element holds the blank and the label logit (wave-uniform)
        const int base = __builtin_amdgcn_readfirstlane(c0 - lane);
        const int pos_b = blank + shift;
        const int pos_y = n.y >= 0 ? n.y + shift : -(1 << 24);
        const int rel_b = pos_b / EPC - base;
        const int rel_y = (pos_y >= 0 ? pos_y / EPC : -(1 << 20)) - base;
        const int kb = __builtin_amdgcn_readfirstlane(rel_b >= 0 ? rel_b >> 6 : -1);
        const int ky = __builtin_amdgcn_readfirstlane(rel_y >= 0 ? rel_y >> 6 : -1);
        const int lane_b = rel_b & (kWave - 1), lane_y = rel_y & (kWave - 1);
        const int jb = pos_b & (EPC - 1), jy = pos_y & (EPC - 1);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int idx = c0 + k * kWave;
            if (idx >= 0 && idx < nchunks) {
                const u32x4 g = grad_chunk<DT>(raw[k], n, nl2, k == kb, lane_b, k == ky, lane_y, lane, jb, jy);
                if (MIS && (idx == 0 || idx == nchunks - 1))
                    store_inside_row<DT>(g, out4 + idx, idx * EPC - shift, V);
                else
                    st16<NTS>(g, out4 + idx);
            }
        }
    }
}

// K3 for small rows: same node-per-lane-group batching as rnnt_k1_lse_small.
template <typename DT, int LPR, int NCH, int NR, bool MIS>
__global__ __launch_bounds__(256) void rnnt_k3_grad_small(
    const typename DT::elem *__restrict__ acts, typename DT::elem *__restrict__ grads,
    const int *__restrict__ labels, const int *__restrict__ act_lens,
    const int *__restrict__ label_lens, Lattice L, const float *__restrict__ grad_out,
    int grad_out_stride, float grad_scale, int maxT, int U1, int V, int blank, long long total_rows, int B) {
    constexpr int EPC = DT::EPC;
    constexpr int G = kWave / LPR;
    constexpr int RPW = G * NR;
    typedef typename DT::elem E;
    const int lane = threadIdx.x & (kWave - 1);
    const int grp = lane / LPR, sub = lane % LPR;
    const long long wave0 = (long long)blockIdx.x * (blockDim.x / kWave) + __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);  // wave-uniform: row arithmetic on the scalar unit
    const long long nwaves = (long long)gridDim.x * (blockDim.x / kWave);
    const long long niter = (total_rows + RPW - 1) / RPW;
    const float inv_T = 1.0f / (float)maxT, inv_U1 = 1.0f / (float)U1;

    for (long long it = wave0; it < niter; it += nwaves) {
        const long long r0 = it * RPW;
        const RowCoord c0 = decode_row(r0, maxT, U1);
        // Phases as in rnnt_k1_lse_small: lengths -> lattice values and rows -> arithmetic, each
        // phase's loads issued for all NR nodes before anything waits for them.
        SmallRow row[NR];
        small_rows<NR, G>(row, c0, grp, r0, total_rows, act_lens, label_lens, maxT, U1, B, inv_T, inv_U1);
        NodeIn in[NR];
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            RowCoord c;  // padding nodes read (and ignore) node (0,0,0)
            c.b = row[n].valid ? row[n].b : 0;
            c.t = row[n].valid ? row[n].t : 0;
            c.u = row[n].valid ? row[n].u : 0;
            in[n] = node_in(L, c, row[n].valid && row[n].u < row[n].Ub, grad_out, grad_out_stride);
        }
        int y[NR];
        if (U1 > 1) {  // (labels may be an empty tensor otherwise)
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                const bool emit = row[n].valid && row[n].u < row[n].Ub;
                y[n] = labels[emit ? (size_t)row[n].b * (U1 - 1) + row[n].u : 0];
            }
        }
        u32x4 raw[NR][NCH];
        int first[NR], nchunks[NR];
        u32x4 *out4[NR];
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            const int o = n * G + grp;
            const long long r = row[n].exists ? r0 + o : 0;
            // host guarantees acts and grads share their misalignment, so one shift serves both
            E *orow = grads + (size_t)r * (size_t)V;
            const E *irow = acts + (size_t)(row[n].valid ? r : 0) * (size_t)V;
            const uintptr_t iaddr = reinterpret_cast<uintptr_t>(irow);
            const int shift = MIS ? (int)((reinterpret_cast<uintptr_t>(orow) & 15) / sizeof(E)) : 0;
            out4[n] = MIS ? align_down16(orow) : reinterpret_cast<u32x4 *>(orow);
            const u32x4 *row4 = MIS ? align_down16(irow) : reinterpret_cast<const u32x4 *>(irow);
            nchunks[n] = MIS ? (shift + V + EPC - 1) / EPC : V / EPC;
            first[n] = sub * EPC - shift;
            // = nchunks unless row 0 stands in for a padding node
            const int in_chunks = MIS ? ((int)((iaddr & 15) / sizeof(E)) + V + EPC - 1) / EPC : V / EPC;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int idx = sub + k * LPR;
                raw[n][k] = __builtin_nontemporal_load(row4 + (idx < in_chunks ? idx : in_chunks - 1));
            }
        }
        NodeCoef coef[NR];
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            const bool emit = U1 > 1 && row[n].valid && row[n].u < row[n].Ub;
            coef[n] = node_coef_from(in[n], row[n].t, row[n].u, row[n].Tb, row[n].Ub, emit ? y[n] : -1, grad_scale);
            if (!row[n].valid) {
                coef[n].sa = coef[n].sbk = coef[n].syk = 0.0f;
                coef[n].m = coef[n].l = 0.0f;
                coef[n].y = -1;
            }
        }
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            if (!row[n].exists) continue;
            const float nl2 = -coef[n].l * kLog2e;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int idx = sub + k * LPR;
                if (idx >= nchunks[n]) continue;
                const int e0 = first[n] + k * LPR * EPC;
                float x[EPC];
                DT::unpack(raw[n][k], x);
#pragma unroll
                for (int j = 0; j < EPC; ++j) {
                    float g = coef[n].sa * __builtin_amdgcn_exp2f(fmaf(x[j] - coef[n].m, kLog2e, nl2));
                    g -= (e0 + j == blank) ? coef[n].sbk : 0.0f;
                    g -= (e0 + j == coef[n].y) ? coef[n].syk : 0.0f;
                    x[j] = row[n].valid ? g : 0.0f;  // padding node: exact zeros (sa = 0 could meet an inf)
                }
                const u32x4 g4 = DT::pack(x);
                if (!MIS || (e0 >= 0 && e0 + EPC <= V))
                    __builtin_nontemporal_store(g4, out4[n] + idx);
                else
                    store_inside_row<DT>(g4, out4[n] + idx, e0, V);
            }
        }
    }
}

template <typename DT>
__global__ __launch_bounds__(1024) void rnnt_k3_grad_generic(
    const typename DT::elem *__restrict__ acts, typename DT::elem *__restrict__ grads,
    const int *__restrict__ labels, const int *__restrict__ act_lens,
    const int *__restrict__ label_lens, Lattice L, const float *__restrict__ grad_out,
    int grad_out_stride, float grad_scale, int maxT, int U1, int V, int blank, long long total_rows, int B,
    long long compact_rows) {
    const int lane = threadIdx.x & (kWave - 1);
    const long long wave0 = (long long)blockIdx.x * (blockDim.x / kWave) + __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);  // wave-uniform: row arithmetic on the scalar unit
    const long long nwaves = (long long)gridDim.x * (blockDim.x / kWave);
    const long long nrows = compact_rows ? min(min(L.spans[B].first, total_rows), compact_rows) : total_rows;  // as rnnt_k3_grad
    const long long per_b = (long long)maxT * U1;
    for (long long r = wave0; r < nrows; r += nwaves) {
        RowCoord c;
        int Tb, Ub;
        if (compact_rows) {
            const Node nd = find_node(L.spans, B, r, per_b, lane);
            c.b = nd.b, c.t = nd.t, c.u = nd.u;
            Tb = nd.Tb, Ub = nd.Ub;
        } else {
            c = decode_row(r, maxT, U1);
            Tb = act_lens[c.b];
            Ub = label_lens[c.b];
        }
        typename DT::elem *out = grads + (size_t)r * (size_t)V;
        if (!lens_ok(Tb, Ub, maxT, U1) || c.t >= Tb || c.u > Ub) {
            for (int v = lane; v < V; v += kWave) out[v] = DT::from_float(0.0f);
            continue;
        }
        const typename DT::elem *row = acts + (size_t)r * (size_t)V;
        const NodeCoef n = node_coef(L, c, Tb, Ub, labels, U1, grad_out, grad_out_stride, grad_scale);
        for (int v = lane; v < V; v += kWave) {
            float g = n.sa * __builtin_amdgcn_exp2f(fmaf(DT::to_float(row[v]) - n.m, kLog2e, -n.l * kLog2e));
            g -= (v == blank) ? n.sbk : 0.0f;
            g -= (v == n.y) ? n.syk : 0.0f;
            out[v] = DT::from_float(g);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
inline size_t align_up(size_t n, size_t a) { return (n + a - 1) / a * a; }
inline int row_stride(int maxU) { return (int)align_up((size_t)maxU, kWave); }

// floats in one fp32 plane of the skewed lattice (+1 row of slack: K3 reads beta at o+US+1)
inline size_t plane_floats(int maxT, int maxU, int B) {
    const size_t ND = (size_t)maxT + maxU - 1;
    const size_t US = row_stride(maxU);
    return (size_t)B * (ND + 1) * US + 2 * US;
}

// workspace = {m,l} (float2) | lp (float2) | alpha | beta | ll[B] | spans[B+1]; every part starts 256-byte aligned
inline size_t ll_floats(int B) { return align_up((size_t)B, 64); }
inline size_t spans_bytes(int B) { return align_up(((size_t)B + 1) * sizeof(NodeSpan), 256); }
inline Lattice carve(void *ws, int maxT, int maxU, int B) {
    const size_t p = plane_floats(maxT, maxU, B);  // multiple of 64 floats = 256 bytes
    float *f = static_cast<float *>(ws);
    Lattice L;
    L.ml = reinterpret_cast<float2 *>(f);
    L.lp = reinterpret_cast<float2 *>(f + 2 * p);
    L.alpha = f + 4 * p;
    L.beta = f + 5 * p;
    L.ll = f + 6 * p;
    L.spans = reinterpret_cast<NodeSpan *>(f + 6 * p + ll_floats(B));
    L.ND = maxT + maxU - 1;
    L.US = row_stride(maxU);
    return L;
}

inline int elem_size(int dtype) { return dtype == RNNT_DTYPE_F32 ? 4 : 2; }

bool check_common(const void *acts, const int *labels, const int *act_lens, const int *label_lens,
                  int V, int B, const void *ws, const rnntOptions &o) {
    if (!acts || !act_lens || !label_lens || !ws) return false;
    if (!labels && o.maxU > 1) return false;
    if (V <= 0 || B <= 0 || o.maxT <= 0 || o.maxU <= 0 || o.maxU > kMaxLatticeColumns) return false;
    if (o.blank_label < 0 || o.blank_label >= V) return false;
    if (o.dtype < RNNT_DTYPE_F32 || o.dtype > RNNT_DTYPE_F16) return false;
    if (reinterpret_cast<uintptr_t>(ws) % 256 != 0) return false;
    if (reinterpret_cast<uintptr_t>(acts) % elem_size(o.dtype) != 0) return false;
    return true;
}

// Fast path: 16-byte aligned rows.  A node's row is cut into `nseg` segments of NCH*64 chunks, one
// wave each (K1: nseg <= 4 waves of one workgroup; K3: any nseg, segments are independent).
struct Tile {
    int nch;   // chunks per lane, one of kNchSizes; 0 = use the element-granular fallback
    int nseg;  // segments (waves) per node
    bool mis;  // rows are not 16-byte aligned: the MIS kernels (boundary chunks masked / stored by element)
};
// K1 also has 10 and 14: the read-only pass gains from a closer fit (bf16 V=4232: 0.81 -> 0.71 ms with 10
// instead of 12 chunks per lane, fp32 V=9000: 1.47 -> 1.30 ms); K3 measured the same or 1-4 % worse with
// them (fp32 V=2400: 3.48 -> 3.61 ms) and keeps the coarser list.
constexpr int kNchSizes[] = {1, 2, 4, 6, 8, 10, 12, 14, 16, 17};
constexpr bool nch_for_k3(int sz) { return sz != 10 && sz != 14; }
inline Tile pick_tile(const void *p0, const void *p1, int V, int dtype, int max_seg, bool k3 = false) {
    Tile t = {0, 0, false};
    const int epc = 16 / elem_size(dtype);
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(p0) % 16, a1 = p1 ? reinterpret_cast<uintptr_t>(p1) % 16 : a0;
    if (a0 != a1) return t;  // logits and gradient rows must share their misalignment
    t.mis = V % epc != 0 || a0 != 0;
    // chunks per lane if one wave took the whole row (+7: the 128-byte alignment shift, which is zero
    // when every row starts on a 128-byte boundary; MIS: a row can straddle one more chunk)
    const bool lines = !t.mis && ((size_t)V * elem_size(dtype)) % 128 == 0 && reinterpret_cast<uintptr_t>(p0) % 128 == 0 &&
                       (!p1 || reinterpret_cast<uintptr_t>(p1) % 128 == 0);
    const int need = ((V + epc - 1) / epc + (t.mis ? 1 : 0) + (lines ? 0 : 7) + kWave - 1) / kWave;
    int nseg = (need + 16) / 17;
    if (max_seg == 4 && nseg == 3) nseg = 4;  // K1 splits a workgroup of 4 waves evenly
    if (nseg > max_seg) {
        t.mis = false;
        return t;
    }
    const int per = (need + nseg - 1) / nseg;
    for (int sz : kNchSizes)
        if (per <= sz && (!k3 || nch_for_k3(sz))) {
            t.nch = sz;
            t.nseg = nseg;
            return t;
        }
    return t;
}

// One wave per lattice node, nodes dealt to workgroups in memory order: the resident workgroups
// then cover one contiguous, advancing window of the logits (and of the gradient), which is what
// HBM wants.  Measured on config 2: K1 1.45 -> 1.31 ms, K3 3.51 -> 3.14 ms against a 2048-block
// grid-stride launch.  The kernels keep their grid-stride loop so any grid is valid.
inline int stream_grid(long long total_rows, int block) {
    const int wpb = block / kWave;
    const long long want = (total_rows + wpb - 1) / wpb;
    const long long cap = 0x7fffffffLL;
    return (int)(want < 1 ? 1 : (want > cap ? cap : want));
}

inline void record(const rnntOptions &o, int i) {
    if (o.stage_events && o.stage_events[i])
        (void)hipEventRecord(static_cast<hipEvent_t>(o.stage_events[i]), static_cast<hipStream_t>(o.stream));
}

// Developer knobs for A/B runs (tools/sweep.sh builds with -DRNNT_TUNING); compiled out of the default
// library, where the constants below are the measured best on config 2 (profiles/r01_tuning.md).
#ifdef RNNT_TUNING  // developer build only (tools/sweep.sh): the shipped library never reads the environment
inline int tuning_knob(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}
#else
constexpr int tuning_knob(const char *, int dflt) { return dflt; }
#endif

#define FOR_EACH_NCH(M) M(1) M(2) M(4) M(6) M(8) M(10) M(12) M(14) M(16) M(17)
#define FOR_EACH_NCH_K3(M) M(1) M(2) M(4) M(6) M(8) M(12) M(16) M(17)

// Small rows: shape index into {LPR, NCH, NR} = {4,1,8} {4,2,4} {8,2,4} {8,4,2} {8,8,1} {8,16,1}, or -1
// when a row can span more than 128 chunks (the one-wave-per-node kernels take those).  Few lanes per
// node whatever its length: one wave instruction then serves 8 or 16 nodes (the per-node bookkeeping
// is ~180 VALU instructions per instruction stream, which bounds these kernels, not HBM).  With
// 8 lanes a wave instruction still moves whole 128-byte lines per node; 4 lanes (64 bytes per node
// and instruction) pay for rows up to 8 chunks (V=16: 1.29 -> 0.86 ms, V=29: 1.61 -> 1.36 ms at
// 2.2e9 logits) but not for 16 (V=64: 1.83 -> 1.99 ms), and 2 lanes never did.
constexpr int kSmallShapes = 6;
constexpr int kSmallCap[kSmallShapes] = {4, 8, 16, 32, 64, 128};
constexpr int kSmallRowsPerWave[kSmallShapes] = {128, 64, 32, 16, 8, 8};
inline int small_shape(int V, int dtype, bool mis, int max_shape, int maxT) {
    if (maxT >= (1 << 21)) return -1;  // small_div is exact below 2^22 (frame index + carry)
    const int epc = 16 / elem_size(dtype);
    const int cover = mis ? (V + 2 * (epc - 1)) / epc : V / epc;  // chunks a row can touch (mis: at the worst offset)
    if (tuning_knob("RNNT_SMALL", 1) == 0) return -1;
    for (int i = 0; i <= max_shape; ++i)
        if (cover <= kSmallCap[i]) return i;
    return -1;
}
#define FOR_EACH_SMALL(M) M(0, 4, 1, 8) M(1, 4, 2, 4) M(2, 8, 2, 4) M(3, 8, 4, 2) M(4, 8, 8, 1) M(5, 8, 16, 1)
#define FOR_EACH_SMALL_K3(M) M(0, 4, 1, 8) M(1, 4, 2, 4) M(2, 8, 2, 4) M(3, 8, 4, 2) M(4, 8, 8, 1)

template <typename DT>
rnntStatus_t launch_k1(const void *acts, const int *labels, const int *act_lens, const int *label_lens,
                       int V, int B, const Lattice &L, const rnntOptions &o, long long compact_rows = 0) {
    typedef typename DT::elem E;
    const long long rows = (long long)B * o.maxT * o.maxU;
    const int block = 256;
    hipStream_t st = static_cast<hipStream_t>(o.stream);
    const bool mis = V % DT::EPC != 0 || reinterpret_cast<uintptr_t>(acts) % 16 != 0;
    const int shape = small_shape(V, o.dtype, mis, 5, o.maxT);
    // (the small-row kernels walk the tensor in memory order: a compacted layout takes the one-wave-per-node kernels)
    if (shape >= 0 && !compact_rows) {
        const long long iters = (rows + kSmallRowsPerWave[shape] - 1) / kSmallRowsPerWave[shape];
        const int sgrid = stream_grid(iters, block);
        switch (shape) {
#define CASE(I, LPR, NCH, NR)                                                                          \
    case I:                                                                                            \
        if (mis)                                                                                       \
            hipLaunchKernelGGL((rnnt_k1_lse_small<DT, LPR, NCH, NR, true>), dim3(sgrid), dim3(block), 0, st,  \
                               static_cast<const E *>(acts), labels, act_lens, label_lens, L, o.maxT,  \
                               o.maxU, V, o.blank_label, rows, B);                                     \
        else                                                                                           \
            hipLaunchKernelGGL((rnnt_k1_lse_small<DT, LPR, NCH, NR, false>), dim3(sgrid), dim3(block), 0, st, \
                               static_cast<const E *>(acts), labels, act_lens, label_lens, L, o.maxT,  \
                               o.maxU, V, o.blank_label, rows, B);                                     \
        break;
            FOR_EACH_SMALL(CASE)
#undef CASE
        }
        return hipGetLastError() == hipSuccess ? RNNT_STATUS_SUCCESS : RNNT_STATUS_EXECUTION_FAILED;
    }
    // every other K1 walks the compacted node order (rnnt_node_spans above)
    hipLaunchKernelGGL(rnnt_node_spans, dim3(1), dim3(kWave), 0, st, act_lens, label_lens, B, o.maxT, o.maxU, L.spans);
    const Tile t = pick_tile(acts, nullptr, V, o.dtype, 4);
    const long long grid_rows = compact_rows && compact_rows < rows ? compact_rows : rows;  // (one wave per row of `acts`)
    int grid = stream_grid(grid_rows * (t.nseg ? t.nseg : 1), block);
    const int g = tuning_knob("RNNT_K1_GRID", 0);
    if (g > 0) grid = g;
#define K1_ARGS static_cast<const E *>(acts), labels, act_lens, label_lens, L, o.maxT, o.maxU, V, o.blank_label, rows, B, compact_rows
#define K1_LAUNCH(N, W, M) hipLaunchKernelGGL((rnnt_k1_lse<DT, N, W, M>), dim3(grid), dim3(block), 0, st, K1_ARGS)
// several waves per node only occur with 9+ chunks per lane (pick_tile): the others are not instantiated
#define CASE(N)                                                                                        \
    case N:                                                                                            \
        if (N < 9 || t.nseg == 1) {                                                                    \
            if (t.mis) K1_LAUNCH(N, 1, true); else K1_LAUNCH(N, 1, false);                             \
        } else if (t.nseg == 2) {                                                                      \
            if (t.mis) K1_LAUNCH(N, (N < 9 ? 1 : 2), true); else K1_LAUNCH(N, (N < 9 ? 1 : 2), false); \
        } else {                                                                                       \
            if (t.mis) K1_LAUNCH(N, (N < 9 ? 1 : 4), true); else K1_LAUNCH(N, (N < 9 ? 1 : 4), false); \
        }                                                                                              \
        break;
    const bool vectorisable = V % DT::EPC == 0 && reinterpret_cast<uintptr_t>(acts) % 16 == 0;
    switch (t.nch) {
        FOR_EACH_NCH(CASE)
        default:
            if (vectorisable) {  // row longer than 4 waves hold: one workgroup per node, single streaming pass
                const int wgrid = g > 0 ? g : (int)(grid_rows > 0x7fffffffLL ? 0x7fffffffLL : grid_rows);
                hipLaunchKernelGGL((rnnt_k1_lse_wide<DT>), dim3(wgrid), dim3(block), 0, st, K1_ARGS);
            } else {
                hipLaunchKernelGGL((rnnt_k1_lse_generic<DT>), dim3(grid), dim3(block), 0, st, K1_ARGS);
            }
    }
#undef CASE
#undef K1_LAUNCH
#undef K1_ARGS
    return hipGetLastError() == hipSuccess ? RNNT_STATUS_SUCCESS : RNNT_STATUS_EXECUTION_FAILED;
}

#define K3_ARGS                                                                                       \
    static_cast<const E *>(acts), static_cast<E *>(grads), labels, act_lens, label_lens, L, grad_out, \
        grad_out_stride, grad_scale, o.maxT, o.maxU, V, o.blank_label, rows

template <typename DT>
rnntStatus_t launch_k3(const void *acts, void *grads, const int *labels, const int *act_lens,
                       const int *label_lens, int V, int B, const float *grad_out, int grad_out_stride,
                       float grad_scale, const Lattice &L, const rnntOptions &o, long long compact_rows = 0) {
    typedef typename DT::elem E;
    const long long rows = (long long)B * o.maxT * o.maxU;
    const int block = tuning_knob("RNNT_K3_BLOCK", 512);  // 8 nodes per workgroup: 3.14 -> 3.04 ms vs 256
    hipStream_t st = static_cast<hipStream_t>(o.stream);
    const bool mis = V % DT::EPC != 0 || reinterpret_cast<uintptr_t>(acts) % 16 != 0;
    // 16 chunks per lane (rows of 64..128 chunks) cost K3 its occupancy: V=512 fp32 1.58 ms against 1.44 ms
    // with one wave per node, so K3 stops one shape earlier than K1 (0.83 against 1.04 ms there)
    const int shape = small_shape(V, o.dtype, mis, 4, o.maxT);
    if (shape >= 0 && !compact_rows && reinterpret_cast<uintptr_t>(acts) % 16 == reinterpret_cast<uintptr_t>(grads) % 16) {
        const long long iters = (rows + kSmallRowsPerWave[shape] - 1) / kSmallRowsPerWave[shape];
        const int sgrid = stream_grid(iters, 256);
        switch (shape) {
#define CASE(I, LPR, NCH, NR)                                                                          \
    case I:                                                                                            \
        if (mis)                                                                                       \
            hipLaunchKernelGGL((rnnt_k3_grad_small<DT, LPR, NCH, NR, true>), dim3(sgrid), dim3(256), 0, st, K3_ARGS, B);  \
        else                                                                                           \
            hipLaunchKernelGGL((rnnt_k3_grad_small<DT, LPR, NCH, NR, false>), dim3(sgrid), dim3(256), 0, st, K3_ARGS, B); \
        break;
            FOR_EACH_SMALL_K3(CASE)
#undef CASE
        }
        return hipGetLastError() == hipSuccess ? RNNT_STATUS_SUCCESS : RNNT_STATUS_EXECUTION_FAILED;
    }
    const Tile t = pick_tile(acts, grads, V, o.dtype, 1 << 20, true);
    const long long grid_rows = compact_rows && compact_rows < rows ? compact_rows : rows;
    int grid = stream_grid(grid_rows * (t.nseg ? t.nseg : 1), block);
    const int g = tuning_knob("RNNT_K3_GRID", 0);
    if (g > 0) grid = g;
    const int lds = tuning_knob("RNNT_K3_LDS", 0);  // unused dynamic LDS: caps workgroups per CU
    switch (t.nch) {
#define CASE(N)                                                                                       \
    case N:                                                                                           \
        if (t.mis)                                                                                    \
            hipLaunchKernelGGL((rnnt_k3_grad<DT, N, true>), dim3(grid), dim3(block), lds, st, K3_ARGS, t.nseg, B, compact_rows);  \
        else                                                                                          \
            hipLaunchKernelGGL((rnnt_k3_grad<DT, N, false>), dim3(grid), dim3(block), lds, st, K3_ARGS, t.nseg, B, compact_rows); \
        break;
        FOR_EACH_NCH_K3(CASE)
#undef CASE
        default:
            hipLaunchKernelGGL((rnnt_k3_grad_generic<DT>), dim3(grid), dim3(block), 0, st, K3_ARGS, B, compact_rows);
    }
    return hipGetLastError() == hipSuccess ? RNNT_STATUS_SUCCESS : RNNT_STATUS_EXECUTION_FAILED;
}

rnntStatus_t launch_k2(const int *act_lens, const int *label_lens, int B, float *costs, const Lattice &L,
                       const rnntOptions &o) {
    const int NW = L.US / kWave;
    hipStream_t st = static_cast<hipStream_t>(o.stream);
#define K2(N)                                                                                         \
    hipLaunchKernelGGL((rnnt_k2_lattice<N>), dim3(B), dim3(2 * N * kWave), 0, st, L, costs, act_lens, \
                       label_lens, o.maxT, o.maxU)
    switch (NW) {
        case 1: K2(1); break;
        case 2: K2(2); break;
        case 3: K2(3); break;
        case 4: K2(4); break;
        case 5: K2(5); break;
        case 6: K2(6); break;
        case 7: K2(7); break;
        case 8: K2(8); break;
        default:  // more than 512 columns
            if (L.US > kMaxLatticeColumns) return RNNT_STATUS_INVALID_VALUE;
            hipLaunchKernelGGL(rnnt_k2_lattice_wide, dim3(B), dim3(1024), (size_t)4 * L.US * sizeof(float), st, L, costs,
                               act_lens, label_lens, o.maxT, o.maxU);
    }
#undef K2
    return hipGetLastError() == hipSuccess ? RNNT_STATUS_SUCCESS : RNNT_STATUS_EXECUTION_FAILED;
}

}  // namespace

namespace rnnt_mi355_detail {
// Everything of the forward pass after the lattice planes {max, log-sum} and {lp_blank, lp_label} are
// written: stage event 1, K2, the cost reduction, stage event 2.
rnntStatus_t launch_lattice_and_costs(const int *act_lens, const int *label_lens, int B, float *costs,
                                      float *loss_out, int reduction, const Lattice &L, const rnntOptions &o) {
    record(o, 1);
    const rnntStatus_t s = launch_k2(act_lens, label_lens, B, costs, L, o);
    if (s != RNNT_STATUS_SUCCESS) return s;
    if (loss_out) {
        hipLaunchKernelGGL(rnnt_reduce_costs_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(o.stream), costs,
                           loss_out, B, reduction);
        if (hipGetLastError() != hipSuccess) return RNNT_STATUS_EXECUTION_FAILED;
    }
    record(o, 2);
    return RNNT_STATUS_SUCCESS;
}
Lattice carve_workspace(void *ws, int maxT, int maxU, int B) { return carve(ws, maxT, maxU, B); }
rnntStatus_t launch_node_spans(const int *act_lens, const int *label_lens, int B, const Lattice &L, const rnntOptions &o) {
    hipLaunchKernelGGL(rnnt_node_spans, dim3(1), dim3(kWave), 0, static_cast<hipStream_t>(o.stream), act_lens, label_lens,
                       B, o.maxT, o.maxU, L.spans);
    return hipGetLastError() == hipSuccess ? RNNT_STATUS_SUCCESS : RNNT_STATUS_EXECUTION_FAILED;
}
rnntStatus_t launch_compact_rows(int *rowmap, const Lattice &L, int B, long long cap, const rnntOptions &o) {
    const long long rows = (long long)B * o.maxT * o.maxU;
    if (rows > 0x7fffffffLL) return RNNT_STATUS_INVALID_VALUE;  // rows are numbered with 32 bits
    hipLaunchKernelGGL(rnnt_compact_rows_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(o.stream), rowmap, L.spans, L.ml, L.lp, L.ND, L.US, o.maxT, o.maxU, rows, cap);
    return hipGetLastError() == hipSuccess ? RNNT_STATUS_SUCCESS : RNNT_STATUS_EXECUTION_FAILED;
}
}  // namespace rnnt_mi355_detail

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

const char *rnntGetStatusString(rnntStatus_t status) {
    switch (status) {
        case RNNT_STATUS_SUCCESS: return "no error";
        case RNNT_STATUS_MEMOPS_FAILED: return "hip memcpy or memset failed";
        case RNNT_STATUS_INVALID_VALUE: return "invalid value";
        case RNNT_STATUS_EXECUTION_FAILED: return "execution failed";
        case RNNT_STATUS_STATIC_WORKSPACE_EXCEEDED:
            return "lattice does not fit the static workspace (RNNT_STATIC_WORKSPACE_BYTES); use compute_rnnt_loss";
        case RNNT_STATUS_UNKNOWN_ERROR:
        default: return "unknown error";
    }
}

int rnnt_mi355_abi_version(void) { return kAbiVersion; }

rnntStatus_t rnnt_get_workspace_size(int maxT, int maxU, int minibatch, size_t *size_bytes) {
    if (!size_bytes || maxT <= 0 || maxU <= 0 || minibatch <= 0) return RNNT_STATUS_INVALID_VALUE;
    if (maxU > kMaxLatticeColumns) return RNNT_STATUS_INVALID_VALUE;
    *size_bytes = (6 * plane_floats(maxT, maxU, minibatch) + ll_floats(minibatch)) * sizeof(float) + spans_bytes(minibatch);
    return RNNT_STATUS_SUCCESS;
}

rnntStatus_t get_workspace_size(int maxT, int maxU, int minibatch, int gpu, size_t *size_bytes) {
    if (!gpu) return RNNT_STATUS_INVALID_VALUE;  // no CPU path in this library
    return rnnt_get_workspace_size(maxT, maxU, minibatch, size_bytes);
}

rnntStatus_t rnnt_forward(const void *acts, const int *labels, const int *act_lens, const int *label_lens,
                          int alphabet_size, int minibatch, float *costs, float *loss_out, int reduction,
                          void *workspace, rnntOptions options) {
    if (!check_common(acts, labels, act_lens, label_lens, alphabet_size, minibatch, workspace, options) ||
        !costs)
        return RNNT_STATUS_INVALID_VALUE;
    if (reduction < RNNT_REDUCTION_NONE || reduction > RNNT_REDUCTION_SUM) return RNNT_STATUS_INVALID_VALUE;
    const Lattice w = carve(workspace, options.maxT, options.maxU, minibatch);
    record(options, 0);
    rnntStatus_t s;
    switch (options.dtype) {
        case RNNT_DTYPE_F32:
            s = launch_k1<F32>(acts, labels, act_lens, label_lens, alphabet_size, minibatch, w, options);
            break;
        case RNNT_DTYPE_BF16:
            s = launch_k1<BF16>(acts, labels, act_lens, label_lens, alphabet_size, minibatch, w, options);
            break;
        default:
            s = launch_k1<F16>(acts, labels, act_lens, label_lens, alphabet_size, minibatch, w, options);
    }
    if (s != RNNT_STATUS_SUCCESS) return s;
    return launch_lattice_and_costs(act_lens, label_lens, minibatch, costs, loss_out, reduction, w, options);
}

rnntStatus_t rnnt_backward(const void *acts, void *grads, const int *labels, const int *act_lens,
                           const int *label_lens, int alphabet_size, int minibatch, const float *grad_out,
                           int grad_out_stride, float grad_scale, const void *workspace,
                           rnntOptions options) {
    if (!check_common(acts, labels, act_lens, label_lens, alphabet_size, minibatch, workspace, options) ||
        !grads)
        return RNNT_STATUS_INVALID_VALUE;
    if (grad_out_stride < 0) return RNNT_STATUS_INVALID_VALUE;
    const Lattice w = carve(const_cast<void *>(workspace), options.maxT, options.maxU, minibatch);
    record(options, 0);
    rnntStatus_t s;
    switch (options.dtype) {
        case RNNT_DTYPE_F32:
            s = launch_k3<F32>(acts, grads, labels, act_lens, label_lens, alphabet_size, minibatch, grad_out,
                               grad_out_stride, grad_scale, w, options);
            break;
        case RNNT_DTYPE_BF16:
            s = launch_k3<BF16>(acts, grads, labels, act_lens, label_lens, alphabet_size, minibatch, grad_out,
                                grad_out_stride, grad_scale, w, options);
            break;
        default:
            s = launch_k3<F16>(acts, grads, labels, act_lens, label_lens, alphabet_size, minibatch, grad_out,
                               grad_out_stride, grad_scale, w, options);
    }
    if (s != RNNT_STATUS_SUCCESS) return s;
    record(options, 1);
    return RNNT_STATUS_SUCCESS;
}

// ---- the three passes on logits kept in the COMPACTED node order (one row per lattice node; include/rnnt_mi355.h) ----
rnntStatus_t rnnt_compact_rows(const int *act_lens, const int *label_lens, int minibatch, int maxT, int maxU,
                               long long capacity, int *rowmap, void *scratch, void *stream) {
    if (!act_lens || !label_lens || !rowmap || !scratch || minibatch <= 0 || maxT <= 0 || maxU <= 0 || capacity <= 0)
        return RNNT_STATUS_INVALID_VALUE;
    if (reinterpret_cast<uintptr_t>(scratch) % 16 != 0 || (long long)minibatch * maxT * maxU > 0x7fffffffLL)
        return RNNT_STATUS_INVALID_VALUE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    NodeSpan *spans = static_cast<NodeSpan *>(scratch);
    const long long rows = (long long)minibatch * maxT * maxU;
    hipLaunchKernelGGL(rnnt_node_spans, dim3(1), dim3(kWave), 0, st, act_lens, label_lens, minibatch, maxT, maxU, spans);
    hipLaunchKernelGGL(rnnt_compact_rows_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, rowmap, spans,
                       static_cast<float2 *>(nullptr), static_cast<float2 *>(nullptr), 0, 0, maxT, maxU, rows, capacity);
    return hipGetLastError() == hipSuccess ? RNNT_STATUS_SUCCESS : RNNT_STATUS_EXECUTION_FAILED;
}

rnntStatus_t rnnt_forward_compact(const void *acts, long long acts_rows, const int *labels, const int *act_lens,
                                  const int *label_lens, int alphabet_size, int minibatch, float *costs, float *loss_out,
                                  int reduction, void *workspace, rnntOptions options) {
    if (!check_common(acts, labels, act_lens, label_lens, alphabet_size, minibatch, workspace, options) || !costs ||
        acts_rows <= 0)
        return RNNT_STATUS_INVALID_VALUE;
    if (reduction < RNNT_REDUCTION_NONE || reduction > RNNT_REDUCTION_SUM) return RNNT_STATUS_INVALID_VALUE;
    const Lattice w = carve(workspace, options.maxT, options.maxU, minibatch);
    record(options, 0);
    rnntStatus_t s;
    switch (options.dtype) {
        case RNNT_DTYPE_F32:
            s = launch_k1<F32>(acts, labels, act_lens, label_lens, alphabet_size, minibatch, w, options, acts_rows);
            break;
        case RNNT_DTYPE_BF16:
            s = launch_k1<BF16>(acts, labels, act_lens, label_lens, alphabet_size, minibatch, w, options, acts_rows);
            break;
        default:
            s = launch_k1<F16>(acts, labels, act_lens, label_lens, alphabet_size, minibatch, w, options, acts_rows);
    }
    if (s != RNNT_STATUS_SUCCESS) return s;
    // nodes past acts_rows (a buffer smaller than the lattices): NaN lattice entries, hence NaN costs
    s = rnnt_mi355_detail::launch_compact_rows(nullptr, w, minibatch, acts_rows, options);
    if (s != RNNT_STATUS_SUCCESS) return s;
    return launch_lattice_and_costs(act_lens, label_lens, minibatch, costs, loss_out, reduction, w, options);
}

rnntStatus_t rnnt_backward_compact(const void *acts, void *grads, long long acts_rows, const int *labels,
                                   const int *act_lens, const int *label_lens, int alphabet_size, int minibatch,
                                   const float *grad_out, int grad_out_stride, float grad_scale, const void *workspace,
                                   rnntOptions options) {
    if (!check_common(acts, labels, act_lens, label_lens, alphabet_size, minibatch, workspace, options) || !grads ||
        acts_rows <= 0)
        return RNNT_STATUS_INVALID_VALUE;
    if (grad_out_stride < 0) return RNNT_STATUS_INVALID_VALUE;
    const Lattice w = carve(const_cast<void *>(workspace), options.maxT, options.maxU, minibatch);
    record(options, 0);
    rnntStatus_t s;
    switch (options.dtype) {
        case RNNT_DTYPE_F32:
            s = launch_k3<F32>(acts, grads, labels, act_lens, label_lens, alphabet_size, minibatch, grad_out,
                               grad_out_stride, grad_scale, w, options, acts_rows);
            break;
        case RNNT_DTYPE_BF16:
            s = launch_k3<BF16>(acts, grads, labels, act_lens, label_lens, alphabet_size, minibatch, grad_out,
                                grad_out_stride, grad_scale, w, options, acts_rows);
            break;
        default:
            s = launch_k3<F16>(acts, grads, labels, act_lens, label_lens, alphabet_size, minibatch, grad_out,
                               grad_out_stride, grad_scale, w, options, acts_rows);
    }
    if (s != RNNT_STATUS_SUCCESS) return s;
    record(options, 1);
    return RNNT_STATUS_SUCCESS;
}

rnntStatus_t compute_rnnt_loss(const void *acts, void *grads, const int *labels, const int *label_lens,
                               const int *act_lens, int alphabet_size, int minibatch, float *costs,
                               void *workspace, rnntOptions options) {
    rnntOptions fo = options;
    fo.stage_events = nullptr;
    rnntStatus_t s = rnnt_forward(acts, labels, act_lens, label_lens, alphabet_size, minibatch, costs,
                                  nullptr, RNNT_REDUCTION_NONE, workspace, fo);
    if (s != RNNT_STATUS_SUCCESS || !grads) return s;
    return rnnt_backward(acts, grads, labels, act_lens, label_lens, alphabet_size, minibatch, nullptr, 0,
                         1.0f, workspace, fo);
}

// ---------------------------------------------------------------------------------------------
// compute_rnnt_loss for callers that cannot allocate (HIP graph capture, serving loops): the
// lattice workspace is a file-scope device array.  Every GPU the library runs on holds its own
// copy at its own address; `scratch` itself is only a host-side handle for hipGetSymbolAddress and
// must never be passed to a kernel or a memset.
// ---------------------------------------------------------------------------------------------
}  // extern "C"
namespace {
static __device__ __align__(256) unsigned char scratch[RNNT_STATIC_WORKSPACE_BYTES];

constexpr int kMaxDevices = 64;
std::atomic<void *> g_scratch_ptr[kMaxDevices];  // device address of `scratch` per device ordinal

// Address of the CURRENT device's copy, looked up once per device (hipGetSymbolAddress is
// idempotent, so two threads racing through the first call store the same value).
rnntStatus_t static_workspace(void **out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return RNNT_STATUS_EXECUTION_FAILED;
    void *p = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!p) {
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return RNNT_STATUS_EXECUTION_FAILED;
        g_scratch_ptr[dev].store(p, std::memory_order_release);
    }
    *out = p;
    return RNNT_STATUS_SUCCESS;
}
}  // namespace
extern "C" {

rnntStatus_t compute_rnnt_loss_static(const void *acts, void *grads, const int *labels, const int *label_lens,
                                      const int *act_lens, int alphabet_size, int minibatch, float *costs,
                                      rnntOptions options) {
    size_t need = 0;
    const rnntStatus_t q = rnnt_get_workspace_size(options.maxT, options.maxU, minibatch, &need);
    if (q != RNNT_STATUS_SUCCESS) return q;
    if (need > (size_t)RNNT_STATIC_WORKSPACE_BYTES) return RNNT_STATUS_STATIC_WORKSPACE_EXCEEDED;
    void *ws = nullptr;
    const rnntStatus_t s = static_workspace(&ws);
    if (s != RNNT_STATUS_SUCCESS) return s;
    // K1 and K2 write every lattice value K2 and K3 read (the workspace of the allocating entry
    // points is uninitialised memory too), so nothing has to be cleared between calls.
    return compute_rnnt_loss(acts, grads, labels, label_lens, act_lens, alphabet_size, minibatch, costs, ws,
                             options);
}

rnntStatus_t rnnt_reduce_costs(const float *costs, float *loss_out, int minibatch, int reduction,
                               void *stream) {
    if (!costs || !loss_out || minibatch <= 0) return RNNT_STATUS_INVALID_VALUE;
    if (reduction < RNNT_REDUCTION_NONE || reduction > RNNT_REDUCTION_SUM) return RNNT_STATUS_INVALID_VALUE;
    hipLaunchKernelGGL(rnnt_reduce_costs_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), costs,
                       loss_out, minibatch, reduction);
    return hipGetLastError() == hipSuccess ? RNNT_STATUS_SUCCESS : RNNT_STATUS_EXECUTION_FAILED;
}

rnntStatus_t rnnt_event_create(void **event) {
    if (!event) return RNNT_STATUS_INVALID_VALUE;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return RNNT_STATUS_EXECUTION_FAILED;
    *event = e;
    return RNNT_STATUS_SUCCESS;
}
rnntStatus_t rnnt_event_destroy(void *event) {
    return hipEventDestroy(static_cast<hipEvent_t>(event)) == hipSuccess ? RNNT_STATUS_SUCCESS
                                                                         : RNNT_STATUS_EXECUTION_FAILED;
}
rnntStatus_t rnnt_event_record(void *event, void *stream) {
    return hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream)) == hipSuccess
               ? RNNT_STATUS_SUCCESS
               : RNNT_STATUS_EXECUTION_FAILED;
}
rnntStatus_t rnnt_event_elapsed_ms(void *start, void *stop, float *ms) {
    if (!ms) return RNNT_STATUS_INVALID_VALUE;
    if (hipEventSynchronize(static_cast<hipEvent_t>(stop)) != hipSuccess) return RNNT_STATUS_EXECUTION_FAILED;
    return hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)) == hipSuccess
               ? RNNT_STATUS_SUCCESS
               : RNNT_STATUS_EXECUTION_FAILED;
}

}  // extern "C"

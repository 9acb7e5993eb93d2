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

#include "../../include/rnnt_mi355.h"

namespace {

constexpr int kWave = 64;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr int kAbiVersion = 1;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

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
        // plain casts: hipcc emits v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN stays NaN)
        const __bf16 a = (__bf16)lo, b = (__bf16)hi;
        return (unsigned)__builtin_bit_cast(unsigned short, a) |
               ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
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
        const _Float16 a = (_Float16)lo, b = (_Float16)hi;
        return (unsigned)__builtin_bit_cast(unsigned short, a) |
               ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
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

struct RowCoord {
    int b, t, u;
};
__device__ __forceinline__ RowCoord decode_row(long long r, int maxT, int U1) {
    const long long per_b = (long long)maxT * U1;
    RowCoord c;
    c.b = (int)(r / per_b);
    const int rem = (int)(r - (long long)c.b * per_b);
    c.t = rem / U1;
    c.u = rem - c.t * U1;
    return c;
}
// Lengths that do not fit the logits tensor never reach an address computation: the utterance is
// treated as all padding by K1/K3 and K2 reports a NaN cost (rnnt_mi355.h: "NaN, not an error").
__device__ __forceinline__ bool lens_ok(int Tb, int Ub, int maxT, int U1) {
    return Tb >= 1 && Tb <= maxT && Ub >= 0 && Ub < U1;
}
__device__ __forceinline__ size_t skew_index(int b, int t, int u, int ND, int U1) {
    return ((size_t)b * ND + (size_t)(t + u)) * U1 + u;
}

// ---------------------------------------------------------------------------------------------
// K1: log-softmax denominators; keep blank and next-label log-probs.  
// Bound: HBM read of N_valid elements.  One wave per node, NCH 16-byte chunks per lane.
// ---------------------------------------------------------------------------------------------
template <typename DT, int NCH>
__global__ __launch_bounds__(256) void rnnt_k1_lse(const typename DT::elem *__restrict__ acts,
                                                   const int *__restrict__ labels,
                                                   const int *__restrict__ act_lens,
                                                   const int *__restrict__ label_lens,
                                                   float *__restrict__ lse, float *__restrict__ lpB,
                                                   float *__restrict__ lpY, int maxT, int U1, int V,
                                                   int blank, int ND, long long total_rows) {
    constexpr int EPC = DT::EPC;
    const int lane = threadIdx.x & (kWave - 1);
    const long long wave0 = (long long)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    const long long nwaves = (long long)gridDim.x * (blockDim.x / kWave);
    const int nchunks = V / EPC;  // host guarantees V % EPC == 0 and nchunks <= NCH*64

    for (long long r = wave0; r < total_rows; r += nwaves) {
        const RowCoord c = decode_row(r, maxT, U1);
        const int Tb = act_lens[c.b];
        const int Ub = label_lens[c.b];
        if (!lens_ok(Tb, Ub, maxT, U1) || c.t >= Tb || c.u > Ub) continue;  // padded node: never read

        const typename DT::elem *row = acts + (size_t)r * (size_t)V;
        const u32x4 *row4 = reinterpret_cast<const u32x4 *>(row);
        u32x4 raw[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int idx = lane + k * kWave;
            if (idx < nchunks) raw[k] = __builtin_nontemporal_load(row4 + idx);
        }
        // blank / label logits: tiny extra loads that hit the lines fetched above
        const int y = (c.u < Ub) ? labels[(size_t)c.b * (U1 - 1) + c.u] : blank;
        const float xb = DT::to_float(row[blank]);
        const float xy = (y >= 0 && y < V) ? DT::to_float(row[y]) : __builtin_nanf("");

        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (lane + k * kWave < nchunks) {
                float x[EPC];
                DT::unpack(raw[k], x);
#pragma unroll
                for (int j = 0; j < EPC; ++j) m = fmaxf(m, x[j]);
            }
        }
        m = wave_max(m);
        // all -inf row (or +inf max): avoid inf-inf; the result is NaN/-inf downstream as in torch
        const float mneg = (m == -INFINITY || m == INFINITY) ? 0.0f : -m * kLog2e;
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (lane + k * kWave < nchunks) {
                float x[EPC];
                DT::unpack(raw[k], x);
#pragma unroll
                for (int j = 0; j < EPC; ++j) s += __builtin_amdgcn_exp2f(fmaf(x[j], kLog2e, mneg));
            }
        }
        s = wave_sum(s);
        if (lane == 0) {
            const float logs = __logf(s);
            const size_t o = skew_index(c.b, c.t, c.u, ND, U1);
            const float mm = (m == -INFINITY || m == INFINITY) ? 0.0f : m;
            lse[o] = mm + logs;
            // same association as torch.log_softmax on the upstream CPU path: (x - max) - log(sum)
            lpB[o] = (xb - mm) - logs;
            lpY[o] = (xy - mm) - logs;
        }
    }
}

// Fallback for any V / alignment (element-granular loads, two passes over an L2-resident row).
template <typename DT>
__global__ __launch_bounds__(256) void rnnt_k1_lse_generic(
    const typename DT::elem *__restrict__ acts, const int *__restrict__ labels,
    const int *__restrict__ act_lens, const int *__restrict__ label_lens, float *__restrict__ lse,
    float *__restrict__ lpB, float *__restrict__ lpY, int maxT, int U1, int V, int blank, int ND,
    long long total_rows) {
    const int lane = threadIdx.x & (kWave - 1);
    const long long wave0 = (long long)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    const long long nwaves = (long long)gridDim.x * (blockDim.x / kWave);
    for (long long r = wave0; r < total_rows; r += nwaves) {
        const RowCoord c = decode_row(r, maxT, U1);
        const int Tb = act_lens[c.b];
        const int Ub = label_lens[c.b];
        if (!lens_ok(Tb, Ub, maxT, U1) || c.t >= Tb || c.u > Ub) continue;
        const typename DT::elem *row = acts + (size_t)r * (size_t)V;
        float m = -INFINITY;
        for (int v = lane; v < V; v += kWave) m = fmaxf(m, DT::to_float(row[v]));
        m = wave_max(m);
        const float mm = (m == -INFINITY || m == INFINITY) ? 0.0f : m;
        float s = 0.0f;
        for (int v = lane; v < V; v += kWave)
            s += __builtin_amdgcn_exp2f(fmaf(DT::to_float(row[v]), kLog2e, -mm * kLog2e));
        s = wave_sum(s);
        if (lane == 0) {
            const int y = (c.u < Ub) ? labels[(size_t)c.b * (U1 - 1) + c.u] : blank;
            const float logs = __logf(s);
            const size_t o = skew_index(c.b, c.t, c.u, ND, U1);
            lse[o] = mm + logs;
            lpB[o] = (DT::to_float(row[blank]) - mm) - logs;
            lpY[o] = (((y >= 0 && y < V) ? DT::to_float(row[y]) : __builtin_nanf("")) - mm) - logs;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K2: alpha / beta over anti-diagonals.  
// Latency-bound: T+U dependent steps.  Block = 2*NW waves; waves [0,NW) alpha, [NW,2NW) beta.
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

// log(e^a + e^b), same arrangement as the upstream CPU helper: big + log1p(exp(small - big)).
__device__ __forceinline__ float log_add_exp(float a, float b) {
    const float big = fmaxf(a, b);
    const float small = fminf(a, b);
    if (small == -INFINITY) return big;
    return big + log1pf(__expf(small - big));
}

constexpr int kPF = 16;  // diagonals of lp_blank/lp_label kept in flight per lane

template <int NW>
__global__ __launch_bounds__(2 * NW * kWave) void rnnt_k2_lattice(
    const float *__restrict__ lpB, const float *__restrict__ lpY, float *__restrict__ alpha,
    float *__restrict__ beta, float *__restrict__ ll, float *__restrict__ costs,
    const int *__restrict__ act_lens, const int *__restrict__ label_lens, int U1, int ND) {
    __shared__ float xch[2][2][NW > 1 ? NW : 1];  // [direction][step parity][wave]: edge lanes
    const int b = blockIdx.x;
    const int T = act_lens[b];
    const int U = label_lens[b];  // labels; lattice columns 0..U
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int dir = wave / NW;
    const int w = wave - dir * NW;
    const int u = w * kWave + lane;
    if (!lens_ok(T, U, ND - U1 + 1, U1)) {  // block-uniform exit, before any barrier
        if (threadIdx.x == 0) {
            ll[b] = __builtin_nanf("");
            costs[b] = __builtin_nanf("");
        }
        return;
    }
    const int nd = T + U;  // diagonals 0 .. nd-1
    const size_t base = (size_t)b * ND * U1;
    const bool col_ok = (u <= U);

    if (dir == 0) {
        // ---- alpha: diagonal d needs row d-1 of lpB (column u) and lpY (column u-1)
        float pb[kPF], py[kPF];
#pragma unroll
        for (int i = 0; i < kPF; ++i) {
            const int dd = i - 1;  // row read at step i
            const bool ok = col_ok && dd >= 0 && dd < nd;
            pb[i] = ok ? lpB[base + (size_t)dd * U1 + u] : 0.0f;
            py[i] = (ok && u > 0) ? lpY[base + (size_t)dd * U1 + u - 1] : 0.0f;
        }
        float a_prev = -INFINITY;
        for (int d0 = 0; d0 < nd; d0 += kPF) {
#pragma unroll
            for (int i = 0; i < kPF; ++i) {
                const int d = d0 + i;
                const float cb = pb[i], cy = py[i];
                {  // refill this ring slot for step d + kPF
                    const int dd = d + kPF - 1;
                    const bool ok = col_ok && dd < nd;
                    pb[i] = ok ? lpB[base + (size_t)dd * U1 + u] : 0.0f;
                    py[i] = (ok && u > 0) ? lpY[base + (size_t)dd * U1 + u - 1] : 0.0f;
                }
                if (d < nd) {  // block-uniform
                    float left = wave_shift_up(a_prev, -INFINITY);
                    if (NW > 1) {
                        if (lane == 0 && w > 0 && d > 0) left = xch[0][(d - 1) & 1][w - 1];
                    }
                    const int t = d - u;
                    const bool active = col_ok && t >= 0 && t < T;
                    const float up_term = a_prev + cb;
                    const float left_term = left + cy;
                    float a;
                    if (t == 0)
                        a = (u == 0) ? 0.0f : left_term;
                    else
                        a = (u == 0) ? up_term : log_add_exp(left_term, up_term);
                    a = active ? a : -INFINITY;
                    if (active) alpha[base + (size_t)d * U1 + u] = a;
                    a_prev = a;
                    if (NW > 1) {
                        if (lane == kWave - 1) xch[0][d & 1][w] = a;
                        __syncthreads();
                    }
                }
            }
        }
        // the last diagonal holds alpha(T-1,U) in lane u == U
        if (u == U) {
            const float v = a_prev + lpB[base + (size_t)(nd - 1) * U1 + U];
            ll[b] = v;
            costs[b] = -v;
        }
    } else {
        // ---- beta: diagonal d (descending) needs row d of lpB and lpY (column u)
        float pb[kPF], py[kPF];
#pragma unroll
        for (int i = 0; i < kPF; ++i) {
            const int dd = nd - 1 - i;
            const bool ok = col_ok && dd >= 0;
            pb[i] = ok ? lpB[base + (size_t)dd * U1 + u] : 0.0f;
            py[i] = ok ? lpY[base + (size_t)dd * U1 + u] : 0.0f;
        }
        float b_prev = -INFINITY;
        for (int s0 = 0; s0 < nd; s0 += kPF) {
#pragma unroll
            for (int i = 0; i < kPF; ++i) {
                const int s = s0 + i;
                const int d = nd - 1 - s;
                const float cb = pb[i], cy = py[i];
                {
                    const int dd = d - kPF;
                    const bool ok = col_ok && dd >= 0;
                    pb[i] = ok ? lpB[base + (size_t)dd * U1 + u] : 0.0f;
                    py[i] = ok ? lpY[base + (size_t)dd * U1 + u] : 0.0f;
                }
                if (s < nd) {
                    float right = wave_shift_down(b_prev, -INFINITY);
                    if (NW > 1) {
                        if (lane == kWave - 1 && w < NW - 1 && s > 0) right = xch[1][(s - 1) & 1][w + 1];
                    }
                    const int t = d - u;
                    const bool active = col_ok && t >= 0 && t < T;
                    const float down_term = b_prev + cb;   // beta(t+1,u) + lp_blank(t,u)
                    const float right_term = right + cy;   // beta(t,u+1) + lp_label(t,u)
                    float v;
                    if (t == T - 1)
                        v = (u == U) ? cb : right_term;
                    else
                        v = (u == U) ? down_term : log_add_exp(right_term, down_term);
                    v = active ? v : -INFINITY;
                    if (active) beta[base + (size_t)d * U1 + u] = v;
                    b_prev = v;
                    if (NW > 1) {
                        if (lane == 0) xch[1][s & 1][w] = v;
                        __syncthreads();
                    }
                }
            }
        }
    }
}

// costs -> loss_out, fixed summation order (bitwise reproducible, identical on every rank)
__global__ void rnnt_reduce_costs_kernel(const float *__restrict__ costs, float *__restrict__ loss_out, int B,
                                  int reduction) {
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
//   g_v = scale * ( softmax_v * e^{alpha+beta-ll}  - [v=blank] Bk - [v=y_u] Yk )
// ---------------------------------------------------------------------------------------------
template <typename DT, int NCH>
__global__ __launch_bounds__(256) void rnnt_k3_grad(
    const typename DT::elem *__restrict__ acts, typename DT::elem *__restrict__ grads,
    const int *__restrict__ labels, const int *__restrict__ act_lens,
    const int *__restrict__ label_lens, const float *__restrict__ lse, const float *__restrict__ lpB,
    const float *__restrict__ lpY, const float *__restrict__ alpha, const float *__restrict__ beta,
    const float *__restrict__ ll, const float *__restrict__ grad_out, int grad_out_stride,
    float grad_scale, int maxT, int U1, int V, int blank, int ND, long long total_rows) {
    constexpr int EPC = DT::EPC;
    const int lane = threadIdx.x & (kWave - 1);
    const long long wave0 = (long long)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    const long long nwaves = (long long)gridDim.x * (blockDim.x / kWave);
    const int nchunks = V / EPC;

    for (long long r = wave0; r < total_rows; r += nwaves) {
        const RowCoord c = decode_row(r, maxT, U1);
        const int Tb = act_lens[c.b];
        const int Ub = label_lens[c.b];
        u32x4 *out4 = reinterpret_cast<u32x4 *>(grads + (size_t)r * (size_t)V);
        if (!lens_ok(Tb, Ub, maxT, U1) || c.t >= Tb || c.u > Ub) {
            const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int idx = lane + k * kWave;
                if (idx < nchunks) __builtin_nontemporal_store(z, out4 + idx);
            }
            continue;
        }
        const u32x4 *row4 = reinterpret_cast<const u32x4 *>(acts + (size_t)r * (size_t)V);
        u32x4 raw[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int idx = lane + k * kWave;
            if (idx < nchunks) raw[k] = __builtin_nontemporal_load(row4 + idx);
        }
        // per-node scalars (wave-uniform addresses)
        const size_t o = skew_index(c.b, c.t, c.u, ND, U1);
        const float llb = ll[c.b];
        const float a = alpha[o];
        const float scale = grad_scale * (grad_out ? grad_out[(size_t)c.b * grad_out_stride] : 1.0f);
        // Same association as the upstream CPU path: alpha + beta_next + lp - ll, and the dense
        // softmax term is weighted by (Bk + Yk) rather than by e^{alpha+beta-ll}.
        float bk = 0.0f, yk = 0.0f;
        int y = -1;
        if (c.t < Tb - 1)
            bk = __expf(a + beta[o + U1] + lpB[o] - llb);  // beta(t+1,u): next diagonal, same column
        else if (c.u == Ub)
            bk = __expf(a + lpB[o] - llb);
        if (c.u < Ub) {
            y = labels[(size_t)c.b * (U1 - 1) + c.u];
            yk = __expf(a + beta[o + U1 + 1] + lpY[o] - llb);  // beta(t,u+1): next diagonal, column+1
        }
        const float nl2 = -lse[o] * kLog2e;   // softmax_v = 2^(x*log2e + nl2)
        const float sa = scale * (bk + yk);
        const float sbk = scale * bk, syk = scale * yk;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int idx = lane + k * kWave;
            if (idx < nchunks) {
                float x[EPC];
                DT::unpack(raw[k], x);
                const int v0 = idx * EPC;
#pragma unroll
                for (int j = 0; j < EPC; ++j) {
                    float g = sa * __builtin_amdgcn_exp2f(fmaf(x[j], kLog2e, nl2));
                    g -= (v0 + j == blank) ? sbk : 0.0f;
                    g -= (v0 + j == y) ? syk : 0.0f;
                    x[j] = g;
                }
                __builtin_nontemporal_store(DT::pack(x), out4 + idx);
            }
        }
    }
}

template <typename DT>
__global__ __launch_bounds__(256) void rnnt_k3_grad_generic(
    const typename DT::elem *__restrict__ acts, typename DT::elem *__restrict__ grads,
    const int *__restrict__ labels, const int *__restrict__ act_lens,
    const int *__restrict__ label_lens, const float *__restrict__ lse, const float *__restrict__ lpB,
    const float *__restrict__ lpY, const float *__restrict__ alpha, const float *__restrict__ beta,
    const float *__restrict__ ll, const float *__restrict__ grad_out, int grad_out_stride,
    float grad_scale, int maxT, int U1, int V, int blank, int ND, long long total_rows) {
    const int lane = threadIdx.x & (kWave - 1);
    const long long wave0 = (long long)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    const long long nwaves = (long long)gridDim.x * (blockDim.x / kWave);
    for (long long r = wave0; r < total_rows; r += nwaves) {
        const RowCoord c = decode_row(r, maxT, U1);
        const int Tb = act_lens[c.b];
        const int Ub = label_lens[c.b];
        typename DT::elem *out = grads + (size_t)r * (size_t)V;
        if (!lens_ok(Tb, Ub, maxT, U1) || c.t >= Tb || c.u > Ub) {
            for (int v = lane; v < V; v += kWave) out[v] = DT::from_float(0.0f);
            continue;
        }
        const typename DT::elem *row = acts + (size_t)r * (size_t)V;
        const size_t o = skew_index(c.b, c.t, c.u, ND, U1);
        const float llb = ll[c.b];
        const float a = alpha[o];
        const float scale = grad_scale * (grad_out ? grad_out[(size_t)c.b * grad_out_stride] : 1.0f);
        float bk = 0.0f, yk = 0.0f;
        int y = -1;
        if (c.t < Tb - 1)
            bk = __expf(a + beta[o + U1] + lpB[o] - llb);
        else if (c.u == Ub)
            bk = __expf(a + lpB[o] - llb);
        if (c.u < Ub) {
            y = labels[(size_t)c.b * (U1 - 1) + c.u];
            yk = __expf(a + beta[o + U1 + 1] + lpY[o] - llb);
        }
        const float nl2 = -lse[o] * kLog2e;
        const float sa = scale * (bk + yk);
        for (int v = lane; v < V; v += kWave) {
            float g = sa * __builtin_amdgcn_exp2f(fmaf(DT::to_float(row[v]), kLog2e, nl2));
            g -= (v == blank) ? scale * bk : 0.0f;
            g -= (v == y) ? scale * yk : 0.0f;
            out[v] = DT::from_float(g);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct Workspace {
    float *lse, *lpB, *lpY, *alpha, *beta, *ll;
};

inline size_t align_up(size_t n, size_t a) { return (n + a - 1) / a * a; }

inline size_t plane_floats(int maxT, int maxU, int B) {
    const size_t ND = (size_t)maxT + maxU - 1;
    return align_up((size_t)B * ND * (size_t)maxU + (size_t)maxU + 1, 64);  // +row slack for o+U1+1
}

inline Workspace carve(void *ws, int maxT, int maxU, int B) {
    const size_t p = plane_floats(maxT, maxU, B);
    float *f = static_cast<float *>(ws);
    Workspace w;
    w.lse = f;
    w.lpB = f + p;
    w.lpY = f + 2 * p;
    w.alpha = f + 3 * p;
    w.beta = f + 4 * p;
    w.ll = f + 5 * p;
    return w;
}

inline int elem_size(int dtype) { return dtype == RNNT_DTYPE_F32 ? 4 : 2; }

bool check_common(const void *acts, const int *labels, const int *act_lens, const int *label_lens,
                  int V, int B, const void *ws, const rnntOptions &o) {
    if (!acts || !act_lens || !label_lens || !ws) return false;
    if (!labels && o.maxU > 1) return false;
    if (V <= 0 || B <= 0 || o.maxT <= 0 || o.maxU <= 0) return false;
    if (o.blank_label < 0 || o.blank_label >= V) return false;
    if (o.dtype < RNNT_DTYPE_F32 || o.dtype > RNNT_DTYPE_F16) return false;
    if (reinterpret_cast<uintptr_t>(ws) % 256 != 0) return false;
    if (reinterpret_cast<uintptr_t>(acts) % elem_size(o.dtype) != 0) return false;
    return true;
}

// fast path: 16-byte aligned rows and the whole row fits the register-resident instantiations
constexpr int kMaxNCH = 32;
inline int pick_nch(const void *p0, const void *p1, int V, int dtype) {
    const int epc = 16 / elem_size(dtype);
    if (V % epc != 0) return 0;
    if (reinterpret_cast<uintptr_t>(p0) % 16 != 0) return 0;
    if (p1 && reinterpret_cast<uintptr_t>(p1) % 16 != 0) return 0;
    const int need = (V / epc + kWave - 1) / kWave;
    static const int sizes[] = {1, 2, 4, 6, 8, 12, 17, 24, 32};
    for (int s : sizes)
        if (need <= s) return s;
    return 0;
}

inline int stream_grid(long long total_rows) {
    // grid-stride over lattice nodes: 256 CUs x 8 blocks of 4 waves
    const long long want = (total_rows + 3) / 4;
    return (int)(want < 2048 ? (want < 1 ? 1 : want) : 2048);
}

inline void record(const rnntOptions &o, int i) {
    if (o.stage_events && o.stage_events[i])
        (void)hipEventRecord(static_cast<hipEvent_t>(o.stage_events[i]), static_cast<hipStream_t>(o.stream));
}

#define FOR_EACH_NCH(M) M(1) M(2) M(4) M(6) M(8) M(12) M(17) M(24) M(32)

template <typename DT>
rnntStatus_t launch_k1(const void *acts, const int *labels, const int *act_lens, const int *label_lens,
                       int V, int B, const Workspace &w, const rnntOptions &o) {
    typedef typename DT::elem E;
    const long long rows = (long long)B * o.maxT * o.maxU;
    const int ND = o.maxT + o.maxU - 1;
    const int grid = stream_grid(rows);
    hipStream_t st = static_cast<hipStream_t>(o.stream);
    const int nch = pick_nch(acts, nullptr, V, o.dtype);
    switch (nch) {
#define CASE(N)                                                                                     \
    case N:                                                                                         \
        hipLaunchKernelGGL((rnnt_k1_lse<DT, N>), dim3(grid), dim3(256), 0, st,                      \
                           static_cast<const E *>(acts), labels, act_lens, label_lens, w.lse, w.lpB, \
                           w.lpY, o.maxT, o.maxU, V, o.blank_label, ND, rows);                      \
        break;
        FOR_EACH_NCH(CASE)
#undef CASE
        default:
            hipLaunchKernelGGL((rnnt_k1_lse_generic<DT>), dim3(grid), dim3(256), 0, st,
                               static_cast<const E *>(acts), labels, act_lens, label_lens, w.lse, w.lpB,
                               w.lpY, o.maxT, o.maxU, V, o.blank_label, ND, rows);
    }
    return hipGetLastError() == hipSuccess ? RNNT_STATUS_SUCCESS : RNNT_STATUS_EXECUTION_FAILED;
}

template <typename DT>
rnntStatus_t launch_k3(const void *acts, void *grads, const int *labels, const int *act_lens,
                       const int *label_lens, int V, int B, const float *grad_out, int grad_out_stride,
                       float grad_scale, const Workspace &w, const rnntOptions &o) {
    typedef typename DT::elem E;
    const long long rows = (long long)B * o.maxT * o.maxU;
    const int ND = o.maxT + o.maxU - 1;
    const int grid = stream_grid(rows);
    hipStream_t st = static_cast<hipStream_t>(o.stream);
    const int nch = pick_nch(acts, grads, V, o.dtype);
    switch (nch) {
#define CASE(N)                                                                                        \
    case N:                                                                                            \
        hipLaunchKernelGGL((rnnt_k3_grad<DT, N>), dim3(grid), dim3(256), 0, st,                        \
                           static_cast<const E *>(acts), static_cast<E *>(grads), labels, act_lens,    \
                           label_lens, w.lse, w.lpB, w.lpY, w.alpha, w.beta, w.ll, grad_out,           \
                           grad_out_stride, grad_scale, o.maxT, o.maxU, V, o.blank_label, ND, rows);   \
        break;
        FOR_EACH_NCH(CASE)
#undef CASE
        default:
            hipLaunchKernelGGL((rnnt_k3_grad_generic<DT>), dim3(grid), dim3(256), 0, st,
                               static_cast<const E *>(acts), static_cast<E *>(grads), labels, act_lens,
                               label_lens, w.lse, w.lpB, w.lpY, w.alpha, w.beta, w.ll, grad_out,
                               grad_out_stride, grad_scale, o.maxT, o.maxU, V, o.blank_label, ND, rows);
    }
    return hipGetLastError() == hipSuccess ? RNNT_STATUS_SUCCESS : RNNT_STATUS_EXECUTION_FAILED;
}

rnntStatus_t launch_k2(const int *act_lens, const int *label_lens, int B, float *costs, const Workspace &w,
                       const rnntOptions &o) {
    const int ND = o.maxT + o.maxU - 1;
    const int NW = (o.maxU + kWave - 1) / kWave;
    hipStream_t st = static_cast<hipStream_t>(o.stream);
#define K2(N)                                                                                          \
    hipLaunchKernelGGL((rnnt_k2_lattice<N>), dim3(B), dim3(2 * N * kWave), 0, st, w.lpB, w.lpY, w.alpha, \
                       w.beta, w.ll, costs, act_lens, label_lens, o.maxU, ND)
    switch (NW) {
        case 1: K2(1); break;
        case 2: K2(2); break;
        case 3: K2(3); break;
        case 4: K2(4); break;
        case 5: case 6: K2(6); break;
        case 7: case 8: K2(8); break;
        default: return RNNT_STATUS_INVALID_VALUE;  // maxU > 512 labels+1: not supported
    }
#undef K2
    return hipGetLastError() == hipSuccess ? RNNT_STATUS_SUCCESS : RNNT_STATUS_EXECUTION_FAILED;
}

}  // namespace

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
        case RNNT_STATUS_UNKNOWN_ERROR:
        default: return "unknown error";
    }
}

int rnnt_mi355_abi_version(void) { return kAbiVersion; }

rnntStatus_t rnnt_get_workspace_size(int maxT, int maxU, int minibatch, size_t *size_bytes) {
    if (!size_bytes || maxT <= 0 || maxU <= 0 || minibatch <= 0) return RNNT_STATUS_INVALID_VALUE;
    if (maxU > 512) return RNNT_STATUS_INVALID_VALUE;
    *size_bytes = (5 * plane_floats(maxT, maxU, minibatch) + align_up((size_t)minibatch, 64)) * sizeof(float);
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
    const Workspace w = carve(workspace, options.maxT, options.maxU, minibatch);
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
    record(options, 1);
    s = launch_k2(act_lens, label_lens, minibatch, costs, w, options);
    if (s != RNNT_STATUS_SUCCESS) return s;
    if (loss_out) {
        hipLaunchKernelGGL(rnnt_reduce_costs_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(options.stream),
                           costs, loss_out, minibatch, reduction);
        if (hipGetLastError() != hipSuccess) return RNNT_STATUS_EXECUTION_FAILED;
    }
    record(options, 2);
    return RNNT_STATUS_SUCCESS;
}

rnntStatus_t rnnt_backward(const void *acts, void *grads, const int *labels, const int *act_lens,
                           const int *label_lens, int alphabet_size, int minibatch, const float *grad_out,
                           int grad_out_stride, float grad_scale, const void *workspace,
                           rnntOptions options) {
    if (!check_common(acts, labels, act_lens, label_lens, alphabet_size, minibatch, workspace, options) ||
        !grads)
        return RNNT_STATUS_INVALID_VALUE;
    if (grad_out_stride < 0) return RNNT_STATUS_INVALID_VALUE;
    const Workspace w = carve(const_cast<void *>(workspace), options.maxT, options.maxU, minibatch);
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

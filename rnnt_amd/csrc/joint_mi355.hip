// joint_mi355.hip — the hidden layer of the transducer's joint network on MI355X (gfx950).
//
// The reference's JointNet.forward (/root/reference/src/rnnt/model.py:39-59) repeats the encoder
// state over U+1 and the decoder state over T, concatenates them to [B,T,U+1,He+Hd], and runs
// Linear(He+Hd -> H) + Tanh over all B*T*(U+1) rows.  Linear(cat(e,d)) = W_e e + W_d d + bias, so
// the layer is two small GEMMs (done by the caller: [B*T,He]x[He,H] and [B*(U+1),Hd]x[Hd,H]) and
//     hidden[b,t,u,:] = tanh(enc_proj[b,t,:] + dec_proj[b,u,:])
// which is what these kernels compute, with its backward.  SURVEY.md §8f rank 2.  Both kernels
// are HBM-bound streams: forward writes B*T*(U+1)*H elements once, backward reads hidden and its
// gradient once.  Nothing of size [.., He+Hd] is ever materialised.
//
// Backward:  g = grad_hidden * (1 - hidden^2);  grad_enc_proj[b,t,:] = sum_u g;
//            grad_dec_proj[b,u,:] = sum_t g.
// The sum over t crosses waves: each wave owns a 16-frame chunk of one utterance and a slice of
// columns, reduces its frames in registers (no LDS, no barrier) and writes one fp32 partial row
// piece per (b, frame-chunk, u); a second small kernel adds the partials in chunk order.  No
// atomics, so results are bit-reproducible.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/rnnt_mi355.h"
#include "elem_types.h"

namespace {

using namespace rnnt_mi355_detail;

constexpr int kJW = 4;    // waves per workgroup, forward
constexpr int kJCH = 2;   // 16-byte (or scalar) pieces per lane per row tile
constexpr int kBW = 4;    // waves per workgroup, backward (independent of each other)
constexpr int kTC = 16;   // frames per wave in the backward kernel

// A lane's piece of a row: VEC elements (VEC = DT::EPC through 16-byte accesses, or 1)
template <typename DT, int VEC>
struct Piece {
    float x[VEC];
    __device__ __forceinline__ void load(const typename DT::elem *p) {
        if constexpr (VEC == 1) {
            x[0] = DT::to_float(*p);
        } else {
            const u32x4 raw = *reinterpret_cast<const u32x4 *>(p);
            DT::unpack(raw, x);
        }
    }
    __device__ __forceinline__ void load_nt(const typename DT::elem *p) {
        if constexpr (VEC == 1) {
            x[0] = DT::to_float(__builtin_nontemporal_load(p));
        } else {
            const u32x4 raw = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
            DT::unpack(raw, x);
        }
    }
    __device__ __forceinline__ void store_nt(typename DT::elem *p) const {
        if constexpr (VEC == 1) {
            __builtin_nontemporal_store(DT::from_float(x[0]), p);
        } else {
            __builtin_nontemporal_store(DT::pack(x), reinterpret_cast<u32x4 *>(p));
        }
    }
    __device__ __forceinline__ void store(typename DT::elem *p) const {
        if constexpr (VEC == 1) {
            *p = DT::from_float(x[0]);
        } else {
            *reinterpret_cast<u32x4 *>(p) = DT::pack(x);
        }
    }
};

// backward pieces: 4 elements per lane (16 bytes of fp32, 8 bytes of a 16-bit type) or 1
template <typename DT, int VEC>
__device__ __forceinline__ void load4_nt(const typename DT::elem *p, float (&x)[VEC]) {
    if constexpr (VEC == 1) {
        x[0] = DT::to_float(__builtin_nontemporal_load(p));
    } else if constexpr (sizeof(typename DT::elem) == 4) {
        const u32x4 raw = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
        x[0] = __uint_as_float(raw.x);
        x[1] = __uint_as_float(raw.y);
        x[2] = __uint_as_float(raw.z);
        x[3] = __uint_as_float(raw.w);
    } else {
        typedef typename DT::elem e4 __attribute__((ext_vector_type(4)));
        const e4 raw = __builtin_nontemporal_load(reinterpret_cast<const e4 *>(p));
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = DT::to_float(raw[j]);
    }
}
template <typename DT, int VEC>
__device__ __forceinline__ void store4(typename DT::elem *p, const float (&x)[VEC]) {
    if constexpr (VEC == 1) {
        *p = DT::from_float(x[0]);
    } else {
        typedef typename DT::elem e4 __attribute__((ext_vector_type(4)));
        e4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = DT::from_float(x[j]);
        *reinterpret_cast<e4 *>(p) = v;
    }
}

// tanh for the forward pass.  fp32 outputs use the device libm (about 1 ulp, as torch's tanh).  16-bit
// outputs are rounded to 2^-9 / 2^-12 relative anyway, so they take (1-t)/(1+t), t = e^(-2|x|), whose
// absolute error is ~1e-7: two transcendental instructions instead of libm's ~25 VALU slots, which is
// what bounds the 16-bit kernel.
template <typename DT>
__device__ __forceinline__ float joint_tanh(float x) {
    if constexpr (sizeof(typename DT::elem) == 4) {
        return tanhf(x);
    } else {
        const float t = __builtin_amdgcn_exp2f(-2.f * 1.4426950408889634f * fabsf(x));
        const float r = (1.f - t) * __builtin_amdgcn_rcpf(1.f + t);
        return copysignf(r, x);  // NaN in, NaN out; +-inf -> +-1
    }
}

// Compacted layout (rnnt_joint_hidden_*_compact): `hidden` holds one row per lattice NODE, in the order the loss's
// compacted kernels walk (lattice.h: utterance by utterance, frame by frame, label position fastest; an utterance whose
// lengths do not fit the tensor has no nodes).  Number of nodes of the utterances before b, the same in every lane.
__device__ __forceinline__ long long nodes_before(const int *__restrict__ act_lens, const int *__restrict__ label_lens,
                                                  int b, int T, int U1, int lane) {
    long long s = 0;
    for (int i = lane; i < b; i += kWave) {
        const int Tb = act_lens[i], Ub = label_lens[i];
        if (Tb >= 1 && Tb <= T && Ub >= 0 && Ub < U1) s += (long long)Tb * (Ub + 1);
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, kWave);
    return s;
}

// ---------------------------------------------------------------------------------------------
// forward: one wave per (b,t); the enc_proj piece stays in registers while u runs over U1
// grid = (ceil(B*T / kJW), ceil(H / CW)),  CW = 64 * VEC * kJCH columns per tile
// ---------------------------------------------------------------------------------------------
template <typename DT, int VEC>
__global__ __launch_bounds__(kJW *kWave) void joint_hidden_fwd(const typename DT::elem *__restrict__ enc_proj,
                                                                const typename DT::elem *__restrict__ dec_proj,
                                                                typename DT::elem *__restrict__ hidden,
                                                                long long BT, int T, int U1, int H,
                                                                const int *__restrict__ act_lens,
                                                                const int *__restrict__ label_lens,
                                                                long long compact_rows) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const long long p = (long long)blockIdx.x * kJW + wave;  // = b*T + t
    if (p >= BT) return;
    const int b = (int)(p / T);
    // ragged call: only the rows inside the utterance's lattice (t < T_b, u <= U_b) are written; an utterance whose
    // lengths do not fit the tensor has none (as everywhere in this library)
    int nu = U1;
    if (act_lens) {
        const int Tb = act_lens[b], Ub = label_lens[b];
        if (!(Tb >= 1 && Tb <= T && Ub >= 0 && Ub < U1) || (int)(p - (long long)b * T) >= Tb) return;  // wave-uniform
        nu = Ub + 1;
    }
    const int col0 = blockIdx.y * (kWave * VEC * kJCH) + lane * VEC;

    Piece<DT, VEC> e[kJCH];
    bool ok[kJCH];
#pragma unroll
    for (int k = 0; k < kJCH; ++k) {
        const int c = col0 + k * kWave * VEC;
        ok[k] = c < H;
        if (ok[k]) {
            e[k].load(enc_proj + (size_t)p * H + c);
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) e[k].x[j] = 0.f;
        }
    }
    const typename DT::elem *drow = dec_proj + (size_t)b * U1 * H;
    typename DT::elem *hrow = hidden + (size_t)p * U1 * H;
    if (compact_rows) {  // (kernel-uniform; lengths given) one row per node: frame t of utterance b starts at node
        // first_b + t (U_b + 1); a frame the buffer has no room for is skipped
        const long long k0 = nodes_before(act_lens, label_lens, b, T, U1, lane) + (p - (long long)b * T) * nu;
        if (k0 + nu > compact_rows) return;  // wave-uniform
        hrow = hidden + (size_t)k0 * H;
    }
#pragma unroll 4
    for (int u = 0; u < nu; ++u) {
#pragma unroll
        for (int k = 0; k < kJCH; ++k) {
            if (!ok[k]) continue;
            const int c = col0 + k * kWave * VEC;
            Piece<DT, VEC> d, y;
            d.load(drow + (size_t)u * H + c);
#pragma unroll
            for (int j = 0; j < VEC; ++j) y.x[j] = joint_tanh<DT>(e[k].x[j] + d.x[j]);
            y.store_nt(hrow + (size_t)u * H + c);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward: one wave per (b, 16-frame chunk, column slice of 64*VEC); waves are independent
// ---------------------------------------------------------------------------------------------
template <typename DT, int VEC>
__global__ __launch_bounds__(kBW *kWave) void joint_hidden_bwd(const typename DT::elem *__restrict__ hidden,
                                                                const typename DT::elem *__restrict__ grad_hidden,
                                                                typename DT::elem *__restrict__ grad_enc_proj,
                                                                float *__restrict__ partial, long long items, int T,
                                                                int U1, int H, int nchunk, int nslice,
                                                                const int *__restrict__ act_lens,
                                                                const int *__restrict__ label_lens,
                                                                long long compact_rows) {
    const int lane = threadIdx.x & (kWave - 1);
    const long long w = (long long)blockIdx.x * kBW + threadIdx.x / kWave;
    if (w >= items) return;
    const int slice = (int)(w % nslice);
    const long long bc = w / nslice;
    const int chunk = (int)(bc % nchunk);
    const int b = (int)(bc / nchunk);
    const int col = slice * kWave * VEC + lane * VEC;
    const bool ok = col < H;
    const int colc = ok ? col : H - VEC;  // surplus lanes read a valid piece and store nothing
    const int t0 = chunk * kTC;
    const int nt = T - t0 < kTC ? T - t0 : kTC;  // wave-uniform
    // ragged call: rows outside the utterance's lattice are not read (whatever they hold) and count as zero gradient:
    // ntv frames of this chunk and nu label positions lie inside
    int ntv = nt, nu = U1;
    if (act_lens) {
        const int Tb = act_lens[b], Ub = label_lens[b];
        const bool fit = Tb >= 1 && Tb <= T && Ub >= 0 && Ub < U1;
        ntv = fit ? min(max(Tb - t0, 0), nt) : 0;
        nu = fit && ntv > 0 ? Ub + 1 : 0;
    }

    float acc_e[kTC][VEC];
#pragma unroll
    for (int i = 0; i < kTC; ++i)
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc_e[i][j] = 0.f;

    size_t frame_stride = (size_t)U1 * H;
    const typename DT::elem *hp = hidden + ((size_t)b * T + t0) * frame_stride + colc;
    const typename DT::elem *gp = grad_hidden + ((size_t)b * T + t0) * frame_stride + colc;
    float *pp = partial + ((size_t)b * nchunk + chunk) * frame_stride + col;
    if (compact_rows && nu > 0) {  // (lengths given) hidden and its gradient hold one row per node: frames of U_b + 1 rows
        const long long first = nodes_before(act_lens, label_lens, b, T, U1, lane);
        const long long fit_frames = compact_rows > first ? (compact_rows - first) / nu : 0;  // frames the buffer has room for
        ntv = (int)min((long long)ntv, max(fit_frames - t0, 0ll));
        if (ntv == 0) nu = 0;
        frame_stride = (size_t)nu * H;
        hp = hidden + ((size_t)first + (size_t)t0 * nu) * H + colc;
        gp = grad_hidden + ((size_t)first + (size_t)t0 * nu) * H + colc;
    }

    if (ntv == kTC) {  // every chunk but the last: unconditional loads, fully unrolled
        for (int u = 0; u < nu; ++u) {
            float acc_d[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc_d[j] = 0.f;
#pragma unroll
            for (int i = 0; i < kTC; ++i) {
                float hx[VEC], gx[VEC];
                load4_nt<DT, VEC>(hp + i * frame_stride + (size_t)u * H, hx);
                load4_nt<DT, VEC>(gp + i * frame_stride + (size_t)u * H, gx);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float v = gx[j] * (1.f - hx[j] * hx[j]);
                    acc_d[j] += v;
                    acc_e[i][j] += v;
                }
            }
            if (ok) store4<F32, VEC>(pp + (size_t)u * H, acc_d);
        }
    } else if (ntv > 0) {
        // a chunk that ends inside (the last frames of the tensor or, ragged, of the utterance): the same unconditional,
        // fully unrolled loads — a frame past the end reads the chunk's last valid frame again and counts as zero (a
        // select: whatever was read is dropped).  With the loads under a per-frame condition hipcc waits for each one.
        for (int u = 0; u < nu; ++u) {
            float acc_d[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc_d[j] = 0.f;
#pragma unroll
            for (int i = 0; i < kTC; ++i) {
                const int ii = i < ntv ? i : ntv - 1;  // wave-uniform
                float hx[VEC], gx[VEC];
                load4_nt<DT, VEC>(hp + ii * frame_stride + (size_t)u * H, hx);
                load4_nt<DT, VEC>(gp + ii * frame_stride + (size_t)u * H, gx);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float v = i < ntv ? gx[j] * (1.f - hx[j] * hx[j]) : 0.f;
                    acc_d[j] += v;
                    acc_e[i][j] += v;
                }
            }
            if (ok) store4<F32, VEC>(pp + (size_t)u * H, acc_d);
        }
    }
    if (ok) {
        const float zero[VEC] = {};
        for (int u = nu; u < U1; ++u) store4<F32, VEC>(pp + (size_t)u * H, zero);  // (ragged: positions past U_b)
#pragma unroll
        for (int i = 0; i < kTC; ++i)  // (ragged: frames past T_b get their zeros)
            if (i < nt) store4<DT, VEC>(grad_enc_proj + ((size_t)b * T + t0 + i) * H + col, acc_e[i]);
    }
}

// grad_dec_proj[b,u,:] = sum over frame chunks, in chunk order
template <typename DT>
__global__ __launch_bounds__(256) void joint_dec_reduce(const float *__restrict__ partial,
                                                        typename DT::elem *__restrict__ grad_dec_proj, int B,
                                                        int U1, int H, int nchunk) {
    const size_t per_b = (size_t)U1 * H;
    const size_t total = (size_t)B * per_b;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / per_b, r = i - b * per_b;
        const float *p = partial + b * nchunk * per_b + r;
        float s = 0.f;
        for (int c = 0; c < nchunk; ++c) s += p[(size_t)c * per_b];
        grad_dec_proj[i] = DT::from_float(s);
    }
}

// ---------------------------------------------------------------------------------------------
// Row argmax for greedy decoding (the reference's per-frame
// softmax -> argmax -> .item(), src/rnnt/model.py:151-154, over a window of frames at once).
// One wave per row; the FIRST index of the maximum wins, a NaN counts as the maximum — the
// conventions of torch.argmax.
// ---------------------------------------------------------------------------------------------
struct Best {
    float v;   // NaN mapped to +inf
    int nan;   // 1 if the value was a NaN (ranks above +inf)
    int i;
};
__device__ __forceinline__ bool better(const Best &a, const Best &b) {  // a beats b
    if (a.nan != b.nan) return a.nan > b.nan;
    if (a.v != b.v) return a.v > b.v;
    return a.i < b.i;
}
// With `logp` the kernel also gives max(log_softmax(row)) = (x_max - max) - log(sum exp(x - max)), the score the
// reference's batched greedy search adds per emitted symbol (src/rnnt/beam_searcher.py:91-93,101); the row is
// read a second time (it is L2-resident: at most a few tens of KB).  NaN / +inf rows give NaN as in torch.
template <typename DT>
__global__ __launch_bounds__(256) void row_argmax_kernel(const typename DT::elem *__restrict__ x, int rows, int V,
                                                         int *__restrict__ out, float *__restrict__ logp) {
    const int lane = threadIdx.x & (kWave - 1);
    const int r = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (r >= rows) return;
    const typename DT::elem *row = x + (size_t)r * V;
    Best best = {-INFINITY, 0, 0x7fffffff};
    for (int v = lane; v < V; v += kWave) {
        const float f = DT::to_float(row[v]);
        const Best c = {f != f ? INFINITY : f, f != f ? 1 : 0, v};
        if (better(c, best)) best = c;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        Best o;
        o.v = __shfl_xor(best.v, off, kWave);
        o.nan = __shfl_xor(best.nan, off, kWave);
        o.i = __shfl_xor(best.i, off, kWave);
        if (better(o, best)) best = o;
    }
    if (lane == 0) out[r] = best.i;
    if (logp == nullptr) return;
    const float m = best.nan ? __builtin_nanf("") : best.v;
    float sum = 0.0f;
    for (int v = lane; v < V; v += kWave) sum += __expf(DT::to_float(row[v]) - m);
    sum = wave_sum(sum);
    if (lane == 0) logp[r] = (m - m) - __logf(sum);  // m - m: NaN for an infinite maximum, as x - max is in torch
}

// The same for rows that start on 16 bytes (x 16-byte aligned, V a multiple of the chunk's element count): 16-byte
// loads and ONE pass — the running {max, sum 2^((x - max) log2e)} of rnnt_k1_lse beside the running argmax; the two
// lanes' worth of state are merged with butterflies at the end.  Same results as the element-wise kernel above up to
// the order of the fp32 additions.
template <typename DT>
__global__ __launch_bounds__(256) void row_argmax_vec_kernel(const typename DT::elem *__restrict__ x, int rows, int V,
                                                             int *__restrict__ out, float *__restrict__ logp) {
    constexpr int EPC = DT::EPC;
    constexpr float kLog2eF = 1.4426950408889634f, kLn2F = 0.6931471805599453f;
    const int lane = threadIdx.x & (kWave - 1);
    const int r = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (r >= rows) return;
    const u32x4 *row4 = reinterpret_cast<const u32x4 *>(x + (size_t)r * V);
    const int nchunks = V / EPC;
    Best best = {-INFINITY, 0, 0x7fffffff};
    float m = -INFINITY, s = 0.0f;  // over this lane's elements; NaN elements poison s, which is what the result wants
    for (int c = lane; c < nchunks; c += kWave) {
        float f[EPC];
        DT::unpack(row4[c], f);
        float cm = -INFINITY;
#pragma unroll
        for (int j = 0; j < EPC; ++j) {
            const Best cand = {f[j] != f[j] ? INFINITY : f[j], f[j] != f[j] ? 1 : 0, c * EPC + j};
            if (better(cand, best)) best = cand;
            cm = fmaxf(cm, f[j]);
        }
        if (logp != nullptr) {
            const float mn = fmaxf(m, cm);
            if (mn != -INFINITY) {  // (all -inf so far: nothing to add, and -inf - -inf must not be formed)
                float acc = (s == 0.0f) ? 0.0f : s * __builtin_amdgcn_exp2f((m - mn) * kLog2eF);
#pragma unroll
                for (int j = 0; j < EPC; ++j) acc += __builtin_amdgcn_exp2f((f[j] - mn) * kLog2eF);
                s = acc;
                m = mn;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        Best o;
        o.v = __shfl_xor(best.v, off, kWave);
        o.nan = __shfl_xor(best.nan, off, kWave);
        o.i = __shfl_xor(best.i, off, kWave);
        if (better(o, best)) best = o;
    }
    if (lane == 0) out[r] = best.i;
    if (logp == nullptr) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float mo = __shfl_xor(m, off, kWave), so = __shfl_xor(s, off, kWave);
        const float mn = fmaxf(m, mo);
        const float a = (s == 0.0f || m == -INFINITY) ? 0.0f : s * __builtin_amdgcn_exp2f((m - mn) * kLog2eF);
        const float b = (so == 0.0f || mo == -INFINITY) ? 0.0f : so * __builtin_amdgcn_exp2f((mo - mn) * kLog2eF);
        // (both lanes of a pair add in the same order: the lower lane's share first)
        s = (lane & off) ? b + a : a + b;
        m = mn;
    }
    if (lane == 0) {
        const float mm = best.nan ? __builtin_nanf("") : best.v;
        logp[r] = (mm - mm) - kLn2F * __builtin_amdgcn_logf(s);  // mm - mm: NaN for an infinite maximum, as torch
    }
}

inline int frame_chunks(int T) { return (T + kTC - 1) / kTC; }

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline bool shape_ok(int B, int T, int U1, int H) {
    // grid.y / grid.z limits and int row arithmetic
    return B >= 1 && T >= 1 && U1 >= 1 && H >= 1 && B <= 65535 && (long long)B * T < (1ll << 40) &&
           (long long)U1 * H < (1ll << 31);
}

template <typename DT>
rnntStatus_t launch_fwd(const void *enc_proj, const void *dec_proj, void *hidden, int B, int T, int U1, int H,
                        hipStream_t st, const int *act_lens = nullptr, const int *label_lens = nullptr,
                        long long compact_rows = 0) {
    typedef typename DT::elem E;
    const bool vec = H % DT::EPC == 0 && aligned16(enc_proj) && aligned16(dec_proj) && aligned16(hidden);
    const long long BT = (long long)B * T;
    const long long gx = (BT + kJW - 1) / kJW;
    if (gx > 0x7fffffffll) return RNNT_STATUS_INVALID_VALUE;
    if (vec) {
        const int cw = kWave * DT::EPC * kJCH;
        hipLaunchKernelGGL((joint_hidden_fwd<DT, DT::EPC>), dim3((unsigned)gx, (H + cw - 1) / cw), dim3(kJW * kWave), 0,
                           st, static_cast<const E *>(enc_proj), static_cast<const E *>(dec_proj),
                           static_cast<E *>(hidden), BT, T, U1, H, act_lens, label_lens, compact_rows);
    } else {
        const int cw = kWave * kJCH;
        hipLaunchKernelGGL((joint_hidden_fwd<DT, 1>), dim3((unsigned)gx, (H + cw - 1) / cw), dim3(kJW * kWave), 0, st,
                           static_cast<const E *>(enc_proj), static_cast<const E *>(dec_proj),
                           static_cast<E *>(hidden), BT, T, U1, H, act_lens, label_lens, compact_rows);
    }
    return hipGetLastError() == hipSuccess ? RNNT_STATUS_SUCCESS : RNNT_STATUS_EXECUTION_FAILED;
}

inline bool aligned_to(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

template <typename DT>
rnntStatus_t launch_bwd(const void *hidden, const void *grad_hidden, void *grad_enc_proj, void *grad_dec_proj,
                        int B, int T, int U1, int H, void *workspace, hipStream_t st, const int *act_lens = nullptr,
                        const int *label_lens = nullptr, long long compact_rows = 0) {
    typedef typename DT::elem E;
    const size_t piece = 4 * sizeof(E);
    const bool vec = H % 4 == 0 && aligned_to(hidden, piece) && aligned_to(grad_hidden, piece) &&
                     aligned_to(grad_enc_proj, piece);
    const int nchunk = frame_chunks(T);
    float *partial = static_cast<float *>(workspace);
    const int width = kWave * (vec ? 4 : 1);
    const int nslice = (H + width - 1) / width;
    const long long items = (long long)B * nchunk * nslice;
    const long long gx = (items + kBW - 1) / kBW;
    if (gx > 0x7fffffffll) return RNNT_STATUS_INVALID_VALUE;
    if (vec) {
        hipLaunchKernelGGL((joint_hidden_bwd<DT, 4>), dim3((unsigned)gx), dim3(kBW * kWave), 0, st,
                           static_cast<const E *>(hidden), static_cast<const E *>(grad_hidden),
                           static_cast<E *>(grad_enc_proj), partial, items, T, U1, H, nchunk, nslice, act_lens, label_lens,
                           compact_rows);
    } else {
        hipLaunchKernelGGL((joint_hidden_bwd<DT, 1>), dim3((unsigned)gx), dim3(kBW * kWave), 0, st,
                           static_cast<const E *>(hidden), static_cast<const E *>(grad_hidden),
                           static_cast<E *>(grad_enc_proj), partial, items, T, U1, H, nchunk, nslice, act_lens, label_lens,
                           compact_rows);
    }
    if (hipGetLastError() != hipSuccess) return RNNT_STATUS_EXECUTION_FAILED;
    const size_t total = (size_t)B * U1 * H;
    const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL((joint_dec_reduce<DT>), dim3(blocks), dim3(256), 0, st, partial,
                       static_cast<E *>(grad_dec_proj), B, U1, H, nchunk);
    return hipGetLastError() == hipSuccess ? RNNT_STATUS_SUCCESS : RNNT_STATUS_EXECUTION_FAILED;
}

}  // namespace

extern "C" {

rnntStatus_t rnnt_joint_hidden_workspace_size(int B, int T, int U1, int H, size_t *size_bytes) {
    if (size_bytes == nullptr || !shape_ok(B, T, U1, H)) return RNNT_STATUS_INVALID_VALUE;
    *size_bytes = (size_t)B * frame_chunks(T) * U1 * H * sizeof(float);
    return RNNT_STATUS_SUCCESS;
}

rnntStatus_t rnnt_joint_hidden_forward(const void *enc_proj, const void *dec_proj, void *hidden, int B, int T,
                                       int U1, int H, rnntDtype_t dtype, void *stream) {
    if (enc_proj == nullptr || dec_proj == nullptr || hidden == nullptr || !shape_ok(B, T, U1, H))
        return RNNT_STATUS_INVALID_VALUE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case RNNT_DTYPE_F32: return launch_fwd<F32>(enc_proj, dec_proj, hidden, B, T, U1, H, st);
        case RNNT_DTYPE_BF16: return launch_fwd<BF16>(enc_proj, dec_proj, hidden, B, T, U1, H, st);
        case RNNT_DTYPE_F16: return launch_fwd<F16>(enc_proj, dec_proj, hidden, B, T, U1, H, st);
        default: return RNNT_STATUS_INVALID_VALUE;
    }
}

rnntStatus_t rnnt_joint_hidden_backward(const void *hidden, const void *grad_hidden, void *grad_enc_proj,
                                        void *grad_dec_proj, int B, int T, int U1, int H, rnntDtype_t dtype,
                                        void *workspace, void *stream) {
    if (hidden == nullptr || grad_hidden == nullptr || grad_enc_proj == nullptr || grad_dec_proj == nullptr ||
        workspace == nullptr || !shape_ok(B, T, U1, H) || !aligned16(workspace))
        return RNNT_STATUS_INVALID_VALUE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case RNNT_DTYPE_F32:
            return launch_bwd<F32>(hidden, grad_hidden, grad_enc_proj, grad_dec_proj, B, T, U1, H, workspace, st);
        case RNNT_DTYPE_BF16:
            return launch_bwd<BF16>(hidden, grad_hidden, grad_enc_proj, grad_dec_proj, B, T, U1, H, workspace, st);
        case RNNT_DTYPE_F16:
            return launch_bwd<F16>(hidden, grad_hidden, grad_enc_proj, grad_dec_proj, B, T, U1, H, workspace, st);
        default: return RNNT_STATUS_INVALID_VALUE;
    }
}

rnntStatus_t rnnt_joint_hidden_forward_ragged(const void *enc_proj, const void *dec_proj, void *hidden,
                                              const int *act_lens, const int *label_lens, int B, int T, int U1, int H,
                                              rnntDtype_t dtype, void *stream) {
    if (enc_proj == nullptr || dec_proj == nullptr || hidden == nullptr || act_lens == nullptr || label_lens == nullptr ||
        !shape_ok(B, T, U1, H))
        return RNNT_STATUS_INVALID_VALUE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case RNNT_DTYPE_F32: return launch_fwd<F32>(enc_proj, dec_proj, hidden, B, T, U1, H, st, act_lens, label_lens);
        case RNNT_DTYPE_BF16: return launch_fwd<BF16>(enc_proj, dec_proj, hidden, B, T, U1, H, st, act_lens, label_lens);
        case RNNT_DTYPE_F16: return launch_fwd<F16>(enc_proj, dec_proj, hidden, B, T, U1, H, st, act_lens, label_lens);
        default: return RNNT_STATUS_INVALID_VALUE;
    }
}

rnntStatus_t rnnt_joint_hidden_backward_ragged(const void *hidden, const void *grad_hidden, void *grad_enc_proj,
                                               void *grad_dec_proj, const int *act_lens, const int *label_lens, int B,
                                               int T, int U1, int H, rnntDtype_t dtype, void *workspace, void *stream) {
    if (hidden == nullptr || grad_hidden == nullptr || grad_enc_proj == nullptr || grad_dec_proj == nullptr ||
        act_lens == nullptr || label_lens == nullptr || workspace == nullptr || !shape_ok(B, T, U1, H) ||
        !aligned16(workspace))
        return RNNT_STATUS_INVALID_VALUE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case RNNT_DTYPE_F32:
            return launch_bwd<F32>(hidden, grad_hidden, grad_enc_proj, grad_dec_proj, B, T, U1, H, workspace, st, act_lens,
                                   label_lens);
        case RNNT_DTYPE_BF16:
            return launch_bwd<BF16>(hidden, grad_hidden, grad_enc_proj, grad_dec_proj, B, T, U1, H, workspace, st, act_lens,
                                    label_lens);
        case RNNT_DTYPE_F16:
            return launch_bwd<F16>(hidden, grad_hidden, grad_enc_proj, grad_dec_proj, B, T, U1, H, workspace, st, act_lens,
                                   label_lens);
        default: return RNNT_STATUS_INVALID_VALUE;
    }
}

rnntStatus_t rnnt_joint_hidden_forward_compact(const void *enc_proj, const void *dec_proj, void *hidden,
                                               long long hidden_rows, const int *act_lens, const int *label_lens, int B,
                                               int T, int U1, int H, rnntDtype_t dtype, void *stream) {
    if (enc_proj == nullptr || dec_proj == nullptr || hidden == nullptr || act_lens == nullptr || label_lens == nullptr ||
        hidden_rows <= 0 || !shape_ok(B, T, U1, H))
        return RNNT_STATUS_INVALID_VALUE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case RNNT_DTYPE_F32:
            return launch_fwd<F32>(enc_proj, dec_proj, hidden, B, T, U1, H, st, act_lens, label_lens, hidden_rows);
        case RNNT_DTYPE_BF16:
            return launch_fwd<BF16>(enc_proj, dec_proj, hidden, B, T, U1, H, st, act_lens, label_lens, hidden_rows);
        case RNNT_DTYPE_F16:
            return launch_fwd<F16>(enc_proj, dec_proj, hidden, B, T, U1, H, st, act_lens, label_lens, hidden_rows);
        default: return RNNT_STATUS_INVALID_VALUE;
    }
}

rnntStatus_t rnnt_joint_hidden_backward_compact(const void *hidden, const void *grad_hidden, long long hidden_rows,
                                                void *grad_enc_proj, void *grad_dec_proj, const int *act_lens,
                                                const int *label_lens, int B, int T, int U1, int H, rnntDtype_t dtype,
                                                void *workspace, void *stream) {
    if (hidden == nullptr || grad_hidden == nullptr || grad_enc_proj == nullptr || grad_dec_proj == nullptr ||
        act_lens == nullptr || label_lens == nullptr || workspace == nullptr || hidden_rows <= 0 ||
        !shape_ok(B, T, U1, H) || !aligned16(workspace))
        return RNNT_STATUS_INVALID_VALUE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case RNNT_DTYPE_F32:
            return launch_bwd<F32>(hidden, grad_hidden, grad_enc_proj, grad_dec_proj, B, T, U1, H, workspace, st, act_lens,
                                   label_lens, hidden_rows);
        case RNNT_DTYPE_BF16:
            return launch_bwd<BF16>(hidden, grad_hidden, grad_enc_proj, grad_dec_proj, B, T, U1, H, workspace, st, act_lens,
                                    label_lens, hidden_rows);
        case RNNT_DTYPE_F16:
            return launch_bwd<F16>(hidden, grad_hidden, grad_enc_proj, grad_dec_proj, B, T, U1, H, workspace, st, act_lens,
                                   label_lens, hidden_rows);
        default: return RNNT_STATUS_INVALID_VALUE;
    }
}

rnntStatus_t rnnt_row_argmax(const void *x, int rows, int V, rnntDtype_t dtype, int *out, void *stream) {
    return rnnt_row_max_logp(x, rows, V, dtype, out, nullptr, stream);
}

rnntStatus_t rnnt_row_max_logp(const void *x, int rows, int V, rnntDtype_t dtype, int *out, float *logp, void *stream) {
    if (x == nullptr || out == nullptr || rows < 1 || V < 1) return RNNT_STATUS_INVALID_VALUE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((rows + 3) / 4), block(256);
    // rows on 16 bytes: the one-pass kernel with 16-byte loads; any other layout: element by element
#define ROW_LAUNCH(DT, E)                                                                                          \
    do {                                                                                                          \
        if (aligned16(x) && V % DT::EPC == 0)                                                                     \
            hipLaunchKernelGGL((row_argmax_vec_kernel<DT>), grid, block, 0, st, static_cast<const E *>(x), rows, V, out, \
                               logp);                                                                             \
        else                                                                                                      \
            hipLaunchKernelGGL((row_argmax_kernel<DT>), grid, block, 0, st, static_cast<const E *>(x), rows, V, out, logp); \
    } while (0)
    switch (dtype) {
        case RNNT_DTYPE_F32: ROW_LAUNCH(F32, float); break;
        case RNNT_DTYPE_BF16: ROW_LAUNCH(BF16, __bf16); break;
        case RNNT_DTYPE_F16: ROW_LAUNCH(F16, _Float16); break;
        default: return RNNT_STATUS_INVALID_VALUE;
    }
#undef ROW_LAUNCH
    return hipGetLastError() == hipSuccess ? RNNT_STATUS_SUCCESS : RNNT_STATUS_EXECUTION_FAILED;
}

}  // extern "C"

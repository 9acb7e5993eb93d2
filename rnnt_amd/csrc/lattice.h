// lattice.h — the lattice workspace and node indexing shared by the kernels of librnnt_mi355.so
// (rnnt_mi355.hip: K1/K2/K3; fused_mi355.hip: the output projection fused with K1).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rnnt_mi355.h"

namespace rnnt_mi355_detail {

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
// log of the row's sum of exponentials given its maximum m.  A row whose maximum is +inf has no
// log-softmax (torch.log_softmax: NaN in every element, x - inf = -inf or NaN and the sum is NaN):
// NaN here makes lp_blank / lp_label of the node NaN, hence the utterance's cost, as upstream's CPU path.
// (An all -inf row, m = -inf, gives log(0 + ...) with s = V after the clamp of m to 0: x - 0 = -inf,
// exp = 0, s = 0, log = -inf and lp = -inf - (-inf) = NaN, also as torch.)
__device__ __forceinline__ float row_log_sum(float s, float m) {
    return m == INFINITY ? __builtin_nanf("") : __logf(s);
}
// Skewed lattice layout: one anti-diagonal d = t+u of one utterance is one row of US floats
// (US = U1 rounded up to a multiple of 64, so a wave reads/writes whole aligned 256-byte rows).
__device__ __forceinline__ size_t skew_index(int b, int t, int u, int ND, int US) {
    return ((size_t)b * (ND + 1) + (size_t)(t + u)) * US + u;  // +1: K2's junk row
}

// One utterance's slice of the COMPACTED node order K1 walks: the nodes inside the utterances' lattices only,
// utterance by utterance, frame by frame, label position fastest — the order they have in memory with the
// padding left out.  Written by rnnt_node_spans from the length tensors on the device (the host never reads them).
struct NodeSpan {
    long long first;  // compacted index of the utterance's node (0, 0); entry [B] holds the total
    int Tb;           // frames; 0 for an utterance whose lengths do not fit the tensor (it has no nodes)
    int Ub1;          // label positions per frame, U_b + 1 (1 where Tb = 0, so that it can divide)
};

// The COMPACTED node order of the fused projection + loss kernels (rnnt_joint_loss_*_compact): the MFMA kernels
// walk the nodes inside the lattices only and keep their logits in that order, so a ragged batch costs matrix work
// and memory for its own nodes, not for the padding of the [B, maxT, U1] tensor.  rowmap[k] = row of that tensor
// (b * maxT + t) * U1 + u of compacted node k, written by rnnt_compact_rows on the device; the number of nodes is
// spans[B].first.  rowmap = nullptr: every row of the tensor is a node (the dense entry points).
struct Compact {
    const int *rowmap;
    const NodeSpan *spans;
    int B;
};
__device__ __forceinline__ long long compact_total(const Compact &c, long long cap) {
    if (!c.rowmap) return cap;
    const long long n = c.spans[c.B].first;  // (a scalar load)
    return n < cap ? n : cap;
}
__device__ __forceinline__ long long compact_row(const Compact &c, long long k) { return c.rowmap ? c.rowmap[k] : k; }

struct Lattice {      // device pointers into the workspace
    float2 *ml;       // {max_v x_v, log sum_v exp(x_v - max)} per node, kept apart: lse = m + l would
                      // round to ulp(|m|) and cost K3 that much relative accuracy on offset logits
    float2 *lp;       // {lp_blank, lp_label} per node
    float *alpha;
    float *beta;
    float *ll;        // [B] forward log-likelihood
    NodeSpan *spans;  // [B + 1], see above
    int ND;           // anti-diagonals per utterance = maxT + maxU - 1
    int US;           // row stride of the skewed layout
};


// ---- per-node coefficients of the gradient (K3 of rnnt_mi355.hip and the fused backward of fused_bwd_mi355.hip) ----
//   g_v = scale * ( softmax_v * (Bk + Yk)  - [v=blank] Bk - [v=y_u] Yk )          (SURVEY.md §8a A5 + A6)
struct NodeCoef {
    float sa;    // scale * (Bk + Yk): weight of softmax_v
    float sbk;   // scale * Bk: taken off the blank logit
    float syk;   // scale * Yk: taken off the label logit
    float m, l;  // softmax_v = 2^((x_v - m)*log2e - l*log2e): x - max first (as torch.log_softmax), so a
                 // large common offset of the logits costs nothing
    int y;       // label of this node's emit arc, -1 if none
};

// Every value the coefficients of one node need, loaded unconditionally (nothing here depends on another
// load), then the arithmetic: the same association as the upstream CPU path (alpha + beta_next + lp - ll, and the
// dense softmax term weighted by Bk + Yk rather than by e^{alpha+beta-ll}).
struct NodeIn {
    float llb, a, go, beta_t, beta_u;
    float2 lp, ml;
};
__device__ __forceinline__ NodeIn node_in(const Lattice &L, const RowCoord &c, bool emit,
                                          const float *__restrict__ grad_out, int grad_out_stride) {
    const size_t o = skew_index(c.b, c.t, c.u, L.ND, L.US);
    NodeIn in;
    in.llb = L.ll[c.b];
    in.a = L.alpha[o];
    in.lp = L.lp[o];
    in.ml = L.ml[o];
    in.go = grad_out ? grad_out[(size_t)c.b * grad_out_stride] : 1.0f;
    in.beta_t = L.beta[o + L.US];                 // next diagonal (the last one is followed by K2's junk row)
    in.beta_u = L.beta[o + L.US + (emit ? 1 : 0)];  // column u+1 only where it exists
    return in;
}
__device__ __forceinline__ NodeCoef node_coef_from(const NodeIn &in, int t, int u, int Tb, int Ub, int y,
                                                   float grad_scale) {
    const float scale = grad_scale * in.go;
    float bk = 0.0f, yk = 0.0f;
    if (t < Tb - 1)
        bk = __expf(in.a + in.beta_t + in.lp.x - in.llb);
    else if (u == Ub)
        bk = __expf(in.a + in.lp.x - in.llb);
    if (u < Ub) yk = __expf(in.a + in.beta_u + in.lp.y - in.llb);
    NodeCoef n;
    n.y = u < Ub ? y : -1;
    n.m = in.ml.x;
    n.l = in.ml.y;
    n.sa = scale * (bk + yk);
    n.sbk = scale * bk;
    n.syk = scale * yk;
    return n;
}


// Launches of rnnt_mi355.hip that the fused forward (fused_mi355.hip) runs after its own kernel.
rnntStatus_t launch_lattice_and_costs(const int *act_lens, const int *label_lens, int B, float *costs,
                                      float *loss_out, int reduction, const Lattice &L, const rnntOptions &o);
Lattice carve_workspace(void *ws, int maxT, int maxU, int B);
// spans[b] = {first compacted node, T_b, U_b + 1}, spans[B].first = the number of nodes (rnnt_node_spans)
rnntStatus_t launch_node_spans(const int *act_lens, const int *label_lens, int B, const Lattice &L, const rnntOptions &o);
// rowmap[k] = tensor row of compacted node k (rowmap may be null); nodes from `cap` on get NaN lattice entries
rnntStatus_t launch_compact_rows(int *rowmap, const Lattice &L, int B, long long cap, const rnntOptions &o);

}  // namespace rnnt_mi355_detail

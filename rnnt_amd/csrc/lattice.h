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


// Launches of rnnt_mi355.hip that the fused forward (fused_mi355.hip) runs after its own kernel.
rnntStatus_t launch_lattice_and_costs(const int *act_lens, const int *label_lens, int B, float *costs,
                                      float *loss_out, int reduction, const Lattice &L, const rnntOptions &o);
Lattice carve_workspace(void *ws, int maxT, int maxU, int B);

}  // namespace rnnt_mi355_detail

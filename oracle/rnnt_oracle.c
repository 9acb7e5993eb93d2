/*
 * rnnt_oracle.c — CPU restatement of the RNN-Transducer loss used as the
 * parity ORACLE for rnnt_amd.
 *
 * *** TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and the
 * *** cpu_baseline leg of bench.py may load this.  The product (rnnt_amd/)
 * *** never imports, links or falls back to it.
 *
 * What it restates.  The reference (jhvmhg/rnnt) computes its loss with ONE
 * call into an un-vendored, un-pinned third-party package:
 *     /root/reference/src/rnnt/model.py:5    from warprnnt_pytorch import RNNTLoss
 *     /root/reference/src/rnnt/model.py:87   self.transducer_loss = RNNTLoss()
 *     /root/reference/src/rnnt/model.py:116  loss = self.transducer_loss(logits, tokens.int(), out_len, tok_len.int())
 * (README.md:6,47 name the package: HawkAaron/warp-transducer, no version).
 * Its sources are not in /root/reference nor in this image, so nothing can be
 * compiled into oracle/_ref.  This file restates that package's published CPU
 * algorithm (Graves 2012, "Sequence Transduction with RNNs", eqs. 16-20, in
 * the arrangement SURVEY.md §8a rows A3-A6 records):
 *   A3  log_softmax over V (materialised, as the upstream CPU path does in Python)
 *   A4  alpha / beta recursions per utterance, ll = alpha(T-1,U-1) + lp_blank(T-1,U-1)
 *   A5  sparse gradient w.r.t. log-probs, densified by the log_softmax backward
 *   A6  per-utterance cost = -ll; reductions are applied by the Python wrapper
 * with an OpenMP `parallel for` over the minibatch (one utterance per thread).
 *
 * PARITY UNPINNED: the reference holds no tests, golden vectors or fixtures
 * for this path.  The anchors are two known-answer vectors
 * recalled from the upstream package's public unit tests (B1·T2·U2·V5, cost
 * 4.4956668; B2·T4·U2·V3, costs 4.2806528590890736 / 3.9384369822503591), which
 * tests/test_oracle.py checks together with a float64 dynamic programme,
 * autograd through it, and brute-force path enumeration.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define REAL float
#define SUFFIX _f32
#define EXP expf
#define LOG logf
#define LOG1P log1pf
#include "rnnt_oracle_impl.h"
#undef REAL
#undef SUFFIX
#undef EXP
#undef LOG
#undef LOG1P

#define REAL double
#define SUFFIX _f64
#define EXP exp
#define LOG log
#define LOG1P log1p
#include "rnnt_oracle_impl.h"
#undef REAL
#undef SUFFIX
#undef EXP
#undef LOG
#undef LOG1P

int rnnt_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/*
 * acts   [B, maxT, maxU, V] contiguous raw logits (maxU = max label length + 1)
 * labels [B, maxU-1] zero padded; act_lens/label_lens [B]
 * costs  [B]  <- -log P(y_b | x_b)
 * grads  [B, maxT, maxU, V] or NULL  <- d costs[b] / d acts[b]; zeros in padding
 * lls_backward [B] or NULL <- beta(0,0), a self-check value
 * Returns 0, or -1 on bad arguments / allocation failure.
 */
#define DEFINE_BATCH(REAL, SUFFIX)                                                                   \
    int rnnt_oracle##SUFFIX(const REAL *acts, const int *labels, const int *act_lens,                \
                            const int *label_lens, int B, int maxT, int maxU, int V, int blank,      \
                            REAL *costs, REAL *grads, REAL *lls_backward, int num_threads) {         \
        if (!acts || !labels || !act_lens || !label_lens || !costs) return -1;                       \
        if (B <= 0 || maxT <= 0 || maxU <= 0 || V <= 0 || blank < 0 || blank >= V) return -1;        \
        for (int b = 0; b < B; ++b) {                                                                \
            if (act_lens[b] < 1 || act_lens[b] > maxT) return -1;                                    \
            if (label_lens[b] < 0 || label_lens[b] + 1 > maxU) return -1;                            \
            for (int u = 0; u < label_lens[b]; ++u) {                                                \
                const int y = labels[(size_t)b * (size_t)(maxU - 1) + u];                            \
                if (y < 0 || y >= V) return -1;                                                      \
            }                                                                                        \
        }                                                                                            \
        const size_t per_utt = (size_t)maxT * (size_t)maxU * (size_t)V;                              \
        if (grads) memset(grads, 0, per_utt * (size_t)B * sizeof(REAL));                             \
        int fail = 0;                                                                                \
        if (num_threads <= 0) num_threads = rnnt_oracle_max_threads();                               \
        _Pragma("omp parallel for schedule(dynamic, 1) num_threads(num_threads)")                    \
        for (int b = 0; b < B; ++b) {                                                                \
            const int T = act_lens[b], U = label_lens[b] + 1;                                        \
            REAL *lp = (REAL *)malloc((size_t)T * U * (size_t)V * sizeof(REAL));                     \
            REAL *ab = (REAL *)malloc((size_t)2 * T * U * sizeof(REAL));                             \
            if (!lp || !ab) {                                                                        \
                fail = 1;                                                                            \
            } else {                                                                                 \
                REAL llb = 0;                                                                        \
                const REAL ll = rnnt_one_utt##SUFFIX(                                                \
                    acts + per_utt * (size_t)b, labels + (size_t)b * (size_t)(maxU - 1), T, U, maxU, \
                    V, blank, lp, ab, ab + (size_t)T * U, grads ? grads + per_utt * (size_t)b : NULL,\
                    &llb);                                                                           \
                costs[b] = -ll;                                                                      \
                if (lls_backward) lls_backward[b] = llb;                                             \
            }                                                                                        \
            free(lp);                                                                                \
            free(ab);                                                                                \
        }                                                                                            \
        return fail ? -1 : 0;                                                                        \
    }

DEFINE_BATCH(float, _f32)
DEFINE_BATCH(double, _f64)

/*
 * The native half of the upstream-shaped CPU path (bench.py's second CPU baseline): log_probs
 * [B, maxT, maxU, V] as torch.log_softmax left them; costs [B]; grads_lp [B, maxT, maxU, V] or NULL <- the
 * sparse d costs[b] / d log_probs[b] (zeros elsewhere).  Same OpenMP loop over the minibatch.
 */
#define DEFINE_BATCH_LOGPROBS(REAL, SUFFIX)                                                          \
    int rnnt_oracle_logprobs##SUFFIX(const REAL *lp, const int *labels, const int *act_lens,         \
                                     const int *label_lens, int B, int maxT, int maxU, int V,        \
                                     int blank, REAL *costs, REAL *grads_lp, int num_threads) {      \
        if (!lp || !labels || !act_lens || !label_lens || !costs) return -1;                         \
        if (B <= 0 || maxT <= 0 || maxU <= 0 || V <= 0 || blank < 0 || blank >= V) return -1;        \
        for (int b = 0; b < B; ++b) {                                                                \
            if (act_lens[b] < 1 || act_lens[b] > maxT) return -1;                                    \
            if (label_lens[b] < 0 || label_lens[b] + 1 > maxU) return -1;                            \
            for (int u = 0; u < label_lens[b]; ++u) {                                                \
                const int y = labels[(size_t)b * (size_t)(maxU - 1) + u];                            \
                if (y < 0 || y >= V) return -1;                                                      \
            }                                                                                        \
        }                                                                                            \
        const size_t per_utt = (size_t)maxT * (size_t)maxU * (size_t)V;                              \
        if (grads_lp) memset(grads_lp, 0, per_utt * (size_t)B * sizeof(REAL));                       \
        int fail = 0;                                                                                \
        if (num_threads <= 0) num_threads = rnnt_oracle_max_threads();                               \
        _Pragma("omp parallel for schedule(dynamic, 1) num_threads(num_threads)")                    \
        for (int b = 0; b < B; ++b) {                                                                \
            const int T = act_lens[b], U = label_lens[b] + 1;                                        \
            REAL *ab = (REAL *)malloc((size_t)2 * T * U * sizeof(REAL));                             \
            if (!ab) {                                                                               \
                fail = 1;                                                                            \
            } else {                                                                                 \
                costs[b] = -rnnt_one_utt_logprobs##SUFFIX(                                           \
                    lp + per_utt * (size_t)b, labels + (size_t)b * (size_t)(maxU - 1), T, U, maxU,   \
                    V, blank, ab, ab + (size_t)T * U, grads_lp ? grads_lp + per_utt * (size_t)b : NULL); \
            }                                                                                        \
            free(ab);                                                                                \
        }                                                                                            \
        return fail ? -1 : 0;                                                                        \
    }

DEFINE_BATCH_LOGPROBS(float, _f32)
DEFINE_BATCH_LOGPROBS(double, _f64)

/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See rnnt_oracle.c for the header.
 *
 * Included twice by rnnt_oracle.c with REAL = float / double.  One utterance
 * per call; rnnt_oracle.c wraps it in the OpenMP loop over the minibatch.
 *
 * Lattice conventions (SURVEY.md §8a rows A3–A5; Graves 2012):
 *   T  = act_lens[b]            number of encoder frames of this utterance
 *   U  = label_lens[b] + 1      lattice width (labels + the "no label yet" column)
 *   acts row (t,u) has V logits, row stride V, u stride V, t stride maxU*V
 */

#ifndef REAL
#error "define REAL, SUFFIX, EXP, LOG, LOG1P before including"
#endif

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

/* log(e^a + e^b) the way the upstream CPU helper does it: explicit -inf
 * short-circuits, then log1p(exp(small - big)) + big. */
static inline REAL FN(log_sum_exp)(REAL a, REAL b) {
    if (a == -(REAL)INFINITY) return b;
    if (b == -(REAL)INFINITY) return a;
    if (a > b) return LOG1P(EXP(b - a)) + a;
    return LOG1P(EXP(a - b)) + b;
}

/*
 * A4: alpha / beta over one utterance's T x U lattice of materialised log-probs; row (t,u) starts at
 * lp + t*t_stride + u*V.  Returns ll = alpha(T-1,U-1) + lp_blank(T-1,U-1); *ll_backward_out = beta(0,0).
 */
static REAL FN(rnnt_lattice)(const REAL *lp, size_t t_stride, const int *labels, int T, int U, int V, int blank,
                             REAL *alphas, REAL *betas, REAL *ll_backward_out) {
#define LP(t, u) (lp + (size_t)(t) * t_stride + (size_t)(u) * (size_t)V)
#define IDX(t, u) ((size_t)(t) * (size_t)U + (size_t)(u))
    /* A4: alphas */
    alphas[0] = 0;
    for (int t = 0; t < T; ++t) {
        for (int u = 0; u < U; ++u) {
            if (u == 0 && t > 0) alphas[IDX(t, 0)] = alphas[IDX(t - 1, 0)] + LP(t - 1, 0)[blank];
            if (t == 0 && u > 0) alphas[IDX(0, u)] = alphas[IDX(0, u - 1)] + LP(0, u - 1)[labels[u - 1]];
            if (t > 0 && u > 0) {
                const REAL no_emit = alphas[IDX(t - 1, u)] + LP(t - 1, u)[blank];
                const REAL emit = alphas[IDX(t, u - 1)] + LP(t, u - 1)[labels[u - 1]];
                alphas[IDX(t, u)] = FN(log_sum_exp)(emit, no_emit);
            }
        }
    }
    const REAL ll = alphas[IDX(T - 1, U - 1)] + LP(T - 1, U - 1)[blank];

    /* A4: betas */
    betas[IDX(T - 1, U - 1)] = LP(T - 1, U - 1)[blank];
    for (int t = T - 1; t >= 0; --t) {
        for (int u = U - 1; u >= 0; --u) {
            if (u == U - 1 && t < T - 1) betas[IDX(t, U - 1)] = betas[IDX(t + 1, U - 1)] + LP(t, U - 1)[blank];
            if (t == T - 1 && u < U - 1) betas[IDX(T - 1, u)] = betas[IDX(T - 1, u + 1)] + LP(T - 1, u)[labels[u]];
            if (t < T - 1 && u < U - 1) {
                const REAL no_emit = betas[IDX(t + 1, u)] + LP(t, u)[blank];
                const REAL emit = betas[IDX(t, u + 1)] + LP(t, u)[labels[u]];
                betas[IDX(t, u)] = FN(log_sum_exp)(emit, no_emit);
            }
        }
    }
    if (ll_backward_out) *ll_backward_out = betas[0];
    return ll;
#undef LP
#undef IDX
}

/*
 * One utterance.  lp is a scratch buffer of T*U*V log-probs (the tensor the
 * upstream CPU path materialises with torch.log_softmax before the native
 * call); alphas/betas are T*U scratch.  grads may be NULL (forward only).
 * Returns the forward log-likelihood.
 */
static REAL FN(rnnt_one_utt)(const REAL *acts, const int *labels, int T, int U, int maxU, int V,
                             int blank, REAL *lp, REAL *alphas, REAL *betas, REAL *grads,
                             REAL *ll_backward_out) {
    const size_t t_stride = (size_t)maxU * (size_t)V;
#define ROW(t, u) (acts + (size_t)(t) * t_stride + (size_t)(u) * (size_t)V)
#define LP(t, u) (lp + ((size_t)(t) * (size_t)U + (size_t)(u)) * (size_t)V)
#define IDX(t, u) ((size_t)(t) * (size_t)U + (size_t)(u))

    /* A3: log_softmax over V, computed as torch does it: (x - max) - log(sum exp(x - max)) */
    for (int t = 0; t < T; ++t) {
        for (int u = 0; u < U; ++u) {
            const REAL *x = ROW(t, u);
            REAL *o = LP(t, u);
            REAL mx = x[0];
            for (int v = 1; v < V; ++v) mx = x[v] > mx ? x[v] : mx;
            REAL s = 0;
            for (int v = 0; v < V; ++v) s += EXP(x[v] - mx);
            const REAL ls = LOG(s);
            for (int v = 0; v < V; ++v) o[v] = (x[v] - mx) - ls;
        }
    }

    /* A4: alphas, betas (shared with the log-probs entry point below) */
    const REAL ll = FN(rnnt_lattice)(lp, (size_t)U * (size_t)V, labels, T, U, V, blank, alphas, betas,
                                     ll_backward_out);

    if (grads) {
        /* A5: sparse d(-ll)/d(lp) (two entries per node), then densified through
         * the log_softmax backward  g_x[v] = g_lp[v] - softmax[v] * sum_v' g_lp[v'],
         * which is what autograd applies on the upstream CPU path. */
        const size_t gt_stride = t_stride;
        for (int t = 0; t < T; ++t) {
            for (int u = 0; u < U; ++u) {
                const REAL *o = LP(t, u);
                REAL *g = grads + (size_t)t * gt_stride + (size_t)u * (size_t)V;
                REAL g_blank = 0, g_label = 0;
                if (t < T - 1) g_blank = -EXP(alphas[IDX(t, u)] + betas[IDX(t + 1, u)] + o[blank] - ll);
                if (t == T - 1 && u == U - 1) g_blank = -EXP(alphas[IDX(t, u)] + o[blank] - ll);
                if (u < U - 1) g_label = -EXP(alphas[IDX(t, u)] + betas[IDX(t, u + 1)] + o[labels[u]] - ll);
                const REAL gsum = g_blank + g_label;
                for (int v = 0; v < V; ++v) g[v] = -EXP(o[v]) * gsum;
                g[blank] += g_blank;
                /* when labels[u] == blank both sparse entries land on the same logit */
                if (u < U - 1) g[labels[u]] += g_label;
            }
        }
    }
    return ll;
#undef ROW
#undef LP
#undef IDX
}

/*
 * One utterance of the upstream-SHAPED CPU path: the caller (upstream's Python glue) has already run
 * torch.log_softmax; the native call walks the lattice on those log-probs and leaves the SPARSE gradient
 * d(-ll)/d(lp) — two entries per node, the rest of the (pre-zeroed) tensor untouched — for autograd to carry
 * back through the log_softmax.  lp and grads_lp are [maxT, maxU, V] slices (row stride maxU*V).
 */
static REAL FN(rnnt_one_utt_logprobs)(const REAL *lp, const int *labels, int T, int U, int maxU, int V, int blank,
                                      REAL *alphas, REAL *betas, REAL *grads_lp) {
    const size_t t_stride = (size_t)maxU * (size_t)V;
#define LP(t, u) (lp + (size_t)(t) * t_stride + (size_t)(u) * (size_t)V)
#define IDX(t, u) ((size_t)(t) * (size_t)U + (size_t)(u))
    const REAL ll = FN(rnnt_lattice)(lp, t_stride, labels, T, U, V, blank, alphas, betas, NULL);
    if (grads_lp) {
        for (int t = 0; t < T; ++t) {
            for (int u = 0; u < U; ++u) {
                const REAL *o = LP(t, u);
                REAL *g = grads_lp + (size_t)t * t_stride + (size_t)u * (size_t)V;
                REAL g_blank = 0, g_label = 0;
                if (t < T - 1) g_blank = -EXP(alphas[IDX(t, u)] + betas[IDX(t + 1, u)] + o[blank] - ll);
                if (t == T - 1 && u == U - 1) g_blank = -EXP(alphas[IDX(t, u)] + o[blank] - ll);
                if (u < U - 1) g_label = -EXP(alphas[IDX(t, u)] + betas[IDX(t, u + 1)] + o[labels[u]] - ll);
                g[blank] += g_blank;
                if (u < U - 1) g[labels[u]] += g_label;
            }
        }
    }
    return ll;
#undef LP
#undef IDX
}

#undef FN
#undef CAT
#undef CAT_

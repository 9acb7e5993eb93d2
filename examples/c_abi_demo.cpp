// c_abi_demo.cpp — calls librnnt_mi355.so through include/rnnt_mi355.h with no PyTorch in sight:
// hipMalloc'ed buffers, the upstream-shaped compute_rnnt_loss(), the split forward/backward and the loss on a ragged
// batch's lattice nodes only (rnnt_compact_rows / rnnt_forward_compact / rnnt_backward_compact).
// Input is the B1·T2·U2·V5 known-answer vector; prints costs and gradients
// so tests/test_gpu_c_abi.py can compare them.  Build: see __graft_entry__.build().
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/rnnt_mi355.h"

#define HIP_OK(x)                                                                 \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            std::fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            return 2;                                                             \
        }                                                                         \
    } while (0)
#define RNNT_OK(x)                                                                \
    do {                                                                          \
        rnntStatus_t s_ = (x);                                                    \
        if (s_ != RNNT_STATUS_SUCCESS) {                                          \
            std::fprintf(stderr, "rnnt error '%s' at line %d\n", rnntGetStatusString(s_), __LINE__); \
            return 3;                                                             \
        }                                                                         \
    } while (0)

int main() {
    const int B = 1, T = 2, U1 = 3, V = 5;
    const std::vector<float> acts = {0.1f, 0.6f, 0.1f, 0.1f, 0.1f, 0.1f, 0.1f, 0.6f, 0.1f, 0.1f,
                                     0.1f, 0.1f, 0.2f, 0.8f, 0.1f, 0.1f, 0.6f, 0.1f, 0.1f, 0.1f,
                                     0.1f, 0.1f, 0.2f, 0.1f, 0.1f, 0.7f, 0.1f, 0.2f, 0.1f, 0.1f};
    const int labels[2] = {1, 2}, act_lens[1] = {2}, label_lens[1] = {2};
    const size_t n = acts.size();

    float *d_acts, *d_grads, *d_costs, *d_loss, *d_go;
    int *d_labels, *d_al, *d_ll;
    void *d_ws;
    size_t ws_bytes = 0;
    RNNT_OK(get_workspace_size(T, U1, B, /*gpu=*/1, &ws_bytes));
    HIP_OK(hipMalloc(&d_acts, n * sizeof(float)));
    HIP_OK(hipMalloc(&d_grads, n * sizeof(float)));
    HIP_OK(hipMalloc(&d_costs, B * sizeof(float)));
    HIP_OK(hipMalloc(&d_loss, sizeof(float)));
    HIP_OK(hipMalloc(&d_go, sizeof(float)));
    HIP_OK(hipMalloc(&d_labels, sizeof(labels)));
    HIP_OK(hipMalloc(&d_al, sizeof(act_lens)));
    HIP_OK(hipMalloc(&d_ll, sizeof(label_lens)));
    HIP_OK(hipMalloc(&d_ws, ws_bytes));
    HIP_OK(hipMemcpy(d_acts, acts.data(), n * sizeof(float), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_labels, labels, sizeof(labels), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_al, act_lens, sizeof(act_lens), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_ll, label_lens, sizeof(label_lens), hipMemcpyHostToDevice));

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    rnntOptions opt = {};
    opt.stream = stream;
    opt.blank_label = 0;
    opt.maxT = T;
    opt.maxU = U1;
    opt.dtype = RNNT_DTYPE_F32;
    opt.stage_events = nullptr;

    // 1. the upstream-shaped call: costs + unscaled gradients
    RNNT_OK(compute_rnnt_loss(d_acts, d_grads, d_labels, d_ll, d_al, V, B, d_costs, d_ws, opt));
    HIP_OK(hipStreamSynchronize(stream));
    float cost = 0;
    std::vector<float> grads(n);
    HIP_OK(hipMemcpy(&cost, d_costs, sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(grads.data(), d_grads, n * sizeof(float), hipMemcpyDeviceToHost));
    std::printf("cost %.9g\n", cost);
    std::printf("grads");
    for (float g : grads) std::printf(" %.9g", g);
    std::printf("\n");

    // 2. the split the autograd binding uses: forward (mean), backward with grad_out = 0.5 on the device
    const float half = 0.5f;
    HIP_OK(hipMemcpy(d_go, &half, sizeof(float), hipMemcpyHostToDevice));
    RNNT_OK(rnnt_forward(d_acts, d_labels, d_al, d_ll, V, B, d_costs, d_loss, RNNT_REDUCTION_MEAN, d_ws, opt));
    RNNT_OK(rnnt_backward(d_acts, d_grads, d_labels, d_al, d_ll, V, B, d_go, 0, 1.0f / B, d_ws, opt));
    HIP_OK(hipStreamSynchronize(stream));
    float loss = 0;
    HIP_OK(hipMemcpy(&loss, d_loss, sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(grads.data(), d_grads, n * sizeof(float), hipMemcpyDeviceToHost));
    std::printf("loss_mean %.9g\n", loss);
    std::printf("grads_half");
    for (float g : grads) std::printf(" %.9g", g);
    std::printf("\n");

    // 3. a ragged batch in the compacted node order: utterance 0 is the vector above, utterance 1 uses one frame and one
    //    label of a [2, 2, 3, 5] tensor.  rnnt_compact_rows numbers the 6 + 2 nodes inside the lattices on the device; the
    //    logits the loss sees hold one row per NODE (a caller that projects only those rows never builds the padding).
    {
        const int B2 = 2, nodes = 2 * 3 + 1 * 2;
        std::vector<float> padded(2 * n, 0.25f);
        for (size_t i = 0; i < n; ++i) padded[i] = acts[i];
        for (size_t i = n; i < 2 * n; ++i) padded[i] = 0.05f * (float)((i * 7) % 11);
        const int labels2[4] = {1, 2, 3, 0}, al2[2] = {2, 1}, ll2[2] = {2, 1};
        float *d_nodes, *d_ngrads, *d_costs2, *d_loss2, *d_go2;
        int *d_labels2, *d_al2, *d_ll2, *d_rowmap;
        void *d_ws2, *d_scratch;
        size_t ws2 = 0;
        RNNT_OK(get_workspace_size(T, U1, B2, 1, &ws2));
        HIP_OK(hipMalloc(&d_nodes, nodes * V * sizeof(float)));
        HIP_OK(hipMalloc(&d_ngrads, nodes * V * sizeof(float)));
        HIP_OK(hipMalloc(&d_costs2, B2 * sizeof(float)));
        HIP_OK(hipMalloc(&d_loss2, B2 * sizeof(float)));
        HIP_OK(hipMalloc(&d_go2, B2 * sizeof(float)));
        HIP_OK(hipMalloc(&d_labels2, sizeof(labels2)));
        HIP_OK(hipMalloc(&d_al2, sizeof(al2)));
        HIP_OK(hipMalloc(&d_ll2, sizeof(ll2)));
        HIP_OK(hipMalloc(&d_rowmap, nodes * sizeof(int)));
        HIP_OK(hipMalloc(&d_ws2, ws2));
        HIP_OK(hipMalloc(&d_scratch, 16 * (B2 + 1)));
        HIP_OK(hipMemcpy(d_labels2, labels2, sizeof(labels2), hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(d_al2, al2, sizeof(al2), hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(d_ll2, ll2, sizeof(ll2), hipMemcpyHostToDevice));
        RNNT_OK(rnnt_compact_rows(d_al2, d_ll2, B2, T, U1, nodes, d_rowmap, d_scratch, stream));
        HIP_OK(hipStreamSynchronize(stream));
        std::vector<int> rowmap(nodes);
        HIP_OK(hipMemcpy(rowmap.data(), d_rowmap, nodes * sizeof(int), hipMemcpyDeviceToHost));
        std::vector<float> rows((size_t)nodes * V);
        std::printf("compact_rowmap");
        for (int k = 0; k < nodes; ++k) {
            std::printf(" %d", rowmap[k]);
            for (int v = 0; v < V; ++v) rows[(size_t)k * V + v] = padded[(size_t)rowmap[k] * V + v];
        }
        std::printf("\n");
        HIP_OK(hipMemcpy(d_nodes, rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice));
        const float ones[2] = {1.0f, 1.0f};
        HIP_OK(hipMemcpy(d_go2, ones, sizeof(ones), hipMemcpyHostToDevice));
        rnntOptions opt2 = opt;
        RNNT_OK(rnnt_forward_compact(d_nodes, nodes, d_labels2, d_al2, d_ll2, V, B2, d_costs2, d_loss2, RNNT_REDUCTION_NONE,
                                     d_ws2, opt2));
        RNNT_OK(rnnt_backward_compact(d_nodes, d_ngrads, nodes, d_labels2, d_al2, d_ll2, V, B2, d_go2, 1, 1.0f, d_ws2, opt2));
        HIP_OK(hipStreamSynchronize(stream));
        float costs2[2];
        std::vector<float> ngrads((size_t)nodes * V);
        HIP_OK(hipMemcpy(costs2, d_costs2, sizeof(costs2), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(ngrads.data(), d_ngrads, ngrads.size() * sizeof(float), hipMemcpyDeviceToHost));
        std::printf("compact_costs %.9g %.9g\n", costs2[0], costs2[1]);
        std::printf("compact_grads");
        for (float g : ngrads) std::printf(" %.9g", g);
        std::printf("\n");
    }

    // 4. errors are status codes
    rnntOptions bad = opt;
    bad.blank_label = V;  // out of range
    std::printf("bad_blank_status %d (%s)\n",
                (int)rnnt_forward(d_acts, d_labels, d_al, d_ll, V, B, d_costs, d_loss, RNNT_REDUCTION_MEAN, d_ws, bad),
                rnntGetStatusString(RNNT_STATUS_INVALID_VALUE));
    return 0;
}

// membench2.hip — which store/load cache policy and access shape stream fastest on MI355X.
// Developer tool. Row-per-wave kernels shaped like K1/K3 (rows of 1058 16-byte chunks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
// AUXL/AUXS: bit0 sc0, bit1 nt, bit4 sc1
template <int NCH, int AUXL, int AUXS, bool DOLOAD, bool DOSTORE>
__global__ __launch_bounds__(512) void k_row(const u32x4* __restrict__ in, u32x4* __restrict__ out, long long rows, int rowchunks, unsigned* sink) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    long long w0 = (long long)blockIdx.x * wpb + (threadIdx.x >> 6), nw = (long long)gridDim.x * wpb;
    unsigned acc = 0;
    for (long long r = w0; r < rows; r += nw) {
        const u32x4* src = in + r * rowchunks; u32x4* dst = out + r * rowchunks;
        const unsigned rb = __builtin_amdgcn_readfirstlane(rowchunks * 16);
        auto rs = mk((const void*)__builtin_amdgcn_readfirstlane((unsigned long long)0) , 0); (void)rs;
        u32x4 v[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) v[k] = u32x4{1u, 2u, 3u, (unsigned)lane};
        if (DOLOAD) {
            __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, rb, 0x00020000);
#pragma unroll
            for (int k = 0; k < NCH; ++k) { int j = lane + 64 * k; v[k] = __builtin_amdgcn_raw_buffer_load_b128(rl, j * 16, 0, AUXL); }
        }
        if (DOSTORE) {
            __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, rb, 0x00020000);
#pragma unroll
            for (int k = 0; k < NCH; ++k) { int j = lane + 64 * k; v[k].x += 1; __builtin_amdgcn_raw_buffer_store_b128(v[k], rsd, j * 16, 0, AUXS); }
        } else {
#pragma unroll
            for (int k = 0; k < NCH; ++k) acc ^= v[k].x ^ v[k].w;
        }
    }
    if (!DOSTORE && acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int rowchunks = 1058; long long rows = 524800;
    size_t n = (size_t)rows * rowchunks;
    u32x4 *a, *b; unsigned* o;
    CK(hipMalloc(&a, n * 16 + 4096)); CK(hipMalloc(&b, n * 16 + 4096)); CK(hipMalloc(&o, 64));
    CK(hipMemset(a, 1, n * 16)); CK(hipMemset(b, 2, n * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double gb, auto launch) {
        for (int i = 0; i < 2; ++i) launch();
        CK(hipDeviceSynchronize());
        float tot = 0; const int reps = 8;
        for (int i = 0; i < reps; ++i) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms; }
        printf("%-40s avg %.3f ms  %.0f GB/s\n", name, tot / reps, gb / (tot / reps) * 1e3 / 1e9); fflush(stdout);
    };
    double gb = (double)n * 16;
#define RUN(NAME, AUXL, AUXS, L, S, BLOCK) { int wpb = BLOCK / 64; int g = (int)((rows + wpb - 1) / wpb); char nm[128]; snprintf(nm, 128, "%s auxL=%d auxS=%d block=%d", NAME, AUXL, AUXS, BLOCK); \
    timeit(nm, gb * ((L ? 1 : 0) + (S ? 1 : 0)), [&] { hipLaunchKernelGGL((k_row<17, AUXL, AUXS, L, S>), dim3(g), dim3(BLOCK), 0, 0, a, b, rows, rowchunks, o); }); }
    RUN("write", 0, 0, false, true, 256) RUN("write", 0, 1, false, true, 256) RUN("write", 0, 2, false, true, 256) RUN("write", 0, 3, false, true, 256)
    RUN("write", 0, 16, false, true, 256) RUN("write", 0, 17, false, true, 256) RUN("write", 0, 18, false, true, 256) RUN("write", 0, 19, false, true, 256)
    RUN("write", 0, 0, false, true, 512) RUN("write", 0, 2, false, true, 512) RUN("write", 0, 18, false, true, 512) RUN("write", 0, 3, false, true, 512)
    RUN("read", 0, 0, true, false, 256) RUN("read", 1, 0, true, false, 256) RUN("read", 2, 0, true, false, 256) RUN("read", 3, 0, true, false, 256)
    RUN("read", 16, 0, true, false, 256) RUN("read", 17, 0, true, false, 256) RUN("read", 18, 0, true, false, 256) RUN("read", 19, 0, true, false, 256)
    RUN("copy", 2, 2, true, true, 512) RUN("copy", 0, 2, true, true, 512) RUN("copy", 2, 0, true, true, 512) RUN("copy", 0, 0, true, true, 512)
    RUN("copy", 2, 18, true, true, 512) RUN("copy", 2, 3, true, true, 512) RUN("copy", 18, 2, true, true, 512) RUN("copy", 3, 2, true, true, 512) RUN("copy", 2, 16, true, true, 512) RUN("copy", 2, 1, true, true, 512)
    RUN("copy", 2, 2, true, true, 256)
    return 0;
}

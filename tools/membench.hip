// membench.hip — streaming ceilings on this MI355X: read-only, write-only, copy; plain vs nontemporal.
// Developer tool (not part of the product): hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <bool NT, int UNR>
__global__ __launch_bounds__(256) void k_read(const u32x4* __restrict__ in, unsigned* __restrict__ out, size_t n) {
    size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    unsigned acc = 0;
    for (size_t i = tid; i < n; i += stride * UNR) {
        u32x4 v[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) { size_t j = i + k * stride; if (j < n) v[k] = NT ? __builtin_nontemporal_load(in + j) : in[j]; else v[k] = u32x4{0,0,0,0}; }
#pragma unroll
        for (int k = 0; k < UNR; ++k) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <bool NT, int UNR>
__global__ __launch_bounds__(256) void k_write(u32x4* __restrict__ out, size_t n) {
    size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    u32x4 z = {1u, 2u, 3u, (unsigned)tid};
    for (size_t i = tid; i < n; i += stride * UNR) {
#pragma unroll
        for (int k = 0; k < UNR; ++k) { size_t j = i + k * stride; if (j < n) { if (NT) __builtin_nontemporal_store(z, out + j); else out[j] = z; } }
    }
}
template <bool NTL, bool NTS, int UNR>
__global__ __launch_bounds__(256) void k_copy(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n) {
    size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (size_t i = tid; i < n; i += stride * UNR) {
        u32x4 v[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) { size_t j = i + k * stride; if (j < n) v[k] = NTL ? __builtin_nontemporal_load(in + j) : in[j]; }
#pragma unroll
        for (int k = 0; k < UNR; ++k) { size_t j = i + k * stride; if (j < n) { v[k].x += 1; if (NTS) __builtin_nontemporal_store(v[k], out + j); else out[j] = v[k]; } }
    }
}
// row-per-wave copy like K3: rows of `rowchunks` 16-byte chunks, one wave per row, grid-stride
template <bool NTL, bool NTS, int NCH>
__global__ __launch_bounds__(256) void k_rowcopy(const u32x4* __restrict__ in, u32x4* __restrict__ out, long long rows, int rowchunks) {
    const int lane = threadIdx.x & 63;
    long long w0 = (long long)blockIdx.x * 4 + threadIdx.x / 64, nw = (long long)gridDim.x * 4;
    for (long long r = w0; r < rows; r += nw) {
        const u32x4* src = in + r * rowchunks; u32x4* dst = out + r * rowchunks;
        u32x4 v[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) { int j = lane + 64 * k; if (j < rowchunks) v[k] = NTL ? __builtin_nontemporal_load(src + j) : src[j]; }
#pragma unroll
        for (int k = 0; k < NCH; ++k) { int j = lane + 64 * k; if (j < rowchunks) { v[k].x += 1; if (NTS) __builtin_nontemporal_store(v[k], dst + j); else dst[j] = v[k]; } }
    }
}

int main(int argc, char** argv) {
    size_t bytes = (size_t)8884 * 1000 * 1000 / 16 * 16;   // ~ config-2 logits
    if (argc > 1) bytes = (size_t)atoll(argv[1]);
    size_t n = bytes / 16;
    u32x4 *a, *b; unsigned* o;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&o, 64));
    CK(hipMemset(a, 1, n * 16)); CK(hipMemset(b, 2, n * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double gb, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        float best = 1e9, tot = 0; const int reps = 10;
        for (int i = 0; i < reps; ++i) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; tot += ms; }
        printf("%-44s avg %.3f ms  %.0f GB/s   best %.3f ms %.0f GB/s\n", name, tot / reps, gb / (tot / reps) * 1e3 / 1e9 * 1e0, best, gb / best * 1e3 / 1e9);
        fflush(stdout);
    };
    double gb = (double)n * 16;
    const int grids[] = {2048, 4096, 8192};
    for (int g : grids) {
        printf("--- grid %d\n", g);
        timeit("read plain unr4", gb, [&] { hipLaunchKernelGGL((k_read<false, 4>), dim3(g), dim3(256), 0, 0, a, o, n); });
        timeit("read nt unr4", gb, [&] { hipLaunchKernelGGL((k_read<true, 4>), dim3(g), dim3(256), 0, 0, a, o, n); });
        timeit("read nt unr8", gb, [&] { hipLaunchKernelGGL((k_read<true, 8>), dim3(g), dim3(256), 0, 0, a, o, n); });
        timeit("write plain unr4", gb, [&] { hipLaunchKernelGGL((k_write<false, 4>), dim3(g), dim3(256), 0, 0, b, n); });
        timeit("write nt unr4", gb, [&] { hipLaunchKernelGGL((k_write<true, 4>), dim3(g), dim3(256), 0, 0, b, n); });
        timeit("copy plain/plain unr4", 2 * gb, [&] { hipLaunchKernelGGL((k_copy<false, false, 4>), dim3(g), dim3(256), 0, 0, a, b, n); });
        timeit("copy nt/nt unr4", 2 * gb, [&] { hipLaunchKernelGGL((k_copy<true, true, 4>), dim3(g), dim3(256), 0, 0, a, b, n); });
        timeit("copy nt/plain unr4", 2 * gb, [&] { hipLaunchKernelGGL((k_copy<true, false, 4>), dim3(g), dim3(256), 0, 0, a, b, n); });
        timeit("copy plain/nt unr4", 2 * gb, [&] { hipLaunchKernelGGL((k_copy<false, true, 4>), dim3(g), dim3(256), 0, 0, a, b, n); });
        timeit("copy nt/nt unr8", 2 * gb, [&] { hipLaunchKernelGGL((k_copy<true, true, 8>), dim3(g), dim3(256), 0, 0, a, b, n); });
    }
    long long rows = (long long)(n / 1058);
    for (int g : grids) {
        printf("--- rowcopy grid %d rows %lld x 1058 chunks\n", g, rows);
        timeit("rowcopy nt/nt 17", 2.0 * rows * 1058 * 16, [&] { hipLaunchKernelGGL((k_rowcopy<true, true, 17>), dim3(g), dim3(256), 0, 0, a, b, rows, 1058); });
        timeit("rowcopy plain/plain 17", 2.0 * rows * 1058 * 16, [&] { hipLaunchKernelGGL((k_rowcopy<false, false, 17>), dim3(g), dim3(256), 0, 0, a, b, rows, 1058); });
        timeit("rowcopy nt/plain 17", 2.0 * rows * 1058 * 16, [&] { hipLaunchKernelGGL((k_rowcopy<true, false, 17>), dim3(g), dim3(256), 0, 0, a, b, rows, 1058); });
        timeit("rowcopy plain/nt 17", 2.0 * rows * 1058 * 16, [&] { hipLaunchKernelGGL((k_rowcopy<false, true, 17>), dim3(g), dim3(256), 0, 0, a, b, rows, 1058); });
    }
    {   // hipMemcpy D2D as an external reference
        timeit("hipMemcpyAsync D2D", 2 * gb, [&] { CK(hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, 0)); });
    }
    return 0;
}

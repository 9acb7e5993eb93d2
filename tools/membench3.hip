// membench3.hip — does a persistent, software-pipelined row copy beat one wave per row on SHORT rows (8.4 KB, the
// 16-bit V = 4232 row of K3)?  Developer tool: hipcc -O3 --offload-arch=gfx950 tools/membench3.hip -o tools/membench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// (a) one wave per row, as K3: the wave loads its row, touches it, stores it, exits
template <int NCH, int WPE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void k_one(const u32x4* __restrict__ in, u32x4* __restrict__ out, long long rows, int rowchunks) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    if (r >= rows) return;
    const u32x4* src = in + r * rowchunks; u32x4* dst = out + r * rowchunks;
    u32x4 v[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) { int j = lane + 64 * k; if (j < rowchunks) v[k] = __builtin_nontemporal_load(src + j); }
#pragma unroll
    for (int k = 0; k < NCH; ++k) { int j = lane + 64 * k; if (j < rowchunks) { v[k].x += 1; __builtin_nontemporal_store(v[k], dst + j); } }
}
// (b) persistent: a wave walks rows r, r + nwaves, ...; the next row's loads are issued before this row's stores
template <int NCH, int WPE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void k_pipe(const u32x4* __restrict__ in, u32x4* __restrict__ out, long long rows, int rowchunks) {
    const int lane = threadIdx.x & 63;
    const long long w0 = (long long)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, nw = (long long)gridDim.x * (blockDim.x / 64);
    u32x4 cur[NCH], nxt[NCH];
    if (w0 < rows) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) { int j = lane + 64 * k; cur[k] = __builtin_nontemporal_load(in + w0 * rowchunks + (j < rowchunks ? j : rowchunks - 1)); }
    }
    for (long long r = w0; r < rows; r += nw) {
        const long long rn = r + nw < rows ? r + nw : r;  // (the last iteration re-reads its own row)
#pragma unroll
        for (int k = 0; k < NCH; ++k) { int j = lane + 64 * k; nxt[k] = __builtin_nontemporal_load(in + rn * rowchunks + (j < rowchunks ? j : rowchunks - 1)); }
        u32x4* dst = out + r * rowchunks;
#pragma unroll
        for (int k = 0; k < NCH; ++k) { int j = lane + 64 * k; if (j < rowchunks) { cur[k].x += 1; __builtin_nontemporal_store(cur[k], dst + j); } }
#pragma unroll
        for (int k = 0; k < NCH; ++k) cur[k] = nxt[k];
    }
}

template <typename F>
float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

template <int NCH>
void run(const char* name, int rowchunks, long long rows, u32x4* in, u32x4* out) {
    const double gb = 2.0 * rows * rowchunks * 16 / 1e9;
    float t;
    t = timeit([&] { hipLaunchKernelGGL((k_one<NCH, 2>), dim3((unsigned)((rows + 7) / 8)), dim3(512), 0, 0, in, out, rows, rowchunks); }, 10);
    printf("%s one wave per row, 512 threads, 2 waves/SIMD: %.3f ms  %.0f GB/s\n", name, t, gb / t * 1e3);
    t = timeit([&] { hipLaunchKernelGGL((k_one<NCH, 2>), dim3((unsigned)((rows + 1) / 2)), dim3(128), 0, 0, in, out, rows, rowchunks); }, 10);
    printf("%s one wave per row, 128 threads, 2 waves/SIMD: %.3f ms  %.0f GB/s\n", name, t, gb / t * 1e3);
    t = timeit([&] { hipLaunchKernelGGL((k_one<NCH, 4>), dim3((unsigned)((rows + 7) / 8)), dim3(512), 0, 0, in, out, rows, rowchunks); }, 10);
    printf("%s one wave per row, 512 threads, 4 waves/SIMD: %.3f ms  %.0f GB/s\n", name, t, gb / t * 1e3);
    for (int wgs : {256, 512, 1024, 2048}) {
        t = timeit([&] { hipLaunchKernelGGL((k_pipe<NCH, 2>), dim3(wgs), dim3(512), 0, 0, in, out, rows, rowchunks); }, 10);
        printf("%s persistent pipelined, %d x 512 threads, 2 waves/SIMD: %.3f ms  %.0f GB/s\n", name, wgs, t, gb / t * 1e3);
    }
    for (int wgs : {512, 1024}) {
        t = timeit([&] { hipLaunchKernelGGL((k_pipe<NCH, 3>), dim3(wgs), dim3(256), 0, 0, in, out, rows, rowchunks); }, 10);
        printf("%s persistent pipelined, %d x 256 threads, 3 waves/SIMD: %.3f ms  %.0f GB/s\n", name, wgs, t, gb / t * 1e3);
    }
}

int main() {
    const long long rows = 524800;
    u32x4 *in, *out;
    const size_t bytes = (size_t)rows * 1058 * 16;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    run<9>("8464-byte rows:", 529, rows, in, out);
    run<17>("16928-byte rows:", 1058, rows, in, out);
    return 0;
}

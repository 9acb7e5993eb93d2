// mfma_valu_bench.hip — how the matrix pipe and the vector ALU of one SIMD share time (gfx950).
//   hipcc -O3 --offload-arch=gfx950 -o tools/mfma_valu_bench tools/mfma_valu_bench.hip && ./tools/mfma_valu_bench
// One workgroup of 8 waves per CU (waves w and w+4 share a SIMD).  Per mode, cycles (s_memtime) that wave 0 / wave 4 need:
//   mma      waves 0-3: N MFMA 32x32x16 bf16 (two independent accumulators), waves 4-7 idle
//   valu     waves 4-7: M v_fma_f32 (8 independent chains), waves 0-3 idle
//   exp      waves 4-7: M v_exp_f32
//   mma|valu both at once, on different waves of the SIMD
//   mma|exp
//   mma+valu ONE wave: each MFMA followed by K independent v_fma_f32 (same wave interleave)
//   mma|mma  both waves run MFMAs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int K>
__device__ __forceinline__ void mma_loop(int n, f32x16 &a0, f32x16 &a1, bf16x8 x, bf16x8 y, float (&v)[8]) {
    for (int i = 0; i < n; i += 2) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], 1.0001f, 0.5f);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) v[(k + 4) & 7] = __builtin_fmaf(v[(k + 4) & 7], 1.0001f, 0.5f);
    }
}

__global__ __launch_bounds__(512) void bench(int mode, int n_mma, int n_valu, long long *out, float *sink) {
    const int wave = threadIdx.x >> 6;
    const bool first = wave < 4;
    f32x16 a0 = {}, a1 = {};
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(0.001f * (threadIdx.x + i)); y[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.001f * (threadIdx.x + i);
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    bool did = false;
    if (first) {
        if (mode == 0 || mode == 3 || mode == 4 || mode == 6) { mma_loop<0>(n_mma, a0, a1, x, y, v); did = true; }
        if (mode == 5) { mma_loop<4>(n_mma, a0, a1, x, y, v); did = true; }
        if (mode == 7) { mma_loop<7>(n_mma, a0, a1, x, y, v); did = true; }
    } else {
        if (mode == 1 || mode == 3) {
            for (int i = 0; i < n_valu; i += 8)
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = __builtin_fmaf(v[k], 1.0001f, 0.5f);
            did = true;
        }
        if (mode == 2 || mode == 4) {
            for (int i = 0; i < n_valu; i += 8)
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = __builtin_amdgcn_exp2f(v[k]);
            did = true;
        }
        if (mode == 6) { mma_loop<0>(n_mma, a0, a1, x, y, v); did = true; }
        if (mode == 8 || mode == 9) {  // ONE dependent chain of fma
            for (int i = 0; i < n_valu; i += 8)
#pragma unroll
                for (int k = 0; k < 8; ++k) v[0] = __builtin_fmaf(v[0], 1.0001f, 0.5f);
            did = true;
        }
        if (mode == 10 || mode == 11) {  // two chains
            for (int i = 0; i < n_valu; i += 8)
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k & 1] = __builtin_fmaf(v[k & 1], 1.0001f, 0.5f);
            did = true;
        }
        if (mode == 12 || mode == 13) {  // exp, then an add that needs it, 8 independent pairs
            for (int i = 0; i < n_valu; i += 16)
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = v[k] * 0.5f + __builtin_amdgcn_exp2f(v[k]);
            did = true;
        }
        if (mode == 14 || mode == 15) {  // exp -> dependent add into ONE accumulator
            float acc1 = 0.0f;
            for (int i = 0; i < n_valu; i += 16) {
#pragma unroll
                for (int k = 0; k < 8; ++k) acc1 += __builtin_amdgcn_exp2f(v[k] + (float)i);
            }
            v[0] += acc1;
            did = true;
        }
        if (mode == 16 || mode == 17) {  // v_pk_fma_f32, 4 independent chains of pairs
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p[4] = {{v[0], v[1]}, {v[2], v[3]}, {v[4], v[5]}, {v[6], v[7]}};
            const f2 c1 = {1.0001f, 1.0002f}, c2 = {0.5f, 0.25f};
            for (int i = 0; i < n_valu; i += 8)
#pragma unroll
                for (int k = 0; k < 8; ++k) p[k & 3] = __builtin_elementwise_fma(p[k & 3], c1, c2);
            for (int k = 0; k < 4; ++k) { v[2 * k] = p[k].x; v[2 * k + 1] = p[k].y; }
            did = true;
        }
    }
    if (first && (mode == 9 || mode == 11 || mode == 13 || mode == 15 || mode == 17)) { mma_loop<0>(n_mma, a0, a1, x, y, v); did = true; }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wave] = did ? t1 - t0 : 0;
}

int main() {
    long long *out; float *sink;
    CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&sink, 4));
    const char *names[] = {"mma (waves 0-3)", "valu fma (waves 4-7)", "exp (waves 4-7)", "mma | valu", "mma | exp",
                           "mma + 4 fma per MFMA, one wave", "mma | mma", "mma + 7 fma per MFMA, one wave",
                           "fma, ONE dependent chain", "mma | fma one chain", "fma, two chains", "mma | fma two chains",
                           "8 x (exp + dependent fma) indep.", "mma | 8 x (exp + dep. fma)", "exp -> add into one accumulator",
                           "mma | exp -> one accumulator", "v_pk_fma_f32, 4 chains", "mma | v_pk_fma_f32"};
    const int n_mma = 2048, n_valu = 16384;
    printf("N = %d MFMA 32x32x16 (ideal %d cycles), M = %d vector instructions (ideal %d cycles; exp at quarter rate %d)\n",
           n_mma, n_mma * 32, n_valu, n_valu * 4, n_valu * 16);
    for (int mode = 0; mode < 18; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(bench, dim3(256), dim3(512), 0, 0, mode, n_mma, n_valu, out, sink);
            CHECK(hipDeviceSynchronize());
        }
        long long h[8];
        CHECK(hipMemcpy(h, out, 64, hipMemcpyDeviceToHost));
        printf("%-36s wave0 %8lld   wave4 %8lld\n", names[mode], h[0], h[4]);
    }
    return 0;
}

// Pure-MFMA issue-rate probe for gfx950: no memory traffic in the loop.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int KIND>   // KIND 0: 16x16x32, 1: 32x32x16
__global__ __launch_bounds__(256) void probe(const half8v* in, float* out, int iters) {
    half8v a = in[threadIdx.x], b = in[threadIdx.x + 256];
    if (KIND == 0) {
        f32x4 acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0;
#pragma unroll
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        f32x16 acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0;
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) s += acc[i][j];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

template <int NACC, int KIND>
void run(const char* name, int blocks, int threads, const half8v* in, float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NACC, KIND>), dim3(blocks), dim3(threads), 0, 0, in, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NACC, KIND>), dim3(blocks), dim3(threads), 0, 0, in, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop_per_mfma = KIND == 0 ? 16.0 * 16 * 32 * 2 : 32.0 * 32 * 16 * 2;
    const double waves = (double)blocks * threads / 64;
    const double tf = waves * iters * NACC * flop_per_mfma / (ms * 1e-3) / 1e12;
    printf("%-44s blocks=%5d thr=%4d  %8.3f ms  %8.1f TF/s\n", name, blocks, threads, ms, tf);
}

int main() {
    half8v* in; float* out;
    hipMalloc(&in, 1024 * 16); hipMalloc(&out, 4096 * 1024 * 4);
    _Float16 h[1024 * 8];
    for (int mode = 0; mode < 2; ++mode) {
        for (int i = 0; i < 1024 * 8; ++i) h[i] = mode ? (_Float16)((rand() % 2001 - 1000) / 1000.0f) : (_Float16)0.f;
        hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
        printf("---- operands: %s\n", mode ? "random in [-1,1]" : "zeros");
        const int it = 20000;
        run<16, 0>("16x16x32 16 acc, 1 wave/SIMD (256 blk x 256)", 256, 256, in, out, it);
        run<16, 0>("16x16x32 16 acc, 2 waves/SIMD (512 blk x 256)", 512, 256, in, out, it);
        run<16, 0>("16x16x32 16 acc, 4 waves/SIMD (1024 blk x 256)", 1024, 256, in, out, it);
        run<50, 0>("16x16x32 50 acc, 1 wave/SIMD", 256, 256, in, out, it / 2);
        run<8, 1>("32x32x16 8 acc, 1 wave/SIMD", 256, 256, in, out, it);
        run<8, 1>("32x32x16 8 acc, 2 waves/SIMD", 512, 256, in, out, it);
        run<16, 0>("16x16x32 16 acc, 1 wave/SIMD, 128 CUs only", 128, 256, in, out, it);
        run<16, 0>("16x16x32 16 acc, 1 wave/SIMD, 32 CUs only", 32, 256, in, out, it);
    }
    return 0;
}

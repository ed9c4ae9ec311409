// Prototype / micro-benchmark for DESIGN section 9 "(1)": is a 256 x 256 block of FOUR waves with 128 x 128 wave tiles (one wave per SIMD,
// 256 accumulator registers in the AGPR half of the register file) faster at the power cap than the library's 8-wave kernels (80 x 80 /
// 64 x 128 wave tiles, two waves per SIMD)?  0.25 fragment reads per MFMA instead of 0.375-0.40, the same LDS-DMA ring and swizzles as
// gemm_sqp_kernel (moca_video_amd/csrc/gemm.hip).  Plain linear only: out[M][N] = A[M][K] . W[N][K]^T, fp16 in / out, fp32 accumulate;
// M % 256 == 0, N % 192 == 0, K % 64 == 0.  Not part of the product.
//
//   hipcc -O3 --offload-arch=gfx950 tools/micro/gemm128.hip -o tools/micro/gemm128 && tools/micro/gemm128 [M N K]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <vector>

typedef _Float16 half_t;
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char* lds_ptr;

constexpr int TM = 256, BN = 192, KS = 32, RB = 64;            // block tile, k per slot, bytes per LDS row
constexpr int A_BYTES = TM * RB, STAGE = A_BYTES + BN * RB;    // 32 KiB per k-tile
constexpr int NS = 5;                                          // ring slots (160 KiB)
constexpr int MT = 8, NT = 6;                                  // MFMA tiles per wave: 128 x 128
constexpr unsigned OOB = 0x80000000u;

template <int V> struct int_c { static constexpr int value = V; };

__global__ __launch_bounds__(256) void gemm128_kernel(const half_t* __restrict__ a, const half_t* __restrict__ w, half_t* __restrict__ out,
                                                      int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int tiles_n = N / BN, tiles_m = M / TM, nblk = tiles_m * tiles_n;
    // XCD-aware remap: each XCD (blockIdx & 7) owns a contiguous range of the tile_m-major raster
    int logical;
    {
        const int b = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = b & 7, j = b >> 3;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tile_m = logical / tiles_n, tile_n = logical - tile_m * tiles_n;
    const int m0 = tile_m * TM, n0 = tile_n * BN;
    const int nk = K / KS;

    // ---- LDS-DMA: piece = 16 rows x 64 B; per k-tile 16 A pieces + 16 W pieces = 8 per wave: A pieces 4 wave .. 4 wave + 3, W likewise ----
    const int lrow = lane >> 2, pch = lane & 3;
    const int lch = pch ^ ((0x78 >> (2 * ((lrow >> 2) & 3))) & 3);
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a), 0, OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(w), 0, OOB, 0x00020000);
    unsigned a_off[4], w_off[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        a_off[g] = (unsigned)(((int64_t)(m0 + (4 * wave + g) * 16 + lrow) * K + lch * 8) * 2);
        w_off[g] = (unsigned)(((int64_t)(n0 + (3 * wave + (g < 3 ? g : 0)) * 16 + lrow) * K + lch * 8) * 2);
    }
    auto issue_tile = [&](int kt, int slot, int g) {          // one A piece and one W piece of k-tile kt
        const lds_ptr s = (lds_ptr)smem + slot * STAGE;
        const unsigned soff = (unsigned)(kt * KS * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, s + (4 * wave + g) * 1024, 16, kt < nk ? a_off[g] : OOB, soff, 0, 0);
        if (g < 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, s + A_BYTES + (3 * wave + g) * 1024, 16, kt < nk ? w_off[g] : OOB, soff, 0, 0);
    };
    auto issue_pair = [&](int kt, int s_even, int s_odd) {    // 16 DMA instructions
#pragma unroll
        for (int g = 0; g < 4; ++g) { issue_tile(kt, s_even, g); issue_tile(kt + 1, s_odd, g); }
    };

    const int fr = lane & 15, fg = lane >> 4;
    const int swz = (fg ^ ((0x78 >> (2 * ((fr >> 2) & 3))) & 3)) << 4;
    const int a_rd = (wave_m * 128 + fr) * RB + swz;
    const int b_rd = A_BYTES + (wave_n * 96 + fr) * RB + swz;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    half8v af[MT], bf[2][NT];          // A fragments are refreshed in place (row mt right after its last MFMA), B fragments double-buffered

    // ---- prologue: pairs (0,1) and (2,3) in flight, the first landed everywhere, its fragments read ----
    issue_pair(0, 0, 1);
    issue_pair(2, 2, 3);
    asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        const char* cur = smem;
#pragma unroll
        for (int r = 0; r < NT; ++r) bf[0][r] = *reinterpret_cast<const half8v*>(cur + b_rd + r * 1024);
#pragma unroll
        for (int r = 0; r < MT; ++r) af[r] = *reinterpret_cast<const half8v*>(cur + a_rd + r * 1024);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    int s0 = 0;                                                // ring slot of k-tile i
    for (int i = 0; i < nk; i += 2) {
        const int s1 = s0 + 1 == NS ? 0 : s0 + 1;              // k-tile i + 1
        const int s2 = s1 + 1 == NS ? 0 : s1 + 1;              // k-tile i + 2
        const int sp = s0 == 0 ? NS - 1 : s0 - 1;              // free (k-tile i - 1)
        // ---- phase E: 64 MFMAs of k-tile i (B set 0); row mt's A fragment is replaced by k-tile i + 1's right after its last use, the 8 B
        //      fragments of k-tile i + 1 go to set 1 ----
        {
            const char* nx = smem + s1 * STAGE;
#pragma unroll
            for (int j = 0; j < MT * NT; ++j) {
                const int mt = j / NT, nt = j % NT;
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[0][nt], af[mt], acc[mt][nt], 0, 0, 0);
                if (nt == 3 && mt < NT) {
                    __builtin_amdgcn_sched_barrier(0);
                    bf[1][mt] = *reinterpret_cast<const half8v*>(nx + b_rd + mt * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (nt == NT - 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    af[mt] = *reinterpret_cast<const half8v*>(nx + a_rd + mt * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // the pair (i + 2, i + 3) -- issued during the previous iteration's phase O -- is this wave's only outstanding DMA: wait for it here, so that
        // ONE barrier per iteration publishes both "k-tiles i, i + 1 are in registers everywhere" and "k-tiles i + 2, i + 3 have landed everywhere"
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase O: 64 MFMAs of k-tile i + 1 (B set 1) with, in the gaps, the 14 DMA instructions of the pair (i + 4, i + 5) -> slots (sp, s0)
        //      and the fragment reads of k-tile i + 2 (A in place, B -> set 0) ----
        {
            const char* nx = smem + s2 * STAGE;
#pragma unroll
            for (int j = 0; j < MT * NT; ++j) {
                const int mt = j / NT, nt = j % NT;
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[1][nt], af[mt], acc[mt][nt], 0, 0, 0);
                if (nt == 1) {
                    const int g = mt;                          // 8 steps x (A piece + W piece)
                    __builtin_amdgcn_sched_barrier(0);
                    if (g < 4) issue_tile(i + 4, sp, g);
                    else issue_tile(i + 5, s0, g - 4);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (nt == 3 && mt < NT) {
                    __builtin_amdgcn_sched_barrier(0);
                    bf[0][mt] = *reinterpret_cast<const half8v*>(nx + b_rd + mt * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (nt == NT - 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    af[mt] = *reinterpret_cast<const half8v*>(nx + a_rd + mt * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        s0 = s2;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- epilogue: fp16 tile staged in LDS (row pitch 256 * 2 + 16 B), whole rows out ----
    constexpr int pitch = BN * 2 + 16;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = wave_n * 96 + nt * 16 + 4 * fg;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = wave_m * 128 + mt * 16 + fr;
            *reinterpret_cast<half4v*>(smem + row * pitch + col * 2) = __builtin_convertvector(acc[mt][nt], half4v);
        }
    }
    __syncthreads();
    for (int idx = tid; idx < TM * (BN / 8); idx += 256) {
        const int row = idx / (BN / 8), ch = idx - row * (BN / 8);
        *reinterpret_cast<half8v*>(out + (int64_t)(m0 + row) * N + n0 + ch * 8) = *reinterpret_cast<const half8v*>(smem + row * pitch + ch * 16);
    }
}

int main(int argc, char** argv) {
    const int M = argc > 3 ? atoi(argv[1]) : 40960, N = argc > 3 ? atoi(argv[2]) : 1280, K = argc > 3 ? atoi(argv[3]) : 5120;
    if (M % 256 || N % 192 || K % 64 || (int64_t)M * K * 2 >= (1ll << 31) || (int64_t)N * K * 2 >= (1ll << 31)) { printf("bad shape\n"); return 1; }
    std::vector<half_t> ha((size_t)M * K), hw((size_t)N * K);
    uint32_t st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };   // uniform (-0.5, 0.5)
    for (auto& x : ha) x = (half_t)(rnd() * 2.0f);
    for (auto& x : hw) x = (half_t)(rnd() * 2.0f / sqrtf((float)K) * 4.0f);
    half_t *da, *dw, *dout;
    if (hipMalloc(&da, ha.size() * 2) != hipSuccess || hipMalloc(&dw, hw.size() * 2) != hipSuccess || hipMalloc(&dout, (size_t)M * N * 2) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemset(dout, 0, (size_t)M * N * 2);
    const int lds = NS * STAGE;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) { printf("lds attr failed\n"); return 1; }
    const int nblk = (M / 256) * (N / 192);
    hipLaunchKernelGGL(gemm128_kernel, dim3(nblk), dim3(256), lds, 0, da, dw, dout, M, N, K);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    // check 512 sampled outputs against fp32 dot products of the fp16 operands
    std::vector<half_t> ho((size_t)M * N);
    hipMemcpy(ho.data(), dout, ho.size() * 2, hipMemcpyDeviceToHost);
    double maxerr = 0.0, maxref = 0.0;
    for (int s = 0; s < 512; ++s) {
        st = st * 1664525u + 1013904223u; const int m = (st >> 4) % M;
        st = st * 1664525u + 1013904223u; const int n = (st >> 4) % N;
        double ref = 0.0;
        for (int k = 0; k < K; ++k) ref += (double)(float)ha[(size_t)m * K + k] * (double)(float)hw[(size_t)n * K + k];
        maxerr = fmax(maxerr, fabs(ref - (double)(float)ho[(size_t)m * N + n]));
        maxref = fmax(maxref, fabs(ref));
    }
    printf("M=%d N=%d K=%d  tiles %d  check: max |err| %.3e of max |ref| %.3e -> %s\n", M, N, K, nblk, maxerr, maxref, maxerr <= 3e-3 * maxref ? "ok" : "WRONG");
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        const int it = 200;
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(gemm128_kernel, dim3(nblk), dim3(256), lds, 0, da, dw, dout, M, N, K);
        hipEventRecord(e0);
        for (int i = 0; i < it; ++i) hipLaunchKernelGGL(gemm128_kernel, dim3(nblk), dim3(256), lds, 0, da, dw, dout, M, N, K);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        printf("  %8.1f us  %7.1f TFLOP/s\n", ms / it * 1e3, 2.0 * M * N * K / (ms / it * 1e-3) / 1e12);
    }
    return 0;
}

// Weight-stationary streaming linear for the HBM-bound 320 -> 320 projections (round 6).
//
// Replaces, for N == K == 320: `proj_in` / `proj_out` of SpatialTransformer / TemporalTransformer (attention.py:242,258,302,328) and
// `to_q`, `to_out[0]` of CrossAttention (attention.py:54-57) at the 320-channel level, with the bias / residual add
// (attention.py:217-219,278,373) and the LayerNorm statistics of the consumer (attention.py:199-201) in the epilogue.
//
// Why another kernel.  These launches move 3 x M x 640 B (A in, residual in, out) for 2 x M x 320 x 320 FLOP: 0.10 of the matrix pipe at
// the HBM rate.  On the staggered 160 x 320 tiling (gemm_w80s_kernel<0, 1>) a CU runs prologue -> main loop -> store loop once per
// 160-row tile and nothing streams during two of the three phases (profiles/r05_g4_phase_stamps.txt: store loop 45 %, main loop
// 31 %, prologue 17 % of a tile), and the 200 KB weight matrix is re-streamed L2 -> LDS for every 100 KB of A: 3.5-4.0 TB/s isolated,
// ~2.9 TB/s inside the graph (cold residual).  Here
//   * W never moves again: each of a block's 4 waves keeps its 80 output columns x 320 k of W as MFMA fragments in 200 registers
//     (one wave per SIMD, 512 registers each; loaded once per block);
//   * A AND the residual rows stream through one LDS ring by LDS-DMA, 3 strips of 32 rows (40 KB each) ahead of the strip being
//     computed -- every byte a block needs is requested ~100 KB ahead and no wave ever waits for a register load (vmcnt is in-order:
//     one register load consumed per strip would pull the whole DMA queue in with it);
//   * a DMA instruction fetches ONE MFMA fragment (16 rows x 64 B of A: lane l takes row l % 16, chunk l / 16) or one accumulator-
//     shaped piece of the residual, so its 1 KB lands contiguously and is read back with ds_read_b128 at base + 16 lane: no swizzle,
//     no bank conflict, addresses are immediates;
//   * the epilogue runs from registers: W rows are assigned to MFMA rows in a permuted order so that a lane's accumulators in two
//     neighbouring tiles are 8 consecutive output columns (16-byte stores, 16 rows x 64 B per instruction);
//   * one s_barrier per strip; the waits are counted (`vmcnt(20)`: the two younger strips stay in flight).
// Algorithmic bytes per launch: M x 640 B x (2 + residual) + 200 KB x blocks of W (from L2).
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) char* lds_ptr;
constexpr unsigned WS_OOB = 0x80000000u;
constexpr int WS_C = 320;                       // N == K
constexpr int WS_ROWS = 32;                     // rows per strip (two MFMA row tiles)
constexpr int WS_KSTEPS = WS_C / 32;            // 10 k-steps of v_mfma_f32_16x16x32_f16
constexpr int WS_A_BYTES = WS_ROWS * WS_C * 2;  // 20 KiB: 20 fragments of 1 KiB, [row tile][k-step]
constexpr int WS_R_BYTES = WS_ROWS * WS_C * 2;  // 20 KiB: per wave 5 pieces of 1 KiB, [wave][piece]
// ring: 160 KiB either way -- 4 slots of (A + residual) = 40 KiB with a residual, 8 slots of 20 KiB without; SLOTS - 1 strips in flight
// ahead of the one being computed

// column (inside a wave's 80) held by MFMA row i of column tile j: tiles (0,1) and (2,3) interleave so that rows 4q..4q+3 of a pair
// are 8 consecutive columns; tile 4 is plain
__device__ __forceinline__ int ws_col(int j, int i) {
    if (j == 4) return 64 + i;
    return 32 * (j >> 1) + 8 * (i >> 2) + 4 * (j & 1) + (i & 3);
}

template <int V> struct int_c { static constexpr int value = V; };

// s_waitcnt vmcnt(BASE + min(k, MAXK) * STEP) with immediate operands
template <int BASE, int STEP, int MAXK, int I = 0>
__device__ __forceinline__ void ws_wait(int k) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (I >= MAXK) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE + MAXK * STEP) : "memory");
    } else {
        if (k == I) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE + I * STEP) : "memory");
        else ws_wait<BASE, STEP, MAXK, I + 1>(k);
    }
#endif
}

template <bool RES, bool ROWSUM>
__global__ __launch_bounds__(256) void gemm_ws_kernel(const moca_gemm_params p, const int nstrips) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // vmcnt retires loads, LDS-DMA and stores together in issue order, so "strip k + 1 has landed" = all but the N youngest operations are
    // done, N = the DMA pieces AND the stores this wave has issued since: AHEAD - 1 groups of each (N <= 63 bounds AHEAD without a residual)
    constexpr int DMA_PER_STRIP = RES ? 10 : 5;  // per wave: 5 fragments of A (+ its 5 residual pieces)
    constexpr int ST_PER_STRIP = ROWSUM ? 8 : 6; // per wave: 2 row tiles x (16 B, 16 B, 8 B per lane) (+ the row partial)
    constexpr int WS_SLOT = WS_A_BYTES + (RES ? WS_R_BYTES : 0), WS_SLOTS = RES ? 4 : 8, WS_AHEAD = RES ? 3 : (ROWSUM ? 5 : 6);
    constexpr int WAIT_N = (WS_AHEAD - 1) * (DMA_PER_STRIP + ST_PER_STRIP);
    static_assert(WAIT_N <= 63 && WS_AHEAD <= WS_SLOTS - 1, "vmcnt is a 6-bit counter; a slot is refilled one barrier after its last read");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;

    // ---- this block's strips: b, b + G, b + 2 G, ... -- at any moment the G blocks stream ONE contiguous window of G x 20 KB, spread over
    //      every HBM channel (a contiguous range per block puts all blocks on addresses a multiple of M / G rows apart: 1.2-1.3 x slower
    //      at M = 655360, profiles/r06_ab_gemm_ws.txt) ----
    const int G = gridDim.x;
    const int n_my = (nstrips - (int)blockIdx.x + G - 1) / G;
    if (n_my <= 0) return;

    // ---- DMA stream.  A: fragment f = 5 wave + g (g < 5) of a strip's 20, f = 10 t + s: lane takes row 16 t + fr, bytes 64 s + 16 fq.
    //      Residual: this wave's own 5 pieces: (t, pair) = 16 rows x 64 B at byte column 160 wave + 64 pair + 16 fq, and the tile-4 piece:
    //      lanes 0..31 row tile 0, 32..63 row tile 1, 16 B at byte column 160 wave + 128 + 16 (fq & 1). ----
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, WS_OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(RES ? p.residual : p.a), 0, WS_OOB, 0x00020000);
    unsigned a_rel[5], r_rel[5];                 // byte offsets inside a strip (row 0 of the strip = 0)
#pragma unroll
    for (int g = 0; g < 5; ++g) {
        const int f = 5 * wave + g, t = f / WS_KSTEPS, s = f - t * WS_KSTEPS;
        a_rel[g] = (unsigned)(((16 * t + fr) * p.lda) * 2 + 64 * s + 16 * fq);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int t = g >> 1, pr = g & 1;
        r_rel[g] = (unsigned)(((16 * t + fr) * p.ldr) * 2 + 160 * wave + 64 * pr + 16 * fq);
    }
    r_rel[4] = (unsigned)(((16 * (fq >> 1) + fr) * p.ldr) * 2 + 160 * wave + 128 + 16 * (fq & 1));
    auto issue = [&](int k) {                    // k-th strip of this block -> ring slot k % 4 (k >= n_my: zero fill, no traffic)
        const bool live = k < n_my;
        const int64_t row0 = ((int64_t)blockIdx.x + (int64_t)k * G) * WS_ROWS;
        const unsigned a0 = (unsigned)(row0 * p.lda * 2), r0 = (unsigned)(row0 * p.ldr * 2);
        const lds_ptr slot = (lds_ptr)smem + (k & (WS_SLOTS - 1)) * WS_SLOT;
#pragma unroll
        for (int g = 0; g < 5; ++g)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, slot + (5 * wave + g) * 1024, 16, live ? a_rel[g] + a0 : WS_OOB, 0, 0, 0);
        if constexpr (RES) {
#pragma unroll
            for (int g = 0; g < 5; ++g)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_r, slot + WS_A_BYTES + (5 * wave + g) * 1024, 16, live ? r_rel[g] + r0 : WS_OOB, 0, 0, 0);
        }
    };
    // ---- W: this wave's 80 columns x 320 k as 5 x 10 MFMA fragments (A operand of the swapped product: lane = W row fr of the tile,
    //      k chunk fq), once per block; bias of the lane's 20 output columns ----
    half8v wf[5][WS_KSTEPS];
    {
        const half_t* w = reinterpret_cast<const half_t*>(p.w);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const half_t* wr = w + (int64_t)(80 * wave + ws_col(j, fr)) * p.ldw + 8 * fq;
#pragma unroll
            for (int s = 0; s < WS_KSTEPS; ++s) wf[j][s] = *reinterpret_cast<const half8v*>(wr + 32 * s);
        }
    }
    float bias[20];                                  // columns 80 wave + {8 fq .. +7, 32 + 8 fq .. +7, 64 + 4 fq .. +3}
#pragma unroll
    for (int c = 0; c < 20; ++c) bias[c] = 0.f;
    if (p.bias) {
        const float* bp = p.bias + 80 * wave;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp + 8 * fq), b1 = *reinterpret_cast<const f32x4*>(bp + 8 * fq + 4);
        const f32x4 b2 = *reinterpret_cast<const f32x4*>(bp + 32 + 8 * fq), b3 = *reinterpret_cast<const f32x4*>(bp + 32 + 8 * fq + 4);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bp + 64 + 4 * fq);
#pragma unroll
        for (int c = 0; c < 4; ++c) { bias[c] = b0[c]; bias[4 + c] = b1[c]; bias[8 + c] = b2[c]; bias[12 + c] = b3[c]; bias[16 + c] = b4[c]; }
    }
    // (W and the bias are requested BEFORE the stream starts: vmcnt retires in order, behind the first strips' DMAs they would arrive
    //  only after those -- an HBM round trip later than needed)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k0 = 0; k0 < WS_AHEAD; ++k0) issue(k0);
    half_t* const out = reinterpret_cast<half_t*>(p.out);
#ifdef MOCA_WS_ABLATE                            // timing-only diagnostic builds (wrong results): 1 one k-step of MFMAs, 2 no stores, 4 no epilogue reads
    constexpr int abl = MOCA_WS_ABLATE;
#else
    constexpr int abl = 0;
#endif
    const unsigned a_rd = (unsigned)lane * 16;                                      // fragment read: base + 16 lane
    const unsigned r_rd = (unsigned)(WS_A_BYTES + 5 * wave * 1024) + (unsigned)lane * 16;
    const unsigned r4_rd = (unsigned)(WS_A_BYTES + (5 * wave + 4) * 1024) + (unsigned)(((fq >> 1) * 16 + fr) * 16 + (fq & 1) * 8);

    // strip 0 has landed (this wave's share; the younger strips may stay in flight)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((WS_AHEAD - 1) * DMA_PER_STRIP) : "memory");      // (no stores yet)
    for (int k = 0; k < n_my; ++k) {
        // strip k has landed everywhere (each wave waited for its share at the end of the previous iteration) and every wave is done
        // with strip k - 1, whose slot the stream refills now
        __builtin_amdgcn_s_barrier();
        issue(k + WS_AHEAD);
        const char* slot = smem + (k & (WS_SLOTS - 1)) * WS_SLOT;
        const int64_t row0 = ((int64_t)blockIdx.x + (int64_t)k * G) * WS_ROWS;
        half8v o0[2], o1[2];
        half4v o2[2];
        float sum[2], sq[2];
        // every fragment read of the strip is issued before the first MFMA (80 registers; the compiler's own schedule kept two reads in
        // flight and the matrix pipe idled for an LDS round trip in each of the 10 k-steps: 2.0 us of arithmetic per strip)
        half8v af[2][WS_KSTEPS];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < WS_KSTEPS; ++s) af[t][s] = *reinterpret_cast<const half8v*>(slot + (t * WS_KSTEPS + s) * 1024 + a_rd);
        half8v r0v[2], r1v[2];
        half4v r2v[2];
        if constexpr (RES) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                r0v[t] = *reinterpret_cast<const half8v*>(slot + r_rd + (2 * t) * 1024);
                r1v[t] = *reinterpret_cast<const half8v*>(slot + r_rd + (2 * t + 1) * 1024);
                r2v[t] = *reinterpret_cast<const half4v*>(slot + r4_rd + t * 512);
            }
        }
        __builtin_amdgcn_sched_barrier(0);           // (keeps the reads above: the machine scheduler sinks them back next to their MFMAs)
        f32x4 acc[2][5];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // epilogue arithmetic of one row tile: lane = row fr, columns 80 wave + {8 fq .. +7, 32 + 8 fq .. +7, 64 + 4 fq .. +3}
        auto finish = [&](auto t_tag) {
            constexpr int t = decltype(t_tag)::value;
            sum[t] = 0.f; sq[t] = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float v0 = acc[t][c >> 2][c & 3] + bias[c], v1 = acc[t][2 + (c >> 2)][c & 3] + bias[8 + c];
                if constexpr (RES) { v0 += (float)r0v[t][c]; v1 += (float)r1v[t][c]; }
                o0[t][c] = (half_t)v0; o1[t][c] = (half_t)v1;
                if constexpr (ROWSUM) {
                    const float h0 = (float)o0[t][c], h1 = (float)o1[t][c];
                    sum[t] += h0 + h1; sq[t] += h0 * h0 + h1 * h1;
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float v = acc[t][4][c] + bias[16 + c];
                if constexpr (RES) v += (float)r2v[t][c];
                o2[t][c] = (half_t)v;
                if constexpr (ROWSUM) { const float h = (float)o2[t][c]; sum[t] += h; sq[t] += h * h; }
            }
            if constexpr (ROWSUM) {                  // partial `wave` of 4: (sum, sum of squares) over this wave's 80 stored columns of a row
                sum[t] += __shfl_xor(sum[t], 16, 64); sq[t] += __shfl_xor(sq[t], 16, 64);
                sum[t] += __shfl_xor(sum[t], 32, 64); sq[t] += __shfl_xor(sq[t], 32, 64);
            }
        };
        // (tile 0's epilogue arithmetic placed in the MFMA gaps of tile 1 with sched_group_barrier: no difference, 253 vs 253 us / 279 vs 270 us
        //  at M = 655360 -- what a strip costs is the sum of what its ONE wave per SIMD issues: DMA pieces, fragment reads, MFMAs, epilogue)
#pragma unroll
        for (int s = 0; s < ((abl & 1) ? 1 : WS_KSTEPS); ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < 5; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j][s], af[t][s], acc[t][j], 0, 0, 0);
        finish(int_c<0>{});
        finish(int_c<1>{});
        // strip k + 1 has landed (this wave's share), counted exactly: the younger DMA groups and the store groups issued since stay in
        // flight.  (First form: the wait counted the DMA pieces only and sat behind the strip's fresh stores -- it then also waited for
        // AHEAD - 1 younger strips' worth of operations, and with `nt` stores, acknowledged late, the kernel was 1.1-1.4 x slower than the
        // tiled one.)  Strip k has min(k, AHEAD - 1) store groups behind DMA(k + 1): counting more than exist would let the wait pass early.
        __builtin_amdgcn_sched_barrier(0);           // (the wait is not hoisted into the MFMA sequence)
        ws_wait<(WS_AHEAD - 1) * DMA_PER_STRIP, ST_PER_STRIP, WS_AHEAD - 1>(k);
        if (!(abl & 2) || p.M < 0)               // (ablation: a never-taken branch keeps the arithmetic alive)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int64_t m = row0 + 16 * t + fr;
            half_t* orow = out + m * p.ldo + 80 * wave;
            // (plain stores also where the tiled kernels stream their output with `nt`: a non-temporal store is acknowledged late, and
            //  vmcnt counts it -- the next strip's wait then pulls the younger loads in with it: 252 -> 222 us / 337 -> 251 us at M = 655360)
            *reinterpret_cast<half8v*>(orow + 8 * fq) = o0[t];
            *reinterpret_cast<half8v*>(orow + 32 + 8 * fq) = o1[t];
            *reinterpret_cast<half4v*>(orow + 64 + 4 * fq) = o2[t];
            if constexpr (ROWSUM) {
                if (fq == 0) *reinterpret_cast<f32x2*>(p.rowsum + ((int64_t)wave * p.M + m) * 2) = f32x2{sum[t], sq[t]};
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                // (the zero-fill tail of the stream)
#endif
}

template <bool RES, bool ROWSUM>
int ws_launch2(const moca_gemm_params& p, hipStream_t st) {
    const int nstrips = p.M / WS_ROWS;
    int dev = 0, cus = 256;
    static int cached_cus = 0;
    if (!cached_cus) {
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cached_cus = prop.multiProcessorCount;
        else
            cached_cus = 256;
    }
    cus = cached_cus;
    const int grid = nstrips < cus ? nstrips : cus;
    constexpr int lds = 160 * 1024;                  // the whole LDS: one block per CU
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ws_kernel<RES, ROWSUM>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return MOCA_E_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_ws_kernel<RES, ROWSUM>), dim3(grid), dim3(256), lds, st, p, nstrips);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

}  // namespace

// which calls the kernel can run (validated, split-normalised parameters): a plain linear 320 -> 320 on whole 32-row strips, bias /
// residual / row sums only, every operand inside the 2 GiB a buffer descriptor's 32-bit offsets reach
bool moca_gemm_ws_ok(const moca_gemm_params& p) {
    if (p.a_mode != MOCA_A_LINEAR || p.a2 || p.splits != 1 || p.N != WS_C || p.K != WS_C || p.rowadd || p.up_phase) return false;
    if (p.flags & ~MOCA_EP_ROWSUM) return false;
    if (p.M % WS_ROWS || p.M < 8192) return false;
    if (p.lda % 8 || p.lda < WS_C || p.ldo % 8 || p.ldo < WS_C || p.ldw % 8 || p.ldw < WS_C || (p.residual && (p.ldr % 8 || p.ldr < WS_C))) return false;
    if (((int64_t)p.M * p.lda + 64) * 2 >= (1ll << 31) || (p.residual && ((int64_t)p.M * p.ldr + 64) * 2 >= (1ll << 31))) return false;
    if ((reinterpret_cast<uintptr_t>(p.a) | reinterpret_cast<uintptr_t>(p.w) | reinterpret_cast<uintptr_t>(p.out) | reinterpret_cast<uintptr_t>(p.residual)) & 15) return false;
    return true;
}

int moca_gemm_ws_launch(const moca_gemm_params& p, hipStream_t st) {
    if (!moca_gemm_ws_ok(p) || ((p.flags & MOCA_EP_ROWSUM) && !p.rowsum)) return MOCA_E_BADARG;
    const bool rs = p.flags & MOCA_EP_ROWSUM;
    if (p.residual) return rs ? ws_launch2<true, true>(p, st) : ws_launch2<true, false>(p, st);
    return rs ? ws_launch2<false, true>(p, st) : ws_launch2<false, false>(p, st);
}

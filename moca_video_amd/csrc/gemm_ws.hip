// Weight-stationary streaming linear for the HBM-bound 320 -> 320 projections (round 6).
//
// Replaces, for N == K == 320: `proj_in` / `proj_out` of SpatialTransformer / TemporalTransformer (attention.py:242,258,302,328) and
// `to_q`, `to_out[0]` of CrossAttention (attention.py:54-57) at the 320-channel level, with the bias / residual add
// (attention.py:217-219,278,373) and the LayerNorm statistics of the consumer (attention.py:199-201) in the epilogue.
//
// Why another kernel.  These launches move 3 x M x 640 B (A in, residual in, out) for 2 x M x 320 x 320 FLOP: 0.10 of the matrix pipe at
// the HBM rate.  On the staggered 160 x 320 tiling (gemm_w80s_kernel<0, 1>) a CU runs prologue -> main loop -> store loop once per
// 160-row tile and nothing streams during two of the three phases (profiles/r05_g4_phase_stamps.txt: store loop 45 %, main loop
// 31 %, prologue 17 % of a tile), and the 200 KB weight matrix is re-streamed L2 -> LDS for every 100 KB of A: 3.7-4.0 TB/s.  Here
//   * W never moves again: wave (c, h) of a block's 8 -- column group c = wave & 3 (on SIMD c), K half h = wave >> 2 -- keeps its 80
//     output columns x 160 k of W as MFMA fragments in 100 registers, loaded once per block;
//   * A AND the residual rows stream through one LDS ring by LDS-DMA, 32-row strips, 2 (with a residual: 40 KB each) or 5 strips ahead
//     of the one being computed, and no wave ever waits for a register load: vmcnt retires loads, LDS-DMA and stores together IN
//     ISSUE ORDER, one register load consumed per strip would pull the whole DMA queue in with it.  For the same reason the wait for
//     the next strip is counted exactly -- younger DMA pieces AND the store groups issued since -- and the output leaves with plain
//     stores (a non-temporal store is acknowledged late and holds the counter: 252 -> 222 us / 337 -> 251 us at M = 655360);
//   * a DMA instruction fetches ONE MFMA fragment (16 rows x 64 B of A: lane l takes row l % 16, chunk l / 16) or one accumulator-
//     shaped piece of the residual, so its 1 KB lands contiguously and is read back with ds_read_b128 at base + 16 lane: no swizzle,
//     no bank conflict, addresses are immediates;
//   * a wave multiplies BOTH row tiles of a strip by its K half (50 MFMAs), hands the partial sums of row tile 1 - h to its partner
//     through LDS (5 KB per wave, plain stores: hipcc pads the MFMA -> LDS-write hazard, an inline-asm ds_write does not and read
//     stale accumulators) and finishes row tile h from registers: W rows are assigned to MFMA rows in a permuted order so that a
//     lane's accumulators in two neighbouring tiles are 8 consecutive output columns (16-byte stores, 16 rows x 64 B per instruction);
//   * strips are dealt round-robin (block b: b, b + G, ...): the G blocks stream one contiguous window; a contiguous range per block
//     put all of them M / G rows apart -- the same HBM channels -- and ran 1.2-1.3 x slower;
//   * two s_barriers per strip (strip landed / partials exchanged).
// The first form had 4 waves (one per SIMD, all of K: 200 registers of W, 512-register budget): what a strip cost was the SUM of what
// its one wave issued -- DMA pieces, fragment reads, 100 MFMAs, epilogue -- 3-8 % behind this form at M = 655360, 5-10 % at M = 81920.
// Measured (profiles/r06_ab_gemm_ws.txt, same box, alternating, cold operands): M = 655360 +res 318 -> 247 us (5.1 TB/s of algorithmic
// bytes), +res +rowsum 321 -> 265, plain 228 -> 188; M = 81920 +res 46.5 -> 42.1, +res +rowsum 47.0 -> 43.2, plain 33.4 -> 37.2 (a block's
// 10 strips do not amortise its start-up: the dispatch takes it only with a residual or from M = 2^17 up).
// Algorithmic bytes per launch: M x 640 B x (2 + residual) + 200 KB x blocks of W (from L2).
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) char* lds_ptr;
[[maybe_unused]] constexpr unsigned WS_OOB = 0x80000000u;
[[maybe_unused]] constexpr int WS_C = 320;                       // N == K
[[maybe_unused]] constexpr int WS_ROWS = 32;                     // rows per strip (two MFMA row tiles)
[[maybe_unused]] constexpr int WS_KSTEPS = WS_C / 32;            // 10 k-steps of v_mfma_f32_16x16x32_f16
[[maybe_unused]] constexpr int WS_A_BYTES = WS_ROWS * WS_C * 2;  // 20 KiB: 20 fragments of 1 KiB, [row tile][k-step]
[[maybe_unused]] constexpr int WS_R_BYTES = WS_ROWS * WS_C * 2;  // 20 KiB: per wave 5 pieces of 1 KiB, [wave][piece]
// ring: 160 KiB either way -- 4 slots of (A + residual) = 40 KiB with a residual, 8 slots of 20 KiB without; SLOTS - 1 strips in flight
// ahead of the one being computed

// column (inside a wave's 80) held by MFMA row i of column tile j: tiles (0,1) and (2,3) interleave so that rows 4q..4q+3 of a pair
// are 8 consecutive columns; tile 4 is plain
__device__ __forceinline__ int ws_col(int j, int i) {
    if (j == 4) return 64 + i;
    return 32 * (j >> 1) + 8 * (i >> 2) + 4 * (j & 1) + (i & 3);
}

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

template <int V> struct ws_ic { static constexpr int value = V; };

// Strip -> block map.  The rows come in GROUPS of `U` strips (one group = the whole matrix; or the rows of one weight group, wgroup_rows;
// or of one GroupNorm statistics group, gstat_rows).  n_groups >= G: block b takes groups b, b + G, ... whole; else `bpg` = G / n_groups
// blocks share a group and deal its strips round-robin (part, part + bpg, ...).  Inside a group a block starts at a ROTATED position
// (13 g + 5 part): the blocks of different groups are a multiple of the group size apart in memory -- without the rotation all of them
// would sit on the same HBM channels at the same moment (the 1.2-1.3 x of a contiguous range per block, profiles/r06_ab_gemm_ws.txt).
template <bool RES, int EPI>
__global__ __launch_bounds__(512) void gemm_ws_kernel(const moca_gemm_params p, const int U, const int n_groups, const int bpg) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool ROWSUM = EPI == 1, GSTAT = EPI == 2;
    constexpr int SLOT = WS_A_BYTES + (RES ? WS_R_BYTES : 0), SLOTS = RES ? 3 : 6, AHEAD = SLOTS - 1;
    constexpr int XCH = SLOTS * SLOT;                // 40 KiB exchange area behind the ring: [wave][column tile][lane] f32x4
    constexpr int ST_PER_STRIP = ROWSUM ? 4 : 3;     // per wave: its row tile's (16 B, 16 B, 8 B per lane) (+ the row partial)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave & 3, h = wave >> 2;
    const int fr = lane & 15, fq = lane >> 4;
    const int G = gridDim.x;
    const bool many = n_groups >= G;
    const int part = many ? 0 : (int)blockIdx.x % bpg;
    const int g_first = many ? (int)blockIdx.x : (int)blockIdx.x / bpg, g_step = many ? G : n_groups;
    const int n_my = (U - part + bpg - 1) / bpg;                         // strips of a group this block handles
    if (g_first >= n_groups || n_my <= 0) return;

    // ---- DMA stream: the pieces of a strip = 20 fragments of A (f = 10 t + s) then, with a residual, its 20 pieces (r = 5 c + g): (t, pair) =
    //      16 rows x 64 B at byte column 160 c + 64 pair + 16 fq, and the tile-4 piece (lanes 0..31 row tile 0, 32..63 row tile 1, 16 B at byte
    //      column 160 c + 128 + 16 (fq & 1)); wave v issues pieces v, v + 8, v + 16 (, v + 24, v + 32): 5 each with a residual, else 3 (v < 4) or 2 ----
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, WS_OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(RES ? p.residual : p.a), 0, WS_OOB, 0x00020000);
    constexpr int NP = RES ? 5 : 3;
    unsigned rel[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int q = wave + 8 * i;
        if (q < 20) {
            const int t = q / WS_KSTEPS, sk = q - t * WS_KSTEPS;
            rel[i] = (unsigned)(((16 * t + fr) * p.lda) * 2 + 64 * sk + 16 * fq);
        } else {
            const int r = q - 20, c = r / 5, g = r - 5 * c;
            rel[i] = g < 4 ? (unsigned)(((16 * (g >> 1) + fr) * p.ldr) * 2 + 160 * c + 64 * (g & 1) + 16 * fq)
                           : (unsigned)(((16 * (fq >> 1) + fr) * p.ldr) * 2 + 160 * c + 128 + 16 * (fq & 1));
        }
    }
    half_t* const out = reinterpret_cast<half_t*>(p.out);
    const unsigned ax_rd = (unsigned)(((1 - h) * WS_KSTEPS + 5 * h) * 1024) + (unsigned)lane * 16;   // fragments of k-steps 5 h .. 5 h + 4 of row tile 1 - h
    const unsigned am_rd = (unsigned)((h * WS_KSTEPS + 5 * h) * 1024) + (unsigned)lane * 16;         // ... of row tile h
    const unsigned r_rd = (unsigned)(WS_A_BYTES + (5 * wc + 2 * h) * 1024) + (unsigned)lane * 16;
    const unsigned r4_rd = (unsigned)(WS_A_BYTES + (5 * wc + 4) * 1024) + (unsigned)((h * 32 + (fq >> 1) * 16 + fr) * 16 + (fq & 1) * 8);
    char* const x_wr = smem + XCH + wave * 5120 + lane * 16;
    const char* const x_rd = smem + XCH + (wc + 4 * (1 - h)) * 5120 + lane * 16;
    const bool per_group_w = p.wgroup_rows > 0;

    half8v wf[5][5];
    float bias[20];                                  // columns 80 wc + {8 fq .. +7, 32 + 8 fq .. +7, 64 + 4 fq .. +3}
    auto load_w = [&](int g) {                       // W: 80 columns x the K half h as 5 x 5 MFMA fragments; bias of the lane's 20 output columns
        const half_t* w = reinterpret_cast<const half_t*>(p.w) + (per_group_w ? (int64_t)g * p.wgroup_stride : 0);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const half_t* wr = w + (int64_t)(80 * wc + ws_col(j, fr)) * p.ldw + 160 * h + 8 * fq;
#pragma unroll
            for (int sk = 0; sk < 5; ++sk) wf[j][sk] = *reinterpret_cast<const half8v*>(wr + 32 * sk);
        }
#pragma unroll
        for (int c = 0; c < 20; ++c) bias[c] = 0.f;
        if (p.bias) {
            const float* bp = p.bias + (per_group_w ? (int64_t)g * p.N : 0) + 80 * wc;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp + 8 * fq), b1 = *reinterpret_cast<const f32x4*>(bp + 8 * fq + 4);
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(bp + 32 + 8 * fq), b3 = *reinterpret_cast<const f32x4*>(bp + 32 + 8 * fq + 4);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bp + 64 + 4 * fq);
#pragma unroll
            for (int c = 0; c < 4; ++c) { bias[c] = b0[c]; bias[4 + c] = b1[c]; bias[8 + c] = b2[c]; bias[12 + c] = b3[c]; bias[16 + c] = b4[c]; }
        }
    };
    if (!per_group_w) load_w(0);                     // (requested before the stream starts: vmcnt retires in order)

    for (int g = g_first; g < n_groups; g += g_step) {
        if (per_group_w) load_w(g);
        const int rot = (13 * g + 5 * part) % n_my;
        const int64_t grow0 = (int64_t)g * U * WS_ROWS;
        auto row_of = [&](int k) -> int64_t {        // first row of the k-th strip this block handles in group g
            int kk = k + rot;
            if (kk >= n_my) kk -= n_my;
            return grow0 + ((int64_t)part + (int64_t)kk * bpg) * WS_ROWS;
        };
        auto issue = [&](int k, int slot_i) {        // k-th strip -> ring slot slot_i (k >= n_my: zero fill, no traffic)
            const bool live = k < n_my;
            const int64_t row0 = live ? row_of(k) : 0;
            const unsigned a0 = (unsigned)(row0 * p.lda * 2), r0 = (unsigned)(row0 * p.ldr * 2);
            const lds_ptr slot = (lds_ptr)smem + slot_i * SLOT;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int q = wave + 8 * i;                              // (wave-uniform)
                if (RES || q < 20) {
                    if (q < 20) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, slot + q * 1024, 16, live ? rel[i] + a0 : WS_OOB, 0, 0, 0);
                    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_r, slot + q * 1024, 16, live ? rel[i] + r0 : WS_OOB, 0, 0, 0);
                }
            }
        };
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k0 = 0; k0 < AHEAD; ++k0) issue(k0, k0);
        [[maybe_unused]] float gs[20], gq[20];       // GSTAT: this lane's column sums / sums of squares over the group's strips (its row fr of each)
        if constexpr (GSTAT) {
#pragma unroll
            for (int c = 0; c < 20; ++c) { gs[c] = 0.f; gq[c] = 0.f; }
        }
        // strip 0 has landed (this wave's share; the younger strips may stay in flight)
        if (RES || h == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * NP) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * 2) : "memory");
        int cur = 0, nxt = AHEAD;                    // ring slots of strip k and of strip k + AHEAD (= the slot of strip k - 1)
        for (int k = 0; k < n_my; ++k) {
            __builtin_amdgcn_s_barrier();            // B1: strip k has landed everywhere; everybody is done with strip k - 1 and its exchange
            issue(k + AHEAD, nxt);
            const char* slot = smem + cur * SLOT;
            const int64_t m = row_of(k) + 16 * h + fr;
            // afx / accx: the row tile the PARTNER finishes (1 - h), afm / accm: this wave's (h) -- selected by address, not by branches
            half8v afx[5], afm[5];
#pragma unroll
            for (int sk = 0; sk < 5; ++sk) afx[sk] = *reinterpret_cast<const half8v*>(slot + ax_rd + sk * 1024);
            if constexpr (!GSTAT) {                  // (GSTAT keeps 40 accumulators more: its second five fragments are read behind the first MFMAs)
#pragma unroll
                for (int sk = 0; sk < 5; ++sk) afm[sk] = *reinterpret_cast<const half8v*>(slot + am_rd + sk * 1024);
            }
            half8v r0v, r1v;
            half4v r2v;
            if constexpr (RES) {
                r0v = *reinterpret_cast<const half8v*>(slot + r_rd);
                r1v = *reinterpret_cast<const half8v*>(slot + r_rd + 1024);
                r2v = *reinterpret_cast<const half4v*>(slot + r4_rd);
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x4 accx[5], accm[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) { accx[j] = f32x4{0.f, 0.f, 0.f, 0.f}; accm[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            // the partner's row tile first: its partial sums leave as early as possible
#pragma unroll
            for (int sk = 0; sk < 5; ++sk)
#pragma unroll
                for (int j = 0; j < 5; ++j) accx[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j][sk], afx[sk], accx[j], 0, 0, 0);
            if constexpr (GSTAT) {
#pragma unroll
                for (int sk = 0; sk < 5; ++sk) afm[sk] = *reinterpret_cast<const half8v*>(slot + am_rd + sk * 1024);
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) *reinterpret_cast<f32x4*>(x_wr + j * 1024) = accx[j];      // (plain stores: hipcc pads the MFMA -> LDS-write hazard)
#pragma unroll
            for (int sk = 0; sk < 5; ++sk)
#pragma unroll
                for (int j = 0; j < 5; ++j) accm[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j][sk], afm[sk], accm[j], 0, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // (the exchange writes have left)
            __builtin_amdgcn_s_barrier();            // B2: every partial is in the exchange area
            f32x4 mine[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) mine[j] = accm[j] + *reinterpret_cast<const f32x4*>(x_rd + j * 1024);
            // ---- epilogue of this wave's 16 rows: lane = row fr, columns 80 wc + {8 fq .. +7, 32 + 8 fq .. +7, 64 + 4 fq .. +3} ----
            half8v o0, o1;
            half4v o2;
            float sum = 0.f, sq = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float v0 = mine[c >> 2][c & 3] + bias[c], v1 = mine[2 + (c >> 2)][c & 3] + bias[8 + c];
                if constexpr (RES) { v0 += (float)r0v[c]; v1 += (float)r1v[c]; }
                o0[c] = (half_t)v0; o1[c] = (half_t)v1;
                if constexpr (ROWSUM) {
                    const float h0 = (float)o0[c], h1 = (float)o1[c];
                    sum += h0 + h1; sq += h0 * h0 + h1 * h1;
                }
                if constexpr (GSTAT) { gs[c] += v0; gq[c] += v0 * v0; gs[8 + c] += v1; gq[8 + c] += v1 * v1; }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float v = mine[4][c] + bias[16 + c];
                if constexpr (RES) v += (float)r2v[c];
                o2[c] = (half_t)v;
                if constexpr (ROWSUM) { const float hh = (float)o2[c]; sum += hh; sq += hh * hh; }
                if constexpr (GSTAT) { gs[16 + c] += v; gq[16 + c] += v * v; }
            }
            if constexpr (ROWSUM) {                  // partial `wc` of 4: (sum, sum of squares) over the 80 stored columns of row m
                sum += __shfl_xor(sum, 16, 64); sq += __shfl_xor(sq, 16, 64);
                sum += __shfl_xor(sum, 32, 64); sq += __shfl_xor(sq, 32, 64);
            }
            // strip k + 1 has landed (this wave's share), counted exactly: AHEAD - 1 younger DMA groups and the store groups issued since
            // (vmcnt retires loads, LDS-DMA and stores together in issue order; strip k has min(k, AHEAD - 1) store groups behind DMA(k + 1))
            __builtin_amdgcn_sched_barrier(0);
            if (RES || h == 0) ws_wait<(AHEAD - 1) * NP, ST_PER_STRIP, AHEAD - 1>(k);
            else ws_wait<(AHEAD - 1) * 2, ST_PER_STRIP, AHEAD - 1>(k);
            half_t* orow = out + m * p.ldo + 80 * wc;
            *reinterpret_cast<half8v*>(orow + 8 * fq) = o0;
            *reinterpret_cast<half8v*>(orow + 32 + 8 * fq) = o1;
            *reinterpret_cast<half4v*>(orow + 64 + 4 * fq) = o2;
            if constexpr (ROWSUM) {
                if (fq == 0) *reinterpret_cast<f32x2*>(p.rowsum + ((int64_t)wc * p.M + m) * 2) = f32x2{sum, sq};
            }
            cur = cur + 1 == SLOTS ? 0 : cur + 1;
            nxt = nxt + 1 == SLOTS ? 0 : nxt + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // (the zero-fill tail of the stream: the ring is free)
        if constexpr (GSTAT) {
            // GroupNorm statistics of the consumer (MOCA_EP_GSTAT; the group IS the statistics group): the lane sums of the whole group are
            // reduced over the 16 rows of the tile layout once -- four DPP steps (xor 1, xor 2, half mirror, mirror) -- staged in the (free)
            // ring, then one lane per (channel group touched by this wave's 80 columns, sum | sum of squares) adds its columns and issues ONE
            // fixed-point atomic (common.h).  Values before the fp16 rounding, as the tiled kernels count them.
            auto dpp_add = [](float v, auto ctrl_tag) {
                constexpr int ctrl = decltype(ctrl_tag)::value;
                return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false));
            };
#pragma unroll
            for (int c = 0; c < 20; ++c) {
                gs[c] = dpp_add(gs[c], ws_ic<0xB1>{}); gq[c] = dpp_add(gq[c], ws_ic<0xB1>{});          // quad_perm [1,0,3,2]
                gs[c] = dpp_add(gs[c], ws_ic<0x4E>{}); gq[c] = dpp_add(gq[c], ws_ic<0x4E>{});          // quad_perm [2,3,0,1]
                gs[c] = dpp_add(gs[c], ws_ic<0x141>{}); gq[c] = dpp_add(gq[c], ws_ic<0x141>{});        // row_half_mirror
                gs[c] = dpp_add(gs[c], ws_ic<0x140>{}); gq[c] = dpp_add(gq[c], ws_ic<0x140>{});        // row_mirror
            }
            __builtin_amdgcn_s_barrier();            // (every wave has left the loop: no exchange / ring read is pending)
            float* scr = reinterpret_cast<float*>(smem + wave * 1024);                   // [2][80] floats of this wave
            if (fr == 0) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    scr[8 * fq + c] = gs[c]; scr[32 + 8 * fq + c] = gs[8 + c];
                    scr[80 + 8 * fq + c] = gq[c]; scr[80 + 32 + 8 * fq + c] = gq[8 + c];
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) { scr[64 + 4 * fq + c] = gs[16 + c]; scr[80 + 64 + 4 * fq + c] = gq[16 + c]; }
            }
            if (lane < 32) {
                const int comp = lane & 1, gi = lane >> 1;
                const int cpg = p.gstat_cpg > 0 ? p.gstat_cpg : WS_C / 32;
                const int c0 = p.gstat_coff + 80 * wc;                                 // consumer channel of this wave's column 0
                const int cg = c0 / cpg + gi;
                const int lo = max(cg * cpg - c0, 0), hi = min((cg + 1) * cpg - c0, 80);
                if (lo < hi) {
                    float a = 0.f;
                    for (int c = lo; c < hi; ++c) a += scr[comp * 80 + c];
                    moca_gstat_add(p.gstat + ((int64_t)g * 32 + cg) * 2 + comp, comp, a);
                }
            }
        }
        __builtin_amdgcn_s_barrier();                // (the next group's stream may overwrite the ring)
    }
#endif
}

template <bool RES, int EPI>
int ws_launch2(const moca_gemm_params& p, hipStream_t st) {
    static int cached_cus = 0;
    if (!cached_cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cached_cus = prop.multiProcessorCount;
        else
            cached_cus = 256;
    }
    // rows per group: a weight group (wgroup_rows), a statistics group (gstat_rows), or the whole matrix
    const int R = p.wgroup_rows > 0 ? p.wgroup_rows : ((p.flags & MOCA_EP_GSTAT) ? p.gstat_rows : p.M);
    const int U = R / WS_ROWS, n_groups = p.M / R;
    int grid, bpg;
    if (n_groups >= cached_cus) { grid = cached_cus; bpg = 1; }
    else {
        bpg = cached_cus / n_groups;
        if (bpg > U) bpg = U;
        grid = bpg * n_groups;
    }
    constexpr int lds = 160 * 1024;                  // the whole LDS: one block per CU
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ws_kernel<RES, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return MOCA_E_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_ws_kernel<RES, EPI>), dim3(grid), dim3(512), lds, st, p, U, n_groups, bpg);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

}  // namespace

// which calls the kernel can run (validated, split-normalised parameters): a plain linear 320 -> 320 on whole 32-row strips; bias /
// residual / row sums OR finished GroupNorm statistics; per-row-group weights; every operand inside the 2 GiB a buffer descriptor's
// 32-bit offsets reach
bool moca_gemm_ws_ok(const moca_gemm_params& p) {
    if (p.a_mode != MOCA_A_LINEAR || p.a2 || p.splits != 1 || p.N != WS_C || p.K != WS_C || p.rowadd || p.up_phase) return false;
    if (p.flags & ~(MOCA_EP_ROWSUM | MOCA_EP_GSTAT)) return false;
    if ((p.flags & MOCA_EP_ROWSUM) && (p.flags & MOCA_EP_GSTAT)) return false;
    if (p.M % WS_ROWS || p.M < 8192) return false;
    if (p.flags & MOCA_EP_GSTAT) {                     // the statistics group = the strip group; <= 16 channel groups per 80 columns
        const int cpg = p.gstat_cpg > 0 ? p.gstat_cpg : WS_C / 32;
        if (p.gstat_rows <= 0 || p.gstat_rows % WS_ROWS || p.M % p.gstat_rows || cpg < 6 || p.gstat_coff < 0) return false;
    }
    if (p.wgroup_rows) {
        if (p.wgroup_rows < 0 || p.wgroup_rows % WS_ROWS || p.M % p.wgroup_rows || p.wgroup_stride < p.N * p.ldw) return false;
        if ((p.flags & MOCA_EP_GSTAT) && p.gstat_rows != p.wgroup_rows) return false;
        if ((int64_t)(p.M / p.wgroup_rows) * p.wgroup_stride * 2 >= (1ll << 31)) return false;
    }
    if (p.lda % 8 || p.lda < WS_C || p.ldo % 8 || p.ldo < WS_C || p.ldw % 8 || p.ldw < WS_C || (p.residual && (p.ldr % 8 || p.ldr < WS_C))) return false;
    if (((int64_t)p.M * p.lda + 64) * 2 >= (1ll << 31) || (p.residual && ((int64_t)p.M * p.ldr + 64) * 2 >= (1ll << 31))) return false;
    if ((reinterpret_cast<uintptr_t>(p.a) | reinterpret_cast<uintptr_t>(p.w) | reinterpret_cast<uintptr_t>(p.out) | reinterpret_cast<uintptr_t>(p.residual)) & 15) return false;
    return true;
}

int moca_gemm_ws_launch(const moca_gemm_params& p, hipStream_t st) {
    if (!moca_gemm_ws_ok(p) || ((p.flags & MOCA_EP_ROWSUM) && !p.rowsum) || ((p.flags & MOCA_EP_GSTAT) && !p.gstat)) return MOCA_E_BADARG;
    const int epi = (p.flags & MOCA_EP_ROWSUM) ? 1 : ((p.flags & MOCA_EP_GSTAT) ? 2 : 0);
    if (p.residual) return epi == 1 ? ws_launch2<true, 1>(p, st) : (epi == 2 ? ws_launch2<true, 2>(p, st) : ws_launch2<true, 0>(p, st));
    return epi == 1 ? ws_launch2<false, 1>(p, st) : (epi == 2 ? ws_launch2<false, 2>(p, st) : ws_launch2<false, 0>(p, st));
}

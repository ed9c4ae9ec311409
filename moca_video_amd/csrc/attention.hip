// Attention kernels for gfx950, head dim 64, fp16 I/O, fp32 softmax.
//
// moca_attention_f16: flash-style softmax(QK^T*scale)V for the spatial self / cross
//   attention (N = H*W up to 2560 queries; 77/154 context keys).  256 threads = 4
//   wavefronts x 32 query rows; K/V tiles of 64 keys double-buffered in LDS, both
//   row-major as they come from HBM (16-byte coalesced staging): K XOR-swizzled for the
//   ds_read_b128 fragment reads, V XOR-swizzled for ds_read_b64_tr_b16, the CDNA4
//   transposing LDS read that delivers the V^T operand of P.V without any data movement.
//   The score tile is computed "swapped" (S^T = K.Q^T with v_mfma_f32_32x32x16_f16) so
//   every lane owns ONE query column: the online-softmax row reduction is 31 in-lane
//   max/adds plus one lane<->lane+32 exchange, and the S^T accumulator registers are
//   directly the B operand of O^T += V^T.P^T (k order of the accumulator-as-operand
//   trick: key = 16s + 8(j>>2) + 4h + (j&3), matched by two 4-key transposed reads).
//   One barrier per key tile; the accumulator rescale is skipped (wave-uniform branch)
//   when no lane's running maximum moved.
//
// moca_temporal_attention_f16: attention over the frame axis (T <= 16) per
//   (pixel, head): one wavefront per problem with v_mfma_f32_16x16x32_f16 (QK^T) and
//   v_mfma_f32_16x16x16_f16 (PV); gathers the T frames of a pixel straight from the
//   channels-last token matrix, so the reference's (b t) c h w <-> (b h w) t c
//   reshuffles (attention.py:335-338,367) never touch memory.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int D = 64;        // head dim
constexpr int KT = 64;       // keys per LDS tile
constexpr int QB = 128;      // queries per block
constexpr int ROWB = 128;    // bytes per LDS row (64 halves)

template <int V> struct int_c { static constexpr int value = V; };
typedef short short4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char* lds_c_ptr;
typedef __attribute__((address_space(3))) short4v* lds_s4_ptr;

// ds_read_b64_tr_b16: a 16-lane group reads a 4-row x 16-column block of 16-bit elements and receives it
// column-major (lane i gets column i of the 4 rows).  Lane i = 4q+p supplies the address of row q, columns 4p..4p+3.
__device__ __forceinline__ half4v tr_read(const char* addr) {
    const short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)addr);
    return __builtin_bit_cast(half4v, v);
}

// XCD-aware block order: the hardware deals consecutive workgroup ids round-robin to the 8 XCDs (each with its own L2).  With
// the natural order the 20 query blocks of one (frame, head) land on all 8 XCDs and every L2 fetches that head's K and V from
// the fabric (rocprofv3 FETCH_SIZE: 2 GB per 2560-token launch, ~6 TB/s during the kernel).  Here all query blocks of a
// (frame, head) pair go to XCD (pair % 8), consecutive in time, so K/V cross the fabric once.
__device__ __forceinline__ void attn_block_coords(int& bx, int& by) {
    const int nx = gridDim.x, ny = gridDim.y;
    if (ny % 8 == 0) {
        const int b = blockIdx.y * nx + blockIdx.x;       // dispatch order
        const int xcd = b & 7, slot = b >> 3;
        by = (slot / nx) * 8 + xcd;
        bx = slot % nx;
    } else {
        bx = blockIdx.x; by = blockIdx.y;
    }
}

template <bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attention_kernel(
    const half_t* __restrict__ q, const half_t* __restrict__ k, const half_t* __restrict__ v, half_t* __restrict__ out,
    int heads, int Nq, int Nk, int ldq, int ldk, int ldv, int ldo, int kv_div, float scale_log2e) {
    // double-buffered K / V tiles: K row-major [key][d] swizzled for ds_read_b128 (chunk ^ ((row>>1)&7)),
    // V row-major [key][d] swizzled for the transposed reads (chunk ^ (((row>>1)&1)<<2))
    __shared__ __attribute__((aligned(16))) char sK[2][KT * ROWB];
    __shared__ __attribute__((aligned(16))) char sV[2][KT * ROWB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx, by;
    attn_block_coords(bx, by);
    const int bq = by / heads, head = by % heads;
    const int bkv = bq / kv_div;
    const int q0 = bx * QB + wave * 32;
    const int fr = lane & 31, fh = lane >> 5;

    const half_t* qb = q + (int64_t)bq * Nq * ldq + head * D;
    const half_t* kb = k + (int64_t)bkv * Nk * ldk + head * D;
    const half_t* vb = v + (int64_t)bkv * Nk * ldv + head * D;

    // Q fragments (B operand of S^T = K.Q^T): lane (q=fr, h=fh) holds Q[q][16ks + 8h + j]
    half8v qf[4];
    const int qrow = q0 + fr;
    const bool q_ok = qrow < Nq;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        half8v t = {0, 0, 0, 0, 0, 0, 0, 0};
        if (q_ok) t = *reinterpret_cast<const half8v*>(qb + (int64_t)qrow * ldq + ks * 16 + fh * 8);
        qf[ks] = t;
    }

    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // staging coordinates: thread -> rows (tid>>3) + 32 i, chunk tid&7
    const int cc = tid & 7, r0 = tid >> 3;
    half8v rk[2], rv[2];
    auto load_kv = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = kt * KT + r0 + 32 * i;
            half8v a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
            if (key < Nk) {
                a = *reinterpret_cast<const half8v*>(kb + (int64_t)key * ldk + cc * 8);
                b = *reinterpret_cast<const half8v*>(vb + (int64_t)key * ldv + cc * 8);
            }
            rk[i] = a; rv[i] = b;
        }
    };
    auto store_kv = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = r0 + 32 * i;
            *reinterpret_cast<half8v*>(sK[buf] + row * ROWB + ((cc ^ ((row >> 1) & 7)) << 4)) = rk[i];
            *reinterpret_cast<half8v*>(sV[buf] + row * ROWB + ((cc ^ (((row >> 1) & 1) << 2)) << 4)) = rv[i];
        }
    };
    // per-lane constants of the transposed V reads: lane i = lane&15 = 4q+p of its 16-lane group
    const int tq = (lane & 15) >> 2, tp = lane & 3;
    const int tcol16 = (lane >> 4) & 1;          // which 16-column half of the 32-wide d tile
    const int tkey4 = 4 * fh;                     // lane half h selects keys +4

    const int nkt = (Nk + KT - 1) / KT;
    load_kv(0);
    store_kv(0);
    if (nkt > 1) load_kv(1);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        // stage the next tile into the other buffer (its last readers passed the previous barrier),
        // then fetch the tile after it into registers: both overlap the MFMA/softmax work below
        if (kt + 1 < nkt) {
            store_kv(cur ^ 1);
            if (kt + 2 < nkt) load_kv(kt + 2);
        }
        const char* kbuf = sK[cur];
        const char* vbuf = sV[cur];

        // ---- S^T = K . Q^T ----
        f32x16 s[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[sub][r] = 0.f;
            const int row = sub * 32 + fr;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int ch = ks * 2 + fh;
                const half8v kf = *reinterpret_cast<const half8v*>(kbuf + row * ROWB + ((ch ^ ((row >> 1) & 7)) << 4));
                s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[sub], 0, 0, 0);
            }
        }
        // mask keys beyond Nk (last tile only); CAUSAL: also keys after this lane's query (text tower of the CLIP encoder)
        if (CAUSAL || (kt + 1) * KT > Nk) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * KT + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    if (key >= Nk || (CAUSAL && key > qrow)) s[sub][r] = -INFINITY;
                }
        }
        // ---- online softmax (this lane = one query column; partner lane^32 has the other keys) ----
        float tmax = s[0][0];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[sub][r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float mb = m_new * scale_log2e;
        float psum = 0.f;
        half8v pf[2][2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(s[sub][r] * scale_log2e - mb);
                psum += pv;
                pf[sub][r >> 3][r & 7] = (half_t)pv;
            }
        // rescale only when some lane's running max moved (wave-uniform branch; alpha == 1 otherwise)
        if (__any(m_new > m_run)) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
        l_run += psum;
        m_run = m_new;
        // ---- O^T += V^T . P^T ; V^T fragments by transposed LDS reads of the row-major V tile ----
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                const int krow0 = sub * 32 + ss * 16 + tkey4 + tq;     // row of the first 4-key block; second is +8
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int dcol = dt * 32 + tcol16 * 16 + 4 * tp;     // first of this lane's 4 address columns
                    const int ch = dcol >> 3, within = (dcol & 7) * 2;
                    const int ra = krow0, rb = krow0 + 8;
                    const half4v lo = tr_read(vbuf + ra * ROWB + ((ch ^ (((ra >> 1) & 1) << 2)) << 4) + within);
                    const half4v hi = tr_read(vbuf + rb * ROWB + ((ch ^ (((rb >> 1) & 1) << 2)) << 4) + within);
                    half8v vf;
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[sub][ss], o[dt], 0, 0, 0);
                }
            }
        __syncthreads();   // everyone is done with buffers [cur]; the other buffers' stores are visible
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_ok) {
        half_t* ob = out + ((int64_t)bq * Nq + qrow) * ldo + head * D;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                half4v h4;
#pragma unroll
                for (int j = 0; j < 4; ++j) h4[j] = (half_t)(o[dt][g * 4 + j] * inv);
                *reinterpret_cast<half4v*>(ob + dt * 32 + 8 * g + 4 * fh) = h4;
            }
    }
}

// ---- cross-attention against a SHORT context (N_k <= 96: the 77 CLIP tokens): ONE key tile of 96 (three 32-key sub-tiles; the
// general kernel runs two 64-key tiles of which the second is 80 % padding, with an online-softmax step and a barrier in
// between), plain softmax (no running maximum), and K / V staged ONCE per block for `q_iters` consecutive 128-query groups
// (the general kernel re-stages them per 128 queries: 20 times per (frame, head) at 2560 tokens).  Fragment maps and LDS images
// are those of attention_kernel.
constexpr int KS96 = 96;
__global__ __launch_bounds__(256, 2) void attention_short_kernel(
    const half_t* __restrict__ q, const half_t* __restrict__ k, const half_t* __restrict__ v, half_t* __restrict__ out,
    int heads, int Nq, int Nk, int ldq, int ldk, int ldv, int ldo, int kv_div, float scale_log2e, int q_iters) {
    __shared__ __attribute__((aligned(16))) char sK[KS96 * ROWB];
    __shared__ __attribute__((aligned(16))) char sV[KS96 * ROWB];
    // Q rows in / O rows out pass through a per-wave LDS tile (32 rows x 128 B, chunk ^ ((row >> 1) & 7)): the global accesses are
    // whole 128-byte rows (8 lanes x 16 B) instead of one 16-byte / 8-byte piece of 32 different rows per instruction.  Each wave
    // touches only its own 32 rows, so no block barrier is needed inside the query loop.
    __shared__ __attribute__((aligned(16))) char sQ[QB * ROWB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx, by;
    attn_block_coords(bx, by);
    const int bq = by / heads, head = by % heads;
    const int bkv = bq / kv_div;
    const int fr = lane & 31, fh = lane >> 5;
    const half_t* qb = q + (int64_t)bq * Nq * ldq + head * D;
    const half_t* kb = k + (int64_t)bkv * Nk * ldk + head * D;
    const half_t* vb = v + (int64_t)bkv * Nk * ldv + head * D;
    {   // stage K and V (rows >= Nk are zero; they are masked below)
        const int cc = tid & 7, r0 = tid >> 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int row = r0 + 32 * i;
            half8v a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
            if (row < Nk) {
                a = *reinterpret_cast<const half8v*>(kb + (int64_t)row * ldk + cc * 8);
                b = *reinterpret_cast<const half8v*>(vb + (int64_t)row * ldv + cc * 8);
            }
            *reinterpret_cast<half8v*>(sK + row * ROWB + ((cc ^ ((row >> 1) & 7)) << 4)) = a;
            *reinterpret_cast<half8v*>(sV + row * ROWB + ((cc ^ (((row >> 1) & 1) << 2)) << 4)) = b;
        }
    }
    __syncthreads();
    const int tq = (lane & 15) >> 2, tp = lane & 3;
    const int tcol16 = (lane >> 4) & 1;
    const int tkey4 = 4 * fh;
    char* wq = sQ + wave * 32 * ROWB;
    half8v nq[4];                                          // the NEXT query group's rows travel while the current one is computed
    auto load_q = [&](int q0n) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {                      // 32 rows x 8 chunks = 256 chunks, 4 per lane, row-contiguous
            const int c = lane + 64 * i, row = c >> 3, ch = c & 7;
            half8v t = {0, 0, 0, 0, 0, 0, 0, 0};
            if (q0n + row < Nq) t = *reinterpret_cast<const half8v*>(qb + (int64_t)(q0n + row) * ldq + ch * 8);
            nq[i] = t;
        }
    };
    load_q(bx * q_iters * QB + wave * 32);
    for (int it = 0; it < q_iters; ++it) {
        const int q0 = (bx * q_iters + it) * QB + wave * 32;
        if (q0 >= Nq) break;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i, row = c >> 3, ch = c & 7;
            *reinterpret_cast<half8v*>(wq + row * ROWB + ((ch ^ ((row >> 1) & 7)) << 4)) = nq[i];
        }
        if (it + 1 < q_iters) load_q(q0 + QB);
        half8v qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[ks] = *reinterpret_cast<const half8v*>(wq + fr * ROWB + (((ks * 2 + fh) ^ ((fr >> 1) & 7)) << 4));
        f32x16 s[3];
#pragma unroll
        for (int sub = 0; sub < 3; ++sub) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[sub][r] = 0.f;
            const int row = sub * 32 + fr;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int ch = ks * 2 + fh;
                const half8v kf = *reinterpret_cast<const half8v*>(sK + row * ROWB + ((ch ^ ((row >> 1) & 7)) << 4));
                s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[sub], 0, 0, 0);
            }
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int sub = 0; sub < 3; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (key >= Nk) s[sub][r] = -INFINITY;
                tmax = fmaxf(tmax, s[sub][r]);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mb = tmax * scale_log2e;
        float psum = 0.f;
        half8v pf[3][2];
#pragma unroll
        for (int sub = 0; sub < 3; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(s[sub][r] * scale_log2e - mb);
                psum += pv;
                pf[sub][r >> 3][r & 7] = (half_t)pv;
            }
        f32x16 o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
        for (int sub = 0; sub < 3; ++sub)
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                const int krow0 = sub * 32 + ss * 16 + tkey4 + tq;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int dcol = dt * 32 + tcol16 * 16 + 4 * tp;
                    const int ch = dcol >> 3, within = (dcol & 7) * 2;
                    const int ra = krow0, rb = krow0 + 8;
                    const half4v lo = tr_read(sV + ra * ROWB + ((ch ^ (((ra >> 1) & 1) << 2)) << 4) + within);
                    const half4v hi = tr_read(sV + rb * ROWB + ((ch ^ (((rb >> 1) & 1) << 2)) << 4) + within);
                    const half8v vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[sub][ss], o[dt], 0, 0, 0);
                }
            }
        const float inv = 1.0f / (psum + __shfl_xor(psum, 32, 64));
        // O^T[d = 32 dt + 8 g + 4 fh + j][query fr] -> row fr of the wave's LDS tile (the Q fragments are in registers), then whole rows out
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                half4v h4;
#pragma unroll
                for (int j = 0; j < 4; ++j) h4[j] = (half_t)(o[dt][g * 4 + j] * inv);
                const int d0 = dt * 32 + 8 * g + 4 * fh;
                *reinterpret_cast<half4v*>(wq + fr * ROWB + (((d0 >> 3) ^ ((fr >> 1) & 7)) << 4) + (d0 & 7) * 2) = h4;
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i, row = c >> 3, ch = c & 7;
            if (q0 + row < Nq)
                *reinterpret_cast<half8v*>(out + ((int64_t)bq * Nq + q0 + row) * ldo + head * D + ch * 8) =
                    *reinterpret_cast<const half8v*>(wq + row * ROWB + ((ch ^ ((row >> 1) & 7)) << 4));
        }
    }
}

// ---- "v4": the VALU-lean flash attention for long key sequences ---------------------------------------------------------
// rocprofv3 counters of attention_kernel at N = 2560 (tools/pmc_attn.sh): 222 VALU instructions per wave per 64-key tile,
// the VALU active in 61 % of all SIMD cycles, the matrix pipe in 30 % -- at head dim 64 the kernel is VALU-bound, so v4 removes
// vector instructions instead of re-arranging them.  Per score element the first-generation kernel spends
// fma (scale, - max) + exp + add (row sum) + 0.5 max3 + ~1.5 conversions + its share of zero-initialising the score
// accumulators and of LDS / global address arithmetic.  Here:
//   * Q is multiplied by scale*log2(e) once, when it is loaded (one fp16 rounding of Q more), and the running reference
//     maximum enters through the ACCUMULATOR INPUT of the first score MFMA (a 16-register operand holding -m_ref, rebuilt
//     only when m_ref moves): S'' = K.Q'^T - m_ref comes out of the matrix pipe ready for exp2 -- no fma, no zero fill;
//   * the reference maximum is lazy (cdna_hip_programming.md T13): it is moved (O, l rescaled) only when a tile maximum
//     exceeds it by more than THR = 8, i.e. P <= 2^8 in fp16, whose relative precision does not depend on magnitude; the
//     first tile always sets it;
//   * the row sums are v_dot2_f32_f16 of the PACKED P against (1, 1): 16 instructions per tile instead of 32 adds (a variant
//     that takes them from the matrix pipe -- l^T += 1^T.P^T, one more MFMA per 16 keys -- is 5 % slower: the matrix pipe is
//     the longer pole once the VALU work is cut; -DMOCA_ATTN_SUM_MFMA);
//   * P is packed with v_cvt_pk_f16_f32 only; LDS fragment addresses are per-lane constants + immediates (key loop
//     unrolled over the two buffers); the staging pointers advance by one 64-bit add per tile.
// What is left per element: 0.5 max3 + exp + 0.5 cvt_pk + 0.5 dot2.  121 VGPRs: four waves per SIMD.  Tiles, LDS images and fragment maps are those of attention_kernel.
[[maybe_unused]] constexpr float LAZY_THR = 8.0f;
[[maybe_unused]] constexpr float REF0_BAND = 6.0f;     // first-tile maxima inside [-6, 6] (in log2 units) leave the reference at 0
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// NW wavefronts (32 queries each) share every K/V tile: 4 (128 queries per block), or 8 (256 queries: half the LDS-DMA and L2 -> LDS
// traffic per query, one piece of K and one of V per wave and tile; the barrier spans 8 waves)
template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : 2) void attention_v4_kernel(
    const half_t* __restrict__ q, const half_t* __restrict__ k, const half_t* __restrict__ v, half_t* __restrict__ out,
    int heads, int Nq, int Nk, int ldq, int ldk, int ldv, int ldo, int kv_div, float scale_log2e) {
#if defined(__HIP_DEVICE_COMPILE__)   // (__amdgpu_buffer_rsrc_t is a device-only type; the host pass only needs the launch stub)
    __shared__ __attribute__((aligned(16))) char smem[4 * KT * ROWB];      // sK[0], sK[1], sV[0], sV[1]
    constexpr int TILE = KT * ROWB;                                       // 8 KiB
    char* const sK = smem;
    char* const sV = smem + 2 * TILE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx, by;
    attn_block_coords(bx, by);
    const int bq = by / heads, head = by % heads;
    const int bkv = bq / kv_div;
    const int q0 = bx * (32 * NW) + wave * 32;
    const int fr = lane & 31, fh = lane >> 5;

    const half_t* qb = q + (int64_t)bq * Nq * ldq + head * D;
    // Q' = Q * scale * log2(e), B operand of S^T = K.Q'^T: lane (q = fr, h = fh) holds Q'[q][16 ks + 8 h + j]
    half8v qf[4];
    const int qrow = q0 + fr;
    const bool q_ok = qrow < Nq;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        half8v t = {0, 0, 0, 0, 0, 0, 0, 0};
        if (q_ok) t = *reinterpret_cast<const half8v*>(qb + (int64_t)qrow * ldq + ks * 16 + fh * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = (half_t)((float)t[j] * scale_log2e);
        qf[ks] = t;
    }

    f32x16 o[2], osum, negm;          // O^T accumulators, row-sum accumulator (all rows equal), -m_ref replicated
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; osum[r] = 0.f; negm[r] = 0.f; }
    float m_ref = 0.f;
    bool ref_zero = true;      // (wave-uniform) no query of this wave has moved its reference off 0
    float lsum = 0.f;          // this lane's share of the row sum (its 32 of the 64 keys of every tile)
    half8v ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (half_t)1.0f;

    // staging by LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers, no ds_write): a 1 KiB piece = 8 key rows x 128 B,
    // lane -> row (lane >> 3), PHYSICAL chunk lane & 7; the swizzles of the two LDS images go on the per-lane SOURCE address
    // (logical chunk = physical ^ swizzle(row)).  Wave w moves pieces w and w + 4 of K and of V (rows 8w.., 32 + 8w..): 4 DMA
    // instructions per wave per key tile.  Rows >= Nk lie beyond num_records of the descriptors: the range check writes zeros.
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int lr = lane >> 3, pc = lane & 7;
    const int srow = wv * 8 + lr;                                          // + 32 for the second piece (same swizzles)
    const unsigned k_voff = (unsigned)(srow * ldk * 2 + ((pc ^ ((srow >> 1) & 7)) << 4));
    const unsigned v_voff = (unsigned)(srow * ldv * 2 + ((pc ^ (((srow >> 1) & 1) << 2)) << 4));
    const half_t* kbase = k + (int64_t)bkv * Nk * ldk + head * D;
    const half_t* vbase = v + (int64_t)bkv * Nk * ldv + head * D;
    const __amdgpu_buffer_rsrc_t rsrc_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(kbase), 0, (unsigned)(Nk * ldk * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(vbase), 0, (unsigned)(Nk * ldv * 2), 0x00020000);
    auto dma_k = [&](int kt, int buf) {
        const lds_c_ptr dk = (lds_c_ptr)sK + buf * TILE + wv * 1024;
        const unsigned ks0 = (unsigned)(kt * KT * ldk * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_k, dk, 16, k_voff, ks0, 0, 0);
        if constexpr (NW == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_k, dk + 4096, 16, k_voff, ks0 + (unsigned)(32 * ldk * 2), 0, 0);
    };
    auto dma_v = [&](int kt, int buf) {
        const lds_c_ptr dv = (lds_c_ptr)sV + buf * TILE + wv * 1024;
        const unsigned vs0 = (unsigned)(kt * KT * ldv * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_v, dv, 16, v_voff, vs0, 0, 0);
        if constexpr (NW == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_v, dv + 4096, 16, v_voff, vs0 + (unsigned)(32 * ldv * 2), 0, 0);
    };
    // fragment read offsets inside a tile: per-lane constants; (sub, ss, buffer) only add immediates
    int k_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) k_off[ks] = fr * ROWB + (((ks * 2 + fh) ^ ((fr >> 1) & 7)) << 4);       // + sub * 32 * ROWB
    const int tq = (lane & 15) >> 2, tp = lane & 3, tcol16 = (lane >> 4) & 1;
    int v_off[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const int dcol = dt * 32 + tcol16 * 16 + 4 * tp;
        const int row = 4 * fh + tq;                                      // + sub*32 + ss*16 (+8): leaves ((row >> 1) & 1) unchanged
        v_off[dt] = row * ROWB + (((dcol >> 3) ^ (((row >> 1) & 1) << 2)) << 4) + (dcol & 7) * 2;
    }
    const int nkt = (Nk + KT - 1) / KT;

    // scores of one key tile: S'' = K.Q'^T - m_ref (the accumulator input of the first MFMA of each chain is -m_ref)
    auto qk = [&](const char* kbuf, f32x16 (&s)[2], auto zero_tag) {
        constexpr bool ZERO_REF = decltype(zero_tag)::value != 0;    // the reference of every query of this wave is 0: the accumulator input is the inline constant
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const half8v kf = *reinterpret_cast<const half8v*>(kbuf + k_off[ks] + sub * 32 * ROWB);
                s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], ks == 0 ? (ZERO_REF ? zero16 : negm) : s[sub], 0, 0, 0);
            }
    };
    // one key tile (K and V in slot B): scores, lazy reference, exp2/pack in four 16-key groups each followed by its 3 MFMAs
    // (row sum + two O tiles), so a group's MFMAs run under the next group's exp2
    auto tile = [&](auto b_tag, f32x16 (&sc)[2], int kt) {
        constexpr int VB = decltype(b_tag)::value;
        // Round 5: while the reference of every query of the wave is still 0 (`ref_zero`, wave-uniform) the score MFMAs take the inline
        // constant 0 as their accumulator input and the 16 v_mov that rebuild the -m_ref operand per tile (64 of the tile's 960 issue-port
        // cycles) are not executed.  The reference starts at 0 unless the first tile's maximum lies outside [-REF0_BAND, REF0_BAND] (then
        // it is that maximum, as before) and moves, as before, when a tile maximum exceeds it by more than LAZY_THR: P <= 2^8 and the
        // row's largest P >= 2^-REF0_BAND either way.
#ifndef MOCA_ATTN_NO_ZERO_REF
        if (ref_zero) {
            qk(sK + VB * TILE, sc, int_c<1>{});
        } else
#endif
        {
#ifndef MOCA_ATTN_NEGM_PERSISTENT   // -m_ref rebuilt per tile (16 v_mov) instead of living across it: 141 -> 121 VGPRs = 4 waves per SIMD, +3.5 %
#pragma unroll
            for (int r = 0; r < 16; ++r) negm[r] = -m_ref;
#endif
            qk(sK + VB * TILE, sc, int_c<0>{});
        }
        if ((kt + 1) * KT > Nk) {                     // keys beyond Nk (last tile only)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * KT + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    if (key >= Nk) sc[sub][r] = -INFINITY;
                }
        }
        // tile maximum as four independent chains of v_max3_f32.  As instructions, not fmaxf: on accumulator outputs the compiler puts a
        // canonicalising v_max_f32 x, x in front of every fmaxf operand (47 vector instructions for these 32 values instead of 18 -- a
        // fifth of the key tile's vector issue, which is the longer pole of this kernel).  Scores are finite or -inf, never NaN.
        float t4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            t4[c] = vmax3(sc[0][c], sc[1][c], sc[0][4 + c]);
            t4[c] = vmax3(t4[c], sc[1][4 + c], sc[0][8 + c]);
            t4[c] = vmax3(t4[c], sc[1][8 + c], sc[0][12 + c]);
            t4[c] = vmax3(t4[c], sc[1][12 + c], sc[1][12 + c]);
        }
        float tmax = vmax3(vmax3(t4[0], t4[1], t4[2]), t4[3], t4[3]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));      // both key halves of this query: the two lanes agree from here on
        if (kt == 0 || __any(tmax > LAZY_THR)) {      // move the reference (first tile: to its maximum unless that lies in the zero band)
#ifndef MOCA_ATTN_NO_ZERO_REF
            const float delta = kt == 0 ? (fabsf(tmax) > REF0_BAND ? tmax : 0.f) : fmaxf(tmax, 0.f);
#else
            const float delta = kt == 0 ? tmax : fmaxf(tmax, 0.f);
#endif
            const float alpha = __builtin_amdgcn_exp2f(-delta);          // (kt == 0: O and l are still zero)
            m_ref += delta;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o[0][r] *= alpha; o[1][r] *= alpha; osum[r] *= alpha;
                if (r == 0) lsum *= alpha;
                sc[0][r] -= delta; sc[1][r] -= delta;
#ifdef MOCA_ATTN_NEGM_PERSISTENT
                negm[r] = -m_ref;
#endif
            }
            ref_zero = !__any(m_ref != 0.f);
        }
        const char* vbuf = sV + VB * TILE;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                half2v h[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float p0 = __builtin_amdgcn_exp2f(sc[sub][ss * 8 + 2 * j]);
                    const float p1 = __builtin_amdgcn_exp2f(sc[sub][ss * 8 + 2 * j + 1]);
                    h[j] = __builtin_convertvector(f32x2{p0, p1}, half2v);
                }
                const half4v plo = __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3);
                const half4v phi = __builtin_shufflevector(h[2], h[3], 0, 1, 2, 3);
                const half8v pf = __builtin_shufflevector(plo, phi, 0, 1, 2, 3, 4, 5, 6, 7);
#ifndef MOCA_ATTN_SUM_MFMA      // row sums by v_dot2_f32_f16 on the packed P (+5 % over a 4th "ones" MFMA per 16 keys, and 16 registers fewer)
#pragma unroll
                for (int j = 0; j < 4; ++j) lsum = __builtin_amdgcn_fdot2(h[j], half2v{(half_t)1.0f, (half_t)1.0f}, lsum, false);
#else
                osum = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, pf, osum, 0, 0, 0);
#endif
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const char* a0 = vbuf + v_off[dt] + (sub * 32 + ss * 16) * ROWB;
                    const half4v lo = tr_read(a0);
                    const half4v hi = tr_read(a0 + 8 * ROWB);
                    const half8v vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[dt], 0, 0, 0);
                }
            }
    };

    // tile kt+1 is fetched while tile kt is computed: DMA issued after the barrier that retired the buffer's last readers,
    // awaited (vmcnt(0): a whole tile of compute later) before the barrier that publishes it.  (A software-pipelined variant
    // -- scores of tile kt+1 issued under the softmax of tile kt, `step(...)` with two score register sets -- needs 196 VGPRs =
    // two waves per SIMD and measured 446 us against 401 us for this form at 157 VGPRs = three waves per SIMD: occupancy wins.)
    f32x16 sa[2];
    dma_k(0, 0); dma_v(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nkt; kt += 2) {
        if (kt + 1 < nkt) { dma_k(kt + 1, 1); dma_v(kt + 1, 1); }
        tile(int_c<0>{}, sa, kt);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nkt) {
            if (kt + 2 < nkt) { dma_k(kt + 2, 0); dma_v(kt + 2, 0); }
            tile(int_c<1>{}, sa, kt + 1);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }

#ifndef MOCA_ATTN_SUM_MFMA
    const float inv = 1.0f / (lsum + __shfl_xor(lsum, 32, 64));
#else
    const float inv = 1.0f / osum[0];
#endif
    // O rows leave as whole 128-byte rows through the (now free) K/V buffers: wave w owns 4 KiB at smem + 4096 w (the last
    // barrier of the key loop retired every fragment read; each wave touches only its own 32 rows, so no further barrier)
    {
        char* wq = smem + wave * 32 * ROWB;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                half4v h4;
#pragma unroll
                for (int j = 0; j < 4; ++j) h4[j] = (half_t)(o[dt][g * 4 + j] * inv);
                const int d0 = dt * 32 + 8 * g + 4 * fh;
                *reinterpret_cast<half4v*>(wq + fr * ROWB + (((d0 >> 3) ^ ((fr >> 1) & 7)) << 4) + (d0 & 7) * 2) = h4;
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i, row = c >> 3, ch = c & 7;
            if (q0 + row < Nq)
                *reinterpret_cast<half8v*>(out + ((int64_t)bq * Nq + q0 + row) * ldo + head * D + ch * 8) =
                    *reinterpret_cast<const half8v*>(wq + row * ROWB + ((ch ^ ((row >> 1) & 7)) << 4));
        }
    }
    (void)q_ok;
#endif
}

// ---- temporal attention: one wavefront per (video, pixel, head), T <= 16 -------
constexpr int TV_ROWB = 144;  // 128 B + 16 B pad per V row in LDS

__global__ __launch_bounds__(256) void temporal_attention_kernel(
    const half_t* __restrict__ q, const half_t* __restrict__ k, const half_t* __restrict__ v, half_t* __restrict__ out,
    int B, int T, int HW, int heads, int ld, int ldo, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) char sV[4][16 * TV_ROWB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t total = (int64_t)B * HW * heads;
    int64_t prob = (int64_t)blockIdx.x * 4 + wave;
    const bool active = prob < total;
    if (!active) prob = total - 1;
    const int head = (int)(prob % heads);
    const int64_t bp = prob / heads;
    const int pix = (int)(bp % HW), b = (int)(bp / HW);
    // token row of frame t: (b*T + t)*HW + pix
    const int64_t row0 = (int64_t)b * T * HW + pix;
    const int64_t rstride = (int64_t)HW * ld;
    const int64_t base = row0 * ld + head * D;

    const int fr = lane & 15, fg = lane >> 4;
    // K (A operand) and Q (B operand) fragments of S^T = K.Q^T: row fr, d = 32 ks + 8 fg + j
    half8v kf[2], qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        half8v a = {0, 0, 0, 0, 0, 0, 0, 0}, c = {0, 0, 0, 0, 0, 0, 0, 0};
        if (fr < T) {
            a = *reinterpret_cast<const half8v*>(k + base + fr * rstride + ks * 32 + fg * 8);
            c = *reinterpret_cast<const half8v*>(q + base + fr * rstride + ks * 32 + fg * 8);
        }
        kf[ks] = a; qf[ks] = c;
    }
    // V -> LDS row-major [key][d]
    char* sv = sV[wave];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (lane >> 3) + 8 * i, ch = lane & 7;
        half8v a = {0, 0, 0, 0, 0, 0, 0, 0};
        if (row < T) a = *reinterpret_cast<const half8v*>(v + base + row * rstride + ch * 8);
        *reinterpret_cast<half8v*>(sv + row * TV_ROWB + ch * 16) = a;
    }
    __syncthreads();

    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    s = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[0], qf[0], s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[1], qf[1], s, 0, 0, 0);
    // lane holds S^T[key = 4 fg + r][q = fr]
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (4 * fg + r >= T) s[r] = -INFINITY;
        mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
    half4v pf;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float pv = exp2f((s[r] - mx) * scale_log2e);
        sum += pv;
        pf[r] = (half_t)pv;
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;

    // O^T[d][q] = V^T[d][key] . P^T[key][q]; A: lane (d = fr, k = 4 fg + j) = V[4fg+j][16dt + fr]
    const bool st_ok = active && fr < T;
    half_t* ob = out + (row0 + (int64_t)fr * HW) * ldo + head * D;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        half4v vf;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            vf[j] = *reinterpret_cast<const half_t*>(sv + (4 * fg + j) * TV_ROWB + (dt * 16 + fr) * 2);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(vf, pf, acc, 0, 0, 0);
        // lane holds O^T[d = 16dt + 4fg + r][q = fr]
        if (st_ok) {
            half4v h4;
#pragma unroll
            for (int r = 0; r < 4; ++r) h4[r] = (half_t)(acc[r] * inv);
            *reinterpret_cast<half4v*>(ob + dt * 16 + 4 * fg) = h4;
        }
    }
}

}  // namespace

extern "C" int moca_attention_f16(const void* q, const void* k, const void* v, void* out,
                                  int32_t Bq, int32_t heads, int32_t Nq, int32_t Nk,
                                  int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                                  int32_t kv_div, float scale, void* stream) {
    if (!q || !k || !v || !out) return MOCA_E_BADARG;
    if (Bq <= 0 || heads <= 0 || Nq <= 0 || Nk <= 0 || kv_div <= 0 || Bq % kv_div) return MOCA_E_BADARG;
    if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4) return MOCA_E_BADARG;
    if (ldq < heads * D || ldk < heads * D || ldv < heads * D || ldo < heads * D) return MOCA_E_BADARG;
    if ((int64_t)Bq * heads > 65535) return MOCA_E_BADARG;
    const dim3 grid((Nq + QB - 1) / QB, Bq * heads), block(256);
    if (Nk >= 2 * KT) {
        // (an 8-wave form, 256 queries per block, measured -2.8 % at 2560 tokens in isolation and -0.1 % on the whole step in round 2:
        //  not kept)
        hipLaunchKernelGGL(attention_v4_kernel<4>, grid, block, 0, moca_stream(stream),
                           reinterpret_cast<const half_t*>(q), reinterpret_cast<const half_t*>(k),
                           reinterpret_cast<const half_t*>(v), reinterpret_cast<half_t*>(out),
                           heads, Nq, Nk, ldq, ldk, ldv, ldo, kv_div, scale * 1.4426950408889634f);
        MOCA_CHECK_LAUNCH();
        return MOCA_OK;
    }
    if (Nk <= KS96) {
        // query groups of 128 per block: as many as keep >= ~3 blocks per CU in the launch (K / V are staged once per block)
        const int qgroups = (Nq + QB - 1) / QB;
        int q_iters = (int)(((int64_t)qgroups * Bq * heads) / 768);
        if (q_iters < 1) q_iters = 1;
        if (q_iters > 8) q_iters = 8;
        if (q_iters > qgroups) q_iters = qgroups;
        const dim3 grid_s((qgroups + q_iters - 1) / q_iters, Bq * heads);
        hipLaunchKernelGGL(attention_short_kernel, grid_s, block, 0, moca_stream(stream),
                           reinterpret_cast<const half_t*>(q), reinterpret_cast<const half_t*>(k),
                           reinterpret_cast<const half_t*>(v), reinterpret_cast<half_t*>(out),
                           heads, Nq, Nk, ldq, ldk, ldv, ldo, kv_div, scale * 1.4426950408889634f, q_iters);
        MOCA_CHECK_LAUNCH();
        return MOCA_OK;
    }
    hipLaunchKernelGGL(attention_kernel<false>, grid, block, 0, moca_stream(stream),
                       reinterpret_cast<const half_t*>(q), reinterpret_cast<const half_t*>(k),
                       reinterpret_cast<const half_t*>(v), reinterpret_cast<half_t*>(out),
                       heads, Nq, Nk, ldq, ldk, ldv, ldo, kv_div, scale * 1.4426950408889634f);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_attention_causal_f16(const void* q, const void* k, const void* v, void* out,
                                         int32_t B, int32_t heads, int32_t N, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                                         float scale, void* stream) {
    if (!q || !k || !v || !out || B <= 0 || heads <= 0 || N <= 0) return MOCA_E_BADARG;
    if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4) return MOCA_E_BADARG;
    if (ldq < heads * D || ldk < heads * D || ldv < heads * D || ldo < heads * D) return MOCA_E_BADARG;
    if ((int64_t)B * heads > 65535) return MOCA_E_BADARG;
    const dim3 grid((N + QB - 1) / QB, B * heads), block(256);
    hipLaunchKernelGGL(attention_kernel<true>, grid, block, 0, moca_stream(stream),
                       reinterpret_cast<const half_t*>(q), reinterpret_cast<const half_t*>(k),
                       reinterpret_cast<const half_t*>(v), reinterpret_cast<half_t*>(out),
                       heads, N, N, ldq, ldk, ldv, ldo, 1, scale * 1.4426950408889634f);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_temporal_attention_f16(const void* q, const void* k, const void* v, void* out,
                                           int32_t B, int32_t T, int32_t HW, int32_t heads,
                                           int32_t ld_qkv, int32_t ldo, float scale, void* stream) {
    if (!q || !k || !v || !out) return MOCA_E_BADARG;
    if (B <= 0 || T <= 0 || T > 16 || HW <= 0 || heads <= 0) return MOCA_E_BADARG;
    if (ld_qkv % 8 || ldo % 4 || ld_qkv < heads * D || ldo < heads * D) return MOCA_E_BADARG;
    const int64_t total = (int64_t)B * HW * heads;
    const dim3 grid((unsigned)((total + 3) / 4)), block(256);
    hipLaunchKernelGGL(temporal_attention_kernel, grid, block, 0, moca_stream(stream),
                       reinterpret_cast<const half_t*>(q), reinterpret_cast<const half_t*>(k),
                       reinterpret_cast<const half_t*>(v), reinterpret_cast<half_t*>(out),
                       B, T, HW, heads, ld_qkv, ldo, scale * 1.4426950408889634f);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

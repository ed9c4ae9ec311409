// Implicit-GEMM for the VideoCrafter2 3D-UNet on gfx950 (MI355X).
//
//   out[M][N] = epilogue( gather(A)[M][K] . W[N][K]^T )
//
// One kernel family covers every contraction on the denoising path: 3x3 conv2d
// (stride 1/2, optional fused nearest x2 upsample), the (3,1,1) temporal conv3d and
// all nn.Linear layers; the A operand is gathered on the fly from channels-last fp16
// activations ([frame][y][x][c]), so no im2col buffer and no layout shuffles exist.
//
// Tiling (CDNA4): 256 threads = 4 wavefronts (2 x 2), block tile 128 x BN x 64
// (BN = 128 or 64), each wave owns a 64 x BN/2 sub-tile as 2 x (BN/64)
// v_mfma_f32_32x32x16_f16 accumulators.  A/B k-tiles are staged through registers
// (the gather needs zero fill) into double-buffered LDS with 128-byte rows and an
// XOR swizzle on the 16-byte chunk index (chunk ^ ((row>>1)&7)) so that the
// ds_read_b128 fragment reads are bank-conflict free.  The epilogue goes through an
// fp32 LDS tile so that bias / time-embedding / residual are applied in fp32 and the
// result leaves as full 16-byte coalesced stores.
#include "common.h"

namespace {

template <int V> struct int_c { static constexpr int value = V; };
struct yes_t { static constexpr bool value = true; };
struct no_t { static constexpr bool value = false; };

constexpr int BM = 128;
constexpr int BK = 64;
constexpr int MOCA_A_LINEAR2 = 3;      // internal: MOCA_A_LINEAR with moca_gemm_params.a2 (two-source A = virtual torch.cat), staggered kernels only
constexpr int ROW_BYTES = BK * 2;  // 128 B per LDS row

struct Geo {
    // conv3x3 / tconv geometry in registers
    int C, inH, inW, outH, outW, stride, up, T, HW;
};

// 2-D XCD partition of the tile grid: XCD (i, j) of an xm x xn arrangement (xm * xn = 8) owns the sub-grid of tiles_m / xm row
// tiles x tiles_n / xn column tiles and walks it column-fastest.  With the 1-D partition above every XCD streams ALL of W once per
// M tile it owns: at the 1280-channel level (K = 1280, N = 10240: W = 26 MB against a 4 MB L2) rocprofv3 shows 559 MB of fabric
// reads per GEGLU launch for 39 MB of operands.  (xm, xn) is chosen on the host to minimise (rows of A + rows of W) per XCD and only
// when both divide; p.reserved4_ >> 8 carries xn (0 / 1: the 1-D partition).
__device__ __forceinline__ void remap_tile_2d(int tiles_m, int tiles_n, int xn, int& tile_m, int& tile_n) {
    const int xm = 8 / xn;
    const int sm = tiles_m / xm, sn = tiles_n / xn;          // sub-grid of one XCD
    const int b = blockIdx.x;
    const int xcd = b & 7, j = b >> 3;                        // j-th tile of this XCD
    const int xi = xcd / xn, xj = xcd - xi * xn;
    tile_m = xi * sm + j / sn;
    tile_n = xj * sn + j % sn;
}

// Blocks nblk .. gridDim.x - 1 of a launch with p.prefetch: no tile, they read the next weight-heavy launch's weights once (plain
// 16-byte loads, 8 in flight per lane; the values are dropped) so that those lines sit in the Infinity Cache when that launch asks
// for them.  They are dispatched after every tile block, i.e. onto the CUs this launch leaves idle or onto its tail.
constexpr int PREFETCH_BLOCKS = 24;
__device__ __forceinline__ bool prefetch_block(const moca_gemm_params& p, int nblk, int nthreads) {
    if ((int)blockIdx.x < nblk) return false;
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(p.prefetch);
    const int64_t n16 = (int64_t)p.prefetch_kib << 6;
    const int64_t stride = (int64_t)((int)gridDim.x - nblk) * nthreads;
    int64_t i = (int64_t)((int)blockIdx.x - nblk) * nthreads + threadIdx.x;
    unsigned acc = 0;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[i + j * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    for (; i < n16; i += stride) { const uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    asm volatile("" :: "v"(acc));                     // keeps the loads; nothing is stored
    return true;
}
static inline int prefetch_blocks(const moca_gemm_params& p) { return (p.prefetch && p.prefetch_kib > 0) ? PREFETCH_BLOCKS : 0; }

template <int BN>
__device__ __forceinline__ void remap_block(int nblk, int& logical) {
    // XCD-aware bijective remap: physical blocks b, b+8, b+16, ... share an XCD (L2);
    // give each XCD a contiguous range of logical tiles so neighbours share A rows.
    const int b = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = b & 7, j = b >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

template <int BN, int AMODE>
__global__ __launch_bounds__(256, 2) void gemm_f16_kernel(const moca_gemm_params p) {
    constexpr int NT = BN / 64;          // 32-wide n tiles per wave
    constexpr int B_SLOTS = BN / 32;     // 16-B chunks of W per thread per k-tile
    constexpr int A_BYTES = BM * ROW_BYTES;
    constexpr int B_BYTES = BN * ROW_BYTES;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                  // [2][A_BYTES]
    char* sB = smem + 2 * A_BYTES;    // [2][B_BYTES]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave >> 1, wave_n = wave & 1;

    const int tiles_m = (p.M + BM - 1) / BM;
    const int tiles_n = p.N / BN;
    const int nblk = tiles_m * tiles_n * p.splits;
    int logical;
    remap_block<BN>(nblk, logical);
    const int split = logical % p.splits;
    const int tile = logical / p.splits;
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int nk_total = (p.K + BK - 1) / BK;      // (W rows are readable and zero beyond K up to the next multiple of 64)
    const int kts = (nk_total + p.splits - 1) / p.splits;
    const int kt_begin = split * kts;
    const int kt_end = min(kt_begin + kts, nk_total);

    // ---- per-thread staging coordinates --------------------------------------
    const int cc = tid & 7;     // 16-byte chunk column inside the k-tile
    const int r0 = tid >> 3;    // rows r0 + 32*i
    const int swz = (r0 >> 1) & 7;  // identical for all rows r0+32i

    const half_t* __restrict__ Aptr = reinterpret_cast<const half_t*>(p.a);
    const half_t* __restrict__ Wptr = reinterpret_cast<const half_t*>(p.w);

    // row descriptors
    int64_t row_off[4];   // LINEAR: m*lda ; CONV: frame pixel base ; TCONV: m
    int row_y[4], row_x[4];
    bool row_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + r0 + 32 * i;
        row_ok[i] = m < p.M;
        const int mm = row_ok[i] ? m : 0;
        if (AMODE == MOCA_A_LINEAR) {
            row_off[i] = (int64_t)mm * p.lda;
            row_y[i] = row_x[i] = 0;
        } else if (AMODE == MOCA_A_CONV3X3) {
            const int ohw = p.outH * p.outW;
            const int f = mm / ohw, rem = mm - f * ohw;
            const int oy = rem / p.outW, ox = rem - oy * p.outW;
            row_off[i] = (int64_t)f * p.inH * p.inW;
            row_y[i] = oy * p.stride - 1 + p.nopad_lo;
            row_x[i] = ox * p.stride - 1 + p.nopad_lo;
        } else {  // TCONV3
            const int frame = mm / p.HW;
            row_off[i] = mm;
            row_y[i] = frame % p.T;  // t
            row_x[i] = 0;
        }
    }

    half8v ra[4], rb[B_SLOTS];

    auto load_tile = [&](int kt) {
        const int k = kt * BK + cc * 8;
        // ---- A ----
        if (AMODE == MOCA_A_LINEAR) {
            const bool kok = k < p.K;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                half8v v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (kok && row_ok[i]) v = *reinterpret_cast<const half8v*>(Aptr + row_off[i] + k);
                ra[i] = v;
            }
        } else if (AMODE == MOCA_A_CONV3X3) {
            const int tap = k / p.C, c = k - tap * p.C;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int limH = p.up ? 2 * p.inH : p.inH, limW = p.up ? 2 * p.inW : p.inW;
            const bool kok = tap < 9;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int iy = row_y[i] + ky, ix = row_x[i] + kx;
                const bool ok = kok && row_ok[i] && iy >= 0 && iy < limH && ix >= 0 && ix < limW;
                if (p.up) { iy >>= 1; ix >>= 1; }
                half8v v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ok) v = *reinterpret_cast<const half8v*>(Aptr + (row_off[i] + (int64_t)iy * p.inW + ix) * p.C + c);
                ra[i] = v;
            }
        } else {
            const int tap = k / p.C, c = k - tap * p.C;
            const bool kok = tap < 3;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int tt = row_y[i] + tap - 1;
                const bool ok = kok && row_ok[i] && tt >= 0 && tt < p.T;
                half8v v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ok) v = *reinterpret_cast<const half8v*>(Aptr + (row_off[i] + (int64_t)(tap - 1) * p.HW) * p.C + c);
                ra[i] = v;
            }
        }
        // ---- W (always in range: N % BN == 0, ldw % 64 == 0, zero padded) ----
#pragma unroll
        for (int i = 0; i < B_SLOTS; ++i) {
            const int n = n0 + r0 + 32 * i;
            rb[i] = *reinterpret_cast<const half8v*>(Wptr + (int64_t)n * p.ldw + k);
        }
    };

    auto store_tile = [&](int buf) {
        char* a = sA + buf * A_BYTES;
        char* b = sB + buf * B_BYTES;
        const int coff = (cc ^ swz) << 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<half8v*>(a + (r0 + 32 * i) * ROW_BYTES + coff) = ra[i];
#pragma unroll
        for (int i = 0; i < B_SLOTS; ++i)
            *reinterpret_cast<half8v*>(b + (r0 + 32 * i) * ROW_BYTES + coff) = rb[i];
    };

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // fragment read coordinates
    const int frow = lane & 31, fh = lane >> 5;
    int a_row_b[2], a_swz[2], b_row_b[NT], b_swz[NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int row = wave_m * 64 + mt * 32 + frow;
        a_row_b[mt] = row * ROW_BYTES;
        a_swz[mt] = (row >> 1) & 7;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int row = wave_n * (BN / 2) + nt * 32 + frow;
        b_row_b[nt] = row * ROW_BYTES;
        b_swz[nt] = (row >> 1) & 7;
    }

    if (kt_begin < kt_end) {
        load_tile(kt_begin);
        store_tile(0);
    }
    __syncthreads();

    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool has_next = kt + 1 < kt_end;
        if (has_next) load_tile(kt + 1);

        const char* a = sA + cur * A_BYTES;
        const char* b = sB + cur * B_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int ch = ks * 2 + fh;
            half8v af[2], bf[NT];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                af[mt] = *reinterpret_cast<const half8v*>(a + a_row_b[mt] + ((ch ^ a_swz[mt]) << 4));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                bf[nt] = *reinterpret_cast<const half8v*>(b + b_row_b[nt] + ((ch ^ b_swz[nt]) << 4));
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
        }
        if (has_next) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- split-K: raw fp32 partials, epilogue runs in splitk_reduce_kernel ----
    if (p.splits > 1) {
        float* ws = p.splitk_ws + (int64_t)split * p.M * p.N;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wave_m * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    const int col = n0 + wave_n * (BN / 2) + nt * 32 + frow;
                    if (row < p.M) ws[(int64_t)row * p.N + col] = acc[mt][nt][r];
                }
        return;
    }

    // ---- epilogue stage 1: registers -> fp32 LDS tile (bias, GEGLU) ----
    // (the main loop's last __syncthreads() makes the pipeline buffers reusable)
    float* sC = reinterpret_cast<float*>(smem);
    const bool geglu = (p.flags & MOCA_EP_GEGLU) != 0;
    const int out_bn = geglu ? BN / 2 : BN;  // output columns of this tile
    if (geglu) {
        if constexpr (NT == 2) {
            const int ncol = n0 + wave_n * 64 + frow;
            const float bv = p.bias ? p.bias[ncol] : 0.f;
            const float bg = p.bias ? p.bias[ncol + 32] : 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wave_m * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    const float v = acc[mt][0][r] + bv;
                    const float g = acc[mt][1][r] + bg;
                    sC[row * out_bn + wave_n * 32 + frow] = v * moca_gelu(g);
                }
        }
    } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = wave_n * (BN / 2) + nt * 32 + frow;
            const float bv = p.bias ? p.bias[n0 + col] : 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wave_m * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    const float v = acc[mt][nt][r] + bv;
                    sC[row * out_bn + col] = (p.flags & MOCA_EP_GELU) ? moca_gelu(v) : v;
                }
        }
    }
    __syncthreads();

    // ---- epilogue stage 2: coalesced 16-byte rows (+rowadd, +residual) ----
    const int chunks_per_row = out_bn / 8;
    const int total_chunks = BM * chunks_per_row;
    const int on0 = geglu ? n0 / 2 : n0;
    const half_t* __restrict__ rowadd = reinterpret_cast<const half_t*>(p.rowadd);
    const half_t* __restrict__ resid = reinterpret_cast<const half_t*>(p.residual);
    for (int idx = tid; idx < total_chunks; idx += 256) {
        const int row = idx / chunks_per_row, ch = idx - row * chunks_per_row;
        const int m = m0 + row;
        if (m >= p.M) continue;
        const int col = on0 + ch * 8;
        float v[8];
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(sC + row * out_bn + ch * 8);
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(sC + row * out_bn + ch * 8 + 4);
        v[0] = c0[0]; v[1] = c0[1]; v[2] = c0[2]; v[3] = c0[3];
        v[4] = c1[0]; v[5] = c1[1]; v[6] = c1[2]; v[7] = c1[3];
        if (rowadd) {
            const half8v e = *reinterpret_cast<const half8v*>(rowadd + (int64_t)(m / p.rowadd_div) * p.ld_rowadd + col);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)e[j];
        }
        if (resid) {
            const half8v e = *reinterpret_cast<const half8v*>(resid + (int64_t)m * p.ldr + col);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)e[j];
        }
        if (p.flags & MOCA_EP_OUT_F32) {
            float* o = reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + col;
            *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
            half8v h;
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = (half_t)v[j];
            *reinterpret_cast<half8v*>(reinterpret_cast<half_t*>(p.out) + (int64_t)m * p.ldo + col) = h;
        }
    }
}

// Sum the split-K partial slabs and run the same epilogue.  One thread per 8 output
// columns; HBM-bound (reads splits*M*N*4 bytes).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const moca_gemm_params p) {
    const bool geglu = (p.flags & MOCA_EP_GEGLU) != 0;
    const int out_n = geglu ? p.N / 2 : p.N;
    const int cpr = out_n / 8;
    const int64_t total = (int64_t)p.M * cpr;
    const half_t* __restrict__ rowadd = reinterpret_cast<const half_t*>(p.rowadd);
    const half_t* __restrict__ resid = reinterpret_cast<const half_t*>(p.residual);
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int m = (int)(idx / cpr), ch = (int)(idx - (int64_t)m * cpr);
        const int col = ch * 8;
        float v[8];
        if (geglu) {
            // output col j <- value col (j/32)*64 + j%32, gate = value + 32; 8 | 32 so a chunk never straddles
            const int vc = (col / 32) * 64 + (col % 32);
            float a[8], g[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { a[j] = p.bias ? p.bias[vc + j] : 0.f; g[j] = p.bias ? p.bias[vc + 32 + j] : 0.f; }
            for (int s = 0; s < p.splits; ++s) {
                const float* w = p.splitk_ws + ((int64_t)s * p.M + m) * p.N + vc;
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(w), a1 = *reinterpret_cast<const f32x4*>(w + 4);
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(w + 32), g1 = *reinterpret_cast<const f32x4*>(w + 36);
#pragma unroll
                for (int j = 0; j < 4; ++j) { a[j] += a0[j]; a[4 + j] += a1[j]; g[j] += g0[j]; g[4 + j] += g1[j]; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = a[j] * moca_gelu(g[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = p.bias ? p.bias[col + j] : 0.f;
            if (p.reserved4_ & 4) {                               // fp16 slabs (MOCA_TUNE_SLAB_F16)
                for (int s = 0; s < p.splits; ++s) {
                    const half8v a = *reinterpret_cast<const half8v*>(reinterpret_cast<const half_t*>(p.splitk_ws) + ((int64_t)s * p.M + m) * p.N + col);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += (float)a[j];
                }
            } else {
                for (int s = 0; s < p.splits; ++s) {
                    const float* w = p.splitk_ws + ((int64_t)s * p.M + m) * p.N + col;
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(w), a1 = *reinterpret_cast<const f32x4*>(w + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] += a0[j]; v[4 + j] += a1[j]; }
                }
            }
        }
        if (rowadd) {
            const half8v e = *reinterpret_cast<const half8v*>(rowadd + (int64_t)(m / p.rowadd_div) * p.ld_rowadd + col);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)e[j];
        }
        if (resid) {
            const half8v e = *reinterpret_cast<const half8v*>(resid + (int64_t)m * p.ldr + col);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)e[j];
        }
        if (p.flags & MOCA_EP_OUT_F32) {
            float* o = reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + col;
            *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
            half8v h;
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = (half_t)v[j];
            *reinterpret_cast<half8v*>(reinterpret_cast<half_t*>(p.out) + (int64_t)m * p.ldo + col) = h;
        }
    }
}

// ---- split-K reduce + the GroupNorm(+SiLU) that consumes the result, in ONE launch (moca_gemm_splitk_groupnorm_f16): at the
// 5 x 8-latent level (M = 1280 at B = 2: 50 tiles per launch, every conv / temporal conv runs split-K) a ResBlock is a chain of
// GEMM -> reduce -> GroupNorm launches of 5-9 us each on tensors of 3 MB.  One block per (statistics group, channel group) slab, as
// gn_slab_reg_kernel: a thread sums the split-K slabs of its <= CPT 16-byte output chunks (+ bias / row add / residual: the epilogue of
// splitk_reduce_kernel, same order), rounds them to fp16 -- the value the two-launch path stores and normalises --, optionally
// writes them (write_x: the residual of a later launch), the block reduces sum / sum of squares and normalises from registers.
template <int CPT>
__global__ __launch_bounds__(1024) void splitk_gn_kernel(const moca_gemm_params p, half_t* __restrict__ y, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int R, int cpg, double inv_count, float eps,
                                                         int silu, int write_x) {
    __shared__ float s_red[2][16];
    __shared__ float s_sc[128], s_sh[128];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int slab = blockIdx.x, sg = slab / 32, g = slab % 32;
    const int vpr = cpg / 8, nchunks = R * vpr;
    const int N = p.N;
    const half_t* __restrict__ rowadd = reinterpret_cast<const half_t*>(p.rowadd);
    const half_t* __restrict__ resid = reinterpret_cast<const half_t*>(p.residual);
    const float gm_t = tid < cpg ? gamma[g * cpg + tid] : 0.f, bt_t = tid < cpg ? beta[g * cpg + tid] : 0.f;   // (in flight under the slab loads)
    half8v v[CPT];
    int mrow[CPT], ccol[CPT];
    float s = 0.f, q = 0.f;
    // (chunk by chunk, split by split.  Prefetching the next split's loads of all CPT chunks -- two register sets -- measured SLOWER, 23.1 against
    //  18.6 us on the 16-frame slabs: what bounds a block is not the dependent round trips but 160-byte row segments of the fp32 slabs at a
    //  5 KiB stride on the 64 CUs the 64 slabs occupy; profiles/r05_ab_splitk_groupnorm.txt)
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int idx = tid + i * nthr;
        mrow[i] = -1;
        if (idx < nchunks) {
            const int row = idx / vpr, c = idx - row * vpr;
            const int m = sg * R + row, col = g * cpg + c * 8;
            mrow[i] = m; ccol[i] = col;
            float a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = p.bias ? p.bias[col + j] : 0.f;
            if (p.reserved4_ & 4) {                               // fp16 slabs (MOCA_TUNE_SLAB_F16)
                for (int sp = 0; sp < p.splits; ++sp) {
                    const half8v h = *reinterpret_cast<const half8v*>(reinterpret_cast<const half_t*>(p.splitk_ws) + ((int64_t)sp * p.M + m) * N + col);
#pragma unroll
                    for (int j = 0; j < 8; ++j) a[j] += (float)h[j];
                }
            } else {
                for (int sp = 0; sp < p.splits; ++sp) {
                    const float* w = p.splitk_ws + ((int64_t)sp * p.M + m) * N + col;
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(w), a1 = *reinterpret_cast<const f32x4*>(w + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { a[j] += a0[j]; a[4 + j] += a1[j]; }
                }
            }
            if (rowadd) {
                const half8v e = *reinterpret_cast<const half8v*>(rowadd + (int64_t)(m / p.rowadd_div) * p.ld_rowadd + col);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] += (float)e[j];
            }
            if (resid) {
                const half8v e = *reinterpret_cast<const half8v*>(resid + (int64_t)m * p.ldr + col);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] += (float)e[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = (half_t)a[j];
            if (write_x) *reinterpret_cast<half8v*>(reinterpret_cast<half_t*>(p.out) + (int64_t)m * p.ldo + col) = v[i];
        }
    }
#pragma unroll
    for (int i = 0; i < CPT; ++i)
        if (mrow[i] >= 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float a = (float)v[i][j]; s += a; q += a * a; }
        }
    s = wave_sum(s); q = wave_sum(q);
    if ((tid & 63) == 0) { s_red[0][tid >> 6] = s; s_red[1][tid >> 6] = q; }
    __syncthreads();
    if (tid < cpg) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < (nthr >> 6); ++w) { a += (double)s_red[0][w]; b += (double)s_red[1][w]; }
        const double mean = a * inv_count;
        double var = b * inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = rstd * gm_t;
        s_sc[tid] = sc;
        s_sh[tid] = bt_t - (float)mean * sc;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < CPT; ++i)
        if (mrow[i] >= 0) {
            const int c0 = ccol[i] - g * cpg;
            half8v r;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float a = (float)v[i][j] * s_sc[c0 + j] + s_sh[c0 + j];
                if (silu) a = moca_silu(a);
                r[j] = (half_t)a;
            }
            *reinterpret_cast<half8v*>(y + (int64_t)mrow[i] * N + ccol[i]) = r;
        }
}


// =====================================================================================
// Large-tile kernel: 256 x BN x 64 block tile (BN = 128 or 160), 512 threads = 8
// wavefronts (4 x 2), wave tile 64 x BN/2 as 4 x (BN/32) v_mfma_f32_16x16x32_f16
// accumulators.  Operands are staged global -> LDS directly (global_load_lds_dwordx4,
// no VGPR round trip) into a 3-slot ring, two k-tiles in flight, ONE raw s_barrier per
// k-tile behind a counted s_waitcnt vmcnt(N) (never 0 in the steady state).  The LDS
// image of a DMA is lane-linear, so the bank-conflict XOR swizzle is applied on the
// per-lane SOURCE address (logical chunk = physical chunk ^ ((row>>1)&7)) and again on
// the fragment reads.  Zero padding (conv halo, M/K tails) is a lane whose source is a
// 16-byte zero page.
// =====================================================================================
// zeros: long enough that `zero + per-tile k offset` of the fast gather path stays inside it
constexpr int ZERO_PAGE_HALVES = 8192 + 64;
__device__ __attribute__((aligned(16))) half_t g_zero_page[ZERO_PAGE_HALVES];

#ifdef MOCA_STAMPS
// diagnostic build only (-DMOCA_STAMPS): per-block s_memtime stamps, read back with moca_debug_stamps()
constexpr int STAMP_SLOTS = 16, STAMP_BLOCKS = 16384;
__device__ unsigned long long g_stamps[STAMP_BLOCKS * STAMP_SLOTS];
#define MOCA_STAMP(k)                                                                              \
    do {                                                                                           \
        if (threadIdx.x == 0 && blockIdx.x < STAMP_BLOCKS) {                                       \
            unsigned long long t__;                                                                \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");            \
            g_stamps[blockIdx.x * STAMP_SLOTS + (k)] = t__;                                        \
        }                                                                                          \
    } while (0)
#define MOCA_STAMP_W(k, w)                                                                         \
    do {                                                                                           \
        if (threadIdx.x == (w) * 64 && blockIdx.x < STAMP_BLOCKS) {                                \
            unsigned long long t__;                                                                \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");            \
            g_stamps[blockIdx.x * STAMP_SLOTS + (k)] = t__;                                        \
        }                                                                                          \
    } while (0)
#define MOCA_STAMP_P(k, dep)                                                                       \
    do {                                                                                           \
        int dep__ = (dep);                                                                         \
        asm volatile("" : "+v"(dep__));                                                            \
        if (threadIdx.x == 0 && blockIdx.x < STAMP_BLOCKS) {                                       \
            unsigned long long t__;                                                                \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");            \
            g_stamps[blockIdx.x * STAMP_SLOTS + (k)] = t__ + (dep__ & 0);                          \
        }                                                                                          \
    } while (0)
#define MOCA_STAMP_HW()                                                                            \
    do {                                                                                           \
        if (threadIdx.x == 0 && blockIdx.x < STAMP_BLOCKS) {                                       \
            unsigned hw__, xcc__;                                                                  \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw__));                     \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc__));                   \
            g_stamps[blockIdx.x * STAMP_SLOTS + 6] = hw__;                                         \
            g_stamps[blockIdx.x * STAMP_SLOTS + 7] = xcc__;                                        \
        }                                                                                          \
    } while (0)
#else
#define MOCA_STAMP(k) do {} while (0)
#define MOCA_STAMP_P(k, dep) do {} while (0)
#define MOCA_STAMP_HW() do {} while (0)
#endif

// q = n / d, r = n % d for 0 <= n < 2^24 via one float multiply and a +-1 fix-up (rd = 1.0f / d);
// the row-descriptor set-up of a block does 8-12 of these and sits on its critical path
__device__ __forceinline__ void divmod24(int n, int d, float rd, int& q, int& r) {
    q = (int)((float)n * rd);
    r = n - q * d;
    if (r < 0) { r += d; --q; }
    else if (r >= d) { r -= d; ++q; }
}

// ---- A-operand gather addressing shared by the direct-to-LDS kernels (glds, g4, w80) -----------------------------------
// A lane owns NAP rows of the block tile (one per DMA instruction it issues) and a 16-byte chunk `lch` of the KS-wide k-tile.
// Fast path (every layer of the UNet except the 4-channel input conv): a k-tile lies inside ONE tap (C % 64 == 0) resp. inside
// K (K % 64 == 0), so per tile every lane just adds a block-uniform element offset to a per-row base pointer.  Rows that must
// read zeros (conv halo, t-1/t+1 outside the clip, M tail) have the zero page as base; the zero page is longer than any
// per-tile offset, so no per-tile select is needed.  Bases are recomputed only when the tap changes (every C/KS tiles).
// Slow path (input conv with C = 8, odd K): the source of every (row, tile) is computed from scratch.
template <int AMODE, bool FAST, int NAP, int KS>
struct AGather {
    const moca_gemm_params& p;
    const half_t* Aptr;
    const half_t* zero;
    int lch;
    int64_t row_off[NAP];
    int row_y[NAP], row_x[NAP];
    bool row_ok[NAP];
    const half_t* a_base[NAP];
    int tap_cur = -1, a_koff = 0, tiles_per_tap;

    __device__ __forceinline__ AGather(const moca_gemm_params& p_, int lch_)
        : p(p_), Aptr(reinterpret_cast<const half_t*>(p_.a)), zero(g_zero_page), lch(lch_),
          tiles_per_tap((AMODE == MOCA_A_LINEAR || !FAST) ? (1 << 30) : p_.C / KS) {}

    // row slot g of this lane is output row m (pixel m of the [frame][y][x] raster for the conv modes)
    __device__ __forceinline__ void init_row(int g, int m) {
        row_ok[g] = m < p.M;
        const int mm = row_ok[g] ? m : 0;
        if (AMODE == MOCA_A_LINEAR) {
            row_off[g] = (int64_t)mm * p.lda;
            row_y[g] = row_x[g] = 0;
        } else if (AMODE == MOCA_A_CONV3X3) {
            const int ohw = p.outH * p.outW;
            int f, rem, oy, ox;
            if (p.M < (1 << 24)) {
                divmod24(mm, ohw, 1.0f / (float)ohw, f, rem);
                divmod24(rem, p.outW, 1.0f / (float)p.outW, oy, ox);
            } else {
                f = mm / ohw; rem = mm - f * ohw;
                oy = rem / p.outW; ox = rem - oy * p.outW;
            }
            row_off[g] = (int64_t)f * p.inH * p.inW;
            row_y[g] = oy * p.stride - 1 + p.nopad_lo + (p.up_phase ? (p.up_phase - 1) >> 1 : 0);
            row_x[g] = ox * p.stride - 1 + p.nopad_lo + (p.up_phase ? (p.up_phase - 1) & 1 : 0);
        } else {
            int frame, pix, vid, t;
            if (p.M < (1 << 24)) {
                divmod24(mm, p.HW, 1.0f / (float)p.HW, frame, pix);
                divmod24(frame, p.T, 1.0f / (float)p.T, vid, t);
            } else {
                frame = mm / p.HW; t = frame % p.T;
            }
            row_off[g] = mm;
            row_y[g] = t;
            row_x[g] = 0;
        }
    }

    __device__ __forceinline__ void set_tap(int tap) {
        tap_cur = tap;
        if (AMODE == MOCA_A_LINEAR) {
#pragma unroll
            for (int g = 0; g < NAP; ++g) a_base[g] = row_ok[g] ? Aptr + row_off[g] + lch * 8 : zero;
        } else if (AMODE == MOCA_A_CONV3X3) {
            // (up_phase: one of the four 2 x 2 convs a nearest-x2 upsample + 3 x 3 conv splits into -- tap t = (t >> 1, t & 1))
            const int ky = p.up_phase ? tap >> 1 : tap / 3, kx = p.up_phase ? tap & 1 : tap - ky * 3;
            const int limH = p.up ? 2 * p.inH : p.inH, limW = p.up ? 2 * p.inW : p.inW;
#pragma unroll
            for (int g = 0; g < NAP; ++g) {
                int iy = row_y[g] + ky, ix = row_x[g] + kx;
                const bool ok = tap < (p.up_phase ? 4 : 9) && row_ok[g] && iy >= 0 && iy < limH && ix >= 0 && ix < limW;
                if (p.up) { iy >>= 1; ix >>= 1; }
                const half_t* src = Aptr + (row_off[g] + (int64_t)iy * p.inW + ix) * p.C + lch * 8;
                a_base[g] = ok ? src : zero;
            }
        } else {
#pragma unroll
            for (int g = 0; g < NAP; ++g) {
                const int tt = row_y[g] + tap - 1;
                const bool ok = tap < 3 && row_ok[g] && tt >= 0 && tt < p.T;
                const half_t* src = Aptr + (row_off[g] + (int64_t)(tap - 1) * p.HW) * p.C + lch * 8;
                a_base[g] = ok ? src : zero;
            }
        }
    }

    // block-uniform per-tile state of the DMA stream (fast path): element offset added to every a_base
    __device__ __forceinline__ void begin_tile(int kt) {
        if constexpr (FAST) {
            const int tap = kt / tiles_per_tap;
            if (tap != tap_cur) set_tap(tap);
            a_koff = (kt - tap * tiles_per_tap) * KS;
        }
    }

    __device__ __forceinline__ const half_t* slow_src(int kt, int g) const {
        const int k = kt * KS + lch * 8;
        if (AMODE == MOCA_A_LINEAR) {
            return (k < p.K && row_ok[g]) ? Aptr + row_off[g] + k : zero;
        } else if (AMODE == MOCA_A_CONV3X3) {
            const int tap = k / p.C, c = k - tap * p.C;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int limH = p.up ? 2 * p.inH : p.inH, limW = p.up ? 2 * p.inW : p.inW;
            int iy = row_y[g] + ky, ix = row_x[g] + kx;
            const bool ok = tap < 9 && row_ok[g] && iy >= 0 && iy < limH && ix >= 0 && ix < limW;
            if (p.up) { iy >>= 1; ix >>= 1; }
            return ok ? Aptr + (row_off[g] + (int64_t)iy * p.inW + ix) * p.C + c : zero;
        } else {
            const int tap = k / p.C, c = k - tap * p.C;
            const int tt = row_y[g] + tap - 1;
            const bool ok = tap < 3 && row_ok[g] && tt >= 0 && tt < p.T;
            return ok ? Aptr + (row_off[g] + (int64_t)(tap - 1) * p.HW) * p.C + c : zero;
        }
    }

    // source of row slot g for k-tile kt; `odd` = kt is the second tile of the pair begin_tile() was called for
    __device__ __forceinline__ const half_t* src(int kt, int g, int odd = 0) const {
        if constexpr (FAST) return a_base[g] + a_koff + odd * KS;
        else return slow_src(kt, g);
    }
};

// ---- A-operand gather for BUFFER-addressed LDS-DMA (buffer_load_dwordx4 ... offen lds) --------------------------------------
// Same sources as AGather's fast path, expressed as  descriptor base (SGPRs) + per-lane 32-bit byte offset (one VGPR per
// row slot, recomputed only when the tap changes) + a block-uniform scalar byte offset per k-tile.  The hot loop then has NO
// 64-bit per-lane pointer arithmetic (the flat-address form costs v_lshl_add_u64 / v_mov_b64 pairs per DMA instruction, which
// compete with the MFMAs for the SIMD's issue slots), and rows that must read zeros (conv halo, t-1/t+1 outside the clip, M
// tail) carry an offset >= 2^31 = num_records of the descriptor: the hardware range check then writes ZEROS into LDS
// (tools/micro/buflds.hip verifies both facts on gfx950: out-of-range lanes of an LDS-DMA store 0, soffset is range-checked).
// Requires every in-range byte offset < 2^31 (host-checked) and C % 64 == 0 resp. K % 64 == 0 (the FAST conditions).
constexpr unsigned OOB_OFF = 0x80000000u;
template <int AMODE, int NAP, int KS>
struct BGather {
    const moca_gemm_params& p;
    int lch;
    int64_t row_off[NAP];
    int row_y[NAP], row_x[NAP];
    bool row_ok[NAP];
    unsigned a_off[NAP];          // per-lane byte offset of row slot g for the current tap
    int tap = -1, tin = 0, tiles_per_tap;
    int kt_next, kt_last;         // even tile of the next pair to issue; last pair of this block's k range
    // The k range is walked tap-major (the packing order of W): all channels of tap 0, then tap 1, ...  A channel-major walk (the
    // nine taps of one 64-channel slice before the next slice, so that the taps hit L2) was measured in round 2: -8 % GEMM fabric
    // reads, +2-5 % on some convs in isolation, -0.4 % on the whole step (profiles/r02_ab_conv_channel_major.txt) -- removed.
    // MOCA_A_LINEAR2 (internal: MOCA_A_LINEAR with p.a2): the virtual torch.cat([a, a2], channels) -- two "taps" of k1 / KS and
    // (K - k1) / KS tiles; tap 0 walks the rows of a (row stride lda), tap 1 those of a2 (lda2).  The kernel switches the buffer
    // descriptor with the tap (`tap` is block-uniform); k1 % 64 == 0, so a k-tile pair never straddles the sources.
    __device__ __forceinline__ BGather(const moca_gemm_params& p_, int lch_, int kt_begin, int kt_last_pair)
        : p(p_), lch(lch_), tiles_per_tap(AMODE == MOCA_A_LINEAR ? (1 << 30) : (AMODE == MOCA_A_LINEAR2 ? p_.k1 / KS : p_.C / KS)),
          kt_next(kt_begin), kt_last(kt_last_pair) {}

    __device__ __forceinline__ void init_row(int g, int m) {
        row_ok[g] = m < p.M;
        const int mm = row_ok[g] ? m : 0;
        if (AMODE == MOCA_A_LINEAR) {
            row_off[g] = (int64_t)mm * p.lda;
            row_y[g] = row_x[g] = 0;
        } else if (AMODE == MOCA_A_LINEAR2) {
            row_off[g] = mm;
            row_y[g] = row_x[g] = 0;
        } else if (AMODE == MOCA_A_CONV3X3) {
            const int ohw = p.outH * p.outW;
            int f, rem, oy, ox;
            if (p.M < (1 << 24)) {
                divmod24(mm, ohw, 1.0f / (float)ohw, f, rem);
                divmod24(rem, p.outW, 1.0f / (float)p.outW, oy, ox);
            } else {
                f = mm / ohw; rem = mm - f * ohw;
                oy = rem / p.outW; ox = rem - oy * p.outW;
            }
            row_off[g] = (int64_t)f * p.inH * p.inW;
            row_y[g] = oy * p.stride - 1 + p.nopad_lo + (p.up_phase ? (p.up_phase - 1) >> 1 : 0);
            row_x[g] = ox * p.stride - 1 + p.nopad_lo + (p.up_phase ? (p.up_phase - 1) & 1 : 0);
        } else {
            int frame, pix, vid, t;
            if (p.M < (1 << 24)) {
                divmod24(mm, p.HW, 1.0f / (float)p.HW, frame, pix);
                divmod24(frame, p.T, 1.0f / (float)p.T, vid, t);
            } else {
                frame = mm / p.HW; t = frame % p.T;
            }
            row_off[g] = mm;
            row_y[g] = t;
            row_x[g] = 0;
        }
    }

    __device__ __forceinline__ void set_tap(int t) {
        tap = t;
        if (AMODE == MOCA_A_LINEAR) {
#pragma unroll
            for (int g = 0; g < NAP; ++g) a_off[g] = row_ok[g] ? (unsigned)((row_off[g] + lch * 8) * 2) : OOB_OFF;
        } else if (AMODE == MOCA_A_LINEAR2) {
            const int ld = t ? p.lda2 : p.lda;
            if (t) tiles_per_tap = 1 << 30;                  // (the second source runs to the end of the k range)
#pragma unroll
            for (int g = 0; g < NAP; ++g) a_off[g] = row_ok[g] ? (unsigned)((row_off[g] * ld + lch * 8) * 2) : OOB_OFF;
        } else if (AMODE == MOCA_A_CONV3X3) {
            const int ky = p.up_phase ? t >> 1 : t / 3, kx = p.up_phase ? t & 1 : t - ky * 3;      // (up_phase: see AGather::set_tap)
            const int limH = p.up ? 2 * p.inH : p.inH, limW = p.up ? 2 * p.inW : p.inW;
#pragma unroll
            for (int g = 0; g < NAP; ++g) {
                int iy = row_y[g] + ky, ix = row_x[g] + kx;
                const bool ok = t < (p.up_phase ? 4 : 9) && row_ok[g] && iy >= 0 && iy < limH && ix >= 0 && ix < limW;
                if (p.up) { iy >>= 1; ix >>= 1; }
                a_off[g] = ok ? (unsigned)(((row_off[g] + (int64_t)iy * p.inW + ix) * p.C + lch * 8) * 2) : OOB_OFF;
            }
        } else {
#pragma unroll
            for (int g = 0; g < NAP; ++g) {
                const int tt = row_y[g] + t - 1;
                const bool ok = t < 3 && row_ok[g] && tt >= 0 && tt < p.T;
                a_off[g] = ok ? (unsigned)(((row_off[g] + (int64_t)(t - 1) * p.HW) * p.C + lch * 8) * 2) : OOB_OFF;
            }
        }
    }

    // position the stream on the pair whose even tile is kt (one integer division, prologue only)
    __device__ __forceinline__ void seek(int kt) {
        kt_next = kt;
        const int t = AMODE == MOCA_A_LINEAR2 ? (kt >= tiles_per_tap ? 1 : 0) : kt / tiles_per_tap;
        tin = kt - t * tiles_per_tap;
        set_tap(t);
    }
    // step to the following pair; past the end of the k range the last pair repeats (uniform DMA accounting)
    __device__ __forceinline__ void advance() {
        if (kt_next + 2 <= kt_last) {
            kt_next += 2;
            tin += 2;
            if (tin >= tiles_per_tap) { tin = 0; set_tap(tap + 1); }
        }
    }
    // block-uniform byte offsets of the pair's EVEN tile (the odd one is + KS*2 bytes)
    __device__ __forceinline__ unsigned a_soff() const { return (unsigned)(tin * KS * 2); }
    __device__ __forceinline__ unsigned w_soff() const { return (unsigned)(kt_next * KS * 2); }
};

// ---- epilogue stage 2 of the direct-to-LDS kernels: the fp16 tile staged in LDS (`rows` x `out_bn`, row pitch `pitch` bytes)
//      leaves as 16-byte coalesced stores; the time-embedding row add and the residual are added in fp32 on the way ----
// Output rows of a launch whose output is at least half the 256 MiB Infinity Cache leave with non-temporal stores: the consumer could
// re-read little of them from a cache anyway, and they stop evicting the operands the running blocks share.  Same box, alternating
// (threshold sweep 0 / 64 / 128 / 256 MiB / never): the B = 16 FIFO iteration 220.7 -> 218.2 ms at every threshold <= 256 MiB; the
// B = 2 step 61.85 UNet-steps/s with plain stores, 62.1-62.2 at 64-128 MiB (the 320-channel GEGLU / q|k|v outputs stream), 60.7 with
// `nt` on every output (the consumers of the small outputs find them in the Infinity Cache).
__device__ __forceinline__ bool out_streams(const moca_gemm_params& p) { return p.reserved4_ & 1; }   // (decided by moca_gemm_f16)
__device__ __forceinline__ void st_out8(half_t* ptr, const half8v v, bool nt) {
#if defined(__HIP_DEVICE_COMPILE__)
    // (an `nt` store in one arm of a branch is merged with the plain one and loses the hint: the streaming arm is an instruction of its own)
    if (nt) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(ptr), "v"(v) : "memory");
    else *reinterpret_cast<half8v*>(ptr) = v;
#endif
}

// Residual rows of a launch whose output streams (>= 128 MiB: out_streams) are read with non-temporal loads: they are read once and,
// at that size, come from HBM -- same box, alternating (profiles/r05_ab_nt_loads.txt): the B = 16 `163840 x 640 x 640 +res` linears
// 218 -> 199 us, `655360 x 320 x 320 +res` -1 %; on the B = 2 tensors (served by the Infinity Cache) the hint costs 8 %, hence the size
// rule.  (Non-temporal LDS-DMA of an A operand that is read once measured 13 % slower.)
// (a compile-time choice: a hinted load in one arm of a runtime branch is merged with the plain one and loses the hint, as with stores;
//  the store loops are instantiated for both policies and chosen once per block)
template <bool NT>
__device__ __forceinline__ half8v ld_res8(const half_t* ptr) {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const half8v*>(ptr));
    else return *reinterpret_cast<const half8v*>(ptr);
}

// Row-add / residual operands of the store loops are fetched several chunks AHEAD of the stores.  `out` may alias `residual`, so the
// compiler keeps every load behind the previous iteration's store, and each 16-byte chunk paid a global-load round trip plus the store
// drain (`s_waitcnt vmcnt(0)`).  With operands served by the Infinity Cache (B = 2 forward) that costs nothing measurable; from HBM
// (B = 16: the FIFO iteration, configs[4]) the +residual linears run 3.5-9 % faster with the operands in flight together
// (tools/bench_gemm.py "linear+res", BG_B=16).  A thread only ever reads the addresses it writes itself.
constexpr int EPI_U = 8;

template <int NTHREADS, bool NT>
__device__ __forceinline__ void store_fp16_tile_impl(const moca_gemm_params& p, const char* stage, int pitch, int rows, int out_bn,
                                                int m0, int on0, int tid) {
    const half_t* __restrict__ rowadd = reinterpret_cast<const half_t*>(p.rowadd);
    const half_t* __restrict__ resid = reinterpret_cast<const half_t*>(p.residual);
    constexpr bool nt_out = NT;                        // (= out_streams(p), chosen by the wrapper below)
    const int chunks_per_row = out_bn / 8;
    const int total_chunks = rows * chunks_per_row;
    if (!rowadd && !resid) {
        // up_phase (a, b): GEMM row m = pixel (f, i, j) of the low-resolution grid is output pixel (f, 2i + a, 2j + b) of the upsampled
        // one: row 4 m - 2 (m mod W) + 2 W a + b
        const int upW = p.up_phase ? p.outW : 0, upC = p.up_phase ? 2 * p.outW * ((p.up_phase - 1) >> 1) + ((p.up_phase - 1) & 1) : 0;
        for (int idx = tid; idx < total_chunks; idx += NTHREADS) {
            const int row = idx / chunks_per_row, ch = idx - row * chunks_per_row;
            const int m = m0 + row;
            if (m >= p.M) continue;
            const int64_t mo = upW ? 4 * (int64_t)m - 2 * (m % upW) + upC : m;
            st_out8(reinterpret_cast<half_t*>(p.out) + mo * p.ldo + on0 + ch * 8, *reinterpret_cast<const half8v*>(stage + row * pitch + ch * 16), nt_out);
        }
        return;
    }
    const half8v zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int base = tid; base < total_chunks; base += EPI_U * NTHREADS) {
        half8v ea[EPI_U], er[EPI_U];
        int soff[EPI_U];
        int64_t ooff[EPI_U];
        bool ok[EPI_U];
#pragma unroll
        for (int u = 0; u < EPI_U; ++u) {
            const int idx = base + u * NTHREADS;
            const int row = idx / chunks_per_row, ch = idx - row * chunks_per_row;
            const int m = m0 + row, col = on0 + ch * 8;
            ok[u] = idx < total_chunks && m < p.M;
            soff[u] = row * pitch + ch * 16;
            ooff[u] = (int64_t)m * p.ldo + col;
            ea[u] = zero8; er[u] = zero8;
            if (ok[u]) {
                if (rowadd) ea[u] = *reinterpret_cast<const half8v*>(rowadd + (int64_t)(m / p.rowadd_div) * p.ld_rowadd + col);
                if (resid) er[u] = ld_res8<NT>(resid + (int64_t)m * p.ldr + col);
            }
        }
#pragma unroll
        for (int u = 0; u < EPI_U; ++u) {
            if (!ok[u]) continue;
            half8v h = *reinterpret_cast<const half8v*>(stage + soff[u]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {                     // (same order as ever: + row add, then + residual, in fp32)
                float v = (float)h[j];
                if (rowadd) v += (float)ea[u][j];
                if (resid) v += (float)er[u][j];
                h[j] = (half_t)v;
            }
            st_out8(reinterpret_cast<half_t*>(p.out) + ooff[u], h, nt_out);
        }
    }
}

template <int NTHREADS>
__device__ __forceinline__ void store_fp16_tile(const moca_gemm_params& p, const char* stage, int pitch, int rows, int out_bn,
                                                int m0, int on0, int tid) {
    if (out_streams(p)) store_fp16_tile_impl<NTHREADS, true>(p, stage, pitch, rows, out_bn, m0, on0, tid);
    else store_fp16_tile_impl<NTHREADS, false>(p, stage, pitch, rows, out_bn, m0, on0, tid);
}

// ---- store loop of the 320 x 160 kernels with GroupNorm statistics (MOCA_EP_COLSUM): as store_fp16_tile, but every thread
//      keeps a FIXED 16-byte column chunk (thread -> chunk tid % 20, rows tid / 20 + 25 i) so that it can accumulate the sum and
//      the sum of squares of its 8 columns in registers on the way out (fp32 values after row add / residual, i.e. what the
//      consumer's GroupNorm would read back, before the fp16 rounding); the 25 row subsets are then combined through LDS in a
//      fixed order (deterministic) and the block writes colsum[tile_m][n0 .. n0+160)[2].  `red` = 32 000 B of LDS scratch
//      behind the staged tile.
template <int ROWS, int BNC, bool NT>
__device__ __forceinline__ void store_fp16_tile_colsum_impl(const moca_gemm_params& p, const char* stage, float* red, int pitch,
                                                       int m0, int n0, int tile_m, int tid) {
    constexpr int CPR = BNC / 8, RS = 512 / CPR;          // 320 x 160: 20 chunks per row, 25 row subsets; 160 x 320: 40 and 12
    const half_t* __restrict__ rowadd = reinterpret_cast<const half_t*>(p.rowadd);
    const half_t* __restrict__ resid = reinterpret_cast<const half_t*>(p.residual);
    constexpr bool nt_out = NT;                        // (= out_streams(p), chosen by the wrapper below)
    const int ch = tid % CPR, rs = tid / CPR;
    // (up_phase: rows scattered to the upsampled grid as in store_fp16_tile; a row tile of the low-resolution grid lies inside one frame)
    const int upW = p.up_phase ? p.outW : 0, upC = p.up_phase ? 2 * p.outW * ((p.up_phase - 1) >> 1) + ((p.up_phase - 1) & 1) : 0;
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
    if (rs < RS) {
        const int col = n0 + ch * 8;
        const half8v zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        constexpr int U = 4;                                       // operands fetched U rows ahead of the stores (see store_fp16_tile)
        for (int row0 = rs; row0 < ROWS && m0 + row0 < p.M; row0 += U * RS) {
            half8v ea[U], er[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = row0 + u * RS, m = m0 + row;
                ea[u] = zero8; er[u] = zero8;
                if (row < ROWS && m < p.M) {
                    if (rowadd) ea[u] = *reinterpret_cast<const half8v*>(rowadd + (int64_t)(m / p.rowadd_div) * p.ld_rowadd + col);
                    if (resid) er[u] = ld_res8<NT>(resid + (int64_t)m * p.ldr + col);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = row0 + u * RS, m = m0 + row;
                if (row >= ROWS || m >= p.M) break;
                const half8v h = *reinterpret_cast<const half8v*>(stage + row * pitch + ch * 16);
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[j] = (float)h[j];
                    if (rowadd) v[j] += (float)ea[u][j];
                    if (resid) v[j] += (float)er[u][j];
                }
                half8v o;
#pragma unroll
                for (int j = 0; j < 8; ++j) { o[j] = (half_t)v[j]; s[j] += v[j]; q[j] += v[j] * v[j]; }
                const int64_t mo = upW ? 4 * (int64_t)m - 2 * (m % upW) + upC : m;
                st_out8(reinterpret_cast<half_t*>(p.out) + mo * p.ldo + col, o, nt_out);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[(rs * BNC + ch * 8 + j) * 2] = s[j]; red[(rs * BNC + ch * 8 + j) * 2 + 1] = q[j]; }
    }
    __syncthreads();
    if (p.flags & MOCA_EP_GSTAT) {
        // finished statistics: column totals -> LDS, then one thread per (GroupNorm channel group touched by this tile, sum or sum
        // of squares) adds its columns and issues ONE fixed-point atomic on gstat[statistics group][channel group] (a row tile lies
        // inside one statistics group).  Sums of <= 320 x 40 values per atomic in fp32; the cross-tile accumulation is 64-bit fixed point (order independent, common.h).
        float a[2] = {0.f, 0.f};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = tid + h * 512;
            if (i < 2 * BNC) {
#pragma unroll 4
                for (int r = 0; r < RS; ++r) a[h] += red[r * BNC * 2 + i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = tid + h * 512;
            if (i < 2 * BNC) red[i] = a[h];
        }
        __syncthreads();
        // (gstat_cpg / gstat_coff: this output is one source of a virtual concat and its columns are channels coff + n of the consumer's
        //  tensor, whose groups are cpg wide -- a group at the seam is completed by the other source's producer)
        const int cpg = p.gstat_cpg > 0 ? p.gstat_cpg : p.N / 32, coff = p.gstat_coff;
        const int g0 = (coff + n0) / cpg, g1 = (coff + n0 + BNC - 1) / cpg;
        if (tid < 2 * (g1 - g0 + 1)) {
            const int g = g0 + (tid >> 1), comp = tid & 1;
            const int c0 = max(g * cpg - coff, n0) - n0, c1 = min((g + 1) * cpg - coff, n0 + BNC) - n0;
            float t = 0.f;
            for (int c = c0; c < c1; ++c) t += red[c * 2 + comp];
            const int sg = m0 / p.gstat_rows;
            moca_gstat_add(p.gstat + ((int64_t)sg * 32 + g) * 2 + comp, comp, t);
        }
        return;
    }
    for (int i = tid; i < 2 * BNC; i += 512) {
        float a = 0.f;
#pragma unroll 4
        for (int r = 0; r < RS; ++r) a += red[r * BNC * 2 + i];
        p.colsum[((int64_t)tile_m * p.N + n0) * 2 + i] = a;
    }
}

template <int ROWS, int BNC>
__device__ __forceinline__ void store_fp16_tile_colsum(const moca_gemm_params& p, const char* stage, float* red, int pitch,
                                                       int m0, int n0, int tile_m, int tid) {
    if (out_streams(p)) store_fp16_tile_colsum_impl<ROWS, BNC, true>(p, stage, red, pitch, m0, n0, tile_m, tid);
    else store_fp16_tile_colsum_impl<ROWS, BNC, false>(p, stage, red, pitch, m0, n0, tile_m, tid);
}

// ---- store loop of the 160 x 320 tile with LayerNorm (MOCA_EP_LN, N == 320: the block owns complete rows): 8 lanes per row,
//      lane l8 -> 16-byte chunks l8, l8 + 8, ..., l8 + 32 (40 columns), 64 rows per pass.  Adds the residual in fp32, stores
//      x = the linear's output (fp16) AND ln_out = LayerNorm(x) * gamma + beta (fp16; statistics of the fp32 values, variance
//      two-pass from registers, reduced over the 8 lanes of a row with shuffles).
template <int ROWS = 160>
__device__ __forceinline__ void store_fp16_tile_ln(const moca_gemm_params& p, const char* stage, float* gb, int pitch, int m0, int tid) {
    constexpr int NC = 320, CPL = 5;
    const half_t* __restrict__ resid = reinterpret_cast<const half_t*>(p.residual);
    const half_t* __restrict__ rowadd = reinterpret_cast<const half_t*>(p.rowadd);
    const bool nt_out = out_streams(p);
    const int l8 = tid & 7, rsub = tid >> 3;
    // gamma / beta go through LDS (`gb` = 2 x 320 floats behind the staged tile): 80 more live registers per thread would not
    // fit beside the row
    for (int i = tid; i < 2 * NC; i += 512) gb[i] = i < NC ? p.ln_gamma[i] : p.ln_beta[i - NC];
    __syncthreads();
    for (int row = rsub; row < ROWS; row += 64) {
        const int m = m0 + row;
        const bool ok = m < p.M;                         // (all 8 lanes of a row agree; the shuffles below need every lane)
        float v[CPL][8];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int col = (l8 + 8 * c) * 8;
            const half8v h = *reinterpret_cast<const half8v*>(stage + row * pitch + col * 2);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[c][j] = (float)h[j];
            if (rowadd && ok) {
                const half8v e = *reinterpret_cast<const half8v*>(rowadd + (int64_t)(m / p.rowadd_div) * p.ld_rowadd + col);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[c][j] += (float)e[j];
            }
            if (resid && ok) {
                const half8v e = *reinterpret_cast<const half8v*>(resid + (int64_t)m * p.ldr + col);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[c][j] += (float)e[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[c][j];
        }
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        const float mean = s * (1.0f / NC);
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[c][j] - mean; q += d * d; }
        q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
        const float rstd = rsqrtf(q * (1.0f / NC) + p.ln_eps);
        if (ok) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int col = (l8 + 8 * c) * 8;
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(gb + col), g1 = *reinterpret_cast<const f32x4*>(gb + col + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(gb + NC + col), b1 = *reinterpret_cast<const f32x4*>(gb + NC + col + 4);
                half8v o, n;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = (half_t)v[c][j]; o[4 + j] = (half_t)v[c][4 + j];
                    n[j] = (half_t)((v[c][j] - mean) * rstd * g0[j] + b0[j]);
                    n[4 + j] = (half_t)((v[c][4 + j] - mean) * rstd * g1[j] + b1[j]);
                }
                st_out8(reinterpret_cast<half_t*>(p.out) + (int64_t)m * p.ldo + col, o, nt_out);
                *reinterpret_cast<half8v*>(reinterpret_cast<half_t*>(p.ln_out) + (int64_t)m * p.ld_ln + col) = n;
            }
        }
    }
}

// ---- store loop with LayerNorm statistics of the consumer (MOCA_EP_ROWSUM): as store_fp16_tile, but LPR lanes share a row
//      (lane l -> 16-byte chunks l, l + LPR, ...), so the sum and the sum of squares of the stored (fp16-rounded) values of a
//      row's BNC columns are reduced with shuffles and written to rowsum[column tile][m][2].  The consumer
//      (MOCA_EP_LNFOLD) combines the N / BNC partials of a row. ----
template <int NTHREADS, int BNC, int LPR, bool NT>
__device__ __forceinline__ void store_fp16_tile_rowsum_impl(const moca_gemm_params& p, const char* stage, int pitch, int rows,
                                                       int m0, int n0, int tid) {
    constexpr int CPL = BNC / 8 / LPR;
    static_assert(CPL * LPR * 8 == BNC, "column tile = LPR lanes x CPL chunks of 8");
    const half_t* __restrict__ resid = reinterpret_cast<const half_t*>(p.residual);
    const half_t* __restrict__ rowadd = reinterpret_cast<const half_t*>(p.rowadd);
    constexpr bool nt_out = NT;                        // (= out_streams(p), chosen by the wrapper below)
    const int l = tid % LPR, rsub = tid / LPR;
    float* dst = p.rowsum + (int64_t)(n0 / BNC) * p.M * 2;
    const half8v zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int RSTEP = NTHREADS / LPR;
    half8v ea[CPL], er[CPL], na[CPL], nr[CPL];           // operands of this row pass / of the next one, fetched a pass ahead of the
    auto fetch = [&](int row, half8v* fa, half8v* fr) {  // stores (see store_fp16_tile)
        const int m = m0 + row;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int col = (l + LPR * c) * 8;
            fa[c] = zero8; fr[c] = zero8;
            if (row < rows && m < p.M) {
                if (rowadd) fa[c] = *reinterpret_cast<const half8v*>(rowadd + (int64_t)(m / p.rowadd_div) * p.ld_rowadd + n0 + col);
                if (resid) fr[c] = ld_res8<NT>(resid + (int64_t)m * p.ldr + n0 + col);
            }
        }
    };
    if (rowadd || resid) fetch(rsub, ea, er);
    for (int row = rsub; row < rows; row += RSTEP) {
        const int m = m0 + row;
        const bool ok = m < p.M;                         // (all lanes of a row agree; the shuffles below need every lane)
        float s = 0.f, q = 0.f;
        if (rowadd || resid) fetch(row + RSTEP, na, nr);
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int col = (l + LPR * c) * 8;
            half8v h = *reinterpret_cast<const half8v*>(stage + row * pitch + col * 2);
            if ((rowadd || resid) && ok) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v = (float)h[j];
                    if (rowadd) v += (float)ea[c][j];
                    if (resid) v += (float)er[c][j];
                    h[j] = (half_t)v;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float a = (float)h[j]; s += a; q += a * a; }
            if (ok) st_out8(reinterpret_cast<half_t*>(p.out) + (int64_t)m * p.ldo + n0 + col, h, nt_out);
        }
#pragma unroll
        for (int o = 1; o < LPR; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
        if (l == 0 && ok) *reinterpret_cast<float2*>(dst + (int64_t)m * 2) = float2{s, q};
#pragma unroll
        for (int c = 0; c < CPL; ++c) { ea[c] = na[c]; er[c] = nr[c]; }
    }
}

template <int NTHREADS, int BNC, int LPR>
__device__ __forceinline__ void store_fp16_tile_rowsum(const moca_gemm_params& p, const char* stage, int pitch, int rows,
                                                       int m0, int n0, int tid) {
    if (out_streams(p)) store_fp16_tile_rowsum_impl<NTHREADS, BNC, LPR, true>(p, stage, pitch, rows, m0, n0, tid);
    else store_fp16_tile_rowsum_impl<NTHREADS, BNC, LPR, false>(p, stage, pitch, rows, m0, n0, tid);
}

// ---- MOCA_EP_LNFOLD, step 1: thread t < TM owns the LayerNorm statistics of A row m0 + t, thread t < BN the wsum / bias of
//      column n0 + t.  lnfold_issue() goes BEFORE the prologue's DMA instructions (loads return in order: by the time the first
//      k-tiles have landed these have too, so their latency costs nothing; measured 7-14 us per launch when they were issued
//      behind the DMAs) and keeps the raw values of the first two row partials in registers; lnfold_finish() (after the
//      prologue DMAs are out) adds any further partials and turns the sums into (rstd, -mean * rstd).  Four registers are carried
//      through the main loop; lnfold_publish() puts them into LDS behind the staged tile for the accumulator -> fp16 staging
//      pass (rows past M repeat the last row; their stores are skipped anyway). ----
struct LnFoldRegs { float rs, rb, ws, b; };
struct LnFoldRaw { float2 p0, p1; float ws, b; };
// (`row` = the global A row of this thread's tile row: m0 + tid, or the gathered row of the temporal-attention tiling)
template <int TM, int BN>
__device__ __forceinline__ LnFoldRaw lnfold_issue(const moca_gemm_params& p, int row, int n0, int tid) {
    LnFoldRaw r = {float2{0.f, 0.f}, float2{0.f, 0.f}, 0.f, 0.f};
    if (tid < BN) {
        r.ws = p.lnf_wsum[n0 + tid];
        r.b = p.bias ? p.bias[n0 + tid] : 0.f;
    }
    if (tid < TM) {
        const int m = min(row, p.M - 1);
        r.p0 = *reinterpret_cast<const float2*>(p.lnf_part + (int64_t)m * 2);
        if (p.lnf_nparts > 1) r.p1 = *reinterpret_cast<const float2*>(p.lnf_part + ((int64_t)p.M + m) * 2);
    }
    return r;
}
template <int TM, int BN>
__device__ __forceinline__ LnFoldRegs lnfold_finish(const moca_gemm_params& p, const LnFoldRaw& raw, int row, int tid) {
    LnFoldRegs r = {0.f, 0.f, raw.ws, raw.b};
    if (tid < TM) {
        const int m = min(row, p.M - 1);
        float s = raw.p0.x + raw.p1.x, q = raw.p0.y + raw.p1.y;
        for (int i = 2; i < p.lnf_nparts; ++i) {
            const float2 v = *reinterpret_cast<const float2*>(p.lnf_part + ((int64_t)i * p.M + m) * 2);
            s += v.x; q += v.y;
        }
        const float inv_k = 1.0f / (float)p.K;
        const float mean = s * inv_k;
        const float var = fmaxf(q * inv_k - mean * mean, 0.f);
        r.rs = rsqrtf(var + p.ln_eps);
        r.rb = -mean * r.rs;
    }
    return r;
}
// lds: [TM] float2 (rstd, -mean * rstd), then [BN] wsum, then [BN] bias
template <int TM, int BN>
__device__ __forceinline__ void lnfold_publish(const LnFoldRegs& r, float* lds, int tid) {
    if (tid < TM) *reinterpret_cast<float2*>(lds + 2 * tid) = float2{r.rs, r.rb};
    if (tid < BN) { lds[2 * TM + tid] = r.ws; lds[2 * TM + BN + tid] = r.b; }
    __syncthreads();
}
// step 2, per accumulator tile: v = rstd * acc + (-mean * rstd) * wsum + bias
__device__ __forceinline__ f32x4 lnfold_apply(const f32x4 acc, const float2 st, const f32x4 ws, const f32x4 b) {
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = st.x * acc[j] + (st.y * ws[j] + b[j]);
    return r;
}
__device__ __forceinline__ f32x4 ld4_or_zero(const float* ptr, int i) {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return ptr ? *reinterpret_cast<const f32x4*>(ptr + i) : z;
}

typedef __attribute__((address_space(3))) char* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

template <int BN, int AMODE, bool FAST>
__global__ __launch_bounds__(512, 2) void gemm_glds_kernel(const moca_gemm_params p) {
    constexpr int TM = 256;
    constexpr int NT = BN / 32;                       // 16-wide n tiles per wave
    constexpr int A_BYTES = TM * ROW_BYTES;           // 32 KiB
    constexpr int B_BYTES = BN * ROW_BYTES;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int B_GROUPS = BN / 8;                  // 1 KiB row groups of the W tile (16 or 20)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    MOCA_STAMP(0);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;

    const int tiles_m = (p.M + TM - 1) / TM;
    const int tiles_n = p.N / BN;
    const int nk_total = (p.K + BK - 1) / BK;      // (W rows are readable and zero beyond K up to the next multiple of 64)
    int split, tile, kt_begin, kt_end;
    {
        const int nblk = tiles_m * tiles_n * p.splits;
        if (prefetch_block(p, nblk, 512)) return;
        int logical;
        remap_block<BN>(nblk, logical);
        split = logical % p.splits;
        tile = logical / p.splits;
        const int kts = (nk_total + p.splits - 1) / p.splits;
        kt_begin = split * kts;
        kt_end = min(kt_begin + kts, nk_total);
    }
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int m0 = tile_m * TM, n0 = tile_n * BN;
    const int nk = kt_end - kt_begin;
    MOCA_STAMP_P(8, m0 + n0 + nk);

    const half_t* __restrict__ Wptr = reinterpret_cast<const half_t*>(p.w);

    // ---- DMA coordinates: instruction g of this wave fills row group q = g*8 + wave,
    //      i.e. rows q*8 + (lane>>3), physical chunk lane&7
    const int lrow = lane >> 3, pch = lane & 7;
    const int swz = (((wave * 8 + lrow) >> 1) & 7);   // same for every g (64 g rows apart)
    const int lch = pch ^ swz;                        // logical 16-byte chunk of the k-tile this lane fetches

    AGather<AMODE, FAST, 4, BK> ga(p, lch);
#pragma unroll
    for (int g = 0; g < 4; ++g) ga.init_row(g, m0 + (g * 8 + wave) * 8 + lrow);
    MOCA_STAMP_P(9, (int)ga.row_ok[0]);
    // W rows of this lane: group q = g*8 + wave (valid while q < B_GROUPS)
    const half_t* w_row[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const int q = g * 8 + wave;
        const int n = n0 + (q < B_GROUPS ? q : 0) * 8 + lrow;
        w_row[g] = Wptr + (int64_t)n * p.ldw + lch * 8;
    }

    // DMA piece j (0..3: A row groups, 4..6: W row groups) of tile kt into ring slot `slot`
    auto dma_piece = [&](int kt, int slot, int j) {
        const lds_ptr sa = (lds_ptr)smem + slot * STAGE;
        if (j < 4) {
            const half_t* src;
            src = ga.src(kt, j);
            __builtin_amdgcn_global_load_lds((glb_ptr)src, sa + (j * 8 + wave) * 1024, 16, 0, 0);
        } else {
            const int g = j - 4;
            if (g * 8 + 7 < B_GROUPS || (g * 8 < B_GROUPS && wave < B_GROUPS - g * 8))
                __builtin_amdgcn_global_load_lds((glb_ptr)(w_row[g] + kt * BK), sa + A_BYTES + (g * 8 + wave) * 1024, 16, 0, 0);
        }
    };
    auto issue = [&](int kt, int slot) {
        ga.begin_tile(kt);
#pragma unroll
        for (int j = 0; j < 7; ++j) dma_piece(kt, slot, j);
    };

    f32x4 acc[4][NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    int a_row_b[4], a_swz[4], b_row_b[NT], b_swz[NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int row = wave_m * 64 + mt * 16 + fr;
        a_row_b[mt] = row * ROW_BYTES;
        a_swz[mt] = (row >> 1) & 7;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int row = wave_n * (BN / 2) + nt * 16 + fr;
        b_row_b[nt] = A_BYTES + row * ROW_BYTES;
        b_swz[nt] = (row >> 1) & 7;
    }

    // number of DMA instructions this wave issues per k-tile (for the counted wait)
    const bool extra_w = (B_GROUPS % 8) != 0 && wave < (B_GROUPS % 8);

    // ---- software-pipelined, hand-interleaved main loop ---------------------------------------
    // Two fragment register sets.  Per k-tile i each wave runs
    //   P0: MFMAs on set0 (k 0..31 of tile i), with the ds_reads of k 32..63 (-> set1) and the 6-7 DMA
    //       instructions of tile i+2 dropped one at a time into the gaps between MFMA pairs
    //   sync: counted vmcnt (tile i+1 landed), lgkmcnt(0), ONE s_barrier
    //   P1: MFMAs on set1, with the ds_reads of k 0..31 of tile i+1 (-> set0) in the gaps
    // sched_barrier(0) pins the interleave the source spells out.
    half8v af0[4], bf0[NT], af1[4], bf1[NT];
    auto read_one = [&](const char* st, int ks, int r, half8v (&af)[4], half8v (&bf)[NT]) {
        const int ch = ks * 4 + fg;
        if (r < 4) af[r] = *reinterpret_cast<const half8v*>(st + a_row_b[r] + ((ch ^ a_swz[r]) << 4));
        else bf[r - 4] = *reinterpret_cast<const half8v*>(st + b_row_b[r - 4] + ((ch ^ b_swz[r - 4]) << 4));
    };
    auto wait_tile = [&](bool more_in_flight) {
        // this wave's DMAs of the awaited tile are complete once at most one tile's worth is outstanding
        if (more_in_flight) {
            if (B_GROUPS % 8 == 0) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
            else if (extra_w) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    };
    constexpr int NMMA = 4 * NT;     // MFMAs per half k-tile
    constexpr int NRD = 4 + NT;      // fragment reads per half k-tile

    LnFoldRaw lraw = {float2{0.f, 0.f}, float2{0.f, 0.f}, 0.f, 0.f};
    if (p.flags & MOCA_EP_LNFOLD) lraw = lnfold_issue<TM, BN>(p, m0 + tid, n0, tid);
    MOCA_STAMP_P(10, a_row_b[0] + b_row_b[0]);
    if (nk > 0) issue(kt_begin, 0);
    MOCA_STAMP_P(11, 0);
    if (nk > 1) issue(kt_begin + 1, 1);
    LnFoldRegs lf = {0.f, 0.f, 0.f, 0.f};
    if (p.flags & MOCA_EP_LNFOLD) lf = lnfold_finish<TM, BN>(p, lraw, m0 + tid, tid);
    MOCA_STAMP(1);
    if (nk > 0) {
        wait_tile(nk > 1);
        MOCA_STAMP(2);
#pragma unroll
        for (int r = 0; r < NRD; ++r) read_one(smem, 0, r, af0, bf0);
    }
    int s_cur = 0, s_nxt = 1, s_far = 2;      // ring slots of tiles i, i+1, i+2 (rotated, no modulo in the loop)
    // one k-tile step; ISSUE (compile time) = tile i+2 exists and its DMA pieces go into the gaps of P0
    auto step = [&](auto issue_tag, int i, bool has_next) {
        constexpr bool do_issue = decltype(issue_tag)::value;
        const char* cur = smem + s_cur * STAGE;
        const int kt2 = kt_begin + i + 2, slot2 = s_far;     // ring slot (i-1)%3: every wave passed sync(i-1) after its last read of it
        if constexpr (do_issue) ga.begin_tile(kt2);
        // ---- P0 ----
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < NMMA; ++j) {
            const int mt = j / NT, nt = j % NT;
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf0[nt], af0[mt], acc[mt][nt], 0, 0, 0);
            if (j & 1) {
                const int r = j >> 1;
                if (r < NRD) read_one(cur, 1, r, af1, bf1);
                if constexpr (do_issue) { if (r >= 1 && r <= 7) dma_piece(kt2, slot2, r - 1); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        // ---- sync ----
        if (has_next) wait_tile(do_issue);
        const char* nxt = smem + (has_next ? s_nxt : s_cur) * STAGE;
        // ---- P1 ----
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < NMMA; ++j) {
            const int mt = j / NT, nt = j % NT;
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf1[nt], af1[mt], acc[mt][nt], 0, 0, 0);
            if (j & 1) {
                const int r = j >> 1;
                if (r < NRD) read_one(nxt, 0, r, af0, bf0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        { const int t = s_cur; s_cur = s_nxt; s_nxt = s_far; s_far = t; }
    };
    int i = 0;
    for (; i + 2 < nk; ++i) step(yes_t{}, i, true);        // steady state: branch-free DMA issue
    for (; i < nk; ++i) step(no_t{}, i, i + 1 < nk);       // last two tiles: nothing left to prefetch
    MOCA_STAMP(3);
    __syncthreads();   // all fragment reads done before the ring is reused by the epilogue

    // The MFMAs above compute the TRANSPOSED tile (W fragment as A operand), so accumulator element r
    // of tile (mt, nt) is row m = wave_m*64 + mt*16 + fr, column n = wave_n*BN/2 + nt*16 + 4*fg + r:
    // each lane owns 4 CONSECUTIVE output columns of one row -> 8-byte fp16 / 16-byte fp32 accesses.
    if (p.splits > 1) {
        if (p.reserved4_ & 4) {
            // fp16 slabs (MOCA_TUNE_SLAB_F16, A/B only): the partial tile is rounded to fp16, staged like the fp16 epilogue and leaves as whole rows
            constexpr int pitch16 = BN * 2 + 16;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = wave_n * (BN / 2) + nt * 16 + 4 * fg;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int row = wave_m * 64 + mt * 16 + fr;
                    *reinterpret_cast<half4v*>(smem + row * pitch16 + col * 2) = __builtin_convertvector(acc[mt][nt], half4v);
                }
            }
            __syncthreads();
            half_t* ws16 = reinterpret_cast<half_t*>(p.splitk_ws) + (int64_t)split * p.M * p.N;
            constexpr int cpr = BN / 8;
            for (int idx = tid; idx < TM * cpr; idx += 512) {
                const int row = idx / cpr, ch = idx - row * cpr;
                const int m = m0 + row;
                if (m < p.M) *reinterpret_cast<half8v*>(ws16 + (int64_t)m * p.N + n0 + ch * 8) = *reinterpret_cast<const half8v*>(smem + row * pitch16 + ch * 16);
            }
            return;
        }
        float* ws = p.splitk_ws + (int64_t)split * p.M * p.N;
        // the fp32 partial tile is staged in LDS and leaves as whole 4 BN-byte rows (round 5: straight from registers it left as 64-byte
        // segments, 16 rows x 4 lanes per instruction -- 1.2-1.4 us slower per launch at the 5 x 8-latent level, profiles/r05_ab_slab_store.txt)
        float* sC = reinterpret_cast<float*>(smem);
        constexpr int PF = BN == 128 ? BN + 4 : BN;           // floats per staged row (+16 B where the 160 KiB allow it: BN = 160 fills them)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = wave_n * (BN / 2) + nt * 16 + 4 * fg;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int row = wave_m * 64 + mt * 16 + fr;
                *reinterpret_cast<f32x4*>(sC + row * PF + col) = acc[mt][nt];
            }
        }
        __syncthreads();
        constexpr int cpr4 = BN / 4;
        for (int idx = tid; idx < TM * cpr4; idx += 512) {
            const int row = idx / cpr4, ch = idx - row * cpr4;
            const int m = m0 + row;
            if (m < p.M) *reinterpret_cast<f32x4*>(ws + (int64_t)m * p.N + n0 + ch * 4) = *reinterpret_cast<const f32x4*>(sC + row * PF + ch * 4);
        }
        return;
    }

    const bool geglu = (p.flags & MOCA_EP_GEGLU) != 0;
    const int out_bn = geglu ? BN / 2 : BN;
    const int on0 = geglu ? n0 / 2 : n0;
    const half_t* __restrict__ rowadd = reinterpret_cast<const half_t*>(p.rowadd);
    const half_t* __restrict__ resid = reinterpret_cast<const half_t*>(p.residual);

    if (p.flags & MOCA_EP_OUT_F32) {
        // fp32 output: stage the tile in fp32 (no intermediate rounding)
        float* sC = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = wave_n * (BN / 2) + nt * 16 + 4 * fg;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + n0 + col);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int row = wave_m * 64 + mt * 16 + fr;
                *reinterpret_cast<f32x4*>(sC + row * BN + col) = acc[mt][nt] + bv;
            }
        }
        __syncthreads();
        const int cpr = BN / 4;
        for (int idx = tid; idx < TM * cpr; idx += 512) {
            const int row = idx / cpr, ch = idx - row * cpr;
            const int m = m0 + row;
            if (m >= p.M) continue;
            const int col = n0 + ch * 4;
            f32x4 v = *reinterpret_cast<const f32x4*>(sC + row * BN + ch * 4);
            if (rowadd) {
                const half4v e = *reinterpret_cast<const half4v*>(rowadd + (int64_t)(m / p.rowadd_div) * p.ld_rowadd + col);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += (float)e[j];
            }
            if (resid) {
                const half4v e = *reinterpret_cast<const half4v*>(resid + (int64_t)m * p.ldr + col);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += (float)e[j];
            }
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + col) = v;
        }
        return;
    }

    // fp16 output: stage fp16 rows (pitch = out_bn*2 + 16 bytes: conflict-free ds_write_b64 / ds_read_b128)
    const int pitch = out_bn * 2 + 16;
    if (geglu) {
        if constexpr (NT == 4) {
            // wave span of 64 packed columns = 32 value columns (tiles 0,1) then their 32 gate columns (tiles 2,3)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int ncol = n0 + wave_n * 64 + nt * 16 + 4 * fg;
                f32x4 bv = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) { bv = *reinterpret_cast<const f32x4*>(p.bias + ncol); bg = *reinterpret_cast<const f32x4*>(p.bias + ncol + 32); }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int row = wave_m * 64 + mt * 16 + fr;
                    const f32x4 va = acc[mt][nt] + bv, ga = acc[mt][nt + 2] + bg;
                    const f32x2 lo = moca_geglu2(f32x2{va[0], va[1]}, f32x2{ga[0], ga[1]});
                    const f32x2 hi = moca_geglu2(f32x2{va[2], va[3]}, f32x2{ga[2], ga[3]});
                    half4v h;
                    h[0] = (half_t)lo[0]; h[1] = (half_t)lo[1]; h[2] = (half_t)hi[0]; h[3] = (half_t)hi[1];
                    *reinterpret_cast<half4v*>(smem + row * pitch + (wave_n * 32 + nt * 16 + 4 * fg) * 2) = h;
                }
            }
        }
    } else {
        const bool fold = (p.flags & MOCA_EP_LNFOLD) != 0;     // Linear(LayerNorm(x)) from x (no GEGLU / split-k on this kernel)
        float* rst = reinterpret_cast<float*>(smem + TM * (BN * 2 + 16));
        if (fold) lnfold_publish<TM, BN>(lf, rst, tid);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = wave_n * (BN / 2) + nt * 16 + 4 * fg;
            f32x4 bv, ws4 = {0.f, 0.f, 0.f, 0.f};
            if (fold) { ws4 = *reinterpret_cast<const f32x4*>(rst + 2 * TM + col); bv = *reinterpret_cast<const f32x4*>(rst + 2 * TM + BN + col); }
            else bv = ld4_or_zero(p.bias, n0 + col);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int row = wave_m * 64 + mt * 16 + fr;
                f32x4 v;
                if (fold) v = lnfold_apply(acc[mt][nt], *reinterpret_cast<const float2*>(rst + 2 * row), ws4, bv);
                else v = acc[mt][nt] + bv;
                *reinterpret_cast<half4v*>(smem + row * pitch + col * 2) = __builtin_convertvector(v, half4v);
            }
        }
    }
    __syncthreads();
    MOCA_STAMP(4);

    // (MOCA_EP_COLSUM / MOCA_EP_ROWSUM are only accepted without GEGLU / split-k: out_bn == BN; 256 x 272 B or 256 x 336 B of
    //  staged rows + the 32 x BN x 2 / 25 x BN x 2 floats of row-subset sums fit inside the 144 / 156 KiB ring)
    if (p.flags & (MOCA_EP_COLSUM | MOCA_EP_GSTAT)) store_fp16_tile_colsum<TM, BN>(p, smem, reinterpret_cast<float*>(smem + TM * (BN * 2 + 16)), pitch, m0, n0, tile_m, tid);
    else if (p.flags & MOCA_EP_ROWSUM) store_fp16_tile_rowsum<512, BN, 4>(p, smem, pitch, TM, m0, n0, tid);
    else store_fp16_tile<512>(p, smem, pitch, TM, out_bn, m0, on0, tid);
    MOCA_STAMP(5);
    MOCA_STAMP_HW();
}

#ifdef MOCA_STAMPS
}  // namespace
extern "C" int moca_debug_stamps(unsigned long long* host_out, int n_blocks) {
    if (n_blocks > STAMP_BLOCKS) n_blocks = STAMP_BLOCKS;
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), (size_t)n_blocks * STAMP_SLOTS * 8) == hipSuccess ? 0 : -2;
}
namespace {
#endif


// =====================================================================================
// "g4" kernel: 256 x 128 x 32 block tile, 256 threads = 4 wavefronts (2 x 2), wave tile 128 x 64 as 8 x 4
// v_mfma_f32_16x16x32_f16 accumulators, 3-slot direct-to-LDS ring of 24 KiB k-tiles = 72 KiB per block, so
// TWO blocks are resident per CU (8 waves, 2 per SIMD, <= 256 VGPRs each).  The two blocks run different
// tiles out of phase: one block's prologue (first-DMA latency), barriers and epilogue (bias/GEGLU/stores)
// sit under the other block's MFMAs -- the structural stall of the 8-wave kernel above, where the single
// resident block leaves the matrix pipe idle in those phases (s_memtime stamps: main loop only 35-42 % of a
// K = 320 tile).  One 32-deep k-tile = one MFMA k-step: per tile 32 MFMAs, 12 fragment ds_read_b128 and 6 DMA
// pieces per wave, one s_barrier; fragments are double-buffered in registers, the DMA runs three tiles ahead.
// LDS rows are 64 B; chunk swizzle phys = chunk ^ f((row>>2)&3), f = {0,2,3,1} (conflict-free for the four
// 16-lane groups of ds_read_b128 on the 16x16x32 operand map; derivation in DESIGN.md).
// =====================================================================================
template <int AMODE, bool FAST>
__global__ __launch_bounds__(256, 2) void gemm_g4_kernel(const moca_gemm_params p) {
    constexpr int TM = 256, BN = 128, KS = 32;
    constexpr int RB = KS * 2;                         // 64-byte LDS rows
    constexpr int MT = 8, NT = 4;
    constexpr int A_BYTES = TM * RB;                   // 16 KiB
    constexpr int B_BYTES = BN * RB;                   // 8 KiB
    constexpr int STAGE = A_BYTES + B_BYTES;           // 24 KiB
    constexpr int NMMA = MT * NT;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    MOCA_STAMP(0);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;

    const int tiles_m = (p.M + TM - 1) / TM;
    const int tiles_n = p.N / BN;
    const int nblk = tiles_m * tiles_n * p.splits;
    if (prefetch_block(p, nblk, 256)) return;
    int split = 0, tile_m, tile_n;
    if ((p.reserved4_ >> 8) > 1) {                       // 2-D XCD partition (see remap_tile_2d)
        remap_tile_2d(tiles_m, tiles_n, p.reserved4_ >> 8, tile_m, tile_n);
    } else {
        int logical;
        remap_block<BN>(nblk, logical);
        split = logical % p.splits;
        const int tile = logical / p.splits;
        tile_m = tile / tiles_n; tile_n = tile % tiles_n;
    }
    const int m0 = tile_m * TM, n0 = tile_n * BN;

    const int nk_total = 2 * ((p.K + 63) / 64);        // 64-deep units -> even
    const int kts = 2 * (((p.K + 63) / 64 + p.splits - 1) / p.splits);   // 64-deep units, as the host sizes the splits -> even
    const int kt_begin = split * kts;
    const int nk = min(kt_begin + kts, nk_total) - kt_begin;

    const half_t* __restrict__ Wptr = reinterpret_cast<const half_t*>(p.w);

    // DMA piece = 1 KiB = 16 rows x 64 B: lane -> row (lane>>2), physical chunk lane&3; A piece g of this
    // wave covers rows (g*4 + wave)*16 .. +15 (g = 0..3), W piece g rows (g*4 + wave)*16 .. (g = 0..1)
    const int lrow = lane >> 2, pch = lane & 3;
    const int lch = pch ^ ((0x78 >> (2 * ((lrow >> 2) & 3))) & 3);     // logical chunk fetched into this lane's slot
    AGather<AMODE, FAST, 4, KS> ga(p, lch);
#pragma unroll
    for (int g = 0; g < 4; ++g) ga.init_row(g, m0 + (g * 4 + wave) * 16 + lrow);
    const half_t* w_row[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int n = n0 + (g * 4 + wave) * 16 + lrow;
        w_row[g] = Wptr + (int64_t)n * p.ldw + lch * 8;
    }

    // DMA piece j (0..3: A, 4..5: W) of tile kt (= the tile begin_tile() was called for, + odd) into ring slot `slot`
    auto dma_piece = [&](int kt, int slot, int j, int odd) {
        const lds_ptr sa = (lds_ptr)smem + slot * STAGE;
        if (j < 4) {
            __builtin_amdgcn_global_load_lds((glb_ptr)ga.src(kt, j, odd), sa + (j * 4 + wave) * 1024, 16, 0, 0);
        } else {
            const int g = j - 4;
            __builtin_amdgcn_global_load_lds((glb_ptr)(w_row[g] + kt * KS), sa + A_BYTES + (g * 4 + wave) * 1024, 16, 0, 0);
        }
    };
    // k-tiles travel in (even, odd) pairs: the two 64-byte halves of every 128-byte line are requested back to back
    // (one phase apart the 32 KiB vector L1 has turned over and every line crosses L2 -> L1 twice: +10-22 % on the short-K
    // linears, measured); the pair's 12 DMA instructions are interleaved, so a pair lands as a unit
    auto issue_pair = [&](int kt_even, int slot_even, int slot_odd) {
        ga.begin_tile(kt_even);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            dma_piece(kt_even, slot_even, j, 0);
            dma_piece(kt_even + 1, slot_odd, j, 1);
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    // fragment byte offsets inside a slot (row base + swizzled chunk): constant per lane
    int a_off[MT], b_off[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = wave_m * 128 + mt * 16 + fr;
        a_off[mt] = row * RB + ((fg ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3)) << 4);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int row = wave_n * 64 + nt * 16 + fr;
        b_off[nt] = A_BYTES + row * RB + ((fg ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3)) << 4);
    }

    // ONE fragment set: the LDS-read latency at the head of a phase is covered by the co-resident block's
    // wave on the same SIMD (the two blocks of a CU run out of phase), not by a second register set --
    // 128 accumulator + 48 fragment registers leave room for two blocks per CU.
    half8v af[MT], bf[NT];
    int s0 = 0, s1 = 1, s2 = 2;                 // ring slots of tiles i, i+1, i+2 (rotating registers, no modulo)
    const int kt_last_pair = kt_begin + nk - 2;  // nk is even
    // phase i: read tile i's fragments, 32 MFMAs.  EVEN phases also issue the pair (i+2, i+3): tile i+2 into the slot of
    // tile i-1, tile i+3 into tile i's own slot -- hence the barrier after the fragment reads (every wave has tile i in
    // registers before any wave's DMA overwrites it).  The pair must be complete at the end of the following odd phase.
    auto phase = [&](auto even_tag, int i) {
        constexpr bool even = decltype(even_tag)::value;
        const char* cur = smem + s0 * STAGE;
        const int kt2 = min(kt_begin + i + 2, kt_last_pair);
        if constexpr (even) ga.begin_tile(kt2);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const half8v*>(cur + a_off[mt]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = *reinterpret_cast<const half8v*>(cur + b_off[nt]);
        if constexpr (even) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < NMMA; ++j) {
            const int mt = j / NT, nt = j % NT;
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[nt], af[mt], acc[mt][nt], 0, 0, 0);   // D^T: lane = row m
            if constexpr (even) {
                if (j % 5 == 2 && j / 5 < 6) {              // after MFMA 2, 7, ..., 27: piece j/5 of both tiles of the pair
                    dma_piece(kt2, s2, j / 5, 0);
                    dma_piece(kt2 + 1, s0, j / 5, 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        { const int t = s0; s0 = s1; s1 = s2; s2 = t; }
    };

    // ---- prologue: pair (0, 1) ----
    LnFoldRaw lraw = {float2{0.f, 0.f}, float2{0.f, 0.f}, 0.f, 0.f};
    if (p.flags & MOCA_EP_LNFOLD) lraw = lnfold_issue<TM, BN>(p, m0 + tid, n0, tid);
    issue_pair(kt_begin, 0, 1);
    LnFoldRegs lf = {0.f, 0.f, 0.f, 0.f};
    if (p.flags & MOCA_EP_LNFOLD) lf = lnfold_finish<TM, BN>(p, lraw, m0 + tid, tid);
    MOCA_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    MOCA_STAMP(2);
    for (int i = 0; i < nk; i += 2) {
        phase(yes_t{}, i);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // tile i+1 landed with its partner; everyone is past tile i's MFMAs
        __builtin_amdgcn_s_barrier();
        phase(no_t{}, i + 1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // pair (i+2, i+3) complete
        __builtin_amdgcn_s_barrier();
    }
    // (the last barrier also ends every fragment read: the ring is free for the epilogue)
    MOCA_STAMP(3);

    // ---- epilogue (same scheme as the 8-wave kernel): lane owns 4 consecutive columns of row
    //      m = wave_m*128 + mt*16 + fr: column n = wave_n*64 + nt*16 + 4*fg + r ----
    if (p.splits > 1) {
        float* ws = p.splitk_ws + (int64_t)split * p.M * p.N;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = m0 + wave_m * 128 + mt * 16 + fr;
            if (row < p.M) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = n0 + wave_n * 64 + nt * 16 + 4 * fg;
                    *reinterpret_cast<f32x4*>(ws + (int64_t)row * p.N + col) = acc[mt][nt];
                }
            }
        }
        return;
    }
    const bool geglu = (p.flags & MOCA_EP_GEGLU) != 0;
    const int out_bn = geglu ? BN / 2 : BN;
    const int on0 = geglu ? n0 / 2 : n0;
    const int pitch = out_bn * 2 + 16;
    const bool fold = (p.flags & MOCA_EP_LNFOLD) != 0;         // Linear(LayerNorm(x)) from x: row statistics -> LDS behind the staged tile
    float* rst = reinterpret_cast<float*>(smem + TM * (BN * 2 + 16));
    if (fold) lnfold_publish<TM, BN>(lf, rst, tid);
    if (geglu) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int lcol = wave_n * 64 + nt * 16 + 4 * fg, ncol = n0 + lcol;
            f32x4 bv, bg, wv = {0.f, 0.f, 0.f, 0.f}, wg = wv;
            if (fold) {
                wv = *reinterpret_cast<const f32x4*>(rst + 2 * TM + lcol); wg = *reinterpret_cast<const f32x4*>(rst + 2 * TM + lcol + 32);
                bv = *reinterpret_cast<const f32x4*>(rst + 2 * TM + BN + lcol); bg = *reinterpret_cast<const f32x4*>(rst + 2 * TM + BN + lcol + 32);
            } else {
                bv = ld4_or_zero(p.bias, ncol); bg = ld4_or_zero(p.bias, ncol + 32);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int row = wave_m * 128 + mt * 16 + fr;
                f32x4 va, ga;
                if (fold) {
                    const float2 st = *reinterpret_cast<const float2*>(rst + 2 * row);
                    va = lnfold_apply(acc[mt][nt], st, wv, bv); ga = lnfold_apply(acc[mt][nt + 2], st, wg, bg);
                } else {
                    va = acc[mt][nt] + bv; ga = acc[mt][nt + 2] + bg;
                }
                const f32x2 lo = moca_geglu2(f32x2{va[0], va[1]}, f32x2{ga[0], ga[1]});
                const f32x2 hi = moca_geglu2(f32x2{va[2], va[3]}, f32x2{ga[2], ga[3]});
                half4v h;
                h[0] = (half_t)lo[0]; h[1] = (half_t)lo[1]; h[2] = (half_t)hi[0]; h[3] = (half_t)hi[1];
                *reinterpret_cast<half4v*>(smem + row * pitch + (wave_n * 32 + nt * 16 + 4 * fg) * 2) = h;
            }
        }
    } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = wave_n * 64 + nt * 16 + 4 * fg;
            f32x4 bv, ws4 = {0.f, 0.f, 0.f, 0.f};
            if (fold) { ws4 = *reinterpret_cast<const f32x4*>(rst + 2 * TM + col); bv = *reinterpret_cast<const f32x4*>(rst + 2 * TM + BN + col); }
            else bv = ld4_or_zero(p.bias, n0 + col);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int row = wave_m * 128 + mt * 16 + fr;
                f32x4 v;
                if (fold) v = lnfold_apply(acc[mt][nt], *reinterpret_cast<const float2*>(rst + 2 * row), ws4, bv);
                else v = acc[mt][nt] + bv;
                *reinterpret_cast<half4v*>(smem + row * pitch + col * 2) = __builtin_convertvector(v, half4v);
            }
        }
    }
    __syncthreads();
    MOCA_STAMP(4);
    store_fp16_tile<256>(p, smem, pitch, TM, out_bn, m0, on0, tid);
    MOCA_STAMP(5);
    MOCA_STAMP_HW();
}


// =====================================================================================
// "g4p" kernel: the two-blocks-per-CU 256 x 128 x 32 structure of g4 as a PERSISTENT kernel for the linears (A mode LINEAR).
// Phase stamps of g4 / the 256 x 256 staggered kernel on the GEGLU projection of the 320-channel level
// (profiles/r05_g4_phase_stamps.txt): of a tile's ~30 k cycles the main loop is 42-44 %; 10-16 % go to the prologue (row set-up,
// first DMA round trip), 30-35 % to the accumulator -> LDS staging pass with the erf-GELU, 8-10 % to the store loop.  Here
//   * a block walks its tiles (XCD-local order: the blocks of an XCD work on consecutive tiles of the tile_m-major raster, so an A
//     row tile is fetched from beyond L2 once and then hit by the blocks that own its other column tiles);
//   * the DMA stream never stops: the last even phase of a tile issues the first k-tile pair of the NEXT tile (per-lane offsets
//     switched just before), so there is no prologue except the block's first;
//   * the ring is never used for staging: W rows are fetched into the LDS tile in a PERMUTED order (free -- the source address of
//     an LDS-DMA is per lane) chosen so that the 4 + 4 accumulator columns a lane holds in two neighbouring 16-column MFMA tiles are
//     8 CONSECUTIVE output columns: the epilogue goes from registers to memory as 16-byte stores (16 rows x 64 B per instruction),
//     no LDS pass, no barrier, and it runs while the next tile's first pair is in flight;
//   * LayerNorm-fold statistics of a tile (row: rstd, -mean rstd; column: wsum, bias) are fetched at the tile's start and
//     published in 3 KiB of LDS behind the ring.
// Buffer-addressed DMA (SGPR descriptors, 32-bit per-lane offsets, out-of-range = zero fill) as in the staggered kernels.
// =====================================================================================
// LDS row rho (0..127) of the W tile <- packed W row n0 + g4p_perm(rho): MFMA tile nt = (rho >> 4) & 3, column j = rho & 15 of wave
// column wn = rho >> 6 holds output column  wn * 64 + (nt >> 1) * 32 + 8 (j >> 2) + 4 (nt & 1) + (j & 3)  of the tile
__device__ __forceinline__ int g4p_perm(int rho) {
    const int wn = rho >> 6, nt = (rho >> 4) & 3, j = rho & 15;
    return wn * 64 + (nt >> 1) * 32 + 8 * (j >> 2) + 4 * (nt & 1) + (j & 3);
}
__device__ __forceinline__ void st_out8_nt(half_t* ptr, const half8v v, bool nt) { st_out8(ptr, v, nt); }

template <bool GEGLU>
__global__ __launch_bounds__(256, 2) void gemm_g4p_kernel(const moca_gemm_params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TM = 256, BN = 128, KS = 32, RB = 64, MT = 8, NT = 4;
    constexpr int A_BYTES = TM * RB, STAGE = A_BYTES + BN * RB, RING = 3 * STAGE;     // 24 KiB per k-tile, 72 KiB ring
    constexpr int NMMA = MT * NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lst = reinterpret_cast<float*>(smem + RING);          // [TM] float2 (rstd, -mean rstd), [BN] wsum, [BN] bias  (MFMA column order)
    float* lws = lst + 2 * TM;
    float* lbi = lws + BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int nper = p.reserved2_;                               // persistent blocks (multiple of 8); the rest of the grid prefetches
    if (prefetch_block(p, nper, 256)) return;

    const int tiles_n = p.N / BN;
    const int ntiles = ((p.M + TM - 1) / TM) * tiles_n;
    // tiles of this block: XCD x = b & 7 owns a contiguous range of the raster, its J = nper / 8 blocks walk it with stride J
    int t_cur, t_end;
    const int J = nper >> 3;
    {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int q = ntiles >> 3, r = ntiles & 7;
        const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        t_cur = start + j;
        t_end = start + q + (x < r ? 1 : 0);
    }
    if (t_cur >= t_end) return;                                  // (block-uniform)
    const int nk = 2 * (p.K / 64);                               // K % 64 == 0 (host-checked)

    // ---- DMA addressing: piece = 16 rows x 64 B, lane -> row lane >> 2, physical chunk lane & 3 ----
    const int lrow = lane >> 2, pch = lane & 3;
    const int lch = pch ^ ((0x78 >> (2 * ((lrow >> 2) & 3))) & 3);
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, OOB_OFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, OOB_OFF, 0x00020000);
    unsigned a_off[4], w_off[2];
    int w_perm[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) w_perm[g] = g4p_perm((g * 4 + wave) * 16 + lrow);
    auto set_dma_tile = [&](int t) {                             // t < 0: no tile (every lane out of range: zero fill, no traffic)
        const int tm = t / tiles_n, tn = t - tm * tiles_n;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = tm * TM + (g * 4 + wave) * 16 + lrow;
            a_off[g] = (t >= 0 && row < p.M) ? (unsigned)(((int64_t)row * p.lda + lch * 8) * 2) : OOB_OFF;
        }
#pragma unroll
        for (int g = 0; g < 2; ++g)
            w_off[g] = t >= 0 ? (unsigned)(((int64_t)(tn * BN + w_perm[g]) * p.ldw + lch * 8) * 2) : OOB_OFF;
    };
    auto dma_piece = [&](int kt, int slot, int j) {              // piece j (0..3: A, 4..5: W) of k-tile kt of the DMA tile
        const lds_ptr sa = (lds_ptr)smem + slot * STAGE;
        const unsigned soff = (unsigned)(kt * KS * 2);
        if (j < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, sa + (j * 4 + wave) * 1024, 16, a_off[j], soff, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, sa + A_BYTES + ((j - 4) * 4 + wave) * 1024, 16, w_off[j - 4], soff, 0, 0);
    };

    const int fr = lane & 15, fg = lane >> 4;
    int fa_off[MT], fb_off[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = wave_m * 128 + mt * 16 + fr;
        fa_off[mt] = row * RB + ((fg ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3)) << 4);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int row = wave_n * 64 + nt * 16 + fr;
        fb_off[nt] = A_BYTES + row * RB + ((fg ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3)) << 4);
    }

    f32x4 acc[MT][NT];
    half8v af[MT], bf[NT];
    int s0 = 0, s1 = 1, s2 = 2;                                  // ring slots of k-tiles i, i+1, i+2 of the running stream
    // one phase = one k-tile: fragments -> registers, 32 MFMAs; EVEN phases issue the pair (kt2, kt2 + 1) of the DMA tile: kt2 into the
    // slot of k-tile i - 1, kt2 + 1 into k-tile i's own slot (hence the barrier behind the fragment reads)
    auto phase = [&](auto even_tag, int kt2) {
        constexpr bool even = decltype(even_tag)::value;
        const char* cur = smem + s0 * STAGE;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const half8v*>(cur + fa_off[mt]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = *reinterpret_cast<const half8v*>(cur + fb_off[nt]);
        if constexpr (even) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < NMMA; ++j) {
            const int mt = j / NT, nt = j % NT;
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[nt], af[mt], acc[mt][nt], 0, 0, 0);   // D^T: lane = row m
            if constexpr (even) {
                if (j % 5 == 2 && j / 5 < 6) {
                    dma_piece(kt2, s2, j / 5);
                    dma_piece(kt2 + 1, s0, j / 5);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        { const int t = s0; s0 = s1; s1 = s2; s2 = t; }
    };

    // statistics of tile t: raw values -> registers (issued early), finished and published behind the tile's main loop
    const bool fold = (p.flags & MOCA_EP_LNFOLD) != 0;
    float2 rp0, rp1;
    float rws, rbi;
    auto stat_issue = [&](int t) {
        const int tm = t / tiles_n, tn = t - tm * tiles_n;
        rp0 = float2{0.f, 0.f}; rp1 = float2{0.f, 0.f}; rws = 0.f; rbi = 0.f;
        if (tid < BN) {
            const int n = tn * BN + g4p_perm(tid);
            if (fold) rws = p.lnf_wsum[n];
            if (p.bias) rbi = p.bias[n];
        }
        if (fold) {
            const int m = min(tm * TM + tid, p.M - 1);
            rp0 = *reinterpret_cast<const float2*>(p.lnf_part + (int64_t)m * 2);
            if (p.lnf_nparts > 1) rp1 = *reinterpret_cast<const float2*>(p.lnf_part + ((int64_t)p.M + m) * 2);
        }
    };
    auto stat_publish = [&](int t) {
        float rs = 1.f, rb = 0.f;
        if (fold) {
            const int tm = t / tiles_n;
            const int m = min(tm * TM + tid, p.M - 1);
            float s = rp0.x + rp1.x, q = rp0.y + rp1.y;
            for (int i = 2; i < p.lnf_nparts; ++i) {
                const float2 v = *reinterpret_cast<const float2*>(p.lnf_part + ((int64_t)i * p.M + m) * 2);
                s += v.x; q += v.y;
            }
            const float inv_k = 1.0f / (float)p.K;
            const float mean = s * inv_k;
            const float var = fmaxf(q * inv_k - mean * mean, 0.f);
            rs = rsqrtf(var + p.ln_eps);
            rb = -mean * rs;
        }
        *reinterpret_cast<float2*>(lst + 2 * tid) = float2{rs, rb};
        if (tid < BN) { lws[tid] = rws; lbi[tid] = rbi; }
    };

    // ---- prologue: first pair of the first tile ----
    stat_issue(t_cur);
    set_dma_tile(t_cur);
#pragma unroll
    for (int j = 0; j < 6; ++j) { dma_piece(0, 0, j); dma_piece(1, 1, j); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const bool nt_out = out_streams(p);
    const half_t* __restrict__ resid = reinterpret_cast<const half_t*>(p.residual);
    const half_t* __restrict__ rowadd = reinterpret_cast<const half_t*>(p.rowadd);
    int tile_no = 0;
    (void)tile_no;
    while (true) {
        const int t_next = t_cur + J < t_end ? t_cur + J : -1;
        const int tm = t_cur / tiles_n, tn = t_cur - tm * tiles_n;
        const int m0 = tm * TM, n0 = tn * BN;
#ifdef MOCA_STAMPS
        const bool stamp_it = tile_no == 2;
        if (stamp_it) MOCA_STAMP(0);
        if (stamp_it) MOCA_STAMP(1);
        if (stamp_it) MOCA_STAMP(2);
#endif
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < nk; i += 2) {
            const bool last = i + 2 >= nk;
            if (last) set_dma_tile(t_next);                      // the stream moves on to the next tile's first pair
            phase(yes_t{}, last ? 0 : i + 2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            phase(no_t{}, 0);
            if (!last) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // pair (i+2, i+3) complete
                __builtin_amdgcn_s_barrier();
            }
        }
#ifdef MOCA_STAMPS
        if (stamp_it) MOCA_STAMP(3);
#endif
        stat_publish(t_cur);
        if (t_next >= 0) stat_issue(t_next);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#ifdef MOCA_STAMPS
        if (stamp_it) MOCA_STAMP(4);
#endif

        // ---- epilogue from registers: lane = row m0 + wave_m * 128 + mt * 16 + fr, 8 consecutive columns per tile pair ----
        f32x4 cw[NT], cb[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            cw[nt] = *reinterpret_cast<const f32x4*>(lws + wave_n * 64 + nt * 16 + 4 * fg);
            cb[nt] = *reinterpret_cast<const f32x4*>(lbi + wave_n * 64 + nt * 16 + 4 * fg);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int rl = wave_m * 128 + mt * 16 + fr;
            const int m = m0 + rl;
            const float2 st = *reinterpret_cast<const float2*>(lst + 2 * rl);
            f32x4 v[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) v[nt] = st.x * acc[mt][nt] + (st.y * cw[nt] + cb[nt]);
            if constexpr (GEGLU) {
                // value tiles 0, 1 and their gate tiles 2, 3 -> output columns on0 + wave_n * 32 + 8 fg + (0..7)
                half8v h;
#pragma unroll
                for (int nv = 0; nv < 2; ++nv) {
                    const f32x2 lo = moca_geglu2(f32x2{v[nv][0], v[nv][1]}, f32x2{v[nv + 2][0], v[nv + 2][1]});
                    const f32x2 hi = moca_geglu2(f32x2{v[nv][2], v[nv][3]}, f32x2{v[nv + 2][2], v[nv + 2][3]});
                    h[4 * nv + 0] = (half_t)lo[0]; h[4 * nv + 1] = (half_t)lo[1]; h[4 * nv + 2] = (half_t)hi[0]; h[4 * nv + 3] = (half_t)hi[1];
                }
                if (m < p.M) st_out8(reinterpret_cast<half_t*>(p.out) + (int64_t)m * p.ldo + (n0 >> 1) + wave_n * 32 + 8 * fg, h, nt_out);
            } else {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int col = n0 + wave_n * 64 + hh * 32 + 8 * fg;
                    float o[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { o[r] = v[2 * hh][r]; o[4 + r] = v[2 * hh + 1][r]; }
                    if (m < p.M) {
                        if (rowadd) {
                            const half8v e = *reinterpret_cast<const half8v*>(rowadd + (int64_t)(m / p.rowadd_div) * p.ld_rowadd + col);
#pragma unroll
                            for (int r = 0; r < 8; ++r) o[r] = (float)(half_t)o[r] + (float)e[r];   // (as the staged kernels: fp16 tile, then + row add, + residual in fp32)
                        }
                        if (resid) {
                            const half8v e = *reinterpret_cast<const half8v*>(resid + (int64_t)m * p.ldr + col);
#pragma unroll
                            for (int r = 0; r < 8; ++r) o[r] = (rowadd ? o[r] : (float)(half_t)o[r]) + (float)e[r];
                        }
                        half8v h;
#pragma unroll
                        for (int r = 0; r < 8; ++r) h[r] = (half_t)o[r];
                        st_out8(reinterpret_cast<half_t*>(p.out) + (int64_t)m * p.ldo + col, h, nt_out);
                    }
                }
            }
        }
#ifdef MOCA_STAMPS
        if (stamp_it) { MOCA_STAMP(5); MOCA_STAMP_HW(); }
        ++tile_no;
#endif
        if (t_next < 0) break;
        t_cur = t_next;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");           // the next tile's first pair has landed (and the stores have left)
        __builtin_amdgcn_s_barrier();
    }
#endif
}

// "g4q": g4p on v_mfma_f32_32x32x16_f16 (wave tile 128 x 64 = 4 x 2 tiles).  The kernel is bound by the SIMDs' issue port (an MFMA of
// either shape holds it for 8 cycles): half as many MFMA instructions for the same matrix cycles, the same fragment bytes.
// W-row permutation for this operand map: LDS row rho -> tile nt = (rho >> 5) & 1 of wave column wn = rho >> 6, MFMA row i = rho & 31 =
// 8 g + 4 h + r (g = accumulator register group, h = lane >> 5) holds column 16 (g >> 1) + 8 h + 4 (g & 1) + r of the tile's 32.
__device__ __forceinline__ int g4q_perm(int rho) {
    const int wn = rho >> 6, nt = (rho >> 5) & 1, i = rho & 31;
    const int g = i >> 3, h = (i >> 2) & 1, r = i & 3;
    return wn * 64 + nt * 32 + 16 * (g >> 1) + 8 * h + 4 * (g & 1) + r;
}
template <bool GEGLU>
__global__ __launch_bounds__(256, 2) void gemm_g4q_kernel(const moca_gemm_params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TM = 256, BN = 128, KS = 32, RB = 64, MT = 4, NT = 2;      // 32 x 32 MFMA tiles per wave (128 x 64)
    constexpr int A_BYTES = TM * RB, STAGE = A_BYTES + BN * RB, RING = 3 * STAGE;     // 24 KiB per k-tile, 72 KiB ring
        extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lst = reinterpret_cast<float*>(smem + RING);          // [TM] float2 (rstd, -mean rstd), [BN] wsum, [BN] bias  (MFMA column order)
    float* lws = lst + 2 * TM;
    float* lbi = lws + BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int nper = p.reserved2_;                               // persistent blocks (multiple of 8); the rest of the grid prefetches
    if (prefetch_block(p, nper, 256)) return;

    const int tiles_n = p.N / BN;
    const int ntiles = ((p.M + TM - 1) / TM) * tiles_n;
    // tiles of this block: XCD x = b & 7 owns a contiguous range of the raster, its J = nper / 8 blocks walk it with stride J
    int t_cur, t_end;
    const int J = nper >> 3;
    {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int q = ntiles >> 3, r = ntiles & 7;
        const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        t_cur = start + j;
        t_end = start + q + (x < r ? 1 : 0);
    }
    if (t_cur >= t_end) return;                                  // (block-uniform)
    const int nk = 2 * (p.K / 64);                               // K % 64 == 0 (host-checked)

    // ---- DMA addressing: piece = 16 rows x 64 B, lane -> row lane >> 2, physical chunk lane & 3 ----
    const int lrow = lane >> 2, pch = lane & 3;
    const int lch = pch ^ ((0x78 >> (2 * ((lrow >> 2) & 3))) & 3);
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, OOB_OFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, OOB_OFF, 0x00020000);
    unsigned a_off[4], w_off[2];
    int w_perm[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) w_perm[g] = g4q_perm((g * 4 + wave) * 16 + lrow);
    auto set_dma_tile = [&](int t) {                             // t < 0: no tile (every lane out of range: zero fill, no traffic)
        const int tm = t / tiles_n, tn = t - tm * tiles_n;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = tm * TM + (g * 4 + wave) * 16 + lrow;
            a_off[g] = (t >= 0 && row < p.M) ? (unsigned)(((int64_t)row * p.lda + lch * 8) * 2) : OOB_OFF;
        }
#pragma unroll
        for (int g = 0; g < 2; ++g)
            w_off[g] = t >= 0 ? (unsigned)(((int64_t)(tn * BN + w_perm[g]) * p.ldw + lch * 8) * 2) : OOB_OFF;
    };
    auto dma_piece = [&](int kt, int slot, int j) {              // piece j (0..3: A, 4..5: W) of k-tile kt of the DMA tile
        const lds_ptr sa = (lds_ptr)smem + slot * STAGE;
        const unsigned soff = (unsigned)(kt * KS * 2);
        if (j < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, sa + (j * 4 + wave) * 1024, 16, a_off[j], soff, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, sa + A_BYTES + ((j - 4) * 4 + wave) * 1024, 16, w_off[j - 4], soff, 0, 0);
    };

    const int fr = lane & 31, fh = lane >> 5;                    // 32x32x16 operand map: row lane & 31, k = 8 (lane >> 5) .. + 7
    int fa_off[MT][2], fb_off[NT][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = wave_m * 128 + mt * 32 + fr;
            fa_off[mt][ks] = row * RB + (((2 * ks + fh) ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3)) << 4);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int row = wave_n * 64 + nt * 32 + fr;
            fb_off[nt][ks] = A_BYTES + row * RB + (((2 * ks + fh) ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3)) << 4);
        }
    }

    f32x16 acc[MT][NT];
    half8v af[MT][2], bf[NT][2];
    int s0 = 0, s1 = 1, s2 = 2;                                  // ring slots of k-tiles i, i+1, i+2 of the running stream
    // one phase = one k-tile: fragments -> registers, 32 MFMAs; EVEN phases issue the pair (kt2, kt2 + 1) of the DMA tile: kt2 into the
    // slot of k-tile i - 1, kt2 + 1 into k-tile i's own slot (hence the barrier behind the fragment reads)
    auto phase = [&](auto even_tag, int kt2) {
        constexpr bool even = decltype(even_tag)::value;
        const char* cur = smem + s0 * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bf[nt][ks] = *reinterpret_cast<const half8v*>(cur + fb_off[nt][ks]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt][ks] = *reinterpret_cast<const half8v*>(cur + fa_off[mt][ks]);
        }
        if constexpr (even) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 2 * MT * NT; ++j) {
            const int ks = j / (MT * NT), mt = (j / NT) % MT, nt = j % NT;
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[nt][ks], af[mt][ks], acc[mt][nt], 0, 0, 0);   // D^T: lane & 31 = row m
            if constexpr (even) {
                if ((j & 1) && j / 2 < 6) {
                    dma_piece(kt2, s2, j / 2);
                    dma_piece(kt2 + 1, s0, j / 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        { const int t = s0; s0 = s1; s1 = s2; s2 = t; }
    };

    // statistics of tile t: raw values -> registers (issued early), finished and published behind the tile's main loop
    const bool fold = (p.flags & MOCA_EP_LNFOLD) != 0;
    float2 rp0, rp1;
    float rws, rbi;
    auto stat_issue = [&](int t) {
        const int tm = t / tiles_n, tn = t - tm * tiles_n;
        rp0 = float2{0.f, 0.f}; rp1 = float2{0.f, 0.f}; rws = 0.f; rbi = 0.f;
        if (tid < BN) {
            const int n = tn * BN + g4q_perm(tid);
            if (fold) rws = p.lnf_wsum[n];
            if (p.bias) rbi = p.bias[n];
        }
        if (fold) {
            const int m = min(tm * TM + tid, p.M - 1);
            rp0 = *reinterpret_cast<const float2*>(p.lnf_part + (int64_t)m * 2);
            if (p.lnf_nparts > 1) rp1 = *reinterpret_cast<const float2*>(p.lnf_part + ((int64_t)p.M + m) * 2);
        }
    };
    auto stat_publish = [&](int t) {
        float rs = 1.f, rb = 0.f;
        if (fold) {
            const int tm = t / tiles_n;
            const int m = min(tm * TM + tid, p.M - 1);
            float s = rp0.x + rp1.x, q = rp0.y + rp1.y;
            for (int i = 2; i < p.lnf_nparts; ++i) {
                const float2 v = *reinterpret_cast<const float2*>(p.lnf_part + ((int64_t)i * p.M + m) * 2);
                s += v.x; q += v.y;
            }
            const float inv_k = 1.0f / (float)p.K;
            const float mean = s * inv_k;
            const float var = fmaxf(q * inv_k - mean * mean, 0.f);
            rs = rsqrtf(var + p.ln_eps);
            rb = -mean * rs;
        }
        *reinterpret_cast<float2*>(lst + 2 * tid) = float2{rs, rb};
        if (tid < BN) { lws[tid] = rws; lbi[tid] = rbi; }
    };

    // ---- prologue: first pair of the first tile ----
    stat_issue(t_cur);
    set_dma_tile(t_cur);
#pragma unroll
    for (int j = 0; j < 6; ++j) { dma_piece(0, 0, j); dma_piece(1, 1, j); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const bool nt_out = out_streams(p);
    const half_t* __restrict__ resid = reinterpret_cast<const half_t*>(p.residual);
    const half_t* __restrict__ rowadd = reinterpret_cast<const half_t*>(p.rowadd);
    while (true) {
        const int t_next = t_cur + J < t_end ? t_cur + J : -1;
        const int tm = t_cur / tiles_n, tn = t_cur - tm * tiles_n;
        const int m0 = tm * TM, n0 = tn * BN;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
        for (int i = 0; i < nk; i += 2) {
            const bool last = i + 2 >= nk;
            if (last) set_dma_tile(t_next);                      // the stream moves on to the next tile's first pair
            phase(yes_t{}, last ? 0 : i + 2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            phase(no_t{}, 0);
            if (!last) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // pair (i+2, i+3) complete
                __builtin_amdgcn_s_barrier();
            }
        }
        stat_publish(t_cur);
        if (t_next >= 0) stat_issue(t_next);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        // ---- epilogue from registers: lane = row m0 + wave_m * 128 + mt * 32 + (lane & 31); accumulator registers 4 g + r of a tile
        //      are (permuted W rows) columns 16 (g >> 1) + 8 (lane >> 5) + 4 (g & 1) + r of the tile's 32 ----
        // (the 16 column constants of a lane stay in registers for the GEGLU form; the plain form, which also holds row add / residual
        //  operands, re-reads them from LDS per use)
        f32x4 cw[NT][4], cb[NT][4];
        if constexpr (GEGLU) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    cw[nt][g] = *reinterpret_cast<const f32x4*>(lws + wave_n * 64 + nt * 32 + 8 * g + 4 * fh);
                    cb[nt][g] = *reinterpret_cast<const f32x4*>(lbi + wave_n * 64 + nt * 32 + 8 * g + 4 * fh);
                }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int rl = wave_m * 128 + mt * 32 + fr;
            const int m = m0 + rl;
            const float2 st = *reinterpret_cast<const float2*>(lst + 2 * rl);
#pragma unroll
            for (int s = 0; s < 2; ++s) {                        // store s: registers g = 2 s, 2 s + 1 -> 8 consecutive columns 16 s + 8 fh
                f32x4 v[NT][2];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {
                        const int g = 2 * s + gg;
                        const f32x4 a4 = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
                        if constexpr (!GEGLU) {
                            cw[nt][g] = *reinterpret_cast<const f32x4*>(lws + wave_n * 64 + nt * 32 + 8 * g + 4 * fh);
                            cb[nt][g] = *reinterpret_cast<const f32x4*>(lbi + wave_n * 64 + nt * 32 + 8 * g + 4 * fh);
                        }
                        v[nt][gg] = st.x * a4 + (st.y * cw[nt][g] + cb[nt][g]);
                    }
                if constexpr (GEGLU) {
                    half8v h;
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {
                        const f32x2 lo = moca_geglu2(f32x2{v[0][gg][0], v[0][gg][1]}, f32x2{v[1][gg][0], v[1][gg][1]});
                        const f32x2 hi = moca_geglu2(f32x2{v[0][gg][2], v[0][gg][3]}, f32x2{v[1][gg][2], v[1][gg][3]});
                        h[4 * gg + 0] = (half_t)lo[0]; h[4 * gg + 1] = (half_t)lo[1]; h[4 * gg + 2] = (half_t)hi[0]; h[4 * gg + 3] = (half_t)hi[1];
                    }
                    if (m < p.M) st_out8(reinterpret_cast<half_t*>(p.out) + (int64_t)m * p.ldo + (n0 >> 1) + wave_n * 32 + 16 * s + 8 * fh, h, nt_out);
                } else {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int col = n0 + wave_n * 64 + nt * 32 + 16 * s + 8 * fh;
                        float o[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { o[r] = v[nt][0][r]; o[4 + r] = v[nt][1][r]; }
                        if (m < p.M) {
                            if (rowadd) {
                                const half8v e = *reinterpret_cast<const half8v*>(rowadd + (int64_t)(m / p.rowadd_div) * p.ld_rowadd + col);
#pragma unroll
                                for (int r = 0; r < 8; ++r) o[r] = (float)(half_t)o[r] + (float)e[r];
                            }
                            if (resid) {
                                const half8v e = *reinterpret_cast<const half8v*>(resid + (int64_t)m * p.ldr + col);
#pragma unroll
                                for (int r = 0; r < 8; ++r) o[r] = (rowadd ? o[r] : (float)(half_t)o[r]) + (float)e[r];
                            }
                            half8v h;
#pragma unroll
                            for (int r = 0; r < 8; ++r) h[r] = (half_t)o[r];
                            st_out8(reinterpret_cast<half_t*>(p.out) + (int64_t)m * p.ldo + col, h, nt_out);
                        }
                    }
                }
            }
        }
        if (t_next < 0) break;
        t_cur = t_next;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");           // the next tile's first pair has landed (and the stores have left)
        __builtin_amdgcn_s_barrier();
    }
#endif
}


// cross-lane steps of the in-epilogue temporal attention as VALU lane swaps (v_permlane16_swap / v_permlane32_swap: with both operands
// equal to x, the two results hold x and its partner 16 / 32 lanes away in every lane -- cdna_hip_programming.md T12 / T21) instead of
// ds_bpermute round trips through the LDS pipe; and the V^T operand by the transposing LDS read (attention.hip: tr_read)
typedef short tq_short4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) tq_short4v* tq_lds_s4_ptr;
__device__ __forceinline__ half4v tattn_tr_read(const char* addr) {
#if defined(__HIP_DEVICE_COMPILE__)
    const tq_short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tq_lds_s4_ptr)addr);
    return __builtin_bit_cast(half4v, v);
#else
    return half4v{0, 0, 0, 0};
#endif
}
#if defined(__HIP_DEVICE_COMPILE__)
#define MOCA_XLANE(NAME, SWAP, OP)                                                                         \
    __device__ __forceinline__ float NAME(float x) {                                                       \
        const auto r = SWAP(__float_as_uint(x), __float_as_uint(x), false, false);                         \
        const float a = __uint_as_float(r[0]), b = __uint_as_float(r[1]);                                  \
        return OP;                                                                                         \
    }
MOCA_XLANE(xlane16_max, __builtin_amdgcn_permlane16_swap, fmaxf(a, b))
MOCA_XLANE(xlane32_max, __builtin_amdgcn_permlane32_swap, fmaxf(a, b))
MOCA_XLANE(xlane16_sum, __builtin_amdgcn_permlane16_swap, a + b)
MOCA_XLANE(xlane32_sum, __builtin_amdgcn_permlane32_swap, a + b)
#undef MOCA_XLANE
#else
__device__ __forceinline__ float xlane16_max(float x) { return x; }
__device__ __forceinline__ float xlane32_max(float x) { return x; }
__device__ __forceinline__ float xlane16_sum(float x) { return x; }
__device__ __forceinline__ float xlane32_sum(float x) { return x; }
#endif

// =====================================================================================
// "w80s" kernel: the 320 x 160 x 32 tile / 80 x 80 wave tile / 5-slot ring of w80b, with the main loop cut into
// LOAD and MFMA SEGMENTS and the two waves of every SIMD running half an iteration apart.
//
// Why.  In w80/w80b every wave interleaves its own ds_reads and DMA instructions with its own MFMAs and all eight waves
// run in lockstep (one barrier per k-tile), so the two waves of a SIMD want the matrix pipe at the same moments and issue
// their memory instructions at the same moments.  Ablations on the same device (tools/ab_run1.sh): without the DMA
// instructions the conv loop runs 20 % faster, without the fragment reads 18 %, without both 39 % (1.6-1.8 PFLOP/s) --
// the costs ADD, i.e. nothing overlaps them; moving the DMA addressing to SGPR descriptors (w80b) or dropping 8 of 9 A
// fetches changes nothing, so it is neither the address arithmetic nor the L2->LDS bytes: it is the in-order issue of
// each wave.  Here a wave's MFMAs issue back to back from registers (25 per k-tile, nothing in between) while its SIMD
// partner is in its LOAD segment (fragment reads for two k-tiles, or the 8 DMA instructions of a k-tile pair, and the
// counted waits), and vice versa: waves 4..7 pass one extra barrier before the loop and waves 0..3 one after it, so the
// halves stay exactly one barrier apart (MI355X_MICROARCH.md, "Two waves per SIMD", items 5 and 9).
//
// One iteration = two k-tiles (i, i+1), four barriers:
//   LOADe : 10 ds_read_b128 -- the fragments of tile i;  lgkmcnt(0);  barrier
//   MFMAe : 25 MFMAs of tile i with the 10 fragment reads of tile i+1 (second register set) in the gaps;  lgkmcnt(0);  barrier
//   LOADo : 8 DMA instructions = the pair (i+4, i+5) into the slots of tiles i-1 and i (both in registers everywhere: the
//           partner half finished its LOADe reads before the barrier this half has just passed);  vmcnt(8): the pair
//           (i+2, i+3) of this wave has landed;  barrier
//   MFMAo : 25 MFMAs of tile i+1;  barrier
// A DMA has two k-tiles of time to land (as in w80); data is read two barriers after the wait that retires it.
// =====================================================================================
// WIDE = false: 320 x 160 block tile, waves 4 (M) x 2 (N).  WIDE = true ("w80t"): 160 x 320, waves 2 x 4 -- the same 80 x 80 wave
// tiles, ring and DMA stream with the roles of A and W swapped (10 A pieces + 20 W pieces per k-tile): a block then owns
// COMPLETE output rows when N = 320, so A is fetched once instead of once per 160-column tile and the LayerNorm that follows the
// attention / projection linears of the 320-channel level (attention.py:199-201,216-219) runs in the store loop (MOCA_EP_LN).
// SHAPE 2 ("sq256"): 256 x 256 block tile, waves 4 x 2, wave tile 64 x 128 (4 x 8 MFMA tiles, 32 MFMAs per k-tile) -- the same
// staggered structure for the wide projections (GEGLU N = 2560 / 5120 / 10240, QKV N = 3840): 7.8 B of L2->LDS traffic per kFLOP
// against 9.4 for 320 x 160 and 11.7 for 256 x 128; 16 + 16 DMA pieces per k-tile = 4 per wave, no repeats; the ring takes all
// 160 KiB of LDS.  GEGLU epilogue: a wave's 128 packed columns = two 64-column groups of 32 value + 32 gate columns.
template <int AMODE, int SHAPE>
__global__ __launch_bounds__(512, 2) void gemm_w80s_kernel(const moca_gemm_params p) {
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass only needs the launch stub; __amdgpu_buffer_rsrc_t is a device-only type)
    constexpr bool WIDE = SHAPE == 1, SQ = SHAPE == 2, TQ = SHAPE == 3;
    constexpr int MT = SQ ? 4 : 5, NT = SQ ? 8 : (TQ ? 6 : 5), KS = 32, RB = 64;
    constexpr int WTM = 16 * MT, WTN = 16 * NT;          // wave tile: 80 x 80, 64 x 128, or 80 x 96
    constexpr int TM = SQ ? 256 : (WIDE ? 160 : 320), BN = SQ ? 256 : (WIDE ? 320 : (TQ ? 192 : 160));
    constexpr int A_BYTES = TM * RB, STAGE = A_BYTES + BN * RB;     // 30 KiB per k-tile (32 KiB for 256 x 256 and 320 x 192)
    constexpr int NS = 5;
    constexpr int PPW = 4;                               // DMA instructions per wave per k-tile (30 pieces + 2 repeats; 32 pieces)
    constexpr int NAP = SQ ? 2 : (WIDE ? 2 : 3);

    extern __shared__ __attribute__((aligned(16))) char smem[];
    MOCA_STAMP(0);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = WIDE ? wave >> 2 : wave >> 1, wave_n = WIDE ? wave & 3 : wave & 1;
    const bool late = wave >= 4;                         // the half of the workgroup that runs one barrier behind

    const int tiles_m = (p.M + TM - 1) / TM;
    const int tiles_n = p.N / BN;
    const int nblk = tiles_m * tiles_n * p.splits;
    if (prefetch_block(p, nblk, 512)) return;
    int split = 0, tile_m, tile_n;
    const int xcd_n = p.reserved4_ >> 8;                 // > 1: 2-D XCD partition (host: splits == 1, both tile counts divide)
    if (xcd_n > 1) {
        remap_tile_2d(tiles_m, tiles_n, xcd_n, tile_m, tile_n);
    } else {
        int logical;
        remap_block<BN>(nblk, logical);
        split = logical % p.splits;
        const int tile = logical / p.splits;
        tile_m = tile / tiles_n; tile_n = tile % tiles_n;
    }
    const int m0 = tile_m * TM, n0 = tile_n * BN;

    const int nk_total = 2 * ((p.K + 63) / 64);
    const int kts = 2 * (((p.K + 63) / 64 + p.splits - 1) / p.splits);
    const int kt_begin = split * kts;
    const int nk = min(kt_begin + kts, nk_total) - kt_begin;

    // DMA pieces (16 rows x 64 B each; lane -> row lane >> 2, physical chunk lane & 3).  The operand with 20 pieces ("big": A of
    // the tall tile, W of the wide one) and the one with 10 ("small") are spread over the 8 waves as in w80 / w80b:
    //   j = 0: big piece w      j = 1: big piece 8 + w      j = 2: big piece 16 + w (w < 4)  or  small piece w - 4 (w >= 4)
    //   j = 3: small piece 4 + w (w < 6)  or  small piece 2 + w (w = 6, 7: a repeat, so that every wave issues 4 per k-tile)
    const int lrow = lane >> 2, pch = lane & 3;
    const int lch = pch ^ ((0x78 >> (2 * ((lrow >> 2) & 3))) & 3);
    const bool flex_is_big = wave < 4;
    const int kt_last_pair = kt_begin + nk - 2;
    BGather<AMODE, NAP, KS> ga(p, lch, kt_begin, kt_last_pair);
    const int small1 = TQ ? 4 + wave : (wave < 6 ? 4 + wave : 2 + wave);    // j = 3 (320 x 192: 12 W pieces, no repeats)
    // SHAPE 3 ("tq", MOCA_EP_TATTN): the rows of an M tile are the 16 frames of 20 neighbouring pixels of one video, gathered by
    // row index -- tile row r = 16 * (pixel - pix0) + frame -- so that the block holds q, k, v of one head for whole temporal
    // sequences and finishes the temporal attention (attention.py:331-352) in its epilogue.  A DMA piece (16 tile rows) = the 16
    // frames of one pixel.
    const int tq_tpv = TQ ? p.HW / 20 : 1;                                   // row tiles per video
    const int tq_b = tile_m / tq_tpv, tq_pb = tile_m - tq_b * tq_tpv;
    auto grow = [&](int tr) -> int { return TQ ? (tq_b * 16 + (tr & 15)) * p.HW + tq_pb * 20 + (tr >> 4) : m0 + tr; };
    // MOCA per-row-group weights (moca_gemm_params.wgroup_rows: GroupNorm folded into the linear that consumes it): the rows of
    // a tile lie inside one group (host-checked), whose W / bias start wg x stride further -- an offset, nothing in the main loop
    const int wg = (AMODE == MOCA_A_LINEAR && p.wgroup_rows > 0) ? m0 / p.wgroup_rows : 0;
    const unsigned wgo = (unsigned)((int64_t)wg * p.wgroup_stride * 2);
    unsigned w_off[3];
    if constexpr (SQ) {                                  // A pieces w and 8 + w, W pieces w and 8 + w
        ga.init_row(0, m0 + wave * 16 + lrow);
        ga.init_row(1, m0 + (8 + wave) * 16 + lrow);
        w_off[0] = (unsigned)(((int64_t)(n0 + wave * 16 + lrow) * p.ldw + lch * 8) * 2);
        w_off[1] = (unsigned)(((int64_t)(n0 + (8 + wave) * 16 + lrow) * p.ldw + lch * 8) * 2);
        w_off[2] = 0;
    } else if constexpr (!WIDE) {
#pragma unroll
        for (int g = 0; g < NAP; ++g) ga.init_row(g, grow((g < 2 ? g * 8 + wave : 16 + (wave & 3)) * 16 + lrow));
        w_off[0] = (unsigned)(((int64_t)(n0 + (wave & 3) * 16 + lrow) * p.ldw + lch * 8) * 2);      // j = 2 (waves 4..7)
        w_off[1] = (unsigned)(((int64_t)(n0 + small1 * 16 + lrow) * p.ldw + lch * 8) * 2);           // j = 3
        w_off[2] = 0;
    } else {
        ga.init_row(0, m0 + (wave & 3) * 16 + lrow);                                                   // j = 2 (waves 4..7)
        ga.init_row(1, m0 + small1 * 16 + lrow);                                                       // j = 3
        w_off[0] = (unsigned)(((int64_t)(n0 + wave * 16 + lrow) * p.ldw + lch * 8) * 2);             // j = 0
        w_off[1] = (unsigned)(((int64_t)(n0 + (8 + wave) * 16 + lrow) * p.ldw + lch * 8) * 2);       // j = 1
        w_off[2] = (unsigned)(((int64_t)(n0 + (16 + (wave & 3)) * 16 + lrow) * p.ldw + lch * 8) * 2); // j = 2 (waves 0..3)
    }
    if (wgo) { w_off[0] += wgo; w_off[1] += wgo; w_off[2] += wgo; }
    // (two-source A, MOCA_A_LINEAR2: the A descriptor follows the gather's block-uniform source index -- `sync_src()` behind every
    //  ga.seek() / ga.advance(); one scalar compare per k-tile pair, the other modes compile to the constant descriptors)
    __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, OOB_OFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, OOB_OFF, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrc_f = (flex_is_big != WIDE) ? rsrc_a : rsrc_w;            // descriptor of this wave's j = 2 piece
    auto sync_src = [&]() {
        if constexpr (AMODE == MOCA_A_LINEAR2) {
            if (ga.tap == 1) {
                rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a2), 0, OOB_OFF, 0x00020000);
                if (flex_is_big != WIDE) rsrc_f = rsrc_a;
            }
        }
    };

    auto dma_piece = [&](int slot, int j, auto odd_tag) {
        constexpr int odd = decltype(odd_tag)::value;
        const lds_ptr sa = (lds_ptr)smem + slot * STAGE;
        const unsigned a_s = ga.a_soff() + odd * KS * 2, w_s = ga.w_soff() + odd * KS * 2;
        if constexpr (SQ) {
            if (j < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, sa + (j * 8 + wave) * 1024, 16, ga.a_off[j], a_s, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, sa + A_BYTES + ((j - 2) * 8 + wave) * 1024, 16, w_off[j - 2], w_s, 0, 0);
        } else if constexpr (!WIDE) {
            if (j < 2) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, sa + (j * 8 + wave) * 1024, 16, ga.a_off[j], a_s, 0, 0);
            } else if (j == 2) {
                const lds_ptr dst = flex_is_big ? sa + (16 + (wave & 3)) * 1024 : sa + A_BYTES + (wave & 3) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_f, dst, 16, flex_is_big ? ga.a_off[2] : w_off[0], flex_is_big ? a_s : w_s, 0, 0);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, sa + A_BYTES + small1 * 1024, 16, w_off[1], w_s, 0, 0);
            }
        } else {
            if (j < 2) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, sa + A_BYTES + (j * 8 + wave) * 1024, 16, w_off[j], w_s, 0, 0);
            } else if (j == 2) {
                const lds_ptr dst = flex_is_big ? sa + A_BYTES + (16 + (wave & 3)) * 1024 : sa + (wave & 3) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_f, dst, 16, flex_is_big ? w_off[2] : ga.a_off[0], flex_is_big ? w_s : a_s, 0, 0);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, sa + small1 * 1024, 16, ga.a_off[1], a_s, 0, 0);
            }
        }
    };
    auto issue_pair = [&](int slot_even, int slot_odd) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            dma_piece(slot_even, j, int_c<0>{});
            dma_piece(slot_odd, j, int_c<1>{});
        }
    };

    const int fr = lane & 15, fg = lane >> 4;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && p.splits == 1 && !(p.flags & MOCA_EP_LNFOLD)) bv = *reinterpret_cast<const f32x4*>(p.bias + (int64_t)wg * p.N + n0 + wave_n * WTN + nt * 16 + 4 * fg);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = bv;
    }

    const int swz = (fg ^ ((0x78 >> (2 * ((fr >> 2) & 3))) & 3)) << 4;
    const int a_off0 = (wave_m * WTM + fr) * RB + swz;
    const int b_off0 = A_BYTES + (wave_n * WTN + fr) * RB + swz;

    half8v af[2][MT], bf[2][NT];
    auto read_tile = [&](auto set_tag, int slot) {
        constexpr int S = decltype(set_tag)::value;
        const char* cur = smem + slot * STAGE;
#pragma unroll
        for (int r = 0; r < NT; ++r) bf[S][r] = *reinterpret_cast<const half8v*>(cur + b_off0 + r * 1024);
#pragma unroll
        for (int r = 0; r < MT; ++r) af[S][r] = *reinterpret_cast<const half8v*>(cur + a_off0 + r * 1024);
    };
    auto mfma_tile = [&](auto set_tag) {
        constexpr int S = decltype(set_tag)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[S][nt], af[S][mt], acc[mt][nt], 0, 0, 0);   // D^T: lane = row m
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: pairs (0,1) and (2,3) in flight, pair (0,1) landed everywhere ----
    LnFoldRaw lraw = {float2{0.f, 0.f}, float2{0.f, 0.f}, 0.f, 0.f};
    if (p.flags & MOCA_EP_LNFOLD) lraw = lnfold_issue<TM, BN>(p, grow(tid), n0, tid);
    ga.seek(kt_begin);
    sync_src();
    issue_pair(0, 1);
    ga.advance();
    sync_src();
    issue_pair(2, 3);
    LnFoldRegs lf = {0.f, 0.f, 0.f, 0.f};
    if (p.flags & MOCA_EP_LNFOLD) lf = lnfold_finish<TM, BN>(p, lraw, grow(tid), tid);
    MOCA_STAMP(1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
    __builtin_amdgcn_s_barrier();
    MOCA_STAMP(2);
    if (late) __builtin_amdgcn_s_barrier();            // from here on waves 4..7 run one barrier behind waves 0..3

    int s0 = 0;                                          // ring slot of tile i
    for (int i = 0; i < nk; i += 2) {
        const int s1 = s0 + 1 == NS ? 0 : s0 + 1;        // tile i+1
        const int sp = s0 == 0 ? NS - 1 : s0 - 1;        // tile i-1 (consumed) -> tile i+4
#if defined(W80S_VARIANT) && W80S_VARIANT == 0    // (A/B build: both fragment sets read in LOADe, MFMAe pure; 1-3 % slower)
        // ---- LOADe ----
        read_tile(int_c<0>{}, s0);
        read_tile(int_c<1>{}, s1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- MFMAe ----
        mfma_tile(int_c<0>{});
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
#else
#if defined(W80S_GN_DIAG)
        // DIAGNOSTIC build only (results wrong by construction; `make gndiag`, profiles/r04_ab_gn_in_consumer.txt): what a GroupNorm
        // apply + SiLU fused into this consumer would cost as a pass over the LANDED A k-tiles -- each wave rewrites its eighth of the
        // A rows of tiles i and i+1 (affine + SiLU per element, as gn_apply does) before anyone reads fragments, one more barrier.
        if constexpr (AMODE != MOCA_A_LINEAR && !SQ && !TQ) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                char* base = smem + (t == 0 ? s0 : s1) * STAGE;
                for (int c = tid; c < A_BYTES / 16; c += 512) {
                    half8v v = *reinterpret_cast<half8v*>(base + c * 16);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (half_t)moca_silu((float)v[j] * 1.0009765625f + 0.0009765625f);
                    *reinterpret_cast<half8v*>(base + c * 16) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
#endif
        // ---- LOADe: tile i only ----
#ifdef MOCA_STAMPS
#ifndef SEG_WAVE
#define SEG_WAVE 0
#endif
#ifndef SEG_ITER
#define SEG_ITER 2
#endif
        const bool seg = i == SEG_ITER;
        if (seg) MOCA_STAMP_W(8, SEG_WAVE);
#endif
        read_tile(int_c<0>{}, s0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#ifdef MOCA_STAMPS
        if (seg) MOCA_STAMP_W(9, SEG_WAVE);
#endif
        __builtin_amdgcn_s_barrier();
#ifdef MOCA_STAMPS
        if (seg) MOCA_STAMP_W(10, SEG_WAVE);
#endif
        // ---- MFMAe with the reads of tile i+1 in the gaps ----
        {
            const char* nx = smem + s1 * STAGE;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < MT * NT; ++j) {
                const int mt = j / NT, nt = j % NT;
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[0][nt], af[0][mt], acc[mt][nt], 0, 0, 0);
                if (j % 2 == 0 && j / 2 < MT + NT) {
                    const int r = j / 2;
                    __builtin_amdgcn_sched_barrier(0);
                    if (r < NT) bf[1][r] = *reinterpret_cast<const half8v*>(nx + b_off0 + r * 1024);
                    else af[1][r - NT] = *reinterpret_cast<const half8v*>(nx + a_off0 + (r - NT) * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#ifdef MOCA_STAMPS
        if (seg) MOCA_STAMP_W(11, SEG_WAVE);
#endif
        __builtin_amdgcn_s_barrier();
#ifdef MOCA_STAMPS
        if (seg) MOCA_STAMP_W(12, SEG_WAVE);
#endif
#endif
        // ---- LOADo ----
        ga.advance();
        sync_src();
        issue_pair(sp, s0);
#ifdef MOCA_STAMPS
        if (seg) MOCA_STAMP_W(13, SEG_WAVE);
#endif
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
        __builtin_amdgcn_sched_barrier(0);
#ifdef MOCA_STAMPS
        if (seg) MOCA_STAMP_W(14, SEG_WAVE);
#endif
        __builtin_amdgcn_s_barrier();
#ifdef MOCA_STAMPS
        if (seg) MOCA_STAMP_W(15, SEG_WAVE);
#endif
        // ---- MFMAo ----
        mfma_tile(int_c<1>{});
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        s0 = s1 + 1 == NS ? 0 : s1 + 1;
    }
    if (!late) __builtin_amdgcn_s_barrier();           // the halves meet again
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // every DMA (incl. the repeats) is done: the ring is free for the epilogue
    MOCA_STAMP(3);

    // ---- epilogue (as w80): lane owns 4 consecutive columns n = wave_n*80 + nt*16 + 4*fg + r of row m = wave_m*80 + mt*16 + fr ----
    if (p.splits > 1) {
        float* ws = p.splitk_ws + (int64_t)split * p.M * p.N;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = m0 + wave_m * WTM + mt * 16 + fr;
            if (row < p.M) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = n0 + wave_n * WTN + nt * 16 + 4 * fg;
                    *reinterpret_cast<f32x4*>(ws + (int64_t)row * p.N + col) = acc[mt][nt];
                }
            }
        }
        return;
    }
    const bool fold = (p.flags & MOCA_EP_LNFOLD) != 0;
    if constexpr (SQ) {
        if (p.flags & MOCA_EP_GEGLU) {                   // per 64-column group: value tiles +0, +1 and their gate tiles +2, +3 (bias is in the accumulators)
            constexpr int gpitch = (BN / 2) * 2 + 16;
            float* rst = reinterpret_cast<float*>(smem + TM * gpitch);
            if (fold) lnfold_publish<TM, BN>(lf, rst, tid);
#pragma unroll
            for (int grp = 0; grp < 2; ++grp)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int lcol = wave_n * WTN + grp * 64 + nt * 16 + 4 * fg;
                    f32x4 wv = {0.f, 0.f, 0.f, 0.f}, wg = wv, bv = wv, bg = wv;
                    if (fold) {
                        wv = *reinterpret_cast<const f32x4*>(rst + 2 * TM + lcol); wg = *reinterpret_cast<const f32x4*>(rst + 2 * TM + lcol + 32);
                        bv = *reinterpret_cast<const f32x4*>(rst + 2 * TM + BN + lcol); bg = *reinterpret_cast<const f32x4*>(rst + 2 * TM + BN + lcol + 32);
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int row = wave_m * WTM + mt * 16 + fr;
                        f32x4 va = acc[mt][4 * grp + nt], gt = acc[mt][4 * grp + nt + 2];
                        if (fold) {
                            const float2 st = *reinterpret_cast<const float2*>(rst + 2 * row);
                            va = lnfold_apply(va, st, wv, bv); gt = lnfold_apply(gt, st, wg, bg);
                        }
                        const f32x2 lo = moca_geglu2(f32x2{va[0], va[1]}, f32x2{gt[0], gt[1]});
                        const f32x2 hi = moca_geglu2(f32x2{va[2], va[3]}, f32x2{gt[2], gt[3]});
                        half4v h;
                        h[0] = (half_t)lo[0]; h[1] = (half_t)lo[1]; h[2] = (half_t)hi[0]; h[3] = (half_t)hi[1];
                        *reinterpret_cast<half4v*>(smem + row * gpitch + (wave_n * 64 + grp * 32 + nt * 16 + 4 * fg) * 2) = h;
                    }
                }
            __syncthreads();
            MOCA_STAMP(4);
            store_fp16_tile<512>(p, smem, gpitch, TM, BN / 2, m0, n0 / 2, tid);
            MOCA_STAMP(5);
            MOCA_STAMP_HW();
            return;
        }
    }
    constexpr int pitch = BN * 2 + 16;
    if (fold) {                                          // Linear(LayerNorm(x)) from x: row statistics -> LDS behind the staged tile
        float* rst = reinterpret_cast<float*>(smem + TM * pitch);
        lnfold_publish<TM, BN>(lf, rst, tid);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = wave_n * WTN + nt * 16 + 4 * fg;
            const f32x4 ws4 = *reinterpret_cast<const f32x4*>(rst + 2 * TM + col);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(rst + 2 * TM + BN + col);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int row = wave_m * WTM + mt * 16 + fr;
                const float2 st = *reinterpret_cast<const float2*>(rst + 2 * row);
                *reinterpret_cast<half4v*>(smem + row * pitch + col * 2) = __builtin_convertvector(lnfold_apply(acc[mt][nt], st, ws4, b4), half4v);
            }
        }
    } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = wave_n * WTN + nt * 16 + 4 * fg;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int row = wave_m * WTM + mt * 16 + fr;
                *reinterpret_cast<half4v*>(smem + row * pitch + col * 2) = __builtin_convertvector(acc[mt][nt], half4v);
            }
        }
    }
    __syncthreads();
    MOCA_STAMP(4);
    if constexpr (TQ) {
        // temporal attention of the 20 pixels of this tile for head tile_n: q | k | v = columns [0,64) | [64,128) | [128,192) of the
        // staged fp16 rows, 16 frames per pixel.  One wavefront per pixel (pixels w, w + 8, w + 16); the arithmetic is that of
        // temporal_attention_kernel (attention.hip): S^T = K.Q^T by v_mfma_f32_16x16x32_f16, softmax over the 16 keys in-lane
        // + two cross-lane steps, O^T = V^T.P^T by v_mfma_f32_16x16x16_f16.  Only O (64 of the 192 columns) goes to memory.
        const float sl2e = p.tattn_scale * 1.4426950408889634f;
        half_t* outp = reinterpret_cast<half_t*>(p.out);
        // Round 5 (profiles/r05_ab_tattn_epilogue.txt): this epilogue was 5.0 k of a tile's 30 k cycles -- a wave ran its 2-3 pixels one
        // after the other, each a chain of LDS reads -> MFMA -> cross-lane maximum -> exp -> cross-lane sum -> 64 two-byte LDS reads of V ->
        // MFMA.  Now (a) the wave's three pixel slots are computed side by side (uniform control flow: slot 2 of waves 4..7 repeats pixel
        // 19 and only skips its stores), so the chains overlap; (b) the cross-lane steps are v_permlane16/32_swap (VALU) instead of
        // ds_bpermute; (c) V^T fragments come from ONE ds_read_b64_tr_b16 per 16 columns instead of 16 two-byte reads.  Same values, same
        // order of additions.
        constexpr int NPX = 3;
        int pixs[NPX];
        half8v kf[NPX][2], qf[NPX][2];
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            pixs[i] = min(wave + 8 * i, 19);
            const char* base = smem + (pixs[i] * 16) * pitch;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                qf[i][ks] = *reinterpret_cast<const half8v*>(base + fr * pitch + (ks * 32 + fg * 8) * 2);
                kf[i][ks] = *reinterpret_cast<const half8v*>(base + fr * pitch + (64 + ks * 32 + fg * 8) * 2);
            }
        }
        half4v vfr[NPX][4];                                   // V^T fragments: lane (d = fr, keys 4 fg .. 4 fg + 3) of column block dt
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            const char* base = smem + (pixs[i] * 16) * pitch;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                vfr[i][dt] = tattn_tr_read(base + (4 * fg + (fr >> 2)) * pitch + (128 + dt * 16 + 4 * (fr & 3)) * 2);
        }
        f32x4 sc[NPX];
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            sc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            sc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[i][0], qf[i][0], sc[i], 0, 0, 0);
            sc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[i][1], qf[i][1], sc[i], 0, 0, 0);
        }
        half4v pf[NPX];
        float inv[NPX];
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            float mx = fmaxf(fmaxf(sc[i][0], sc[i][1]), fmaxf(sc[i][2], sc[i][3]));       // lane: S^T[key = 4 fg + r][query = fr]
            mx = xlane16_max(mx);
            mx = xlane32_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = exp2f((sc[i][r] - mx) * sl2e);
                sum += pv;
                pf[i][r] = (half_t)pv;
            }
            sum = xlane16_sum(sum);
            sum = xlane32_sum(sum);
            inv[i] = 1.0f / sum;
        }
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            half_t* ob = outp + (int64_t)grow(pixs[i] * 16 + fr) * p.ldo + tile_n * 64;
            const bool live = wave + 8 * i < 20;               // (wave-uniform)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 o4 = {0.f, 0.f, 0.f, 0.f};
                o4 = __builtin_amdgcn_mfma_f32_16x16x16f16(vfr[i][dt], pf[i], o4, 0, 0, 0);     // lane: O^T[d = 16 dt + 4 fg + r][query = fr]
                half4v h4;
#pragma unroll
                for (int r = 0; r < 4; ++r) h4[r] = (half_t)(o4[r] * inv[i]);
                if (live) *reinterpret_cast<half4v*>(ob + dt * 16 + 4 * fg) = h4;
            }
        }
    } else if constexpr (SQ) {
        store_fp16_tile<512>(p, smem, pitch, TM, BN, m0, n0, tid);
    } else if constexpr (WIDE) {
        if (p.flags & MOCA_EP_LN) store_fp16_tile_ln(p, smem, reinterpret_cast<float*>(smem + TM * pitch), pitch, m0, tid);
        else if (p.flags & (MOCA_EP_COLSUM | MOCA_EP_GSTAT)) store_fp16_tile_colsum<TM, BN>(p, smem, reinterpret_cast<float*>(smem + TM * pitch), pitch, m0, n0, tile_m, tid);
        else if (p.flags & MOCA_EP_ROWSUM) store_fp16_tile_rowsum<512, BN, 8>(p, smem, pitch, TM, m0, n0, tid);
        else store_fp16_tile<512>(p, smem, pitch, TM, BN, m0, n0, tid);
    } else {
        if (p.flags & (MOCA_EP_COLSUM | MOCA_EP_GSTAT)) store_fp16_tile_colsum<TM, BN>(p, smem, reinterpret_cast<float*>(smem + TM * pitch), pitch, m0, n0, tile_m, tid);
        else if (p.flags & MOCA_EP_ROWSUM) store_fp16_tile_rowsum<512, BN, 4>(p, smem, pitch, TM, m0, n0, tid);
        else store_fp16_tile<512>(p, smem, pitch, TM, BN, m0, n0, tid);
    }
    MOCA_STAMP(5);
    MOCA_STAMP_HW();
#endif
}

// =====================================================================================
// "sqp" kernel: the 256 x 256 staggered kernel (SHAPE 2 above: 8 waves as 4 x 2, wave tile 64 x 128, 5-slot ring of 32 KiB k-tiles, the
// two waves of a SIMD half an iteration apart) as a PERSISTENT kernel with a REGISTER epilogue, for the wide linears (GEGLU
// projections first).  Per 256 x 256 tile of the 320-channel GEGLU the round-4 kernel spent 31.5 k cycles: 5.0 k in the prologue, 14.0 k in
// the main loop, 9.5 k in the accumulator -> LDS staging pass with the erf-GELU, 2.5 k in the store loop
// (profiles/r05_g4_phase_stamps.txt).  Here
//   * one block per CU walks its tiles; the DMA stream is CONTINUOUS: the pair issued in an iteration's LOADo segment simply moves on to
//     the next tile's k-tiles when the current tile's are exhausted, so a tile's first two pairs are landing while the previous tile's
//     last iterations and epilogue run -- no prologue, no drain, no repeated pieces;
//   * W rows are fetched into the LDS tile in the permuted order of g4p_perm, so a lane's accumulators in two neighbouring MFMA tiles
//     are 8 consecutive output columns and the epilogue stores 16 bytes per lane straight from registers (16 rows x 64 B per
//     instruction): the ring is never used for staging, nothing waits for it to drain;
//   * tile statistics (LayerNorm fold: rstd, -mean rstd per row; wsum, bias per column) travel in four registers through the main
//     loop, as in the staggered kernel, and are published in the ONE ring slot that is free during an epilogue (the slot of the tile's
//     last k-tile: the stream refills it in the next tile's first LOADo, two barriers after the late half's epilogue has ended).
// =====================================================================================
// LDS accesses the compiler must not see: a read / write of ring memory that it cannot prove disjoint from an LDS-DMA in flight is
// given an `s_waitcnt vmcnt(0)` of its own (eight per epilogue here, each behind a global store).  Addresses are LDS byte offsets;
// lds_wait() = lgkmcnt(0) + the scheduling fence that keeps consumers below it (cdna_hip_programming.md rule 18).
__device__ __forceinline__ void lds_wr_f2(unsigned a, f32x2 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_wr_f1(unsigned a, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ f32x2 lds_rd_f2(unsigned a) { f32x2 v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a) : "memory"); return v; }
__device__ __forceinline__ f32x4 lds_rd_f4(unsigned a) { f32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory"); return v; }
__device__ __forceinline__ float lds_rd_f1(unsigned a) { float v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a) : "memory"); return v; }
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }

__device__ __forceinline__ int sqp_perm(int rho) {              // LDS row of the 256-row W tile -> packed W row of the tile (64-column groups as g4p_perm)
    return (rho & ~63) + (g4p_perm(rho & 127) & 63);
}

template <bool GEGLU>
__global__ __launch_bounds__(512, 2) void gemm_sqp_kernel(const moca_gemm_params p) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifndef SEG_WAVE
#define SEG_WAVE 0
#endif
    constexpr int MT = 4, NT = 8, KS = 32, RB = 64, WTM = 64, WTN = 128, TM = 256, BN = 256;
    constexpr int A_BYTES = TM * RB, STAGE = A_BYTES + BN * RB, NS = 5, PPW = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const bool late = wave >= 4;
    const int nper = p.reserved2_;                               // persistent blocks (multiple of 8); the rest of the grid prefetches
    if (prefetch_block(p, nper, 512)) return;

    // ---- this block's tiles: XCD x = b & 7 owns either a contiguous range of the tile_m-major raster or (xcd_n > 1) an xm x xn sub-grid,
    //      its J = nper / 8 blocks walk it with stride J ----
    const int tiles_n = p.N / BN, tiles_m = (p.M + TM - 1) / TM;
    const int xcd_n = p.reserved4_ >> 8;
    const int J = nper >> 3;
    int q_base, q_cnt, sub_n, tm_base, tn_base;
    {
        const int x = blockIdx.x & 7;
        if (xcd_n > 1) {
            const int xm = 8 / xcd_n, sm = tiles_m / xm;
            sub_n = tiles_n / xcd_n;
            const int xi = x / xcd_n, xj = x - xi * xcd_n;
            tm_base = xi * sm; tn_base = xj * sub_n;
            q_base = 0; q_cnt = sm * sub_n;
        } else {
            const int ntiles = tiles_m * tiles_n, q = ntiles >> 3, r = ntiles & 7;
            q_base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
            q_cnt = q + (x < r ? 1 : 0);
            sub_n = tiles_n; tm_base = 0; tn_base = 0;
        }
    }
    auto tile_of = [&](int q, int& tm, int& tn) {                // q-th tile of this XCD's set
        const int l = q_base + q;
        const int a = l / sub_n;
        tm = tm_base + a; tn = tn_base + (l - a * sub_n);
    };
    // walk of the XCD's set (p.reserved4_ bit 1): STRIDED -- block j takes tiles j, j + J, ...: at any time the XCD's J blocks hold J
    // consecutive tiles of the raster (3.2 rows of 10 column tiles), every A row tile is shared by ~10 blocks that all request it at the
    // same moment, i.e. all of them wait out the same HBM miss; or CONTIGUOUS -- block j takes tiles [j cnt / J, (j + 1) cnt / J): it walks
    // a row's column tiles one after the other, so its A row tile misses to HBM once and is then re-read from L2 / the Infinity Cache,
    // while the J blocks of the XCD, in step, share ONE W panel at a time.
    int q_cur, q_step, q_end;
    {
        const int j = blockIdx.x >> 3;
        if (p.reserved4_ & 2) {
            q_cur = (int)((int64_t)q_cnt * j / J);
            q_end = (int)((int64_t)q_cnt * (j + 1) / J);
            q_step = 1;
        } else {
            q_cur = j; q_end = q_cnt; q_step = J;
        }
    }
    if (q_cur >= q_end) return;                                  // (block-uniform)
    const int nk = p.K / 32;                                     // K % 64 == 0 (host-checked): an even number of k-tiles

    // ---- DMA stream: piece = 16 rows x 64 B; A pieces w and 8 + w, W pieces w and 8 + w of every k-tile (4 per wave) ----
    const int lrow = lane >> 2, pch = lane & 3;
    const int lch = pch ^ ((0x78 >> (2 * ((lrow >> 2) & 3))) & 3);
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, OOB_OFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, OOB_OFF, 0x00020000);
    unsigned a_off[2], w_off[2];
    int d_q = q_cur, d_k = 0;                                    // tile / even k-tile of the next pair the stream issues
    auto set_dma_tile = [&](int q) {                             // q >= q_end: no tile (every lane out of range: zero fill, no traffic)
        int tm, tn;
        tile_of(q, tm, tn);
        const bool any = q < q_end;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int row = tm * TM + (g * 8 + wave) * 16 + lrow;
            a_off[g] = (any && row < p.M) ? (unsigned)(((int64_t)row * p.lda + lch * 8) * 2) : OOB_OFF;
            w_off[g] = any ? (unsigned)(((int64_t)(tn * BN + sqp_perm((g * 8 + wave) * 16 + lrow)) * p.ldw + lch * 8) * 2) : OOB_OFF;
        }
    };
    auto issue_pair = [&](int slot_even, int slot_odd) {
        const unsigned soff = (unsigned)(d_k * KS * 2);
        const lds_ptr se = (lds_ptr)smem + slot_even * STAGE, so = (lds_ptr)smem + slot_odd * STAGE;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, se + (g * 8 + wave) * 1024, 16, a_off[g], soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, so + (g * 8 + wave) * 1024, 16, a_off[g], soff + KS * 2, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, se + A_BYTES + (g * 8 + wave) * 1024, 16, w_off[g], soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, so + A_BYTES + (g * 8 + wave) * 1024, 16, w_off[g], soff + KS * 2, 0, 0);
        }
        d_k += 2;
        if (d_k >= nk) { d_k = 0; d_q += q_step; set_dma_tile(d_q); }
    };

    const int fr = lane & 15, fg = lane >> 4;
    const int swz = (fg ^ ((0x78 >> (2 * ((fr >> 2) & 3))) & 3)) << 4;
    const int a_off0 = (wave_m * WTM + fr) * RB + swz;
    const int b_off0 = A_BYTES + (wave_n * WTN + fr) * RB + swz;
    f32x4 acc[MT][NT];
    half8v af[2][MT], bf[2][NT];
    auto read_tile = [&](auto set_tag, int slot) {
        constexpr int S = decltype(set_tag)::value;
        const char* cur = smem + slot * STAGE;
#pragma unroll
        for (int r = 0; r < NT; ++r) bf[S][r] = *reinterpret_cast<const half8v*>(cur + b_off0 + r * 1024);
#pragma unroll
        for (int r = 0; r < MT; ++r) af[S][r] = *reinterpret_cast<const half8v*>(cur + a_off0 + r * 1024);
    };

    // (asm volatile: recomputed where it is used -- as a loop invariant it is hoisted, kept across the main loop and spilled)
    auto tid_now = [&]() -> int {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return wave * 64 + l;
    };
    // ---- tile statistics: thread t < 256 (the early half) owns row t and LDS column t of the tile.  The values of tile q + 1 are
    //      fetched at the head of tile q's epilogue and finished behind its stores; they travel through the main loop as four registers
    //      (rstd, -mean rstd, wsum, bias).  The fetch is LDS-DMA (4 bytes per lane, into 22 KiB of the free slot behind the published
    //      statistics, one private area per wavefront), read back with inline-asm ds_reads behind a counted wait -- no VGPR is the
    //      destination of a load: a VGPR load the compiler knows about makes it guard every later write of those registers (the fragment
    //      reads of the main loop) with `s_waitcnt vmcnt(0)`, draining the DMA stream once per iteration; an inline-asm VGPR load is
    //      copied by the register allocator BEFORE the asm wait that retires it (`v_mov` of the load destinations in front of the
    //      `s_waitcnt` statement that names them "+v": seen in the ISA of the first form of this path -- stale statistics whenever a load
    //      is slower than the epilogue's arithmetic). ----
    const bool fold = (p.flags & MOCA_EP_LNFOLD) != 0;
    const float* const dummy = reinterpret_cast<const float*>(p.w);
    constexpr unsigned STAT_SCRATCH = 8192;                      // offset of the raw-value areas inside the free slot
    const int stat_floats = 2 + 2 * (fold ? p.lnf_nparts : 0);   // per lane: wsum, bias, (sum, sum of squares) per row partial
    auto stat_dma = [&](int q, unsigned slot_off) {              // waves 0..3 only
        int tm, tn;
        tile_of(q < q_end ? q : 0, tm, tn);
        const int t = tid_now();
        const int n = tn * BN + sqp_perm(t);
        const int m = min(tm * TM + t, p.M - 1);
        const lds_ptr area = (lds_ptr)smem + slot_off + STAT_SCRATCH + wave * (22 * 256);
        __builtin_amdgcn_global_load_lds((glb_ptr)(fold ? p.lnf_wsum + n : dummy), area, 4, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_ptr)(p.bias ? p.bias + n : dummy), area + 256, 4, 0, 0);
        if (fold) {
            for (int i = 0; i < p.lnf_nparts; ++i) {             // (block-uniform; <= 10)
                const float* src = p.lnf_part + ((int64_t)i * p.M + m) * 2;
                __builtin_amdgcn_global_load_lds((glb_ptr)src, area + (2 + 2 * i) * 256, 4, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_ptr)(src + 1), area + (3 + 2 * i) * 256, 4, 0, 0);
            }
        }
    };
    // (`after` = vector-memory instructions this wave has issued behind the DMAs, at least: the wait leaves that many outstanding)
    auto stat_read = [&](unsigned slot_off, auto after_tag) -> LnFoldRegs {
        constexpr int after = decltype(after_tag)::value;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(after) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        int lane_;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_));
        const unsigned area = (unsigned)(size_t)((lds_ptr)smem + slot_off + STAT_SCRATCH + wave * (22 * 256)) + 4 * lane_;
        const float ws = lds_rd_f1(area), b = lds_rd_f1(area + 256);
        float s = 0.f, qq = 0.f;
        LnFoldRegs f = {1.f, 0.f, 0.f, 0.f};
        if (fold) {
            for (int i = 0; i < p.lnf_nparts; ++i) {
                const float x = lds_rd_f1(area + (2 + 2 * i) * 256), y = lds_rd_f1(area + (3 + 2 * i) * 256);
                lds_wait();
                s += x; qq += y;
            }
        }
        lds_wait();
        f.ws = fold ? ws : 0.f;
        f.b = p.bias ? b : 0.f;
        if (fold) {
            const float inv_k = 1.0f / (float)p.K;
            const float mean = s * inv_k;
            const float var = fmaxf(qq * inv_k - mean * mean, 0.f);
            f.rs = rsqrtf(var + p.ln_eps);
            f.rb = -mean * f.rs;
        }
        return f;
    };
    (void)stat_floats;

    // ---- prologue (once per block): two pairs in flight, the first landed everywhere ----
    LnFoldRegs lf = {1.f, 0.f, 0.f, 0.f};
    if (!late) {                                                 // (slot 4 is not part of the prologue's four k-tiles)
        stat_dma(q_cur, (NS - 1) * STAGE);
        lf = stat_read((NS - 1) * STAGE, int_c<0>{});
    }
    __builtin_amdgcn_s_barrier();                                // (every read of slot 4 done before the stream reaches it)
    set_dma_tile(q_cur);
    issue_pair(0, 1);
    issue_pair(2, 3);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
    __builtin_amdgcn_s_barrier();
    if (late) __builtin_amdgcn_s_barrier();                      // from here on waves 4..7 run one barrier behind waves 0..3

    const bool nt_out = out_streams(p);
    const half_t* __restrict__ resid = reinterpret_cast<const half_t*>(p.residual);
    const half_t* __restrict__ rowadd = reinterpret_cast<const half_t*>(p.rowadd);
    int s0 = 0;                                                  // ring slot of the running stream's current k-tile
#ifdef MOCA_STAMPS
    int tile_no = 0;
#endif
    while (true) {
        int tm, tn;
        tile_of(q_cur, tm, tn);
        const int m0 = tm * TM, n0 = tn * BN;
#ifdef MOCA_STAMPS
        const bool stamp_it = tile_no == 3;          // (phases: [0, 0, main loop, publish + barrier, epilogue arithmetic + stores] + the statistics wait)
        if (stamp_it) { MOCA_STAMP(0); }
#endif
        // (plain zeros, default unrolling: the compiler peels the first iteration -- literal C operand -- and reconciles the peeled copy's
        //  result registers with ~100 v_mov per steady-state iteration; opaque zeros and / or `unroll(disable)` remove the moves and
        //  measure 1-8 % SLOWER on every GEGLU shape, same box, profiles/r05_ab_sqp_loop_shape.txt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < nk; i += 2) {
            const int s1 = s0 + 1 == NS ? 0 : s0 + 1;
            const int sp = s0 == 0 ? NS - 1 : s0 - 1;
            // ---- LOADe: k-tile i ----
#ifdef MOCA_STAMPS
            const bool seg = stamp_it && i == SEG_ITER;
            if (seg) MOCA_STAMP_W(8, SEG_WAVE);
#endif
            read_tile(int_c<0>{}, s0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#ifdef MOCA_STAMPS
            if (seg) MOCA_STAMP_W(9, SEG_WAVE);
#endif
            __builtin_amdgcn_s_barrier();
#ifdef MOCA_STAMPS
            if (seg) MOCA_STAMP_W(10, SEG_WAVE);
#endif
            // ---- MFMAe with the reads of k-tile i + 1 in the gaps ----
            {
                const char* nx = smem + s1 * STAGE;
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int j = 0; j < MT * NT; ++j) {
                    const int mt = j / NT, nt = j % NT;
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[0][nt], af[0][mt], acc[mt][nt], 0, 0, 0);
                    if (j % 2 == 0 && j / 2 < MT + NT) {
                        const int r = j / 2;
                        __builtin_amdgcn_sched_barrier(0);
                        if (r < NT) bf[1][r] = *reinterpret_cast<const half8v*>(nx + b_off0 + r * 1024);
                        else af[1][r - NT] = *reinterpret_cast<const half8v*>(nx + a_off0 + (r - NT) * 1024);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_s_setprio(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#ifdef MOCA_STAMPS
            if (seg) MOCA_STAMP_W(11, SEG_WAVE);
#endif
            __builtin_amdgcn_s_barrier();
#ifdef MOCA_STAMPS
            if (seg) MOCA_STAMP_W(12, SEG_WAVE);
#endif
            // ---- LOADo: the stream's next pair into the slots of k-tiles i - 1 and i; the pair before it has landed ----
            issue_pair(sp, s0);
#ifdef MOCA_STAMPS
            if (seg) MOCA_STAMP_W(13, SEG_WAVE);
#endif
            // (behind an epilogue this also waits for the epilogue's stores, which are older than the pair just issued: leaving them
            //  outstanding -- vmcnt(8 + stores) in a tile's first LOADo -- measured no different, profiles/r05_ab_sqp_store_wait.txt)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
            __builtin_amdgcn_sched_barrier(0);
#ifdef MOCA_STAMPS
            if (seg) MOCA_STAMP_W(14, SEG_WAVE);
#endif
            __builtin_amdgcn_s_barrier();
#ifdef MOCA_STAMPS
            if (seg) MOCA_STAMP_W(15, SEG_WAVE);
#endif
            // ---- MFMAo ----
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[1][nt], af[1][mt], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
#ifdef MOCA_STAMPS
            if (seg) MOCA_STAMP_W(1, SEG_WAVE);
#endif
            __builtin_amdgcn_s_barrier();
#ifdef MOCA_STAMPS
            if (seg) MOCA_STAMP_W(2, SEG_WAVE);
#endif
            s0 = s1 + 1 == NS ? 0 : s1 + 1;
        }
#ifdef MOCA_STAMPS
        if (stamp_it) MOCA_STAMP(3);
#endif
        // ---- epilogue: statistics -> the free slot (that of the tile's last k-tile), then registers -> memory.  The halves MEET first
        //      (waves 0..3 wait for waves 4..7's last MFMAo) and part again behind it: one barrier apart, with no barrier inside, the
        //      two epilogues would run one after the other -- each half's 6 k cycles of arithmetic inside the other's barrier wait ----
        if (!late) __builtin_amdgcn_s_barrier();
        const unsigned lst = (unsigned)(size_t)((lds_ptr)smem + (s0 == 0 ? NS - 1 : s0 - 1) * STAGE);   // [TM] float2, [BN] wsum, [BN] bias
        const unsigned lws = lst + 8 * TM, lbi = lws + 4 * BN;
        if (!late) {
            const int t = tid_now();                              // (recomputed: a thread id kept across the main loop is the one value it spills)
            lds_wr_f2(lst + 8 * t, f32x2{lf.rs, lf.rb});
            lds_wr_f1(lws + 4 * t, lf.ws);
            lds_wr_f1(lbi + 4 * t, lf.b);
        }
        lds_wait();
        __builtin_amdgcn_s_barrier();
#ifdef MOCA_STAMPS
        if (stamp_it) MOCA_STAMP(4);
#endif
        f32x2 st[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) st[mt] = lds_rd_f2(lst + 8 * (wave_m * WTM + mt * 16 + fr));
        const int q_next = q_cur + q_step;
        const unsigned stat_slot = (unsigned)((s0 == 0 ? NS - 1 : s0 - 1) * STAGE);
        if (!late) stat_dma(q_next, stat_slot);                   // (in flight under the epilogue's arithmetic; read back behind its stores)
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
            f32x4 cw[4], cb[4];                                   // (per 64-column group: 32 registers instead of 64)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                cw[t] = lds_rd_f4(lws + 4 * (wave_n * WTN + (grp * 4 + t) * 16 + 4 * fg));
                cb[t] = lds_rd_f4(lbi + 4 * (wave_n * WTN + (grp * 4 + t) * 16 + 4 * fg));
            }
            lds_wait();
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = m0 + wave_m * WTM + mt * 16 + fr;
                f32x4 v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = st[mt][0] * acc[mt][grp * 4 + t] + (st[mt][1] * cw[t] + cb[t]);
                if constexpr (GEGLU) {
                    half8v h;
#pragma unroll
                    for (int nv = 0; nv < 2; ++nv) {
                        const f32x4 r4 = moca_geglu4(v[nv], v[nv + 2]);        // (4-wide: 7 instead of 133 s_nop per epilogue, -1 % at K = 320)
                        h[4 * nv + 0] = (half_t)r4[0]; h[4 * nv + 1] = (half_t)r4[1]; h[4 * nv + 2] = (half_t)r4[2]; h[4 * nv + 3] = (half_t)r4[3];
                    }
                    if (m < p.M) st_out8(reinterpret_cast<half_t*>(p.out) + (int64_t)m * p.ldo + (n0 >> 1) + wave_n * 64 + grp * 32 + 8 * fg, h, nt_out);
                } else {
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int col = n0 + wave_n * WTN + grp * 64 + hh * 32 + 8 * fg;
                        float o[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { o[r] = v[2 * hh][r]; o[4 + r] = v[2 * hh + 1][r]; }
                        if (m < p.M) {
                            if (rowadd) {
                                const half8v e = *reinterpret_cast<const half8v*>(rowadd + (int64_t)(m / p.rowadd_div) * p.ld_rowadd + col);
#pragma unroll
                                for (int r = 0; r < 8; ++r) o[r] = (float)(half_t)o[r] + (float)e[r];
                            }
                            if (resid) {
                                const half8v e = *reinterpret_cast<const half8v*>(resid + (int64_t)m * p.ldr + col);
#pragma unroll
                                for (int r = 0; r < 8; ++r) o[r] = (rowadd ? o[r] : (float)(half_t)o[r]) + (float)e[r];
                            }
                            half8v h;
#pragma unroll
                            for (int r = 0; r < 8; ++r) h[r] = (half_t)o[r];
                            st_out8(reinterpret_cast<half_t*>(p.out) + (int64_t)m * p.ldo + col, h, nt_out);
                        }
                    }
                }
            }
        }
        // the next tile's statistics: complete once at most the stores issued behind their loads are outstanding (GEGLU: 8 per lane, plain
        // 16; a tile with rows past M may have skipped stores: full drain there)
#ifdef MOCA_STAMPS
        if (stamp_it) { MOCA_STAMP(5); MOCA_STAMP_HW(); }
        ++tile_no;
#endif
        if (!late) {
            if (m0 + TM > p.M) lf = stat_read(stat_slot, int_c<0>{});
            else lf = stat_read(stat_slot, int_c<(GEGLU ? 8 : 16)>{});
        }
        if (q_next >= q_end) break;
        q_cur = q_next;
        if (late) __builtin_amdgcn_s_barrier();                  // waves 4..7 fall one barrier behind again
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (the stream's last pairs were zero fill; the halves are together here)
#endif
}

// XCD partition (xm x xn = 8, xn returned; 1 = the 1-D partition) of a tiles_m x tiles_n grid of TM x BN tiles of a LINEAR
// launch: estimated fabric bytes = W part + A part.  W: an XCD whose W sub-range ((tiles_n / xn) BN x K) fits its L2 (<= 3 MB)
// fetches it once -> xm |W| in total; one that does not streams it again for every M tile it owns -> tiles_m |W| whatever
// the partition.  A is consumed row tile by row tile -> xn |A|.  Only partitions that divide both tile counts; a 2-D partition is
// taken when it saves >= 20 %.
static int choose_xcd_n(const moca_gemm_params& p, int tiles_m, int tiles_n, int TM, int BN) {
    if (p.splits != 1 || p.a_mode != MOCA_A_LINEAR) return 1;
    const double Wtot = (double)p.N * p.K * 2, Atot = (double)p.M * p.K * 2;
    auto cost = [&](int xn) {
        const int xm = 8 / xn;
        const double wsub = (double)(tiles_n / xn) * BN * p.K * 2;
        return (wsub <= 3.6e6 ? xm * Wtot : tiles_m * Wtot) + xn * Atot;      // (3.3 MB of W + the A tiles in flight still mostly hit: -3..6 % on the 1280-channel GEGLU)
    };
    double best = cost(1);
    int best_xn = 1;
    for (int xn = 2; xn <= 8; xn *= 2) {
        if (tiles_m % (8 / xn) || tiles_n % xn) continue;
        const double c = cost(xn);
        if (c < 0.8 * best) { best = c; best_xn = xn; }
    }
    (void)TM;
    return best_xn;
}

template <int AMODE, int SHAPE>
int launch_gemm_w80s(const moca_gemm_params& p, hipStream_t st) {
    constexpr int TM = SHAPE == 2 ? 256 : (SHAPE == 1 ? 160 : 320), BN = SHAPE == 2 ? 256 : (SHAPE == 1 ? 320 : (SHAPE == 3 ? 192 : 160));
    const int tiles_m = (p.M + TM - 1) / TM, tiles_n = p.N / BN;
    const int nblk = tiles_m * tiles_n * p.splits;
    moca_gemm_params pl = p;
    pl.reserved4_ = (pl.reserved4_ & 0xff) | ((SHAPE == 3 ? 1 : choose_xcd_n(p, tiles_m, tiles_n, TM, BN)) << 8);
    constexpr int lds = 5 * (TM + BN) * 64;              // 150 KiB (160 KiB for 256 x 256); the fp16 epilogue tile fits inside the ring
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w80s_kernel<AMODE, SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return MOCA_E_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_w80s_kernel<AMODE, SHAPE>), dim3(nblk + prefetch_blocks(pl)), dim3(512), lds, st, pl);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

// bytes spanned by the A operand / the W operand: the buffer-addressed kernels need every in-range offset below 2^31
static inline int64_t a_span_bytes(const moca_gemm_params& p) {
    if (p.a_mode == MOCA_A_LINEAR) return ((int64_t)p.M * p.lda + 64) * 2;
    if (p.a_mode == MOCA_A_CONV3X3) return ((int64_t)(p.M / (p.outH * p.outW)) * p.inH * p.inW * p.C + 64) * 2;
    return ((int64_t)p.M * p.C + 64) * 2;
}
static inline bool buffer_addressable(const moca_gemm_params& p) {
    if (p.a2 && ((int64_t)p.M * p.lda2 + 64) * 2 >= (1ll << 31)) return false;
    return a_span_bytes(p) < (1ll << 31) && (int64_t)p.N * p.ldw * 2 < (1ll << 31);
}

static inline int sq256_mode() { return moca_tuning_get(MOCA_TUNE_GEMM_SQ256); }
static inline bool fast_gather(const moca_gemm_params& p);
// g4 (4 waves, two blocks per CU): the GEGLU projections with K <= 640 -- but not the very tall ones (M >= 2^17: B = 16 forwards), where
// the 256 x 256 staggered kernel is 4-5 % ahead (tools/bench_gemm.py geglu, BG_B=16 BG_TUNE=2:2).  MOCA_TUNE_GEMM_G4 = 0 / 2: never / always.
static inline bool wants_g4(const moca_gemm_params& p) {
    const int g4_mode = moca_tuning_get(MOCA_TUNE_GEMM_G4);
    return g4_mode == 2 || (g4_mode == 1 && (p.flags & MOCA_EP_GEGLU) && p.K <= 640 && p.M < (1 << 17));
}
static inline bool buffer_addressable(const moca_gemm_params& p);
static inline bool g4p_ok(const moca_gemm_params& p);
static inline bool sqp_ok(const moca_gemm_params& p);
// the persistent 256 x 256 kernel (MOCA_TUNE_GEMM_SQP = 0: never, 1: the GEGLU projections, 2: every linear it can run -- tests, A/B)
static inline bool takes_sqp(const moca_gemm_params& p) {
    const int mode = moca_tuning_get(MOCA_TUNE_GEMM_SQP);
    if (!mode || !sqp_ok(p)) return false;
    return mode == 2 || (p.flags & MOCA_EP_GEGLU);
}
// the persistent two-blocks-per-CU kernel (MOCA_TUNE_GEMM_G4P = 0: never, 1: the GEGLU projections, 2: every linear it can run -- tests, A/B)
static inline bool takes_g4p(const moca_gemm_params& p) {
    const int mode = moca_tuning_get(MOCA_TUNE_GEMM_G4P);
    if (!mode || !g4p_ok(p)) return false;
    return mode == 2 || (p.flags & MOCA_EP_GEGLU);
}
// the staggered kernel on 256 x 256 tiles for the wide projections (MOCA_TUNE_GEMM_SQ256 = 0: never, 1: not where g4 is preferred,
// 2: every wide linear -- A/B runs); asked after takes_w80()
static inline bool takes_sq256(const moca_gemm_params& p, bool use_g4) {
    const int mode = sq256_mode();
    return mode && p.a_mode == MOCA_A_LINEAR && p.N % 256 == 0 && p.N >= 2560 && p.M >= 512 && fast_gather(p) && buffer_addressable(p) &&
           !(p.flags & (MOCA_EP_OUT_F32 | MOCA_FORCE_SMALL_TILE)) && ((p.M + 255) / 256) * (p.N / 256) * p.splits >= 200 &&
           (mode == 2 || !use_g4);
}
static inline bool fast_gather(const moca_gemm_params& p) {
    return (p.a_mode == MOCA_A_LINEAR) ? (p.K % BK == 0 && p.K <= 8192) : (p.C % BK == 0 && p.C <= 8192);
}
// the weight-stationary streaming kernel of the 320 -> 320 linears (gemm_ws.hip; MOCA_TUNE_GEMM_WS = 0: never).  Asked FIRST by every
// predicate below: a call it takes runs on none of the tiled kernels.  (The queries -- moca_gemm_colsum_rows / _rowsum_cols / _lnfold_ok /
// _ln_ok -- add the flag they ask about before they come here: the kernel has no column sums, no LayerNorm fold, no LayerNorm store.)
static inline bool takes_ws(const moca_gemm_params& p) {
    const int mode = moca_tuning_get(MOCA_TUNE_GEMM_WS);
    if (!mode || !moca_gemm_ws_ok(p)) return false;
    // 1: where it was measured ahead of the tiled kernel (profiles/r06_ab_gemm_ws.txt): with a residual at every size, without one from
    // M = 2^17 up (B = 16 forwards; at M = 81920 a block's 10 strips do not amortise its start-up: W fetch + first strip).  2: wherever it applies
    return mode == 2 || p.residual || p.M >= (1 << 17);
}
// does this (validated, split-normalised) call run on the 320 x 160 kernels / on their staggered buffer-addressed form?
static inline bool takes_w80(const moca_gemm_params& p) {
    if (takes_ws(p)) return false;
    const int w80_mode = moca_tuning_get(MOCA_TUNE_GEMM_W80);
    const int tiles320 = ((p.M + 319) / 320) * (p.N / 160);
    return w80_mode && p.N % 160 == 0 && !(p.flags & (MOCA_EP_GEGLU | MOCA_EP_OUT_F32 | MOCA_FORCE_SMALL_TILE)) && p.M > 160 &&
           (tiles320 * p.splits >= 200 || w80_mode == 2) && fast_gather(p) && buffer_addressable(p);
}
static inline bool takes_w80s(const moca_gemm_params& p) {
    return takes_w80(p);
}
// which form of the staggered kernel a call that takes_w80s() runs on.  The 160 x 320 tiling is required by the LayerNorm store
// loop and taken by every N % 320 == 0 contraction: A is fetched once per 320 columns instead of once per 160 (linears: 37 vs 38 us
// at M = 81920, N = K = 320; 99 vs 103 at K = 1280; 182 vs 193 / 366 vs 384 at M = 327680; N = 640: 22.7 vs 25.1).
// Convs / temporal convs gain nothing from it (+-1 %, A/B on one device) and stay on the 320 x 160 form.
// MOCA_TUNE_GEMM_WIDE = 0: only with MOCA_EP_LN; 1 (default): linears; 2: convs / temporal convs too (tests)
static inline bool w80s_wide(const moca_gemm_params& p) {
    if (p.flags & MOCA_EP_LN) return true;
    const int mode = moca_tuning_get(MOCA_TUNE_GEMM_WIDE);
    if (mode == 0 || p.N % 320) return false;
    return p.a_mode == MOCA_A_LINEAR || mode == 2;
}
static inline bool takes_w80t_ln(const moca_gemm_params& pp) {     // the 160 x 320 tiling with the LayerNorm store loop
    moca_gemm_params p = pp;
    p.flags |= MOCA_EP_LN;
    return p.a_mode == MOCA_A_LINEAR && p.N == 320 && p.splits == 1 && takes_w80s(p) && !(p.flags & (MOCA_EP_COLSUM | MOCA_EP_GSTAT));
}

int launch_gemm_w80_mode(const moca_gemm_params& p, hipStream_t st) {
    const bool wide = w80s_wide(p);
    if (p.a2) return wide ? launch_gemm_w80s<MOCA_A_LINEAR2, 1>(p, st) : launch_gemm_w80s<MOCA_A_LINEAR2, 0>(p, st);
    if (p.a_mode == MOCA_A_LINEAR) return wide ? launch_gemm_w80s<MOCA_A_LINEAR, 1>(p, st) : launch_gemm_w80s<MOCA_A_LINEAR, 0>(p, st);
    if (p.a_mode == MOCA_A_CONV3X3) return wide ? launch_gemm_w80s<MOCA_A_CONV3X3, 1>(p, st) : launch_gemm_w80s<MOCA_A_CONV3X3, 0>(p, st);
    return wide ? launch_gemm_w80s<MOCA_A_TCONV3, 1>(p, st) : launch_gemm_w80s<MOCA_A_TCONV3, 0>(p, st);
}

template <int AMODE, bool FAST>
int launch_gemm_g4(const moca_gemm_params& p, hipStream_t st) {
    const int tiles_m = (p.M + 255) / 256, tiles_n = p.N / 128;
    const int nblk = tiles_m * tiles_n * p.splits;
    constexpr int lds = 3 * (256 + 128) * 64;       // 72 KiB ring; the fp16 epilogue tile (256 x 272 B) fits inside it
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_g4_kernel<AMODE, FAST>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return MOCA_E_LAUNCH;
        attr_set = true;
    }
    moca_gemm_params pl = p;
    pl.reserved4_ = (pl.reserved4_ & 0xff) | (choose_xcd_n(p, tiles_m, tiles_n, 256, 128) << 8);
    hipLaunchKernelGGL((gemm_g4_kernel<AMODE, FAST>), dim3(nblk + 2 * prefetch_blocks(pl)), dim3(256), lds, st, pl);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

// persistent two-blocks-per-CU kernel: which calls it can run (linears whose 256 x 128 tiles fill two blocks on every CU; bias,
// LayerNorm fold, GEGLU or row add / residual -- none of the statistics epilogues)
constexpr int G4P_BLOCKS = 512;
static inline bool g4p_ok(const moca_gemm_params& p) {
    if (p.a_mode != MOCA_A_LINEAR || p.splits != 1 || p.N % 128 || p.K % 64 || !buffer_addressable(p)) return false;
    if (p.flags & ~(MOCA_EP_GEGLU | MOCA_EP_LNFOLD)) return false;
    if ((p.flags & MOCA_EP_GEGLU) && (p.residual || p.rowadd)) return false;
    return ((p.M + 255) / 256) * (p.N / 128) >= G4P_BLOCKS;
}
template <bool GEGLU>
int launch_gemm_g4p(const moca_gemm_params& p, hipStream_t st) {
    constexpr int lds = 3 * (256 + 128) * 64 + (2 * 256 + 2 * 128) * 4;      // 72 KiB ring + 3 KiB of tile statistics
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_g4p_kernel<GEGLU>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return MOCA_E_LAUNCH;
        attr_set = true;
    }
    moca_gemm_params pl = p;
    pl.reserved2_ = G4P_BLOCKS;
    if (moca_tuning_get(MOCA_TUNE_GEMM_MF32)) {
        static bool attr_set_q = false;
        if (!attr_set_q) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_g4q_kernel<GEGLU>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
                return MOCA_E_LAUNCH;
            attr_set_q = true;
        }
        hipLaunchKernelGGL((gemm_g4q_kernel<GEGLU>), dim3(G4P_BLOCKS + 2 * prefetch_blocks(pl)), dim3(256), lds, st, pl);
    } else {
        hipLaunchKernelGGL((gemm_g4p_kernel<GEGLU>), dim3(G4P_BLOCKS + 2 * prefetch_blocks(pl)), dim3(256), lds, st, pl);
    }
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

// persistent 256 x 256 staggered kernel: the linears whose 256 x 256 tiles give every CU at least one (bias, LayerNorm fold, GEGLU or
// row add / residual -- none of the statistics epilogues)
constexpr int SQP_BLOCKS = 256;
static inline bool sqp_ok(const moca_gemm_params& p) {
    if (p.a_mode != MOCA_A_LINEAR || p.splits != 1 || p.N % 256 || p.K % 64 || !buffer_addressable(p)) return false;
    if (p.flags & ~(MOCA_EP_GEGLU | MOCA_EP_LNFOLD)) return false;
    if ((p.flags & MOCA_EP_GEGLU) && (p.residual || p.rowadd)) return false;
    return ((p.M + 255) / 256) * (p.N / 256) >= SQP_BLOCKS;
}
template <bool GEGLU>
int launch_gemm_sqp(const moca_gemm_params& p, hipStream_t st) {
    constexpr int lds = 5 * (256 + 256) * 64;                    // the whole 160 KiB: the ring; tile statistics live in its one free slot
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_sqp_kernel<GEGLU>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return MOCA_E_LAUNCH;
        attr_set = true;
    }
    moca_gemm_params pl = p;
    pl.reserved2_ = SQP_BLOCKS;
    pl.reserved4_ = (pl.reserved4_ & 0xfd) | (choose_xcd_n(p, (p.M + 255) / 256, p.N / 256, 256, 256) << 8) |
                    (moca_tuning_get(MOCA_TUNE_SQP_WALK) ? 2 : 0);
    hipLaunchKernelGGL((gemm_sqp_kernel<GEGLU>), dim3(SQP_BLOCKS + prefetch_blocks(pl)), dim3(512), lds, st, pl);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

template <int BN, int AMODE, bool FAST>
int launch_gemm_glds(const moca_gemm_params& p, hipStream_t st) {
    const int tiles_m = (p.M + 255) / 256, tiles_n = p.N / BN;
    const int nblk = tiles_m * tiles_n * p.splits;
    constexpr int lds_pipe = 3 * (256 + BN) * ROW_BYTES;
    constexpr int lds_epi = 256 * BN * 4;
    constexpr int lds = lds_pipe > lds_epi ? lds_pipe : lds_epi;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<BN, AMODE, FAST>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return MOCA_E_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_glds_kernel<BN, AMODE, FAST>), dim3(nblk + prefetch_blocks(p)), dim3(512), lds, st, p);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

template <int BN, int AMODE>
int launch_gemm(const moca_gemm_params& p, hipStream_t st) {
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
    const int nblk = tiles_m * tiles_n * p.splits;
    constexpr int lds_pipe = 2 * (BM + BN) * ROW_BYTES;
    constexpr int lds_epi = BM * BN * 4;
    constexpr int lds = lds_pipe > lds_epi ? lds_pipe : lds_epi;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_kernel<BN, AMODE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return MOCA_E_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_f16_kernel<BN, AMODE>), dim3(nblk), dim3(256), lds, st, p);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

}  // namespace

static void normalise_splits(moca_gemm_params& p) {
    if (p.splits < 1) p.splits = 1;
    if (p.splits > (p.K + BK - 1) / BK) p.splits = (p.K + BK - 1) / BK;
    if (p.splits > 1) {   // no empty k range: every split writes its slab
        const int nkt = (p.K + BK - 1) / BK, kts = (nkt + p.splits - 1) / p.splits;
        p.splits = (nkt + kts - 1) / kts;
    }
}

// does this (validated, split-normalised) call run on the 256-row direct-to-LDS kernel (gemm_glds_kernel), and with which BN?
// (mirrors the dispatch order of moca_gemm_f16: w80 family first, then sq256, then g4, then glds)
static int takes_glds_bn(const moca_gemm_params& p) {
    if (takes_ws(p) || takes_w80(p)) return 0;
    const int big_bn = (p.N % 128 == 0) ? 128 : (p.N % 160 == 0 ? 160 : 0);
    if (!(big_bn != 0 && p.M > 128 && !(p.flags & MOCA_FORCE_SMALL_TILE))) return 0;
    const bool use_g4 = !(p.flags & MOCA_EP_OUT_F32) && wants_g4(p);
    if (takes_sq256(p, use_g4)) return 0;
    if (big_bn == 128 && use_g4) return 0;
    return big_bn;
}
// rows per tile of the column sums a MOCA_EP_COLSUM launch leaves behind (0: this call cannot): 320 / 160 on the staggered kernel,
// 256 on the 256-row kernel (fp16 output, no GEGLU, no split-k)
static int colsum_rows(const moca_gemm_params& pp) {
    moca_gemm_params p = pp;
    if (!(p.flags & (MOCA_EP_COLSUM | MOCA_EP_GSTAT))) p.flags |= MOCA_EP_COLSUM;      // (the question is about the call WITH column sums)
    if (p.splits != 1) return 0;
    if ((p.flags & MOCA_EP_GSTAT) && takes_ws(p)) return 32;      // the weight-stationary kernel: finished statistics only, strips of 32 rows
    if (takes_w80s(p)) return w80s_wide(p) ? 160 : 320;
    if (!(p.flags & (MOCA_EP_GEGLU | MOCA_EP_OUT_F32)) && takes_glds_bn(p) != 0) return 256;
    return 0;
}

// columns per column tile of the row sums a MOCA_EP_ROWSUM launch leaves behind (0: this call cannot)
static int rowsum_cols(const moca_gemm_params& pp) {
    moca_gemm_params p = pp;
    p.flags |= MOCA_EP_ROWSUM;
    if (takes_ws(p)) return 80;                       // one partial per wave of the weight-stationary kernel
    if (p.splits != 1 || (p.flags & (MOCA_EP_GEGLU | MOCA_EP_OUT_F32 | MOCA_EP_COLSUM | MOCA_EP_GSTAT | MOCA_EP_LN | MOCA_EP_GELU | MOCA_FORCE_SMALL_TILE))) return 0;
    if (takes_w80s(p)) return w80s_wide(p) ? 320 : 160;
    return takes_glds_bn(p);
}
// does the kernel this call runs on have the MOCA_EP_LNFOLD epilogue?
static bool lnfold_ok(const moca_gemm_params& pp) {
    moca_gemm_params p = pp;
    p.flags |= MOCA_EP_LNFOLD;
    if (p.a_mode != MOCA_A_LINEAR || p.splits != 1) return false;
    if (p.flags & (MOCA_EP_OUT_F32 | MOCA_EP_COLSUM | MOCA_EP_GSTAT | MOCA_EP_LN | MOCA_EP_ROWSUM | MOCA_EP_GELU | MOCA_FORCE_SMALL_TILE)) return false;
    if (takes_sqp(p) || takes_g4p(p)) return true;
    if (takes_w80(p)) return !(p.flags & MOCA_EP_GEGLU) && takes_w80s(p);
    const int big_bn = (p.N % 128 == 0) ? 128 : (p.N % 160 == 0 ? 160 : 0);
    if (!(big_bn != 0 && p.M > 128)) return false;
    const bool use_g4 = wants_g4(p);
    if (takes_sq256(p, use_g4)) return true;
    if (big_bn == 128 && use_g4) return true;
    return !(p.flags & MOCA_EP_GEGLU);                  // the 256-row kernel: plain epilogue only
}

// MOCA_EP_TATTN: the q|k|v projection of a temporal self-attention with the attention finished in the epilogue (SHAPE 3 of the
// staggered kernel: 320 x 192 tiles, rows = 16 frames x 20 pixels, columns = one head)
static bool tattn_ok(const moca_gemm_params& p) {
    if (p.a_mode != MOCA_A_LINEAR || p.splits != 1 || !fast_gather(p) || !buffer_addressable(p)) return false;
    if (p.flags & ~(MOCA_EP_TATTN | MOCA_EP_LNFOLD)) return false;
    if (p.residual || p.rowadd) return false;
    if (p.N % 192 || p.T != 16 || p.HW <= 0 || p.HW % 20 || p.M % (16 * p.HW)) return false;
    if (p.ldo % 4 || p.ldo < p.N / 3) return false;
    return true;
}

extern "C" int moca_gemm_tattn_ok(const moca_gemm_params* pp) {
    if (!pp) return 0;
    moca_gemm_params p = *pp;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.N % 64 || p.K % 8) return 0;
    if (p.splits < 1) p.splits = 1;
    p.flags |= MOCA_EP_TATTN;
    return tattn_ok(p) ? 1 : 0;
}

// two-source A (the virtual torch.cat in front of a ResBlock's skip_connection): a plain linear on the staggered kernels, sources
// split at a multiple of 64 columns, nothing but bias / residual / row add / row sums / GroupNorm statistics in the epilogue
static bool cat_ok(const moca_gemm_params& p) {
    if (!p.a2 || p.a_mode != MOCA_A_LINEAR || p.splits != 1) return false;
    if (p.k1 <= 0 || p.k1 >= p.K || p.k1 % 64 || (p.K - p.k1) % 64 || p.lda % 8 || p.lda2 % 8 || p.lda < p.k1 || p.lda2 < p.K - p.k1) return false;
    if (p.flags & ~(MOCA_EP_COLSUM | MOCA_EP_GSTAT | MOCA_EP_ROWSUM)) return false;
    return takes_w80s(p);
}
extern "C" int moca_gemm_cat_ok(const moca_gemm_params* pp) {
    if (!pp) return 0;
    moca_gemm_params p = *pp;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.N % 64 || p.K % 8) return 0;
    normalise_splits(p);
    return cat_ok(p) ? 1 : 0;
}

// per-row-group weights (wgroup_rows / wgroup_stride): a plain linear on the staggered kernels whose row tiles (160 rows on the
// 160 x 320 tiling, 320 on 320 x 160) lie inside one group; bias / residual / row sums / LayerNorm store loop / column statistics
static bool wgroup_ok(const moca_gemm_params& p) {
    if (p.wgroup_rows <= 0 || p.wgroup_stride < p.N * p.ldw || p.a_mode != MOCA_A_LINEAR || p.a2 || p.splits != 1) return false;
    if (takes_ws(p)) return true;                     // the weight-stationary kernel deals its strips by weight group
    if (p.flags & ~(MOCA_EP_COLSUM | MOCA_EP_GSTAT | MOCA_EP_ROWSUM | MOCA_EP_LN)) return false;
    if (!takes_w80s(p) || p.M % p.wgroup_rows) return false;
    const int tm = w80s_wide(p) ? 160 : 320;
    if (p.wgroup_rows % tm) return false;
    return (int64_t)(p.M / p.wgroup_rows) * p.wgroup_stride * 2 < (1ll << 31);
}
extern "C" int moca_gemm_wgroup_ok(const moca_gemm_params* pp) {
    if (!pp) return 0;
    moca_gemm_params p = *pp;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.N % 64 || p.K % 8) return 0;
    normalise_splits(p);
    return wgroup_ok(p) ? 1 : 0;
}

extern "C" int moca_gemm_rowsum_cols(const moca_gemm_params* pp) {
    if (!pp) return 0;
    moca_gemm_params p = *pp;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.N % 64 || p.K % 8) return 0;
    normalise_splits(p);
    return rowsum_cols(p);
}

extern "C" int moca_gemm_lnfold_ok(const moca_gemm_params* pp) {
    if (!pp) return 0;
    moca_gemm_params p = *pp;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.N % 64 || p.K % 8) return 0;
    normalise_splits(p);
    return lnfold_ok(p) ? 1 : 0;
}

extern "C" int moca_gemm_colsum_rows(const moca_gemm_params* pp) {
    if (!pp) return 0;
    moca_gemm_params p = *pp;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.N % 64 || p.K % 8) return 0;
    normalise_splits(p);
    return colsum_rows(p);
}

extern "C" int moca_gemm_ln_ok(const moca_gemm_params* pp) {
    if (!pp) return 0;
    moca_gemm_params p = *pp;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.N % 64 || p.K % 8) return 0;
    normalise_splits(p);
    return takes_w80t_ln(p) ? 1 : 0;
}

// fp16 split-K slabs (MOCA_TUNE_SLAB_F16): a split-K call of the 256-row kernel with an fp16 output and no GEGLU
static bool slab_f16(const moca_gemm_params& p) {
    return moca_tuning_get(MOCA_TUNE_SLAB_F16) && p.splits > 1 && !(p.flags & (MOCA_EP_GEGLU | MOCA_EP_OUT_F32)) && takes_glds_bn(p) != 0;
}
// can moca_gemm_splitk_groupnorm_f16 finish this (validated, split-normalised) MOCA_EP_SLABS call?  fp16 plain epilogue, one block per
// (statistics group, channel group) slab with the slab in registers
static bool splitk_gn_ok(const moca_gemm_params& p, int HW, int fps) {
    if (p.splits < 2 || (p.flags & ~MOCA_EP_SLABS) || p.N % 32 || (p.N / 32) % 8 || p.N / 32 > 128 || p.up_phase) return false;
    if (HW <= 0 || fps <= 0 || p.M % (HW * fps)) return false;
    const int64_t nchunks = (int64_t)HW * fps * (p.N / 32 / 8);
    return nchunks >= 64 && nchunks <= 4096;
}
extern "C" int moca_gemm_splitk_groupnorm_ok(const moca_gemm_params* pp, int32_t HW, int32_t frames_per_stat) {
    if (!pp) return 0;
    moca_gemm_params p = *pp;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.N % 64 || p.K % 8) return 0;
    normalise_splits(p);
    return splitk_gn_ok(p, HW, frames_per_stat) ? 1 : 0;
}
extern "C" int moca_gemm_splitk_groupnorm_f16(const moca_gemm_params* pp, void* y, const float* gamma, const float* beta, int32_t HW,
                                              int32_t frames_per_stat, float eps, int32_t silu, int32_t write_x, void* stream) {
    if (!pp || !y || !gamma || !beta) return MOCA_E_BADARG;
    moca_gemm_params p = *pp;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.N % 64 || p.K % 8 || !p.splitk_ws || (write_x && !p.out)) return MOCA_E_BADARG;
    if (p.ldo % 8 || (p.residual && p.ldr % 8) || (p.rowadd && (p.ld_rowadd % 8 || p.rowadd_div <= 0))) return MOCA_E_BADARG;
    normalise_splits(p);
    if (!splitk_gn_ok(p, HW, frames_per_stat)) return MOCA_E_BADARG;
    p.reserved4_ = slab_f16(p) ? 4 : 0;
    const int cpg = p.N / 32, R = HW * frames_per_stat;
    const int nchunks = R * (cpg / 8);
    const int cpt = (nchunks + 1023) / 1024;
    int thr = ((nchunks + cpt - 1) / cpt + 63) / 64 * 64;
    if (thr < (cpg + 63) / 64 * 64) thr = (cpg + 63) / 64 * 64;      // threads tid < cpg fill s_sc / s_sh (cpg = 128 with 64 chunks: ADVICE r5)
    const int n_slabs = (p.M / R) * 32;
    const double inv_count = 1.0 / ((double)R * cpg);
    hipStream_t st = moca_stream(stream);
    half_t* yo = reinterpret_cast<half_t*>(y);
#define MOCA_SKGN(CPT) hipLaunchKernelGGL(splitk_gn_kernel<CPT>, dim3(n_slabs), dim3(thr), 0, st, p, yo, gamma, beta, R, cpg, inv_count, eps, silu, write_x)
    if (cpt == 1) MOCA_SKGN(1); else if (cpt == 2) MOCA_SKGN(2); else if (cpt == 3) MOCA_SKGN(3); else MOCA_SKGN(4);
#undef MOCA_SKGN
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int64_t moca_gemm_splitk_ws_bytes(int32_t M, int32_t N, int32_t splits) {
    return splits > 1 ? (int64_t)splits * M * N * 4 : 0;
}

extern "C" int moca_gemm_f16(const moca_gemm_params* pp, void* stream) {
    if (!pp) return MOCA_E_BADARG;
    moca_gemm_params p = *pp;
    if (p.splits < 1) p.splits = 1;
    if (!p.a || !p.w || !p.out) return MOCA_E_BADARG;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return MOCA_E_BADARG;
    if (p.N % 64 || p.K % 8 || p.ldw % BK || p.ldw < ((p.K + BK - 1) / BK) * BK) return MOCA_E_BADARG;
    const bool geglu = p.flags & MOCA_EP_GEGLU;
    if (geglu && p.N % 128) return MOCA_E_BADARG;
    if (p.ldo % 8 || (p.residual && p.ldr % 8) || (p.rowadd && (p.ld_rowadd % 8 || p.rowadd_div <= 0))) return MOCA_E_BADARG;
    if (p.splits > 1 && !p.splitk_ws) return MOCA_E_BADARG;
    if (p.prefetch_kib < 0 || (p.prefetch_kib > 0 && (!p.prefetch || (reinterpret_cast<uintptr_t>(p.prefetch) & 15)))) return MOCA_E_BADARG;
    // the plain-GELU epilogue (CLIP text MLP) exists in the 128-row kernel only
    if ((p.flags & MOCA_EP_GELU) && (geglu || p.splits != 1 || !((p.flags & MOCA_FORCE_SMALL_TILE) || p.M <= 128))) return MOCA_E_BADARG;
    normalise_splits(p);
    if ((p.flags & MOCA_EP_SLABS) && (p.splits < 2 || (p.flags & ~MOCA_EP_SLABS))) return MOCA_E_BADARG;   // ask moca_gemm_splitk_groupnorm_ok() first
    if (p.up_phase && p.a_mode != MOCA_A_CONV3X3) return MOCA_E_BADARG;
    switch (p.a_mode) {
        case MOCA_A_LINEAR:
            if (p.a2) {                               // ask moca_gemm_cat_ok() first
                if (!cat_ok(p)) return MOCA_E_BADARG;
            } else if (p.lda % 8 || p.lda < p.K) return MOCA_E_BADARG;
            break;
        case MOCA_A_CONV3X3:
            if (p.up_phase < 0 || p.up_phase > 4) return MOCA_E_BADARG;
            if (p.C % 8 || p.K != (p.up_phase ? 4 : 9) * p.C || p.inH <= 0 || p.inW <= 0 || p.outH <= 0 || p.outW <= 0) return MOCA_E_BADARG;
            // one phase of upsample + conv: a 2 x 2 conv on the low-resolution grid, rows scattered by the plain store loop of the
            // 256- / 320-row kernels (fast gather), nothing else in the epilogue
            if (p.up_phase && (p.stride != 1 || p.up || p.nopad_lo || p.splits != 1 || p.M <= 160 || !fast_gather(p) || p.residual || p.rowadd ||
                               (p.flags & (MOCA_EP_GEGLU | MOCA_EP_OUT_F32 | MOCA_EP_COLSUM | MOCA_EP_ROWSUM | MOCA_EP_LN | MOCA_EP_LNFOLD |
                                          MOCA_EP_TATTN | MOCA_EP_GELU | MOCA_FORCE_SMALL_TILE)) || (p.N % 128 && p.N % 160))) return MOCA_E_BADARG;
            if (p.stride != 1 && p.stride != 2) return MOCA_E_BADARG;
            if (p.nopad_lo != 0 && (p.nopad_lo != 1 || p.stride != 2 || p.up || (p.inH | p.inW) & 1)) return MOCA_E_BADARG;
            if (p.up && (p.stride != 1 || p.outH != 2 * p.inH || p.outW != 2 * p.inW)) return MOCA_E_BADARG;
            if (!p.up && p.stride == 1 && (p.outH != p.inH || p.outW != p.inW)) return MOCA_E_BADARG;
            if (p.stride == 2 && (p.outH != (p.inH - 1) / 2 + 1 || p.outW != (p.inW - 1) / 2 + 1)) return MOCA_E_BADARG;
            if (p.M % (p.outH * p.outW)) return MOCA_E_BADARG;
            break;
        case MOCA_A_TCONV3:
            if (p.C % 8 || p.K != 3 * p.C || p.T <= 0 || p.HW <= 0 || p.M % (p.T * p.HW)) return MOCA_E_BADARG;
            break;
        default:
            return MOCA_E_BADARG;
    }
    if (p.a2 && p.a_mode != MOCA_A_LINEAR) return MOCA_E_BADARG;
    if (p.gstat_cpg < 0 || p.gstat_coff < 0 || ((p.gstat_cpg || p.gstat_coff) && !(p.flags & MOCA_EP_GSTAT))) return MOCA_E_BADARG;
    hipStream_t st = moca_stream(stream);
    const bool wide = (p.N % 128 == 0);
    int rc;
    // large-tile direct-to-LDS kernel whenever a 256-row tile is at least half full
    const int big_bn = (p.N % 128 == 0) ? 128 : (p.N % 160 == 0 ? 160 : 0);
    const bool use_big = big_bn != 0 && p.M > 128 && !(p.flags & MOCA_FORCE_SMALL_TILE);
    const bool fastp = fast_gather(p);
    // g4 (4 waves, two blocks per CU) wins where the epilogue is VALU-heavy and K is short (GEGLU at C = 320 / 640:
    // one block's erf-GELU epilogue runs under the other block's MFMAs, -5 % on the same device); the 8-wave kernel's
    // deeper pipeline wins everywhere else (K >= 1280: 1137 vs 880 TFLOP/s).  MOCA_TUNE_GEMM_G4 = 0 / 2 forces never / always (tests).
    const bool use_g4 = !(p.flags & MOCA_EP_OUT_F32) && wants_g4(p);
    // w80 (320 x 160 tiles, 80 x 80 wave tiles): every non-GEGLU contraction whose N is a multiple of 160 and whose
    // 320-row tiles fill the chip.  (MOCA_TUNE_GEMM_W80: 0 never, 2 drops the tile-count rule -- tests.)
    const bool use_w80 = takes_w80(p);
    if ((p.flags & MOCA_EP_COLSUM) && !(p.colsum && colsum_rows(p) != 0)) return MOCA_E_BADARG;   // ask moca_gemm_colsum_rows() first
    if ((p.flags & MOCA_EP_LN) && !(p.ln_gamma && p.ln_beta && p.ln_out && p.ld_ln % 8 == 0 && takes_w80t_ln(p))) return MOCA_E_BADARG;   // ask moca_gemm_ln_ok() first
    if (p.flags & MOCA_EP_GSTAT) {                    // same kernels as MOCA_EP_COLSUM; a row tile must lie inside one statistics group
        const int rows = colsum_rows(p);
        if (!(p.gstat && rows != 0 && !(p.flags & MOCA_EP_COLSUM) && p.gstat_rows > 0 && p.gstat_rows % rows == 0 && p.M % p.gstat_rows == 0 &&
              (p.gstat_cpg > 0 ? (p.gstat_coff + p.N - 1) / p.gstat_cpg < 32 : (p.N % 32 == 0 && p.gstat_coff == 0)))) return MOCA_E_BADARG;
    }
    if ((p.flags & MOCA_EP_ROWSUM) && !(p.rowsum && rowsum_cols(p) != 0)) return MOCA_E_BADARG;             // ask moca_gemm_rowsum_cols() first
    if (p.wgroup_rows < 0 || (p.wgroup_rows > 0 && !wgroup_ok(p))) return MOCA_E_BADARG;                    // ask moca_gemm_wgroup_ok() first
    p.reserved4_ = 0;                                 // (bits 8.. carry the XCD partition chosen by the launcher)
    // bit 0: output rows leave with non-temporal stores when the output is at least half the 256 MiB Infinity Cache (see out_streams)
    if ((int64_t)p.M * (geglu ? p.N / 2 : p.N) * 2 >= (128ll << 20)) p.reserved4_ |= 1;
    if (slab_f16(p)) p.reserved4_ |= 4;               // bit 2: fp16 split-K slabs (MOCA_TUNE_SLAB_F16; the 256-row kernel only)
    if (p.flags & MOCA_EP_TATTN) {                    // ask moca_gemm_tattn_ok() first
        if (!tattn_ok(p) || ((p.flags & MOCA_EP_LNFOLD) && !(p.lnf_part && p.lnf_wsum && p.lnf_nparts >= 1))) return MOCA_E_BADARG;
        return launch_gemm_w80s<MOCA_A_LINEAR, 3>(p, st);
    }
    if ((p.flags & MOCA_EP_LNFOLD) && !(p.lnf_part && p.lnf_wsum && p.lnf_nparts >= 1 && lnfold_ok(p))) return MOCA_E_BADARG;   // ask moca_gemm_lnfold_ok() first
    if (takes_ws(p)) {
        rc = moca_gemm_ws_launch(p, st);
    } else if (p.a2) {                                // (cat_ok: a staggered-kernel call)
        rc = launch_gemm_w80_mode(p, st);
    } else if (takes_sqp(p)) {
        rc = (p.flags & MOCA_EP_GEGLU) ? launch_gemm_sqp<true>(p, st) : launch_gemm_sqp<false>(p, st);
    } else if (takes_g4p(p)) {
        rc = (p.flags & MOCA_EP_GEGLU) ? launch_gemm_g4p<true>(p, st) : launch_gemm_g4p<false>(p, st);
    } else if (use_w80) {
        rc = launch_gemm_w80_mode(p, st);
    } else if (takes_sq256(p, use_g4)) {
        rc = launch_gemm_w80s<MOCA_A_LINEAR, 2>(p, st);
    } else if (use_big && big_bn == 128 && use_g4) {
        if (p.a_mode == MOCA_A_LINEAR) rc = fastp ? launch_gemm_g4<MOCA_A_LINEAR, true>(p, st) : launch_gemm_g4<MOCA_A_LINEAR, false>(p, st);
        else if (p.a_mode == MOCA_A_CONV3X3) rc = fastp ? launch_gemm_g4<MOCA_A_CONV3X3, true>(p, st) : launch_gemm_g4<MOCA_A_CONV3X3, false>(p, st);
        else rc = fastp ? launch_gemm_g4<MOCA_A_TCONV3, true>(p, st) : launch_gemm_g4<MOCA_A_TCONV3, false>(p, st);
    } else if (use_big && big_bn == 128) {
        if (p.a_mode == MOCA_A_LINEAR) rc = fastp ? launch_gemm_glds<128, MOCA_A_LINEAR, true>(p, st) : launch_gemm_glds<128, MOCA_A_LINEAR, false>(p, st);
        else if (p.a_mode == MOCA_A_CONV3X3) rc = fastp ? launch_gemm_glds<128, MOCA_A_CONV3X3, true>(p, st) : launch_gemm_glds<128, MOCA_A_CONV3X3, false>(p, st);
        else rc = fastp ? launch_gemm_glds<128, MOCA_A_TCONV3, true>(p, st) : launch_gemm_glds<128, MOCA_A_TCONV3, false>(p, st);
    } else if (use_big) {
        if (p.a_mode == MOCA_A_LINEAR) rc = fastp ? launch_gemm_glds<160, MOCA_A_LINEAR, true>(p, st) : launch_gemm_glds<160, MOCA_A_LINEAR, false>(p, st);
        else if (p.a_mode == MOCA_A_CONV3X3) rc = fastp ? launch_gemm_glds<160, MOCA_A_CONV3X3, true>(p, st) : launch_gemm_glds<160, MOCA_A_CONV3X3, false>(p, st);
        else rc = fastp ? launch_gemm_glds<160, MOCA_A_TCONV3, true>(p, st) : launch_gemm_glds<160, MOCA_A_TCONV3, false>(p, st);
    } else if (wide) {
        if (p.a_mode == MOCA_A_LINEAR) rc = launch_gemm<128, MOCA_A_LINEAR>(p, st);
        else if (p.a_mode == MOCA_A_CONV3X3) rc = launch_gemm<128, MOCA_A_CONV3X3>(p, st);
        else rc = launch_gemm<128, MOCA_A_TCONV3>(p, st);
    } else {
        if (p.a_mode == MOCA_A_LINEAR) rc = launch_gemm<64, MOCA_A_LINEAR>(p, st);
        else if (p.a_mode == MOCA_A_CONV3X3) rc = launch_gemm<64, MOCA_A_CONV3X3>(p, st);
        else rc = launch_gemm<64, MOCA_A_TCONV3>(p, st);
    }
    if (rc != MOCA_OK) return rc;
    if (p.splits > 1 && !(p.flags & MOCA_EP_SLABS)) {
        const int out_n = geglu ? p.N / 2 : p.N;
        const int64_t total = (int64_t)p.M * (out_n / 8);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
        MOCA_CHECK_LAUNCH();
    }
    return MOCA_OK;
}

// The library's version string lives beside the kernels a diagnostic build changes (`make stamps` / `gndiag` / `diagx` recompile this file
// only): such a build -- stamps, timing-only variants with wrong results -- reports "DIAG:<name>", and moca_video_amd/lib.py refuses to
// load it unless MOCA_HIP_DIAG=1 is set.
#ifdef MOCA_DIAG_NAME
extern "C" const char* moca_version(void) { return "moca_hip 0.1 (gfx950) DIAG:" MOCA_DIAG_NAME; }
#else
extern "C" const char* moca_version(void) { return "moca_hip 0.1 (gfx950)"; }
#endif

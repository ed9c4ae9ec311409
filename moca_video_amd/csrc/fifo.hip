// One outer iteration of the MoCA-FIFO loop (scripts/evaluation/funcs.py:305-371) as device-side work: the 72-frame
// latent queue is a RING in HBM whose head index, iteration counter and RNG seed live in a small device-resident state
// block, so that an iteration -- window gather, batched UNet, classifier-free guidance + MoCA ddim_step of all 2n windows,
// write-back, emission, FreeInit mix, queue shift -- is a fixed launch sequence the host captures ONCE into a hipGraph and
// replays without reading anything back.  fp32 throughout, compiled with -ffp-contract=off like sampler.hip (the reference
// evaluates these expressions as separate fp32 torch ops).
#include "common.h"

namespace {

// ---- Philox4x32-10 (Salmon et al., SC'11) + Box-Muller: the noise of ddim.py:561 / funcs.py:92 drawn on the device ------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1;
    c[3] = (uint32_t)p0;
    c[0] = n0;
    c[2] = n2;
}

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

__global__ __launch_bounds__(256) void fifo_randn_kernel(const moca_fifo_state* __restrict__ st, float* __restrict__ out, int64_t n) {
    if (st->ext_noise) return;                      // the host filled the buffer for this iteration (fixtures)
    const uint32_t k0 = st->seed_lo, k1 = st->seed_hi, it = (uint32_t)st->iter;
    const int64_t n4 = (n + 3) / 4;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
        uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), it, 0x4d6f4341u};
        philox4x32_10(c, k0, k1);
        float z[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float u1 = ((float)c[2 * h] + 0.5f) * 2.3283064365386963e-10f;         // (0, 1]
            const float u2 = ((float)c[2 * h + 1] + 0.5f) * 2.3283064365386963e-10f;
            const float r = sqrtf(-2.0f * logf(u1));
            float sn, cs;
            sincosf(6.283185307179586f * u2, &sn, &cs);
            z[2 * h] = r * cs;
            z[2 * h + 1] = r * sn;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (q * 4 + j < n) out[q * 4 + j] = z[j];
    }
}

// window w, repeat r: x[(r nW + w)][c][j][p] = queue[c][(head + win_start[w] + j) mod Q][p] (funcs.py:315, the .clone() of
// the window); anchor[c][p] = queue[c][head][p] (funcs.py:88: the frame the shift dequeues, read AFTER the iteration's write-backs.
// With lookahead no write-back touches frame 0, so the pre-UNet gather may fetch it; without lookahead the rank-0 window rewrites
// frame 0 (funcs.py:353-354) and the host asks for the anchor in a second call behind the step kernel: x == NULL, nW == 0)
__global__ __launch_bounds__(256) void fifo_gather_kernel(const moca_fifo_state* __restrict__ st, const float* __restrict__ queue,
                                                          float* __restrict__ x, float* __restrict__ anchor,
                                                          const int32_t* __restrict__ win_start, int nW, int reps, int C, int Q,
                                                          int f, int HW) {
    const int head = st->head;
    const int64_t per_rep = x ? (int64_t)nW * C * f * HW : 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_rep; i += (int64_t)gridDim.x * 256) {
        const int p = (int)(i % HW);
        int64_t r = i / HW;
        const int j = (int)(r % f); r /= f;
        const int c = (int)(r % C);
        const int w = (int)(r / C);
        int fr = head + win_start[w] + j;
        fr %= Q;
        const float v = queue[((int64_t)c * Q + fr) * HW + p];
        for (int rr = 0; rr < reps; ++rr) x[rr * per_rep + i] = v;
    }
    if (anchor) {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)C * HW; i += (int64_t)gridDim.x * 256) {
            const int p = (int)(i % HW), c = (int)(i / HW);
            anchor[i] = queue[((int64_t)c * Q + head) * HW + p];
        }
    }
}

// classifier-free guidance (ddim.py:372) + MoCA ddim_step (ddim.py:405-430,556-609) of every window of an iteration + the
// write-back of funcs.py:351-354.  One thread per (window, channel, pixel); the frame axis is walked sequentially (momentum
// EMA, previous-frame dependence).  x is the gathered PRE-iteration copy of the windows: window r+1 reads the frames window
// r writes back, and in the reference's reversed-rank order every window sees pre-iteration values.
__global__ __launch_bounds__(256) void fifo_step_windows_kernel(moca_fifo_step_params P) {
    const int head = P.state->head;
    const int C = P.C, f = P.f, HW = P.HW, Q = P.Q;
    const int64_t total = (int64_t)P.nW * C * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int p = (int)(i % HW);
        const int64_t wc = i / HW;
        const int c = (int)(wc % C);
        const int w = (int)(wc / C);
        const int64_t base = wc * f * HW + p;
        float prev = 0.f, mom_prev = P.momentum[base];        // momentum[:, :, 0] is never written (ddim.py:424: frames i >= 1 only)
        const float cnd = P.cond ? P.cond[(int64_t)c * HW + p] : 0.f;
        const int start = P.win_start[w];
        for (int j = 0; j < f; ++j) {
            const float* cf = P.coef + ((int64_t)w * f + j) * 6;
            const int64_t o = base + (int64_t)j * HW;
            float et = P.eps_c[o];
            if (P.eps_u) {
                const float u = P.eps_u[o];
                et = u + P.cfg_scale * (et - u);               // :372
            }
            float p0 = (P.x[o] - cf[3] * et) / cf[0];          // :415
            const float dir = cf[4] * et;                      // :418
            if (j >= 1) {
                float g = p0 - prev;                           // :422
                g = g + 1.5f * dir;                            // :423
                const float m = P.beta * mom_prev + P.one_minus_beta * g;   // :424-427
                P.momentum[o] = m;
                mom_prev = m;
                p0 = p0 + cf[5] * m;                           // :428-430,557
            }
            prev = p0;                                         // :559
            const float nz = cf[2] * P.noise[o];               // :561
            const float xp = cf[1] * p0 + dir + nz;            // :562
            if (P.x_prev) P.x_prev[o] = xp;
            if (P.queue && j >= P.wb_from) {                   // funcs.py:351-354
                int fr = head + start + j;
                fr %= Q;
                P.queue[((int64_t)c * Q + fr) * HW + p] = xp;
            }
            if (P.pred_x0) {
                const int mf = P.mask ? P.mask_frame[w * f + j] : -1;
                if (mf >= 0) {                                 // :565-590
                    const int ms = (head + mf) % Q;
                    if (P.mask_sums[ms] != 0.f && P.mask[(int64_t)ms * HW + p] > 0.5f) p0 = cnd * P.enh[w * f + j];
                } else if (P.sam_eff && P.sam_idx[w * f + j] >= 0) {    // :592-606 -> _apply_segmentation :847,897-901 (factor 2)
                    if (P.sam_eff[((int64_t)w * f + j) * HW + p] > 0.5f) p0 = cnd * 2.0f;
                }
                P.pred_x0[o] = P.one_minus_gamma * p0 + P.gamma * nz;   // :609
            }
        }
    }
}

// funcs.py:357-371 without the decode: emitted[iter mod n_slots] = queue frame `emit_frame`; the slot of the dequeued frame 0
// receives the FreeInit-mixed frame (it becomes the tail once the head moves on); the mask queue keeps its last frame (:116)
__global__ __launch_bounds__(256) void fifo_advance_copy_kernel(const moca_fifo_state* __restrict__ st, float* __restrict__ queue,
                                                                const float* __restrict__ newframe, float* __restrict__ emitted,
                                                                int n_slots, int emit_frame, float* __restrict__ mask,
                                                                float* __restrict__ mask_sums, int C, int Q, int HW) {
    const int head = st->head;
    const int slot = n_slots > 0 ? st->iter % n_slots : 0;
    const int ef = (head + emit_frame) % Q, tail = (head + Q - 1) % Q;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)C * HW; i += (int64_t)gridDim.x * 256) {
        const int p = (int)(i % HW), c = (int)(i / HW);
        if (emitted) emitted[(int64_t)slot * C * HW + i] = queue[((int64_t)c * Q + ef) * HW + p];
        queue[((int64_t)c * Q + head) * HW + p] = newframe[i];
        if (mask && c == 0) mask[(int64_t)head * HW + p] = mask[(int64_t)tail * HW + p];
    }
    if (mask_sums && blockIdx.x == 0 && threadIdx.x == 0) mask_sums[head] = mask_sums[tail];
}

__global__ void fifo_advance_bump_kernel(moca_fifo_state* st, int Q) {
    st->head = (st->head + 1) % Q;
    st->iter = st->iter + 1;
    st->ext_noise = 0;
}

// ---- base sampling (ddim.py:226-252): one DDIM step of `ddim_sampling` as device-side work.  Step i of the loop uses schedule
// index S - 1 - i (ddim.py:238); i = state->iter mod S, so a captured step replays for every i.
// rows[0:n] = table[S - 1 - i]: the timestep rows of the UNet plan (ddim.py:239 `ts = torch.full((b,), step)`)
__global__ __launch_bounds__(256) void base_set_timestep_kernel(const moca_fifo_state* __restrict__ st, const int64_t* __restrict__ table,
                                                                int S, int64_t* __restrict__ rows, int n) {
    const int64_t t = table[S - 1 - st->iter % S];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) rows[i] = t;
}

// guidance (ddim.py:304) + the tail of p_sample_ddim (ddim.py:328-357, as ddim_update_kernel) with the coefficients of schedule
// index S - 1 - i: coef[idx] = {sqrt(a_t), sqrt(a_prev), sigma_t, sqrt(1 - a_t), sqrt(1 - a_prev - sigma_t^2), scale_t, scale_prev, -}.
// x is updated IN PLACE (it is the UNet plan's input buffer: the next step reads it there).
__global__ __launch_bounds__(256) void base_step_kernel(const moca_fifo_state* __restrict__ st, float* __restrict__ x,
                                                        const float* __restrict__ eps_c, const float* __restrict__ eps_u,
                                                        const float* __restrict__ noise, float* __restrict__ pred_x0,
                                                        const float* __restrict__ coef, int S, float cfg_scale, int use_scale, int64_t n) {
    const float* cf = coef + (int64_t)(S - 1 - st->iter % S) * 8;
    const float sqrt_at = cf[0], sqrt_aprev = cf[1], sigma_t = cf[2], s1m = cf[3], dir_coef = cf[4], scale_t = cf[5], scale_prev = cf[6];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float et = eps_c[i];
        if (eps_u) {
            const float u = eps_u[i];
            et = u + cfg_scale * (et - u);                      // :304
        }
        float p0 = (x[i] - s1m * et) / sqrt_at;                 // :339
        const float dir = dir_coef * et;                        // :343
        const float nz = sigma_t * noise[i];                    // :345
        float xp;
        if (use_scale) {
            p0 = p0 / scale_t;                                  // :353
            xp = sqrt_aprev * scale_prev * p0 + dir + nz;       // :354
        } else {
            xp = sqrt_aprev * p0 + dir + nz;                    // :356
        }
        x[i] = xp;
        if (pred_x0) pred_x0[i] = p0;
    }
}

__global__ void state_bump_kernel(moca_fifo_state* st) {
    st->iter = st->iter + 1;
    st->ext_noise = 0;
}

__global__ __launch_bounds__(256) void mask_frame_sums_kernel(const float* __restrict__ mask, float* __restrict__ sums, int HW) {
    __shared__ float red[4];
    const int fm = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < HW; i += 256) s += mask[(int64_t)fm * HW + i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[fm] = red[0] + red[1] + red[2] + red[3];
}

// funcs.py:53-79: queue frame j = sqrt(a_j) z[frame_idx_j] + sqrt(1 - a_j) noise_j, all B videos; coefficients come from the host,
// evaluated as the reference's 0-dim fp32 tensors are (alpha ** 0.5, (1 - alpha) ** 0.5); mul, mul, add: no contraction
__global__ __launch_bounds__(256) void fifo_prepare_kernel(const float* __restrict__ z, const float* __restrict__ noise, float* __restrict__ out,
                                                           const float* __restrict__ ca, const float* __restrict__ cb,
                                                           const int32_t* __restrict__ fidx, int BC, int Tz, int Q, int HW) {
    const int64_t total = (int64_t)BC * Q * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int p = (int)(i % HW);
        const int64_t r = i / HW;
        const int j = (int)(r % Q);
        const int64_t bc = r / Q;
        const float a = ca[j] * z[(bc * Tz + fidx[j]) * HW + p];
        const float b = cb[j] * noise[i];
        out[i] = a + b;
    }
}

// ---- the mask bookkeeping of `_apply_segmentation` (ddim.py:739-903) for the windows of one iteration, candidates precomputed ----
// counts of (a > 0.5 && b > 0.5), (a > 0.5 || b > 0.5) over a frame, every thread gets both (calculate_iou, ddim.py:919-930)
__device__ __forceinline__ void sam_block_count2(const float* __restrict__ a, const float* __restrict__ b, int HW, int* sh, int& inter,
                                                 int& uni) {
    int ci = 0, cu = 0;
    for (int p = threadIdx.x; p < HW; p += 256) {
        const bool x = a[p] > 0.5f, y = b[p] > 0.5f;
        ci += (x && y) ? 1 : 0;
        cu += (x || y) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ci += __shfl_xor(ci, o, 64);
        cu += __shfl_xor(cu, o, 64);
    }
    __syncthreads();                                   // (the previous reduction's readers are done with sh)
    if ((threadIdx.x & 63) == 0) {
        sh[2 * (threadIdx.x >> 6)] = ci;
        sh[2 * (threadIdx.x >> 6) + 1] = cu;
    }
    __syncthreads();
    inter = sh[0] + sh[2] + sh[4] + sh[6];
    uni = sh[1] + sh[3] + sh[5] + sh[7];
}

__device__ __forceinline__ float sam_block_sum(const float* __restrict__ a, int HW, float* sh) {
    float s = 0.f;
    for (int p = threadIdx.x; p < HW; p += 256) s += a[p];
    s = wave_sum(s);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// One block per window (= one `ddim_step` call: `pre_masks = None` at its start, ddim.py:391), frames in order.  Per frame with
// t <= 300 (:592): no detection -> the previous masks, or no injection if there are none (:788-793); IoU(new, previous) < 0.5 ->
// the previous masks (:804-807, mean over zip() pairs :905-943); then the masks are applied in order, one that covers > 80 % of
// the frame resets what the earlier ones injected (:820-822).  "previous masks" is always some earlier frame's candidate set, so
// it is carried as (offset, count) into the candidate pool.  Every branch condition is block-uniform.
__global__ __launch_bounds__(256) void sam_select_kernel(const float* __restrict__ cand, const int32_t* __restrict__ cand_off,
                                                         const int32_t* __restrict__ ncand, const int64_t* __restrict__ t_rows,
                                                         float* __restrict__ eff, int32_t* __restrict__ eff_idx, int f, int HW) {
    __shared__ int sh_i[8];
    __shared__ float sh_f[4];
    const int w = blockIdx.x;
    int pre_off = -1, pre_n = 0;
    const float big = (float)(0.8 * (double)HW);       // `mask.sum() > 0.8 * mask.numel()`: a python float against an fp32 sum
    for (int i = 0; i < f; ++i) {
        const int wf = w * f + i;
        int use_off = -1, use_n = 0;
        if (t_rows[wf] <= 300) {
            const int n = ncand[wf];
            if (n > 0) {
                use_off = cand_off[wf];
                use_n = n;
                if (pre_off >= 0) {
                    const int pairs = n < pre_n ? n : pre_n;
                    float acc = 0.f;
                    for (int k = 0; k < pairs; ++k) {
                        int inter, uni;
                        sam_block_count2(cand + (int64_t)(use_off + k) * HW, cand + (int64_t)(pre_off + k) * HW, HW, sh_i, inter, uni);
                        acc += uni == 0 ? 1.0f : (float)inter / (float)uni;
                    }
                    if (acc / (float)pairs < 0.5f) {
                        use_off = pre_off;
                        use_n = pre_n;
                    }
                }
            } else if (pre_off >= 0) {
                use_off = pre_off;
                use_n = pre_n;
            }
        }
        if (use_off < 0) {                             // t > 300, or nothing detected yet: pred_x0 is left alone
            if (threadIdx.x == 0) eff_idx[wf] = -1;
            continue;
        }
        pre_off = use_off;
        pre_n = use_n;
        float* e = eff + (int64_t)wf * HW;
        for (int p = threadIdx.x; p < HW; p += 256) e[p] = 0.f;
        for (int m = 0; m < use_n; ++m) {
            const float* mk = cand + (int64_t)(use_off + m) * HW;
            const bool reset = sam_block_sum(mk, HW, sh_f) > big;
            for (int p = threadIdx.x; p < HW; p += 256) e[p] = reset ? 0.f : ((mk[p] > 0.5f) ? 1.0f : e[p]);
        }
        if (threadIdx.x == 0) eff_idx[wf] = i;
    }
}

inline int grid_for(int64_t total) {
    int64_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int moca_fifo_randn_f32(const moca_fifo_state* state, float* out, int64_t n, void* stream) {
    if (!state || !out || n <= 0) return MOCA_E_BADARG;
    hipLaunchKernelGGL(fifo_randn_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, moca_stream(stream), state, out, n);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_fifo_gather_windows_f32(const moca_fifo_state* state, const float* queue, float* x, float* anchor,
                                            const int32_t* win_start, int32_t nW, int32_t reps, int32_t C, int32_t Q, int32_t f,
                                            int32_t HW, void* stream) {
    if (!state || !queue || C <= 0 || Q <= 0 || HW <= 0) return MOCA_E_BADARG;
    if (x ? (!win_start || nW <= 0 || reps <= 0 || f <= 0 || f > Q) : !anchor) return MOCA_E_BADARG;     // x == NULL: the anchor only
    hipLaunchKernelGGL(fifo_gather_kernel, dim3(grid_for(x ? (int64_t)nW * C * f * HW : (int64_t)C * HW)), dim3(256), 0, moca_stream(stream),
                       state, queue, x, anchor, win_start, nW, reps, C, Q, f, HW);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_fifo_step_windows_f32(const moca_fifo_step_params* p, void* stream) {
    if (!p || !p->state || !p->x || !p->eps_c || !p->noise || !p->momentum || !p->coef || !p->win_start) return MOCA_E_BADARG;
    if (p->nW <= 0 || p->C <= 0 || p->f <= 0 || p->HW <= 0 || p->wb_from < 0) return MOCA_E_BADARG;
    if (p->queue && (p->Q < p->f)) return MOCA_E_BADARG;
    if (p->mask && (!p->mask_sums || !p->mask_frame || !p->enh || p->Q <= 0)) return MOCA_E_BADARG;
    if (p->sam_eff && (!p->sam_idx || p->mask)) return MOCA_E_BADARG;      // one branch of ddim.py:565-606 per call
    if (!p->queue && !p->x_prev && !p->pred_x0) return MOCA_E_BADARG;
    hipLaunchKernelGGL(fifo_step_windows_kernel, dim3(grid_for((int64_t)p->nW * p->C * p->HW)), dim3(256), 0, moca_stream(stream), *p);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_sam_select_masks_f32(const float* cand, const int32_t* cand_off, const int32_t* ncand, const int64_t* t_rows,
                                         float* eff, int32_t* eff_idx, int32_t nW, int32_t f, int32_t HW, void* stream) {
    if (!cand || !cand_off || !ncand || !t_rows || !eff || !eff_idx || nW <= 0 || f <= 0 || HW <= 0) return MOCA_E_BADARG;
    hipLaunchKernelGGL(sam_select_kernel, dim3(nW), dim3(256), 0, moca_stream(stream), cand, cand_off, ncand, t_rows, eff, eff_idx, f, HW);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_fifo_advance_f32(moca_fifo_state* state, float* queue, const float* newframe, float* emitted, int32_t n_slots,
                                     int32_t emit_frame, float* mask, float* mask_sums, int32_t C, int32_t Q, int32_t HW,
                                     void* stream) {
    if (!state || !queue || !newframe || C <= 0 || Q <= 0 || HW <= 0 || emit_frame < 0 || emit_frame >= Q) return MOCA_E_BADARG;
    if (emitted && n_slots <= 0) return MOCA_E_BADARG;
    if (mask && !mask_sums) return MOCA_E_BADARG;
    hipStream_t st = moca_stream(stream);
    hipLaunchKernelGGL(fifo_advance_copy_kernel, dim3(grid_for((int64_t)C * HW)), dim3(256), 0, st, state, queue, newframe, emitted,
                       n_slots, emit_frame, mask, mask_sums, C, Q, HW);
    MOCA_CHECK_LAUNCH();
    hipLaunchKernelGGL(fifo_advance_bump_kernel, dim3(1), dim3(1), 0, st, state, Q);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_fifo_prepare_queue_f32(const float* z, const float* noise, float* queue, const float* coef_z, const float* coef_noise,
                                           const int32_t* frame_idx, int32_t BC, int32_t Tz, int32_t Q, int32_t HW, void* stream) {
    if (!z || !noise || !queue || !coef_z || !coef_noise || !frame_idx || BC <= 0 || Tz <= 0 || Q <= 0 || HW <= 0) return MOCA_E_BADARG;
    hipLaunchKernelGGL(fifo_prepare_kernel, dim3(grid_for((int64_t)BC * Q * HW)), dim3(256), 0, moca_stream(stream), z, noise, queue, coef_z,
                       coef_noise, frame_idx, BC, Tz, Q, HW);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_base_set_timestep(const moca_fifo_state* state, const int64_t* table, int32_t S, int64_t* rows, int32_t n, void* stream) {
    if (!state || !table || !rows || S <= 0 || n <= 0) return MOCA_E_BADARG;
    hipLaunchKernelGGL(base_set_timestep_kernel, dim3(grid_for(n)), dim3(256), 0, moca_stream(stream), state, table, S, rows, n);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_base_ddim_step_f32(moca_fifo_state* state, float* x, const float* eps_c, const float* eps_u, const float* noise,
                                       float* pred_x0, const float* coef, int32_t S, float cfg_scale, int32_t use_scale, int64_t n,
                                       void* stream) {
    if (!state || !x || !eps_c || !noise || !coef || S <= 0 || n <= 0) return MOCA_E_BADARG;
    hipStream_t st = moca_stream(stream);
    hipLaunchKernelGGL(base_step_kernel, dim3(grid_for(n)), dim3(256), 0, st, state, x, eps_c, eps_u, noise, pred_x0, coef, S, cfg_scale,
                       use_scale, n);
    MOCA_CHECK_LAUNCH();
    hipLaunchKernelGGL(state_bump_kernel, dim3(1), dim3(1), 0, st, state);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_mask_frame_sums_f32(const float* mask, float* sums, int32_t frames, int32_t HW, void* stream) {
    if (!mask || !sums || frames <= 0 || HW <= 0) return MOCA_E_BADARG;
    hipLaunchKernelGGL(mask_frame_sums_kernel, dim3(frames), dim3(256), 0, moca_stream(stream), mask, sums, HW);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

// Issue rate of packed fp32 VALU against the scalar pair it replaces (gfx950): 8 independent chains per lane.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/pk_f32.hip -o /tmp/pk_f32 && /tmp/pk_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f32x2{(float)threadIdx.x + i, 1.0f + i};
    const f32x2 m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            else if (MODE == 1) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i][0]) : "v"(m[0]), "v"(c[0]));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i][1]) : "v"(m[1]), "v"(c[1]));
            } else if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            else if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            else if (MODE == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i][0]));
            else if (MODE == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i][0]));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
static void run(const char* name, int per_iter) {
    float* out;
    hipMalloc(&out, 1024 * 256 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 1; waves <= 2; ++waves) {          // blocks per CU: 1 wave / 2 waves per SIMD
        const int blocks = 256 * waves;
        k<MODE><<<blocks, 256>>>(out, 100);
        hipEventRecord(e0);
        k<MODE><<<blocks, 256>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: `waves` waves each issuing iters * 8 * per_iter instructions
        const double instr = (double)iters * 8 * per_iter * waves;
        printf("%-28s %d wave(s)/SIMD: %.2f ms -> %.2f ns per instruction per SIMD (%.1f cycles at 2.4 GHz)\n", name, waves, ms, ms * 1e6 / instr,
               ms * 1e6 / instr * 2.4);
    }
    hipFree(out);
}
int main() {
    run<0>("v_pk_fma_f32", 1);
    run<1>("2 x v_fma_f32", 2);
    run<2>("v_pk_mul_f32", 1);
    run<3>("v_pk_add_f32", 1);
    run<4>("v_exp_f32", 1);
    run<5>("v_rcp_f32", 1);
    return 0;
}

// Probe: buffer_load_dwordx4 ... lds (raw buffer, SGPR descriptor + 32-bit voffset + SGPR soffset).
// Questions: (1) does an out-of-range lane write ZEROS into LDS (or leave LDS untouched)?  (2) is soffset part of the
// range check?  (3) does the builtin compile for gfx950 with 16-byte pieces?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) char* lds_ptr;

__global__ void probe(const uint32_t* src, uint32_t nbytes, uint32_t* out, uint32_t soff) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) lds[i] = 0xDEADBEEF;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    // lanes 0..31 in range, lanes 32..47: voffset = 0x80000000 + lane*16 (out of range), lanes 48..63: just past the end
    uint32_t voff = lane * 16;
    if (lane >= 32 && lane < 48) voff = 0x80000000u + lane * 16;
    if (lane >= 48) voff = nbytes - soff + (lane - 48) * 16;      // in range w.r.t. voffset alone? beyond the end once soffset is added
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)lds, 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = lds[i];
}

int main() {
    const uint32_t N = 4096;   // bytes
    uint32_t* h = (uint32_t*)malloc(N + 4096);
    for (uint32_t i = 0; i < (N + 4096) / 4; ++i) h[i] = 0x1000 + i;
    uint32_t *d, *o;
    hipMalloc(&d, N + 4096); hipMalloc(&o, 1024);
    hipMemcpy(d, h, N + 4096, hipMemcpyHostToDevice);
    for (uint32_t soff : {0u, 64u}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, N, o, soff);
        uint32_t r[256];
        hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
        printf("soffset=%u\n", soff);
        for (int l : {0, 1, 31, 32, 40, 47, 48, 49, 50, 51, 52, 63}) printf("  lane %2d: %08x %08x %08x %08x\n", l, r[l * 4], r[l * 4 + 1], r[l * 4 + 2], r[l * 4 + 3]);
    }
    return 0;
}

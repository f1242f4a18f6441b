// Probe: semantics of buffer_load ... lds (LDS-DMA) on gfx950:
//  (a) destination = wave-uniform LDS base + lane*16 ?   (b) out-of-range lanes write zeros ?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) int i32x4;
__global__ void probe(const int* src, int nbytes, int* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* lds = (int*)smem;
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = -1;   // poison
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // lane l loads 16 B from a PERMUTED source chunk; odd lanes of wave 1 go out of range
    uint32_t voff = (uint32_t)((wave * 64 + (lane ^ 5)) * 16);
    if (wave == 1 && (lane & 1)) voff = 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += blockDim.x) out[i] = lds[i];
}
int main() {
    int h[512]; for (int i = 0; i < 512; ++i) h[i] = i + 1000;
    int *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(h)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    probe<<<1, 128, 8192>>>(d, sizeof(h), o);
    int r[512]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int ok_lin = 1, ok_zero = 1;
    for (int w = 0; w < 2; ++w) for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
        int got = r[(w * 64 + l) * 4 + e];
        bool oob = (w == 1 && (l & 1));
        int want = oob ? 0 : 1000 + (w * 64 + (l ^ 5)) * 4 + e;
        if (got != want) { if (oob) ok_zero = 0; else ok_lin = 0; if (l < 4) printf("w%d l%d e%d got %d want %d\n", w, l, e, got, want); }
    }
    printf("lane-linear destination with per-lane source: %s\nout-of-range lanes write zeros: %s\n", ok_lin ? "YES" : "NO", ok_zero ? "YES" : "NO");
    return 0;
}

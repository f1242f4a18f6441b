// Probe: is a packed-fp32 VALU instruction with a lane-crossing op_sel reliable on gfx950 while a wave of ANOTHER kernel shares
// the SIMD?  (the fused sums of two conv_gemm_nt instances went wrong beside a weight-gradient
// workgroup, and exactly in the lanes such instructions computed.)
//
// victim: every thread holds register pairs a, b and repeats   r = v_pk_add_f32(a, b)   in three forms -- natural lanes, the
//         crosswise op_sel:[0,1] op_sel_hi:[1,0] the compiler emitted, and op_sel on the first source -- and compares both
//         lanes of every result with scalar v_add_f32 of the same registers.  Operands change every iteration.
// co-runners on a second stream: an MFMA loop, a plain VALU loop, or none; small enough that both kernels share every SIMD.
//
//   hipcc --offload-arch=gfx950 -O2 pk_opsel_probe.hip -o pk_opsel_probe && ./pk_opsel_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

// counts[0..1]: natural form lo / hi lane mismatches; [2..3]: op_sel:[0,1] op_sel_hi:[1,0]; [4..5]: op_sel:[1,0] op_sel_hi:[0,1];
// [6]: wrong low lanes of the op_sel:[0,1] form that equal a.lo + b.LO (the op_sel bit of source 1 ignored);
// [7]: wrong low lanes of v_pk_mul_f32 op_sel:[0,1]; [8]: of those, equal to a.lo * b.LO;
// [9]: wrong low lanes of v_pk_fma_f32 op_sel:[0,1,0]; [10..11]: v_pk_mov_b32 op_sel:[0,1] lo / hi; [12..13]: v_pk_mov_b32 op_sel:[1,0] lo / hi
__global__ __launch_bounds__(256) void victim(const float* seed, int iters, unsigned long long* counts) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {seed[(t * 4 + 0) & 65535], seed[(t * 4 + 1) & 65535]};
    f32x2 b = {seed[(t * 4 + 2) & 65535], seed[(t * 4 + 3) & 65535]};
    unsigned bad[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        f32x2 rn, rx, ry;
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(rn) : "v"(a), "v"(b));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(rx) : "v"(a), "v"(b));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(ry) : "v"(a), "v"(b));
        float e0, e1, e2, e3;
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(e0) : "v"(a.x), "v"(b.x));     // lo + lo
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(e1) : "v"(a.y), "v"(b.y));     // hi + hi
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(e2) : "v"(a.x), "v"(b.y));     // lo + hi
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(e3) : "v"(a.y), "v"(b.x));     // hi + lo
        bad[0] += __float_as_uint(rn.x) != __float_as_uint(e0);
        bad[1] += __float_as_uint(rn.y) != __float_as_uint(e1);
        bad[2] += __float_as_uint(rx.x) != __float_as_uint(e2);      // lo lane: a.lo + b.HI
        bad[3] += __float_as_uint(rx.y) != __float_as_uint(e3);      // hi lane: a.hi + b.LO
        bad[6] += __float_as_uint(rx.x) != __float_as_uint(e2) && __float_as_uint(rx.x) == __float_as_uint(e0);
        {
            f32x2 rm;
            float m_hi, m_lo;
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(rm) : "v"(a), "v"(b));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m_hi) : "v"(a.x), "v"(b.y));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m_lo) : "v"(a.x), "v"(b.x));
            bad[7] += __float_as_uint(rm.x) != __float_as_uint(m_hi);
            bad[8] += __float_as_uint(rm.x) != __float_as_uint(m_hi) && __float_as_uint(rm.x) == __float_as_uint(m_lo);
        }
        {
            f32x2 rf, m1, m2;
            float f_lo;
            asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(rf) : "v"(a), "v"(b));
            asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(f_lo) : "v"(a.x), "v"(b.y));
            bad[9] += __float_as_uint(rf.x) != __float_as_uint(f_lo);
            asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(m1) : "v"(a), "v"(b));      // (a.lo, b.hi)
            asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(m2) : "v"(a), "v"(b));      // (a.hi, b.lo)
            bad[10] += __float_as_uint(m1.x) != __float_as_uint(a.x);
            bad[11] += __float_as_uint(m1.y) != __float_as_uint(b.y);
            bad[12] += __float_as_uint(m2.x) != __float_as_uint(a.y);
            bad[13] += __float_as_uint(m2.y) != __float_as_uint(b.x);
        }
        bad[4] += __float_as_uint(ry.x) != __float_as_uint(e3);      // lo lane: a.HI + b.lo
        bad[5] += __float_as_uint(ry.y) != __float_as_uint(e2);      // hi lane: a.LO + b.hi
        // new operands (bounded: no overflow to inf / NaN, which would compare unequal to nothing useful)
        a.x = a.x * 0.75f + 0.3f * b.y;
        a.y = a.y * 0.5f - 0.2f * b.x;
        b.x = b.x * 0.625f + 0.11f * (float)(i & 15);
        b.y = b.y * 0.875f - 0.07f * (float)(t & 7);
    }
    for (int k = 0; k < 14; ++k)
        if (bad[k]) atomicAdd(counts + k, (unsigned long long)bad[k]);
}

__global__ __launch_bounds__(256) void mfma_spin(int iters, float* sink) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.678f) sink[0] = c0[0];
}

__global__ __launch_bounds__(256) void valu_spin(int iters, float* sink) {
    float x = 0.001f * threadIdx.x, y = 1.0f;
    for (int i = 0; i < iters; ++i) {
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(y) : "v"(x));
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
        asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(y));
    }
    if (x + y == 12345.678f) sink[0] = x;
}

int main() {
    int ncu = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) ncu = prop.multiProcessorCount;
    std::vector<float> h(65536);
    uint32_t s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 2000 - 1000) * 0.001f; }
    float *seed, *sink;
    unsigned long long* counts;
    CHECK(hipMalloc(&seed, h.size() * 4));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&counts, 14 * 8));
    CHECK(hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipStream_t sa, sb;
    CHECK(hipStreamCreate(&sa));
    CHECK(hipStreamCreate(&sb));
    const int iters = 200000;                       // per thread; 4 waves x 2 blocks per CU x 256 CUs
    const char* names[3] = {"alone", "beside an MFMA loop", "beside a VALU loop"};
    for (int rep = 0; rep < 2; ++rep)
        for (int co = 0; co < 3; ++co) {
            CHECK(hipMemset(counts, 0, 112));
            CHECK(hipDeviceSynchronize());
            // co-runner first: 4 blocks of 4 waves per CU (one wave per SIMD each), long enough to outlast the victim
            if (co == 1) mfma_spin<<<dim3(ncu * 4), dim3(256), 0, sb>>>(4000000, sink);
            if (co == 2) valu_spin<<<dim3(ncu * 4), dim3(256), 0, sb>>>(6000000, sink);
            victim<<<dim3(ncu * 2), dim3(256), 0, sa>>>(seed, iters, counts);
            CHECK(hipStreamSynchronize(sa));
            const bool co_running = co == 0 || hipStreamQuery(sb) == hipErrorNotReady;     // still busy when the victim ended?
            CHECK(hipDeviceSynchronize());
            unsigned long long c[14];
            CHECK(hipMemcpy(c, counts, 112, hipMemcpyDeviceToHost));
            const double total = (double)iters * ncu * 2 * 256;
            printf("victim %-22s (co-runner outlasted it: %s): mismatching lanes of %.3g results each -- natural lo %llu hi %llu | "
                   "op_sel:[0,1] lo %llu (of which = a.lo + b.LO: %llu) hi %llu | op_sel:[1,0] lo %llu hi %llu | v_pk_mul_f32 op_sel:[0,1] lo %llu "
                   "(of which = a.lo * b.LO: %llu) | v_pk_fma_f32 op_sel:[0,1,0] lo %llu | v_pk_mov_b32 op_sel:[0,1] lo %llu hi %llu, op_sel:[1,0] lo %llu hi %llu\n",
                   names[co], co_running ? "yes" : "NO", total, c[0], c[1], c[2], c[6], c[3], c[4], c[5], c[7], c[8], c[9], c[10], c[11], c[12], c[13]);
        }
    return 0;
}

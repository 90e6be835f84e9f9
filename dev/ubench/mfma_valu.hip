// Micro-benchmark: can the matrix core and the vector ALU of a gfx950 SIMD work
// at the same time?  A wave runs dependent chains of four
// v_mfma_f32_32x32x16_f16 (as the codeword scan of decode.hip does) with NV
// independent v_min3_f32 after each matrix instruction; 1, 2 and 3 waves per
// SIMD.  Prints SIMD cycles per matrix instruction (32 = the matrix core's own
// time: 8 passes of 4 clocks).
// Build: hipcc --offload-arch=gfx950 -O3 dev/ubench/mfma_valu.hip -o dev/ubench/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ITER = 4096;

// DEP: the four matrix instructions of an iteration form a dependent chain
// (same accumulator); else they alternate between two accumulators
template <int NV, bool DEP>
__global__ void __launch_bounds__(256) k(float *out, float a, float b)
{
    f16x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (_Float16)(threadIdx.x * 0.01f + i); bv[i] = (_Float16)(i * 0.5f); }
    f32x16 acc0 = {0}, acc1 = {0};
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (DEP || (m & 1) == 0)
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(av), "v"(bv));
            else
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(av), "v"(bv));
#pragma unroll
            for (int v = 0; v < NV; ++v)
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(r[v % 8]) : "v"(a), "v"(b));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += r[i];
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    if (s == 12345.678f) out[0] = s;
}

template <int NV, bool DEP>
void run(int cus)
{
    float *out;
    CHECK(hipMalloc(&out, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps : {1, 2, 3}) {
        const int blocks = cus * wps;       // 256-thread blocks: one wave per SIMD each
        hipLaunchKernelGGL((k<NV, DEP>), dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<NV, DEP>), dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double mfma_per_simd = (double)ITER * 4 * wps;
        printf("%s chain, %d min3 per mfma, waves/SIMD %d: %.3f ms, %.1f SIMD cycles per mfma (at 2.4 GHz)\n",
               DEP ? "dependent  " : "two accums ", NV, wps, ms, ms * 1e-3 * 2.4e9 / mfma_per_simd);
    }
    CHECK(hipFree(out));
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
    run<0, true>(cus); run<2, true>(cus); run<4, true>(cus); run<6, true>(cus); run<8, true>(cus); run<12, true>(cus);
    run<0, false>(cus); run<4, false>(cus); run<6, false>(cus); run<8, false>(cus);
    return 0;
}

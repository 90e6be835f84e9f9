// Micro-benchmark: SIMD cycles per wave64 VALU instruction on gfx950, by
// instruction kind and waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3
// dev/ubench/valu_rate.hip -o dev/ubench/valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITER = 2048;
constexpr int CHAINS = 8;     // independent dependency chains per lane
constexpr int UNROLL = 4;     // per iteration: CHAINS * UNROLL instructions

template <int KIND>
__global__ void __launch_bounds__(256) k(float *out, float a, float b)
{
    float r[CHAINS];
    double d[CHAINS];
    unsigned u[CHAINS];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[CHAINS];
    for (int i = 0; i < CHAINS; ++i) {
        r[i] = threadIdx.x * 0.001f + i; d[i] = r[i]; u[i] = threadIdx.x + i; p[i] = (f2){r[i], r[i] + 1};
    }
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int uu = 0; uu < UNROLL; ++uu)
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"((f2){a, a}), "v"((f2){b, b}));
                if (KIND == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"((double)a), "v"((double)b));
                if (KIND == 3) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(0xfffffff0u), "v"(3u));
                if (KIND == 4) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
                if (KIND == 5) asm volatile("v_dot2_u32_u16 %0, %0, %1, %0" : "+v"(u[i]) : "v"(0x00010001u));
                if (KIND == 6) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(u[i]));
                if (KIND == 7) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"((double)a));
                if (KIND == 8) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(3u));
                if (KIND == 9) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[(i + 1) % CHAINS]));
                if (KIND == 10) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(0x12345678u), "v"(0x07060302u));
                if (KIND == 11) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(3u));
                if (KIND == 12) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(u[i]) : "v"(d[i]));
                if (KIND == 13) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"((double)a));
                // operand patterns as the decode kernel has them
                if (KIND == 14) asm volatile("v_and_or_b32 %0, %0, -16, 3" : "+v"(u[i]));
                if (KIND == 15) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 1) % CHAINS]), "v"(r[(i + 2) % CHAINS]));
                if (KIND == 16) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(u[i]) : "v"(u[(i + 1) % CHAINS]), "s"(0x00010001u));
                if (KIND == 17) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "s"(a), "v"(b));
                if (KIND == 18) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(r[i]) : "v"(a));
                if (KIND == 19) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]) : "v"(3u), "s"(0x5555555555555555ull));
                if (KIND == 20) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(u[i]) : "v"(a), "v"(b), "v"(3u) : "vcc");
                if (KIND == 21) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"((f2){a, a}));
                if (KIND == 22) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"((f2){a, a}));
                if (KIND == 23) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"((f2){a, a}));
                if (KIND == 24) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
                if (KIND == 25) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(0xffff0000u));
                if (KIND == 26) asm volatile("v_bfe_u32 %0, %0, 3, 1" : "+v"(u[i]));
                if (KIND == 27) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(r[i]));
                if (KIND == 28) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"((double)a));
                if (KIND == 29) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(d[(i + 1) % CHAINS]), "s"((double)a));
                if (KIND == 30) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(r[i]) : "v"(u[i]));
                if (KIND == 31) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 1) % CHAINS]));
                if (KIND == 32) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
                if (KIND == 33) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
            }
    }
    float s = 0;
    for (int i = 0; i < CHAINS; ++i) s += r[i] + (float)d[i] + (float)u[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s;
}

template <int KIND>
void run(const char *name, int cus)
{
    float *out;
    CHECK(hipMalloc(&out, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps : {2, 3, 8}) {
        // 256-thread blocks: one wave per SIMD each; wps blocks per CU
        const int blocks = cus * wps;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_wave = (double)ITER * CHAINS * UNROLL;
        // assume 2.4 GHz; cycles the SIMD spends per wave-instruction
        const double cyc = ms * 1e-3 * 2.4e9 / (instr_per_wave * wps);
        printf("%-16s waves/SIMD %d: %.3f ms, %.2f SIMD cycles per wave-instruction (at 2.4 GHz)\n", name, wps, ms, cyc);
    }
    CHECK(hipFree(out));
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
    run<0>("v_fma_f32", cus);
    run<1>("v_pk_fma_f32", cus);
    run<2>("v_fma_f64", cus);
    run<3>("v_and_or_b32", cus);
    run<4>("v_min3_f32", cus);
    run<5>("v_dot2_u32_u16", cus);
    run<6>("v_cvt_f64_u32", cus);
    run<7>("v_add_f64", cus);
    run<8>("v_add_u32", cus);
    run<9>("v_mov_b32", cus);
    run<10>("v_perm_b32", cus);
    run<11>("v_cndmask_b32", cus);
    run<12>("v_cvt_u32_f64", cus);
    run<13>("v_mul_f64", cus);
    run<14>("and_or v,-16,3", cus);
    run<15>("min3 3 vgpr", cus);
    run<16>("dot2 v,s,acc", cus);
    run<17>("fma_f32 v,s,v", cus);
    run<18>("fma_f32 v,v,1.0", cus);
    run<19>("cndmask_e64 sgpr", cus);
    run<20>("cmp+cndmask vcc", cus);
    run<21>("pk_fma 2 vgpr", cus);
    run<22>("pk_add_f32", cus);
    run<23>("pk_mul_f32", cus);
    run<24>("v_mul_f32", cus);
    run<25>("v_and_b32", cus);
    run<26>("v_bfe_u32 imm", cus);
    run<27>("v_cvt_f64_f32", cus);
    run<28>("fma_f64 2 vgpr", cus);
    run<29>("fma_f64 v,s,acc", cus);
    run<30>("v_cvt_f32_u32", cus);
    run<31>("permlane32_swap", cus);
    run<32>("v_fmac_f32", cus);
    run<33>("v_sub_f32", cus);
    return 0;
}

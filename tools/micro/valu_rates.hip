// Issue cost of the vector instructions the read kernel is made of, in cycles per wave64 instruction on one SIMD (4 waves per SIMD, independent chains):
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates tools/micro/valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kIter = 4096, kUnroll = 16;
template <int OP> __global__ void __launch_bounds__(1024) k(uint32_t *out, uint32_t seed) {
    uint32_t a = threadIdx.x ^ seed, b = a * 3u + 1u, c = a + 7u, d = a ^ 0x55u;
    uint64_t p = a, q = b;
    float f0 = (float)a, f1 = (float)b, f2 = 1.0001f, f3 = 0.9999f;
    double g0 = (double)a, g1 = 1.0000001;
    for (int i = 0; i < kIter; ++i) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            if constexpr (OP == 0) { asm volatile("v_xor_b32 %0, %0, %2\n v_xor_b32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(c)); }
            if constexpr (OP == 1) { asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1" : "+v"(p), "+v"(q) : "v"(c), "v"(d) : "vcc"); }
            if constexpr (OP == 2) { asm volatile("v_mul_lo_u32 %0, %0, %2\n v_mul_lo_u32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(c)); }
            if constexpr (OP == 3) { asm volatile("v_mul_hi_u32 %0, %0, %2\n v_mul_hi_u32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(c)); }
            if constexpr (OP == 4) { asm volatile("v_mul_u32_u24 %0, %0, %2\n v_mul_u32_u24 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(c)); }
            if constexpr (OP == 5) { asm volatile("v_mad_u32_u24 %0, %0, %2, %3\n v_mad_u32_u24 %1, %1, %2, %3" : "+v"(a), "+v"(b) : "v"(c), "v"(d)); }
            if constexpr (OP == 6) { asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %0" : "+v"(p), "+v"(q)); }
            if constexpr (OP == 7) { asm volatile("v_fma_f64 %0, %0, %1, %1\n v_mul_f64 %1, %1, %1" : "+v"(g0), "+v"(g1)); }
            if constexpr (OP == 8) { asm volatile("v_mul_f32 %0, %0, %2\n v_add_f32 %1, %1, %2" : "+v"(f0), "+v"(f1) : "v"(f2)); }
            if constexpr (OP == 9) { asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n v_cmp_lt_u32 vcc, %1, %2" : "+v"(a), "+v"(b) : "v"(c) : "vcc"); }
            if constexpr (OP == 10) { asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %0" : "+v"(p), "+v"(q)); }
            if constexpr (OP == 11) { asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %0" : "+v"(a), "+v"(b)); }
            if constexpr (OP == 12) { asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1" : "+v"(f0), "+v"(f1)); }
            if constexpr (OP == 13) { asm volatile("v_cvt_f32_u32 %0, %2\n v_cvt_u32_f32 %1, %3" : "+v"(f0), "+v"(a) : "v"(b), "v"(f1)); }
            if constexpr (OP == 14) { asm volatile("v_bitop3_b32 %0, %0, %2, %3 bitop3:0x36\n v_bitop3_b32 %1, %1, %2, %3 bitop3:0x36" : "+v"(a), "+v"(b) : "v"(c), "v"(d)); }
            if constexpr (OP == 15) { asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %0" : "+v"(p), "+v"(q)); }
            if constexpr (OP == 16) { asm volatile("v_mul_lo_u32 %0, %0, s0\n v_mul_lo_u32 %1, %1, s0" : "+v"(a), "+v"(b) : : "s0"); }
            if constexpr (OP == 17) { asm volatile("v_mad_u64_u32 %0, vcc, %2, s2, 0\n v_mad_u64_u32 %1, vcc, %3, s2, 0" : "+v"(p), "+v"(q) : "v"(c), "v"(d) : "vcc", "s2"); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + (uint32_t)p + (uint32_t)q + (uint32_t)f0 + (uint32_t)f1 + (uint32_t)g0 + (uint32_t)g1 + (uint32_t)f2 + (uint32_t)f3;
}
template <int OP> int run(const char *name, uint32_t *out, double ghz, int cus) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(cus), dim3(1024), 0, 0, out, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(cus), dim3(1024), 0, 0, out, 2u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: 4 waves x kIter x kUnroll x 2 instructions
    const double insts = 4.0 * kIter * kUnroll * 2.0, cycles = ms * 1e-3 * ghz * 1e9;
    printf("%-28s %8.3f ms  %6.2f cycles per instruction (at %.2f GHz)\n", name, ms, cycles / insts, ghz);
    return 0;
}
int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz\n", prop.name, cus, ghz);
    uint32_t *out;
    CHECK(hipMalloc(&out, (size_t)cus * 1024 * 4));
    run<0>("v_xor_b32", out, ghz, cus);
    run<11>("v_mov_b32", out, ghz, cus);
    run<9>("v_cndmask / v_cmp", out, ghz, cus);
    run<14>("v_bitop3_b32", out, ghz, cus);
    run<1>("v_mad_u64_u32 (vgpr addend)", out, ghz, cus);
    run<17>("v_mad_u64_u32 (sgpr, +0)", out, ghz, cus);
    run<2>("v_mul_lo_u32", out, ghz, cus);
    run<16>("v_mul_lo_u32 (sgpr)", out, ghz, cus);
    run<3>("v_mul_hi_u32", out, ghz, cus);
    run<4>("v_mul_u32_u24", out, ghz, cus);
    run<5>("v_mad_u32_u24", out, ghz, cus);
    run<10>("v_lshl_add_u64", out, ghz, cus);
    run<8>("v_mul_f32 / v_add_f32", out, ghz, cus);
    run<6>("v_pk_mul_f32", out, ghz, cus);
    run<15>("v_pk_add_f32", out, ghz, cus);
    run<7>("v_fma_f64 / v_mul_f64", out, ghz, cus);
    run<12>("v_rcp_f32", out, ghz, cus);
    run<13>("v_cvt_f32_u32 / v_cvt_u32_f32", out, ghz, cus);
    return 0;
}

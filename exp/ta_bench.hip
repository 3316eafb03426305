// Microbenchmark: cost of a wave-level global_load_dwordx4 as a function of the active lanes and of the address pattern.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
struct P { double x, y; };
template <int MODE>
__global__ void __launch_bounds__(768) k(const P *tab, uint32_t n_pairs, int active, int iters, double *out) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    double acc = 0.0;
    if ((int)lane < active) {
        for (int it = 0; it < iters; ++it) {
            P v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                h = h * 1664525u + 1013904223u;
                uint32_t idx;
                if (MODE == 0) idx = (h >> 8) % n_pairs;                        // random per lane
                else if (MODE == 1) idx = ((h >> 8) % (n_pairs / 32)) * 21 % n_pairs + j;   // random row start, sequential pairs
                else idx = (uint32_t)(it * 8 + j) % n_pairs;                     // same address for all lanes
                v[j] = tab[idx];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j].x * v[j].y;
        }
    }
    if (acc == 123.456) out[0] = acc;
}
int main() {
    const uint32_t n_pairs = 1u << 17;    // 2 MB table: L2 resident
    P *tab; double *out;
    hipMalloc(&tab, n_pairs * sizeof(P)); hipMalloc(&out, 8);
    std::vector<P> h(n_pairs, P{1.0, 2.0});
    hipMemcpy(tab, h.data(), n_pairs * sizeof(P), hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000, blocks = 256;
    for (int mode = 0; mode < 3; ++mode)
        for (int active : {64, 32, 16, 8, 4, 1}) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(768), 0, 0, tab, n_pairs, active, iters, out);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(768), 0, 0, tab, n_pairs, active, iters, out);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(768), 0, 0, tab, n_pairs, active, iters, out);
                hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
            }
            const double instr_per_cu = 12.0 * iters * 8;      // wave-level loads per CU (1 block of 12 waves per CU)
            printf("mode %d active %2d: %.3f ms  -> %.1f cycles per wave-load per CU (2.4 GHz)\n", mode, active, ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
        }
    return 0;
}

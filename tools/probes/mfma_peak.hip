// Sustained dense fp16 MFMA rate of the part: waves that do nothing but v_mfma_f32_32x32x16_f16 on registers.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(threadIdx.x * 0.001f + i);
        b[i] = (_Float16)(threadIdx.x * 0.002f - i);
    }
    f16v acc[NACC];
    for (int k = 0; k < NACC; ++k)
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < NACC; ++k)
        for (int i = 0; i < 16; ++i) s += acc[k][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int iters, const char *label) {
    float *out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = (double)blocks * 4 * iters * NACC * 2.0 * 32 * 32 * 16;
    printf("%s: %d blocks x 4 waves, %d MFMAs per wave: %.3f ms -> %.0f TFLOP/s\n", label, blocks, iters * NACC, best, flops / best / 1e9);
    hipFree(out);
}
int main() {
    run<4>(256 * 2, 20000, "2 waves/SIMD, 4 accumulators, ~40 ms");
    run<4>(256 * 2, 2000, "2 waves/SIMD, 4 accumulators, ~4 ms");
    run<4>(256 * 2, 200, "2 waves/SIMD, 4 accumulators, ~0.4 ms");
    run<4>(256 * 1, 2000, "1 wave/SIMD, 4 accumulators");
    run<2>(256 * 2, 4000, "2 waves/SIMD, 2 accumulators");
    run<1>(256 * 2, 8000, "2 waves/SIMD, 1 accumulator (dependent chain)");
    return 0;
}

// Probe: HBM write rate of the conv epilogue's store pattern (lane = pixel row, 16 B per lane, 1 KB row pitch)
// against a fully coalesced pattern, same bytes.  hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void epilogue_pattern(float *y, int M, int Cout) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const int ntiles = Cout / 128;
    const int mt = blockIdx.x / ntiles, nt = blockIdx.x % ntiles;
    const int m0 = mt * 128, n0 = nt * 128;
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (int mi = 0; mi < 2; ++mi) {
        const int m = m0 + (wm * 2 + mi) * 32 + (lane & 31);
        for (int ni = 0; ni < 2; ++ni)
            for (int rg = 0; rg < 4; ++rg) {
                const int n = n0 + (wn * 2 + ni) * 32 + rg * 8 + (lane >> 5) * 4;
                *reinterpret_cast<f32x4 *>(y + (size_t)m * Cout + n) = v;
            }
    }
}
// same tile, but 8 lanes cover one pixel's 128-byte (32-channel) line: one store instruction = 8 full lines
__global__ __launch_bounds__(256) void line_pattern(float *y, int M, int Cout) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const int ntiles = Cout / 128;
    const int mt = blockIdx.x / ntiles, nt = blockIdx.x % ntiles;
    const int m0 = mt * 128, n0 = nt * 128;
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (int mi = 0; mi < 2; ++mi)
        for (int ni = 0; ni < 2; ++ni)
            for (int g = 0; g < 4; ++g) {
                const int m = m0 + (wm * 2 + mi) * 32 + g * 8 + (lane >> 3);
                const int n = n0 + (wn * 2 + ni) * 32 + (lane & 7) * 4;
                *reinterpret_cast<f32x4 *>(y + (size_t)m * Cout + n) = v;
            }
}
__global__ __launch_bounds__(256) void linear_pattern(float *y, size_t n4) {
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) reinterpret_cast<f32x4 *>(y)[i] = v;
}
int main() {
    const int M = 32 * 120 * 160, Cout = 256;
    float *y;
    hipMalloc(&y, (size_t)M * Cout * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = (M / 128) * (Cout / 128);
    for (int k = 0; k < 3; ++k) {
        float ms;
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(epilogue_pattern, dim3(blocks), dim3(256), 0, 0, y, M, Cout);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("epilogue pattern: %.3f ms  %.2f TB/s\n", ms / 10, (double)M * Cout * 4 / (ms / 10) / 1e9);
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(line_pattern, dim3(blocks), dim3(256), 0, 0, y, M, Cout);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("line pattern:     %.3f ms  %.2f TB/s\n", ms / 10, (double)M * Cout * 4 / (ms / 10) / 1e9);
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(linear_pattern, dim3(4096), dim3(256), 0, 0, y, (size_t)M * Cout / 4);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("linear pattern:   %.3f ms  %.2f TB/s\n", ms / 10, (double)M * Cout * 4 / (ms / 10) / 1e9);
    }
    return 0;
}

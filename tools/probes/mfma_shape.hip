// Bare fp32 MFMA loops on random operands: does the 16x16x4 shape hold a higher clock than 32x32x2 (same FLOP per cycle)?
// MI355X_MICROARCH.md (DVFS give-back, item 7) reports 1.12-1.15x for the bf16 pair 16x16x32 vs 32x32x16.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip ; run: ./mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256) void loop(const float *in, float *out, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = in[(tid * 16 + i) & 0xfffff];
        b[i] = in[(tid * 16 + 8 + i) & 0xfffff];
    }
    if (SHAPE == 32) {
        f32x16 acc[4];
        for (int k = 0; k < 4; ++k)
            for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(j + k) & 7], b[j], acc[k], 0, 0, 0);
        }
        float s = 0.f;
        for (int k = 0; k < 4; ++k)
            for (int r = 0; r < 16; ++r) s += acc[k][r];
        out[tid] = s;
    } else {
        f32x4 acc[16];
        for (int k = 0; k < 16; ++k)
            for (int r = 0; r < 4; ++r) acc[k][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(j + k) & 7], b[(j + (k >> 2)) & 7], acc[k], 0, 0, 0);
        }
        float s = 0.f;
        for (int k = 0; k < 16; ++k)
            for (int r = 0; r < 4; ++r) s += acc[k][r];
        out[tid] = s;
    }
}

int main() {
    const int blocks = 256 * 3, iters = 20000;
    std::vector<float> h(1 << 20);
    for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    float *in, *out;
    hipMalloc(&in, h.size() * 4);
    hipMalloc(&out, blocks * 256 * 4);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep)
        for (int shape : {32, 16}) {
            // per iteration per wave: SHAPE 32: 32 MFMAs x 4096 flop; SHAPE 16: 64 MFMAs x 2048 flop  (same flop, same cycles)
            hipEventRecord(e0);
            for (int l = 0; l < 4; ++l) {
                if (shape == 32) hipLaunchKernelGGL(loop<32>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
                else hipLaunchKernelGGL(loop<16>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = 4.0 * blocks * 4.0 * iters * 32.0 * 4096.0;
            printf("shape %dx%d: %.2f ms  %.1f TFLOP/s\n", shape, shape, ms, flop / ms / 1e9);
        }
    return 0;
}

// Bare bf16 MFMA loops on random operands: 32x32x16 vs 16x16x32 (same FLOP per cycle) -- which clock does each hold?
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape_bf16 mfma_shape_bf16.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(256) void loop(const float *in, float *out, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 8; ++k) {
            a[i][k] = (__bf16)in[(tid * 64 + i * 8 + k) & 0xfffff];
            b[i][k] = (__bf16)in[(tid * 64 + 32 + i * 8 + k) & 0xfffff];
        }
    if (SHAPE == 32) {
        f32x16 acc[4];
        for (int k = 0; k < 4; ++k)
            for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(j + k) & 3], b[j], acc[k], 0, 0, 0);
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
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(j + k) & 3], b[(j + (k >> 2)) & 3], acc[k], 0, 0, 0);
        }
        float s = 0.f;
        for (int k = 0; k < 16; ++k)
            for (int r = 0; r < 4; ++r) s += acc[k][r];
        out[tid] = s;
    }
}

int main() {
    const int blocks = 256 * 2, iters = 40000;
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
            // per iteration per wave: SHAPE 32: 16 MFMAs x 32768 flop; SHAPE 16: 32 MFMAs x 16384 flop
            hipEventRecord(e0);
            for (int l = 0; l < 4; ++l) {
                if (shape == 32) hipLaunchKernelGGL(loop<32>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
                else hipLaunchKernelGGL(loop<16>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = 4.0 * blocks * 4.0 * iters * 16.0 * 32768.0;
            printf("bf16 shape %d: %.2f ms  %.1f TFLOP/s\n", shape, ms, flop / ms / 1e9);
        }
    return 0;
}

// Shared device/host helpers for the gfx950 kernels of the PlaneRCNN detection path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define A3D_OK 0
#define A3D_ERR_ARG (-1)
#define A3D_ERR_LAUNCH (-2)
#define A3D_ERR_UNSUPPORTED (-3)

#define A3D_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// hipGetLastError() is per-thread and also reports non-sticky codes left behind by OTHER users of the
// runtime in this thread (e.g. hipErrorNotReady from an event query of the host framework).  Every entry
// point therefore clears the slot before its first launch (a3d_begin) and reads it after its last.
static inline void a3d_begin() { (void)hipGetLastError(); }
static inline int a3d_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? A3D_OK : A3D_ERR_LAUNCH;
}

// Bijective XCD-aware remap of a 1-D block id: blocks b and b+8 share an XCD (observed round-robin
// placement, used for speed only); give every XCD a contiguous chunk of logical tiles so tiles that
// share an operand panel hit the same L2.
__device__ __forceinline__ int a3d_xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// Shared device/host helpers for the gfx950 kernels of the PlaneRCNN detection path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define A3D_OK 0
#define A3D_ERR_ARG (-1)
#define A3D_ERR_LAUNCH (-2)
#define A3D_ERR_UNSUPPORTED (-3)

#define A3D_WAVE 64

// The library reads NO environment variable and keeps no process-global switch (include/a3d.h: re-entrant, no global state): every A/B
// choice that survives in the tree is either a descriptor field or a compile-time constant.  Developer builds (-DA3D_ABLATIONS, used by
// the tools/*_abl.sh and A/B scripts only) may override such a constant from the environment, read at every call.
#ifdef A3D_ABLATIONS
#include <stdlib.h>
static inline long a3d_dev_knob(const char *name, long dflt) {
    const char *e = getenv(name);
    return e ? atol(e) : dflt;
}
#else
#define a3d_dev_knob(name, dflt) (dflt)
#endif

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

// Magnitudes that count towards an image's maximum: FINITE ones.  An Inf / NaN (or a value at the very top of fp32's range) must not
// set the image's power-of-two scale -- an infinite maximum would leave the image unscaled and a NaN is dropped by fmaxf anyway --
// so the finite values of such an image keep their full 22 bits, and the non-finite value itself splits into (Inf | NaN, NaN) and
// poisons exactly the outputs whose receptive field holds it, as it does in fp32 (tests/test_gpu_precision.py).
__device__ __forceinline__ float a3d_finite_mag(const float v) {
    const float a = fabsf(v);
    return a < 1.7e38f ? a : 0.f;
}
__device__ __forceinline__ float a3d_absmax4(const f32x4 v) {
    float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));  // (fmaxf drops NaNs; an Inf wins)
    if (__builtin_expect(!(m < 1.7e38f), 0))  // rare: re-take the maximum over the finite members only
        m = fmaxf(fmaxf(a3d_finite_mag(v[0]), a3d_finite_mag(v[1])), fmaxf(a3d_finite_mag(v[2]), a3d_finite_mag(v[3])));
    return m;
}

// hipFuncSetAttribute (the > 64 KiB dynamic-LDS opt-in) is per DEVICE: a launcher keeps one bit per device ordinal, so a second
// device used by the same process gets its own opt-in (a process-wide flag made every launch there fail).
struct a3d_attr_once {
    unsigned long long done = 0;
    bool needed() const {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return true;
        return dev >= 64 || !((done >> dev) & 1ull);
    }
    void mark() {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev < 64) done |= 1ull << dev;  // (benign race: the attribute call is idempotent)
    }
};

// Bijective XCD-aware remap of a 1-D block id: blocks b and b+8 share an XCD (observed round-robin
// placement, used for speed only); give every XCD a contiguous chunk of logical tiles so tiles that
// share an operand panel hit the same L2.
__device__ __forceinline__ int a3d_xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

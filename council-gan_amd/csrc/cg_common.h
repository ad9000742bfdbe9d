// Shared helpers for the Council-GAN gfx950 kernels.  CDNA4 only: 64-lane wavefronts, no
// portability macros.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/council_gan_hip.h"

extern thread_local char cg_err_buf[512];
int cg_set_error(int code, const char* fmt, ...);

#define CG_CHECK_ARG(cond, ...)                         \
    do {                                                \
        if (!(cond)) return cg_set_error(CG_ERR_ARG, __VA_ARGS__); \
    } while (0)

#define CG_LAUNCH_CHECK(name)                                                                   \
    do {                                                                                        \
        hipError_t e__ = hipGetLastError();                                                     \
        if (e__ != hipSuccess) return cg_set_error(CG_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

static inline hipStream_t cg_s(cg_stream_t s) { return (hipStream_t)s; }

__device__ __forceinline__ float cg_apply_act(float v, int act) {
    switch (act) {
        case CG_ACT_RELU: return v > 0.f ? v : 0.f;
        case CG_ACT_LRELU: return v > 0.f ? v : 0.2f * v;
        case CG_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// derivative of the activation expressed through its OUTPUT y
__device__ __forceinline__ float cg_act_grad_from_out(float y, int act) {
    switch (act) {
        case CG_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case CG_ACT_LRELU: return y > 0.f ? 1.f : 0.2f;
        case CG_ACT_TANH: return 1.f - y * y;
        default: return 1.f;
    }
}

// 64-lane wavefront reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline unsigned cg_div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

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

// An unconditional use of a loaded per-lane value right behind its load: retires hipcc's "pending load" state on every path, so
// that the exec-masked per-value blocks of an epilogue do not each start with s_waitcnt vmcnt(0) -- which on gfx950 also waits for
// the previous block's STORE (stores count in vmcnt).  See conv_x3.inc, "Epilogue hygiene".
__device__ __forceinline__ void cg_touch(float v) { asm volatile("" ::"v"(v)); }

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
// ---- layout of the {hi, lo} fp16 form of a tensor (split-precision path) ------------------------------------------
// CG_X3_INTERLEAVE = 1 (default): the planes are interleaved per 32 elements -- element i (flat physical index, NHWC /
// [O][KH][KW][I]; channel counts are multiples of 32) has its hi half at cg_il(i) = 64*(i/32) + i%32 and its lo half
// CG_X3_LO_ELEMS = 32 halves further, so the 32 channels x {hi, lo} of one K-slice of a pixel / weight row are ONE
// 128-byte line (the planar layout fetched two half lines: +4...19 % on the forward / data-gradient kernels).
// CG_X3_INTERLEAVE = 0: two separate planes, lo plane `lo_elems` halves after the hi plane (A/B builds).
#ifndef CG_X3_INTERLEAVE
#define CG_X3_INTERLEAVE 1
#endif
__host__ __device__ __forceinline__ size_t cg_il(size_t i) {
#if CG_X3_INTERLEAVE
    return ((i >> 5) << 6) | (i & 31);
#else
    return i;
#endif
}
// byte offset of channel c (relative to its pixel / row, c a multiple of 8) and shift of a pixel / row base (elements -> bytes)
__host__ __device__ __forceinline__ unsigned cg_il_cbytes(unsigned c) {
#if CG_X3_INTERLEAVE
    return ((c & ~31u) << 2) + ((c & 31u) << 1);
#else
    return c << 1;
#endif
}
constexpr int CG_IL_SHIFT = CG_X3_INTERLEAVE ? 2 : 1;



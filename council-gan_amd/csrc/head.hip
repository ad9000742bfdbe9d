// Fused decoder head for the tape-free generator passes (the two discriminator-side updates decode without an autograd tape):
//   /root/reference/networks.py:393-395  three 1x1 Conv2dBlocks  64 -> 64 (ReLU) -> 64 (ReLU) -> 12 (tanh)
//   /root/reference/networks.py:398-407  mask / blend head: mask_j = (tanh(10 x[9+j]) + 1) / 2,  im <- (1 - m_j) im + m_j x[3j..3j+2]
// as ONE kernel.  Layer by layer this chain is pure HBM traffic at full resolution (a 64-channel fp32 tensor written and a
// {hi, lo} copy written and read per layer: 44 bytes per element and pixel all told, 3.4-4.5 TB/s); fused, a pixel's 64
// channels are read once (256 bytes, the {hi, lo} planes the last AdaIN apply wrote) and 24 bytes leave (image + mask).
//
// Split-precision arithmetic exactly as conv_fwd_x3_kernel (conv_x3.inc): v_mfma_f32_32x32x16_f16, a*b ~= ah*bh + ah*bl +
// al*bh, fp32 accumulation, per k-step the small terms first.  The GEMM is evaluated TRANSPOSED -- out^T[cout][pixel] =
// W[cout][k] * x^T[k][pixel]: weights are the MFMA's A operand, the 32 pixels of a wave's tile its B operand -- because
// then the accumulator layout (lane = pixel, 16 output channels per lane) is almost the next layer's B-operand layout (lane
// = pixel, 8 consecutive input channels per lane): the two lanes of a pixel (lane, lane + 32) swap half of their packed
// halves with one ds_bpermute per dword and the layer's output never leaves the registers -- no LDS round trip, no
// barrier inside the chain.  Weights (40 KiB of fragments for the three layers) sit in LDS in fragment order, loaded once
// per block; a wave walks a strided list of 32-pixel tiles.
//
// Needs the interleaved {hi, lo} layout, 64 trunk channels, output_dim 3, three masks (every shipped config).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/council_gan_hip.h"
#include "cg_common.h"

#if CG_X3_INTERLEAVE
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int HC = 64;            // trunk channels
constexpr int OD = 3, NK = 3;     // image channels, masks
constexpr int C9 = OD * NK + NK;  // 12 channels out of the last convolution
// fragment table in LDS: [layer 7: 2 cout tiles][4 k-steps][hi, lo] = 16, layer 8: 16, layer 9: 1 x 4 x 2 = 8
constexpr int NFRAG = 40;

__device__ __forceinline__ f16x8 as_h8(float4 v) { return __builtin_bit_cast(f16x8, v); }

struct HeadArgs {
    const void* xs;               // {hi, lo} planes of the trunk output, [pixels][64], interleaved per 32
    const void* w[3];             // split weights of the three layers (lead member), [cout][64] interleaved
    const float* b[3];            // biases (lead member)
    const float* w_scale_dev;     // power-of-two scale of the weight mirror (device), may be null
    float w_scale;                // static part of the weight scale
    const float* im_in;           // [pixels][3]
    float* im_out;                // [pixels][3]
    float* mask;                  // [pixels][3]
    long long npix_member;        // pixels per member
    long long w_stride_bytes;     // between members: split weights
    long long b_stride_bytes;     // between members: biases
};

// the 16 accumulator values of one 32-channel tile (this lane: pixel l31, channels 8g + 4lh + 0..3, g = 0..3) after bias +
// activation, as {hi, lo} halves packed per quad: q[g] = 4 halves of quad g
struct Quads {
    f16x4 h[4], l[4];
};

template <int ACT>
__device__ __forceinline__ Quads epilogue(const f32x16& acc, float scale, const float* __restrict__ bias, int cbase, int lh,
                                          int cmax) {
    Quads q;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = cbase + 8 * g + 4 * lh + i;
            float v = acc[4 * g + i] * scale + (c < cmax ? bias[c] : 0.f);
            v = ACT == CG_ACT_RELU ? (v > 0.f ? v : 0.f) : v;
            const _Float16 h = (_Float16)fminf(fmaxf(v, -65504.f), 65504.f);
            q.h[g][i] = h;
            q.l[g][i] = (_Float16)(v - (float)h);
        }
    return q;
}

__device__ __forceinline__ f16x4 swap32(f16x4 v) {      // the same quad of the partner lane (lane ^ 32)
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    i32x2 w = __builtin_bit_cast(i32x2, v);
    w[0] = __shfl_xor(w[0], 32, 64);
    w[1] = __shfl_xor(w[1], 32, 64);
    return __builtin_bit_cast(f16x4, w);
}

// B-operand fragments (this lane: pixel l31, k = 16u + 8lh + 0..7 of the 32-channel tile) for k-steps u = 0, 1 from the
// tile's quads: lane lh = 0 keeps its quads 2u and takes the partner's quads 2u (channels +4..7); lane lh = 1 takes the
// partner's quads 2u + 1 (channels 16u + 8..11) and keeps its own quads 2u + 1 (channels 16u + 12..15)
__device__ __forceinline__ void next_frags(const Quads& q, int lh, f16x8 (&fh)[2], f16x8 (&fl)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const f16x4 send_h = lh ? q.h[2 * u] : q.h[2 * u + 1], send_l = lh ? q.l[2 * u] : q.l[2 * u + 1];
        const f16x4 recv_h = swap32(send_h), recv_l = swap32(send_l);
        const f16x4 own_h = lh ? q.h[2 * u + 1] : q.h[2 * u], own_l = lh ? q.l[2 * u + 1] : q.l[2 * u];
        const f16x4 lo4_h = lh ? recv_h : own_h, hi4_h = lh ? own_h : recv_h;     // k 0..3, k 4..7
        const f16x4 lo4_l = lh ? recv_l : own_l, hi4_l = lh ? own_l : recv_l;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fh[u][i] = lo4_h[i];
            fh[u][4 + i] = hi4_h[i];
            fl[u][i] = lo4_l[i];
            fl[u][4 + i] = hi4_l[i];
        }
    }
}

__device__ __forceinline__ f32x16 mma3(f32x16 acc, f16x8 wh, f16x8 wl, f16x8 xh, f16x8 xl) {
    // the order of conv_fwd_x3_kernel's k-step: x_lo * w_hi, x_hi * w_lo, then the dominant x_hi * w_hi
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc, 0, 0, 0);
    return acc;
}

__global__ __launch_bounds__(256) void head_fwd_x3_kernel(HeadArgs a) {
    __shared__ __attribute__((aligned(16))) float4 wf[NFRAG][64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int z = blockIdx.z;

    // ---- weights -> LDS in fragment order.  fragment f of layer L: (cout tile j, k-step ks, plane); lane l holds the 8 halves
    // of row cout = 32j + l%32 at k = 16ks + 8(l/32): flat index cout*64 + k of the interleaved [cout][64] tensor
    for (int f = wid; f < NFRAG; f += 4) {
        const int L = f < 16 ? 0 : (f < 32 ? 1 : 2);
        const int r = f - (L == 0 ? 0 : (L == 1 ? 16 : 32));
        const int j = L < 2 ? (r >> 3) : 0, ks = (r >> 1) & 3, plane = r & 1;
        const int cout = 32 * j + l31, k0 = 16 * ks + 8 * lh;
        const int cmax = L < 2 ? HC : C9;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cout < cmax) {
            const size_t flat = (size_t)cout * HC + k0;
            const _Float16* base = reinterpret_cast<const _Float16*>(reinterpret_cast<const char*>(a.w[L]) + (long long)z * a.w_stride_bytes);
            v = *reinterpret_cast<const float4*>(base + cg_il(flat) + (plane ? CG_X3_LO_ELEMS : 0));
        }
        wf[f][lane] = v;
    }
    __syncthreads();
    const float* b7 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.b[0]) + (long long)z * a.b_stride_bytes);
    const float* b8 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.b[1]) + (long long)z * a.b_stride_bytes);
    const float* b9 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.b[2]) + (long long)z * a.b_stride_bytes);
    float wscale = 1.f / a.w_scale;
    if (a.w_scale_dev) wscale *= 1.f / a.w_scale_dev[0];

    const long long npm = a.npix_member;
    const long long ntiles = (npm + 31) / 32;
    const _Float16* xs = reinterpret_cast<const _Float16*>(a.xs);
    for (long long t = (long long)blockIdx.x * 4 + wid; t < ntiles; t += (long long)gridDim.x * 4) {
        // the weight fragments are re-read from LDS for every tile: hoisted out of this loop they would occupy 160 VGPRs and
        // leave ONE wave per SIMD with nothing to hide the pixel loads behind (measured: 400 us for a 268 MB launch); the
        // reads cost 320 LDS cycles per tile
        asm volatile("" ::: "memory");
        const long long pm = t * 32 + l31;                         // pixel inside the member
        const bool live = pm < npm;
        const long long p = (long long)z * npm + (live ? pm : npm - 1);        // clamp: dead lanes read a valid pixel
        // ---- this pixel's 64 channels as B-operand fragments: k-step ks, k = 16ks + 8lh .. +7
        f16x8 xh[4], xl[4];
        const _Float16* px = xs + (size_t)p * (2 * HC);            // 64 hi + 64 lo halves per pixel, interleaved per 32
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int c0 = 16 * ks + 8 * lh;
            const _Float16* q = px + ((c0 >> 5) << 6) + (c0 & 31);
            xh[ks] = as_h8(*reinterpret_cast<const float4*>(q));
            xl[ks] = as_h8(*reinterpret_cast<const float4*>(q + CG_X3_LO_ELEMS));
        }
        // ---- layers 7 and 8: 64 -> 64, ReLU
#pragma unroll
        for (int L = 0; L < 2; ++L) {
            f32x16 acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int f = 16 * L + 8 * j + 2 * ks;
                    acc[j] = mma3(acc[j], as_h8(wf[f][lane]), as_h8(wf[f + 1][lane]), xh[ks], xl[ks]);
                }
            }
            const float* bias = L == 0 ? b7 : b8;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const Quads q = epilogue<CG_ACT_RELU>(acc[j], wscale, bias, 32 * j, lh, HC);
                f16x8 fh[2], fl[2];
                next_frags(q, lh, fh, fl);
                xh[2 * j] = fh[0];
                xh[2 * j + 1] = fh[1];
                xl[2 * j] = fl[0];
                xl[2 * j + 1] = fl[1];
            }
        }
        // ---- layer 9: 64 -> 12, tanh; then the mask / blend head
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = mma3(acc, as_h8(wf[32 + 2 * ks][lane]), as_h8(wf[33 + 2 * ks][lane]), xh[ks], xl[ks]);
        // this lane: channels 4lh + 0..3 (g = 0) and 8 + 4lh + 0..3 (g = 1); 12 channels = lane0 {0-3, 8-11} + lane1 {4-7}
        float v0[4], v1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c0 = 4 * lh + i, c1 = 8 + 4 * lh + i;
            v0[i] = tanhf(acc[i] * wscale + b9[c0]);                 // c0 < 8 <= 12 always valid
            v1[i] = c1 < C9 ? tanhf(acc[4 + i] * wscale + b9[c1]) : 0.f;
        }
        float nx[C9];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float other = __shfl_xor(v0[i], 32, 64);            // the partner's channels 4(1-lh) + i
            nx[i] = lh ? other : v0[i];
            nx[4 + i] = lh ? v0[i] : other;
            nx[8 + i] = v1[i];                                         // (only lane 0 of a pixel assembles all 12 channels)
        }
        if (lh == 0 && live) {
            float im[OD];
#pragma unroll
            for (int c = 0; c < OD; ++c) im[c] = a.im_in[p * OD + c];
#pragma unroll
            for (int j = 0; j < NK; ++j) {
                const float m = (tanhf(10.f * nx[OD * NK + j]) + 1.f) * 0.5f;
                a.mask[p * NK + j] = m;
#pragma unroll
                for (int c = 0; c < OD; ++c) im[c] = (1.f - m) * im[c] + m * nx[OD * j + c];
            }
#pragma unroll
            for (int c = 0; c < OD; ++c) a.im_out[p * OD + c] = im[c];
        }
    }
}

}  // namespace
#endif  // CG_X3_INTERLEAVE

extern "C" int cg_decoder_head_fwd_x3(const void* xs, size_t x_lo_elems, const void* w7s, const void* w8s, const void* w9s,
                                      size_t w_lo_elems, float w_scale, const float* w_scale_dev, const float* b7,
                                      const float* b8, const float* b9, const cg_group* group, const float* im_in,
                                      float* im_out, float* mask, long long npix, int channels, int out_dim, int nmask,
                                      cg_stream_t stream) {
#if CG_X3_INTERLEAVE
    CG_CHECK_ARG(xs && w7s && w8s && w9s && b7 && b8 && b9 && im_in && im_out && mask && npix > 0 && w_scale > 0.f,
                 "cg_decoder_head_fwd_x3: null pointer / bad size");
    CG_CHECK_ARG(channels == HC && out_dim == OD && nmask == NK,
                 "cg_decoder_head_fwd_x3: built for %d trunk channels, %d image channels, %d masks (got %d, %d, %d)", HC, OD, NK,
                 channels, out_dim, nmask);
    CG_CHECK_ARG(x_lo_elems == CG_X3_LO_ELEMS && w_lo_elems == CG_X3_LO_ELEMS, "cg_decoder_head_fwd_x3: needs the interleaved layout");
    const int n = group ? group->n : 1;
    CG_CHECK_ARG(n >= 1 && n <= 64 && npix % n == 0 && (n == 1 || (group->stride > 0 && group->stride % 32 == 0)),
                 "cg_decoder_head_fwd_x3: bad member group");
    CG_CHECK_ARG((double)npix * 2 * HC * 2 < 9.0e18, "cg_decoder_head_fwd_x3: too large");
    HeadArgs a;
    a.xs = xs;
    a.w[0] = w7s; a.w[1] = w8s; a.w[2] = w9s;
    a.b[0] = b7; a.b[1] = b8; a.b[2] = b9;
    a.w_scale_dev = w_scale_dev;
    a.w_scale = w_scale;
    a.im_in = im_in; a.im_out = im_out; a.mask = mask;
    a.npix_member = npix / n;
    a.w_stride_bytes = n > 1 ? (long long)group->stride * 4 : 0;      // interleaved {hi, lo}: 4 bytes per element
    a.b_stride_bytes = n > 1 ? (long long)group->stride * 4 : 0;
    const long long tiles = (a.npix_member + 31) / 32;
    long long blocks = (tiles + 3) / 4;                                // 4 waves per block, one tile each per trip
    const long long cap = (2048 + n - 1) / n;                          // ~8 blocks per CU over all members
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(head_fwd_x3_kernel, dim3((unsigned)blocks, 1, n), dim3(256), 0, cg_s(stream), a);
    CG_LAUNCH_CHECK("head_fwd_x3_kernel");
    return CG_OK;
#else
    (void)xs; (void)x_lo_elems; (void)w7s; (void)w8s; (void)w9s; (void)w_lo_elems; (void)w_scale; (void)w_scale_dev; (void)b7;
    (void)b8; (void)b9; (void)group; (void)im_in; (void)im_out; (void)mask; (void)npix; (void)channels; (void)out_dim; (void)nmask;
    (void)stream;
    return cg_set_error(CG_ERR_ARG, "cg_decoder_head_fwd_x3: needs the interleaved operand layout");
#endif
}

// InstanceNorm / AdaIN / LayerNorm for NHWC fp32 activations on gfx950.
//
// Reference call sites: nn.InstanceNorm2d (networks.py:483), AdaptiveInstanceNorm2d.forward =
// F.batch_norm on the (1, B*C, H, W) view (networks.py:640-653), LayerNorm.forward
// (networks.py:670-686).  All of these are HBM-bound passes; statistics are accumulated in
// fp64 (the vector fp64 rate is far above what HBM can feed) so that E[x^2]-E[x]^2 cannot
// cancel, then applied in fp32.  Channels are the contiguous dimension, so a wavefront reads
// 64 consecutive channels of one pixel (coalesced 256 B) and reductions over H*W run down the
// rows; H*W is split across blocks to fill the 256 CUs and combined in a fixed order
// (deterministic, no atomics).
#include "cg_common.h"

namespace {

constexpr int NC_SPLIT_ROWS = 64;  // rows (pixels) per block in the per-(n,c) reductions

__host__ __device__ inline int nc_splits(int HW) { return (HW + NC_SPLIT_ROWS - 1) / NC_SPLIT_ROWS; }

// ws[((n*C + c) * S + s) * 2 + {0,1}] = {sum x, sum x^2} over the rows of split s
__global__ __launch_bounds__(256) void in_stats_partial(const float* __restrict__ x, double* __restrict__ ws, int HW,
                                                        int C, int S) {
    __shared__ double red[2][4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl, n = blockIdx.y, s = blockIdx.z;
    const int r0 = s * NC_SPLIT_ROWS, r1 = min(r0 + NC_SPLIT_ROWS, HW);
    double a = 0.0, b = 0.0;
    if (c < C) {
        const float* p = x + (size_t)n * HW * C + c;
        for (int r = r0 + rl; r < r1; r += 4) {
            double v = (double)p[(size_t)r * C];
            a += v;
            b += v * v;
        }
    }
    red[0][rl][cl] = a;
    red[1][rl][cl] = b;
    __syncthreads();
    if (rl == 0 && c < C) {
        double* o = ws + ((size_t)(n * C + c) * S + s) * 2;
        o[0] = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
        o[1] = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
    }
}

// one wavefront per (n, c): lanes stride over the S split partials, fixed-order butterfly combine
__global__ __launch_bounds__(256) void in_stats_final(const double* __restrict__ ws, float* __restrict__ mean,
                                                      float* __restrict__ rstd, int NC, int S, int HW, float eps) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= NC) return;
    const int lane = threadIdx.x & 63;
    double a = 0.0, b = 0.0;
    for (int s = lane; s < S; s += 64) {
        a += ws[((size_t)i * S + s) * 2];
        b += ws[((size_t)i * S + s) * 2 + 1];
    }
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    if (lane == 0) {
        double m = a / HW;
        double var = b / HW - m * m;  // biased variance (batch_norm training / InstanceNorm2d)
        if (var < 0.0) var = 0.0;
        mean[i] = (float)m;
        rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// statistics from the partials a convolution epilogue emitted (cg_conv2d_fwd_stats): part[((n*S + s)*C + c)*2 + {0,1}]
// = {sum y, sum y^2} over rows [s*R, (s+1)*R) of sample n.  A block owns 32 consecutive channels of one sample: thread
// (c = tid % 32, g = tid / 32) adds the partials s = g, g + 8, ... (32 channels x 16 bytes = one 512-byte run per s: coalesced,
// S / 8 independent loads per thread), the eight groups are joined in a fixed order.  (One wavefront per (n, c) with its lanes
// striding over s touched a different 128-byte line per lane: 11 us per call, 54 calls per benchmark step.)
__global__ __launch_bounds__(256) void in_stats_final_tiles(const double* __restrict__ part, float* __restrict__ mean,
                                                            float* __restrict__ rstd, int C, int S, int HW, float eps) {
    const int n = blockIdx.y, c = blockIdx.x * 32 + (threadIdx.x & 31), g = threadIdx.x >> 5;
    double a = 0.0, b = 0.0;
    if (c < C) {
        for (int s = g; s < S; s += 8) {
            const double2 p = *reinterpret_cast<const double2*>(part + ((size_t)(n * S + s) * C + c) * 2);
            a += p.x;
            b += p.y;
        }
    }
    __shared__ double red[8][32][2];
    red[g][threadIdx.x & 31][0] = a;
    red[g][threadIdx.x & 31][1] = b;
    __syncthreads();
    if (g == 0 && c < C) {
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            a += red[k][threadIdx.x][0];
            b += red[k][threadIdx.x][1];
        }
        const double m = a / HW;
        double var = b / HW - m * m;
        if (var < 0.0) var = 0.0;
        mean[(size_t)n * C + c] = (float)m;
        rstd[(size_t)n * C + c] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// y = act((x - mean) * rstd * gamma + beta) + residual
template <bool VEC>
__global__ __launch_bounds__(256) void in_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ residual, float* __restrict__ y,
                                                       size_t total, int HW, int C, int act, int gs) {
    constexpr int V = VEC ? 4 : 1;
    const size_t nvec = total / V;
    const size_t per_n = (size_t)HW * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * V;
        const int n = (int)(e / per_n);
        const int c = (int)(e % C);
        float xv[V], rv[V], ov[V];
        if constexpr (VEC) {
            float4 t = *reinterpret_cast<const float4*>(x + e);
            xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
            if (residual) {
                float4 r = *reinterpret_cast<const float4*>(residual + e);
                rv[0] = r.x; rv[1] = r.y; rv[2] = r.z; rv[3] = r.w;
            }
        } else {
            xv[0] = x[e];
            if (residual) rv[0] = residual[e];
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int s = n * C + c + k;
            float z = (xv[k] - mean[s]) * rstd[s];
            if (gamma) z = z * gamma[n * gs + c + k] + beta[n * gs + c + k];
            z = cg_apply_act(z, act);
            if (residual) z += rv[k];
            ov[k] = z;
        }
        if constexpr (VEC)
            *reinterpret_cast<float4*>(y + e) = make_float4(ov[0], ov[1], ov[2], ov[3]);
        else
            y[e] = ov[0];
    }
}

__device__ __forceinline__ float in_dz(float dy, float xhat, float g, float b, int act) {
    if (act == CG_ACT_NONE) return dy;
    float z = xhat * g + b;
    if (act == CG_ACT_RELU) return z > 0.f ? dy : 0.f;
    if (act == CG_ACT_LRELU) return z > 0.f ? dy : 0.2f * dy;
    float t = tanhf(z);
    return dy * (1.f - t * t);
}

// ws[((n*C+c)*S + s)*2 + {0,1}] = {sum dz, sum dz*xhat}
__global__ __launch_bounds__(256) void in_bwd_partial(const float* __restrict__ dy, const float* __restrict__ x,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      double* __restrict__ ws, int HW, int C, int S, int act, int gs) {
    __shared__ double red[2][4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl, n = blockIdx.y, s = blockIdx.z;
    const int r0 = s * NC_SPLIT_ROWS, r1 = min(r0 + NC_SPLIT_ROWS, HW);
    double a = 0.0, b = 0.0;
    if (c < C) {
        const int sc = n * C + c;
        const float m = mean[sc], rs = rstd[sc];
        const float g = gamma ? gamma[n * gs + c] : 1.f, bt = gamma ? beta[n * gs + c] : 0.f;
        const size_t base = (size_t)n * HW * C + c;
        for (int r = r0 + rl; r < r1; r += 4) {
            const size_t e = base + (size_t)r * C;
            const float xh = (x[e] - m) * rs;
            const float dz = in_dz(dy[e], xh, g, bt, act);
            a += (double)dz;
            b += (double)dz * (double)xh;
        }
    }
    red[0][rl][cl] = a;
    red[1][rl][cl] = b;
    __syncthreads();
    if (rl == 0 && c < C) {
        double* o = ws + ((size_t)(n * C + c) * S + s) * 2;
        o[0] = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
        o[1] = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
    }
}

// s12[i*2+{0,1}] = {S1/HW, S2/HW} as floats; dgamma = S2, dbeta = S1.  One wavefront per (n, c).
__global__ __launch_bounds__(256) void in_bwd_final(const double* __restrict__ ws, float* __restrict__ s12,
                                                    float* __restrict__ dgamma, float* __restrict__ dbeta, int NC, int S,
                                                    int HW, int C, int gs) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= NC) return;
    const int lane = threadIdx.x & 63;
    double a = 0.0, b = 0.0;
    for (int s = lane; s < S; s += 64) {
        a += ws[((size_t)i * S + s) * 2];
        b += ws[((size_t)i * S + s) * 2 + 1];
    }
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    if (lane == 0) {
        s12[i * 2] = (float)(a / HW);
        s12[i * 2 + 1] = (float)(b / HW);
        const int gi = (i / C) * gs + (i % C);
        if (dgamma) dgamma[gi] = (float)b;
        if (dbeta) dbeta[gi] = (float)a;
    }
}

// dx = rstd * gamma * (dz - S1/HW - xhat * S2/HW)
__global__ __launch_bounds__(256) void in_bwd_apply(const float* __restrict__ dy, const float* __restrict__ x,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ s12, float* __restrict__ dx, size_t total,
                                                    int HW, int C, int act, int gs) {
    const size_t per_n = (size_t)HW * C;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(e / per_n);
        const int c = (int)(e % C);
        const int sc = n * C + c;
        const float rs = rstd[sc];
        const float g = gamma ? gamma[n * gs + c] : 1.f, bt = gamma ? beta[n * gs + c] : 0.f;
        const float xh = (x[e] - mean[sc]) * rs;
        const float dz = in_dz(dy[e], xh, g, bt, act);
        dx[e] = rs * g * (dz - s12[sc * 2] - xh * s12[sc * 2 + 1]);
    }
}

// ---------------------------------------------------------------------------------------
// Bandwidth-shaped variants (C % 4 == 0 and 256 % (C/4) == 0, i.e. every width of the shipped networks):
// every thread owns ONE channel quad for its whole life, so mean / rstd / gamma / beta sit in registers, all index
// arithmetic is 32-bit and loop-invariant, and every access is a 16-byte vector.  blockIdx.y = sample.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

struct Quad {
    float m[4], r[4], g[4], b[4];
};
__device__ __forceinline__ Quad load_quad(const float* mean, const float* rstd, const float* gamma, const float* beta,
                                          int n, int C, int c, int gs) {
    Quad q;
    const float4 m = ld4(mean + n * C + c), r = ld4(rstd + n * C + c);
    q.m[0] = m.x; q.m[1] = m.y; q.m[2] = m.z; q.m[3] = m.w;
    q.r[0] = r.x; q.r[1] = r.y; q.r[2] = r.z; q.r[3] = r.w;
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // gamma / beta live inside the MLP output row: 4-byte aligned only
        q.g[k] = gamma ? gamma[n * gs + c + k] : 1.f;
        q.b[k] = gamma ? beta[n * gs + c + k] : 0.f;
    }
    return q;
}

// y = act((x - mean) * rstd * gamma + beta) + residual
template <bool RES>
__global__ __launch_bounds__(256) void in_apply_q(const float* __restrict__ x, const float* __restrict__ mean,
                                                  const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, const float* __restrict__ residual,
                                                  float* __restrict__ y, int HW, int C, int act, int gs) {
    const int n = blockIdx.y;
    const int nq = HW * (C >> 2);                       // quads per sample
    const int step = gridDim.x * 256;                   // multiple of C/4: the channel quad is loop-invariant
    int i = blockIdx.x * 256 + threadIdx.x;
    const int c = (i % (C >> 2)) << 2;
    const Quad q = load_quad(mean, rstd, gamma, beta, n, C, c, gs);
    const size_t base = (size_t)n * HW * C;
    x += base; y += base;
    if (RES) residual += base;
    for (; i < nq; i += step) {
        const float4 v = ld4(x + 4 * (size_t)i);
        float4 rr;
        if (RES) rr = ld4(residual + 4 * (size_t)i);
        float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = cg_apply_act((o[k] - q.m[k]) * q.r[k] * q.g[k] + q.b[k], act);
        if (RES) { o[0] += rr.x; o[1] += rr.y; o[2] += rr.z; o[3] += rr.w; }
        st4(y + 4 * (size_t)i, make_float4(o[0], o[1], o[2], o[3]));
    }
}

// same, writing the {hi, lo} fp16 planes the split-precision convolution consumes (and optionally fp32 as well)
template <bool RES, bool F32>
__global__ __launch_bounds__(256) void in_apply_split_q(const float* __restrict__ x, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ residual,
                                                        float* __restrict__ y, _Float16* __restrict__ ys, size_t lo_elems,
                                                        int HW, int C, int act, int gs) {
    const int n = blockIdx.y;
    const int nq = HW * (C >> 2);
    const int step = gridDim.x * 256;
    int i = blockIdx.x * 256 + threadIdx.x;
    const int c = (i % (C >> 2)) << 2;
    const Quad q = load_quad(mean, rstd, gamma, beta, n, C, c, gs);
    const size_t base = (size_t)n * HW * C;
    x += base; ys += cg_il(base);        // per-sample element counts are multiples of 32
    if (F32) y += base;
    if (RES) residual += base;
    for (; i < nq; i += step) {
        const float4 v = ld4(x + 4 * (size_t)i);
        float4 rr;
        if (RES) rr = ld4(residual + 4 * (size_t)i);
        float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = cg_apply_act((o[k] - q.m[k]) * q.r[k] * q.g[k] + q.b[k], act);
        if (RES) { o[0] += rr.x; o[1] += rr.y; o[2] += rr.z; o[3] += rr.w; }
        if (F32) st4(y + 4 * (size_t)i, make_float4(o[0], o[1], o[2], o[3]));
        _Float16 h[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h[k] = (_Float16)fminf(fmaxf(o[k], -65504.f), 65504.f);
            l[k] = (_Float16)(o[k] - (float)h[k]);
        }
        *reinterpret_cast<uint2*>(ys + cg_il(4 * (size_t)i)) = *reinterpret_cast<const uint2*>(h);
        *reinterpret_cast<uint2*>(ys + lo_elems + cg_il(4 * (size_t)i)) = *reinterpret_cast<const uint2*>(l);
    }
}

// the reference applies gamma as a multiplier of the normalised value: keep (x-m)*r first, then *g (same rounding
// order as in_apply_kernel)

// Octet variant (C % 8 == 0, 256 % (C/8) == 0, interleaved {hi, lo} layout): a thread owns EIGHT channels, so the halves
// leave as two 16-byte stores (8 hi, 8 lo) instead of four 8-byte ones -- 8-byte stores run at 0.54-0.70x the 16-byte rate
// (MI355X_MICROARCH.md).  Same arithmetic, element by element, as in_apply_split_q.
template <bool RES, bool F32>
__global__ __launch_bounds__(256) void in_apply_split_o(const float* __restrict__ x, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ residual,
                                                        float* __restrict__ y, _Float16* __restrict__ ys, size_t lo_elems,
                                                        int HW, int C, int act, int gs) {
    const int n = blockIdx.y;
    const int no = HW * (C >> 3);                       // octets per sample
    const int step = gridDim.x * 256;                   // multiple of C/8: the channel octet is loop-invariant
    int i = blockIdx.x * 256 + threadIdx.x;
    const int c = (i % (C >> 3)) << 3;
    const Quad q0 = load_quad(mean, rstd, gamma, beta, n, C, c, gs), q1 = load_quad(mean, rstd, gamma, beta, n, C, c + 4, gs);
    const size_t base = (size_t)n * HW * C;
    x += base; ys += cg_il(base);        // per-sample element counts are multiples of 32
    if (F32) y += base;
    if (RES) residual += base;
    for (; i < no; i += step) {
        const size_t e = 8 * (size_t)i;
        const float4 v0 = ld4(x + e), v1 = ld4(x + e + 4);
        float4 r0, r1;
        if (RES) { r0 = ld4(residual + e); r1 = ld4(residual + e + 4); }
        float o[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o[k] = cg_apply_act((o[k] - q0.m[k]) * q0.r[k] * q0.g[k] + q0.b[k], act);
            o[4 + k] = cg_apply_act((o[4 + k] - q1.m[k]) * q1.r[k] * q1.g[k] + q1.b[k], act);
        }
        if (RES) { o[0] += r0.x; o[1] += r0.y; o[2] += r0.z; o[3] += r0.w; o[4] += r1.x; o[5] += r1.y; o[6] += r1.z; o[7] += r1.w; }
        if (F32) {
            st4(y + e, make_float4(o[0], o[1], o[2], o[3]));
            st4(y + e + 4, make_float4(o[4], o[5], o[6], o[7]));
        }
        _Float16 h[8], l[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            h[k] = (_Float16)fminf(fmaxf(o[k], -65504.f), 65504.f);
            l[k] = (_Float16)(o[k] - (float)h[k]);
        }
        *reinterpret_cast<uint4*>(ys + cg_il(e)) = *reinterpret_cast<const uint4*>(h);
        *reinterpret_cast<uint4*>(ys + lo_elems + cg_il(e)) = *reinterpret_cast<const uint4*>(l);
    }
}

// ws[((n*C + c)*S + s)*2 + {0,1}] = {sum dz, sum dz*xhat} over the rows of split s.  Block = (C/4) channel quads x
// 256/(C/4) row lanes; each thread streams float4s of its quad down the rows.
__global__ __launch_bounds__(256) void in_bwd_partial_q(const float* __restrict__ dy, const float* __restrict__ x,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        double* __restrict__ ws, int HW, int C, int S, int rows_per_split,
                                                        int act, int gs, float* __restrict__ mx) {
    // mx (optional): mx[((n*C + c)*S + s)*2 + {0,1}] = {max |dz|, max |xhat|} over the rows of split s -- what
    // cg_instnorm_bwd_split needs to bound |dx| before dx exists
    __shared__ double red[256 * 8];
    const int Q = C >> 2, RL = 256 / Q;
    const int qi = threadIdx.x % Q, rl = threadIdx.x / Q;
    const int n = blockIdx.y, s = blockIdx.x;
    const int c = qi << 2;
    const Quad q = load_quad(mean, rstd, gamma, beta, n, C, c, gs);
    const int r0 = s * rows_per_split, r1 = min(r0 + rows_per_split, HW);
    const size_t base = (size_t)n * HW * C + c;
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    float mz[4] = {0.f, 0.f, 0.f, 0.f}, mh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += RL) {
        const float4 xv = ld4(x + base + (size_t)r * C), dv = ld4(dy + base + (size_t)r * C);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (xs[k] - q.m[k]) * q.r[k];
            const float dz = in_dz(ds[k], xh, q.g[k], q.b[k], act);
            a[k] += (double)dz;
            b[k] += (double)dz * (double)xh;
            mz[k] = fmaxf(mz[k], fabsf(dz));
            mh[k] = fmaxf(mh[k], fabsf(xh));
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[threadIdx.x * 8 + k] = a[k];
        red[threadIdx.x * 8 + 4 + k] = b[k];
    }
    __syncthreads();
    if (rl == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double sa = 0.0, sb = 0.0;
            for (int j = 0; j < RL; ++j) {
                sa += red[(j * Q + qi) * 8 + k];
                sb += red[(j * Q + qi) * 8 + 4 + k];
            }
            double* o = ws + ((size_t)(n * C + c + k) * S + s) * 2;
            o[0] = sa;
            o[1] = sb;
        }
    }
    if (mx) {      // same reduction over the row lanes for the two maxima (floats reuse the LDS array)
        __syncthreads();
        float* redf = reinterpret_cast<float*>(red);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            redf[threadIdx.x * 8 + k] = mz[k];
            redf[threadIdx.x * 8 + 4 + k] = mh[k];
        }
        __syncthreads();
        if (rl == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float sa = 0.f, sb = 0.f;
                for (int j = 0; j < RL; ++j) {
                    sa = fmaxf(sa, redf[(j * Q + qi) * 8 + k]);
                    sb = fmaxf(sb, redf[(j * Q + qi) * 8 + 4 + k]);
                }
                float* o = mx + ((size_t)(n * C + c + k) * S + s) * 2;
                o[0] = sa;
                o[1] = sb;
            }
        }
    }
}

// bound[i] >= max |dx| of (sample, channel) i:  |dx| = rstd |gamma| |dz - S1/HW - xhat S2/HW| <= rstd |gamma| (max|dz| + |S1/HW| +
// max|xhat| |S2/HW|).  One wavefront per (n, c), after in_bwd_final wrote s12.
__global__ __launch_bounds__(256) void in_bwd_bound(const float* __restrict__ mx, const float* __restrict__ s12,
                                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                    float* __restrict__ bound, int NC, int S, int C, int gs) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= NC) return;
    const int lane = threadIdx.x & 63;
    float a = 0.f, b = 0.f;
    for (int s = lane; s < S; s += 64) {
        a = fmaxf(a, mx[((size_t)i * S + s) * 2]);
        b = fmaxf(b, mx[((size_t)i * S + s) * 2 + 1]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a = fmaxf(a, __shfl_xor(a, o, 64));
        b = fmaxf(b, __shfl_xor(b, o, 64));
    }
    if (lane == 0) {
        const float g = gamma ? fabsf(gamma[(i / C) * gs + (i % C)]) : 1.f;
        bound[i] = rstd[i] * g * (a + fabsf(s12[i * 2]) + b * fabsf(s12[i * 2 + 1]));
    }
}

// dx = rstd * gamma * (dz - S1/HW - xhat * S2/HW) written DIRECTLY as {hi, lo} fp16 planes of scale * dx (interleaved layout,
// a thread owns a channel octet: two 16-byte stores) -- the form the split-precision data- and weight-gradient kernels of
// the convolution in front of this norm read.  scale = the power of two that puts max(bound) into [4096, 8192): every
// block reduces the N*C bounds again (a few KiB from L2).  Optionally dx itself in fp32 too.
__global__ __launch_bounds__(256) void in_bwd_apply_split_o(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ s12, const float* __restrict__ bound,
                                                            int NC, float* __restrict__ state, _Float16* __restrict__ out,
                                                            size_t lo_elems, float* __restrict__ dx, int HW, int C, int act,
                                                            int gs) {
    __shared__ float red[4];
    float m = 0.f;
    for (int i = threadIdx.x; i < NC; i += 256) m = fmaxf(m, bound[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const unsigned mb = __float_as_uint(m);
    const int e8 = (int)((mb >> 23) & 255u);
    float scale = 1.f;                                  // dyn_scale of conv_x3.inc: 2^(12 - floor(log2 m)); NaN / Inf / 0 -> 1
    if (e8 != 0 && e8 != 255) {
        int se = 127 + 12 - (e8 - 127);
        se = se < 1 ? 1 : (se > 254 ? 254 : se);
        scale = __uint_as_float((unsigned)se << 23);
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        state[0] = m;
        state[1] = scale;
    }
    const int n = blockIdx.y;
    const int no = HW * (C >> 3);
    const int step = gridDim.x * 256;
    int i = blockIdx.x * 256 + threadIdx.x;
    const int c = (i % (C >> 3)) << 3;
    const Quad q0 = load_quad(mean, rstd, gamma, beta, n, C, c, gs), q1 = load_quad(mean, rstd, gamma, beta, n, C, c + 4, gs);
    float s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        s1[k] = s12[(n * C + c + k) * 2];
        s2[k] = s12[(n * C + c + k) * 2 + 1];
    }
    const size_t base = (size_t)n * HW * C;
    x += base; dy += base; out += cg_il(base);
    if (dx) dx += base;
    for (; i < no; i += step) {
        const size_t e = 8 * (size_t)i;
        const float4 x0 = ld4(x + e), x1 = ld4(x + e + 4), d0 = ld4(dy + e), d1 = ld4(dy + e + 4);
        const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        const float ds[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        float o[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh0 = (xs[k] - q0.m[k]) * q0.r[k], xh1 = (xs[4 + k] - q1.m[k]) * q1.r[k];
            const float dz0 = in_dz(ds[k], xh0, q0.g[k], q0.b[k], act), dz1 = in_dz(ds[4 + k], xh1, q1.g[k], q1.b[k], act);
            o[k] = q0.r[k] * q0.g[k] * (dz0 - s1[k] - xh0 * s2[k]);
            o[4 + k] = q1.r[k] * q1.g[k] * (dz1 - s1[4 + k] - xh1 * s2[4 + k]);
        }
        if (dx) {
            st4(dx + e, make_float4(o[0], o[1], o[2], o[3]));
            st4(dx + e + 4, make_float4(o[4], o[5], o[6], o[7]));
        }
        _Float16 h[8], l[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = o[k] * scale;
            h[k] = (_Float16)fminf(fmaxf(v, -65504.f), 65504.f);
            l[k] = (_Float16)(v - (float)h[k]);
        }
        *reinterpret_cast<uint4*>(out + cg_il(e)) = *reinterpret_cast<const uint4*>(h);
        *reinterpret_cast<uint4*>(out + lo_elems + cg_il(e)) = *reinterpret_cast<const uint4*>(l);
    }
}

// dx = rstd * gamma * (dz - S1/HW - xhat * S2/HW)
__global__ __launch_bounds__(256) void in_bwd_apply_q(const float* __restrict__ dy, const float* __restrict__ x,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ s12, float* __restrict__ dx, int HW, int C,
                                                      int act, int gs, float* __restrict__ amax_state) {
    __shared__ float red[4];
    float amax = 0.f;
    const int n = blockIdx.y;
    const int nq = HW * (C >> 2);
    const int step = gridDim.x * 256;
    int i = blockIdx.x * 256 + threadIdx.x;
    const int c = (i % (C >> 2)) << 2;
    const Quad q = load_quad(mean, rstd, gamma, beta, n, C, c, gs);
    float s1[4], s2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        s1[k] = s12[(n * C + c + k) * 2];
        s2[k] = s12[(n * C + c + k) * 2 + 1];
    }
    const size_t base = (size_t)n * HW * C;
    x += base; dy += base; dx += base;
    for (; i < nq; i += step) {
        const float4 xv = ld4(x + 4 * (size_t)i), dv = ld4(dy + 4 * (size_t)i);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (xs[k] - q.m[k]) * q.r[k];
            const float dz = in_dz(ds[k], xh, q.g[k], q.b[k], act);
            o[k] = q.r[k] * q.g[k] * (dz - s1[k] - xh * s2[k]);
            amax = fmaxf(amax, fabsf(o[k]));
        }
        st4(dx + 4 * (size_t)i, make_float4(o[0], o[1], o[2], o[3]));
    }
    if (amax_state) {   // per-block max |dx| for the split-precision consumers (cg_split_f16_dynamic with nslots = grid size)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
        __syncthreads();
        if (threadIdx.x == 0)
            amax_state[2 + blockIdx.y * gridDim.x + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    }
}

__host__ inline bool quad_ok(int C) { return (C & 3) == 0 && C >= 4 && (256 % (C >> 2)) == 0; }
// blocks along x for the streaming kernels: enough to fill the chip, each thread keeps a few float4 in flight
__host__ inline unsigned quad_grid(int HW, int C, int N) {
    const long nq = (long)HW * (C >> 2);
    long b = (nq + 256 * 4 - 1) / (256 * 4);      // ~4 float4 per thread
    const long cap = (2048 + N - 1) / N;          // ~8 blocks per CU over all samples
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// ---------------------------------------------------------------------------------------
// LayerNorm (networks.py:670-686): per-sample over D = C*HW, unbiased std, y = (x-mean)/(std+eps)*gamma+beta
// ---------------------------------------------------------------------------------------
constexpr int LN_CHUNK = 8192;  // elements per block in the per-sample reductions

__global__ __launch_bounds__(256) void ln_partial(const float* __restrict__ x, const float* __restrict__ dy,
                                                  const float* __restrict__ gamma, const float* __restrict__ mean,
                                                  double* __restrict__ ws, size_t D, int C, int chunks, int mode) {
    // mode 0: {sum x, sum x^2};  mode 1: {sum dxhat, sum dxhat*(x-mean)} with dxhat = dy*gamma[c]
    __shared__ double red[2][4];
    const int n = blockIdx.y, ch = blockIdx.x;
    const size_t e0 = (size_t)ch * LN_CHUNK, e1 = min(e0 + (size_t)LN_CHUNK, D);
    const float* px = x + (size_t)n * D;
    const float mu = mode ? mean[n] : 0.f;
    double a = 0.0, b = 0.0;
    for (size_t e = e0 + threadIdx.x; e < e1; e += 256) {
        if (mode == 0) {
            double v = (double)px[e];
            a += v;
            b += v * v;
        } else {
            float dxh = dy[(size_t)n * D + e] * gamma[e % C];
            a += (double)dxh;
            b += (double)dxh * (double)(px[e] - mu);
        }
    }
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = a;
        red[1][threadIdx.x >> 6] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ws[((size_t)n * chunks + ch) * 2] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        ws[((size_t)n * chunks + ch) * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

__global__ void ln_stats_final(const double* __restrict__ ws, float* __restrict__ mean, float* __restrict__ stdv, int N,
                               int chunks, size_t D) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double a = 0.0, b = 0.0;
    for (int k = 0; k < chunks; ++k) {
        a += ws[((size_t)n * chunks + k) * 2];
        b += ws[((size_t)n * chunks + k) * 2 + 1];
    }
    double m = a / (double)D;
    double var = (b - (double)D * m * m) / (double)(D - 1);  // torch.std default: unbiased
    if (var < 0.0) var = 0.0;
    mean[n] = (float)m;
    stdv[n] = (float)sqrt(var);
}

__global__ __launch_bounds__(256) void ln_apply(const float* __restrict__ x, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, const float* __restrict__ mean,
                                                const float* __restrict__ stdv, float* __restrict__ y, size_t total,
                                                size_t D, int C, float eps) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(e / D);
        const int c = (int)(e % C);
        y[e] = (x[e] - mean[n]) / (stdv[n] + eps) * gamma[c] + beta[c];
    }
}

// ab[n*2+{0,1}] = {A, B} summed over chunks
__global__ void ln_bwd_final(const double* __restrict__ ws, float* __restrict__ ab, int N, int chunks) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double a = 0.0, b = 0.0;
    for (int k = 0; k < chunks; ++k) {
        a += ws[((size_t)n * chunks + k) * 2];
        b += ws[((size_t)n * chunks + k) * 2 + 1];
    }
    ab[n * 2] = (float)a;
    ab[n * 2 + 1] = (float)b;
}

// dx = inv*(dxhat - A/D) - B*inv^2/((D-1)*std) * (x - mean)
__global__ __launch_bounds__(256) void ln_bwd_apply(const float* __restrict__ dy, const float* __restrict__ x,
                                                    const float* __restrict__ gamma, const float* __restrict__ mean,
                                                    const float* __restrict__ stdv, const float* __restrict__ ab,
                                                    float* __restrict__ dx, size_t total, size_t D, int C, float eps) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(e / D);
        const int c = (int)(e % C);
        const float sd = stdv[n];
        const float inv = 1.f / (sd + eps);
        const float dxh = dy[e] * gamma[c];
        const float A = ab[n * 2], B = ab[n * 2 + 1];
        const float k2 = sd > 0.f ? B * inv * inv / ((float)(D - 1) * sd) : 0.f;
        dx[e] = inv * (dxh - A / (float)D) - k2 * (x[e] - mean[n]);
    }
}

// dgamma[c] = sum_{n,hw} dy * xhat ; dbeta[c] = sum dy  -- partial over row chunks then ordered sum
__global__ __launch_bounds__(256) void ln_dgamma_partial(const float* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ stdv, float* __restrict__ part,
                                                         int rows_total, int HW, int C, int rows_per_chunk, float eps) {
    __shared__ float red[2][4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(r0 + rows_per_chunk, rows_total);
    float a = 0.f, b = 0.f;
    if (c < C)
        for (int r = r0 + rl; r < r1; r += 4) {
            const int n = r / HW;
            const size_t e = (size_t)r * C + c;
            const float xh = (x[e] - mean[n]) / (stdv[n] + eps);
            a += dy[e] * xh;
            b += dy[e];
        }
    red[0][rl][cl] = a;
    red[1][rl][cl] = b;
    __syncthreads();
    if (rl == 0 && c < C) {
        part[((size_t)blockIdx.y * 2) * C + c] = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
        part[((size_t)blockIdx.y * 2 + 1) * C + c] = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
    }
}
__global__ void ln_dgamma_final(const float* __restrict__ part, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                int C, int chunks) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float a = 0.f, b = 0.f;
    for (int k = 0; k < chunks; ++k) {
        a += part[((size_t)k * 2) * C + c];
        b += part[((size_t)k * 2 + 1) * C + c];
    }
    dgamma[c] = a;
    dbeta[c] = b;
}

constexpr int LN_ROWS = 512;
inline unsigned ew_grid(size_t n) {
    size_t b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" size_t cg_instnorm_workspace(int N, int HW, int C) {
    // split partials (doubles) + {S1/HW, S2/HW} floats for the backward apply
    return (size_t)N * C * nc_splits(HW) * 2 * sizeof(double) + (size_t)N * C * 2 * sizeof(float);
}

extern "C" int cg_instnorm_stats(const float* x, int N, int HW, int C, float eps, float* mean, float* rstd, void* ws,
                                 size_t ws_bytes, cg_stream_t stream) {
    CG_CHECK_ARG(x && mean && rstd && N > 0 && HW > 0 && C > 0, "cg_instnorm_stats: bad args");
    if (!ws || ws_bytes < cg_instnorm_workspace(N, HW, C))
        return cg_set_error(CG_ERR_WORKSPACE, "cg_instnorm_stats: workspace too small");
    const int S = nc_splits(HW);
    hipLaunchKernelGGL(in_stats_partial, dim3(cg_div_up(C, 64), N, S), dim3(256), 0, cg_s(stream), x, (double*)ws, HW, C, S);
    CG_LAUNCH_CHECK("in_stats_partial");
    hipLaunchKernelGGL(in_stats_final, dim3(cg_div_up((size_t)N * C, 4)), dim3(256), 0, cg_s(stream), (const double*)ws,
                       mean, rstd, N * C, S, HW, eps);
    CG_LAUNCH_CHECK("in_stats_final");
    return CG_OK;
}

extern "C" int cg_instnorm_stats_from_partials(const double* part, int N, int HW, int C, int rows_per_partial, float eps,
                                               float* mean, float* rstd, cg_stream_t stream) {
    CG_CHECK_ARG(part && mean && rstd && N > 0 && HW > 0 && C > 0 && rows_per_partial > 0 && HW % rows_per_partial == 0,
                 "cg_instnorm_stats_from_partials: bad args");
    hipLaunchKernelGGL(in_stats_final_tiles, dim3(cg_div_up(C, 32), N), dim3(256), 0, cg_s(stream), part, mean, rstd, C,
                       HW / rows_per_partial, HW, eps);
    CG_LAUNCH_CHECK("in_stats_final_tiles");
    return CG_OK;
}

extern "C" int cg_instnorm_apply(const float* x, const float* mean, const float* rstd, const float* gamma,
                                 const float* beta, int gstride, const float* residual, float* y, int N, int HW, int C,
                                 int act, cg_stream_t stream) {
    CG_CHECK_ARG(x && mean && rstd && y && N > 0 && HW > 0 && C > 0, "cg_instnorm_apply: bad args");
    CG_CHECK_ARG((gamma == nullptr) == (beta == nullptr), "cg_instnorm_apply: gamma and beta go together");
    const size_t total = (size_t)N * HW * C;
    if (quad_ok(C) && (size_t)HW * C < (size_t)0x7fffffff) {
        dim3 grid(quad_grid(HW, C, N), N);
        if (residual)
            hipLaunchKernelGGL((in_apply_q<true>), grid, dim3(256), 0, cg_s(stream), x, mean, rstd, gamma, beta, residual, y, HW,
                               C, act, gstride);
        else
            hipLaunchKernelGGL((in_apply_q<false>), grid, dim3(256), 0, cg_s(stream), x, mean, rstd, gamma, beta, residual, y, HW,
                               C, act, gstride);
    } else if ((C & 3) == 0)
        hipLaunchKernelGGL((in_apply_kernel<true>), dim3(ew_grid(total / 4)), dim3(256), 0, cg_s(stream), x, mean, rstd,
                           gamma, beta, residual, y, total, HW, C, act, gstride);
    else
        hipLaunchKernelGGL((in_apply_kernel<false>), dim3(ew_grid(total)), dim3(256), 0, cg_s(stream), x, mean, rstd,
                           gamma, beta, residual, y, total, HW, C, act, gstride);
    CG_LAUNCH_CHECK("in_apply_kernel");
    return CG_OK;
}

extern "C" int cg_instnorm_apply_split(const float* x, const float* mean, const float* rstd, const float* gamma,
                                       const float* beta, int gstride, const float* residual, float* y, void* y_split,
                                       size_t y_lo_elems, int N, int HW, int C, int act, cg_stream_t stream) {
    CG_CHECK_ARG(x && mean && rstd && y_split && N > 0 && HW > 0 && C > 0, "cg_instnorm_apply_split: bad args");
    CG_CHECK_ARG((gamma == nullptr) == (beta == nullptr), "cg_instnorm_apply_split: gamma and beta go together");
    CG_CHECK_ARG(quad_ok(C) && (size_t)HW * C < (size_t)0x7fffffff &&
                     (CG_X3_INTERLEAVE ? (y_lo_elems == CG_X3_LO_ELEMS && C % 32 == 0) : y_lo_elems >= (size_t)N * HW * C),
                 "cg_instnorm_apply_split: channel count %d / plane offset not supported", C);
    _Float16* ys = (_Float16*)y_split;
    const bool oct = CG_X3_INTERLEAVE && (C & 7) == 0 && (256 % (C >> 3)) == 0;      // 16-byte stores of the halves
    if (oct) {
        const long no = (long)HW * (C >> 3);
        long b = (no + 256 * 4 - 1) / (256 * 4);
        const long cap = (2048 + N - 1) / N;
        if (b > cap) b = cap;
        if (b < 1) b = 1;
        dim3 grid((unsigned)b, N);
#define CG_AO(RES_, F32_)                                                                                                  \
    hipLaunchKernelGGL((in_apply_split_o<RES_, F32_>), grid, dim3(256), 0, cg_s(stream), x, mean, rstd, gamma, beta, residual, \
                       y, ys, y_lo_elems, HW, C, act, gstride)
        if (residual) { if (y) CG_AO(true, true); else CG_AO(true, false); }
        else { if (y) CG_AO(false, true); else CG_AO(false, false); }
#undef CG_AO
        CG_LAUNCH_CHECK("in_apply_split_o");
        return CG_OK;
    }
    dim3 grid(quad_grid(HW, C, N), N);
#define CG_AS(RES_, F32_)                                                                                                  \
    hipLaunchKernelGGL((in_apply_split_q<RES_, F32_>), grid, dim3(256), 0, cg_s(stream), x, mean, rstd, gamma, beta, residual, \
                       y, ys, y_lo_elems, HW, C, act, gstride)
    if (residual) { if (y) CG_AS(true, true); else CG_AS(true, false); }
    else { if (y) CG_AS(false, true); else CG_AS(false, false); }
#undef CG_AS
    CG_LAUNCH_CHECK("in_apply_split_q");
    return CG_OK;
}

extern "C" int cg_instnorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                               const float* gamma, const float* beta, int gstride, float* dx, float* dgamma,
                               float* dbeta, int N, int HW, int C, int act, void* ws, size_t ws_bytes,
                               float* amax_state, int* amax_nslots, cg_stream_t stream) {
    if (amax_nslots) *amax_nslots = 0;
    CG_CHECK_ARG(dy && x && mean && rstd && dx && N > 0 && HW > 0 && C > 0, "cg_instnorm_bwd: bad args");
    CG_CHECK_ARG((gamma == nullptr) == (beta == nullptr), "cg_instnorm_bwd: gamma and beta go together");
    if (!ws || ws_bytes < cg_instnorm_workspace(N, HW, C))
        return cg_set_error(CG_ERR_WORKSPACE, "cg_instnorm_bwd: workspace too small");
    const int S = nc_splits(HW);
    double* part = (double*)ws;
    float* s12 = (float*)(part + (size_t)N * C * S * 2);
    const bool quad = quad_ok(C) && (size_t)HW * C < (size_t)0x7fffffff;
    if (quad)
        hipLaunchKernelGGL(in_bwd_partial_q, dim3(S, N), dim3(256), 0, cg_s(stream), dy, x, mean, rstd, gamma, beta, part, HW, C,
                           S, NC_SPLIT_ROWS, act, gstride, (float*)nullptr);
    else
        hipLaunchKernelGGL(in_bwd_partial, dim3(cg_div_up(C, 64), N, S), dim3(256), 0, cg_s(stream), dy, x, mean, rstd, gamma,
                           beta, part, HW, C, S, act, gstride);
    CG_LAUNCH_CHECK("in_bwd_partial");
    hipLaunchKernelGGL(in_bwd_final, dim3(cg_div_up((size_t)N * C, 4)), dim3(256), 0, cg_s(stream), (const double*)part,
                       s12, dgamma, dbeta, N * C, S, HW, C, gstride);
    CG_LAUNCH_CHECK("in_bwd_final");
    const size_t total = (size_t)N * HW * C;
    if (quad) {
        unsigned gx = quad_grid(HW, C, N);
        float* st = nullptr;
        if (amax_state && amax_nslots && N <= 1024) {
            if (gx * (unsigned)N > 1024u) gx = 1024u / (unsigned)N;
            st = amax_state;
            *amax_nslots = (int)(gx * (unsigned)N);
        }
        hipLaunchKernelGGL(in_bwd_apply_q, dim3(gx, N), dim3(256), 0, cg_s(stream), dy, x, mean, rstd, gamma, beta,
                           (const float*)s12, dx, HW, C, act, gstride, st);
    }
    else
        hipLaunchKernelGGL(in_bwd_apply, dim3(ew_grid(total)), dim3(256), 0, cg_s(stream), dy, x, mean, rstd, gamma, beta,
                           (const float*)s12, dx, total, HW, C, act, gstride);
    CG_LAUNCH_CHECK("in_bwd_apply");
    return CG_OK;
}

// ---- instance-norm backward that hands dx to the split-precision convolution kernels directly ----------------------
static bool bwd_split_ok(int N, int HW, int C) {
    // quad_ok: the reduction pass (in_bwd_partial_q) walks the rows with 256 / (C / 4) lanes per row -- C <= 1024
    return CG_X3_INTERLEAVE && (C & 31) == 0 && (256 % (C >> 3)) == 0 && quad_ok(C) && (size_t)HW * C < (size_t)0x7fffffff && N >= 1;
}
extern "C" size_t cg_instnorm_bwd_split_workspace(int N, int HW, int C) {
    if (N <= 0 || HW <= 0 || C <= 0 || !bwd_split_ok(N, HW, C)) return 0;      // 0: this shape takes cg_instnorm_bwd
    return cg_instnorm_workspace(N, HW, C) + (size_t)N * C * nc_splits(HW) * 2 * sizeof(float) + (size_t)N * C * sizeof(float);
}
extern "C" int cg_instnorm_bwd_split(const float* dy, const float* x, const float* mean, const float* rstd,
                                     const float* gamma, const float* beta, int gstride, void* dx_split, size_t dx_lo_elems,
                                     float* state, float* dx, float* dgamma, float* dbeta, int N, int HW, int C, int act,
                                     void* ws, size_t ws_bytes, cg_stream_t stream) {
    CG_CHECK_ARG(dy && x && mean && rstd && dx_split && state && N > 0 && HW > 0 && C > 0, "cg_instnorm_bwd_split: bad args");
    CG_CHECK_ARG((gamma == nullptr) == (beta == nullptr), "cg_instnorm_bwd_split: gamma and beta go together");
    CG_CHECK_ARG(bwd_split_ok(N, HW, C) && dx_lo_elems == CG_X3_LO_ELEMS,
                 "cg_instnorm_bwd_split: needs the interleaved layout and C %% 32 == 0 (C = %d)", C);
    if (!ws || ws_bytes < cg_instnorm_bwd_split_workspace(N, HW, C))
        return cg_set_error(CG_ERR_WORKSPACE, "cg_instnorm_bwd_split: workspace too small");
    const int S = nc_splits(HW);
    double* part = (double*)ws;
    float* s12 = (float*)(part + (size_t)N * C * S * 2);
    float* mx = s12 + (size_t)N * C * 2;
    float* bound = mx + (size_t)N * C * S * 2;
    hipStream_t st = cg_s(stream);
    hipLaunchKernelGGL(in_bwd_partial_q, dim3(S, N), dim3(256), 0, st, dy, x, mean, rstd, gamma, beta, part, HW, C, S,
                       NC_SPLIT_ROWS, act, gstride, mx);
    CG_LAUNCH_CHECK("in_bwd_partial_q");
    hipLaunchKernelGGL(in_bwd_final, dim3(cg_div_up((size_t)N * C, 4)), dim3(256), 0, st, (const double*)part, s12, dgamma,
                       dbeta, N * C, S, HW, C, gstride);
    CG_LAUNCH_CHECK("in_bwd_final");
    hipLaunchKernelGGL(in_bwd_bound, dim3(cg_div_up((size_t)N * C, 4)), dim3(256), 0, st, (const float*)mx, (const float*)s12,
                       rstd, gamma, bound, N * C, S, C, gstride);
    CG_LAUNCH_CHECK("in_bwd_bound");
    const long no = (long)HW * (C >> 3);
    long b = (no + 256 * 4 - 1) / (256 * 4);
    const long cap = (2048 + N - 1) / N;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    hipLaunchKernelGGL(in_bwd_apply_split_o, dim3((unsigned)b, N), dim3(256), 0, st, dy, x, mean, rstd, gamma, beta,
                       (const float*)s12, (const float*)bound, N * C, state, (_Float16*)dx_split, dx_lo_elems, dx, HW, C, act,
                       gstride);
    CG_LAUNCH_CHECK("in_bwd_apply_split_o");
    return CG_OK;
}

extern "C" size_t cg_layernorm_workspace(int N, int HW, int C) {
    const size_t D = (size_t)HW * C;
    const size_t chunks = (D + LN_CHUNK - 1) / LN_CHUNK;
    const size_t rows = (size_t)N * HW;
    const size_t rchunks = (rows + LN_ROWS - 1) / LN_ROWS;
    return (size_t)N * chunks * 2 * sizeof(double) + (size_t)N * 2 * sizeof(float) + rchunks * 2 * C * sizeof(float);
}

extern "C" int cg_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                float* stdv, int N, int HW, int C, float eps, void* ws, size_t ws_bytes,
                                cg_stream_t stream) {
    CG_CHECK_ARG(x && gamma && beta && y && mean && stdv && N > 0 && HW > 0 && C > 0, "cg_layernorm_fwd: bad args");
    if (!ws || ws_bytes < cg_layernorm_workspace(N, HW, C))
        return cg_set_error(CG_ERR_WORKSPACE, "cg_layernorm_fwd: workspace too small");
    const size_t D = (size_t)HW * C;
    CG_CHECK_ARG(D > 1, "cg_layernorm_fwd: needs more than one element per sample");
    const int chunks = (int)((D + LN_CHUNK - 1) / LN_CHUNK);
    hipLaunchKernelGGL(ln_partial, dim3(chunks, N), dim3(256), 0, cg_s(stream), x, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (double*)ws, D, C, chunks, 0);
    CG_LAUNCH_CHECK("ln_partial");
    hipLaunchKernelGGL(ln_stats_final, dim3(cg_div_up(N, 64)), dim3(64), 0, cg_s(stream), (const double*)ws, mean, stdv, N,
                       chunks, D);
    CG_LAUNCH_CHECK("ln_stats_final");
    const size_t total = (size_t)N * D;
    hipLaunchKernelGGL(ln_apply, dim3(ew_grid(total)), dim3(256), 0, cg_s(stream), x, gamma, beta, (const float*)mean,
                       (const float*)stdv, y, total, D, C, eps);
    CG_LAUNCH_CHECK("ln_apply");
    return CG_OK;
}

extern "C" int cg_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                                const float* stdv, float* dx, float* dgamma, float* dbeta, int N, int HW, int C,
                                float eps, void* ws, size_t ws_bytes, cg_stream_t stream) {
    CG_CHECK_ARG(dy && x && gamma && mean && stdv && dx && dgamma && dbeta && N > 0 && HW > 0 && C > 0,
                 "cg_layernorm_bwd: bad args");
    if (!ws || ws_bytes < cg_layernorm_workspace(N, HW, C))
        return cg_set_error(CG_ERR_WORKSPACE, "cg_layernorm_bwd: workspace too small");
    const size_t D = (size_t)HW * C;
    const int chunks = (int)((D + LN_CHUNK - 1) / LN_CHUNK);
    double* part = (double*)ws;
    float* ab = (float*)(part + (size_t)N * chunks * 2);
    float* gpart = ab + (size_t)N * 2;
    hipLaunchKernelGGL(ln_partial, dim3(chunks, N), dim3(256), 0, cg_s(stream), x, dy, gamma, mean, part, D, C, chunks, 1);
    CG_LAUNCH_CHECK("ln_partial(bwd)");
    hipLaunchKernelGGL(ln_bwd_final, dim3(cg_div_up(N, 64)), dim3(64), 0, cg_s(stream), (const double*)part, ab, N, chunks);
    CG_LAUNCH_CHECK("ln_bwd_final");
    const size_t total = (size_t)N * D;
    hipLaunchKernelGGL(ln_bwd_apply, dim3(ew_grid(total)), dim3(256), 0, cg_s(stream), dy, x, gamma, mean, stdv,
                       (const float*)ab, dx, total, D, C, eps);
    CG_LAUNCH_CHECK("ln_bwd_apply");
    const int rows = N * HW;
    const int rchunks = (rows + LN_ROWS - 1) / LN_ROWS;
    hipLaunchKernelGGL(ln_dgamma_partial, dim3(cg_div_up(C, 64), rchunks), dim3(256), 0, cg_s(stream), dy, x, mean, stdv,
                       gpart, rows, HW, C, LN_ROWS, eps);
    CG_LAUNCH_CHECK("ln_dgamma_partial");
    hipLaunchKernelGGL(ln_dgamma_final, dim3(cg_div_up(C, 256)), dim3(256), 0, cg_s(stream), (const float*)gpart, dgamma,
                       dbeta, C, rchunks);
    CG_LAUNCH_CHECK("ln_dgamma_final");
    return CG_OK;
}

// HBM-bound elementwise / stencil / reduction kernels of the Council-GAN step (gfx950).
// Each cites the reference arithmetic it replaces.  All are grid-stride, 256-thread blocks,
// coalesced along the contiguous channel dimension of NHWC.
#include <stdarg.h>
#include <math.h>
#include "cg_common.h"

thread_local char cg_err_buf[512] = {0};

int cg_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(cg_err_buf, sizeof(cg_err_buf), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* cg_last_error(void) { return cg_err_buf; }
extern "C" int cg_version(void) { return 100; }

namespace {

inline unsigned ew_grid(size_t n) {
    size_t b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}
#define GRID_STRIDE(i, n) \
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (size_t)gridDim.x * blockDim.x)

__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dz,
                               size_t n, int act) {
    GRID_STRIDE(i, n) dz[i] = dy[i] * cg_act_grad_from_out(y[i], act);
}

__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, int act) {
    GRID_STRIDE(i, n) y[i] = cg_apply_act(x[i], act);
}
__global__ void fill_kernel(float* __restrict__ p, size_t n, float v) { GRID_STRIDE(i, n) p[i] = v; }
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, size_t n) {
    GRID_STRIDE(i, n) o[i] = a[i] + b[i];
}
__global__ void axpby_kernel(float alpha, const float* __restrict__ a, float beta, float* __restrict__ b, size_t n) {
    GRID_STRIDE(i, n) b[i] = alpha * a[i] + (beta == 0.f ? 0.f : beta * b[i]);
}

// ---- AvgPool2d(3, stride 2, pad 1, count_include_pad=False): networks.py:32,129 ---------------
__global__ void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C,
                                   int Ho, int Wo) {
    const size_t total = (size_t)N * Ho * Wo * C;
    GRID_STRIDE(i, total) {
        int c = (int)(i % C);
        size_t p = i / C;
        int ox = (int)(p % Wo);
        p /= Wo;
        int oy = (int)(p % Ho);
        int n = (int)(p / Ho);
        int y0 = max(2 * oy - 1, 0), y1 = min(2 * oy + 1, H - 1);
        int x0 = max(2 * ox - 1, 0), x1 = min(2 * ox + 1, W - 1);
        float s = 0.f;
        for (int yy = y0; yy <= y1; ++yy)
            for (int xx = x0; xx <= x1; ++xx) s += x[(((size_t)n * H + yy) * W + xx) * C + c];
        y[i] = s / (float)((y1 - y0 + 1) * (x1 - x0 + 1));
    }
}
__global__ void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int H, int W, int C,
                                   int Ho, int Wo) {
    const size_t total = (size_t)N * H * W * C;
    GRID_STRIDE(i, total) {
        int c = (int)(i % C);
        size_t p = i / C;
        int ix = (int)(p % W);
        p /= W;
        int iy = (int)(p % H);
        int n = (int)(p / H);
        // windows containing (iy, ix): oy in [ceil((iy-1)/2), floor((iy+1)/2)]
        int oy0 = max(iy / 2, 0), oy1 = min((iy + 1) / 2, Ho - 1);
        int ox0 = max(ix / 2, 0), ox1 = min((ix + 1) / 2, Wo - 1);
        float s = 0.f;
        for (int oy = oy0; oy <= oy1; ++oy) {
            int ny = min(2 * oy + 1, H - 1) - max(2 * oy - 1, 0) + 1;
            for (int ox = ox0; ox <= ox1; ++ox) {
                int nx = min(2 * ox + 1, W - 1) - max(2 * ox - 1, 0) + 1;
                s += dy[(((size_t)n * Ho + oy) * Wo + ox) * C + c] / (float)(ny * nx);
            }
        }
        dx[i] = s;
    }
}

// ---- nearest 2x upsample, networks.py:385 -----------------------------------------------------
__global__ void upsample_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C) {
    const size_t total = (size_t)N * 2 * H * 2 * W * C;
    GRID_STRIDE(i, total) {
        int c = (int)(i % C);
        size_t p = i / C;
        int ox = (int)(p % (2 * W));
        p /= 2 * W;
        int oy = (int)(p % (2 * H));
        int n = (int)(p / (2 * H));
        y[i] = x[(((size_t)n * H + (oy >> 1)) * W + (ox >> 1)) * C + c];
    }
}
__global__ void upsample_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int H, int W, int C) {
    const size_t total = (size_t)N * H * W * C;
    GRID_STRIDE(i, total) {
        int c = (int)(i % C);
        size_t p = i / C;
        int ix = (int)(p % W);
        p /= W;
        int iy = (int)(p % H);
        int n = (int)(p / H);
        const size_t rs = (size_t)2 * W * C;
        const size_t b = (((size_t)n * 2 * H + 2 * iy) * 2 * W + 2 * ix) * C + c;
        dx[i] = (dy[b] + dy[b + C]) + (dy[b + rs] + dy[b + rs + C]);
    }
}

// ---- AdaptiveAvgPool2d(1), networks.py:347 -----------------------------------------------------
__global__ __launch_bounds__(256) void gap_kernel(const float* __restrict__ x, float* __restrict__ y, int HW, int C) {
    __shared__ double red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl, n = blockIdx.y;
    double s = 0.0;
    if (c < C)
        for (int r = rl; r < HW; r += 4) s += (double)x[((size_t)n * HW + r) * C + c];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) y[(size_t)n * C + c] = (float)((red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]) / HW);
}

// ---- mask / blend head, networks.py:398-407 ----------------------------------------------------
// od (image channels) and k (number of masks) are template parameters so the per-pixel arrays stay
// in registers (a runtime-indexed local array would live in scratch).
template <int OD, int K>
__global__ void mask_blend_fwd_kernel(const float* __restrict__ nx, const float* __restrict__ im_in,
                                      float* __restrict__ im_out, float* __restrict__ mask, size_t npix) {
    constexpr int CH = OD * K + K;
    GRID_STRIDE(p, npix) {
        float v[CH], im[OD];
#pragma unroll
        for (int c = 0; c < CH; ++c) v[c] = nx[p * CH + c];
#pragma unroll
        for (int c = 0; c < OD; ++c) im[c] = im_in[p * OD + c];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const float m = (tanhf(10.f * v[OD * K + j]) + 1.f) * 0.5f;
            mask[p * K + j] = m;
#pragma unroll
            for (int c = 0; c < OD; ++c) im[c] = (1.f - m) * im[c] + m * v[OD * j + c];
        }
#pragma unroll
        for (int c = 0; c < OD; ++c) im_out[p * OD + c] = im[c];
    }
}
template <int OD, int K>
__global__ void mask_blend_bwd_kernel(const float* __restrict__ nx, const float* __restrict__ im_in,
                                      const float* __restrict__ d_out, const float* __restrict__ d_mask,
                                      float* __restrict__ d_nx, size_t npix) {
    constexpr int CH = OD * K + K;
    GRID_STRIDE(p, npix) {
        float v[CH], ims[K][OD], m[K], th[K], g[OD];  // ims[j] = image entering blend step j
#pragma unroll
        for (int c = 0; c < CH; ++c) v[c] = nx[p * CH + c];
#pragma unroll
        for (int c = 0; c < OD; ++c) ims[0][c] = im_in[p * OD + c];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            th[j] = tanhf(10.f * v[OD * K + j]);
            m[j] = (th[j] + 1.f) * 0.5f;
            if (j + 1 < K) {
#pragma unroll
                for (int c = 0; c < OD; ++c) ims[j + 1 < K ? j + 1 : 0][c] = (1.f - m[j]) * ims[j][c] + m[j] * v[OD * j + c];
            }
        }
#pragma unroll
        for (int c = 0; c < OD; ++c) g[c] = d_out[p * OD + c];
#pragma unroll
        for (int j = K - 1; j >= 0; --j) {
            float dm = d_mask ? d_mask[p * K + j] : 0.f;
#pragma unroll
            for (int c = 0; c < OD; ++c) {
                d_nx[p * CH + OD * j + c] = g[c] * m[j];
                dm += g[c] * (v[OD * j + c] - ims[j][c]);
                g[c] *= (1.f - m[j]);
            }
            // d/dt (tanh(10 t) + 1)/2 = 5 (1 - tanh^2(10 t))
            d_nx[p * CH + OD * K + j] = dm * 5.f * (1.f - th[j] * th[j]);
        }
    }
}

// ---- LSGAN, networks.py:64,90,166,194 ------------------------------------------------------------
// One block per council member (blockIdx.x): nb = samples of ONE member, whose rows follow the previous member's.
__global__ __launch_bounds__(256) void lsgan_fwd_kernel(const float* __restrict__ out, const float* __restrict__ tgt,
                                                        const float* __restrict__ wt, int nb, int hw, int group,
                                                        float* __restrict__ loss, int accumulate) {
    __shared__ double red[4];
    double s = 0.0;
    const int total = nb * hw;
    const int s0 = (int)blockIdx.x * nb;
    out += (size_t)s0 * hw;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int sidx = s0 + i / hw;
        const float d = out[i] - tgt[sidx];
        s += (double)wt[sidx] * (double)d * (double)d;
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t = (red[0] + red[1] + red[2] + red[3]) / ((double)group * hw);
        loss[blockIdx.x] = (accumulate ? loss[blockIdx.x] : 0.f) + (float)t;
    }
}
__global__ void lsgan_bwd_kernel(const float* __restrict__ out, const float* __restrict__ tgt,
                                 const float* __restrict__ wt, const float* __restrict__ gscale, int nb, int hw,
                                 int group, float* __restrict__ d_out, int nmember) {
    const size_t total = (size_t)nb * nmember * hw;      // nb = samples of ONE member; gscale has one entry per member
    const float k = 2.f / ((float)group * (float)hw);
    GRID_STRIDE(i, total) {
        const int sidx = (int)(i / hw);
        d_out[i] = gscale[sidx / nb] * k * wt[sidx] * (out[i] - tgt[sidx]);
    }
}

// ---- focus-loss criteria, trainer_council.py:230-250 ------------------------------------------
constexpr int FOCUS_BLOCKS = 256;
// stage 1: FOCUS_BLOCKS partial triples (fp64); stage 2: one wave adds them in a fixed order
// blockIdx.y = council member (its `total` mask elements follow the previous member's; partials likewise)
__global__ __launch_bounds__(256) void focus_partial_kernel(const float* __restrict__ mask, int H, int W, int k,
                                                            size_t total, float center, float eps,
                                                            double* __restrict__ part) {
    mask += (size_t)blockIdx.y * total;
    part += (size_t)blockIdx.y * gridDim.x * 3;
    __shared__ double red[3][4];
    double a = 0.0, b = 0.0, t = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const float m = mask[i];
        size_t p = i / k;
        const int x = (int)(p % W);
        p /= W;
        const int y = (int)(p % H);
        a += 1.0 / ((double)fabsf(m - center) + (double)eps);
        b += (double)m;
        if (y + 1 < H) t += (double)fabsf(mask[i + (size_t)W * k] - m);
        if (x + 1 < W) t += (double)fabsf(mask[i + k] - m);
    }
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    t = wave_sum_d(t);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = a;
        red[1][threadIdx.x >> 6] = b;
        red[2][threadIdx.x >> 6] = t;
    }
    __syncthreads();
    if (threadIdx.x < 3)
        part[(size_t)blockIdx.x * 3 + threadIdx.x] =
            red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}
__global__ __launch_bounds__(64) void focus_final_kernel(const double* __restrict__ part, int nblocks,
                                                         float* __restrict__ sums) {
    part += (size_t)blockIdx.x * nblocks * 3;       // blockIdx.x = council member
    sums += blockIdx.x * 3;
    for (int q = 0; q < 3; ++q) {
        double s = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += 64) s += part[(size_t)b * 3 + q];
        s = wave_sum_d(s);
        if (threadIdx.x == 0) sums[q] = (float)s;
    }
}
__global__ void focus_total_kernel(const float* __restrict__ sums, float numel, float w_zo, float w_total, float w_tv,
                                   int use_abs, int use_square, float* __restrict__ out) {
    if (threadIdx.x != 0) return;
    sums += blockIdx.x * 3;                                 // blockIdx.x = council member
    out += blockIdx.x * 4;
    const float zo = sums[0] / numel;                       // trainer_council.py:230-231
    float small = 0.f;
    if (use_abs) small += fabsf(sums[1]) / numel;           // :245-246
    if (use_square) small += (sums[1] / numel) * (sums[1] / numel);  // :242-243
    const float tv = sums[2] / numel;                       // :248-250
    out[0] = w_zo * zo + w_total * small + w_tv * tv;
    out[1] = zo;
    out[2] = small;
    out[3] = tv;
}
__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }
__global__ void focus_bwd_kernel(const float* __restrict__ mask, const float* __restrict__ sums,
                                 const float* __restrict__ gscale, int N, int H, int W, int k, float center, float eps,
                                 float w_zo, float w_total, float w_tv, int use_abs, int use_square,
                                 float* __restrict__ d_mask) {
    // N = samples of ONE council member; blockIdx.y = member (its mask block, its three sums, its upstream gradient)
    const size_t total = (size_t)N * H * W * k;
    const float numel = (float)total;
    mask += (size_t)blockIdx.y * total;
    d_mask += (size_t)blockIdx.y * total;
    sums += blockIdx.y * 3;
    const float gs = gscale[blockIdx.y];
    // mask_small: abs -> |sum m|/numel ; square -> (sum m/numel)^2   (trainer_council.py:242-246)
    float dsmall = 0.f;
    if (use_abs) dsmall += sgn(sums[1]) / numel;
    if (use_square) dsmall += 2.f * sums[1] / (numel * numel);
    GRID_STRIDE(i, total) {
        const float m = mask[i];
        size_t p = i / k;
        const int x = (int)(p % W);
        p /= W;
        const int y = (int)(p % H);
        float g = 0.f;
        if (w_zo != 0.f) {
            const float d = m - center;
            const float den = fabsf(d) + eps;
            g += w_zo * (-sgn(d) / (den * den)) / numel;
        }
        if (w_total != 0.f) g += w_total * dsmall;
        if (w_tv != 0.f) {
            float tv = 0.f;
            const size_t rs = (size_t)W * k;
            if (y + 1 < H) tv -= sgn(mask[i + rs] - m);
            if (y > 0) tv += sgn(m - mask[i - rs]);
            if (x + 1 < W) tv -= sgn(mask[i + k] - m);
            if (x > 0) tv += sgn(m - mask[i - k]);
            g += w_tv * tv / numel;
        }
        d_mask[i] = gs * g;
    }
}

// ---- L1 mean (recon / council-abs criteria), trainer_council.py:207-208,227-228 ----------------
__global__ __launch_bounds__(1024) void l1_mean_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           size_t n, float* __restrict__ loss) {
    __shared__ double red[16];
    double s = 0.0;
    for (size_t i = threadIdx.x; i < n; i += 1024) s += (double)fabsf(a[i] - b[i]);
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[w];
        loss[0] = (float)(t / (double)n);
    }
}
__global__ void l1_mean_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                   const float* __restrict__ gscale, size_t n, float* __restrict__ da) {
    const float gs = gscale[0] / (float)n;
    GRID_STRIDE(i, n) da[i] = gs * sgn(a[i] - b[i]);
}

// ---- torch.optim.Adam (L2 weight decay), trainer_council.py:170-179 ----------------------------
// hyper_dev (optional): {step_size, sqrt(bias_correction2)} read from device memory instead of the kernel arguments -- the two
// numbers that change from step to step, so that a launch captured in a hipGraph stays valid (cg_adam_step_dev)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, size_t n, float beta1, float beta2, float eps, float wd,
                            float step_size, float bc2_sqrt, long long mstride, const float* __restrict__ hyper_dev) {
    if (hyper_dev) {
        step_size = hyper_dev[0];
        bc2_sqrt = hyper_dev[1];
    }
    const float w1 = 1.f - beta1;
    const long long mo = (long long)blockIdx.y * mstride;       // blockIdx.y = council member (same run of its pool slice)
    p += mo; g += mo; m += mo; v += mo;
    GRID_STRIDE(i, n) {
        const float pi = p[i];
        const float gi = g[i] + wd * pi;                      // grad.add(param, alpha=weight_decay)
        float mi = m[i];
        // exp_avg.lerp_(grad, 1 - beta1) with ATen's two-sided lerp formula
        mi = (w1 < 0.5f) ? mi + w1 * (gi - mi) : gi - (gi - mi) * (1.f - w1);
        const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;  // mul_(beta2).addcmul_(g, g, 1-beta2)
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
        m[i] = mi;
        v[i] = vi;
    }
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                   float* __restrict__ out, int nidx, size_t row_elems) {
    const size_t total = (size_t)nidx * row_elems;
    GRID_STRIDE(i, total) {
        const size_t r = i / row_elems, e = i - r * row_elems;
        out[i] = src[(size_t)idx[r] * row_elems + e];
    }
}

// out[i] = (idx[i] >= 0 ? src_a[idx[i]] : src_b[-idx[i] - 1]) along dim 0: the discriminator batches [own fake | real],
// [own translation | colleagues' translations] are assembled from two tensors without a torch.cat
__global__ void gather_rows2_kernel(const float* __restrict__ a, const float* __restrict__ b, const int32_t* __restrict__ idx,
                                    float* __restrict__ out, int nidx, size_t row4) {
    const size_t total = (size_t)nidx * row4;      // float4 units (row_elems % 4 == 0)
    GRID_STRIDE(i, total) {
        const size_t r = i / row4, e = i - r * row4;
        const int k = idx[r];
        const float4* src = reinterpret_cast<const float4*>(k >= 0 ? a : b) + (size_t)(k >= 0 ? k : -k - 1) * row4;
        reinterpret_cast<float4*>(out)[i] = src[e];
    }
}

// per member m: council[m] = council_w * (w_match ? w_match[m] : 1) * lc[m];  total[m] = focus[m] + gan_w * adv[m] +
// council[m];  gcouncil[m] = council_w * w_match[m] (the upstream gradient of lc).  NULL terms count as zero.
__global__ void gen_total_kernel(const float* __restrict__ focus, const float* __restrict__ adv, const float* __restrict__ lc,
                                 const float* __restrict__ w_match, float gan_w, float council_w, float* __restrict__ total,
                                 float* __restrict__ council, float* __restrict__ gcouncil, int n) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n) return;
    const float wm = w_match ? w_match[m] : 1.f;
    const float c = lc ? council_w * wm * lc[m] : 0.f;
    if (council) council[m] = c;
    if (gcouncil) gcouncil[m] = council_w * wm;
    total[m] = (focus ? focus[m * 4] : 0.f) + (adv ? gan_w * adv[m] : 0.f) + c;
}

// ring[pos % n] = value ; optional w = mean(ring_a) / mean(ring_b)  (trainer_council.py:518-524,576-581)
// blockIdx.x = council member: rings [member][n], value / w_out one entry per member
__global__ void ring_kernel(float* __restrict__ ring_a, float* __restrict__ ring_b, int n, int pos,
                            const float* __restrict__ value, float* __restrict__ w_out, int push_b,
                            const int32_t* __restrict__ pos_dev) {
    if (threadIdx.x != 0) return;
    if (pos_dev) pos = pos_dev[0];      // write position from device memory (hipGraph-captured launches)
    ring_a += (size_t)blockIdx.x * n;
    if (ring_b) ring_b += (size_t)blockIdx.x * n;
    value += blockIdx.x;
    if (w_out) w_out += blockIdx.x;
    if (push_b)
        ring_b[pos % n] = value[0];
    else
        ring_a[pos % n] = value[0];
    if (w_out) {
        double sa = 0.0, sb = 0.0;
        for (int i = 0; i < n; ++i) {
            sa += (double)ring_a[i];
            sb += (double)ring_b[i];
        }
        w_out[0] = (float)((sa / n) / (sb / n));
    }
}

// ---- two 1x1 convolutions with nothing between them, composed (MsImageDisCouncil's last two layers, networks.py:142-143:
//      Conv2d(dim, dim, 1) -> Conv2d(dim, 1, 1), no activation): W2 (W1 y + b1) + b2 = (W2 W1) y + (W2 b1 + b2) -- a dim -> 1
//      convolution instead of a dim -> dim one.  out (per member, stride S floats): [0, C) = w_eff, [C] = b_eff.
//      Grid (ceil(C / 64) + 1, members), 512 threads: a block owns 64 columns of w_eff, its eight waves each a slice of the j range
//      (independent loads in flight; a thread per column walking all C rows alone is a chain of C dependent-latency steps: 227 us
//      at C = 512), partials added in wave order; the last block computes b_eff.  Fixed summation order: deterministic.
__global__ __launch_bounds__(512) void compose1x1_fwd_kernel(const float* __restrict__ W1, const float* __restrict__ b1,
                                                             const float* __restrict__ W2, const float* __restrict__ b2, int C,
                                                             long long pstride, float* __restrict__ out, int S) {
    const int m = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    W1 += (long long)m * pstride; b1 += (long long)m * pstride; W2 += (long long)m * pstride; b2 += (long long)m * pstride;
    out += (size_t)m * S;
    __shared__ float red[8][64];
    if (blockIdx.x + 1 < gridDim.x) {
        const int k = blockIdx.x * 64 + lane;
        const int per = (C + 7) / 8, j0 = wv * per, j1 = min(C, j0 + per);
        float a = 0.f;
        if (k < C) {
#pragma unroll 8
            for (int j = j0; j < j1; ++j) a = fmaf(W2[j], W1[(size_t)j * C + k], a);
        }
        red[wv][lane] = a;
        __syncthreads();
        if (wv == 0 && k < C) {
            float t = red[0][lane];
#pragma unroll
            for (int w = 1; w < 8; ++w) t += red[w][lane];
            out[k] = t;
        }
    } else {
        float a = 0.f;
        for (int j = threadIdx.x; j < C; j += 512) a = fmaf(W2[j], b1[j], a);
        a = wave_sum(a);
        if (lane == 0) red[0][wv] = a;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = b2[0];
#pragma unroll
            for (int w = 0; w < 8; ++w) t += red[0][w];
            out[C] = t;
        }
    }
}
// d (per member, stride S): [0, C) = d w_eff, [C] = d b_eff  ->  dW1[j][k] += W2[j] d w_eff[k];  db1[j] += W2[j] d b_eff;
// dW2[j] += sum_k d w_eff[k] W1[j][k] + d b_eff b1[j];  db2 += d b_eff
__global__ __launch_bounds__(256) void compose1x1_bwd_kernel(const float* __restrict__ d, int S, const float* __restrict__ W1,
                                                             const float* __restrict__ b1, const float* __restrict__ W2, int C,
                                                             long long pstride, float* __restrict__ dW1, float* __restrict__ db1,
                                                             float* __restrict__ dW2, float* __restrict__ db2) {
    const int m = blockIdx.y, j = blockIdx.x;
    d += (size_t)m * S;
    const long long off = (long long)m * pstride;
    const float w2 = W2[off + j], dbe = d[C];
    float a = 0.f;
    for (int k = threadIdx.x; k < C; k += 256) {
        const float dk = d[k];
        dW1[off + (size_t)j * C + k] += w2 * dk;
        a = fmaf(dk, W1[off + (size_t)j * C + k], a);
    }
    __shared__ float red[4];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        dW2[off + j] += (red[0] + red[1]) + (red[2] + red[3]) + dbe * b1[off + j];
        db1[off + j] += w2 * dbe;
        if (j == 0) db2[off] += dbe;
    }
}

}  // namespace

#define EW_LAUNCH(kernel, n, ...)                                                                  \
    hipLaunchKernelGGL(kernel, dim3(ew_grid(n)), dim3(256), 0, cg_s(stream), __VA_ARGS__);         \
    CG_LAUNCH_CHECK(#kernel);                                                                      \
    return CG_OK

extern "C" int cg_act_bwd(const float* dy, const float* y, float* dz, size_t n, int act, cg_stream_t stream) {
    CG_CHECK_ARG(dy && y && dz, "cg_act_bwd: null pointer");
    if (n == 0) return CG_OK;
    EW_LAUNCH(act_bwd_kernel, n, dy, y, dz, n, act);
}
extern "C" int cg_act_fwd(const float* x, float* y, size_t n, int act, cg_stream_t stream) {
    CG_CHECK_ARG(x && y, "cg_act_fwd: null pointer");
    if (n == 0) return CG_OK;
    EW_LAUNCH(act_fwd_kernel, n, x, y, n, act);
}
// input pipeline tail: crop window + horizontal flip + ToTensor + Normalize, uint8 HWC -> fp32 NHWC.  One thread per
// output pixel (C <= 4 channels: 3 bytes in, 12 bytes out); the divisions are IEEE fp32, as torch's div / sub / div.
__global__ __launch_bounds__(256) void u8_to_f32_kernel(const uint8_t* __restrict__ src, int N, int Hs, int Ws, int C,
                                                        const int32_t* __restrict__ crop_tl, const uint8_t* __restrict__ flip,
                                                        int H, int W, float mean, float std, float* __restrict__ dst) {
    const size_t total = (size_t)N * H * W;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(p % W), y = (int)((p / W) % H), n = (int)(p / ((size_t)W * H));
        const int top = crop_tl ? crop_tl[2 * n] : 0, left = crop_tl ? crop_tl[2 * n + 1] : 0;
        const int sx = left + ((flip && flip[n]) ? W - 1 - x : x);
        const uint8_t* s = src + (((size_t)n * Hs + top + y) * Ws + sx) * C;
        float* d = dst + p * C;
        for (int c = 0; c < C; ++c) d[c] = __fdiv_rn(__fdiv_rn((float)s[c], 255.0f) - mean, std);
    }
}
extern "C" int cg_u8_to_f32_nhwc(const uint8_t* src, int N, int Hs, int Ws, int C, const int32_t* crop_tl,
                                 const uint8_t* flip, int H, int W, float mean, float std, float* dst, cg_stream_t stream) {
    CG_CHECK_ARG(src && dst, "cg_u8_to_f32_nhwc: null pointer");
    CG_CHECK_ARG(N > 0 && C >= 1 && C <= 4 && H > 0 && W > 0 && H <= Hs && W <= Ws && std != 0.f,
                 "cg_u8_to_f32_nhwc: bad sizes (crop window larger than the image?)");
    const size_t n = (size_t)N * H * W;
    EW_LAUNCH(u8_to_f32_kernel, n, src, N, Hs, Ws, C, crop_tl, flip, H, W, mean, std, dst);
}
extern "C" int cg_fill(float* p, size_t n, float value, cg_stream_t stream) {
    CG_CHECK_ARG(p, "cg_fill: null pointer");
    if (n == 0) return CG_OK;
    EW_LAUNCH(fill_kernel, n, p, n, value);
}
// {hi, lo} planes -> fp32 (hi + lo, divided by the device-side scale when there is one): the way back for the rare consumer that
// reads fp32 from a tensor its producer emitted in split form only (layer widths the split-precision weight gradient does not take)
__global__ __launch_bounds__(256) void unsplit_f16_kernel(const _Float16* __restrict__ zs, size_t lo_elems, const float* __restrict__ scale_dev,
                                                          float* __restrict__ out, size_t n) {
    const float inv = scale_dev ? 1.f / scale_dev[0] : 1.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = ((float)zs[cg_il(i)] + (float)zs[lo_elems + cg_il(i)]) * inv;
}
extern "C" int cg_unsplit_f16(const void* z_split, size_t lo_elems, const float* scale_dev, float* out, size_t n, cg_stream_t stream) {
    CG_CHECK_ARG(z_split && out, "cg_unsplit_f16: null pointer");
    if (n == 0) return CG_OK;
    EW_LAUNCH(unsplit_f16_kernel, n, (const _Float16*)z_split, lo_elems, scale_dev, out, n);
}
extern "C" int cg_compose1x1_fwd(const cg_group* group, const float* W1, const float* b1, const float* W2, const float* b2, int C,
                                 float* out, int out_stride, cg_stream_t stream) {
    CG_CHECK_ARG(W1 && b1 && W2 && b2 && out && C > 0 && out_stride > C, "cg_compose1x1_fwd: bad args");
    const int n = group ? group->n : 1;
    hipLaunchKernelGGL(compose1x1_fwd_kernel, dim3((C + 63) / 64 + 1, n), dim3(512), 0, cg_s(stream), W1, b1, W2, b2, C,
                       group ? (long long)group->stride : 0, out, out_stride);
    CG_LAUNCH_CHECK("compose1x1_fwd_kernel");
    return CG_OK;
}
extern "C" int cg_compose1x1_bwd(const cg_group* group, const float* d, int d_stride, const float* W1, const float* b1,
                                 const float* W2, int C, float* dW1, float* db1, float* dW2, float* db2, cg_stream_t stream) {
    CG_CHECK_ARG(d && W1 && b1 && W2 && dW1 && db1 && dW2 && db2 && C > 0 && d_stride > C, "cg_compose1x1_bwd: bad args");
    const int n = group ? group->n : 1;
    hipLaunchKernelGGL(compose1x1_bwd_kernel, dim3(C, n), dim3(256), 0, cg_s(stream), d, d_stride, W1, b1, W2, C,
                       group ? (long long)group->stride : 0, dW1, db1, dW2, db2);
    CG_LAUNCH_CHECK("compose1x1_bwd_kernel");
    return CG_OK;
}
extern "C" int cg_add(const float* a, const float* b, float* out, size_t n, cg_stream_t stream) {
    CG_CHECK_ARG(a && b && out, "cg_add: null pointer");
    if (n == 0) return CG_OK;
    EW_LAUNCH(add_kernel, n, a, b, out, n);
}
extern "C" int cg_axpby(float alpha, const float* a, float beta, float* b, size_t n, cg_stream_t stream) {
    CG_CHECK_ARG(a && b, "cg_axpby: null pointer");
    if (n == 0) return CG_OK;
    EW_LAUNCH(axpby_kernel, n, alpha, a, beta, b, n);
}

static inline int pool_out(int h) { return (h - 1) / 2 + 1; }

extern "C" int cg_avgpool3s2_fwd(const float* x, float* y, int N, int H, int W, int C, cg_stream_t stream) {
    CG_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0, "cg_avgpool3s2_fwd: bad args");
    const int Ho = pool_out(H), Wo = pool_out(W);
    EW_LAUNCH(avgpool_fwd_kernel, (size_t)N * Ho * Wo * C, x, y, N, H, W, C, Ho, Wo);
}
extern "C" int cg_avgpool3s2_bwd(const float* dy, float* dx, int N, int H, int W, int C, cg_stream_t stream) {
    CG_CHECK_ARG(dy && dx && N > 0 && H > 0 && W > 0 && C > 0, "cg_avgpool3s2_bwd: bad args");
    const int Ho = pool_out(H), Wo = pool_out(W);
    EW_LAUNCH(avgpool_bwd_kernel, (size_t)N * H * W * C, dy, dx, N, H, W, C, Ho, Wo);
}
extern "C" int cg_upsample2x_fwd(const float* x, float* y, int N, int H, int W, int C, cg_stream_t stream) {
    CG_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0, "cg_upsample2x_fwd: bad args");
    EW_LAUNCH(upsample_fwd_kernel, (size_t)N * 4 * H * W * C, x, y, N, H, W, C);
}
extern "C" int cg_upsample2x_bwd(const float* dy, float* dx, int N, int H, int W, int C, cg_stream_t stream) {
    CG_CHECK_ARG(dy && dx && N > 0 && H > 0 && W > 0 && C > 0, "cg_upsample2x_bwd: bad args");
    EW_LAUNCH(upsample_bwd_kernel, (size_t)N * H * W * C, dy, dx, N, H, W, C);
}
extern "C" int cg_global_avgpool_fwd(const float* x, float* y, int N, int HW, int C, cg_stream_t stream) {
    CG_CHECK_ARG(x && y && N > 0 && HW > 0 && C > 0, "cg_global_avgpool_fwd: bad args");
    hipLaunchKernelGGL(gap_kernel, dim3(cg_div_up(C, 64), N), dim3(256), 0, cg_s(stream), x, y, HW, C);
    CG_LAUNCH_CHECK("gap_kernel");
    return CG_OK;
}

#define MB_DISPATCH(KERNEL, ...)                                                                         \
    do {                                                                                                 \
        const unsigned grid__ = ew_grid(npix);                                                           \
        if (od == 3 && k == 3) hipLaunchKernelGGL((KERNEL<3, 3>), dim3(grid__), dim3(256), 0, cg_s(stream), __VA_ARGS__); \
        else if (od == 3 && k == 1) hipLaunchKernelGGL((KERNEL<3, 1>), dim3(grid__), dim3(256), 0, cg_s(stream), __VA_ARGS__); \
        else if (od == 3 && k == 2) hipLaunchKernelGGL((KERNEL<3, 2>), dim3(grid__), dim3(256), 0, cg_s(stream), __VA_ARGS__); \
        else if (od == 3 && k == 4) hipLaunchKernelGGL((KERNEL<3, 4>), dim3(grid__), dim3(256), 0, cg_s(stream), __VA_ARGS__); \
        else if (od == 1 && k == 1) hipLaunchKernelGGL((KERNEL<1, 1>), dim3(grid__), dim3(256), 0, cg_s(stream), __VA_ARGS__); \
        else if (od == 1 && k == 2) hipLaunchKernelGGL((KERNEL<1, 2>), dim3(grid__), dim3(256), 0, cg_s(stream), __VA_ARGS__); \
        else if (od == 1 && k == 3) hipLaunchKernelGGL((KERNEL<1, 3>), dim3(grid__), dim3(256), 0, cg_s(stream), __VA_ARGS__); \
        else if (od == 1 && k == 4) hipLaunchKernelGGL((KERNEL<1, 4>), dim3(grid__), dim3(256), 0, cg_s(stream), __VA_ARGS__); \
        else return cg_set_error(CG_ERR_ARG, "mask_blend: od must be 1 or 3 and k in 1..4 (got od=%d k=%d)", od, k); \
    } while (0)

extern "C" int cg_mask_blend_fwd(const float* new_x, const float* im_in, float* im_out, float* mask, size_t npix,
                                 int od, int k, cg_stream_t stream) {
    CG_CHECK_ARG(new_x && im_in && im_out && mask, "cg_mask_blend_fwd: null pointer");
    if (npix == 0) return CG_OK;
    MB_DISPATCH(mask_blend_fwd_kernel, new_x, im_in, im_out, mask, npix);
    CG_LAUNCH_CHECK("mask_blend_fwd_kernel");
    return CG_OK;
}
extern "C" int cg_mask_blend_bwd(const float* new_x, const float* im_in, const float* d_im_out, const float* d_mask,
                                 float* d_new_x, size_t npix, int od, int k, cg_stream_t stream) {
    CG_CHECK_ARG(new_x && im_in && d_im_out && d_new_x, "cg_mask_blend_bwd: null pointer");
    if (npix == 0) return CG_OK;
    MB_DISPATCH(mask_blend_bwd_kernel, new_x, im_in, d_im_out, d_mask, d_new_x, npix);
    CG_LAUNCH_CHECK("mask_blend_bwd_kernel");
    return CG_OK;
}

// grouped forms: nb = samples of ALL members (member m owns samples [m*nb/nmember, (m+1)*nb/nmember)); loss / gscale have
// one entry per member.  A member's value does not depend on the other members in the launch.
extern "C" int cg_lsgan_fwd_g(const float* out, const float* tgt, const float* wt, int nb, int hw, int group, int nmember,
                              float* loss, int accumulate, cg_stream_t stream) {
    CG_CHECK_ARG(out && tgt && wt && loss && nb > 0 && hw > 0 && group > 0 && nmember > 0 && nb % nmember == 0,
                 "cg_lsgan_fwd: bad args");
    hipLaunchKernelGGL(lsgan_fwd_kernel, dim3(nmember), dim3(256), 0, cg_s(stream), out, tgt, wt, nb / nmember, hw, group, loss,
                       accumulate);
    CG_LAUNCH_CHECK("lsgan_fwd_kernel");
    return CG_OK;
}
extern "C" int cg_lsgan_bwd_g(const float* out, const float* tgt, const float* wt, const float* gscale, int nb, int hw,
                              int group, int nmember, float* d_out, cg_stream_t stream) {
    CG_CHECK_ARG(out && tgt && wt && gscale && d_out && nb > 0 && hw > 0 && group > 0 && nmember > 0 && nb % nmember == 0,
                 "cg_lsgan_bwd: bad args");
    EW_LAUNCH(lsgan_bwd_kernel, (size_t)nb * hw, out, tgt, wt, gscale, nb / nmember, hw, group, d_out, nmember);
}
extern "C" int cg_lsgan_fwd(const float* out, const float* tgt, const float* wt, int nb, int hw, int group,
                            float* loss, int accumulate, cg_stream_t stream) {
    return cg_lsgan_fwd_g(out, tgt, wt, nb, hw, group, 1, loss, accumulate, stream);
}
extern "C" int cg_lsgan_bwd(const float* out, const float* tgt, const float* wt, const float* gscale, int nb, int hw,
                            int group, float* d_out, cg_stream_t stream) {
    return cg_lsgan_bwd_g(out, tgt, wt, gscale, nb, hw, group, 1, d_out, stream);
}

extern "C" size_t cg_focus_workspace(void) { return (size_t)FOCUS_BLOCKS * 3 * sizeof(double); }     // per member
// grouped forms: N = samples of ALL members; sums [nmember][3], out [nmember][4], gscale [nmember]; ws: nmember * cg_focus_workspace()
extern "C" int cg_focus_sums_g(const float* mask, int N, int H, int W, int k, int nmember, float center, float eps,
                               float* sums, void* ws, size_t ws_bytes, cg_stream_t stream) {
    CG_CHECK_ARG(mask && sums && N > 0 && H > 0 && W > 0 && k > 0 && nmember > 0 && N % nmember == 0, "cg_focus_sums: bad args");
    if (!ws || ws_bytes < (size_t)nmember * cg_focus_workspace())
        return cg_set_error(CG_ERR_WORKSPACE, "cg_focus_sums: workspace too small");
    const size_t total = (size_t)(N / nmember) * H * W * k;
    size_t nb = (total + 255) / 256;
    if (nb > FOCUS_BLOCKS) nb = FOCUS_BLOCKS;
    hipLaunchKernelGGL(focus_partial_kernel, dim3((unsigned)nb, nmember), dim3(256), 0, cg_s(stream), mask, H, W, k, total,
                       center, eps, (double*)ws);
    CG_LAUNCH_CHECK("focus_partial_kernel");
    hipLaunchKernelGGL(focus_final_kernel, dim3(nmember), dim3(64), 0, cg_s(stream), (const double*)ws, (int)nb, sums);
    CG_LAUNCH_CHECK("focus_final_kernel");
    return CG_OK;
}
extern "C" int cg_focus_total_g(const float* sums, size_t numel_per_member, int nmember, float w_zo, float w_total,
                                float w_tv, int use_abs, int use_square, float* out, cg_stream_t stream) {
    CG_CHECK_ARG(sums && out && numel_per_member > 0 && nmember > 0, "cg_focus_total: bad args");
    hipLaunchKernelGGL(focus_total_kernel, dim3(nmember), dim3(64), 0, cg_s(stream), sums, (float)numel_per_member, w_zo,
                       w_total, w_tv, use_abs, use_square, out);
    CG_LAUNCH_CHECK("focus_total_kernel");
    return CG_OK;
}
extern "C" int cg_focus_bwd_g(const float* mask, const float* sums, const float* gscale, int N, int H, int W, int k,
                              int nmember, float center, float eps, float w_zo, float w_total, float w_tv, int use_abs,
                              int use_square, float* d_mask, cg_stream_t stream) {
    CG_CHECK_ARG(mask && sums && gscale && d_mask && N > 0 && H > 0 && W > 0 && k > 0 && nmember > 0 && N % nmember == 0,
                 "cg_focus_bwd: bad args");
    const int Nm = N / nmember;
    hipLaunchKernelGGL(focus_bwd_kernel, dim3(ew_grid((size_t)Nm * H * W * k), nmember), dim3(256), 0, cg_s(stream), mask, sums,
                       gscale, Nm, H, W, k, center, eps, w_zo, w_total, w_tv, use_abs, use_square, d_mask);
    CG_LAUNCH_CHECK("focus_bwd_kernel");
    return CG_OK;
}
extern "C" int cg_focus_sums(const float* mask, int N, int H, int W, int k, float center, float eps, float* sums,
                             void* ws, size_t ws_bytes, cg_stream_t stream) {
    return cg_focus_sums_g(mask, N, H, W, k, 1, center, eps, sums, ws, ws_bytes, stream);
}
extern "C" int cg_focus_total(const float* sums, size_t numel, float w_zo, float w_total, float w_tv, int use_abs,
                              int use_square, float* out, cg_stream_t stream) {
    return cg_focus_total_g(sums, numel, 1, w_zo, w_total, w_tv, use_abs, use_square, out, stream);
}
extern "C" int cg_focus_bwd(const float* mask, const float* sums, const float* gscale, int N, int H, int W, int k,
                            float center, float eps, float w_zo, float w_total, float w_tv, int use_abs,
                            int use_square, float* d_mask, cg_stream_t stream) {
    return cg_focus_bwd_g(mask, sums, gscale, N, H, W, k, 1, center, eps, w_zo, w_total, w_tv, use_abs, use_square, d_mask, stream);
}

extern "C" int cg_l1_mean_fwd(const float* a, const float* b, size_t n, float* loss, cg_stream_t stream) {
    CG_CHECK_ARG(a && b && loss && n > 0, "cg_l1_mean_fwd: bad args");
    hipLaunchKernelGGL(l1_mean_fwd_kernel, dim3(1), dim3(1024), 0, cg_s(stream), a, b, n, loss);
    CG_LAUNCH_CHECK("l1_mean_fwd_kernel");
    return CG_OK;
}
extern "C" int cg_l1_mean_bwd(const float* a, const float* b, const float* gscale, size_t n, float* da,
                              cg_stream_t stream) {
    CG_CHECK_ARG(a && b && gscale && da && n > 0, "cg_l1_mean_bwd: bad args");
    EW_LAUNCH(l1_mean_bwd_kernel, n, a, b, gscale, n, da);
}

// grouped: the same run [p, p+n) of `nmember` members whose pool slices sit `mstride` elements apart (one launch)
extern "C" int cg_adam_step_g(float* p, const float* g, float* m, float* v, size_t n, int nmember, long long mstride,
                              float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                              cg_stream_t stream) {
    CG_CHECK_ARG(p && g && m && v && step >= 1 && nmember >= 1 && (nmember == 1 || mstride > 0), "cg_adam_step: bad args");
    if (n == 0) return CG_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n), nmember), dim3(256), 0, cg_s(stream), p, g, m, v, n, beta1, beta2, eps,
                       weight_decay, step_size, bc2_sqrt, mstride, (const float*)nullptr);
    CG_LAUNCH_CHECK("adam_kernel");
    return CG_OK;
}
// The two per-step scalars of cg_adam_step_g as the HOST computes them (double precision, rounded once): out[0] = lr /
// (1 - beta1^step), out[1] = sqrt(1 - beta2^step).  cg_adam_step_dev with these two floats in device memory is bit-identical
// to cg_adam_step_g(..., lr, ..., step).
extern "C" int cg_adam_hyper(float lr, float beta1, float beta2, int step, float* out2_host) {
    CG_CHECK_ARG(out2_host && step >= 1, "cg_adam_hyper: bad args");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    out2_host[0] = (float)((double)lr / bc1);
    out2_host[1] = (float)sqrt(bc2);
    return CG_OK;
}
extern "C" int cg_adam_step_dev(float* p, const float* g, float* m, float* v, size_t n, int nmember, long long mstride,
                                float beta1, float beta2, float eps, float weight_decay, const float* hyper_dev,
                                cg_stream_t stream) {
    CG_CHECK_ARG(p && g && m && v && hyper_dev && nmember >= 1 && (nmember == 1 || mstride > 0), "cg_adam_step_dev: bad args");
    if (n == 0) return CG_OK;
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n), nmember), dim3(256), 0, cg_s(stream), p, g, m, v, n, beta1, beta2, eps,
                       weight_decay, 0.f, 1.f, mstride, hyper_dev);
    CG_LAUNCH_CHECK("adam_kernel");
    return CG_OK;
}
extern "C" int cg_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                            float beta2, float eps, float weight_decay, int step, cg_stream_t stream) {
    return cg_adam_step_g(p, g, m, v, n, 1, 0, lr, beta1, beta2, eps, weight_decay, step, stream);
}

extern "C" int cg_gather_rows(const float* src, const int32_t* idx_dev, float* out, int nidx, size_t row_elems,
                              cg_stream_t stream) {
    CG_CHECK_ARG(src && idx_dev && out && nidx > 0 && row_elems > 0, "cg_gather_rows: bad args");
    EW_LAUNCH(gather_rows_kernel, (size_t)nidx * row_elems, src, idx_dev, out, nidx, row_elems);
}
extern "C" int cg_gather_rows2(const float* src_a, const float* src_b, const int32_t* idx_dev, float* out, int nidx,
                               size_t row_elems, cg_stream_t stream) {
    CG_CHECK_ARG(src_a && idx_dev && out && nidx > 0 && row_elems > 0 && row_elems % 4 == 0, "cg_gather_rows2: bad args");
    EW_LAUNCH(gather_rows2_kernel, (size_t)nidx * (row_elems / 4), src_a, src_b ? src_b : src_a, idx_dev, out, nidx, row_elems / 4);
}
extern "C" int cg_gen_total(const float* focus, const float* adv, const float* lc, const float* w_match, float gan_w,
                            float council_w, float* total, float* council, float* gcouncil, int nmember,
                            cg_stream_t stream) {
    CG_CHECK_ARG(total && nmember > 0, "cg_gen_total: bad args");
    hipLaunchKernelGGL(gen_total_kernel, dim3(cg_div_up(nmember, 64)), dim3(64), 0, cg_s(stream), focus, adv, lc, w_match, gan_w,
                       council_w, total, council, gcouncil, nmember);
    CG_LAUNCH_CHECK("gen_total_kernel");
    return CG_OK;
}

// grouped rings: [nmember][n], one value / weight per member
extern "C" int cg_ring_push_g(float* ring, int n, int pos, const float* value, int nmember, cg_stream_t stream) {
    CG_CHECK_ARG(ring && value && n > 0 && pos >= 0 && nmember > 0, "cg_ring_push: bad args");
    hipLaunchKernelGGL(ring_kernel, dim3(nmember), dim3(64), 0, cg_s(stream), ring, (float*)nullptr, n, pos, value,
                       (float*)nullptr, 0, (const int32_t*)nullptr);
    CG_LAUNCH_CHECK("ring_kernel");
    return CG_OK;
}
// the same two operations with the write position read from device memory (pos_dev[0] >= 0)
extern "C" int cg_ring_push_dev(float* ring, int n, const int32_t* pos_dev, const float* value, int nmember, cg_stream_t stream) {
    CG_CHECK_ARG(ring && value && pos_dev && n > 0 && nmember > 0, "cg_ring_push_dev: bad args");
    hipLaunchKernelGGL(ring_kernel, dim3(nmember), dim3(64), 0, cg_s(stream), ring, (float*)nullptr, n, 0, value,
                       (float*)nullptr, 0, pos_dev);
    CG_LAUNCH_CHECK("ring_kernel");
    return CG_OK;
}
extern "C" int cg_loss_match_dev(float* ring_gan, float* ring_council, int n, const int32_t* pos_dev, const float* council_loss,
                                 float* w_out, int nmember, cg_stream_t stream) {
    CG_CHECK_ARG(ring_gan && ring_council && council_loss && w_out && pos_dev && n > 0 && nmember > 0, "cg_loss_match_dev: bad args");
    hipLaunchKernelGGL(ring_kernel, dim3(nmember), dim3(64), 0, cg_s(stream), ring_gan, ring_council, n, 0, council_loss,
                       w_out, 1, pos_dev);
    CG_LAUNCH_CHECK("ring_kernel");
    return CG_OK;
}
extern "C" int cg_loss_match_g(float* ring_gan, float* ring_council, int n, int pos, const float* council_loss,
                               float* w_out, int nmember, cg_stream_t stream) {
    CG_CHECK_ARG(ring_gan && ring_council && council_loss && w_out && n > 0 && pos >= 0 && nmember > 0, "cg_loss_match: bad args");
    hipLaunchKernelGGL(ring_kernel, dim3(nmember), dim3(64), 0, cg_s(stream), ring_gan, ring_council, n, pos, council_loss,
                       w_out, 1, (const int32_t*)nullptr);
    CG_LAUNCH_CHECK("ring_kernel");
    return CG_OK;
}
extern "C" int cg_ring_push(float* ring, int n, int pos, const float* value, cg_stream_t stream) {
    return cg_ring_push_g(ring, n, pos, value, 1, stream);
}
extern "C" int cg_loss_match(float* ring_gan, float* ring_council, int n, int pos, const float* council_loss,
                             float* w_out, cg_stream_t stream) {
    return cg_loss_match_g(ring_gan, ring_council, n, pos, council_loss, w_out, 1, stream);
}

// The hybrid product on hardware, end to end on a small GEMM (next-round groundwork, DESIGN.md section 8 "Open" item 1):
//     a.b ~ ah.bh  [fp16 MFMA]  +  Q(ah).Q(bl) + Q(al).Q(bh)  [block-scaled fp6 MFMA, one e8m0 scale per 32 K elements]
// C[M][N] = sum_k A[m][k] * B[n][k] with M = N = 64, K = 512; operands are split on the DEVICE (prep kernel: fp16 hi plane,
// packed e2m3 planes of hi and of the residual, their block exponents), the GEMM is one wave per 32x32 tile straight from
// global memory (no LDS: this checks arithmetic and layouts, not speed).  Checked against (1) the same quantisation
// evaluated on the host in double -- agreement to fp32 accumulation round-off pins packing, scale semantics and the cross-term
// assembly -- and (2) the exact double product: the error the datapath costs (CPU emulation says 1.2e-5).
// Build + run:  hipcc --offload-arch=gfx950 -O2 tools/probe/x3_mx6_gemm.hip -o /tmp/x3mx6 && /tmp/x3mx6
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef int i8v __attribute__((ext_vector_type(8)));

constexpr int M = 64, N = 64, K = 512, KB = K / 32;

// e2m3: sign(1) exponent(2, bias 1) mantissa(3); max 7.5; sub-normal step 1/8
__host__ __device__ inline unsigned enc_e2m3(float x) {
    const unsigned sign = x < 0.f ? 32u : 0u;
    float a = fminf(fabsf(x), 7.5f);
    unsigned code;
    if (a < 1.f) {
        code = (unsigned)rintf(a * 8.f);             // 0..8: 8 carries into exponent 1, mantissa 0
    } else {
        int e = a >= 4.f ? 2 : (a >= 2.f ? 1 : 0);
        const float step = e == 2 ? 0.5f : (e == 1 ? 0.25f : 0.125f);
        int m = (int)rintf(a / step);                // 8..16
        if (m == 16) { m = 8; ++e; }
        code = ((unsigned)(e + 1) << 3) | (unsigned)(m - 8);
    }
    return sign | code;
}
static double dec_e2m3(unsigned c) {
    const double s = (c & 32u) ? -1.0 : 1.0;
    const unsigned e = (c >> 3) & 3u, m = c & 7u;
    return s * (e == 0 ? m * 0.125 : (1.0 + m / 8.0) * ldexp(1.0, (int)e - 1));
}

// one thread per (row, 32-block): fp16 hi plane, packed fp6 of hi and of the residual, block exponents (biased e8m0 bytes)
__global__ void prep(const float* x, float scale, _Float16* hi, uint8_t* q_hi, uint8_t* e_hi, uint8_t* q_lo, uint8_t* e_lo, int rows) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * KB) return;
    const float* src = x + (size_t)t * 32;
    float h[32], r[32], mh = 0.f, mr = 0.f;
    for (int j = 0; j < 32; ++j) {
        const float v = src[j] * scale;
        const _Float16 f = (_Float16)v;
        hi[(size_t)t * 32 + j] = f;
        h[j] = (float)f;
        r[j] = v - h[j];
        mh = fmaxf(mh, fabsf(h[j]));
        mr = fmaxf(mr, fabsf(r[j]));
    }
    for (int pl = 0; pl < 2; ++pl) {
        const float mx = pl ? mr : mh;
        const int e = mx > 0.f ? (int)floorf(log2f(mx)) - 2 : -127;
        const float inv = ldexpf(1.f, -e);
        uint8_t bytes[24];
        for (int j = 0; j < 24; ++j) bytes[j] = 0;
        for (int j = 0; j < 32; ++j) {
            const unsigned c = enc_e2m3((pl ? r[j] : h[j]) * inv), bit = 6 * j;
            bytes[bit >> 3] |= (uint8_t)(c << (bit & 7));
            if ((bit & 7) > 2) bytes[(bit >> 3) + 1] |= (uint8_t)(c >> (8 - (bit & 7)));
        }
        uint8_t* q = (pl ? q_lo : q_hi) + (size_t)t * 24;
        for (int j = 0; j < 24; ++j) q[j] = bytes[j];
        (pl ? e_lo : e_hi)[t] = (uint8_t)(e + 127);
    }
}

__device__ inline i8v load6(const uint8_t* p) {
    const int* w = (const int*)p;
    i8v v;
    v[0] = w[0]; v[1] = w[1]; v[2] = w[2]; v[3] = w[3]; v[4] = w[4]; v[5] = w[5]; v[6] = 0; v[7] = 0;
    return v;
}

template <int CROSS>
__global__ void gemm(const _Float16* a_hi, const uint8_t* a_qh, const uint8_t* a_eh, const uint8_t* a_ql, const uint8_t* a_el,
                     const _Float16* b_hi, const uint8_t* b_qh, const uint8_t* b_eh, const uint8_t* b_ql, const uint8_t* b_el,
                     float inv_scale, float* c) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = (wave >> 1) * 32, n0 = (wave & 1) * 32;
    const int row = m0 + (lane & 31), col = n0 + (lane & 31), half = lane >> 5;
    f16v acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int s = 0; s < K / 64; ++s) {
        if (CROSS) {
            const int blk = 2 * s + half;
            const i8v qah = load6(a_qh + ((size_t)row * KB + blk) * 24), qal = load6(a_ql + ((size_t)row * KB + blk) * 24);
            const i8v qbh = load6(b_qh + ((size_t)col * KB + blk) * 24), qbl = load6(b_ql + ((size_t)col * KB + blk) * 24);
            const int eah = a_eh[row * KB + blk], eal = a_el[row * KB + blk], ebh = b_eh[col * KB + blk], ebl = b_el[col * KB + blk];
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qal, qbh, acc, 2, 2, 0, eal, 0, ebh);     // small terms first
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qah, qbl, acc, 2, 2, 0, eah, 0, ebl);
        }
        for (int ks = 0; ks < 4; ++ks) {
            const int k = 64 * s + 16 * ks + 8 * half;
            const h8 a = *(const h8*)(a_hi + (size_t)row * K + k), b = *(const h8*)(b_hi + (size_t)col * K + k);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
    }
    for (int r = 0; r < 16; ++r) {
        const int i = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        c[(size_t)i * N + col] = acc[r] * inv_scale;
    }
}

struct Planes {
    _Float16* hi; uint8_t *qh, *eh, *ql, *el;
    void alloc(int rows) {
        hipMalloc(&hi, (size_t)rows * K * 2); hipMalloc(&qh, (size_t)rows * KB * 24); hipMalloc(&ql, (size_t)rows * KB * 24);
        hipMalloc(&eh, (size_t)rows * KB); hipMalloc(&el, (size_t)rows * KB);
    }
};

static float pow2_scale(const float* x, size_t n) {
    float m = 0.f;
    for (size_t i = 0; i < n; ++i) m = fmaxf(m, fabsf(x[i]));
    return ldexpf(1.f, (int)floorf(log2f(8192.f / m)));
}

int main() {
    static float A[M * K], B[N * K];
    srand(3);
    for (int i = 0; i < M * K; ++i) { const float u = rand() / (float)RAND_MAX - 0.5f; A[i] = u * u * u * 40.f + (i % 7 == 0 ? 0.f : u); }
    for (int i = 0; i < N * K; ++i) { const float u = rand() / (float)RAND_MAX - 0.5f; B[i] = 0.05f * u * (1.f + (i % 13)); }
    const float sA = pow2_scale(A, M * K), sB = pow2_scale(B, N * K);
    float *dA, *dB, *dC;
    hipMalloc(&dA, sizeof(A)); hipMalloc(&dB, sizeof(B)); hipMalloc(&dC, M * N * 4);
    hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice);
    hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
    Planes pa, pb;
    pa.alloc(M); pb.alloc(N);
    hipLaunchKernelGGL(prep, dim3((M * KB + 63) / 64), dim3(64), 0, 0, dA, sA, pa.hi, pa.qh, pa.eh, pa.ql, pa.el, M);
    hipLaunchKernelGGL(prep, dim3((N * KB + 63) / 64), dim3(64), 0, 0, dB, sB, pb.hi, pb.qh, pb.eh, pb.ql, pb.el, N);

    // host copies of the planes for the emulation
    static _Float16 ahi[M * K], bhi[N * K];
    static uint8_t aqh[M * KB * 24], aql[M * KB * 24], bqh[N * KB * 24], bql[N * KB * 24], aeh[M * KB], ael[M * KB], beh[N * KB], bel[N * KB];
    hipMemcpy(ahi, pa.hi, sizeof(ahi), hipMemcpyDeviceToHost); hipMemcpy(bhi, pb.hi, sizeof(bhi), hipMemcpyDeviceToHost);
    hipMemcpy(aqh, pa.qh, sizeof(aqh), hipMemcpyDeviceToHost); hipMemcpy(aql, pa.ql, sizeof(aql), hipMemcpyDeviceToHost);
    hipMemcpy(bqh, pb.qh, sizeof(bqh), hipMemcpyDeviceToHost); hipMemcpy(bql, pb.ql, sizeof(bql), hipMemcpyDeviceToHost);
    hipMemcpy(aeh, pa.eh, sizeof(aeh), hipMemcpyDeviceToHost); hipMemcpy(ael, pa.el, sizeof(ael), hipMemcpyDeviceToHost);
    hipMemcpy(beh, pb.eh, sizeof(beh), hipMemcpyDeviceToHost); hipMemcpy(bel, pb.el, sizeof(bel), hipMemcpyDeviceToHost);
    auto q = [](const uint8_t* p, int row, int k) {
        const uint8_t* b = p + ((size_t)row * KB + k / 32) * 24;
        const unsigned bit = 6 * (k % 32);
        unsigned v = b[bit >> 3] >> (bit & 7);
        if ((bit & 7) > 2) v |= (unsigned)b[(bit >> 3) + 1] << (8 - (bit & 7));
        return dec_e2m3(v & 63u);
    };
    static double exact[M * N], emu1[M * N], emu3[M * N];
    double cmax = 0;
    for (int i = 0; i < M; ++i)
        for (int n = 0; n < N; ++n) {
            double ex = 0, main_ = 0, cross = 0;
            for (int k = 0; k < K; ++k) {
                ex += (double)A[i * K + k] * B[n * K + k];
                main_ += (double)(float)ahi[i * K + k] * (double)(float)bhi[n * K + k];
                const int kb = k / 32;
                cross += q(aqh, i, k) * ldexp(1.0, aeh[i * KB + kb] - 127) * q(bql, n, k) * ldexp(1.0, bel[n * KB + kb] - 127)
                       + q(aql, i, k) * ldexp(1.0, ael[i * KB + kb] - 127) * q(bqh, n, k) * ldexp(1.0, beh[n * KB + kb] - 127);
            }
            exact[i * N + n] = ex;
            emu1[i * N + n] = main_ / ((double)sA * sB);
            emu3[i * N + n] = (main_ + cross) / ((double)sA * sB);
            cmax = fmax(cmax, fabs(ex));
        }
    static float C[M * N];
    for (int cross = 0; cross < 2; ++cross) {
        if (cross)
            hipLaunchKernelGGL(gemm<1>, dim3(1), dim3(256), 0, 0, pa.hi, pa.qh, pa.eh, pa.ql, pa.el, pb.hi, pb.qh, pb.eh, pb.ql, pb.el, 1.f / (sA * sB), dC);
        else
            hipLaunchKernelGGL(gemm<0>, dim3(1), dim3(256), 0, 0, pa.hi, pa.qh, pa.eh, pa.ql, pa.el, pb.hi, pb.qh, pb.eh, pb.ql, pb.el, 1.f / (sA * sB), dC);
        hipMemcpy(C, dC, sizeof(C), hipMemcpyDeviceToHost);
        double d_emu = 0, d_ex = 0, rms = 0, rr = 0;
        for (int i = 0; i < M * N; ++i) {
            d_emu = fmax(d_emu, fabs(C[i] - (cross ? emu3 : emu1)[i]));
            d_ex = fmax(d_ex, fabs(C[i] - exact[i]));
            rms += (C[i] - exact[i]) * (C[i] - exact[i]);
            rr += exact[i] * exact[i];
        }
        printf("%-34s vs host evaluation of the same operands: max %.2e of max|C|;   vs exact: max %.2e, rms-rel %.2e   (%s)\n",
               cross ? "fp16 main + MX-fp6 cross terms" : "fp16 main term only", d_emu / cmax, d_ex / cmax, sqrt(rms / rr),
               hipGetErrorString(hipGetLastError()));
    }
    return 0;
}

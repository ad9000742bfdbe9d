// Implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Replaces nn.Conv2d-after-ZeroPad2d (reference networks.py:513,515-516), the bare 1x1 convs
// (:44,142-143,348) and nn.Linear (:531, as a 1x1 conv on an Nx1x1xC tensor): forward,
// data-gradient (same kernel, transposed weights + a different tap table) and weight-gradient.
//
// fp32 MFMA is the only datapath that meets the 1e-3 parity target; it is
// bit-for-bit an fmaf chain.  One MFMA = 32x32x2, 64 cycles per SIMD, so the matrix pipe is
// 16x slower than bf16 and LDS/global feeding is cheap in comparison: a 128x128 block tile with
// BK = 32, register-staged prefetch of the next K-slice and ds_read_b128 operand fetches keeps
// the pipe busy.
//
// GEMM view (forward / dgrad):  D[m][j] = sum_k A[m][k] * Wp[j][k]
//   m = (n, oy, ox) output position, j = output channel, k = (tap, c) with c fastest.
//   A is gathered on the fly from NHWC activations (zero padding, optional nearest-2x upsample,
//   optional two-source channel concat); Wp is [Cout][T][Ct], k-contiguous.
// GEMM view (wgrad):  dW[j][k] = sum_m dz[m][j] * A[m][k]   (split over m, deterministic reduce)
#include <stddef.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>
#include "cg_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;       // K-slice per LDS stage
constexpr int LDK = BK + 4;  // 36-float rows: 16-B aligned, conflict-free for ds_read_b128 (16 rows x 4 banks)

struct RowInfo {
    int base;     // n*H*W (pixel index of the sample), -1 = row beyond M
    int ly0;      // oy*stride
    int lx0;      // ox*stride
    int out_off;  // element offset of the output pixel (forward) / of the dz row (wgrad)
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8).  Give each XCD a
    // contiguous chunk of tile indices so the n-tiles of one m-tile (same gathered activations)
    // share an L2.  Bijective for any nwg.  Speed only -- never correctness.
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (bid >> 3);
}

// The geometry struct is the FIRST kernel argument: read its tap table with a per-lane index straight
// from the kernarg segment (constant address space).  Indexing the by-value struct dynamically
// would make the compiler spill it to scratch.
__device__ __forceinline__ int load_tap(int t, int geom_offset = 0) {
    const int8_t* ka = (const int8_t*)__builtin_amdgcn_kernarg_segment_ptr() + geom_offset;
    const int dy = ka[offsetof(cg_conv_geom, dy) + t];
    const int dx = ka[offsetof(cg_conv_geom, dx) + t];
    return (dy & 0xffff) | (dx << 16);
}

// the scalar head of cg_conv_geom (same layout, without the tap tables)
struct GeomS {
    int32_t N, H, W, C1, C2, up, Ho, Wo, HoF, WoF, osy, osx, ooy, oox, stride, T, Cout, act;
};
static_assert(sizeof(GeomS) == offsetof(cg_conv_geom, dy), "GeomS mirrors the head of cg_conv_geom");

// One launch of the pipelined kernel covers up to four geometry "classes" (blockIdx.y): the four output-parity
// passes of a stride-2 data-gradient run as ONE grid.  A forward convolution is a batch of one.
struct PipeClass {
    cg_conv_geom g;
    const float* w;
    int32_t M, K, tiles_n, ntiles;     // M = output rows of ONE member, ntiles = tiles of one member (grid.x)
    uint32_t w_bytes;
    int32_t pad_;
};
struct PipeBatch {
    PipeClass c[4];
};

// Member-batched ("grouped") launches: the council members' SAME layer runs as one grid.  Activations of the members
// are sample blocks of one batched NHWC tensor (member z owns samples [z*N/n, (z+1)*N/n), i.e. output rows
// [z*M, (z+1)*M)); their parameters sit at a uniform stride in one pool (optim.py), so member z reads its weights /
// bias `z * stride` further on.  blockIdx.z = member for forward / data-gradient launches, blockIdx.y for
// weight-gradient launches.  n = 1 is an ordinary launch.
struct Members {
    int32_t n;
    int32_t pad;
    long long w_stride;     // bytes between consecutive members' weights AS THE KERNEL READS THEM
    long long b_stride;     // bytes between consecutive members' biases (fp32)
};
inline Members one_member() { return Members{1, 0, 0, 0}; }

template <class G>
__device__ __forceinline__ RowInfo decode_row(const G& g, int m, int M, bool fwd_out) {
    const bool ok = m < M;
    const int mm = ok ? m : 0;
    const int hw = g.Ho * g.Wo;
    const int n = mm / hw;
    const int rem = mm - n * hw;
    const int oy = rem / g.Wo;
    const int ox = rem - oy * g.Wo;
    RowInfo ri;
    ri.base = ok ? n * g.H * g.W : -1;
    ri.ly0 = oy * g.stride;
    ri.lx0 = ox * g.stride;
    ri.out_off = fwd_out ? ((n * g.HoF + oy * g.osy + g.ooy) * g.WoF + ox * g.osx + g.oox) * g.Cout : mm * g.Cout;
    return ri;
}

// one gathered input element / float4 (zero outside the image = ZeroPad2d)
template <class G>
__device__ __forceinline__ bool tap_pixel(const G& g, const RowInfo& ri, int tap_dydx, int& pix) {
    int dy = (int)(short)(tap_dydx & 0xffff);
    int dx = tap_dydx >> 16;
    int ly = ri.ly0 + dy, lx = ri.lx0 + dx;
    bool ok = ri.base >= 0 && ly >= 0 && lx >= 0 && ly < (g.H << g.up) && lx < (g.W << g.up);
    pix = ri.base + (ly >> g.up) * g.W + (lx >> g.up);
    return ok;
}

// max |v| of a block's outputs -> state[2 + blockIdx.x] (the slot layout cg_split_f16_dynamic reduces again): lets the
// convolution that consumes this output in split form skip its own max-reduction pass over the tensor.  Launches with
// more than 1024 blocks fold into 1024 slots with an integer atomic max (non-negative floats order like their bit
// patterns); the launcher zeroes the slots first.
template <int NT>
__device__ __forceinline__ void block_amax_store(float vmax, float* __restrict__ state) {
    __shared__ float amax_red[NT / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    if ((threadIdx.x & 63) == 0) amax_red[threadIdx.x >> 6] = vmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = amax_red[0];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) m = fmaxf(m, amax_red[w]);
        // members ride on grid.z, the output-parity classes of a batched data-gradient launch on grid.y
        const unsigned nb = gridDim.x * gridDim.y * gridDim.z, b = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if (nb <= 1024) state[2 + b] = m;
        else atomicMax(reinterpret_cast<unsigned*>(state + 2 + (b & 1023)), __float_as_uint(m));
    }
}
// a block that has no tile (a class with fewer tiles than the launch's widest) still owns a slot: it reports zero
__device__ __forceinline__ void block_amax_idle(float* __restrict__ state) {
    if (threadIdx.x == 0) {
        const unsigned nb = gridDim.x * gridDim.y * gridDim.z, b = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if (nb <= 1024) state[2 + b] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// forward / data-gradient kernel
//   BM x BN block tile, WM x WN wave tile, (BM/WM)*(BN/WN) waves, STAGES LDS buffers.
//   STAGES = 1: load(k+1) -> compute(k) -> barrier -> store(k+1) -> barrier
//   STAGES = 2: store(k+1) -> load(k+2) -> compute(k) -> barrier        (one barrier per K-slice)
// ------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool FAST, int STAGES>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void conv_fwd_kernel(
    cg_conv_geom g, const float* __restrict__ x1, const float* __restrict__ x2, const float* __restrict__ w,
    const float* __restrict__ bias, float* __restrict__ y, int M, int K, int tiles_n, float* __restrict__ amax_state,
    Members mb) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    static_assert(NT == 256 || NT == 512 || NT == 1024, "4, 8 or 16 waves per block");
    __shared__ __attribute__((aligned(16))) float As[STAGES][BM * LDK];
    __shared__ __attribute__((aligned(16))) float Bs[STAGES][BN * LDK];
    __shared__ RowInfo rows[BM];
    __shared__ int taps[CG_MAX_TAPS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m_lo = (int)blockIdx.z * M;            // this member's rows: [m_lo, m_lo + M)
    const int m0 = m_lo + (tile / tiles_n) * BM;
    const int n0 = (tile % tiles_n) * BN;
    const int Ct = g.C1 + g.C2;
    w = reinterpret_cast<const float*>(reinterpret_cast<const char*>(w) + (long long)blockIdx.z * mb.w_stride);
    if (bias) bias = reinterpret_cast<const float*>(reinterpret_cast<const char*>(bias) + (long long)blockIdx.z * mb.b_stride);

    if (tid < g.T) taps[tid] = load_tap(tid);
    for (int r = tid; r < BM; r += NT) rows[r] = decode_row(g, m0 + r, m_lo + M, true);
    __syncthreads();

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging registers
    constexpr int RPV = NT / 8;                        // FAST: rows per pass (8 float4 per 32-float row)
    constexpr int A_V4 = BM / RPV, B_V4 = BN / RPV;    // float4 per thread
    constexpr int RPS = NT / 32;                       // generic: rows per pass
    constexpr int A_S = BM / RPS, B_S = BN / RPS;      // scalars per thread
    static_assert(BM % RPV == 0 && BN % RPV == 0, "tile rows must be a multiple of the loader rows");
    float4 av[FAST ? A_V4 : 1], bv[FAST ? B_V4 : 1];
    float as[FAST ? 1 : A_S], bs[FAST ? 1 : B_S];

    const int nk = (K + BK - 1) / BK;

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        if constexpr (FAST) {
            const int kc = tid & 7, r0 = tid >> 3;
            const int tap = k0 / Ct;
            const int c0 = k0 - tap * Ct + kc * 4;
            const int td = taps[tap];
#pragma unroll
            for (int i = 0; i < A_V4; ++i) {
                const RowInfo ri = rows[r0 + RPV * i];
                int pix;
                bool ok = tap_pixel(g, ri, td, pix);
                av[i] = ok ? *reinterpret_cast<const float4*>(x1 + (size_t)pix * g.C1 + c0)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < B_V4; ++i) {
                const int j = n0 + r0 + RPV * i;
                bv[i] = j < g.Cout ? *reinterpret_cast<const float4*>(w + (size_t)j * K + k0 + kc * 4)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            const int kk = tid & 31, rg = tid >> 5;
            const int k = k0 + kk;
            const bool kv = k < K;
            const int tap = kv ? k / Ct : 0;
            const int c = k - tap * Ct;
            const int td = taps[tap];
            const bool second = c >= g.C1;
            const float* src = second ? x2 : x1;
            const int cs = second ? g.C2 : g.C1;
            const int cc = second ? c - g.C1 : c;
#pragma unroll
            for (int i = 0; i < A_S; ++i) {
                const RowInfo ri = rows[rg + RPS * i];
                int pix;
                bool ok = tap_pixel(g, ri, td, pix) && kv;
                as[i] = ok ? src[(size_t)pix * cs + cc] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < B_S; ++i) {
                const int j = n0 + rg + RPS * i;
                bs[i] = (kv && j < g.Cout) ? w[(size_t)j * K + k] : 0.f;
            }
        }
    };

    auto store_tile = [&](int buf) {
        if constexpr (FAST) {
            const int kc = tid & 7, r0 = tid >> 3;
#pragma unroll
            for (int i = 0; i < A_V4; ++i)
                *reinterpret_cast<float4*>(&As[buf][(r0 + RPV * i) * LDK + kc * 4]) = av[i];
#pragma unroll
            for (int i = 0; i < B_V4; ++i)
                *reinterpret_cast<float4*>(&Bs[buf][(r0 + RPV * i) * LDK + kc * 4]) = bv[i];
        } else {
            const int kk = tid & 31, rg = tid >> 5;
#pragma unroll
            for (int i = 0; i < A_S; ++i) As[buf][(rg + RPS * i) * LDK + kk] = as[i];
#pragma unroll
            for (int i = 0; i < B_S; ++i) Bs[buf][(rg + RPS * i) * LDK + kk] = bs[i];
        }
    };

    const int wm0 = (wid / WAVES_N) * WM, wn0 = (wid % WAVES_N) * WN;
    const int l31 = lane & 31, lh = lane >> 5;

    auto compute = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < BK / 8; ++ks) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const float4*>(&As[buf][(wm0 + i * 32 + l31) * LDK + ks * 8 + lh * 4]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = *reinterpret_cast<const float4*>(&Bs[buf][(wn0 + j * 32 + l31) * LDK + ks * 8 + lh * 4]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    // lane (l31, lh) feeds A[row l31][k], B[k][col l31] with the SAME k = ks*8 + lh*4 + e
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
    };

    load_tile(0);
    store_tile(0);
    if constexpr (STAGES == 1) {
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) load_tile(kt + 1);  // global loads in flight under the MFMAs below
            compute(0);
            __syncthreads();
            if (kt + 1 < nk) {
                store_tile(0);
                __syncthreads();
            }
        }
    } else {
        if (nk > 1) load_tile(1);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) store_tile((kt + 1) & 1);  // buffer last read in iteration kt-1, released by its barrier
            if (kt + 2 < nk) load_tile(kt + 2);
            compute(kt & 1);
            __syncthreads();
        }
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float vmax = 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn0 + j * 32 + l31;
        if (col >= g.Cout) continue;
        const float bj = bias ? bias[col] : 0.f;
        cg_touch(bj);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const RowInfo ri = rows[row];
                if (ri.base >= 0) {
                    const float v = cg_apply_act(acc[i][j][r] + bj, g.act);
                    y[(size_t)ri.out_off + col] = v;
                    vmax = fmaxf(vmax, fabsf(v));
                }
            }
        }
    }
    if (amax_state) block_amax_store<NT>(vmax, amax_state);
}

// ------------------------------------------------------------------------------------------
// conv_fwd_thin_kernel: forward convolution of the THIN-input layers (3 / 6 / 12 input channels -> 64), exact fp32 MFMA.
//
// These layers (the generators' 7x7 first convolution, the discriminators' 4x4 stride-2 first convolution, the council
// discriminator's two-source 3x3, /root/reference/networks.py:44,152,385-386) have K = taps x channels of 12...147 with
// three channels per pixel: the implicit-GEMM gather of conv_fwd_kernel pays an address computation, a bounds check and
// a 4-byte load per A element and re-stages the weights with every K-slice (29-57 TFLOP/s, profiles/r02_h_final_conv_shapes.txt).
// Here the block is spatial instead:
//   * a block owns 16 x 16 output pixels x all 64 output channels of one image at a time and walks a strided list of
//     such tiles (persistent: grid.x blocks per member);
//   * the input PATCH of a tile -- ((16-1)*S + KH) x ((16-1)*S + KW) pixels x CT channels, zero outside the image, both
//     sources of a two-source layer interleaved per pixel -- is copied into LDS once, with coalesced row loads;
//   * the weights never pass through LDS in the loop: every lane keeps ITS column of B (output channel wn*32 + lane%32,
//     k = 2*kp + lane/32) for the whole K range in registers (K/2 VGPRs), loaded once per block;
//   * an A element is then ONE ds_read_b32 at patch[row base + koff(k)], koff a compile-time constant of the unrolled
//     K loop: no address arithmetic, no bounds checks, no barrier inside the K loop.
// 8 waves: wave (wm, wn) computes rows [64 wm, 64 wm + 64) x channels [32 wn, 32 wn + 32) with v_mfma_f32_32x32x2_f32.
// Taps must be the regular KH x KW raster (dy = kh - pad_y, dx = kw - pad_x), no upsampled source, plain output grid.
// ------------------------------------------------------------------------------------------
template <int KH, int KW, int CT, int S>
__global__ __launch_bounds__(512) void conv_fwd_thin_kernel(
    cg_conv_geom g, const float* __restrict__ x1, const float* __restrict__ x2, const float* __restrict__ w,
    const float* __restrict__ bias, float* __restrict__ y, int imgs_per_member, int pad_y, int pad_x,
    float* __restrict__ amax_state, Members mb) {
    constexpr int TH = 16, TW = 16, NT = 512, BN = 64;
    // K runs over (kh, j), j = kw * CT + c: inside one kernel row the patch offsets of consecutive k are consecutive, so an
    // MFMA (two k's: lane / 32 selects which) takes the pair j = 2 jp, 2 jp + 1 of one row -- the upper half-wave's LDS
    // address is the lower one's + 1 and everything else is an immediate.  An odd row length pads its last pair with a
    // zero weight (7x7x3: 77 MFMAs for 147 k's).
    constexpr int K = KH * KW * CT, RL = KW * CT, JP = (RL + 1) / 2, KP = KH * JP;
    constexpr int KS = K | 1;                                  // odd row stride of the weights in LDS: 32 channels, 32 banks
    constexpr int PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW, PN = PH * PW * CT;
    constexpr int PV = (PN + NT - 1) / NT;                      // patch elements per thread
    __shared__ float wl[BN * KS + 1];
    __shared__ float patch[2][PN + 4];                          // + zeroed pad words: a padded pair may read one element past

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    w = reinterpret_cast<const float*>(reinterpret_cast<const char*>(w) + (long long)blockIdx.z * mb.w_stride);
    if (bias) bias = reinterpret_cast<const float*>(reinterpret_cast<const char*>(bias) + (long long)blockIdx.z * mb.b_stride);

    // this lane's column of B, once per block: w is [Cout][K] (K = (kh, kw, c) raster, c over source 1 then source 2)
    for (int idx = tid; idx < BN * K; idx += NT) {
        const int col = idx / K, k = idx - col * K;
        wl[col * KS + k] = w[idx];
    }
    if (tid == 0) wl[BN * KS] = 0.f;
    if (tid < 8) patch[tid >> 2][PN + (tid & 3)] = 0.f;
    __syncthreads();
    float breg[KP];
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) {
        const int j = 2 * (kp % JP) + lh;
        const float v = wl[(wn * 32 + l31) * KS + (kp / JP) * RL + j];   // j == RL (padded pair) reads the next k / the pad word
        breg[kp] = j < RL ? v : 0.f;
    }
    // this lane's 16 output channels: wn * 32 + 8 q + 4 lh + {0..3}, q = 0..3 (the D rows of its half-wave)
    float4 bj4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float* bq = bias + wn * 32 + 8 * q + 4 * lh;      // (scalar loads: a pool-resident bias is only 4-byte aligned)
        bj4[q] = bias ? make_float4(bq[0], bq[1], bq[2], bq[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        cg_touch(bj4[q].x);
        cg_touch(bj4[q].y);
        cg_touch(bj4[q].z);
        cg_touch(bj4[q].w);
    }

    // LDS offsets of this lane's two A rows: row r of the tile is pixel (r / 16, r % 16)
    int rb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = wm * 64 + i * 32 + l31;
        rb[i] = (((r >> 4) * S) * PW + (r & 15) * S) * CT + lh;
    }

    const int tiles_x = (g.Wo + TW - 1) / TW, tiles_y = (g.Ho + TH - 1) / TH;
    const int tiles_img = tiles_x * tiles_y, ntiles = imgs_per_member * tiles_img;
    const int img0 = (int)blockIdx.z * imgs_per_member;
    float vmax = 0.f;
    // the patch of the NEXT tile travels through registers while the current one is multiplied (two LDS buffers, one
    // barrier per tile)
    float pv[PV];
    auto fetch_patch = [&](int t) {
        const int n = img0 + t / tiles_img, tr = t % tiles_img;
        const int iy0 = (tr / tiles_x) * TH * S - pad_y, ix0 = (tr % tiles_x) * TW * S - pad_x;
#pragma unroll
        for (int j = 0; j < PV; ++j) {
            const int e = tid + j * NT;
            const int pix = e / CT, c = e - pix * CT;
            const int py = pix / PW, px = pix - py * PW;
            const int iy = iy0 + py, ix = ix0 + px;
            float v = 0.f;
            if (e < PN && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W) {
                const size_t p = ((size_t)n * g.H + iy) * g.W + ix;
                v = c < g.C1 ? x1[p * g.C1 + c] : x2[p * g.C2 + (c - g.C1)];
            }
            pv[j] = v;
        }
    };
    auto store_patch = [&](int buf) {
#pragma unroll
        for (int j = 0; j < PV; ++j) {
            const int e = tid + j * NT;
            if (e < PN) patch[buf][e] = pv[j];
        }
    };
    int cur = 0;
    if ((int)blockIdx.x < ntiles) {
        fetch_patch(blockIdx.x);
        store_patch(0);
    }
    __syncthreads();
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int n = img0 + t / tiles_img, tr = t % tiles_img;
        const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * TW;
        const bool more = t + (int)gridDim.x < ntiles;
        if (more) fetch_patch(t + gridDim.x);                   // global loads in flight under the MFMAs below
        const float* __restrict__ pt = patch[cur];

        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
        // A elements are fetched DEPTH k-pairs ahead of the MFMAs that consume them (register ring, static indices)
        constexpr int DEPTH = KP < 4 ? KP : 4;
        auto a_off = [&](int kp) { return (kp / JP) * (PW * CT) + 2 * (kp % JP); };      // compile-time after unrolling
        float a0[DEPTH], a1[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int off = a_off(d);
            a0[d] = pt[rb[0] + off];
            a1[d] = pt[rb[1] + off];
        }
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
            const float u0 = a0[kp % DEPTH], u1 = a1[kp % DEPTH];
            if (kp + DEPTH < KP) {
                const int off = a_off(kp + DEPTH);
                a0[kp % DEPTH] = pt[rb[0] + off];
                a1[kp % DEPTH] = pt[rb[1] + off];
            }
            // operands SWAPPED (round 6): the weights are the MFMA's A operand (rows of D = output channels), the patch elements its B
            // operand (columns of D = pixels) -- the same products in the same order, but the D layout then gives every lane ONE pixel
            // and, per four registers, FOUR CONSECUTIVE channels: the epilogue stores 16 bytes per lane (8 dwordx4 per tile and lane
            // instead of 32 dword stores), which is what a kernel that writes 256 bytes per 54-MFMA pixel needs
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(breg[kp], u0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(breg[kp], u1, acc1, 0, 0, 0);
        }

        // C/D layout of the 32x32 MFMA: col = lane % 32 (pixel), row = (r & 3) + 8 * (r >> 2) + 4 * (lane / 32) (channel)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wm * 64 + i * 32 + l31;
            const int oy = oy0 + (row >> 4), ox = ox0 + (row & 15);
            if (oy < g.Ho && ox < g.Wo) {
                float* __restrict__ dst = y + (((size_t)n * g.Ho + oy) * g.Wo + ox) * BN + wn * 32 + 4 * lh;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v;
                    v.x = cg_apply_act((i ? acc1[4 * q + 0] : acc0[4 * q + 0]) + bj4[q].x, g.act);
                    v.y = cg_apply_act((i ? acc1[4 * q + 1] : acc0[4 * q + 1]) + bj4[q].y, g.act);
                    v.z = cg_apply_act((i ? acc1[4 * q + 2] : acc0[4 * q + 2]) + bj4[q].z, g.act);
                    v.w = cg_apply_act((i ? acc1[4 * q + 3] : acc0[4 * q + 3]) + bj4[q].w, g.act);
                    *reinterpret_cast<float4*>(dst + 8 * q) = v;
                    vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                }
            }
        }
        if (more) store_patch(cur ^ 1);                         // last read by the previous tile, released by its barrier
        __syncthreads();
        cur ^= 1;
    }
    if (amax_state) block_amax_store<NT>(vmax, amax_state);
}

// ------------------------------------------------------------------------------------------
// conv_wgrad_thin_kernel: weight (+ bias) gradient of the same thin-input layers, exact fp32 MFMA.
//   dW[co][k] = sum over output pixels of dz[pix][co] * patch[pix][k]
// Same spatial blocking as conv_fwd_thin_kernel (16 x 16 output pixels of one image per step, persistent over a strided
// tile list), with the reduction index on the MFMA's K: per pixel pair (2p, 2p + 1) the A operand is dz (row = output
// channel) and the B operand is the input patch gathered at this lane's own k (column = k; its patch offset is
// computed once per lane, the pixel pair's is an immediate of the unrolled loop).  Wave w owns output channels
// [32 (w & 1), +32) x all k-tiles for the pixel rows [4 (w >> 1), +4) of a tile, keeps its accumulators for the block's
// whole tile list, and the four pixel slices are summed through LDS at the very end in a fixed order.  A block is one
// split: partial layout and the final reduce are conv_wgrad_kernel's (splitk_reduce_kernel).
// ------------------------------------------------------------------------------------------
template <int KH, int KW, int CT, int S>
__global__ __launch_bounds__(512) void conv_wgrad_thin_kernel(
    cg_conv_geom g, const float* __restrict__ x1, const float* __restrict__ x2, const float* __restrict__ dz,
    float* __restrict__ part, int imgs_per_member, int pad_y, int pad_x, int want_bias, const float* __restrict__ yact, int act) {
    // yact != NULL: `dz` is the gradient of the layer's ACTIVATED output y = act(conv) and yact is that output -- the activation
    // backward dz = dy * act'(y) happens on the way into LDS (same arithmetic as act_bwd_kernel: bit-identical weight gradients),
    // so the layers that need no data gradient (the discriminators' first layers) never materialise dz
    constexpr int TH = 16, TW = 16, NT = 512, CO = 64;
    constexpr int K = KH * KW * CT, NKT = (K + 31) / 32;
    constexpr int PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW, PN = PH * PW * CT;
    constexpr int PV = (PN + NT - 1) / NT;                      // patch elements per thread
    constexpr int DZN = TH * TW * CO, DV = DZN / 4 / NT;        // dz tile: floats, float4 per thread
    static_assert(2 * NKT * 1024 + CO <= DZN, "the final reduction reuses the dz tile");
    __shared__ __attribute__((aligned(16))) float dzl[DZN];
    __shared__ float patch[PN + 4];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int cot = wid & 1, slice = wid >> 1;
    const int l31 = lane & 31, lh = lane >> 5;

    // this lane's column k of every k-tile -> offset inside a patch (k >= K: any valid offset, never stored)
    int koff[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        const int k = kt * 32 + l31;
        const int kk = k < K ? k : 0;
        const int tap = kk / CT, c = kk - tap * CT;
        koff[kt] = ((tap / KW) * PW + (tap % KW)) * CT + c;
    }
    // pixel 64 slice + 2 q + lh of the tile: (py, px) = (4 slice + q / 8, 2 (q % 8) + lh)
    const int pbase = ((4 * slice * S) * PW) * CT + lh * S * CT;           // + the immediate of q below
    const int abase = (64 * slice + lh) * CO + cot * 32 + l31;             // + 2 q CO

    const int tiles_x = (g.Wo + TW - 1) / TW, tiles_y = (g.Ho + TH - 1) / TH;
    const int tiles_img = tiles_x * tiles_y, ntiles = imgs_per_member * tiles_img;
    const int img0 = (int)blockIdx.z * imgs_per_member;

    float pv[PV];
    float4 dv[DV];
    auto fetch = [&](int t) {
        const int n = img0 + t / tiles_img, tr = t % tiles_img;
        const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * TW;
        const int iy0 = oy0 * S - pad_y, ix0 = ox0 * S - pad_x;
#pragma unroll
        for (int j = 0; j < PV; ++j) {
            const int e = tid + j * NT;
            const int pix = e / CT, c = e - pix * CT;
            const int py = pix / PW, px = pix - py * PW;
            const int iy = iy0 + py, ix = ix0 + px;
            float v = 0.f;
            if (e < PN && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W) {
                const size_t p = ((size_t)n * g.H + iy) * g.W + ix;
                v = c < g.C1 ? x1[p * g.C1 + c] : x2[p * g.C2 + (c - g.C1)];
            }
            pv[j] = v;
        }
#pragma unroll
        for (int j = 0; j < DV; ++j) {
            const int f = tid + j * NT;                          // float4 f of the tile: pixel f / 16, channels 4 (f % 16) ..
            const int pix = f >> 4, c4 = f & 15;
            const int oy = oy0 + (pix >> 4), ox = ox0 + (pix & 15);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (oy < g.Ho && ox < g.Wo) {
                const size_t e = (((size_t)n * g.Ho + oy) * g.Wo + ox) * CO + c4 * 4;
                v = *reinterpret_cast<const float4*>(dz + e);
                if (yact) {
                    const float4 yv = *reinterpret_cast<const float4*>(yact + e);
                    v.x *= cg_act_grad_from_out(yv.x, act);
                    v.y *= cg_act_grad_from_out(yv.y, act);
                    v.z *= cg_act_grad_from_out(yv.z, act);
                    v.w *= cg_act_grad_from_out(yv.w, act);
                }
            }
            dv[j] = v;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int j = 0; j < PV; ++j) {
            const int e = tid + j * NT;
            if (e < PN) patch[e] = pv[j];
        }
#pragma unroll
        for (int j = 0; j < DV; ++j) *reinterpret_cast<float4*>(&dzl[(tid + j * NT) * 4]) = dv[j];
    };

    f32x16 acc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kt][r] = 0.f;
    float bsum = 0.f;

    if (tid < 4) patch[PN + tid] = 0.f;
    fetch(blockIdx.x);                                          // the launcher guarantees gridDim.x <= ntiles
    stash();
    __syncthreads();
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const bool more = t + (int)gridDim.x < ntiles;
        if (more) fetch(t + gridDim.x);                         // global loads in flight under the MFMAs below
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const float a = dzl[abase + 2 * q * CO];
            bsum += a;
            const int pq = ((q >> 3) * S * PW + 2 * (q & 7) * S) * CT;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
                acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, patch[pbase + pq + koff[kt]], acc[kt], 0, 0, 0);
        }
        __syncthreads();                                        // every wave is done with this tile's LDS image
        if (more) stash();
        __syncthreads();
    }

    // sum the four pixel slices in a fixed order: red[(cot * NKT + kt) * 16 + r][lane], then the bias row
    float* red = dzl;
    const float bs = bsum + __shfl_xor(bsum, 32, 64);
    for (int rnd = 0; rnd < 4; ++rnd) {
        if (slice == rnd) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int idx = ((cot * NKT + kt) * 16 + r) * 64 + lane;
                    red[idx] = (rnd ? red[idx] : 0.f) + acc[kt][r];
                }
            if (lh == 0) {
                const int idx = 2 * NKT * 1024 + cot * 32 + l31;
                red[idx] = (rnd ? red[idx] : 0.f) + bs;
            }
        }
        __syncthreads();
    }
    float* dst = part + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * ((size_t)CO * K + CO);   // per (member, split)
    for (int e = tid; e < CO * K; e += NT) {
        const int co = e / K, k = e - co * K;
        const int row = co & 31, col = k & 31;
        // C/D layout of the 32x32 MFMA: lane = col + 32 * ((row >> 2) & 1), register (row & 3) + 4 * (row >> 3)
        dst[e] = red[(((co >> 5) * NKT + (k >> 5)) * 16 + (row & 3) + 4 * (row >> 3)) * 64 + col + 32 * ((row >> 2) & 1)];
    }
    if (want_bias && tid < CO) dst[(size_t)CO * K + tid] = red[2 * NKT * 1024 + tid];
}

// ------------------------------------------------------------------------------------------
// forward / data-gradient kernel, software-pipelined ("pipe") -- the hot one.
//
// Same GEMM view and LDS image as conv_fwd_kernel<.., FAST>, restricted to single-source inputs whose
// channel count is a multiple of BK (every K-slice lies inside one tap), but scheduled so that the
// matrix pipe never drains around the block barrier:
//   * two LDS buffers, ONE barrier per K-slice;
//   * operand fragments are double-buffered in registers: the fragment for k-step s+1 is fetched from LDS
//     while the MFMAs of k-step s run, and the first fragment of the NEXT slice is fetched right after the
//     barrier while the last MFMAs of the current slice run -- so every wave arrives at the barrier with
//     its next 4*TM*TN MFMAs ready to issue (operands already in VGPRs), and leaves it issuing them;
//   * the staging work of a slice (ds_write of the prefetched global data, address arithmetic and
//     buffer_loads of the slice after next) sits at a DIFFERENT point of the k-step sequence for the two
//     waves that share a SIMD (waves w and w + NW/2), so one of them always feeds the matrix pipe;
//   * the gather is branch-free: buffer_load with a hardware range check (padded taps and rows beyond
//     M / Cout use an out-of-range offset and read as zero).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    // keep `auto`: converting the builtin's vector type to an ext_vector splats lane 0 (a dword load)
    auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    static_assert(sizeof(v) == sizeof(float4), "buffer_load_dwordx4");
    return __builtin_bit_cast(float4, v);
}

constexpr unsigned CG_OOB = 0x80000000u;

// timing probe of the pipelined kernel (tile configuration 31 only): per block {shader clock, 100 MHz wall clock} at
// entry / loop start / loop end / exit, and the hardware id -- read back with cg_debug_fetch()
constexpr int CG_DBG_WORDS = 16;
__device__ long long cg_dbg[4096 * CG_DBG_WORDS];
__device__ __forceinline__ void dbg_stamp(int tile, int slot) {
    if (threadIdx.x == 0 && tile < 4096) {
        cg_dbg[tile * CG_DBG_WORDS + slot * 2] = (long long)__builtin_amdgcn_s_memtime();
        cg_dbg[tile * CG_DBG_WORDS + slot * 2 + 1] = (long long)wall_clock64();
    }
}  // >= num_records of any tensor validate_geom lets through here

// FOLD > 0 ("chunked sum", round 6): the accumulator is folded into a second register set every FOLD K-slices.  v_mfma_f32_32x32x2_f32
// rounds once per TWO products, so a K = 2304 contraction is a chain of 1152 dependent fp32 roundings -- a random walk of ~sqrt(K/2)
// half-ulps, 2.6x the round-off of the reference's CPU kernels (16-lane FMA chains of K/16) and the reason this datapath drew more
// ReLU / focus-loss sign flips than the reference arithmetic in the generator-gradient statistic.  Summing
// chunks of FOLD x 16 MFMAs first and the chunk sums afterwards shortens both chains (blocked summation): for FOLD = 4 the expected
// round-off drops ~3.7x -- below the CPU kernels'.  Costs TM x TN x 16 registers and one v_add per accumulator value and chunk.
template <int BM, int BN, int WM, int WN, int PF, int ABL, int FOLD = 0>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, (FOLD > 0 && BM == 128 && BN == 128 && WM == 64 && WN == 32) ? 4 : 1) void conv_fwd_pipe_kernel(
    PipeBatch batch, const float* __restrict__ x1, const float* __restrict__ bias, float* __restrict__ y,
    unsigned x_bytes, double* __restrict__ stats, Members mb) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int NW = (BM / WM) * (BN / WN);
    constexpr int NT = NW * 64;
    constexpr int RPV = NT / 8;                      // rows per loader pass (8 float4 per 32-float row)
    constexpr int A_V4 = BM / RPV, B_V4 = BN / RPV;  // float4 per thread and slice
    static_assert(BM % RPV == 0 && BN % RPV == 0, "tile rows must be a multiple of the loader rows");
    __shared__ __attribute__((aligned(16))) float As[2][BM * LDK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDK];
    __shared__ RowInfo rows[BM];
    __shared__ int taps[CG_MAX_TAPS];

    // this block's class: read straight from the kernarg segment (scalar loads; `batch` is the FIRST argument --
    // indexing the by-value struct with blockIdx.y would make the compiler copy it to scratch)
    typedef const __attribute__((address_space(4))) PipeClass* KernargClass;
    const int cls_off = (int)blockIdx.y * (int)sizeof(PipeClass);
    KernargClass pc = (KernargClass)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + cls_off);
    const int ntiles = pc->ntiles;
    if ((int)blockIdx.x >= ntiles) return;
    GeomS g;
    {
        const __attribute__((address_space(4))) int32_t* gi = (const __attribute__((address_space(4))) int32_t*)pc;
        int32_t* go = reinterpret_cast<int32_t*>(&g);
#pragma unroll
        for (int i = 0; i < (int)(sizeof(GeomS) / 4); ++i) go[i] = gi[i];
    }
    const float* __restrict__ w = reinterpret_cast<const float*>(reinterpret_cast<const char*>(pc->w) + (long long)blockIdx.z * mb.w_stride);
    if (bias) bias = reinterpret_cast<const float*>(reinterpret_cast<const char*>(bias) + (long long)blockIdx.z * mb.b_stride);
    const int m_lo = (int)blockIdx.z * pc->M;        // this member's rows: [m_lo, m_lo + pc->M)
    const int M = m_lo + pc->M, K = pc->K, tiles_n = pc->tiles_n;
    const unsigned w_bytes = pc->w_bytes;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int tile = xcd_remap(blockIdx.x, ntiles);
    const int m0 = m_lo + (tile / tiles_n) * BM;
    const int n0 = (tile % tiles_n) * BN;
    const int Ct = g.C1;

    if constexpr (ABL == 3) {
        dbg_stamp(tile, 0);
        if (tid == 0 && tile < 4096) {
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            cg_dbg[tile * CG_DBG_WORDS + 8] = ((long long)xcc << 32) | hwid;
            cg_dbg[tile * CG_DBG_WORDS + 9] = blockIdx.x;
        }
    }
    if (tid < g.T) taps[tid] = load_tap(tid, cls_off);
    for (int r = tid; r < BM; r += NT) rows[r] = decode_row(g, m0 + r, M, true);
    __syncthreads();

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x1, 0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)w_bytes, 0x00020000);

    // loader role of this thread: float4 column kc of rows r0 + RPV*i of both operand tiles
    const int kc = tid & 7, r0 = tid >> 3;
    int rbase[A_V4], rly[A_V4], rlx[A_V4];
#pragma unroll
    for (int i = 0; i < A_V4; ++i) {
        const RowInfo ri = rows[r0 + RPV * i];
        rbase[i] = ri.base;
        rly[i] = ri.ly0;
        rlx[i] = ri.lx0;
    }
    unsigned woff[B_V4];
#pragma unroll
    for (int i = 0; i < B_V4; ++i) {
        const int j = n0 + r0 + RPV * i;
        woff[i] = (((unsigned)j * (unsigned)K + kc * 4) << 2) | (j < g.Cout ? 0u : CG_OOB);
    }
    const int Hl = g.H << g.up, Wl = g.W << g.up;
    const int nk = K / BK;
    // PF = prefetch distance in K-slices = number of staging register sets (the data of slice kt+1+PF is
    // requested while slice kt is computed).  ABL: timing ablations only (1 = every load re-reads slice 0, results
    // wrong; 2 = no stagger between the two waves of a SIMD).
    static_assert(PF == 1 || PF == 2, "one or two staging register sets");
    float4 av[PF][A_V4], bv[PF][B_V4];
    int ld_kt = 0, ld_tap = 0, ld_c0 = 0;  // slice the next load_tile() fetches (clamped to the last slice)

    auto load_tile = [&](auto set_c) {
        constexpr int SET = decltype(set_c)::value;
        const int td = __builtin_amdgcn_readfirstlane(taps[ld_tap]);
        const int dy = (int)(short)(td & 0xffff), dx = td >> 16;
        const unsigned cb = (unsigned)(ld_c0 + kc * 4);
#pragma unroll
        for (int i = 0; i < A_V4; ++i) {
            const int ly = rly[i] + dy, lx = rlx[i] + dx;
            const bool ok = rbase[i] >= 0 && (unsigned)ly < (unsigned)Hl && (unsigned)lx < (unsigned)Wl;
            const unsigned pix = (unsigned)(rbase[i] + (ly >> g.up) * g.W + (lx >> g.up));
            // branch-free: an invalid tap only sets the top offset bit, the range check then returns zeros
            av[SET][i] = buf_load4(xr, ((pix * (unsigned)Ct + cb) << 2) | (ok ? 0u : CG_OOB));
        }
        const unsigned kb = (unsigned)ld_kt * (BK * 4u);
#pragma unroll
        for (int i = 0; i < B_V4; ++i) bv[SET][i] = buf_load4(wr, woff[i] + kb);
        if (ABL != 1 && ld_kt + 1 < nk) {
            ++ld_kt;
            ld_c0 += BK;
            if (ld_c0 == Ct) {
                ld_c0 = 0;
                ++ld_tap;
            }
        }
    };
    auto store_tile = [&](int buf, auto set_c) {
        constexpr int SET = decltype(set_c)::value;
#pragma unroll
        for (int i = 0; i < A_V4; ++i) *reinterpret_cast<float4*>(&As[buf][(r0 + RPV * i) * LDK + kc * 4]) = av[SET][i];
#pragma unroll
        for (int i = 0; i < B_V4; ++i) *reinterpret_cast<float4*>(&Bs[buf][(r0 + RPV * i) * LDK + kc * 4]) = bv[SET][i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm0 = (wid / WAVES_N) * WM, wn0 = (wid % WAVES_N) * WN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int a_off = (wm0 + l31) * LDK + lh * 4, b_off = (wn0 + l31) * LDK + lh * 4;

    auto read_frag = [&](int buf, int ks, float4(&a)[TM], float4(&b)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4*>(&As[buf][a_off + i * 32 * LDK + ks * 8]);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4*>(&Bs[buf][b_off + j * 32 * LDK + ks * 8]);
    };
    auto mma = [&](const float4(&a)[TM], const float4(&b)[TN]) {
        // lane (l31, lh) feeds A[row l31][k], B[k][col l31] with the SAME k = ks*8 + lh*4 + e
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    constexpr bool SB = FOLD > 0 && BM == 128 && BN == 128 && WM == 64 && WN == 32;      // single fragment set, see slice()
    float4 a0[TM], b0[TN], a1[TM], b1[TN];      // (SB: a1 / b1 are never live)
    load_tile(I0());  // slice 0
    store_tile(0, I0());
    if constexpr (PF == 2) {
        load_tile(I1());  // slice 1 -> set 1
        load_tile(I0());  // slice 2 -> set 0
    } else {
        load_tile(I0());  // slice 1 (or slice 0 again when nk == 1) stays in registers
    }
    __syncthreads();
    read_frag(0, 0, a0, b0);
    if constexpr (ABL == 3) dbg_stamp(tile, 1);

    // inside one k-step group: the LDS fetches of the NEXT fragment issue first, the MFMAs of the current one follow
    auto group_order = [&]() {
        __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);   // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);  // MFMA
        __builtin_amdgcn_sched_barrier(0);
    };
    // One K-slice out of LDS buffer CUR.  Staging = write slice kt+1 (register set SET) to the other buffer and
    // request slice kt+1+PF into the same set; STAGE_AT selects where in the k-step sequence this wave does it.
    auto slice = [&](auto cur_c, auto stage_at) {
        constexpr int CUR = decltype(cur_c)::value;
        constexpr int STAGE_AT = decltype(stage_at)::value;
        using SET = std::integral_constant<int, PF == 2 ? (CUR ^ 1) : 0>;
        if constexpr (SB) {
            // single fragment set (the chunked-sum 128x128 / 8-wave tile: the second accumulator set must not cost the second block
            // per CU): a k-step's MFMAs issue, then the next k-step's fragments are read into the same registers -- the other three
            // waves of the SIMD keep the matrix pipe busy meanwhile
            if constexpr (STAGE_AT == 0) {
                store_tile(CUR ^ 1, SET());
                load_tile(SET());
                __builtin_amdgcn_sched_barrier(0);
            }
            mma(a0, b0);
            read_frag(CUR, 1, a0, b0);
            if constexpr (STAGE_AT == 1) {
                store_tile(CUR ^ 1, SET());
                load_tile(SET());
                __builtin_amdgcn_sched_barrier(0);
            }
            mma(a0, b0);
            read_frag(CUR, 2, a0, b0);
            mma(a0, b0);
            read_frag(CUR, 3, a0, b0);
            mma(a0, b0);
            __syncthreads();
            read_frag(CUR ^ 1, 0, a0, b0);
            return;
        }
        if constexpr (STAGE_AT == 0) {
            store_tile(CUR ^ 1, SET());
            load_tile(SET());
            __builtin_amdgcn_sched_barrier(0);
        }
        read_frag(CUR, 1, a1, b1);
        mma(a0, b0);
        group_order();
        if constexpr (STAGE_AT == 1) {
            store_tile(CUR ^ 1, SET());
            load_tile(SET());
            __builtin_amdgcn_sched_barrier(0);
        }
        read_frag(CUR, 2, a0, b0);
        mma(a1, b1);
        group_order();
        read_frag(CUR, 3, a1, b1);
        mma(a0, b0);
        group_order();
        __syncthreads();
        read_frag(CUR ^ 1, 0, a0, b0);
        mma(a1, b1);
        group_order();
    };
    f32x16 tot[FOLD ? TM : 1][FOLD ? TN : 1];
    if constexpr (FOLD > 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;
    }
    auto fold = [&](bool last) {
        if constexpr (FOLD > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float t = tot[i][j][r] + acc[i][j][r];
                        tot[i][j][r] = t;
                        acc[i][j][r] = last ? t : 0.f;       // the epilogue reads acc
                    }
        }
    };
    static_assert(FOLD == 0 || FOLD == 4, "chunks of four K-slices");
    if (ABL != 2 && NW >= 8 && wid >= NW / 2) {
        for (int kt = 0; kt + 1 < nk; kt += 2) {
            slice(I0(), I1());
            slice(I1(), I1());
            if (FOLD > 0 && (kt & 2)) fold(false);      // every second pair of slices
        }
        if (nk & 1) slice(I0(), I1());
    } else {
        for (int kt = 0; kt + 1 < nk; kt += 2) {
            slice(I0(), I0());
            slice(I1(), I0());
            if (FOLD > 0 && (kt & 2)) fold(false);
        }
        if (nk & 1) slice(I0(), I0());
    }
    fold(true);
    if constexpr (ABL == 3) dbg_stamp(tile, 2);

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // With `stats` (instance-norm fusion; the launcher guarantees M % BM == 0 and act == none) the block also emits
    // {sum y, sum y^2} of its BM rows per output column, accumulated in fp64.
    double cs[TN], cq[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        cs[j] = 0.0;
        cq[j] = 0.0;
        const int col = n0 + wn0 + j * 32 + l31;
        if (col >= g.Cout) continue;
        const float bj = bias ? bias[col] : 0.f;
        cg_touch(bj);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const RowInfo ri = rows[row];
                const float v = cg_apply_act(acc[i][j][r] + bj, g.act);
                if (ri.base >= 0) y[(size_t)ri.out_off + col] = v;
                if (stats) {
                    cs[j] += (double)v;
                    cq[j] += (double)v * (double)v;
                }
            }
        }
    }
    if (stats) {
        constexpr int WAVES_M = BM / WM;
        double* red = reinterpret_cast<double*>(&As[0][0]);   // [WAVES_M][BN][2], the operand buffers are free now
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const double a = cs[j] + __shfl_xor(cs[j], 32, 64), b = cq[j] + __shfl_xor(cq[j], 32, 64);
            if (lh == 0) {
                red[((wid / WAVES_N) * BN + wn0 + j * 32 + l31) * 2] = a;
                red[((wid / WAVES_N) * BN + wn0 + j * 32 + l31) * 2 + 1] = b;
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < g.Cout) {
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int wmi = 0; wmi < WAVES_M; ++wmi) {
                a += red[(wmi * BN + tid) * 2];
                b += red[(wmi * BN + tid) * 2 + 1];
            }
            double* o = stats + ((size_t)(m0 / BM) * g.Cout + n0 + tid) * 2;      // global row tile (members included)
            o[0] = a;
            o[1] = b;
        }
    }
    if constexpr (ABL == 3) {
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        dbg_stamp(tile, 3);
    }
}

// ------------------------------------------------------------------------------------------
// weight-gradient kernel:  part[split][co][k] = sum_{m in split} dz[m][co] * A[m][k]
// ------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool FAST>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void conv_wgrad_kernel(cg_conv_geom g, const float* __restrict__ x1,
                                                         const float* __restrict__ x2,
                                                         const float* __restrict__ dz, float* __restrict__ out,
                                                         int M, int K, int tiles_n, int slices_per_split,
                                                         int want_bias) {
    // grouped launches: blockIdx.y = council member; M = output rows of ONE member, whose rows are [m_lo, m_lo + M)
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int BP = 32;  // output positions per stage
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    static_assert(NT == 256 || NT == 512, "4 or 8 waves per block");
    __shared__ __attribute__((aligned(16))) float Ds[BP * BM];
    __shared__ __attribute__((aligned(16))) float Xs[BP * BN];
    __shared__ RowInfo rows[2][BP];
    __shared__ int taps[CG_MAX_TAPS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int tile = blockIdx.x;
    const int co0 = (tile / tiles_n) * BM;
    const int j0 = (tile % tiles_n) * BN;
    const int Ct = g.C1 + g.C2;
    const int split = blockIdx.z;
    const int nslices_total = (M + BP - 1) / BP;
    const int s_begin = split * slices_per_split;
    const int s_end = min(s_begin + slices_per_split, nslices_total);

    if (tid < g.T) taps[tid] = load_tap(tid);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int D_V4 = (BP * BM / 4) / NT;  // float4 per thread for the dz tile
    constexpr int X_V4 = (BP * BN / 4) / NT;
    constexpr int D_S = BP * BM / NT, X_S = BP * BN / NT;
    static_assert((BP * BM / 4) % NT == 0 || BP * BM / 4 < NT, "dz tile / loader mismatch");
    const bool dvec = (g.Cout & 3) == 0;
    float4 dv[D_V4 > 0 ? D_V4 : 1];
    float ds[D_S > 0 ? D_S : 1];
    float4 xv[FAST ? (X_V4 > 0 ? X_V4 : 1) : 1];
    float xs[FAST ? 1 : (X_S > 0 ? X_S : 1)];

    auto load_tile = [&](int buf) {
        // dz tile: rows = positions, cols = co (contiguous in memory)
        if (dvec && D_V4 > 0) {
            constexpr int CH = BM / 4;        // float4 chunks per row
            constexpr int RP = NT / CH;      // rows per pass
            const int ch = tid % CH, r0 = tid / CH;
#pragma unroll
            for (int i = 0; i < (D_V4 > 0 ? D_V4 : 1); ++i) {
                if (r0 + RP * i >= BP) break;
                const RowInfo ri = rows[buf][r0 + RP * i];
                const int co = co0 + ch * 4;
                dv[i] = (ri.base >= 0 && co < g.Cout)
                            ? *reinterpret_cast<const float4*>(dz + (size_t)ri.out_off + co)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            const int cc = tid % BM, r0 = tid / BM;
            constexpr int RP = NT / BM > 0 ? NT / BM : 1;
#pragma unroll
            for (int i = 0; i < (D_S > 0 ? D_S : 1); ++i) {
                if (r0 + RP * i >= BP) break;
                const RowInfo ri = rows[buf][r0 + RP * i];
                const int co = co0 + cc;
                ds[i] = (ri.base >= 0 && co < g.Cout) ? dz[(size_t)ri.out_off + co] : 0.f;
            }
        }
        if constexpr (FAST) {
            constexpr int CH = BN / 4;
            constexpr int RP = NT / CH;
            const int ch = tid % CH, r0 = tid / CH;
            const int tap = j0 / Ct;
            const int c0 = j0 - tap * Ct + ch * 4;
            const int td = taps[tap];
#pragma unroll
            for (int i = 0; i < (X_V4 > 0 ? X_V4 : 1); ++i) {
                if (r0 + RP * i >= BP) break;
                const RowInfo ri = rows[buf][r0 + RP * i];
                int pix;
                bool ok = tap_pixel(g, ri, td, pix);
                xv[i] = ok ? *reinterpret_cast<const float4*>(x1 + (size_t)pix * g.C1 + c0)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            const int jj = tid % BN, r0 = tid / BN;
            constexpr int RP = NT / BN;
            const int k = j0 + jj;
            const bool kv = k < K;
            const int tap = kv ? k / Ct : 0;
            const int c = k - tap * Ct;
            const int td = taps[tap];
            const bool second = c >= g.C1;
            const float* src = second ? x2 : x1;
            const int cs = second ? g.C2 : g.C1;
            const int cc = second ? c - g.C1 : c;
#pragma unroll
            for (int i = 0; i < (X_S > 0 ? X_S : 1); ++i) {
                if (r0 + RP * i >= BP) break;
                const RowInfo ri = rows[buf][r0 + RP * i];
                int pix;
                bool ok = tap_pixel(g, ri, td, pix) && kv;
                xs[i] = ok ? src[(size_t)pix * cs + cc] : 0.f;
            }
        }
    };

    auto store_tile = [&]() {
        if (dvec && D_V4 > 0) {
            constexpr int CH = BM / 4;
            constexpr int RP = NT / CH;
            const int ch = tid % CH, r0 = tid / CH;
#pragma unroll
            for (int i = 0; i < (D_V4 > 0 ? D_V4 : 1); ++i)
                if (r0 + RP * i < BP) *reinterpret_cast<float4*>(&Ds[(r0 + RP * i) * BM + ch * 4]) = dv[i];
        } else {
            const int cc = tid % BM, r0 = tid / BM;
            constexpr int RP = NT / BM > 0 ? NT / BM : 1;
#pragma unroll
            for (int i = 0; i < (D_S > 0 ? D_S : 1); ++i)
                if (r0 + RP * i < BP) Ds[(r0 + RP * i) * BM + cc] = ds[i];
        }
        if constexpr (FAST) {
            constexpr int CH = BN / 4;
            constexpr int RP = NT / CH;
            const int ch = tid % CH, r0 = tid / CH;
#pragma unroll
            for (int i = 0; i < (X_V4 > 0 ? X_V4 : 1); ++i)
                if (r0 + RP * i < BP) *reinterpret_cast<float4*>(&Xs[(r0 + RP * i) * BN + ch * 4]) = xv[i];
        } else {
            const int jj = tid % BN, r0 = tid / BN;
            constexpr int RP = NT / BN;
#pragma unroll
            for (int i = 0; i < (X_S > 0 ? X_S : 1); ++i)
                if (r0 + RP * i < BP) Xs[(r0 + RP * i) * BN + jj] = xs[i];
        }
    };

    const int m_lo = (int)blockIdx.y * M;
    auto decode_rows = [&](int slice, int buf) {
        if (tid < BP) {
            rows[buf][tid] = decode_row(g, m_lo + slice * BP + tid, m_lo + M, false);
        }
    };

    const int wm0 = (wid / WAVES_N) * WM, wn0 = (wid % WAVES_N) * WN;
    const int l31 = lane & 31, lh = lane >> 5;
    // bias gradient = column sums of dz: the first k-tile of every co-tile adds up the dz tile it stages anyway
    const bool do_bias = want_bias && (tile % tiles_n == 0) && tid < BM;
    float bsum = 0.f;

    if (s_begin < s_end) {
        decode_rows(s_begin, 0);
        __syncthreads();
        load_tile(0);
        store_tile();
        if (s_begin + 1 < s_end) decode_rows(s_begin + 1, 1);
        __syncthreads();
        for (int s = s_begin; s < s_end; ++s) {
            const int nb = (s - s_begin + 1) & 1;
            const bool more = s + 1 < s_end;
            if (more) load_tile(nb);  // in flight under the MFMAs
            if (do_bias) {
#pragma unroll 8
                for (int p = 0; p < BP; ++p) bsum += Ds[p * BM + tid];
            }
#pragma unroll 4
            for (int p = 0; p < BP / 2; ++p) {
                float a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = Ds[(2 * p + lh) * BM + wm0 + i * 32 + l31];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = Xs[(2 * p + lh) * BN + wn0 + j * 32 + l31];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
            if (more) {
                store_tile();
                if (s + 2 < s_end) decode_rows(s + 2, nb ^ 1);
                __syncthreads();
            }
        }
    }

    // partial layout per (member, split): [Cout*K weight partials | Cout bias partials]
    float* dst = out + ((size_t)blockIdx.y * gridDim.z + split) * ((size_t)g.Cout * K + g.Cout);
    if (do_bias && co0 + tid < g.Cout) dst[(size_t)g.Cout * K + co0 + tid] = bsum;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = j0 + wn0 + j * 32 + l31;
        if (col >= K) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = co0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < g.Cout) dst[(size_t)row * K + col] = acc[i][j][r];
            }
    }
}

// ------------------------------------------------------------------------------------------
// weight-gradient kernel, software-pipelined (same schedule as conv_fwd_pipe_kernel):
//   part[split][co][k] = sum_{m in split} dz[m][co] * A[m][k]
// Restrictions (the launcher falls back to conv_wgrad_kernel otherwise): one source, every k-tile inside one
// tap (C1 % BN == 0), Cout % 4 == 0, Ho*Wo and Wo powers of two (row decode by shifts, per thread, no LDS
// round trip), operands addressable with 31-bit byte offsets.
// LDS image: Ds[pos][co], Xs[pos][k] (position-major, as loaded).  MFMA tile i of a wave covers the
// INTERLEAVED rows wm0 + l*TM + i (l = 0..31), so one lane's TM operands of a position are adjacent in LDS
// and come with one ds_read_b64; likewise for the k columns when TN = 2.
// ------------------------------------------------------------------------------------------
template <int N>
struct FVec;
template <>
struct FVec<1> {
    float v[1];
    __device__ __forceinline__ void load(const float* p) { v[0] = *p; }
};
template <>
struct FVec<2> {
    float v[2];
    __device__ __forceinline__ void load(const float* p) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        v[0] = t.x;
        v[1] = t.y;
    }
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void conv_wgrad_pipe_kernel(
    cg_conv_geom g, const float* __restrict__ x1, const float* __restrict__ dz, float* __restrict__ out, int M, int K,
    int tiles_n, int slices_per_split, int want_bias, int lg_hw, int lg_wo, unsigned x_bytes, unsigned dz_bytes) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int NW = (BM / WM) * (BN / WN);
    constexpr int NT = NW * 64;
    constexpr int BP = 32;  // output positions per K-slice
    constexpr int D_CH = BM / 4, D_RP = NT / D_CH, D_V4 = BP / D_RP;
    constexpr int X_CH = BN / 4, X_RP = NT / X_CH, X_V4 = BP / X_RP;
    static_assert(TM <= 2 && TN <= 2, "wave tile at most 64 x 64");
    static_assert(NT % D_CH == 0 && NT % X_CH == 0 && D_V4 >= 1 && X_V4 >= 1 && BP % D_RP == 0 && BP % X_RP == 0,
                  "tile / loader mismatch");
    __shared__ __attribute__((aligned(16))) float Ds[2][BP * BM];
    __shared__ __attribute__((aligned(16))) float Xs[2][BP * BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int tile = blockIdx.x;
    const int co0 = (tile / tiles_n) * BM;
    const int j0 = (tile % tiles_n) * BN;
    const int Ct = g.C1;
    const int split = blockIdx.z;
    const int nslices_total = (M + BP - 1) / BP;
    const int s_begin = split * slices_per_split;
    const int s_end = min(s_begin + slices_per_split, nslices_total);
    const int nk = s_end - s_begin;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x1, 0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void*)dz, 0, (int)dz_bytes, 0x00020000);

    // this block's k-tile lies inside ONE tap: (dy, dx) and the channel offset are block constants
    const int tap = j0 / Ct;
    const int td = load_tap(tap);
    const int dy = (int)(short)(td & 0xffff), dx = td >> 16;
    const int d_ch = tid % D_CH, d_r0 = tid / D_CH;
    const int x_ch = tid % X_CH, x_r0 = tid / X_CH;
    const unsigned d_col = (unsigned)(co0 + d_ch * 4);
    const unsigned d_oob = d_col < (unsigned)g.Cout ? 0u : CG_OOB;
    const unsigned x_col = (unsigned)(j0 - tap * Ct + x_ch * 4);
    const int Hl = g.H << g.up, Wl = g.W << g.up;
    const int hw_mask = (1 << lg_hw) - 1, wo_mask = (1 << lg_wo) - 1;
    const int img = g.H * g.W;

    float4 dv[D_V4], xv[X_V4];
    int ld_s = s_begin;  // slice the next load_tile() fetches (clamped to the last slice of this split)
    const int m_lo = (int)blockIdx.y * M, m_hi = m_lo + M;     // grouped launches: blockIdx.y = member, M = ITS rows
    auto load_tile = [&]() {
        const int mb = m_lo + ld_s * BP;
#pragma unroll
        for (int i = 0; i < D_V4; ++i) {
            const int m = mb + d_r0 + D_RP * i;
            dv[i] = buf_load4(dr, ((((unsigned)m * (unsigned)g.Cout) + d_col) << 2) | d_oob | (m < m_hi ? 0u : CG_OOB));
        }
#pragma unroll
        for (int i = 0; i < X_V4; ++i) {
            const int m = mb + x_r0 + X_RP * i;
            const int n = m >> lg_hw, rem = m & hw_mask;
            const int ly = (rem >> lg_wo) * g.stride + dy, lx = (rem & wo_mask) * g.stride + dx;
            const bool ok = m < m_hi && (unsigned)ly < (unsigned)Hl && (unsigned)lx < (unsigned)Wl;
            const unsigned pix = (unsigned)(n * img + (ly >> g.up) * g.W + (lx >> g.up));
            xv[i] = buf_load4(xr, ((pix * (unsigned)Ct + x_col) << 2) | (ok ? 0u : CG_OOB));
        }
        if (ld_s + 1 < s_end) ++ld_s;
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < D_V4; ++i) *reinterpret_cast<float4*>(&Ds[buf][(d_r0 + D_RP * i) * BM + d_ch * 4]) = dv[i];
#pragma unroll
        for (int i = 0; i < X_V4; ++i) *reinterpret_cast<float4*>(&Xs[buf][(x_r0 + X_RP * i) * BN + x_ch * 4]) = xv[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm0 = (wid / WAVES_N) * WM, wn0 = (wid % WAVES_N) * WN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int a_off = lh * BM + wm0 + l31 * TM, b_off = lh * BN + wn0 + l31 * TN;

    // fragment of k-step group grp (8 positions = 4 MFMA k-steps): lane (l31, lh) takes position 8*grp + 2*q + lh
    auto read_frag = [&](int buf, int grp, FVec<TM>(&a)[4], FVec<TN>(&b)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q].load(&Ds[buf][a_off + (grp * 8 + 2 * q) * BM]);
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q].load(&Xs[buf][b_off + (grp * 8 + 2 * q) * BN]);
    };
    auto mma = [&](const FVec<TM>(&a)[4], const FVec<TN>(&b)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].v[i], b[q].v[j], acc[i][j], 0, 0, 0);
    };
    auto group_order = [&]() {
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);            // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);  // MFMA
        __builtin_amdgcn_sched_barrier(0);
    };

    // bias gradient = column sums of dz: the first k-tile of every co-tile adds up the dz tile it stages anyway
    const bool do_bias = want_bias && (tile % tiles_n == 0) && tid < BM;
    float bsum = 0.f;

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    FVec<TM> a0[4], a1[4];
    FVec<TN> b0[4], b1[4];
    if (nk > 0) {
        load_tile();
        store_tile(0);
        load_tile();
        __syncthreads();
        read_frag(0, 0, a0, b0);

        auto slice = [&](auto cur_c) {
            constexpr int CUR = decltype(cur_c)::value;
            store_tile(CUR ^ 1);
            load_tile();
            if (do_bias) {
#pragma unroll 8
                for (int p = 0; p < BP; ++p) bsum += Ds[CUR][p * BM + tid];
            }
            __builtin_amdgcn_sched_barrier(0);
            read_frag(CUR, 1, a1, b1);
            mma(a0, b0);
            group_order();
            read_frag(CUR, 2, a0, b0);
            mma(a1, b1);
            group_order();
            read_frag(CUR, 3, a1, b1);
            mma(a0, b0);
            group_order();
            __syncthreads();
            read_frag(CUR ^ 1, 0, a0, b0);
            mma(a1, b1);
            group_order();
        };
        for (int kt = 0; kt + 1 < nk; kt += 2) {
            slice(I0());
            slice(I1());
        }
        if (nk & 1) slice(I0());
    }

    // partial layout per (member, split): [Cout*K weight partials | Cout bias partials]
    float* dst = out + ((size_t)blockIdx.y * gridDim.z + split) * ((size_t)g.Cout * K + g.Cout);
    if (do_bias && co0 + tid < g.Cout) dst[(size_t)g.Cout * K + co0 + tid] = bsum;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = j0 + wn0 + l31 * TN + j;
        if (col >= K) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = co0 + wm0 + ((r & 3) + 8 * (r >> 2) + 4 * lh) * TM + i;
                if (row < g.Cout) dst[(size_t)row * K + col] = acc[i][j][r];
            }
    }
}

// dw[i] (+)= sum_s part[s][i] for the Cout*K weight partials, dbias[c] (+)= sum_s part[s][Cout*K + c].
// L lanes cooperate on one output element (lane j adds splits j, j+L, ...; fixed-order butterfly combine), so the
// many-split / tiny-output layers (first convs: 512 splits of a 64x54 gradient) do not serialise 512 loads per thread.
template <int L>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                            float* __restrict__ dbias, size_t nw, int nb, int splits,
                                                            int accumulate, long long w_mstride, long long b_mstride) {
    // blockIdx.y = council member: its partial blocks follow the previous member's, its gradients live *_mstride floats on
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t i = t / L;
    const int j = (int)(t % L);
    const size_t stride = nw + (size_t)nb;
    const size_t n = nw + (dbias ? (size_t)nb : 0);
    part += (size_t)blockIdx.y * splits * stride;
    dw += (long long)blockIdx.y * w_mstride;
    if (dbias) dbias += (long long)blockIdx.y * b_mstride;
    float s = 0.f;
    if (i < n)
        for (int k = j; k < splits; k += L) s += part[(size_t)k * stride + i];
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (i < n && j == 0) {
        float* dst = i < nw ? dw + i : dbias + (i - nw);
        *dst = accumulate ? *dst + s : s;
    }
}

__device__ __forceinline__ void split_f16(float v, _Float16& h, _Float16& l);   // conv_x3.inc

// Weight re-layout for the data-gradient passes.  Entry z of the table moves one tap:
//   out[dst_base[z] + (ci * dst_T[z] + dst_tc[z]) * Cout + co] = w[(co * T + src_tap[z]) * Cin + ci0 + ci]
// (one 32x32 LDS-tiled transpose per tap and 32x32 (ci, co) patch)
struct TransTable {
    int32_t src_tap[CG_MAX_TAPS];
    int32_t dst_base[CG_MAX_TAPS];
    int32_t dst_tc[CG_MAX_TAPS];
    int32_t dst_T[CG_MAX_TAPS];
};
struct TransArgs {  // the kernel's kernarg layout (for the offset of the table)
    const float* w;
    float* out;
    int32_t Cout, T, Cin, ci0, nci;
    TransTable tt;
};
template <bool SPLIT>
__global__ __launch_bounds__(256) void weight_transpose_kernel(const float* __restrict__ w, float* __restrict__ out,
                                                               int Cout, int T, int Cin, int ci0, int nci, TransTable tt,
                                                               float scale, unsigned lo_elems, int nz, long long w_mstride,
                                                               long long out_mstride, const float* __restrict__ scale_dev) {
    // blockIdx.z = member * nz + table entry; member m reads w + m*w_mstride floats and writes out + m*out_mstride
    // (floats for the fp32 layout, fp16 ELEMENTS -- i.e. 4 bytes each in the interleaved {hi, lo} form -- when SPLIT)
    __shared__ float tile[32][33];
    typedef const __attribute__((address_space(4))) int32_t* KI;
    KI tab = (KI)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() +
                  offsetof(TransArgs, tt));
    const int member = blockIdx.z / nz;
    const int z = blockIdx.z - member * nz;
    w += (long long)member * w_mstride;
    out += (long long)member * out_mstride;
    if (SPLIT && scale_dev) scale *= scale_dev[0];
    const int tap = tab[z], base = tab[CG_MAX_TAPS + z], tc = tab[2 * CG_MAX_TAPS + z], Tc = tab[3 * CG_MAX_TAPS + z];
    const int cib = blockIdx.x * 32, cob = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        int co = cob + r, ci = cib + tx;
        tile[r][tx] = (co < Cout && ci < nci) ? w[((size_t)co * T + tap) * Cin + ci0 + ci] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int ci = cib + r, co = cob + tx;
        if (ci < nci && co < Cout) {
            const size_t o = (size_t)base + ((size_t)ci * Tc + tc) * Cout + co;
            if constexpr (SPLIT) {      // {hi, lo} fp16 planes of scale * w for the split-precision data-gradient
                _Float16 h, l;
                split_f16(tile[tx][r] * scale, h, l);
                reinterpret_cast<_Float16*>(out)[cg_il(o)] = h;
                reinterpret_cast<_Float16*>(out)[lo_elems + cg_il(o)] = l;
            } else {
                out[o] = tile[tx][r];
            }
        }
    }
}

// ---- opt-in launch timing ---------------------------------------------------------------------
struct ProfRec {
    int slot;
    double flops;
    hipEvent_t e0, e1;
    char key[96];
};
std::string prof_report;
std::mutex prof_mu;
bool prof_on = false;
std::vector<ProfRec> prof_recs;
char prof_names[CG_PROF_SLOTS][64];

int tile_id(int bm, int bn) {
    if (bm == 128 && bn == 128) return 0;
    if (bm == 128 && bn == 64) return 1;
    if (bm == 128 && bn == 32) return 2;
    if (bm == 64 && bn == 128) return 3;
    if (bm == 64 && bn == 64) return 4;
    if (bm == 32 && bn == 128) return 5;
    if (bm == 256 && bn == 128) return 6;
    if (bm == 256 && bn == 256) return 8;
    return 7;  // 128x32 / 64x64-class leftovers
}
struct ProfScope {
    bool active = false;
    ProfRec rec;
    hipStream_t st;
    ProfScope(int family, int bm, int bn, bool fast, double flops, hipStream_t s, const cg_conv_geom* g = nullptr,
              int ncls = 1)
        : st(s) {
        if (!prof_on) return;
        active = true;
        rec.key[0] = 0;
        if (g)
            snprintf(rec.key, sizeof(rec.key), "f%d %3dx%-3d N%-2d %3dx%-3d C%-3d->%-3d T%-2d s%d u%d out%dx%d x%d", family, bm,
                     bn, g->N, g->H, g->W, g->C1 + g->C2, g->Cout, g->T, g->stride, g->up, g->Ho, g->Wo, ncls);
        rec.slot = family * 20 + tile_id(bm, bn) * 2 + (fast ? 1 : 0);
        rec.flops = flops;
        static const char* const fam[10] = {"conv_fwd_kernel",     "conv_wgrad_kernel",     "conv_fwd_pipe_kernel",
                                           "conv_wgrad_pipe_kernel", "conv_fwd_x3_kernel",  "conv_wgrad_x3_kernel",
                                           "conv_fwd_x3w_kernel", "conv_wgrad_x3t_kernel", "conv_fwd_thin_kernel",
                                           "conv_wgrad_thin_kernel"};
        snprintf(prof_names[rec.slot], sizeof(prof_names[0]), "%s<%d,%d,%s>", fam[family], bm, bn, fast ? "fast" : "generic");
        (void)hipEventCreate(&rec.e0);
        (void)hipEventCreate(&rec.e1);
        (void)hipEventRecord(rec.e0, st);
    }
    ~ProfScope() {
        if (!active) return;
        (void)hipEventRecord(rec.e1, st);
        std::lock_guard<std::mutex> lk(prof_mu);
        prof_recs.push_back(rec);
    }
};

int validate_geom(const cg_conv_geom* g, const char* who) {
    CG_CHECK_ARG(g != nullptr, "%s: null geometry", who);
    CG_CHECK_ARG(g->T >= 1 && g->T <= CG_MAX_TAPS, "%s: T=%d out of range", who, g->T);
    CG_CHECK_ARG(g->N > 0 && g->H > 0 && g->W > 0 && g->C1 > 0 && g->C2 >= 0 && g->Cout > 0, "%s: bad dims", who);
    CG_CHECK_ARG(g->Ho > 0 && g->Wo > 0 && g->HoF > 0 && g->WoF > 0 && g->stride > 0, "%s: bad output dims", who);
    CG_CHECK_ARG(g->up == 0 || g->up == 1, "%s: up must be 0/1", who);
    const double in_elems = (double)g->N * g->H * g->W * (g->C1 + g->C2);
    const double out_elems = (double)g->N * g->HoF * g->WoF * g->Cout;
    CG_CHECK_ARG(in_elems < 2.0e9 && out_elems < 2.0e9, "%s: tensor exceeds 2^31 elements", who);
    return CG_OK;
}

// ---- kernel-selection table (cg_tuning, include/council_gan_hip.h) -------------------------------------------------
// The ONLY process-wide state of the library: which of several result-equivalent kernels / tile shapes a launch gets.
// It is read from the CG_* environment ONCE, at the first use, and afterwards changes only through cg_tuning_set()
// (or the single-field wrappers kept for the A/B tools).  Nothing in it changes what a call computes beyond the last
// bits; a host that launches from several threads sets it before the first launch.
static cg_tuning g_tune;
static std::once_flag g_tune_once;
static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
static void tuning_from_env() {
    cg_tuning t{};
    t.fwd_thin = env_int("CG_FWD_THIN", 1) != 0;
    t.wgrad_thin = env_int("CG_WGRAD_THIN", 1) != 0;
    const int bm = env_int("CG_WGRAD_X3_BM256", 2);
    t.wgrad_x3_bm256 = (bm == 0 || bm == 1) ? bm : 2;
    const int ww = env_int("CG_WGRAD_X3_WIDE", 2);
    t.wgrad_x3_wide = (ww == 0 || ww == 1) ? ww : 2;
    t.wgrad_x3_perm = getenv("CG_WGRAD_X3_PERM") != nullptr;
    t.wgrad_legacy = 0;
    t.x3_wide = env_int("CG_X3_WIDE", 16);
    const int to = env_int("CG_X3_THIN_OUT", 20);
    t.x3_thin_out = (to == 20 || to == 21) ? to : 0;
    const int ko = env_int("CG_X3_KORDER", 0);
    t.x3_korder = ko < 0 ? 0 : (ko > 2 ? 2 : ko);
    t.tile_rows_scale = env_int("CG_TILE_ROWS_SCALE", 1) < 1 ? 1 : env_int("CG_TILE_ROWS_SCALE", 1);
    t.no_amax_atomic = getenv("CG_NO_AMAX_ATOMIC") != nullptr;
    t.wgrad_x3_multitap = env_int("CG_WGRAD_X3_MULTITAP", 1) != 0;
    t.x3_cls_minor = env_int("CG_X3_CLS_MINOR", 1) != 0;
    t.x3_generic_epilogue = env_int("CG_X3_GENERIC_EPILOGUE", 0) != 0;
    t.wgrad_xcd_group = env_int("CG_WGRAD_XCD_GROUP", 1) != 0;
    t.fp32_chunked_sum = env_int("CG_FP32_CHUNKED_SUM", 1) != 0;
    g_tune = t;
}
static cg_tuning& tune() {
    std::call_once(g_tune_once, tuning_from_env);
    return g_tune;
}

// Per-launch option handed from the C entry point to the launcher that ends up running (set and cleared around one call on
// the calling thread): where the kernel's epilogue should leave its per-block output maxima, and how many it left.
struct FwdAmax {
    float* state = nullptr;
    int nslots = 0;
    bool all_classes = false;     // a data-gradient launch that carries every output-parity class: multi-class launches report too
};
static thread_local FwdAmax fwd_amax;
constexpr int CG_AMAX_SLOTS_MAX = 1024;     // == CG_AMAX_MAX_SLOTS of conv_x3.inc (state[2 .. 2 + 1024))
__global__ __launch_bounds__(256) void zero_slots_kernel(float* __restrict__ slots) { slots[blockIdx.x * 256 + threadIdx.x] = 0.f; }
__global__ void zero2_kernel(float* __restrict__ p) { if (threadIdx.x < 2) p[threadIdx.x] = 0.f; }
// slots a launch of `blocks` blocks fills (block_amax_store): one each, or 1024 shared ones that must start at zero
static int amax_slots_for(long blocks, float* state, hipStream_t st) {
    if (blocks <= CG_AMAX_SLOTS_MAX) return (int)blocks;
    if (tune().no_amax_atomic) return 0;      // A/B switch
    // a KERNEL, not hipMemsetAsync: a memset node captured into a hipGraph did not keep its place in front of the kernel that
    // fills the slots when the graph was replayed (measured: the consumer then saw zeroed maxima), a kernel node does
    hipLaunchKernelGGL(zero_slots_kernel, dim3(CG_AMAX_SLOTS_MAX / 256), dim3(256), 0, st, state + 2);
    return CG_AMAX_SLOTS_MAX;
}

// ---- member grouping on the host side -----------------------------------------------------------------------------
// cg_group (public): n members whose parameters sit `stride` fp32 ELEMENTS apart (weights, biases and their gradients
// alike: one pool per optimizer kind, optim.py).  A NULL group or n == 1 is an ordinary single-member call.
struct Grp {
    int n = 1;
    long long stride = 0;      // elements
};
static int grp_from(const cg_group* g, int N, Grp& out, const char* who) {
    out = Grp();
    if (!g) return CG_OK;
    CG_CHECK_ARG(g->n >= 1 && g->n <= 64, "%s: group of %d members (1..64)", who, g->n);
    CG_CHECK_ARG(N % g->n == 0, "%s: batch %d is not a multiple of the %d members", who, N, g->n);
    CG_CHECK_ARG(g->n == 1 || g->stride > 0, "%s: grouped launch needs the members' parameter stride", who);
    out.n = g->n;
    out.stride = g->n > 1 ? (long long)g->stride : 0;
    return CG_OK;
}
static Members members_f32(const Grp& gr) { return Members{gr.n, 0, gr.stride * 4, gr.stride * 4}; }

template <int BM, int BN, int WM, int WN, int STAGES>
int launch_fwd(const cg_conv_geom* g, const float* x1, const float* x2, const float* w, const float* bias, float* y,
               int M, int K, bool fast, hipStream_t st, const Grp& gr) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (g->Cout + BN - 1) / BN;     // M = rows of one member
    dim3 grid(tiles_m * tiles_n, 1, gr.n), block(NT);
    float* amax = nullptr;
    if (fwd_amax.state && g->osy == 1 && g->osx == 1) {
        fwd_amax.nslots = amax_slots_for((long)tiles_m * tiles_n * gr.n, fwd_amax.state, st);
        if (fwd_amax.nslots) amax = fwd_amax.state;
    }
    ProfScope prof(0, BM, BN, fast, 2.0 * (double)M * gr.n * (double)g->Cout * (double)K, st, g, gr.n);
    const Members mb = members_f32(gr);
    if (fast)
        hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WM, WN, true, STAGES>), grid, block, 0, st, *g, x1, x2, w, bias, y, M,
                           K, tiles_n, amax, mb);
    else
        hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WM, WN, false, STAGES>), grid, block, 0, st, *g, x1, x2, w, bias, y,
                           M, K, tiles_n, amax, mb);
    CG_LAUNCH_CHECK("conv_fwd_kernel");
    return CG_OK;
}

template <int BM, int BN, int WM, int WN, int PF = 1, int ABL = 0>
int launch_pipe_batch(PipeBatch& b, int ncls, const float* x1, const float* bias, float* y, unsigned x_bytes, hipStream_t st,
                      double* stats, const Members& mb) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    int max_tiles = 0;
    double flops = 0.0;
    for (int c = 0; c < ncls; ++c) {
        PipeClass& pc = b.c[c];
        pc.tiles_n = (pc.g.Cout + BN - 1) / BN;
        pc.ntiles = ((pc.M + BM - 1) / BM) * pc.tiles_n;
        if (pc.ntiles > max_tiles) max_tiles = pc.ntiles;
        flops += 2.0 * (double)pc.M * (double)pc.g.Cout * (double)pc.K * mb.n;
    }
    for (int c = ncls; c < 4; ++c) b.c[c].ntiles = 0;
    dim3 grid(max_tiles, ncls, mb.n), block(NT);
    ProfScope prof(2, BM, BN, true, flops, st, &b.c[0].g, ncls * mb.n);
    // (the chunked-sum variant of the 128 x 128 / 8-wave tile keeps ONE fragment register set so that its second accumulator set does not
    // cost the second block per CU -- 142 registers with two; held to 128 by launch bounds it spilled 96, prefetch distance 2 did not help)
    constexpr int PFF = PF;
    if (ABL == 0 && tune().fp32_chunked_sum)
        hipLaunchKernelGGL((conv_fwd_pipe_kernel<BM, BN, WM, WN, PFF, ABL, ABL == 0 ? 4 : 0>), grid, block, 0, st, b, x1, bias, y, x_bytes, stats, mb);
    else
        hipLaunchKernelGGL((conv_fwd_pipe_kernel<BM, BN, WM, WN, PF, ABL>), grid, block, 0, st, b, x1, bias, y, x_bytes, stats, mb);
    CG_LAUNCH_CHECK("conv_fwd_pipe_kernel");
    return CG_OK;
}

// configurations whose launch forwards the statistics pointer to the kernel
bool pipe_cfg_has_stats(int cfg) { return (cfg >= 20 && cfg <= 26) || cfg == 32; }
int pipe_cfg_bm(int cfg) { return cfg == 23 || cfg == 26 ? 64 : (cfg == 24 || cfg == 32 ? 256 : 128); }

int launch_pipe_cfg(int cfg, PipeBatch& b, int ncls, const float* x1, const float* bias, float* y, unsigned x_bytes,
                    hipStream_t st, double* stats, const Members& mb) {
    switch (cfg) {
        case 20: return launch_pipe_batch<128, 128, 64, 32>(b, ncls, x1, bias, y, x_bytes, st, stats, mb);  // 8 waves
        case 21: return launch_pipe_batch<128, 128, 64, 64>(b, ncls, x1, bias, y, x_bytes, st, stats, mb);  // 4 waves
        case 22: return launch_pipe_batch<128, 64, 64, 32>(b, ncls, x1, bias, y, x_bytes, st, stats, mb);   // 4 waves
        case 23: return launch_pipe_batch<64, 64, 32, 32>(b, ncls, x1, bias, y, x_bytes, st, stats, mb);    // 4 waves
        case 24: return launch_pipe_batch<256, 128, 64, 64>(b, ncls, x1, bias, y, x_bytes, st, stats, mb);  // 8 waves
        case 25: return launch_pipe_batch<128, 64, 32, 32>(b, ncls, x1, bias, y, x_bytes, st, stats, mb);   // 8 waves
        case 26: return launch_pipe_batch<64, 128, 32, 64>(b, ncls, x1, bias, y, x_bytes, st, stats, mb);   // 4 waves
        case 31: return launch_pipe_batch<128, 128, 64, 32, 1, 3>(b, ncls, x1, bias, y, x_bytes, st, nullptr, mb);  // timing probe
        case 32: return launch_pipe_batch<256, 64, 64, 32>(b, ncls, x1, bias, y, x_bytes, st, stats, mb);  // 8 waves
        default: return cg_set_error(CG_ERR_ARG, "pipelined conv: unknown tile configuration %d", cfg);
    }
}

// one class of a (possibly grouped) launch: M = output rows of ONE member
void fill_class(PipeClass& pc, const cg_conv_geom* g, const float* w, int nmember = 1) {
    pc.g = *g;
    pc.w = w;
    pc.M = (g->N / nmember) * g->Ho * g->Wo;
    pc.K = g->T * g->C1;
    pc.w_bytes = (unsigned)((size_t)g->Cout * pc.K * sizeof(float));
    pc.pad_ = 0;
}

#include "conv_x3.inc"

// {hi, lo} operand bookkeeping under the build's layout (cg_common.h): is the caller's lo offset valid for a tensor of
// `plane_bytes` bytes per plane, and how many bytes do the two planes span from the hi pointer
inline bool x3_lo_ok(size_t lo_elems, size_t plane_bytes) {
    return CG_X3_INTERLEAVE ? (lo_elems == CG_X3_LO_ELEMS && plane_bytes % 64 == 0) : lo_elems * 2 >= plane_bytes;
}
inline size_t x3_span(size_t lo_elems, size_t plane_bytes) {
    return CG_X3_INTERLEAVE ? 2 * plane_bytes : lo_elems * 2 + plane_bytes;
}

// the pipelined kernel needs: one source, channels a multiple of BK, operands addressable with 31-bit byte offsets
bool pipe_ok(const cg_conv_geom* g, int K) {
    return g->C2 == 0 && g->C1 % BK == 0 && (size_t)g->N * g->H * g->W * g->C1 * sizeof(float) < (size_t)CG_OOB &&
           (size_t)g->Cout * K * sizeof(float) < (size_t)CG_OOB;
}

// ---- thin-input layers (conv_fwd_thin_kernel) -----------------------------------------------------------------------
// Layers that match a compiled (KH, KW, channels, stride) variant run on the spatial-tile kernel (tile configuration 40);
// on by default since round 3 (measured in the step), CG_FWD_THIN=0 in the environment / cg_conv2d_fwd_thin(0) turns it off.
static bool fwd_thin_on() { return tune().fwd_thin != 0; }

struct ThinVariant { int kh, kw, ct, s; };
static const ThinVariant thin_variants[] = {{7, 7, 3, 1}, {4, 4, 3, 2}, {3, 3, 6, 1}, {3, 3, 3, 1}, {1, 1, 12, 1}};

// index into thin_variants, or -1
static int thin_match(const cg_conv_geom* g) {
    if (g->up != 0 || g->osy != 1 || g->osx != 1 || g->ooy != 0 || g->oox != 0 || g->HoF != g->Ho || g->WoF != g->Wo) return -1;
    if (g->Cout != 64) return -1;
    const int ct = g->C1 + g->C2;
    for (int v = 0; v < (int)(sizeof(thin_variants) / sizeof(thin_variants[0])); ++v) {
        const ThinVariant& tv = thin_variants[v];
        if (g->T != tv.kh * tv.kw || ct != tv.ct || g->stride != tv.s) continue;
        bool raster = true;
        for (int t = 0; t < g->T && raster; ++t)
            raster = g->dy[t] == g->dy[0] + t / tv.kw && g->dx[t] == g->dx[0] + t % tv.kw;
        if (raster) return v;
    }
    return -1;
}

template <int KH, int KW, int CT, int S>
int launch_fwd_thin(const cg_conv_geom* g, const float* x1, const float* x2, const float* w, const float* bias, float* y, int M,
                    hipStream_t st, const Grp& gr) {
    const int imgs = M / (g->Ho * g->Wo);                       // images of ONE member
    const int ntiles = imgs * ((g->Ho + 15) / 16) * ((g->Wo + 15) / 16);
    int gx = 512 / gr.n;                                        // two resident blocks per CU over all members
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    dim3 grid(gx, 1, gr.n), block(512);
    float* amax = nullptr;
    if (fwd_amax.state) {
        fwd_amax.nslots = amax_slots_for((long)gx * gr.n, fwd_amax.state, st);
        if (fwd_amax.nslots) amax = fwd_amax.state;
    }
    ProfScope prof(8, 256, 64, true, 2.0 * (double)M * gr.n * 64.0 * (double)(KH * KW * CT), st, g, gr.n);
    hipLaunchKernelGGL((conv_fwd_thin_kernel<KH, KW, CT, S>), grid, block, 0, st, *g, x1, x2, w, bias, y, imgs, -(int)g->dy[0],
                       -(int)g->dx[0], amax, members_f32(gr));
    CG_LAUNCH_CHECK("conv_fwd_thin_kernel");
    return CG_OK;
}

// ------------------------------------------------------------------------------------------
// conv_fwd_thin_x3_kernel (round 6): the THIN-input layers (3 / 6 / 12 -> 64 channels) of the split-precision datapath on the fp16 MFMA.
// Same spatial blocking, persistent tile walk, LDS patch and 16-byte stores as conv_fwd_thin_kernel, but the K = taps x channels
// products are evaluated as  wh*ph + wh*pl + wl*ph  on v_mfma_f32_32x32x16_f16 (K padded to a multiple of 16 with zero weights):
// 12 MFMAs of 32 cycles per 32x32 output tile for the 3x3x6 layer instead of 27 fp32 MFMAs of 64 -- the fp32 kernel spends half
// its time in the matrix pipe, this one is bound by its 256 output bytes per pixel.
//   * the patch is stored in LDS as ONE dword per element, {hi half | lo half << 16} of the fp32 value (converted once, when the
//     tile is staged): a lane gathers the 8 k's of its fragment with 8 ds_read_b32 at offsets it reads from a small LDS table
//     (k -> patch offset, computed once per block) and separates the planes with two v_perm per dword pair;
//   * this lane's rows of the weight matrix, both planes, stay in registers for the block's whole tile list (K16 / 2 VGPRs),
//     pre-multiplied by CG_X3_WSCALE like every split weight.
// Operands as in conv_fwd_thin_kernel after round 6's swap: A = weights (D rows = output channels), B = patch (D columns = pixels).
// ------------------------------------------------------------------------------------------
// (144-184 registers: one block per CU; held to 128 registers for a second block the 3x3x6 / 4x4x3 variants spill and lose more than
// the second block gains -- 453 vs 333 us on the council discriminator's first layer, gpurun_out/s15)
template <int KH, int KW, int CT, int S>
__global__ __launch_bounds__(512) void conv_fwd_thin_x3_kernel(
    cg_conv_geom g, const float* __restrict__ x1, const float* __restrict__ x2, const float* __restrict__ w,
    const float* __restrict__ bias, float* __restrict__ y, int imgs_per_member, int pad_y, int pad_x,
    float* __restrict__ amax_state, Members mb) {
    constexpr int TH = 16, TW = 16, NT = 512, BN = 64;
    constexpr int K = KH * KW * CT, RL = KW * CT, K16 = (K + 15) / 16 * 16, KSTEPS = K16 / 16;
    constexpr int PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW, PN = PH * PW * CT;
    constexpr int PV = (PN + NT - 1) / NT;                      // patch elements per thread
    __shared__ unsigned patch[2][PN];                           // {hi | lo << 16} per element
    __shared__ __attribute__((aligned(16))) int koffs[K16];     // k -> offset inside a patch (k >= K: 0, its weight is zero)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    w = reinterpret_cast<const float*>(reinterpret_cast<const char*>(w) + (long long)blockIdx.z * mb.w_stride);
    if (bias) bias = reinterpret_cast<const float*>(reinterpret_cast<const char*>(bias) + (long long)blockIdx.z * mb.b_stride);

    for (int k = tid; k < K16; k += NT) koffs[k] = k < K ? (k / RL) * (PW * CT) + (k % RL) : 0;

    // this lane's row of the weight matrix (output channel wn * 32 + l31), k = 16 ks + 8 lh + e, as {hi, lo} halves of 1024 w
    cg_f16x8 wh[KSTEPS], wl[KSTEPS];
    {
        const float* wrow = w + (size_t)(wn * 32 + l31) * K;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = ks * 16 + lh * 8 + e;
                const float v = k < K ? wrow[k] * CG_X3_WSCALE : 0.f;
                _Float16 h, l;
                split_f16(v, h, l);
                wh[ks][e] = h;
                wl[ks][e] = l;
            }
        }
    }
    // this lane's 16 output channels: wn * 32 + 8 q + 4 lh + {0..3}, q = 0..3 (the D rows of its half-wave)
    float4 bj4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float* bq = bias + wn * 32 + 8 * q + 4 * lh;
        bj4[q] = bias ? make_float4(bq[0], bq[1], bq[2], bq[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        cg_touch(bj4[q].x);
        cg_touch(bj4[q].y);
        cg_touch(bj4[q].z);
        cg_touch(bj4[q].w);
    }
    // LDS offsets of this lane's two pixel columns: column r of the tile is pixel (r / 16, r % 16)
    int rb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = wm * 64 + i * 32 + l31;
        rb[i] = (((r >> 4) * S) * PW + (r & 15) * S) * CT;
    }

    const int tiles_x = (g.Wo + TW - 1) / TW, tiles_y = (g.Ho + TH - 1) / TH;
    const int tiles_img = tiles_x * tiles_y, ntiles = imgs_per_member * tiles_img;
    const int img0 = (int)blockIdx.z * imgs_per_member;
    float vmax = 0.f;
    float pv[PV];
    auto fetch_patch = [&](int t) {
        const int n = img0 + t / tiles_img, tr = t % tiles_img;
        const int iy0 = (tr / tiles_x) * TH * S - pad_y, ix0 = (tr % tiles_x) * TW * S - pad_x;
#pragma unroll
        for (int j = 0; j < PV; ++j) {
            const int e = tid + j * NT;
            const int pix = e / CT, c = e - pix * CT;
            const int py = pix / PW, px = pix - py * PW;
            const int iy = iy0 + py, ix = ix0 + px;
            float v = 0.f;
            if (e < PN && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W) {
                const size_t p = ((size_t)n * g.H + iy) * g.W + ix;
                v = c < g.C1 ? x1[p * g.C1 + c] : x2[p * g.C2 + (c - g.C1)];
            }
            pv[j] = v;
        }
    };
    auto store_patch = [&](int buf) {
#pragma unroll
        for (int j = 0; j < PV; ++j) {
            const int e = tid + j * NT;
            if (e < PN) {
                _Float16 h, l;
                split_f16(pv[j], h, l);
                patch[buf][e] = (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
            }
        }
    };
    int cur = 0;
    if ((int)blockIdx.x < ntiles) {
        fetch_patch(blockIdx.x);
        store_patch(0);
    }
    __syncthreads();
    const float inv_scale = 1.f / CG_X3_WSCALE;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int n = img0 + t / tiles_img, tr = t % tiles_img;
        const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * TW;
        const bool more = t + (int)gridDim.x < ntiles;
        if (more) fetch_patch(t + gridDim.x);                   // global loads in flight under the MFMAs below
        const unsigned* __restrict__ pt = patch[cur];

        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int4 o0 = *reinterpret_cast<const int4*>(&koffs[ks * 16 + lh * 8]);
            const int4 o1 = *reinterpret_cast<const int4*>(&koffs[ks * 16 + lh * 8 + 4]);
            const int off[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                unsigned d[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) d[e] = pt[rb[i] + off[e]];
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                u32x4 ph, pl;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ph[q] = __builtin_amdgcn_perm(d[2 * q + 1], d[2 * q], 0x05040100u);      // the two hi halves
                    pl[q] = __builtin_amdgcn_perm(d[2 * q + 1], d[2 * q], 0x07060302u);      // the two lo halves
                }
                // small terms first, the dominant hi*hi product last (as every split-precision kernel)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks], __builtin_bit_cast(cg_f16x8, ph), acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], __builtin_bit_cast(cg_f16x8, pl), acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], __builtin_bit_cast(cg_f16x8, ph), acc[i], 0, 0, 0);
            }
        }

        // C/D layout of the 32x32 MFMA: col = lane % 32 (pixel), row = (r & 3) + 8 * (r >> 2) + 4 * (lane / 32) (channel)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wm * 64 + i * 32 + l31;
            const int oy = oy0 + (row >> 4), ox = ox0 + (row & 15);
            if (oy < g.Ho && ox < g.Wo) {
                float* __restrict__ dst = y + (((size_t)n * g.Ho + oy) * g.Wo + ox) * BN + wn * 32 + 4 * lh;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v;
                    v.x = cg_apply_act(acc[i][4 * q + 0] * inv_scale + bj4[q].x, g.act);
                    v.y = cg_apply_act(acc[i][4 * q + 1] * inv_scale + bj4[q].y, g.act);
                    v.z = cg_apply_act(acc[i][4 * q + 2] * inv_scale + bj4[q].z, g.act);
                    v.w = cg_apply_act(acc[i][4 * q + 3] * inv_scale + bj4[q].w, g.act);
                    *reinterpret_cast<float4*>(dst + 8 * q) = v;
                    vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                }
            }
        }
        if (more) store_patch(cur ^ 1);                         // last read by the previous tile, released by its barrier
        __syncthreads();
        cur ^= 1;
    }
    if (amax_state) block_amax_store<NT>(vmax, amax_state);
}

template <int KH, int KW, int CT, int S>
int launch_fwd_thin_x3(const cg_conv_geom* g, const float* x1, const float* x2, const float* w, const float* bias, float* y, int M,
                       hipStream_t st, const Grp& gr) {
    const int imgs = M / (g->Ho * g->Wo);                       // images of ONE member
    const int ntiles = imgs * ((g->Ho + 15) / 16) * ((g->Wo + 15) / 16);
    int gx = 512 / gr.n;                                        // two resident blocks per CU over all members
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    dim3 grid(gx, 1, gr.n), block(512);
    float* amax = nullptr;
    if (fwd_amax.state) {
        fwd_amax.nslots = amax_slots_for((long)gx * gr.n, fwd_amax.state, st);
        if (fwd_amax.nslots) amax = fwd_amax.state;
    }
    ProfScope prof(8, 256, 64, true, 2.0 * (double)M * gr.n * 64.0 * (double)(KH * KW * CT), st, g, gr.n);
    hipLaunchKernelGGL((conv_fwd_thin_x3_kernel<KH, KW, CT, S>), grid, block, 0, st, *g, x1, x2, w, bias, y, imgs, -(int)g->dy[0],
                       -(int)g->dx[0], amax, members_f32(gr));
    CG_LAUNCH_CHECK("conv_fwd_thin_x3_kernel");
    return CG_OK;
}

int launch_fwd_thin_x3_variant(int v, const cg_conv_geom* g, const float* x1, const float* x2, const float* w, const float* bias,
                               float* y, int M, hipStream_t st, const Grp& gr) {
    switch (v) {
        case 0: return launch_fwd_thin_x3<7, 7, 3, 1>(g, x1, x2, w, bias, y, M, st, gr);
        case 1: return launch_fwd_thin_x3<4, 4, 3, 2>(g, x1, x2, w, bias, y, M, st, gr);
        case 2: return launch_fwd_thin_x3<3, 3, 6, 1>(g, x1, x2, w, bias, y, M, st, gr);
        case 3: return launch_fwd_thin_x3<3, 3, 3, 1>(g, x1, x2, w, bias, y, M, st, gr);
        case 4: return launch_fwd_thin_x3<1, 1, 12, 1>(g, x1, x2, w, bias, y, M, st, gr);
        default: return cg_set_error(CG_ERR_ARG, "thin conv (split precision): no such variant");
    }
}

int launch_fwd_thin_variant(int v, const cg_conv_geom* g, const float* x1, const float* x2, const float* w, const float* bias,
                            float* y, int M, hipStream_t st, const Grp& gr) {
    switch (v) {
        case 0: return launch_fwd_thin<7, 7, 3, 1>(g, x1, x2, w, bias, y, M, st, gr);
        case 1: return launch_fwd_thin<4, 4, 3, 2>(g, x1, x2, w, bias, y, M, st, gr);
        case 2: return launch_fwd_thin<3, 3, 6, 1>(g, x1, x2, w, bias, y, M, st, gr);
        case 3: return launch_fwd_thin<3, 3, 3, 1>(g, x1, x2, w, bias, y, M, st, gr);
        case 4: return launch_fwd_thin<1, 1, 12, 1>(g, x1, x2, w, bias, y, M, st, gr);
        default: return cg_set_error(CG_ERR_ARG, "thin conv: no such variant");
    }
}

// weight gradient of the thin-input layers (conv_wgrad_thin_kernel): on unless CG_WGRAD_THIN=0 / cg_conv2d_wgrad_thin(0)
// (1.6 ... 3.3x the generic kernel on the step's shapes, profiles/r02_i_ab_optin.txt)
static bool wgrad_thin_on() { return tune().wgrad_thin != 0; }
// blocks (= splits) per member of a thin weight-gradient launch; M = rows of one member
static int thin_wgrad_splits(const cg_conv_geom* g, int nmember) {
    const int imgs = g->N / nmember;
    const int ntiles = imgs * ((g->Ho + 15) / 16) * ((g->Wo + 15) / 16);
    // one resident block per CU over all members (two -- held to 128 registers -- spill and gain nothing: 793 vs 807 us on the council
    // discriminator's first layer, the other variants slower, gpurun_out/s20)
    int gx = 256 / nmember;
    if (gx < 1) gx = 1;
    return gx < ntiles ? gx : ntiles;
}

template <int KH, int KW, int CT, int S>
int launch_wgrad_thin(const cg_conv_geom* g, const float* x1, const float* x2, const float* dz, float* part, int want_bias,
                      hipStream_t st, int nmember, const float* yact, int act) {
    const int imgs = g->N / nmember;
    dim3 grid(thin_wgrad_splits(g, nmember), 1, nmember), block(512);
    ProfScope prof(9, 256, 64, true, 2.0 * (double)g->N * g->Ho * g->Wo * 64.0 * (double)(KH * KW * CT), st, g, nmember);
    hipLaunchKernelGGL((conv_wgrad_thin_kernel<KH, KW, CT, S>), grid, block, 0, st, *g, x1, x2, dz, part, imgs, -(int)g->dy[0],
                       -(int)g->dx[0], want_bias, yact, act);
    CG_LAUNCH_CHECK("conv_wgrad_thin_kernel");
    return CG_OK;
}

int launch_wgrad_thin_variant(int v, const cg_conv_geom* g, const float* x1, const float* x2, const float* dz, float* part,
                              int want_bias, hipStream_t st, int nmember, const float* yact = nullptr, int act = 0) {
    switch (v) {
        case 0: return launch_wgrad_thin<7, 7, 3, 1>(g, x1, x2, dz, part, want_bias, st, nmember, yact, act);
        case 1: return launch_wgrad_thin<4, 4, 3, 2>(g, x1, x2, dz, part, want_bias, st, nmember, yact, act);
        case 2: return launch_wgrad_thin<3, 3, 6, 1>(g, x1, x2, dz, part, want_bias, st, nmember, yact, act);
        case 3: return launch_wgrad_thin<3, 3, 3, 1>(g, x1, x2, dz, part, want_bias, st, nmember, yact, act);
        case 4: return launch_wgrad_thin<1, 1, 12, 1>(g, x1, x2, dz, part, want_bias, st, nmember, yact, act);
        default: return cg_set_error(CG_ERR_ARG, "thin weight gradient: no such variant");
    }
}

// tile configurations of the forward kernel (id -> BM, BN, WM, WN, STAGES); M = rows of one member
int launch_fwd_cfg(int cfg, const cg_conv_geom* g, const float* x1, const float* x2, const float* w, const float* bias,
                   float* y, int M, int K, bool fast, hipStream_t st, double* stats, const Grp& gr) {
#define FW(...) return launch_fwd<__VA_ARGS__>(g, x1, x2, w, bias, y, M, K, fast, st, gr)
    switch (cfg) {
        case 0: FW(128, 128, 64, 64, 1);
        case 1: FW(128, 64, 64, 32, 1);
        case 2: FW(128, 32, 32, 32, 1);
        case 3: FW(64, 64, 32, 32, 1);
        case 4: FW(128, 128, 64, 64, 2);
        case 5: FW(128, 64, 64, 32, 2);
        case 6: FW(128, 128, 64, 32, 1);   // 8 waves
        case 7: FW(128, 128, 64, 32, 2);   // 8 waves
        case 8: FW(64, 128, 32, 64, 1);
        case 9: FW(64, 128, 32, 64, 2);
        case 10: FW(64, 64, 32, 32, 2);
        case 11: FW(256, 128, 64, 64, 1);  // 8 waves
        case 12: FW(256, 128, 64, 64, 2);  // 8 waves
        case 13: FW(128, 32, 32, 32, 2);
        case 14: FW(128, 64, 32, 32, 1);   // 8 waves
        case 15: FW(256, 64, 64, 32, 1);   // 8 waves
        case 16: FW(64, 128, 32, 32, 1);   // 8 waves
        case 17: FW(128, 128, 32, 32, 1);  // 16 waves
        case 18: FW(256, 128, 64, 32, 1);  // 16 waves
        case 20: case 21: case 22: case 23: case 24: case 25: case 26: case 31: case 32: {
            if (!pipe_ok(g, K)) return cg_set_error(CG_ERR_ARG, "conv forward: configuration %d needs the pipelined path", cfg);
            PipeBatch b;
            fill_class(b.c[0], g, w, gr.n);
            return launch_pipe_cfg(cfg, b, 1, x1, bias, y, (unsigned)((size_t)g->N * g->H * g->W * g->C1 * sizeof(float)), st,
                                   stats, members_f32(gr));
        }
        case 40: {
            const int v = thin_match(g);
            if (v < 0) return cg_set_error(CG_ERR_ARG, "conv forward: configuration 40 needs a thin-input layer (3/6/12 -> 64 channels)");
            return launch_fwd_thin_variant(v, g, x1, x2, w, bias, y, M, st, gr);
        }
        case 41: {     // the same layers on the fp16 x 3 MFMA (cg_conv2d_fwd_thin_x3_g)
            const int v = thin_match(g);
            if (v < 0) return cg_set_error(CG_ERR_ARG, "conv forward: configuration 41 needs a thin-input layer (3/6/12 -> 64 channels)");
            return launch_fwd_thin_x3_variant(v, g, x1, x2, w, bias, y, M, st, gr);
        }
        default: return cg_set_error(CG_ERR_ARG, "conv forward: unknown tile configuration %d", cfg);
    }
#undef FW
}

// Measured on MI355X (profiles/r01_conv_tiles.txt): 8-wave 128x128 blocks (two waves per SIMD hide each
// other's barrier / LDS phases) reach 112-123 TFLOP/s once >= ~192 such tiles exist; problems with
// fewer tiles fill the 256 CUs better with 64x64 tiles (two LDS stages when very few tiles).  M = rows of the LAUNCH.
int pick_fwd_cfg(const cg_conv_geom* g, long M, bool pipe) {
    if (fwd_thin_on() && thin_match(g) >= 0) return 40;
    M *= tune().tile_rows_scale;      // test hook: choose tiles as if the launch had k x the rows (cg_tuning.tile_rows_scale)
    const long blocks128 = ((M + 127) / 128) * ((g->Cout + 127) / 128);
    if (g->Cout > 64) {
        // (chunked sums, cg_tuning.fp32_chunked_sum: the 128 x 128 / 8-wave tile keeps its two blocks per CU with ONE fragment register set,
        // 128 registers; preferring the 256 x 128 tile for launches that fill the chip measured slower -- 133.6 vs 131.3 ms per step)
        if (blocks128 >= 192) return pipe ? 20 : 6;
        if (pipe) return 23;
        return blocks128 < 96 ? 10 : 3;
    }
    if (g->Cout > 32) {
        // 64 output channels: 8 waves of 32x32 on a 128x64 tile beat 4 waves of 64x32 by 4-6 % (two waves per SIMD);
        // short-K (1x1) layers are bandwidth-bound and prefer the smaller tile (profiles/r01_conv_tiles_pipe.txt)
        if ((M + 127) / 128 < 192) return pipe ? 23 : 3;
        if (pipe) return g->T * g->C1 <= 128 ? 23 : 25;
        return 1;
    }
    return 2;
}

static bool wgrad_force_legacy() { return tune().wgrad_legacy != 0; }  // A/B switch (cg_conv2d_wgrad_legacy)

struct WgradPlan {
    int bm, bn;
    bool fast;
    int tiles_m, tiles_n, splits, slices_per_split;
};

// Split plan of ONE member's weight gradient (M = its output rows).  A grouped launch runs the members' identical plans
// side by side (grid.y), so a member's result does not depend on how many members share the launch; with several members
// fewer splits per member already fill the chip.
// CG_WGRAD_X3_BM256 (environment) / cg_conv2d_wgrad_x3_bm256(): the split-precision weight gradient of layers with
// Cout % 256 == 0 and a 128-wide K-tile on a 256 x 128 tile / 16 waves (the shape that pays for the forward kernel from
// two tiles per CU) instead of 128 x 128 / 8 waves.  0 = never, 1 = wherever the layer qualifies, unset / 2 = where it was
// measured to win (profiles/r02_i_ab_optin.txt: +15...25 % from 64 such tiles over all members, -4 % below).
static int wgrad_x3_bm256() { return tune().wgrad_x3_bm256; }

// CG_WGRAD_X3_WIDE / cg_conv2d_wgrad_x3_wide(): the 256 x 256 LDS-DMA tile (conv_wgrad_x3tw_kernel) for layers with
// Cout % 256 == 0 and C1 % 256 == 0: 0 = never, 1 = wherever the layer qualifies, unset / 2 = where it was measured to win
// (profiles/r03_h_ab_wgrad_wide.txt, with one block per CU's worth of splits: +14 % on the member-batched res-block shape,
// +16 % on 256 -> 512 4x4 s2, +20 % on 512 -> 512 1x1 against the best of the other tiles, reduce included -- from 16 such
// tiles over all members; -5 % on a single member's res-block launch with 9)
static int wgrad_x3_wide() { return tune().wgrad_x3_wide; }

WgradPlan plan_wgrad(const cg_conv_geom* g, int nmember = 1, bool x3 = false) {
    // Every instantiated tile has exactly four 32x32 MFMA wave tiles or more (4 waves / block):
    //   bm=128: bn in {128, 64, 32};  bm=64: bn in {128, 64};  bm=32: bn = 128.
    // FAST (float4 gather, one tap per k-tile) needs a single source and Ct % bn == 0.
    WgradPlan p;
    const int Ct = g->C1 + g->C2;
    const int K = g->T * Ct;
    const int M = (g->N / nmember) * g->Ho * g->Wo;
    p.bm = g->Cout > 64 ? 128 : (g->Cout > 32 ? 64 : 32);
    const int bn_min = p.bm == 128 ? 32 : (p.bm == 64 ? 64 : 128);
    p.fast = false;
    p.bn = 0;
    if (g->C2 == 0) {
        for (int bn = 128; bn >= bn_min; bn >>= 1)
            if (Ct % bn == 0) { p.bn = bn; p.fast = true; break; }
    }
    if (!p.fast) {
        p.bn = bn_min;
        while (p.bn < 128 && p.bn < K) p.bn <<= 1;
        if (p.bn > 128) p.bn = 128;
    }
    // split-precision plans on the transposing-read kernel: a K-tile may span several taps (every 32-channel group of the tile
    // carries its own tap), so layers with 32 / 64 input channels take 128-wide tiles too -- twice / four times the MFMAs per
    // staged dz row (profiles/r03_p_ab_wgrad_multitap.txt); columns past K read zeros and are not written
    if (x3 && tune().wgrad_x3_multitap && CG_X3_INTERLEAVE && !tune().wgrad_x3_perm && p.fast && p.bn < 128 &&
        (Ct == 32 || Ct == 64) && K >= 128)
        p.bn = 128;
    if (x3 && wgrad_x3_bm256() && CG_X3_INTERLEAVE && p.fast && p.bm == 128 && p.bn == 128 && g->Cout % 256 == 0) {
        const long tiles256 = (long)(g->Cout / 256) * ((K + 127) / 128) * nmember;
        if (wgrad_x3_bm256() == 1 || tiles256 >= 64) p.bm = 256;
    }
    if (x3 && wgrad_x3_wide() && CG_X3_INTERLEAVE && p.fast && g->Cout % 256 == 0 && Ct % 256 == 0) {
        const long tiles = (long)(g->Cout / 256) * (K / 256) * nmember;
        if (wgrad_x3_wide() == 1 || tiles >= 16) {
            p.bm = 256;
            p.bn = 256;
        }
    }
    p.tiles_m = (g->Cout + p.bm - 1) / p.bm;
    p.tiles_n = (K + p.bn - 1) / p.bn;
    const int slices = (M + 31) / 32;
    const int tiles = p.tiles_m * p.tiles_n * nmember;
    int want = (2 * 256) / tiles;                      // two co-resident blocks per CU, and no partial second round
    if (p.bm == 256 && p.bn == 256) want = 256 / tiles;   // the 256 x 256 LDS-DMA tile: one block per CU (136 KiB of LDS) -- and
    if (want < 1) want = 1;                               // half the partial sums for the reduce kernel to add up
    if (K <= 128 && g->Cout <= 128) want *= 4;         // 1x1-class gradients are bandwidth-bound: more loads in flight
    int max_by_work = slices / 8 > 0 ? slices / 8 : 1; // >= 8 slices (256 positions) per split
    int s = want < max_by_work ? want : max_by_work;
    if (s < 1) s = 1;
    if (s > 2048) s = 2048;
    p.slices_per_split = (slices + s - 1) / s;
    p.splits = (slices + p.slices_per_split - 1) / p.slices_per_split;
    return p;
}

template <int BM, int BN, int WM, int WN>
int launch_wgrad(const cg_conv_geom* g, const WgradPlan& p, const float* x1, const float* x2, const float* dz,
                 float* out, int M, int K, int want_bias, hipStream_t st, int nmember) {
    dim3 grid(p.tiles_m * p.tiles_n, nmember, p.splits), block((BM / WM) * (BN / WN) * 64);
    ProfScope prof(1, BM, BN, p.fast, 2.0 * (double)M * nmember * (double)g->Cout * (double)K, st, g, nmember);
    if (p.fast)
        hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, WM, WN, true>), grid, block, 0, st, *g, x1, x2, dz, out, M, K,
                           p.tiles_n, p.slices_per_split, want_bias);
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, WM, WN, false>), grid, block, 0, st, *g, x1, x2, dz, out, M, K,
                           p.tiles_n, p.slices_per_split, want_bias);
    CG_LAUNCH_CHECK("conv_wgrad_kernel");
    return CG_OK;
}

int ilog2_exact(int v) {  // log2 of a power of two, -1 otherwise
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

bool wgrad_pipe_ok(const cg_conv_geom* g, const WgradPlan& p) {
    if (!p.fast || g->C2 != 0 || (g->Cout & 3)) return false;
    if (ilog2_exact(g->Ho * g->Wo) < 0 || ilog2_exact(g->Wo) < 0) return false;
    if (!((p.bm == 128 && p.bn == 128) || (p.bm == 128 && p.bn == 64) || (p.bm == 64 && p.bn == 64) ||
          (p.bm == 64 && p.bn == 128) || (p.bm == 256 && p.bn == 128) || (p.bm == 256 && p.bn == 256)))     // 256-row tiles: split-precision plans only
        return false;
    return (size_t)g->N * g->H * g->W * g->C1 * sizeof(float) < (size_t)CG_OOB &&
           (size_t)g->N * g->Ho * g->Wo * g->Cout * sizeof(float) < (size_t)CG_OOB;
}

template <int BM, int BN, int WM, int WN>
int launch_wgrad_pipe(const cg_conv_geom* g, const WgradPlan& p, const float* x1, const float* dz, float* out, int M, int K,
                      int want_bias, hipStream_t st, int nmember) {
    dim3 grid(p.tiles_m * p.tiles_n, nmember, p.splits), block((BM / WM) * (BN / WN) * 64);
    ProfScope prof(3, BM, BN, true, 2.0 * (double)M * nmember * (double)g->Cout * (double)K, st, g, nmember);
    hipLaunchKernelGGL((conv_wgrad_pipe_kernel<BM, BN, WM, WN>), grid, block, 0, st, *g, x1, dz, out, M, K, p.tiles_n,
                       p.slices_per_split, want_bias, ilog2_exact(g->Ho * g->Wo), ilog2_exact(g->Wo),
                       (unsigned)((size_t)g->N * g->H * g->W * g->C1 * sizeof(float)),
                       (unsigned)((size_t)g->N * g->Ho * g->Wo * g->Cout * sizeof(float)));
    CG_LAUNCH_CHECK("conv_wgrad_pipe_kernel");
    return CG_OK;
}

// partials of (member, split) -> the members' gradient tensors, fixed order
int launch_splitk_reduce(const float* part, float* dw, float* dbias, size_t nw, int Cout, int splits, int accumulate,
                         const Grp& gr, hipStream_t st) {
    const size_t n = nw + (dbias ? Cout : 0);
    const long long ms = gr.stride;
    if (splits >= 128)
        hipLaunchKernelGGL(splitk_reduce_kernel<64>, dim3(cg_div_up(n * 64, 256), gr.n), dim3(256), 0, st, part, dw, dbias, nw,
                           Cout, splits, accumulate, ms, ms);
    else if (splits >= 24)
        hipLaunchKernelGGL(splitk_reduce_kernel<8>, dim3(cg_div_up(n * 8, 256), gr.n), dim3(256), 0, st, part, dw, dbias, nw,
                           Cout, splits, accumulate, ms, ms);
    else
        hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3(cg_div_up(n, 256), gr.n), dim3(256), 0, st, part, dw, dbias, nw, Cout,
                           splits, accumulate, ms, ms);
    CG_LAUNCH_CHECK("splitk_reduce_kernel");
    return CG_OK;
}

}  // namespace

// ---- forward ------------------------------------------------------------------------------------------------------
// One implementation behind every fp32 forward entry point: optional instance-norm partials (`stats`), optional
// per-block output maxima (through the thread-local fwd_amax), optional member grouping.
static int conv2d_fwd_impl(const cg_conv_geom* g, const cg_group* group, const float* x1, const float* x2, const float* w,
                           const float* bias, float* y, int cfg, double* stats, size_t stats_bytes, int* rows_per_partial,
                           cg_stream_t stream, const char* who) {
    if (rows_per_partial) *rows_per_partial = 0;
    int rc = validate_geom(g, who);
    if (rc) return rc;
    CG_CHECK_ARG(x1 && w && y, "%s: null pointer", who);
    CG_CHECK_ARG(g->C2 == 0 || x2, "%s: C2 > 0 needs x2", who);
    Grp gr;
    rc = grp_from(group, g->N, gr, who);
    if (rc) return rc;
    const int Ct = g->C1 + g->C2;
    const int K = g->T * Ct;
    const int Mm = (g->N / gr.n) * g->Ho * g->Wo;          // rows of one member
    const bool fast = (g->C2 == 0) && (Ct % 32 == 0);
    if (cfg < 0) cfg = pick_fwd_cfg(g, (long)Mm * gr.n, fast && pipe_ok(g, K));
    double* st_ptr = nullptr;
    if (rows_per_partial && pipe_cfg_has_stats(cfg) && stats && g->act == CG_ACT_NONE && g->osy == 1 && g->osx == 1) {
        const int bm = pipe_cfg_bm(cfg);
        if ((g->Ho * g->Wo) % bm == 0 && stats_bytes >= (size_t)((size_t)Mm * gr.n / bm) * g->Cout * 2 * sizeof(double)) {
            st_ptr = stats;
            *rows_per_partial = bm;
        }
    }
    return launch_fwd_cfg(cfg, g, x1, x2, w, bias, y, Mm, K, fast, cg_s(stream), st_ptr, gr);
}

extern "C" int cg_conv2d_fwd(const cg_conv_geom* g, const float* x1, const float* x2, const float* w,
                             const float* bias, float* y, cg_stream_t stream) {
    return conv2d_fwd_impl(g, nullptr, x1, x2, w, bias, y, -1, nullptr, 0, nullptr, stream, "cg_conv2d_fwd");
}

extern "C" int cg_conv2d_fwd_amax(const cg_conv_geom* g, const float* x1, const float* x2, const float* w,
                                  const float* bias, float* y, float* amax_state, int* amax_nslots, cg_stream_t stream) {
    CG_CHECK_ARG(amax_state && amax_nslots, "cg_conv2d_fwd_amax: null pointer");
    fwd_amax.state = amax_state;
    fwd_amax.nslots = 0;
    const int rc = conv2d_fwd_impl(g, nullptr, x1, x2, w, bias, y, -1, nullptr, 0, nullptr, stream, "cg_conv2d_fwd_amax");
    *amax_nslots = rc ? 0 : fwd_amax.nslots;
    fwd_amax.state = nullptr;
    return rc;
}

extern "C" int cg_conv2d_fwd_stats(const cg_conv_geom* g, const float* x1, const float* x2, const float* w,
                                   const float* bias, float* y, double* stats, size_t stats_bytes, int* rows_per_partial,
                                   cg_stream_t stream) {
    CG_CHECK_ARG(rows_per_partial, "cg_conv2d_fwd_stats: null pointer");
    return conv2d_fwd_impl(g, nullptr, x1, x2, w, bias, y, -1, stats, stats_bytes, rows_per_partial, stream,
                           "cg_conv2d_fwd_stats");
}

// grouped / general form: stats and amax are optional services (NULL = off)
extern "C" int cg_conv2d_fwd_g(const cg_conv_geom* g, const cg_group* group, const float* x1, const float* x2, const float* w,
                               const float* bias, float* y, double* stats, size_t stats_bytes, int* rows_per_partial,
                               float* amax_state, int* amax_nslots, cg_stream_t stream) {
    CG_CHECK_ARG((amax_state == nullptr) == (amax_nslots == nullptr), "cg_conv2d_fwd_g: amax_state and amax_nslots go together");
    fwd_amax.state = amax_state;
    fwd_amax.nslots = 0;
    const int rc = conv2d_fwd_impl(g, group, x1, x2, w, bias, y, -1, stats, stats_bytes, rows_per_partial, stream,
                                   "cg_conv2d_fwd_g");
    if (amax_nslots) *amax_nslots = rc ? 0 : fwd_amax.nslots;
    fwd_amax.state = nullptr;
    return rc;
}

// thin-input layers (3 / 6 / 12 -> 64 channels) of the split-precision datapath: fp32 tensors in and out like cg_conv2d_fwd_g, the
// products on the fp16 x 3 MFMA (conv_fwd_thin_x3_kernel)
// (not the 12 -> 64 1x1 layer: one k-step of work per tile, the fp32 kernel is faster there -- 77 vs 83 us, gpurun_out/s14)
extern "C" int cg_conv2d_fwd_thin_x3_ok(const cg_conv_geom* g) {
    if (!g || !CG_X3_INTERLEAVE) return 0;
    const int v = thin_match(g);
    return v >= 0 && v != 4;
}
extern "C" int cg_conv2d_fwd_thin_x3_g(const cg_conv_geom* g, const cg_group* group, const float* x1, const float* x2, const float* w,
                                       const float* bias, float* y, float* amax_state, int* amax_nslots, cg_stream_t stream) {
    CG_CHECK_ARG((amax_state == nullptr) == (amax_nslots == nullptr), "cg_conv2d_fwd_thin_x3_g: amax_state and amax_nslots go together");
    CG_CHECK_ARG(cg_conv2d_fwd_thin_x3_ok(g), "cg_conv2d_fwd_thin_x3_g: not a thin-input layer (see cg_conv2d_fwd_thin_x3_ok)");
    fwd_amax.state = amax_state;
    fwd_amax.nslots = 0;
    const int rc = conv2d_fwd_impl(g, group, x1, x2, w, bias, y, 41, nullptr, 0, nullptr, stream, "cg_conv2d_fwd_thin_x3_g");
    if (amax_nslots) *amax_nslots = rc ? 0 : fwd_amax.nslots;
    fwd_amax.state = nullptr;
    return rc;
}

extern "C" int cg_conv2d_fwd_tile(const cg_conv_geom* g, const float* x1, const float* x2, const float* w,
                                  const float* bias, float* y, int tile_cfg, cg_stream_t stream) {
    return conv2d_fwd_impl(g, nullptr, x1, x2, w, bias, y, tile_cfg, nullptr, 0, nullptr, stream, "cg_conv2d_fwd_tile");
}

// ---- split-precision forward (conv_x3.inc) --------------------------------------------------------
extern "C" int cg_split_f16(const float* x, void* out, size_t n, size_t lo_elems, float scale, cg_stream_t stream) {
    CG_CHECK_ARG(x && out && n > 0 && x3_lo_ok(lo_elems, n * 2) && scale > 0.f, "cg_split_f16: bad args");
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(split_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, cg_s(stream), x, (_Float16*)out, n, lo_elems,
                       scale);
    CG_LAUNCH_CHECK("split_f16_kernel");
    return CG_OK;
}

static int x3_epilogue_check(const cg_x3_epilogue* epi, const void* y_split, const float* amax_state, const char* who) {
    if (!epi) return CG_OK;
    CG_CHECK_ARG(CG_X3_INTERLEAVE, "%s: the bounded-split epilogue needs the interleaved operand layout", who);
    CG_CHECK_ARG(!epi->l1_ctl || (y_split && epi->out_state && epi->in_state && epi->in_nslots >= 0 && epi->in_nslots <= CG_AMAX_MAX_SLOTS),
                 "%s: bounded split needs the output planes, the input's state and an output state", who);
    CG_CHECK_ARG(!epi->l1_ctl || amax_state == nullptr || amax_state == epi->out_state,
                 "%s: the block maxima of a bounded-split output go to its own state (amax_state == out_state)", who);
    CG_CHECK_ARG(!epi->act_src || epi->act_type == CG_ACT_RELU || epi->act_type == CG_ACT_LRELU,
                 "%s: the fused activation backward takes relu / lrelu (sign-only derivatives)", who);
    return CG_OK;
}
static X3Epi x3_epilogue_of(const cg_x3_epilogue* epi) {
    X3Epi e;
    if (epi) {
        e.ctl = epi->l1_ctl;
        e.in_state = epi->in_state;
        e.in_nslots = epi->in_nslots;
        e.act_type = epi->act_src ? epi->act_type : 0;
        e.act_src = (const _Float16*)epi->act_src;
        e.out_state = epi->l1_ctl ? epi->out_state : nullptr;
        e.addend = epi->addend;
    }
    return e;
}

static int conv2d_fwd_x3_impl(const cg_conv_geom* g, const cg_group* group, const void* xs, size_t x_lo_elems, const void* ws,
                              size_t w_lo_elems, float w_scale, const float* w_scale_dev, const float* x_scale_dev,
                              const float* bias, float* y, void* y_split, size_t y_lo_elems, double* stats,
                              size_t stats_bytes, int* rows_per_partial, int tile_cfg, float* amax_state, int* amax_nslots,
                              cg_stream_t stream, const char* who, const cg_x3_epilogue* epi = nullptr) {
    int rc = validate_geom(g, who);
    CG_CHECK_ARG((amax_state == nullptr) == (amax_nslots == nullptr), "%s: amax_state and amax_nslots go together", who);
    if (amax_nslots) *amax_nslots = 0;
    if (rows_per_partial) *rows_per_partial = 0;
    if (rc) return rc;
    CG_CHECK_ARG(xs && ws && (y || (y_split && epi)) && w_scale > 0.f, "%s: null pointer / bad scale", who);
    rc = x3_epilogue_check(epi, y_split, amax_state, who);
    if (rc) return rc;
    Grp gr;
    rc = grp_from(group, g->N, gr, who);
    if (rc) return rc;
    CG_CHECK_ARG(!y_split || x3_lo_ok(y_lo_elems, (size_t)g->N * g->HoF * g->WoF * g->Cout * 2), "%s: bad y lo offset", who);
    const int K = g->T * g->C1;
    const int Mm = (g->N / gr.n) * g->Ho * g->Wo;
    const size_t x_plane = (size_t)g->N * g->H * g->W * g->C1 * 2, w_plane = (size_t)g->Cout * K * 2;
    CG_CHECK_ARG(x3_lo_ok(x_lo_elems, x_plane) && x3_lo_ok(w_lo_elems, w_plane),
                 "%s: lo offset does not match the operand layout (CG_X3_LO_ELEMS)", who);
    CG_CHECK_ARG(gr.n == 1 || (CG_X3_INTERLEAVE && gr.stride % 32 == 0),
                 "%s: grouped launches need the interleaved operand layout and a member stride that is a multiple of 32", who);
    const size_t x_span = x3_span(x_lo_elems, x_plane), w_span = x3_span(w_lo_elems, w_plane);
    CG_CHECK_ARG(g->C2 == 0 && g->C1 % BK == 0 && x_span < (size_t)CG_OOB && w_span < (size_t)CG_OOB,
                 "%s: needs one source with C %% 32 == 0 and operands spanning < 2 GiB", who);
    PipeBatch b;
    fill_class(b.c[0], g, (const float*)ws, gr.n);
    b.c[0].w_bytes = (unsigned)(w_lo_elems * 2);
    b.c[0].pad_ = (int32_t)w_span;
    const int cfg = tile_cfg < 0 ? pick_x3_cfg(g->Cout, (long)Mm * gr.n, g->C1, g->up ? 1 : g->T) : tile_cfg;
    CG_CHECK_ARG((cfg != 6 && cfg != 7) || g->C1 % 64 == 0, "%s: tile configuration %d needs C %% 64 == 0", who, cfg);
    const int bm = x3_cfg_bm(cfg);
    double* st_ptr = nullptr;
    if (rows_per_partial && stats && g->act == CG_ACT_NONE && g->osy == 1 && g->osx == 1 && (g->Ho * g->Wo) % bm == 0 &&
        stats_bytes >= (size_t)((size_t)Mm * gr.n / bm) * g->Cout * 2 * sizeof(double)) {
        st_ptr = stats;
        *rows_per_partial = bm;
    }
    hipStream_t st = cg_s(stream);
    X3Extra ex;
    ex.w_scale_dev = w_scale_dev;
    ex.mb = Members{gr.n, 0, gr.stride * 4, gr.stride * 4};       // interleaved {hi, lo}: 4 bytes per element
    ex.epi = x3_epilogue_of(epi);
    fwd_amax.state = amax_state;
    fwd_amax.nslots = 0;
    rc = launch_x3_cfg(cfg, b, 1, xs, bias, y, (unsigned)(x_lo_elems * 2), (unsigned)x_span, 1.0f / w_scale, x_scale_dev, st,
                       st_ptr, y_split, y_lo_elems, ex);
    if (amax_nslots && !rc) *amax_nslots = fwd_amax.nslots;
    fwd_amax.state = nullptr;
    return rc;
}

extern "C" int cg_conv2d_fwd_x3(const cg_conv_geom* g, const void* xs, size_t x_lo_elems, const void* ws,
                                size_t w_lo_elems, float w_scale, const float* x_scale_dev, const float* bias, float* y,
                                void* y_split, size_t y_lo_elems, double* stats, size_t stats_bytes,
                                int* rows_per_partial, int tile_cfg, float* amax_state, int* amax_nslots,
                                cg_stream_t stream) {
    return conv2d_fwd_x3_impl(g, nullptr, xs, x_lo_elems, ws, w_lo_elems, w_scale, nullptr, x_scale_dev, bias, y, y_split,
                              y_lo_elems, stats, stats_bytes, rows_per_partial, tile_cfg, amax_state, amax_nslots, stream,
                              "cg_conv2d_fwd_x3");
}

// The same layer of `group->n` council members as ONE launch (blockIdx.z = member): member z's samples are block z of the
// batched activation tensor, its weights / bias sit z * group->stride elements after member 0's.  w_scale_dev: the
// device-side power-of-two scale the weights were split with (cg_split_f16_dynamic; NULL = the static w_scale only).
extern "C" int cg_conv2d_fwd_x3_g(const cg_conv_geom* g, const cg_group* group, const void* xs, size_t x_lo_elems,
                                  const void* ws, size_t w_lo_elems, float w_scale, const float* w_scale_dev,
                                  const float* x_scale_dev, const float* bias, float* y, void* y_split, size_t y_lo_elems,
                                  double* stats, size_t stats_bytes, int* rows_per_partial, int tile_cfg, float* amax_state,
                                  int* amax_nslots, cg_stream_t stream) {
    return conv2d_fwd_x3_impl(g, group, xs, x_lo_elems, ws, w_lo_elems, w_scale, w_scale_dev, x_scale_dev, bias, y, y_split,
                              y_lo_elems, stats, stats_bytes, rows_per_partial, tile_cfg, amax_state, amax_nslots, stream,
                              "cg_conv2d_fwd_x3_g");
}

// cg_conv2d_fwd_x3_g with the bounded-split epilogue (include/council_gan_hip.h, cg_x3_epilogue): y may be NULL (planes only)
extern "C" int cg_conv2d_fwd_x3_e(const cg_conv_geom* g, const cg_group* group, const void* xs, size_t x_lo_elems,
                                  const void* ws, size_t w_lo_elems, float w_scale, const float* w_scale_dev,
                                  const float* x_scale_dev, const float* bias, float* y, void* y_split, size_t y_lo_elems,
                                  const cg_x3_epilogue* epi, int tile_cfg, int* amax_nslots, cg_stream_t stream) {
    CG_CHECK_ARG(epi && amax_nslots, "cg_conv2d_fwd_x3_e: the epilogue descriptor and amax_nslots are required");
    return conv2d_fwd_x3_impl(g, group, xs, x_lo_elems, ws, w_lo_elems, w_scale, w_scale_dev, x_scale_dev, bias, y, y_split,
                              y_lo_elems, nullptr, 0, nullptr, tile_cfg, epi->out_state, amax_nslots, stream,
                              "cg_conv2d_fwd_x3_e", epi);
}

// {largest row (by_input_channel = 0: over output channels of sum_{t,ci} |w|) or column (1: over input channels of
// sum_{co,t} |w|) 1-norm of a convolution weight [Cout][T][Cin], largest |bias|}, the maximum over the members of the group
namespace {
__global__ __launch_bounds__(256) void weight_l1_rows_kernel(const float* __restrict__ w, const float* __restrict__ bias, int K,
                                                             long long w_mstride, long long b_mstride, unsigned* __restrict__ out_bits) {
    // one block per (output row, member): sum_k |w[row][k]|; positive floats order like their bit patterns -> integer atomic max
    const int member = blockIdx.y, row = blockIdx.x;
    const float* r = w + (long long)member * w_mstride + (size_t)row * K;
    __shared__ float red[4];
    float a = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) a += fabsf(r[k]);
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(out_bits, __float_as_uint(red[0] + red[1] + red[2] + red[3]));
        if (bias) atomicMax(out_bits + 1, __float_as_uint(fabsf(bias[(long long)member * b_mstride + row])));
    }
}
constexpr int L1_CHUNKS = 32;
// column sums in two deterministic stages: (32 channels x 8 row lanes) per block over one chunk of the Cout * T rows -> partial
// [member][chunk][ci]; then per channel the chunks are added in order and the maximum taken
__global__ __launch_bounds__(256) void weight_l1_cols_partial(const float* __restrict__ w, int R, int Cin, long long w_mstride,
                                                              float* __restrict__ part) {
    const int member = blockIdx.z, chunk = blockIdx.y;
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), rl = threadIdx.x >> 5;
    const int per = (R + L1_CHUNKS - 1) / L1_CHUNKS, r0 = chunk * per, r1 = min(R, r0 + per);
    const float* wm = w + (long long)member * w_mstride;
    float a = 0.f;
    if (c < Cin)
        for (int r = r0 + rl; r < r1; r += 8) a += fabsf(wm[(size_t)r * Cin + c]);
    __shared__ float red[8][33];
    red[rl][threadIdx.x & 31] = a;
    __syncthreads();
    if (threadIdx.x < 32 && c < Cin) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x];
        part[((size_t)member * L1_CHUNKS + chunk) * Cin + c] = t;
    }
}
__global__ __launch_bounds__(256) void weight_l1_cols_final(const float* __restrict__ part, int Cin, int nmember, float* __restrict__ out2) {
    __shared__ float red[4];
    float m = 0.f;
    for (int i = threadIdx.x; i < Cin * nmember; i += 256) {
        const int member = i / Cin, c = i - member * Cin;
        float t = 0.f;
        for (int k = 0; k < L1_CHUNKS; ++k) t += part[((size_t)member * L1_CHUNKS + k) * Cin + c];
        m = fmaxf(m, t);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        out2[0] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        out2[1] = 0.f;
    }
}
}  // namespace
extern "C" size_t cg_weight_l1_workspace(int Cin, int nmember) { return (size_t)(nmember < 1 ? 1 : nmember) * L1_CHUNKS * Cin * sizeof(float); }
extern "C" int cg_weight_l1_bound(const cg_group* group, const float* w, int Cout, int T, int Cin, const float* bias,
                                  int by_input_channel, float* out2, void* ws, size_t ws_bytes, cg_stream_t stream) {
    CG_CHECK_ARG(w && out2 && Cout > 0 && T > 0 && Cin > 0, "cg_weight_l1_bound: bad args");
    const int n = group ? group->n : 1;
    const long long ms = group ? group->stride : 0;
    hipStream_t st = cg_s(stream);
    if (!by_input_channel) {
        hipLaunchKernelGGL(zero2_kernel, dim3(1), dim3(64), 0, st, out2);
        hipLaunchKernelGGL(weight_l1_rows_kernel, dim3(Cout, n), dim3(256), 0, st, w, bias, T * Cin, ms, ms,
                           reinterpret_cast<unsigned*>(out2));
    } else {
        if (!ws || ws_bytes < cg_weight_l1_workspace(Cin, n)) return cg_set_error(CG_ERR_WORKSPACE, "cg_weight_l1_bound: workspace too small");
        hipLaunchKernelGGL(weight_l1_cols_partial, dim3((Cin + 31) / 32, L1_CHUNKS, n), dim3(256), 0, st, w, Cout * T, Cin, ms, (float*)ws);
        hipLaunchKernelGGL(weight_l1_cols_final, dim3(1), dim3(256), 0, st, (const float*)ws, Cin, n, out2);
    }
    CG_LAUNCH_CHECK("weight_l1_kernel");
    return CG_OK;
}

extern "C" int cg_split_f16_dynamic(const float* x, void* out, size_t n, size_t lo_elems, float* state, int nslots,
                                    cg_stream_t stream) {
    return cg_split_f16_dynamic_capped(x, out, n, lo_elems, state, nslots, 0.f, stream);
}

// max_scale > 0: never scale UP by more than this power of two (weights: 2^10 keeps the small-weight regime of the
// static CG_X3_WSCALE; larger magnitudes get the smaller scale that keeps their hi halves finite)
extern "C" int cg_split_f16_dynamic_capped(const float* x, void* out, size_t n, size_t lo_elems, float* state, int nslots,
                                           float max_scale, cg_stream_t stream) {
    CG_CHECK_ARG(x && out && state && n > 0 && x3_lo_ok(lo_elems, n * 2) && nslots >= 0 && nslots <= CG_AMAX_MAX_SLOTS &&
                     max_scale >= 0.f,
                 "cg_split_f16_dynamic: bad args");
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipStream_t st = cg_s(stream);
    if (nslots == 0) {      // nobody measured the tensor yet: one reduction pass; otherwise the producer filled the slots
        nslots = amax_blocks(n);
        hipLaunchKernelGGL(amax_kernel, dim3(nslots), dim3(256), 0, st, x, n, state);
        CG_LAUNCH_CHECK("amax_kernel");
    }
    hipLaunchKernelGGL(split_f16_dyn_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, (_Float16*)out, n, lo_elems, state,
                       nslots, max_scale);
    CG_LAUNCH_CHECK("split_f16_dyn_kernel");
    return CG_OK;
}

// ---- weight gradient -----------------------------------------------------------------------------------------------
static size_t wgrad_workspace(const cg_conv_geom* g, int nmember) {
    if (!g || g->T < 1 || g->T > CG_MAX_TAPS || nmember < 1 || g->N % nmember) return 0;
    WgradPlan p = plan_wgrad(g, nmember);
    const WgradPlan px = plan_wgrad(g, nmember, true);     // the split-precision plan may split further (larger tile)
    int splits = p.splits > px.splits ? p.splits : px.splits;
    if (wgrad_thin_on() && thin_match(g) >= 0) {
        const int ts = thin_wgrad_splits(g, nmember);
        if (ts > splits) splits = ts;
    }
    const size_t K = (size_t)g->T * (g->C1 + g->C2);
    return (size_t)nmember * splits * ((size_t)g->Cout * K + g->Cout) * sizeof(float);
}
extern "C" size_t cg_conv2d_wgrad_workspace(const cg_conv_geom* g) { return wgrad_workspace(g, 1); }
extern "C" size_t cg_conv2d_wgrad_workspace_g(const cg_conv_geom* g, const cg_group* group) {
    return wgrad_workspace(g, group ? group->n : 1);
}

static int conv2d_wgrad_impl(const cg_conv_geom* g, const cg_group* group, const float* x1, const float* x2, const float* dz,
                             float* dw, float* dbias, int accumulate, void* ws, size_t ws_bytes, cg_stream_t stream,
                             const char* who, const float* yact = nullptr, int act = 0) {
    int rc = validate_geom(g, who);
    if (rc) return rc;
    CG_CHECK_ARG(x1 && dz && dw, "%s: null pointer", who);
    CG_CHECK_ARG(g->C2 == 0 || x2, "%s: C2 > 0 needs x2", who);
    Grp gr;
    rc = grp_from(group, g->N, gr, who);
    if (rc) return rc;
    const size_t need = wgrad_workspace(g, gr.n);
    if (!ws || ws_bytes < need) return cg_set_error(CG_ERR_WORKSPACE, "%s: workspace %zu < %zu", who, ws_bytes, need);
    const int Ct = g->C1 + g->C2;
    const int K = g->T * Ct;
    const int M = (g->N / gr.n) * g->Ho * g->Wo;
    hipStream_t st = cg_s(stream);
    WgradPlan p = plan_wgrad(g, gr.n);
    float* part = (float*)ws;
    const int want_bias = dbias != nullptr;
    if (yact && !(wgrad_thin_on() && thin_match(g) >= 0))
        return cg_set_error(CG_ERR_ARG, "%s: only the thin-input weight gradient folds an activation backward in (cg_conv2d_wgrad_act_ok)", who);
    if (wgrad_thin_on() && thin_match(g) >= 0) {
        rc = launch_wgrad_thin_variant(thin_match(g), g, x1, x2, dz, part, want_bias, st, gr.n, yact, act);
        if (rc) return rc;
        return launch_splitk_reduce(part, dw, dbias, (size_t)g->Cout * K, g->Cout, thin_wgrad_splits(g, gr.n), accumulate, gr, st);
    }
#define WG(BM_, BN_, WM_, WN_) rc = launch_wgrad<BM_, BN_, WM_, WN_>(g, p, x1, x2, dz, part, M, K, want_bias, st, gr.n)
#define WGP(BM_, BN_, WM_, WN_) rc = launch_wgrad_pipe<BM_, BN_, WM_, WN_>(g, p, x1, dz, part, M, K, want_bias, st, gr.n)
    if (wgrad_pipe_ok(g, p) && !wgrad_force_legacy()) {
        if (p.bm == 128 && p.bn == 128) WGP(128, 128, 64, 32);   // 8 waves
        else if (p.bm == 128 && p.bn == 64) WGP(128, 64, 64, 32);
        else if (p.bm == 64 && p.bn == 64) WGP(64, 64, 32, 32);
        else WGP(64, 128, 32, 64);
    }
    else if (p.bm == 128 && p.bn == 128) WG(128, 128, 64, 32);   // 8 waves
    else if (p.bm == 128 && p.bn == 64) WG(128, 64, 64, 32);
    else if (p.bm == 128 && p.bn == 32) WG(128, 32, 32, 32);
    else if (p.bm == 64 && p.bn == 128) WG(64, 128, 32, 64);
    else if (p.bm == 64 && p.bn == 64) WG(64, 64, 32, 32);
    else if (p.bm == 32 && p.bn == 128) WG(32, 128, 32, 32);
    else return cg_set_error(CG_ERR_ARG, "%s: no tile for %dx%d", who, p.bm, p.bn);
#undef WG
#undef WGP
    if (rc) return rc;
    return launch_splitk_reduce(part, dw, dbias, (size_t)g->Cout * K, g->Cout, p.splits, accumulate, gr, st);
}

extern "C" int cg_conv2d_wgrad(const cg_conv_geom* g, const float* x1, const float* x2, const float* dz, float* dw,
                               float* dbias, int accumulate, void* ws, size_t ws_bytes, cg_stream_t stream) {
    return conv2d_wgrad_impl(g, nullptr, x1, x2, dz, dw, dbias, accumulate, ws, ws_bytes, stream, "cg_conv2d_wgrad");
}
extern "C" int cg_conv2d_wgrad_g(const cg_conv_geom* g, const cg_group* group, const float* x1, const float* x2,
                                 const float* dz, float* dw, float* dbias, int accumulate, void* ws, size_t ws_bytes,
                                 cg_stream_t stream) {
    return conv2d_wgrad_impl(g, group, x1, x2, dz, dw, dbias, accumulate, ws, ws_bytes, stream, "cg_conv2d_wgrad_g");
}

extern "C" int cg_conv2d_wgrad_act_ok(const cg_conv_geom* g) { return g && wgrad_thin_on() && thin_match(g) >= 0; }
extern "C" int cg_conv2d_wgrad_act_g(const cg_conv_geom* g, const cg_group* group, const float* x1, const float* x2,
                                     const float* dy, const float* y, int act, float* dw, float* dbias, int accumulate, void* ws,
                                     size_t ws_bytes, cg_stream_t stream) {
    CG_CHECK_ARG(y && (act == CG_ACT_RELU || act == CG_ACT_LRELU || act == CG_ACT_TANH),
                 "cg_conv2d_wgrad_act_g: needs the activated output and its activation (relu / lrelu / tanh)");
    return conv2d_wgrad_impl(g, group, x1, x2, dy, dw, dbias, accumulate, ws, ws_bytes, stream, "cg_conv2d_wgrad_act_g", y, act);
}

// split-precision weight gradient (conv_x3.inc): x and dz arrive as {hi, lo} fp16 planes with their power-of-two
// scales (device-side pointers, NULL = 1); same planning, partial layout and deterministic reduce as cg_conv2d_wgrad
namespace {
template <int BM, int BN, int WM, int WN>
int launch_wgrad_x3(const cg_conv_geom* g, const WgradPlan& p, const void* xs, size_t x_lo, const float* x_scale,
                    const void* dzs, size_t dz_lo, const float* dz_scale, float* out, int M, int K, int want_bias,
                    hipStream_t st, int nmember) {
    dim3 grid(p.tiles_m * p.tiles_n, nmember, p.splits), block((BM / WM) * (BN / WN) * 64);
    const size_t x_plane = (size_t)g->N * g->H * g->W * g->C1 * 2, dz_plane = (size_t)g->N * g->Ho * g->Wo * g->Cout * 2;
    ProfScope prof(5, BM, BN, true, 2.0 * (double)M * nmember * (double)g->Cout * (double)K, st, g, nmember);
    hipLaunchKernelGGL((conv_wgrad_x3_kernel<BM, BN, WM, WN>), grid, block, 0, st, *g, xs, (unsigned)(x_lo * 2),
                       (unsigned)x3_span(x_lo, x_plane), x_scale, dzs, (unsigned)(dz_lo * 2), (unsigned)x3_span(dz_lo, dz_plane),
                       dz_scale, out, M, K, p.tiles_n, p.slices_per_split, want_bias, ilog2_exact(g->Ho * g->Wo),
                       ilog2_exact(g->Wo));
    CG_LAUNCH_CHECK("conv_wgrad_x3_kernel");
    return CG_OK;
}
}  // namespace

#if CG_X3_INTERLEAVE
namespace {
template <int BM, int BN, int WM, int WN>
int launch_wgrad_x3t(const cg_conv_geom* g, const WgradPlan& p, const void* xs, size_t x_lo, const float* x_scale,
                     const void* dzs, size_t dz_lo, const float* dz_scale, float* out, int M, int K, int want_bias,
                     hipStream_t st, int nmember) {
    dim3 grid(p.tiles_m * p.tiles_n, nmember, p.splits), block((BM / WM) * (BN / WN) * 64);
    const size_t x_plane = (size_t)g->N * g->H * g->W * g->C1 * 2, dz_plane = (size_t)g->N * g->Ho * g->Wo * g->Cout * 2;
    ProfScope prof(7, BM, BN, true, 2.0 * (double)M * nmember * (double)g->Cout * (double)K, st, g, nmember);
    hipLaunchKernelGGL((conv_wgrad_x3t_kernel<BM, BN, WM, WN>), grid, block, 0, st, *g, xs, (unsigned)x3_span(x_lo, x_plane),
                       x_scale, dzs, (unsigned)x3_span(dz_lo, dz_plane), dz_scale, out, M, K, p.tiles_n, p.slices_per_split,
                       want_bias, ilog2_exact(g->Ho * g->Wo), ilog2_exact(g->Wo), tune().wgrad_xcd_group);
    CG_LAUNCH_CHECK("conv_wgrad_x3t_kernel");
    return CG_OK;
}
int launch_wgrad_x3tw(const cg_conv_geom* g, const WgradPlan& p, const void* xs, size_t x_lo, const float* x_scale,
                      const void* dzs, size_t dz_lo, const float* dz_scale, float* out, int M, int K, int want_bias, hipStream_t st,
                      int nmember) {
    dim3 grid(p.tiles_m * p.tiles_n, nmember, p.splits), block(512);
    const size_t x_plane = (size_t)g->N * g->H * g->W * g->C1 * 2, dz_plane = (size_t)g->N * g->Ho * g->Wo * g->Cout * 2;
    ProfScope prof(7, 256, 256, true, 2.0 * (double)M * nmember * (double)g->Cout * (double)K, st, g, nmember);
    hipLaunchKernelGGL((conv_wgrad_x3tw_kernel<256, 256, 128, 64, 1>), grid, block, 0, st, *g, xs, (unsigned)x3_span(x_lo, x_plane),
                       x_scale, dzs, (unsigned)x3_span(dz_lo, dz_plane), dz_scale, out, M, K, p.tiles_n, p.slices_per_split,
                       want_bias, ilog2_exact(g->Ho * g->Wo), ilog2_exact(g->Wo), tune().wgrad_xcd_group);
    CG_LAUNCH_CHECK("conv_wgrad_x3tw_kernel");
    return CG_OK;
}
}  // namespace
#endif
// CG_WGRAD_X3_PERM=1 in the environment keeps the v_perm / ds_write_b32 loader (A/B against the transposing LDS read)
static bool wgrad_x3_use_tr() { return CG_X3_INTERLEAVE && !tune().wgrad_x3_perm; }

static int wgrad_x3_ok(const cg_conv_geom* g, int nmember) {
    if (!g || g->T < 1 || g->T > CG_MAX_TAPS || nmember < 1 || g->N % nmember) return 0;
    WgradPlan p = plan_wgrad(g, nmember, true);
    return wgrad_pipe_ok(g, p) && (g->Cout & 31) == 0 && (g->C1 & 31) == 0;
}
extern "C" int cg_conv2d_wgrad_x3_ok(const cg_conv_geom* g) { return wgrad_x3_ok(g, 1); }
extern "C" int cg_conv2d_wgrad_x3_ok_g(const cg_conv_geom* g, const cg_group* group) { return wgrad_x3_ok(g, group ? group->n : 1); }

static int conv2d_wgrad_x3_impl(const cg_conv_geom* g, const cg_group* group, const void* xs, size_t x_lo_elems,
                                const float* x_scale_dev, const void* dzs, size_t dz_lo_elems, const float* dz_scale_dev,
                                float* dw, float* dbias, int accumulate, void* ws, size_t ws_bytes, cg_stream_t stream,
                                const char* who) {
    int rc = validate_geom(g, who);
    if (rc) return rc;
    CG_CHECK_ARG(xs && dzs && dw, "%s: null pointer", who);
    Grp gr;
    rc = grp_from(group, g->N, gr, who);
    if (rc) return rc;
    CG_CHECK_ARG(wgrad_x3_ok(g, gr.n), "%s: layer does not qualify (see cg_conv2d_wgrad_x3_ok)", who);
    const size_t need = wgrad_workspace(g, gr.n);
    if (!ws || ws_bytes < need) return cg_set_error(CG_ERR_WORKSPACE, "%s: workspace %zu < %zu", who, ws_bytes, need);
    const int K = g->T * g->C1;
    const int M = (g->N / gr.n) * g->Ho * g->Wo;
    const size_t x_plane = (size_t)g->N * g->H * g->W * g->C1 * 2, dz_plane = (size_t)g->N * g->Ho * g->Wo * g->Cout * 2;
    CG_CHECK_ARG(x3_lo_ok(x_lo_elems, x_plane) && x3_lo_ok(dz_lo_elems, dz_plane) && x3_span(x_lo_elems, x_plane) < (size_t)CG_OOB &&
                     x3_span(dz_lo_elems, dz_plane) < (size_t)CG_OOB && g->Cout % 32 == 0 && g->C1 % 32 == 0,
                 "%s: operand planes out of range / lo offset does not match the layout", who);
    hipStream_t st = cg_s(stream);
    WgradPlan p = plan_wgrad(g, gr.n, true);
    float* part = (float*)ws;
    const int want_bias = dbias != nullptr;
#if CG_X3_INTERLEAVE
#define WGX(BM_, BN_, WM_, WN_)                                                                                                  \
    rc = wgrad_x3_use_tr()                                                                                                       \
             ? launch_wgrad_x3t<BM_, BN_, WM_, WN_>(g, p, xs, x_lo_elems, x_scale_dev, dzs, dz_lo_elems, dz_scale_dev, part, M, K, want_bias, st, gr.n) \
             : launch_wgrad_x3<BM_, BN_, WM_, WN_>(g, p, xs, x_lo_elems, x_scale_dev, dzs, dz_lo_elems, dz_scale_dev, part, M, K, want_bias, st, gr.n)
#else
#define WGX(BM_, BN_, WM_, WN_) \
    rc = launch_wgrad_x3<BM_, BN_, WM_, WN_>(g, p, xs, x_lo_elems, x_scale_dev, dzs, dz_lo_elems, dz_scale_dev, part, M, K, want_bias, st, gr.n)
#endif
    // (64x64 wave tiles on the 128x128 / 256x128 tiles -- 4 / 8 waves, 0.67 transposing reads per MFMA instead of 1.0 -- measured in round 6:
    // -6 ... -15 % at half the occupancy, gpurun_out/s12; not kept)
    if (p.bm == 128 && p.bn == 128) WGX(128, 128, 64, 32);   // 8 waves
#if CG_X3_INTERLEAVE
    else if (p.bm == 256 && p.bn == 256)                     // experimental wide tile (CG_WGRAD_X3_WIDE)
        rc = launch_wgrad_x3tw(g, p, xs, x_lo_elems, x_scale_dev, dzs, dz_lo_elems, dz_scale_dev, part, M, K, want_bias, st, gr.n);
    else if (p.bm == 256 && p.bn == 128)                     // 16 waves (CG_WGRAD_X3_BM256), transposing-read kernel only
        rc = launch_wgrad_x3t<256, 128, 64, 32>(g, p, xs, x_lo_elems, x_scale_dev, dzs, dz_lo_elems, dz_scale_dev, part, M, K, want_bias, st, gr.n);
#endif
    else if (p.bm == 128 && p.bn == 64) WGX(128, 64, 64, 32);
    else if (p.bm == 64 && p.bn == 64) WGX(64, 64, 32, 32);
    else WGX(64, 128, 32, 64);
#undef WGX
    if (rc) return rc;
    return launch_splitk_reduce(part, dw, dbias, (size_t)g->Cout * K, g->Cout, p.splits, accumulate, gr, st);
}

extern "C" int cg_conv2d_wgrad_x3(const cg_conv_geom* g, const void* xs, size_t x_lo_elems, const float* x_scale_dev,
                                  const void* dzs, size_t dz_lo_elems, const float* dz_scale_dev, float* dw, float* dbias,
                                  int accumulate, void* ws, size_t ws_bytes, cg_stream_t stream) {
    return conv2d_wgrad_x3_impl(g, nullptr, xs, x_lo_elems, x_scale_dev, dzs, dz_lo_elems, dz_scale_dev, dw, dbias, accumulate,
                                ws, ws_bytes, stream, "cg_conv2d_wgrad_x3");
}
extern "C" int cg_conv2d_wgrad_x3_g(const cg_conv_geom* g, const cg_group* group, const void* xs, size_t x_lo_elems,
                                    const float* x_scale_dev, const void* dzs, size_t dz_lo_elems, const float* dz_scale_dev,
                                    float* dw, float* dbias, int accumulate, void* ws, size_t ws_bytes, cg_stream_t stream) {
    return conv2d_wgrad_x3_impl(g, group, xs, x_lo_elems, x_scale_dev, dzs, dz_lo_elems, dz_scale_dev, dw, dbias, accumulate, ws,
                                ws_bytes, stream, "cg_conv2d_wgrad_x3_g");
}

// ---- the kernel-selection table behind the ABI (see cg_tuning above) ------------------------------------------------------
extern "C" int cg_tuning_get(cg_tuning* out) {
    CG_CHECK_ARG(out != nullptr, "cg_tuning_get: null pointer");
    *out = tune();
    return CG_OK;
}
extern "C" int cg_tuning_set(const cg_tuning* in) {
    CG_CHECK_ARG(in != nullptr, "cg_tuning_set: null pointer");
    CG_CHECK_ARG(in->wgrad_x3_bm256 >= 0 && in->wgrad_x3_bm256 <= 2 && in->wgrad_x3_wide >= 0 && in->wgrad_x3_wide <= 2 && in->tile_rows_scale >= 1 && in->tile_rows_scale <= 64 &&
                     (in->x3_thin_out == 0 || in->x3_thin_out == 20 || in->x3_thin_out == 21) && in->x3_korder >= 0 && in->x3_korder <= 2 &&
                     (in->x3_wide == 0 || in->x3_wide == 1 || in->x3_wide == 16 || in->x3_wide == 17),
                 "cg_tuning_set: field out of range");
    tune() = *in;
    return CG_OK;
}
// single-field wrappers (A/B tools and tests); each returns the previous setting
extern "C" int cg_conv2d_wgrad_x3_wide(int mode) {      // 0 / 1 / 2 as CG_WGRAD_X3_WIDE
    const int prev = tune().wgrad_x3_wide;
    tune().wgrad_x3_wide = (mode == 0 || mode == 1) ? mode : 2;
    return prev;
}
extern "C" int cg_conv2d_wgrad_thin(int on) {           // workspace queries follow it
    const int prev = tune().wgrad_thin;
    tune().wgrad_thin = on != 0;
    return prev;
}
extern "C" int cg_conv2d_fwd_thin(int on) {
    const int prev = tune().fwd_thin;
    tune().fwd_thin = on != 0;
    return prev;
}
extern "C" int cg_conv2d_wgrad_x3_bm256(int mode) {    // 0 / 1 / 2 as CG_WGRAD_X3_BM256
    const int prev = tune().wgrad_x3_bm256;
    tune().wgrad_x3_bm256 = (mode == 0 || mode == 1) ? mode : 2;
    return prev;
}
extern "C" int cg_conv2d_wgrad_legacy(int on) {
    tune().wgrad_legacy = on != 0;
    return CG_OK;
}

// Re-layout (and optionally split) of the weights of `nmember` members for the data-gradient passes: member m reads
// w + m*w_mstride and writes out + m*out_mstride (elements)
static int launch_transpose(const float* w, float* out, int Cout, int T, int Cin, int ci0, int nci, const TransTable& tt,
                            int nz, hipStream_t st, bool split, float scale, unsigned lo_elems, int nmember,
                            long long w_mstride, long long out_mstride, const float* scale_dev) {
    dim3 grid(cg_div_up(nci, 32), cg_div_up(Cout, 32), nz * nmember);
    if (split)
        hipLaunchKernelGGL(weight_transpose_kernel<true>, grid, dim3(256), 0, st, w, out, Cout, T, Cin, ci0, nci, tt, scale,
                           lo_elems, nz, w_mstride, out_mstride, scale_dev);
    else
        hipLaunchKernelGGL(weight_transpose_kernel<false>, grid, dim3(256), 0, st, w, out, Cout, T, Cin, ci0, nci, tt, 1.f, 0u,
                           nz, w_mstride, out_mstride, (const float*)nullptr);
    CG_LAUNCH_CHECK("weight_transpose_kernel");
    return CG_OK;
}

extern "C" int cg_weight_transpose(const float* w, float* out, int Cout, int T, int Cin, int ci0, int nci,
                                   const int32_t* tapmap_host, int Tc, cg_stream_t stream) {
    CG_CHECK_ARG(w && out && tapmap_host, "cg_weight_transpose: null pointer");
    CG_CHECK_ARG(Tc >= 1 && Tc <= CG_MAX_TAPS && T >= 1 && ci0 >= 0 && nci >= 1 && ci0 + nci <= Cin,
                 "cg_weight_transpose: bad sizes");
    TransTable tt;
    memset(&tt, 0, sizeof(tt));
    for (int i = 0; i < Tc; ++i) {
        CG_CHECK_ARG(tapmap_host[i] >= 0 && tapmap_host[i] < T, "cg_weight_transpose: tap %d out of range", tapmap_host[i]);
        tt.src_tap[i] = tapmap_host[i];
        tt.dst_base[i] = 0;
        tt.dst_tc[i] = i;
        tt.dst_T[i] = Tc;
    }
    return launch_transpose(w, out, Cout, T, Cin, ci0, nci, tt, Tc, cg_s(stream), false, 1.f, 0u, 1, 0, 0, nullptr);
}

// ---- data gradient of a convolution -------------------------------------------------------------
// dx[ly][lx] = sum over taps t and output positions with  oy*stride + dy[t] == ly  of  dz[oy][ox] * W[t].
// Per output-parity class (ph, pw) = (ly % stride, lx % stride) this is a stride-1 convolution OVER dz with the taps
// t for which stride | (ph - dy[t]), at offsets (ph - dy[t]) / stride, and with transposed weights; the classes
// interleave into dx through the strided epilogue.  All classes of a layer run as ONE launch (blockIdx.y).
namespace {
struct DgradPlan {
    int ncls;
    cg_conv_geom cg[CG_MAX_TAPS];  // at most stride^2 classes; only the first ncls are valid (<= 4 batched)
    int tap_src[CG_MAX_TAPS][CG_MAX_TAPS];
    size_t w_off[CG_MAX_TAPS];
    size_t ws_floats;              // elements of ONE member's re-laid-out weights
};
inline int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

int plan_dgrad(const cg_conv_geom* g, int nci, DgradPlan& p) {
    const int s = g->stride;
    const int Hl = g->H << g->up, Wl = g->W << g->up;
    CG_CHECK_ARG(s >= 1 && s * s <= CG_MAX_TAPS, "cg_conv2d_dgrad: stride %d not supported", s);
    p.ncls = 0;
    p.ws_floats = 0;
    for (int ph = 0; ph < s; ++ph)
        for (int pw = 0; pw < s; ++pw) {
            const int Hc = (Hl - ph + s - 1) / s, Wc = (Wl - pw + s - 1) / s;
            if (Hc <= 0 || Wc <= 0) continue;
            cg_conv_geom& c = p.cg[p.ncls];
            memset(&c, 0, sizeof(c));
            int tc = 0;
            for (int t = 0; t < g->T; ++t) {
                const int ay = ph - g->dy[t], ax = pw - g->dx[t];
                if (ay - floordiv(ay, s) * s != 0 || ax - floordiv(ax, s) * s != 0) continue;
                c.dy[tc] = (int8_t)floordiv(ay, s);
                c.dx[tc] = (int8_t)floordiv(ax, s);
                p.tap_src[p.ncls][tc] = t;
                ++tc;
            }
            CG_CHECK_ARG(tc > 0, "cg_conv2d_dgrad: input positions without a tap (stride > kernel size)");
            c.N = g->N; c.H = g->Ho; c.W = g->Wo; c.C1 = g->Cout; c.C2 = 0; c.up = 0;
            c.Ho = Hc; c.Wo = Wc; c.HoF = Hl; c.WoF = Wl;
            c.osy = c.osx = s; c.ooy = ph; c.oox = pw;
            c.stride = 1; c.T = tc; c.Cout = nci; c.act = CG_ACT_NONE;
            p.w_off[p.ncls] = p.ws_floats;
            p.ws_floats += (size_t)nci * tc * g->Cout;
            ++p.ncls;
        }
    return CG_OK;
}

void dgrad_table(const DgradPlan& p, TransTable& tt, int& nz) {
    memset(&tt, 0, sizeof(tt));
    nz = 0;
    for (int c = 0; c < p.ncls; ++c)
        for (int tc = 0; tc < p.cg[c].T; ++tc, ++nz) {
            tt.src_tap[nz] = p.tap_src[c][tc];
            tt.dst_base[nz] = (int32_t)p.w_off[c];
            tt.dst_tc[nz] = tc;
            tt.dst_T[nz] = p.cg[c].T;
        }
}
}  // namespace

static size_t dgrad_workspace(const cg_conv_geom* g, int nci, int nmember) {
    if (!g || g->T < 1 || g->T > CG_MAX_TAPS || nci < 1 || nmember < 1) return 0;
    return (size_t)nmember * g->T * nci * g->Cout * sizeof(float);   // every tap belongs to exactly one class
}
extern "C" size_t cg_conv2d_dgrad_workspace(const cg_conv_geom* g, int nci) { return dgrad_workspace(g, nci, 1); }
extern "C" size_t cg_conv2d_dgrad_workspace_g(const cg_conv_geom* g, const cg_group* group, int nci) {
    return dgrad_workspace(g, nci, group ? group->n : 1);
}

static int conv2d_dgrad_impl(const cg_conv_geom* g, const cg_group* group, const float* dz, const float* w, int ci0, int nci,
                             float* dx, void* ws, size_t ws_bytes, cg_stream_t stream, const char* who) {
    int rc = validate_geom(g, who);
    if (rc) return rc;
    CG_CHECK_ARG(dz && w && dx, "%s: null pointer", who);
    Grp gr;
    rc = grp_from(group, g->N, gr, who);
    if (rc) return rc;
    const int Cin = g->C1 + g->C2;
    CG_CHECK_ARG(ci0 >= 0 && nci >= 1 && ci0 + nci <= Cin, "%s: channel range [%d, %d) outside %d", who, ci0, ci0 + nci, Cin);
    if (!ws || ws_bytes < dgrad_workspace(g, nci, gr.n)) return cg_set_error(CG_ERR_WORKSPACE, "%s: workspace too small", who);
    static thread_local DgradPlan p;
    rc = plan_dgrad(g, nci, p);
    if (rc) return rc;
    hipStream_t st = cg_s(stream);
    float* wt = (float*)ws;
    // one launch re-lays-out the weights of every class (and member): wt_c[ci][tc][co]
    TransTable tt;
    int nz;
    dgrad_table(p, tt, nz);
    CG_CHECK_ARG(p.ws_floats < (size_t)0x7fffffff, "%s: weight tensor too large", who);
    rc = launch_transpose(w, wt, g->Cout, g->T, Cin, ci0, nci, tt, nz, st, false, 1.f, 0u, gr.n, gr.stride,
                          (long long)p.ws_floats, nullptr);
    if (rc) return rc;
    // batched pipelined launch when every class qualifies
    bool pipe = p.ncls <= 4;
    long m_total = 0;
    for (int c = 0; c < p.ncls && pipe; ++c) {
        pipe = pipe_ok(&p.cg[c], p.cg[c].T * p.cg[c].C1);
        m_total += (long)p.cg[c].N * p.cg[c].Ho * p.cg[c].Wo;
    }
    const Members mb{gr.n, 0, (long long)p.ws_floats * 4, 0};
    if (pipe) {
        PipeBatch b;
        for (int c = 0; c < p.ncls; ++c) fill_class(b.c[c], &p.cg[c], wt + p.w_off[c], gr.n);
        const int cfg = pick_fwd_cfg(&p.cg[0], m_total, true);
        if (cfg >= 20)
            return launch_pipe_cfg(cfg, b, p.ncls, dz, nullptr, dx,
                                   (unsigned)((size_t)g->N * g->Ho * g->Wo * g->Cout * sizeof(float)), st, nullptr, mb);
    }
    Grp gt;                      // the transposed weights of the members sit ws_floats apart
    gt.n = gr.n;
    gt.stride = (long long)p.ws_floats;
    for (int c = 0; c < p.ncls; ++c) {
        const cg_conv_geom* cgm = &p.cg[c];
        const int Ct = cgm->C1, K = cgm->T * Ct;
        const int Mm = (cgm->N / gr.n) * cgm->Ho * cgm->Wo;
        const bool fast = Ct % 32 == 0;
        const int cfg = pick_fwd_cfg(cgm, (long)Mm * gr.n, fast && pipe_ok(cgm, K));
        rc = launch_fwd_cfg(cfg, cgm, dz, nullptr, wt + p.w_off[c], nullptr, dx, Mm, K, fast, st, nullptr, gt);
        if (rc) return rc;
    }
    return CG_OK;
}

extern "C" int cg_conv2d_dgrad(const cg_conv_geom* g, const float* dz, const float* w, int ci0, int nci, float* dx,
                               void* ws, size_t ws_bytes, cg_stream_t stream) {
    return conv2d_dgrad_impl(g, nullptr, dz, w, ci0, nci, dx, ws, ws_bytes, stream, "cg_conv2d_dgrad");
}
extern "C" int cg_conv2d_dgrad_g(const cg_conv_geom* g, const cg_group* group, const float* dz, const float* w, int ci0,
                                 int nci, float* dx, void* ws, size_t ws_bytes, cg_stream_t stream) {
    return conv2d_dgrad_impl(g, group, dz, w, ci0, nci, dx, ws, ws_bytes, stream, "cg_conv2d_dgrad_g");
}

extern "C" int cg_act_bwd_split(const float* dy, const float* y, size_t n, int act, void* out, size_t lo_elems,
                                float* state, int dy_nslots, float* dz, cg_stream_t stream) {
    CG_CHECK_ARG(dy && y && out && state && n > 0 && x3_lo_ok(lo_elems, n * 2) && dy_nslots >= 0 && dy_nslots <= CG_AMAX_MAX_SLOTS,
                 "cg_act_bwd_split: bad args");
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipStream_t st = cg_s(stream);
    int nslots = dy_nslots;
    if (nslots <= 0 || dz) {      // nobody measured dy (or the fp32 dz is wanted too): one pass that measures dz itself
        nslots = amax_blocks(n);
        hipLaunchKernelGGL(act_bwd_amax_kernel, dim3(nslots), dim3(256), 0, st, dy, y, dz, n, act, state);
        CG_LAUNCH_CHECK("act_bwd_amax_kernel");
    }
    hipLaunchKernelGGL(act_bwd_split_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dy, y, (_Float16*)out, n, lo_elems, act,
                       state, nslots);
    CG_LAUNCH_CHECK("act_bwd_split_kernel");
    return CG_OK;
}

// ---- split-precision data-gradient ---------------------------------------------------------------------------------
// Two steps, so that the re-laid-out weights can be cached across the many launches that use one weight version:
//   cg_conv2d_dgrad_x3_prep:  w (fp32, [Cout][T][Cin]) of every member -> {hi, lo} fp16 planes of scale*w in the per-class
//                             [ci][tc][co] layout the data-gradient kernel reads, member m at wt + m * wt_elems elements
//                             (cg_conv2d_dgrad_x3_wt_elems; 4 bytes per element).  scale = w_scale, or the device-side
//                             value *w_scale_dev when given.
//   cg_conv2d_dgrad_x3_run:   dx from dz ({hi, lo} planes with its device-side scale) and the prepared weights.
// cg_conv2d_dgrad_x3 = both, with the prepared weights in the caller's workspace.
extern "C" size_t cg_conv2d_dgrad_x3_wt_elems(const cg_conv_geom* g, int nci) {
    if (!g || g->T < 1 || g->T > CG_MAX_TAPS || nci < 1) return 0;
    return (size_t)g->T * nci * g->Cout;
}

static int dgrad_x3_checks(const cg_conv_geom* g, const cg_group* group, int ci0, int nci, Grp& gr, DgradPlan& p, const char* who) {
    int rc = validate_geom(g, who);
    if (rc) return rc;
    rc = grp_from(group, g->N, gr, who);
    if (rc) return rc;
    const int Cin = g->C1 + g->C2;
    CG_CHECK_ARG(ci0 >= 0 && nci >= 1 && ci0 + nci <= Cin, "%s: channel range outside %d", who, Cin);
    CG_CHECK_ARG(g->Cout % BK == 0, "%s: Cout %% 32 != 0", who);
    rc = plan_dgrad(g, nci, p);
    if (rc) return rc;
    CG_CHECK_ARG(p.ncls <= 4, "%s: stride %d has more than 4 output classes", who, g->stride);
    CG_CHECK_ARG(4 * p.ws_floats < (size_t)CG_OOB, "%s: weight tensor too large", who);
    CG_CHECK_ARG(gr.n == 1 || (CG_X3_INTERLEAVE && p.ws_floats % 32 == 0), "%s: grouped launches need the interleaved layout", who);
    return CG_OK;
}

extern "C" int cg_conv2d_dgrad_x3_prep(const cg_conv_geom* g, const cg_group* group, const float* w, int ci0, int nci,
                                       float w_scale, const float* w_scale_dev, void* wt, size_t wt_bytes,
                                       cg_stream_t stream) {
    static thread_local DgradPlan p;
    Grp gr;
    int rc = dgrad_x3_checks(g, group, ci0, nci, gr, p, "cg_conv2d_dgrad_x3_prep");
    if (rc) return rc;
    CG_CHECK_ARG(w && wt && w_scale > 0.f, "cg_conv2d_dgrad_x3_prep: null pointer / bad scale");
    if (wt_bytes < (size_t)gr.n * p.ws_floats * 4) return cg_set_error(CG_ERR_WORKSPACE, "cg_conv2d_dgrad_x3_prep: buffer too small");
    TransTable tt;
    int nz;
    dgrad_table(p, tt, nz);
    const size_t wt_lo = CG_X3_INTERLEAVE ? (size_t)CG_X3_LO_ELEMS : p.ws_floats;      // lo offset of the re-laid-out weights
    CG_CHECK_ARG(gr.n == 1 || CG_X3_INTERLEAVE, "cg_conv2d_dgrad_x3_prep: grouped launches need the interleaved layout");
    return launch_transpose(w, (float*)wt, g->Cout, g->T, g->C1 + g->C2, ci0, nci, tt, nz, cg_s(stream), true, w_scale,
                            (unsigned)wt_lo, gr.n, gr.stride, (long long)p.ws_floats, w_scale_dev);
}

static int conv2d_dgrad_x3_run_impl(const cg_conv_geom* g, const cg_group* group, const void* dz_split, size_t dz_lo_elems,
                                    const float* dz_scale_dev, const void* wt, float w_scale, const float* w_scale_dev,
                                    int ci0, int nci, float* dx, void* dx_split, size_t dx_lo_elems, const cg_x3_epilogue* epi,
                                    float* amax_state, int* amax_nslots, cg_stream_t stream) {
    static thread_local DgradPlan p;
    Grp gr;
    CG_CHECK_ARG((amax_state == nullptr) == (amax_nslots == nullptr), "cg_conv2d_dgrad_x3_run: amax_state and amax_nslots go together");
    if (amax_nslots) *amax_nslots = 0;
    int rc = dgrad_x3_checks(g, group, ci0, nci, gr, p, "cg_conv2d_dgrad_x3_run");
    if (rc) return rc;
    CG_CHECK_ARG(dz_split && wt && (dx || (dx_split && epi)) && dz_scale_dev && w_scale > 0.f, "cg_conv2d_dgrad_x3_run: null pointer / bad scale");
    rc = x3_epilogue_check(epi, dx_split, amax_state, "cg_conv2d_dgrad_x3_run");
    if (rc) return rc;
    CG_CHECK_ARG(!dx_split || x3_lo_ok(dx_lo_elems, (size_t)g->N * (g->H << g->up) * (g->W << g->up) * nci * 2), "cg_conv2d_dgrad_x3_run: bad dx lo offset");
    const size_t wt_elems = p.ws_floats;          // one fp16 plane = as many elements as the fp32 layout had floats
    const size_t dz_plane = (size_t)g->N * g->Ho * g->Wo * g->Cout * 2;
    CG_CHECK_ARG(x3_lo_ok(dz_lo_elems, dz_plane) && x3_span(dz_lo_elems, dz_plane) < (size_t)CG_OOB,
                 "cg_conv2d_dgrad_x3_run: operand planes out of range / lo offset does not match the layout");
    const size_t wt_lo = CG_X3_INTERLEAVE ? (size_t)CG_X3_LO_ELEMS : wt_elems;
    PipeBatch b;
    long m_total = 0;
    for (int c = 0; c < p.ncls; ++c) {
        fill_class(b.c[c], &p.cg[c], (const float*)((const _Float16*)wt + cg_il(p.w_off[c])), gr.n);   // class sizes: multiples of 32
        b.c[c].w_bytes = (unsigned)(wt_lo * 2);                                                     // lo offset
        b.c[c].pad_ = (int32_t)(4 * wt_elems - 2 * cg_il(p.w_off[c]));                             // span from this class's base
        m_total += (long)b.c[c].M * gr.n;
    }
    X3Extra ex;
    ex.w_scale_dev = w_scale_dev;
    ex.mb = Members{gr.n, 0, (long long)wt_elems * 4, 0};
    // per-block maxima of dx for whoever splits it next (cg_act_bwd_split): every output-parity class is part of this launch,
    // so the blocks' maxima cover the whole tensor
    fwd_amax.state = amax_state;
    fwd_amax.nslots = 0;
    fwd_amax.all_classes = true;
    ex.epi = x3_epilogue_of(epi);
    rc = launch_x3_cfg(pick_x3_cfg(nci, m_total, g->Cout, p.cg[0].T), b, p.ncls, dz_split, nullptr, dx, (unsigned)(dz_lo_elems * 2),
                       (unsigned)x3_span(dz_lo_elems, dz_plane), 1.0f / w_scale, dz_scale_dev, cg_s(stream), nullptr, dx_split,
                       dx_lo_elems, ex);
    if (amax_nslots && !rc) *amax_nslots = fwd_amax.nslots;
    fwd_amax.state = nullptr;
    fwd_amax.all_classes = false;
    return rc;
}

extern "C" int cg_conv2d_dgrad_x3_run(const cg_conv_geom* g, const cg_group* group, const void* dz_split, size_t dz_lo_elems,
                                      const float* dz_scale_dev, const void* wt, float w_scale, const float* w_scale_dev,
                                      int ci0, int nci, float* dx, float* amax_state, int* amax_nslots, cg_stream_t stream) {
    return conv2d_dgrad_x3_run_impl(g, group, dz_split, dz_lo_elems, dz_scale_dev, wt, w_scale, w_scale_dev, ci0, nci, dx, nullptr, 0,
                                    nullptr, amax_state, amax_nslots, stream);
}
// ... with the epilogue extras: dx as {hi, lo} planes on an a-priori scale (dx itself may then be NULL) and / or multiplied by
// the activation derivative of the layer below (cg_x3_epilogue)
extern "C" int cg_conv2d_dgrad_x3_run_e(const cg_conv_geom* g, const cg_group* group, const void* dz_split, size_t dz_lo_elems,
                                        const float* dz_scale_dev, const void* wt, float w_scale, const float* w_scale_dev,
                                        int ci0, int nci, float* dx, void* dx_split, size_t dx_lo_elems, const cg_x3_epilogue* epi,
                                        float* amax_state, int* amax_nslots, cg_stream_t stream) {
    CG_CHECK_ARG(epi, "cg_conv2d_dgrad_x3_run_e: the epilogue descriptor is required");
    return conv2d_dgrad_x3_run_impl(g, group, dz_split, dz_lo_elems, dz_scale_dev, wt, w_scale, w_scale_dev, ci0, nci, dx, dx_split,
                                    dx_lo_elems, epi, amax_state, amax_nslots, stream);
}

extern "C" int cg_conv2d_dgrad_x3(const cg_conv_geom* g, const void* dz_split, size_t dz_lo_elems,
                                  const float* dz_scale_dev, const float* w, int ci0, int nci, float* dx, void* ws,
                                  size_t ws_bytes, cg_stream_t stream) {
    if (!ws || ws_bytes < cg_conv2d_dgrad_workspace(g, nci))
        return cg_set_error(CG_ERR_WORKSPACE, "cg_conv2d_dgrad_x3: workspace too small");
    int rc = cg_conv2d_dgrad_x3_prep(g, nullptr, w, ci0, nci, CG_X3_WSCALE, nullptr, ws, ws_bytes, stream);
    if (rc) return rc;
    return cg_conv2d_dgrad_x3_run(g, nullptr, dz_split, dz_lo_elems, dz_scale_dev, ws, CG_X3_WSCALE, nullptr, ci0, nci, dx, nullptr,
                                  nullptr, stream);
}

// ---- nearest-2x upsample + 3x3 convolution = a 4x4 stride-2 TRANSPOSED convolution ---------------------------------
// nn.Upsample(scale_factor=2) -> ZeroPad2d(1) -> Conv2d(3x3) (networks.py:385-386, Conv2dBlock :513-516).  The upsampled image
// repeats every source pixel 2x2, so the 3x3 taps that land on the SAME source pixel can be added up first: output row
// Y = 2i - 1 + u (u = 0..3) receives source row i through the summed kernel rows  S(0) = {2}, S(1) = {1,2}, S(2) = {0,1},
// S(3) = {0}  (likewise for columns) -- 16 effective taps for 4 output pixels instead of 36: 2.25x fewer multiply-adds,
// forward and backward.  With W_F[u][v] = sum_{kh in S(u), kw in S(v)} W[kh][kw]:
//   forward        y  = conv_transpose(x, W_F, stride 2, pad 1) = the four output-parity classes (2x2 taps each) of the
//                       data-gradient machinery above, with bias and instance-norm partials        (cg_upconv2d_fwd_x3)
//   data gradient  dx = conv(dz, W_F as [Cin][4][4][Cout], stride 2, pad 1)                         (cg_conv2d_fwd_x3_g)
//   weight grad    dW_F[ci][u][v][co] = wgrad of that 4x4 stride-2 conv (input dz, output gradient x)  (cg_conv2d_wgrad_x3_g),
//                  folded back onto the nine taps: dW[co][kh][kw][ci] += sum_{u: kh in S(u), v: kw in S(v)} dW_F[ci][u][v][co]
// The sums are formed in fp32 from the fp32 weights and then split: same 22-bit products as every other convolution here.
namespace {
__host__ __device__ inline int upc_first(int u) { return u == 0 ? 2 : (u == 1 ? 1 : 0); }      // S(u) = {first, .., first + n - 1}
__host__ __device__ inline int upc_count(int u) { return (u == 1 || u == 2) ? 2 : 1; }

// mode 0: forward classes.  class (p, q), tap (a, b) <-> u = p + 1 - 2*dy with dy = a - 1 (p == 0) / a (p == 1):
//         out[cls * Cout*4*Cin + (co*4 + a*2+b) * Cin + ci];   mode 1: backward, out[(ci*16 + u*4+v) * Cout + co]
__global__ __launch_bounds__(256) void upconv_prep_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int Cout,
                                                          int Cin, float scale, const float* __restrict__ scale_dev,
                                                          unsigned lo_elems, long long w_mstride, int mode) {
    const size_t per = (size_t)16 * Cout * Cin;
    const int member = blockIdx.y;
    w += (long long)member * w_mstride;
    out += (size_t)member * per * 2;          // 4 bytes per element in the interleaved {hi, lo} form
    if (scale_dev) scale *= scale_dev[0];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (size_t)gridDim.x * 256) {
        int co, ci, u, v;
        size_t o;
        if (mode == 0) {      // i = ((cls*Cout + co)*4 + tap)*Cin + ci
            ci = (int)(i % Cin);
            size_t r = i / Cin;
            const int tap = (int)(r & 3);
            r >>= 2;
            co = (int)(r % Cout);
            const int cls = (int)(r / Cout);
            const int p = cls >> 1, q = cls & 1, a = tap >> 1, b = tap & 1;
            u = p + 1 - 2 * (p ? a : a - 1);
            v = q + 1 - 2 * (q ? b : b - 1);
            o = i;
        } else {              // i = (ci*16 + uv)*Cout + co
            co = (int)(i % Cout);
            size_t r = i / Cout;
            const int uv = (int)(r & 15);
            ci = (int)(r >> 4);
            u = uv >> 2;
            v = uv & 3;
            o = i;
        }
        float acc = 0.f;
        for (int kh = upc_first(u); kh < upc_first(u) + upc_count(u); ++kh)
            for (int kw = upc_first(v); kw < upc_first(v) + upc_count(v); ++kw) acc += w[((size_t)co * 9 + kh * 3 + kw) * Cin + ci];
        _Float16 h, l;
        split_f16(acc * scale, h, l);
        out[cg_il(o)] = h;
        out[lo_elems + cg_il(o)] = l;
    }
}

// dw[co][kh][kw][ci] (+)= sum over the (u, v) whose S(u) x S(v) contains (kh, kw) of dwf[ci][u][v][co]
__global__ __launch_bounds__(256) void upconv_fold_kernel(const float* __restrict__ dwf, float* __restrict__ dw, int Cout, int Cin,
                                                          long long dw_mstride, int accumulate) {
    __shared__ float tile[32][33];
    const int member = blockIdx.z / 9, t = blockIdx.z % 9, kh = t / 3, kw = t % 3;
    dwf += (size_t)member * 16 * Cout * Cin;
    dw += (long long)member * dw_mstride;
    const int cib = blockIdx.x * 32, cob = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {        // tile[ci][co], co fastest in dwf
        const int ci = cib + r, co = cob + tx;
        float a = 0.f;
        if (ci < Cin && co < Cout)
            for (int u = 0; u < 4; ++u) {
                if (kh < upc_first(u) || kh >= upc_first(u) + upc_count(u)) continue;
                for (int v = 0; v < 4; ++v) {
                    if (kw < upc_first(v) || kw >= upc_first(v) + upc_count(v)) continue;
                    a += dwf[((size_t)ci * 16 + u * 4 + v) * Cout + co];
                }
            }
        tile[r][tx] = a;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {        // ci fastest in dw
        const int co = cob + r, ci = cib + tx;
        if (co < Cout && ci < Cin) {
            float* o = dw + ((size_t)co * 9 + t) * Cin + ci;
            *o = accumulate ? *o + tile[tx][r] : tile[tx][r];
        }
    }
}

// column sums of a {hi, lo} tensor [rows][C] (the bias gradient of a layer whose dz exists in split form only): one block per
// (member, row chunk) writes C partial sums, a second launch adds the chunks in a fixed order
__global__ __launch_bounds__(256) void colsum_split_kernel(const _Float16* __restrict__ zs, unsigned lo_elems, int rows, int C,
                                                           int chunks, float* __restrict__ part) {
    // a thread owns 8 consecutive channels (one 16-byte load per plane and row: cg_il keeps 8-aligned runs contiguous), C / 8
    // threads a row, 256 / (C / 8) rows in flight per block; rows of a chunk are summed in a fixed order
    const int member = blockIdx.y, chunk = blockIdx.x;
    const int per = (rows + chunks - 1) / chunks;
    const int r0 = chunk * per, r1 = min(rows, r0 + per);
    const int tpr = C >> 3, cg = threadIdx.x % tpr, rr = threadIdx.x / tpr, rstep = 256 / tpr;      // C % 8 == 0, 256 % (C / 8) == 0
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = r0 + rr; r < r1; r += rstep) {
        const size_t e = cg_il(((size_t)member * rows + r) * C + (size_t)cg * 8);
        const uint4 h4 = *reinterpret_cast<const uint4*>(zs + e), l4 = *reinterpret_cast<const uint4*>(zs + lo_elems + e);
        const _Float16* h = reinterpret_cast<const _Float16*>(&h4);
        const _Float16* l = reinterpret_cast<const _Float16*>(&l4);
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] += (float)h[i] + (float)l[i];
    }
    __shared__ float red[256 * 8];
#pragma unroll
    for (int i = 0; i < 8; ++i) red[(rr * tpr + cg) * 8 + i] = a[i];
    __syncthreads();
    if (threadIdx.x < C) {
        float t = 0.f;
        for (int k = 0; k < rstep; ++k) t += red[k * C + threadIdx.x];      // (k * tpr + c / 8) * 8 + c % 8 = k * C + c
        part[((size_t)member * chunks + chunk) * C + threadIdx.x] = t;
    }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int C, int chunks,
                                                           const float* __restrict__ scale_dev, float* __restrict__ db,
                                                           long long db_mstride, int accumulate) {
    // 256 / C threads per column, each over every (256 / C)-th chunk (independent loads), joined in a fixed order
    const int member = blockIdx.x, c = threadIdx.x % C, pp = threadIdx.x / C, np = 256 / C;
    float a = 0.f;
    for (int k = pp; k < chunks; k += np) a += part[((size_t)member * chunks + k) * C + c];
    __shared__ float red[256];
    red[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x >= C) return;
    for (int k = 1; k < np; ++k) a += red[k * C + c];
    if (scale_dev) a *= 1.f / scale_dev[0];
    float* o = db + (long long)member * db_mstride + c;
    *o = accumulate ? *o + a : a;
}
}  // namespace

extern "C" size_t cg_upconv_wt_elems(int Cout, int Cin) { return (size_t)16 * Cout * Cin; }

extern "C" int cg_upconv_prep_x3(const cg_group* group, const float* w, int Cout, int Cin, float w_scale,
                                 const float* w_scale_dev, void* wt_fwd, void* wt_bwd, cg_stream_t stream) {
    CG_CHECK_ARG(w && (wt_fwd || wt_bwd) && Cout > 0 && Cin > 0 && Cout % 32 == 0 && Cin % 32 == 0 && w_scale > 0.f,
                 "cg_upconv_prep_x3: bad args (channel counts must be multiples of 32)");
    CG_CHECK_ARG(CG_X3_INTERLEAVE, "cg_upconv_prep_x3: needs the interleaved operand layout");
    const int n = group ? group->n : 1;
    const long long ms = group ? group->stride : 0;
    CG_CHECK_ARG(n >= 1, "cg_upconv_prep_x3: bad group");
    const size_t per = cg_upconv_wt_elems(Cout, Cin);
    dim3 grid((unsigned)std::min<size_t>((per + 255) / 256, 1024), n);
    if (wt_fwd)
        hipLaunchKernelGGL(upconv_prep_kernel, grid, dim3(256), 0, cg_s(stream), w, (_Float16*)wt_fwd, Cout, Cin, w_scale,
                           w_scale_dev, (unsigned)CG_X3_LO_ELEMS, ms, 0);
    if (wt_bwd)
        hipLaunchKernelGGL(upconv_prep_kernel, grid, dim3(256), 0, cg_s(stream), w, (_Float16*)wt_bwd, Cout, Cin, w_scale,
                           w_scale_dev, (unsigned)CG_X3_LO_ELEMS, ms, 1);
    CG_LAUNCH_CHECK("upconv_prep_kernel");
    return CG_OK;
}

extern "C" int cg_upconv2d_fwd_x3(const cg_conv_geom* g, const cg_group* group, const void* xs, size_t x_lo_elems,
                                  const float* x_scale_dev, const void* wt_fwd, float w_scale, const float* w_scale_dev,
                                  const float* bias, float* y, double* stats, size_t stats_bytes, int* rows_per_partial,
                                  cg_stream_t stream) {
    const char* who = "cg_upconv2d_fwd_x3";
    int rc = validate_geom(g, who);
    if (rc) return rc;
    if (rows_per_partial) *rows_per_partial = 0;
    CG_CHECK_ARG(xs && wt_fwd && y && w_scale > 0.f, "%s: null pointer / bad scale", who);
    CG_CHECK_ARG(CG_X3_INTERLEAVE && g->up == 1 && g->T == 9 && g->stride == 1 && g->C2 == 0 && g->C1 % 32 == 0 && g->Cout % 32 == 0 &&
                     g->Ho == 2 * g->H && g->Wo == 2 * g->W && g->osy == 1 && g->osx == 1,
                 "%s: needs a 3x3 stride-1 pad-1 convolution on a 2x upsampled source, channels multiples of 32", who);
    for (int t = 0; t < 9; ++t)
        CG_CHECK_ARG(g->dy[t] == t / 3 - 1 && g->dx[t] == t % 3 - 1, "%s: taps must be the 3x3 pad-1 window", who);
    Grp gr;
    rc = grp_from(group, g->N, gr, who);
    if (rc) return rc;
    const size_t x_plane = (size_t)g->N * g->H * g->W * g->C1 * 2;
    CG_CHECK_ARG(x3_lo_ok(x_lo_elems, x_plane) && x3_span(x_lo_elems, x_plane) < (size_t)CG_OOB, "%s: operand planes out of range", who);
    const size_t cls_elems = (size_t)g->Cout * 4 * g->C1, wt_elems = 4 * cls_elems;
    CG_CHECK_ARG(4 * wt_elems < (size_t)CG_OOB, "%s: weight tensor too large", who);
    PipeBatch b;
    long m_total = 0;
    for (int c = 0; c < 4; ++c) {
        const int p = c >> 1, q = c & 1;
        cg_conv_geom cg;
        memset(&cg, 0, sizeof(cg));
        cg.N = g->N; cg.H = g->H; cg.W = g->W; cg.C1 = g->C1; cg.C2 = 0; cg.up = 0;
        cg.Ho = g->H; cg.Wo = g->W; cg.HoF = g->Ho; cg.WoF = g->Wo;
        cg.osy = cg.osx = 2; cg.ooy = p; cg.oox = q;
        cg.stride = 1; cg.T = 4; cg.Cout = g->Cout; cg.act = g->act;
        for (int a = 0; a < 2; ++a)
            for (int bb = 0; bb < 2; ++bb) {
                cg.dy[a * 2 + bb] = (int8_t)(p ? a : a - 1);
                cg.dx[a * 2 + bb] = (int8_t)(q ? bb : bb - 1);
            }
        fill_class(b.c[c], &cg, (const float*)((const _Float16*)wt_fwd + cg_il(c * cls_elems)), gr.n);
        b.c[c].w_bytes = (unsigned)(CG_X3_LO_ELEMS * 2);
        b.c[c].pad_ = (int32_t)(4 * wt_elems - 2 * cg_il(c * cls_elems));
        m_total += (long)b.c[c].M * gr.n;
    }
    const int cfg = pick_x3_cfg(g->Cout, m_total, g->C1, 4);
    const int bm = x3_cfg_bm(cfg);
    double* st_ptr = nullptr;
    if (rows_per_partial && stats && g->act == CG_ACT_NONE && (g->H * g->W) % bm == 0 &&
        stats_bytes >= (size_t)(m_total / bm) * g->Cout * 2 * sizeof(double)) {
        st_ptr = stats;              // partial T = (row tile) * 4 + class: the tiles of a sample stay consecutive
        *rows_per_partial = bm;
    }
    X3Extra ex;
    ex.w_scale_dev = w_scale_dev;
    ex.mb = Members{gr.n, 0, (long long)wt_elems * 4, gr.stride * 4};
    fwd_amax.state = nullptr;
    return launch_x3_cfg(cfg, b, 4, xs, bias, y, (unsigned)(x_lo_elems * 2), (unsigned)x3_span(x_lo_elems, x_plane), 1.0f / w_scale,
                         x_scale_dev, cg_s(stream), st_ptr, nullptr, 0, ex);
}

extern "C" int cg_upconv_fold_dw(const cg_group* group, const float* dwf, float* dw, int Cout, int Cin, int accumulate,
                                 cg_stream_t stream) {
    CG_CHECK_ARG(dwf && dw && Cout > 0 && Cin > 0, "cg_upconv_fold_dw: bad args");
    const int n = group ? group->n : 1;
    const long long ms = group ? group->stride : 0;
    hipLaunchKernelGGL(upconv_fold_kernel, dim3(cg_div_up(Cin, 32), cg_div_up(Cout, 32), 9 * n), dim3(256), 0, cg_s(stream), dwf, dw,
                       Cout, Cin, ms, accumulate);
    CG_LAUNCH_CHECK("upconv_fold_kernel");
    return CG_OK;
}

extern "C" size_t cg_colsum_split_workspace(int C, int nmember) { return (size_t)nmember * 256 * C * sizeof(float); }

extern "C" int cg_colsum_split(const cg_group* group, const void* zs, size_t lo_elems, const float* scale_dev, long rows_total,
                               int C, float* db, int accumulate, void* ws, size_t ws_bytes, cg_stream_t stream) {
    const int n = group ? group->n : 1;
    const long long ms = group ? group->stride : 0;
    CG_CHECK_ARG(zs && db && ws && C >= 8 && C <= 256 && 256 % C == 0 && rows_total > 0 && rows_total % n == 0 &&
                     x3_lo_ok(lo_elems, (size_t)rows_total * C * 2),
                 "cg_colsum_split: bad args (C must divide 256 and be a multiple of 8)");
    if (ws_bytes < cg_colsum_split_workspace(C, n)) return cg_set_error(CG_ERR_WORKSPACE, "cg_colsum_split: workspace too small");
    const int rows = (int)(rows_total / n), chunks = 256;
    hipLaunchKernelGGL(colsum_split_kernel, dim3(chunks, n), dim3(256), 0, cg_s(stream), (const _Float16*)zs, (unsigned)lo_elems, rows,
                       C, chunks, (float*)ws);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(n), dim3(256), 0, cg_s(stream), (const float*)ws, C, chunks, scale_dev, db, ms,
                       accumulate);
    CG_LAUNCH_CHECK("colsum_split_kernel");
    return CG_OK;
}

extern "C" int cg_x3_interleaved(void) { return CG_X3_INTERLEAVE; }

extern "C" int cg_debug_fetch(long long* host, int nwords) {
    CG_CHECK_ARG(host && nwords > 0 && nwords <= 4096 * CG_DBG_WORDS, "cg_debug_fetch: bad args");
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(cg_dbg), (size_t)nwords * sizeof(long long));
    if (e != hipSuccess) return cg_set_error(CG_ERR_LAUNCH, "cg_debug_fetch: %s", hipGetErrorString(e));
    return CG_OK;
}

extern "C" int cg_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(prof_mu);
    prof_on = on != 0;
    return CG_OK;
}

extern "C" int cg_prof_collect(int64_t* counts, double* ms, double* flops) {
    CG_CHECK_ARG(counts && ms && flops, "cg_prof_collect: null pointer");
    std::lock_guard<std::mutex> lk(prof_mu);
    for (int i = 0; i < CG_PROF_SLOTS; ++i) {
        counts[i] = 0;
        ms[i] = 0.0;
        flops[i] = 0.0;
    }
    struct Agg { long n = 0; double ms = 0, fl = 0; };
    std::map<std::string, Agg> by_shape;
    for (auto& r : prof_recs) {
        (void)hipEventSynchronize(r.e1);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, r.e0, r.e1);
        counts[r.slot] += 1;
        ms[r.slot] += (double)t;
        flops[r.slot] += r.flops;
        Agg& a = by_shape[r.key];
        a.n += 1;
        a.ms += (double)t;
        a.fl += r.flops;
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    prof_recs.clear();
    prof_report.clear();
    char line[192];
    for (auto& kv : by_shape) {
        snprintf(line, sizeof(line), "%-70s n=%4ld  %9.3f ms  avg %8.1f us  %6.1f TF\n", kv.first.c_str(), kv.second.n, kv.second.ms,
                 1000.0 * kv.second.ms / kv.second.n, kv.second.ms > 0 ? kv.second.fl / (kv.second.ms * 1e9) : 0.0);
        prof_report += line;
    }
    return CG_OK;
}

extern "C" const char* cg_prof_report(void) { return prof_report.c_str(); }

extern "C" const char* cg_prof_slot_name(int slot) {
    if (slot < 0 || slot >= CG_PROF_SLOTS) return "";
    return prof_names[slot];
}

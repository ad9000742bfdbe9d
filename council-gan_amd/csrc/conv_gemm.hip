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
    int32_t M, K, tiles_n, ntiles;
    uint32_t w_bytes;
    int32_t pad_;
    // grouped launches (one class per council member, conv_fwd_x3_kernel only): when xs is set the class brings its own
    // activation / bias / output / scale pointers instead of the launch-wide ones
    const void* xs;
    const float* bias;
    float* y;
    const float* x_scale;
};
struct PipeBatch {
    PipeClass c[4];
};

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
        if (gridDim.x <= 1024) state[2 + blockIdx.x] = m;
        else atomicMax(reinterpret_cast<unsigned*>(state + 2 + (blockIdx.x & 1023)), __float_as_uint(m));
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
    const float* __restrict__ bias, float* __restrict__ y, int M, int K, int tiles_n, float* __restrict__ amax_state) {
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
    const int m0 = (tile / tiles_n) * BM;
    const int n0 = (tile % tiles_n) * BN;
    const int Ct = g.C1 + g.C2;

    if (tid < g.T) taps[tid] = load_tap(tid);
    for (int r = tid; r < BM; r += NT) rows[r] = decode_row(g, m0 + r, M, true);
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

template <int BM, int BN, int WM, int WN, int PF, int ABL>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void conv_fwd_pipe_kernel(
    PipeBatch batch, const float* __restrict__ x1, const float* __restrict__ bias, float* __restrict__ y,
    unsigned x_bytes, double* __restrict__ stats) {
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
    const float* __restrict__ w = pc->w;
    const int M = pc->M, K = pc->K, tiles_n = pc->tiles_n;
    const unsigned w_bytes = pc->w_bytes;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int tile = xcd_remap(blockIdx.x, ntiles);
    const int m0 = (tile / tiles_n) * BM;
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
    float4 a0[TM], b0[TN], a1[TM], b1[TN];
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
    if (ABL != 2 && NW >= 8 && wid >= NW / 2) {
        for (int kt = 0; kt + 1 < nk; kt += 2) {
            slice(I0(), I1());
            slice(I1(), I1());
        }
        if (nk & 1) slice(I0(), I1());
    } else {
        for (int kt = 0; kt + 1 < nk; kt += 2) {
            slice(I0(), I0());
            slice(I1(), I0());
        }
        if (nk & 1) slice(I0(), I0());
    }
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
            double* o = stats + ((size_t)(tile / tiles_n) * g.Cout + n0 + tid) * 2;
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

    auto decode_rows = [&](int slice, int buf) {
        if (tid < BP) {
            rows[buf][tid] = decode_row(g, slice * BP + tid, M, false);
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

    // partial layout per split: [Cout*K weight partials | Cout bias partials]
    float* dst = out + (size_t)split * ((size_t)g.Cout * K + g.Cout);
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
    auto load_tile = [&]() {
        const int mb = ld_s * BP;
#pragma unroll
        for (int i = 0; i < D_V4; ++i) {
            const int m = mb + d_r0 + D_RP * i;
            dv[i] = buf_load4(dr, ((((unsigned)m * (unsigned)g.Cout) + d_col) << 2) | d_oob | (m < M ? 0u : CG_OOB));
        }
#pragma unroll
        for (int i = 0; i < X_V4; ++i) {
            const int m = mb + x_r0 + X_RP * i;
            const int n = m >> lg_hw, rem = m & hw_mask;
            const int ly = (rem >> lg_wo) * g.stride + dy, lx = (rem & wo_mask) * g.stride + dx;
            const bool ok = m < M && (unsigned)ly < (unsigned)Hl && (unsigned)lx < (unsigned)Wl;
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

    // partial layout per split: [Cout*K weight partials | Cout bias partials]
    float* dst = out + (size_t)split * ((size_t)g.Cout * K + g.Cout);
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
                                                            int accumulate) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t i = t / L;
    const int j = (int)(t % L);
    const size_t stride = nw + (size_t)nb;
    const size_t n = nw + (dbias ? (size_t)nb : 0);
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
                                                               float scale, unsigned lo_elems) {
    __shared__ float tile[32][33];
    typedef const __attribute__((address_space(4))) int32_t* KI;
    KI tab = (KI)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() +
                  offsetof(TransArgs, tt));
    const int z = blockIdx.z;
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
    if (bm == 256) return 6;
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
        rec.slot = family * 16 + tile_id(bm, bn) * 2 + (fast ? 1 : 0);
        rec.flops = flops;
        static const char* const fam[6] = {"conv_fwd_kernel", "conv_wgrad_kernel", "conv_fwd_pipe_kernel",
                                           "conv_wgrad_pipe_kernel", "conv_fwd_x3_kernel", "conv_wgrad_x3_kernel"};
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

// Per-launch option handed from the C entry point to the launcher that ends up running (set and cleared around one call on
// the calling thread): where the kernel's epilogue should leave its per-block output maxima, and how many it left.
struct FwdAmax {
    float* state = nullptr;
    int nslots = 0;
};
static thread_local FwdAmax fwd_amax;
constexpr int CG_AMAX_SLOTS_MAX = 1024;     // == CG_AMAX_MAX_SLOTS of conv_x3.inc (state[2 .. 2 + 1024))
// slots a launch of `blocks` blocks fills (block_amax_store): one each, or 1024 shared ones that must start at zero
static int amax_slots_for(long blocks, float* state, hipStream_t st) {
    if (blocks <= CG_AMAX_SLOTS_MAX) return (int)blocks;
    static const bool no_atomic = getenv("CG_NO_AMAX_ATOMIC") != nullptr;      // A/B switch
    if (no_atomic) return 0;
    (void)hipMemsetAsync(state + 2, 0, CG_AMAX_SLOTS_MAX * sizeof(float), st);
    return CG_AMAX_SLOTS_MAX;
}

template <int BM, int BN, int WM, int WN, int STAGES>
int launch_fwd(const cg_conv_geom* g, const float* x1, const float* x2, const float* w, const float* bias, float* y,
               int M, int K, bool fast, hipStream_t st) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (g->Cout + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n), block(NT);
    float* amax = nullptr;
    if (fwd_amax.state && g->osy == 1 && g->osx == 1) {
        fwd_amax.nslots = amax_slots_for((long)tiles_m * tiles_n, fwd_amax.state, st);
        if (fwd_amax.nslots) amax = fwd_amax.state;
    }
    ProfScope prof(0, BM, BN, fast, 2.0 * (double)M * (double)g->Cout * (double)K, st, g);
    if (fast)
        hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WM, WN, true, STAGES>), grid, block, 0, st, *g, x1, x2, w, bias, y, M,
                           K, tiles_n, amax);
    else
        hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WM, WN, false, STAGES>), grid, block, 0, st, *g, x1, x2, w, bias, y,
                           M, K, tiles_n, amax);
    CG_LAUNCH_CHECK("conv_fwd_kernel");
    return CG_OK;
}

template <int BM, int BN, int WM, int WN, int PF = 1, int ABL = 0>
int launch_pipe_batch(PipeBatch& b, int ncls, const float* x1, const float* bias, float* y, unsigned x_bytes, hipStream_t st,
                      double* stats = nullptr, int* stats_rows = nullptr) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    int max_tiles = 0;
    double flops = 0.0;
    for (int c = 0; c < ncls; ++c) {
        PipeClass& pc = b.c[c];
        pc.tiles_n = (pc.g.Cout + BN - 1) / BN;
        pc.ntiles = ((pc.M + BM - 1) / BM) * pc.tiles_n;
        if (pc.ntiles > max_tiles) max_tiles = pc.ntiles;
        flops += 2.0 * (double)pc.M * (double)pc.g.Cout * (double)pc.K;
    }
    for (int c = ncls; c < 4; ++c) b.c[c].ntiles = 0;
    dim3 grid(max_tiles, ncls), block(NT);
    ProfScope prof(2, BM, BN, true, flops, st, &b.c[0].g, ncls);
    if (stats_rows) *stats_rows = BM;
    hipLaunchKernelGGL((conv_fwd_pipe_kernel<BM, BN, WM, WN, PF, ABL>), grid, block, 0, st, b, x1, bias, y, x_bytes, stats);
    CG_LAUNCH_CHECK("conv_fwd_pipe_kernel");
    return CG_OK;
}

// configurations whose launch forwards the statistics pointer to the kernel (the measurement variants 27-31 do not)
bool pipe_cfg_has_stats(int cfg) { return (cfg >= 20 && cfg <= 26) || cfg == 32; }
int pipe_cfg_bm(int cfg) { return cfg == 23 || cfg == 26 ? 64 : (cfg == 24 || cfg == 32 ? 256 : 128); }

int launch_pipe_cfg(int cfg, PipeBatch& b, int ncls, const float* x1, const float* bias, float* y, unsigned x_bytes,
                    hipStream_t st, double* stats = nullptr) {
    switch (cfg) {
        case 20: return launch_pipe_batch<128, 128, 64, 32>(b, ncls, x1, bias, y, x_bytes, st, stats);  // 8 waves
        case 21: return launch_pipe_batch<128, 128, 64, 64>(b, ncls, x1, bias, y, x_bytes, st, stats);  // 4 waves
        case 22: return launch_pipe_batch<128, 64, 64, 32>(b, ncls, x1, bias, y, x_bytes, st, stats);   // 4 waves
        case 23: return launch_pipe_batch<64, 64, 32, 32>(b, ncls, x1, bias, y, x_bytes, st, stats);    // 4 waves
        case 24: return launch_pipe_batch<256, 128, 64, 64>(b, ncls, x1, bias, y, x_bytes, st, stats);  // 8 waves
        case 25: return launch_pipe_batch<128, 64, 32, 32>(b, ncls, x1, bias, y, x_bytes, st, stats);   // 8 waves
        case 26: return launch_pipe_batch<64, 128, 32, 64>(b, ncls, x1, bias, y, x_bytes, st, stats);   // 4 waves
        case 27: return launch_pipe_batch<128, 128, 64, 32, 2>(b, ncls, x1, bias, y, x_bytes, st);     // prefetch 2
        case 28: return launch_pipe_batch<128, 128, 64, 32, 1, 1>(b, ncls, x1, bias, y, x_bytes, st);  // ablation
        case 29: return launch_pipe_batch<128, 128, 64, 32, 1, 2>(b, ncls, x1, bias, y, x_bytes, st);  // ablation
        case 30: return launch_pipe_batch<128, 128, 64, 64, 2>(b, ncls, x1, bias, y, x_bytes, st);     // 4 waves, prefetch 2
        case 31: return launch_pipe_batch<128, 128, 64, 32, 1, 3>(b, ncls, x1, bias, y, x_bytes, st);  // timing probe
        case 32: return launch_pipe_batch<256, 64, 64, 32>(b, ncls, x1, bias, y, x_bytes, st, stats);  // 8 waves
        default: return cg_set_error(CG_ERR_ARG, "pipelined conv: unknown tile configuration %d", cfg);
    }
}

void fill_class(PipeClass& pc, const cg_conv_geom* g, const float* w) {
    pc.g = *g;
    pc.w = w;
    pc.M = g->N * g->Ho * g->Wo;
    pc.K = g->T * g->C1;
    pc.w_bytes = (unsigned)((size_t)g->Cout * pc.K * sizeof(float));
    pc.pad_ = 0;
    pc.xs = nullptr;
    pc.bias = nullptr;
    pc.y = nullptr;
    pc.x_scale = nullptr;
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

// tile configurations of the forward kernel (id -> BM, BN, WM, WN, STAGES)
int launch_fwd_cfg(int cfg, const cg_conv_geom* g, const float* x1, const float* x2, const float* w, const float* bias,
                   float* y, int M, int K, bool fast, hipStream_t st, double* stats = nullptr) {
    switch (cfg) {
        case 0: return launch_fwd<128, 128, 64, 64, 1>(g, x1, x2, w, bias, y, M, K, fast, st);
        case 1: return launch_fwd<128, 64, 64, 32, 1>(g, x1, x2, w, bias, y, M, K, fast, st);
        case 2: return launch_fwd<128, 32, 32, 32, 1>(g, x1, x2, w, bias, y, M, K, fast, st);
        case 3: return launch_fwd<64, 64, 32, 32, 1>(g, x1, x2, w, bias, y, M, K, fast, st);
        case 4: return launch_fwd<128, 128, 64, 64, 2>(g, x1, x2, w, bias, y, M, K, fast, st);
        case 5: return launch_fwd<128, 64, 64, 32, 2>(g, x1, x2, w, bias, y, M, K, fast, st);
        case 6: return launch_fwd<128, 128, 64, 32, 1>(g, x1, x2, w, bias, y, M, K, fast, st);   // 8 waves
        case 7: return launch_fwd<128, 128, 64, 32, 2>(g, x1, x2, w, bias, y, M, K, fast, st);   // 8 waves
        case 8: return launch_fwd<64, 128, 32, 64, 1>(g, x1, x2, w, bias, y, M, K, fast, st);
        case 9: return launch_fwd<64, 128, 32, 64, 2>(g, x1, x2, w, bias, y, M, K, fast, st);
        case 10: return launch_fwd<64, 64, 32, 32, 2>(g, x1, x2, w, bias, y, M, K, fast, st);
        case 11: return launch_fwd<256, 128, 64, 64, 1>(g, x1, x2, w, bias, y, M, K, fast, st);  // 8 waves
        case 12: return launch_fwd<256, 128, 64, 64, 2>(g, x1, x2, w, bias, y, M, K, fast, st);  // 8 waves
        case 13: return launch_fwd<128, 32, 32, 32, 2>(g, x1, x2, w, bias, y, M, K, fast, st);
        case 14: return launch_fwd<128, 64, 32, 32, 1>(g, x1, x2, w, bias, y, M, K, fast, st);   // 8 waves
        case 15: return launch_fwd<256, 64, 64, 32, 1>(g, x1, x2, w, bias, y, M, K, fast, st);   // 8 waves
        case 16: return launch_fwd<64, 128, 32, 32, 1>(g, x1, x2, w, bias, y, M, K, fast, st);   // 8 waves
        case 17: return launch_fwd<128, 128, 32, 32, 1>(g, x1, x2, w, bias, y, M, K, fast, st);  // 16 waves
        case 18: return launch_fwd<256, 128, 64, 32, 1>(g, x1, x2, w, bias, y, M, K, fast, st);  // 16 waves
        case 20: case 21: case 22: case 23: case 24: case 25: case 26: case 27: case 28: case 29: case 30: case 31:
        case 32: {
            if (!pipe_ok(g, K)) return cg_set_error(CG_ERR_ARG, "conv forward: configuration %d needs the pipelined path", cfg);
            PipeBatch b;
            fill_class(b.c[0], g, w);
            return launch_pipe_cfg(cfg, b, 1, x1, bias, y, (unsigned)((size_t)g->N * g->H * g->W * g->C1 * sizeof(float)), st,
                                   stats);
        }
        default: return cg_set_error(CG_ERR_ARG, "conv forward: unknown tile configuration %d", cfg);
    }
}

// Measured on MI355X (profiles/r01_conv_tiles.txt): 8-wave 128x128 blocks (two waves per SIMD hide each
// other's barrier / LDS phases) reach 112-123 TFLOP/s once >= ~192 such tiles exist; problems with
// fewer tiles fill the 256 CUs better with 64x64 tiles (two LDS stages when very few tiles).
int pick_fwd_cfg(const cg_conv_geom* g, int M, bool pipe) {
    const long blocks128 = (long)((M + 127) / 128) * ((g->Cout + 127) / 128);
    if (g->Cout > 64) {
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

bool cg_wgrad_force_legacy = false;  // A/B switch (cg_conv2d_wgrad_legacy)

struct WgradPlan {
    int bm, bn;
    bool fast;
    int tiles_m, tiles_n, splits, slices_per_split;
};

WgradPlan plan_wgrad(const cg_conv_geom* g) {
    // Every instantiated tile has exactly four 32x32 MFMA wave tiles or more (4 waves / block):
    //   bm=128: bn in {128, 64, 32};  bm=64: bn in {128, 64};  bm=32: bn = 128.
    // FAST (float4 gather, one tap per k-tile) needs a single source and Ct % bn == 0.
    WgradPlan p;
    const int Ct = g->C1 + g->C2;
    const int K = g->T * Ct;
    const int M = g->N * g->Ho * g->Wo;
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
    p.tiles_m = (g->Cout + p.bm - 1) / p.bm;
    p.tiles_n = (K + p.bn - 1) / p.bn;
    const int slices = (M + 31) / 32;
    const int tiles = p.tiles_m * p.tiles_n;
    int want = (2 * 256) / tiles;                      // two co-resident blocks per CU, and no partial second round
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
                 float* out, int M, int K, int want_bias, hipStream_t st) {
    dim3 grid(p.tiles_m * p.tiles_n, 1, p.splits), block((BM / WM) * (BN / WN) * 64);
    ProfScope prof(1, BM, BN, p.fast, 2.0 * (double)M * (double)g->Cout * (double)K, st, g);
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

bool wgrad_pipe_ok(const cg_conv_geom* g, const WgradPlan& p, int M) {
    if (!p.fast || g->C2 != 0 || (g->Cout & 3)) return false;
    if (ilog2_exact(g->Ho * g->Wo) < 0 || ilog2_exact(g->Wo) < 0) return false;
    if (!((p.bm == 128 && p.bn == 128) || (p.bm == 128 && p.bn == 64) || (p.bm == 64 && p.bn == 64) ||
          (p.bm == 64 && p.bn == 128)))
        return false;
    return (size_t)g->N * g->H * g->W * g->C1 * sizeof(float) < (size_t)CG_OOB &&
           (size_t)M * g->Cout * sizeof(float) < (size_t)CG_OOB;
}

template <int BM, int BN, int WM, int WN>
int launch_wgrad_pipe(const cg_conv_geom* g, const WgradPlan& p, const float* x1, const float* dz, float* out, int M, int K,
                      int want_bias, hipStream_t st) {
    dim3 grid(p.tiles_m * p.tiles_n, 1, p.splits), block((BM / WM) * (BN / WN) * 64);
    ProfScope prof(3, BM, BN, true, 2.0 * (double)M * (double)g->Cout * (double)K, st, g);
    hipLaunchKernelGGL((conv_wgrad_pipe_kernel<BM, BN, WM, WN>), grid, block, 0, st, *g, x1, dz, out, M, K, p.tiles_n,
                       p.slices_per_split, want_bias, ilog2_exact(g->Ho * g->Wo), ilog2_exact(g->Wo),
                       (unsigned)((size_t)g->N * g->H * g->W * g->C1 * sizeof(float)),
                       (unsigned)((size_t)M * g->Cout * sizeof(float)));
    CG_LAUNCH_CHECK("conv_wgrad_pipe_kernel");
    return CG_OK;
}

}  // namespace

static int conv2d_fwd_impl(const cg_conv_geom* g, const float* x1, const float* x2, const float* w, const float* bias,
                           float* y, int cfg, cg_stream_t stream) {
    int rc = validate_geom(g, "cg_conv2d_fwd");
    if (rc) return rc;
    CG_CHECK_ARG(x1 && w && y, "cg_conv2d_fwd: null pointer");
    CG_CHECK_ARG(g->C2 == 0 || x2, "cg_conv2d_fwd: C2 > 0 needs x2");
    const int Ct = g->C1 + g->C2;
    const int K = g->T * Ct;
    const int M = g->N * g->Ho * g->Wo;
    const bool fast = (g->C2 == 0) && (Ct % 32 == 0);
    if (cfg < 0) cfg = pick_fwd_cfg(g, M, fast && pipe_ok(g, K));
    return launch_fwd_cfg(cfg, g, x1, x2, w, bias, y, M, K, fast, cg_s(stream));
}

extern "C" int cg_conv2d_fwd(const cg_conv_geom* g, const float* x1, const float* x2, const float* w,
                             const float* bias, float* y, cg_stream_t stream) {
    return conv2d_fwd_impl(g, x1, x2, w, bias, y, -1, stream);
}

extern "C" int cg_conv2d_fwd_amax(const cg_conv_geom* g, const float* x1, const float* x2, const float* w,
                                  const float* bias, float* y, float* amax_state, int* amax_nslots, cg_stream_t stream) {
    CG_CHECK_ARG(amax_state && amax_nslots, "cg_conv2d_fwd_amax: null pointer");
    fwd_amax.state = amax_state;
    fwd_amax.nslots = 0;
    const int rc = conv2d_fwd_impl(g, x1, x2, w, bias, y, -1, stream);
    *amax_nslots = rc ? 0 : fwd_amax.nslots;
    fwd_amax.state = nullptr;
    return rc;
}

extern "C" int cg_conv2d_fwd_stats(const cg_conv_geom* g, const float* x1, const float* x2, const float* w,
                                   const float* bias, float* y, double* stats, size_t stats_bytes, int* rows_per_partial,
                                   cg_stream_t stream) {
    int rc = validate_geom(g, "cg_conv2d_fwd_stats");
    if (rc) return rc;
    CG_CHECK_ARG(x1 && w && y && rows_per_partial, "cg_conv2d_fwd_stats: null pointer");
    CG_CHECK_ARG(g->C2 == 0 || x2, "cg_conv2d_fwd_stats: C2 > 0 needs x2");
    const int Ct = g->C1 + g->C2;
    const int K = g->T * Ct;
    const int M = g->N * g->Ho * g->Wo;
    const bool fast = (g->C2 == 0) && (Ct % 32 == 0);
    const int cfg = pick_fwd_cfg(g, M, fast && pipe_ok(g, K));
    *rows_per_partial = 0;
    double* st_ptr = nullptr;
    if (pipe_cfg_has_stats(cfg) && stats && g->act == CG_ACT_NONE && g->osy == 1 && g->osx == 1) {
        const int bm = pipe_cfg_bm(cfg);
        if ((g->Ho * g->Wo) % bm == 0 && stats_bytes >= (size_t)(M / bm) * g->Cout * 2 * sizeof(double)) {
            st_ptr = stats;
            *rows_per_partial = bm;
        }
    }
    return launch_fwd_cfg(cfg, g, x1, x2, w, bias, y, M, K, fast, cg_s(stream), st_ptr);
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

extern "C" int cg_conv2d_fwd_x3(const cg_conv_geom* g, const void* xs, size_t x_lo_elems, const void* ws,
                                size_t w_lo_elems, float w_scale, const float* x_scale_dev, const float* bias, float* y,
                                void* y_split, size_t y_lo_elems, double* stats, size_t stats_bytes,
                                int* rows_per_partial, int tile_cfg, float* amax_state, int* amax_nslots,
                                cg_stream_t stream) {
    int rc = validate_geom(g, "cg_conv2d_fwd_x3");
    CG_CHECK_ARG((amax_state == nullptr) == (amax_nslots == nullptr), "cg_conv2d_fwd_x3: amax_state and amax_nslots go together");
    if (amax_nslots) *amax_nslots = 0;
    if (rc) return rc;
    CG_CHECK_ARG(xs && ws && y && w_scale > 0.f, "cg_conv2d_fwd_x3: null pointer / bad scale");
    CG_CHECK_ARG(!y_split || x3_lo_ok(y_lo_elems, (size_t)g->N * g->HoF * g->WoF * g->Cout * 2), "cg_conv2d_fwd_x3: bad y lo offset");
    const int K = g->T * g->C1;
    const int M = g->N * g->Ho * g->Wo;
    const size_t x_plane = (size_t)g->N * g->H * g->W * g->C1 * 2, w_plane = (size_t)g->Cout * K * 2;
    CG_CHECK_ARG(x3_lo_ok(x_lo_elems, x_plane) && x3_lo_ok(w_lo_elems, w_plane),
                 "cg_conv2d_fwd_x3: lo offset does not match the operand layout (CG_X3_LO_ELEMS)");
    const size_t x_span = x3_span(x_lo_elems, x_plane), w_span = x3_span(w_lo_elems, w_plane);
    CG_CHECK_ARG(g->C2 == 0 && g->C1 % BK == 0 && x_span < (size_t)CG_OOB && w_span < (size_t)CG_OOB,
                 "cg_conv2d_fwd_x3: needs one source with C %% 32 == 0 and operands spanning < 2 GiB");
    PipeBatch b;
    fill_class(b.c[0], g, (const float*)ws);
    b.c[0].w_bytes = (unsigned)(w_lo_elems * 2);
    b.c[0].pad_ = (int32_t)w_span;
    const int cfg = tile_cfg < 0 ? pick_x3_cfg(g->Cout, M, g->C1) : tile_cfg;
    CG_CHECK_ARG((cfg != 6 && cfg != 7) || g->C1 % 64 == 0, "cg_conv2d_fwd_x3: tile configuration %d needs C %% 64 == 0", cfg);
    const int bm = (cfg == 3 || cfg == 10) ? 64 : ((cfg == 5 || cfg == 13) ? 256 : 128);
    double* st_ptr = nullptr;
    if (rows_per_partial) {
        *rows_per_partial = 0;
        if (stats && g->act == CG_ACT_NONE && g->osy == 1 && g->osx == 1 && (g->Ho * g->Wo) % bm == 0 &&
            stats_bytes >= (size_t)(M / bm) * g->Cout * 2 * sizeof(double)) {
            st_ptr = stats;
            *rows_per_partial = bm;
        }
    }
    hipStream_t st = cg_s(stream);
    fwd_amax.state = amax_state;
    fwd_amax.nslots = 0;
    rc = launch_x3_cfg(cfg, b, 1, xs, bias, y, (unsigned)(x_lo_elems * 2), (unsigned)x_span, 1.0f / w_scale, x_scale_dev, st,
                       st_ptr, y_split, y_lo_elems);
    if (amax_nslots && !rc) *amax_nslots = fwd_amax.nslots;
    fwd_amax.state = nullptr;
    return rc;
}

// One launch for the same layer of up to four council members (same geometry, each with its own activations, weights,
// bias and output): blockIdx.y = member.  Four times the rows per launch -> the large tiles and full waves of CUs the
// single-member launches cannot use.  No instance-norm partials / split output / maxima yet.
extern "C" int cg_conv2d_fwd_x3_group(int n, const cg_conv_geom* g, const void* const* x_hi, size_t x_lo_elems,
                                      const void* const* w_hi, size_t w_lo_elems, float w_scale,
                                      const float* const* x_scale_dev, const float* const* bias, float* const* y, int tile_cfg,
                                      cg_stream_t stream) {
    int rc = validate_geom(g, "cg_conv2d_fwd_x3_group");
    if (rc) return rc;
    CG_CHECK_ARG(n >= 1 && n <= 4 && x_hi && w_hi && y && w_scale > 0.f, "cg_conv2d_fwd_x3_group: 1..4 members, non-null tables");
    const int K = g->T * g->C1;
    const int M = g->N * g->Ho * g->Wo;
    const size_t x_plane = (size_t)g->N * g->H * g->W * g->C1 * 2, w_plane = (size_t)g->Cout * K * 2;
    CG_CHECK_ARG(x3_lo_ok(x_lo_elems, x_plane) && x3_lo_ok(w_lo_elems, w_plane),
                 "cg_conv2d_fwd_x3_group: lo offset does not match the operand layout (CG_X3_LO_ELEMS)");
    const size_t x_span = x3_span(x_lo_elems, x_plane), w_span = x3_span(w_lo_elems, w_plane);
    CG_CHECK_ARG(g->C2 == 0 && g->C1 % BK == 0 && x_span < (size_t)CG_OOB && w_span < (size_t)CG_OOB && g->osy == 1 && g->osx == 1,
                 "cg_conv2d_fwd_x3_group: needs one source with C %% 32 == 0 and operands spanning < 2 GiB");
    PipeBatch b;
    for (int c = 0; c < n; ++c) {
        CG_CHECK_ARG(x_hi[c] && w_hi[c] && y[c], "cg_conv2d_fwd_x3_group: null pointer for member %d", c);
        fill_class(b.c[c], g, (const float*)w_hi[c]);
        b.c[c].w_bytes = (unsigned)(w_lo_elems * 2);
        b.c[c].pad_ = (int32_t)w_span;
        b.c[c].xs = x_hi[c];
        b.c[c].bias = bias ? bias[c] : nullptr;
        b.c[c].y = y[c];
        b.c[c].x_scale = x_scale_dev ? x_scale_dev[c] : nullptr;
    }
    const int cfg = tile_cfg < 0 ? pick_x3_cfg(g->Cout, (long)M * n, g->C1) : tile_cfg;
    CG_CHECK_ARG((cfg != 6 && cfg != 7) || g->C1 % 64 == 0, "cg_conv2d_fwd_x3_group: tile configuration %d needs C %% 64 == 0", cfg);
    return launch_x3_cfg(cfg, b, n, x_hi[0], nullptr, y[0], (unsigned)(x_lo_elems * 2), (unsigned)x_span, 1.0f / w_scale, nullptr,
                         cg_s(stream), nullptr, nullptr, 0);
}

extern "C" int cg_split_f16_dynamic(const float* x, void* out, size_t n, size_t lo_elems, float* state, int nslots,
                                    cg_stream_t stream) {
    CG_CHECK_ARG(x && out && state && n > 0 && x3_lo_ok(lo_elems, n * 2) && nslots >= 0 && nslots <= CG_AMAX_MAX_SLOTS,
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
                       nslots);
    CG_LAUNCH_CHECK("split_f16_dyn_kernel");
    return CG_OK;
}

extern "C" int cg_conv2d_fwd_tile(const cg_conv_geom* g, const float* x1, const float* x2, const float* w,
                                  const float* bias, float* y, int tile_cfg, cg_stream_t stream) {
    return conv2d_fwd_impl(g, x1, x2, w, bias, y, tile_cfg, stream);
}

extern "C" size_t cg_conv2d_wgrad_workspace(const cg_conv_geom* g) {
    if (!g || g->T < 1 || g->T > CG_MAX_TAPS) return 0;
    WgradPlan p = plan_wgrad(g);
    const size_t K = (size_t)g->T * (g->C1 + g->C2);
    return (size_t)p.splits * ((size_t)g->Cout * K + g->Cout) * sizeof(float);
}

extern "C" int cg_conv2d_wgrad(const cg_conv_geom* g, const float* x1, const float* x2, const float* dz, float* dw,
                               float* dbias, int accumulate, void* ws, size_t ws_bytes, cg_stream_t stream) {
    int rc = validate_geom(g, "cg_conv2d_wgrad");
    if (rc) return rc;
    CG_CHECK_ARG(x1 && dz && dw, "cg_conv2d_wgrad: null pointer");
    CG_CHECK_ARG(g->C2 == 0 || x2, "cg_conv2d_wgrad: C2 > 0 needs x2");
    const size_t need = cg_conv2d_wgrad_workspace(g);
    if (!ws || ws_bytes < need) return cg_set_error(CG_ERR_WORKSPACE, "cg_conv2d_wgrad: workspace %zu < %zu", ws_bytes, need);
    const int Ct = g->C1 + g->C2;
    const int K = g->T * Ct;
    const int M = g->N * g->Ho * g->Wo;
    hipStream_t st = cg_s(stream);
    WgradPlan p = plan_wgrad(g);
    float* part = (float*)ws;
    const int want_bias = dbias != nullptr;
#define WG(BM_, BN_, WM_, WN_) rc = launch_wgrad<BM_, BN_, WM_, WN_>(g, p, x1, x2, dz, part, M, K, want_bias, st)
#define WGP(BM_, BN_, WM_, WN_) rc = launch_wgrad_pipe<BM_, BN_, WM_, WN_>(g, p, x1, dz, part, M, K, want_bias, st)
    if (wgrad_pipe_ok(g, p, M) && !cg_wgrad_force_legacy) {
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
    else return cg_set_error(CG_ERR_ARG, "cg_conv2d_wgrad: no tile for %dx%d", p.bm, p.bn);
#undef WG
#undef WGP
    if (rc) return rc;
    const size_t nw = (size_t)g->Cout * K;
    const size_t n = nw + (want_bias ? g->Cout : 0);
    if (p.splits >= 128)
        hipLaunchKernelGGL(splitk_reduce_kernel<64>, dim3(cg_div_up(n * 64, 256)), dim3(256), 0, st, (const float*)part, dw,
                           dbias, nw, g->Cout, p.splits, accumulate);
    else if (p.splits >= 24)
        hipLaunchKernelGGL(splitk_reduce_kernel<8>, dim3(cg_div_up(n * 8, 256)), dim3(256), 0, st, (const float*)part, dw, dbias,
                           nw, g->Cout, p.splits, accumulate);
    else
        hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3(cg_div_up(n, 256)), dim3(256), 0, st, (const float*)part, dw, dbias,
                           nw, g->Cout, p.splits, accumulate);
    CG_LAUNCH_CHECK("splitk_reduce_kernel");
    return CG_OK;
}

// split-precision weight gradient (conv_x3.inc): x and dz arrive as {hi, lo} fp16 planes with their power-of-two
// scales (device-side pointers, NULL = 1); same planning, partial layout and deterministic reduce as cg_conv2d_wgrad
namespace {
template <int BM, int BN, int WM, int WN>
int launch_wgrad_x3(const cg_conv_geom* g, const WgradPlan& p, const void* xs, size_t x_lo, const float* x_scale,
                    const void* dzs, size_t dz_lo, const float* dz_scale, float* out, int M, int K, int want_bias,
                    hipStream_t st) {
    dim3 grid(p.tiles_m * p.tiles_n, 1, p.splits), block((BM / WM) * (BN / WN) * 64);
    const size_t x_plane = (size_t)g->N * g->H * g->W * g->C1 * 2, dz_plane = (size_t)M * g->Cout * 2;
    ProfScope prof(5, BM, BN, true, 2.0 * (double)M * (double)g->Cout * (double)K, st, g);
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
                     hipStream_t st) {
    dim3 grid(p.tiles_m * p.tiles_n, 1, p.splits), block((BM / WM) * (BN / WN) * 64);
    const size_t x_plane = (size_t)g->N * g->H * g->W * g->C1 * 2, dz_plane = (size_t)M * g->Cout * 2;
    ProfScope prof(5, BM, BN, true, 2.0 * (double)M * (double)g->Cout * (double)K, st, g);
    hipLaunchKernelGGL((conv_wgrad_x3t_kernel<BM, BN, WM, WN>), grid, block, 0, st, *g, xs, (unsigned)x3_span(x_lo, x_plane),
                       x_scale, dzs, (unsigned)x3_span(dz_lo, dz_plane), dz_scale, out, M, K, p.tiles_n, p.slices_per_split,
                       want_bias, ilog2_exact(g->Ho * g->Wo), ilog2_exact(g->Wo));
    CG_LAUNCH_CHECK("conv_wgrad_x3t_kernel");
    return CG_OK;
}
}  // namespace
#endif
// CG_WGRAD_X3_PERM=1 in the environment keeps the v_perm / ds_write_b32 loader (A/B against the transposing LDS read)
static bool wgrad_x3_use_tr() {
    static const bool on = CG_X3_INTERLEAVE && getenv("CG_WGRAD_X3_PERM") == nullptr;
    return on;
}

extern "C" int cg_conv2d_wgrad_x3_ok(const cg_conv_geom* g) {
    if (!g || g->T < 1 || g->T > CG_MAX_TAPS) return 0;
    const int M = g->N * g->Ho * g->Wo;
    WgradPlan p = plan_wgrad(g);
    return wgrad_pipe_ok(g, p, M) && (g->Cout & 31) == 0 && (g->C1 & 31) == 0;
}

extern "C" int cg_conv2d_wgrad_x3(const cg_conv_geom* g, const void* xs, size_t x_lo_elems, const float* x_scale_dev,
                                  const void* dzs, size_t dz_lo_elems, const float* dz_scale_dev, float* dw, float* dbias,
                                  int accumulate, void* ws, size_t ws_bytes, cg_stream_t stream) {
    int rc = validate_geom(g, "cg_conv2d_wgrad_x3");
    if (rc) return rc;
    CG_CHECK_ARG(xs && dzs && dw, "cg_conv2d_wgrad_x3: null pointer");
    CG_CHECK_ARG(cg_conv2d_wgrad_x3_ok(g), "cg_conv2d_wgrad_x3: layer does not qualify (see cg_conv2d_wgrad_x3_ok)");
    const size_t need = cg_conv2d_wgrad_workspace(g);
    if (!ws || ws_bytes < need) return cg_set_error(CG_ERR_WORKSPACE, "cg_conv2d_wgrad_x3: workspace %zu < %zu", ws_bytes, need);
    const int K = g->T * g->C1;
    const int M = g->N * g->Ho * g->Wo;
    const size_t x_plane = (size_t)g->N * g->H * g->W * g->C1 * 2, dz_plane = (size_t)M * g->Cout * 2;
    CG_CHECK_ARG(x3_lo_ok(x_lo_elems, x_plane) && x3_lo_ok(dz_lo_elems, dz_plane) && x3_span(x_lo_elems, x_plane) < (size_t)CG_OOB &&
                     x3_span(dz_lo_elems, dz_plane) < (size_t)CG_OOB && g->Cout % 32 == 0 && g->C1 % 32 == 0,
                 "cg_conv2d_wgrad_x3: operand planes out of range / lo offset does not match the layout");
    hipStream_t st = cg_s(stream);
    WgradPlan p = plan_wgrad(g);
    float* part = (float*)ws;
    const int want_bias = dbias != nullptr;
#if CG_X3_INTERLEAVE
#define WGX(BM_, BN_, WM_, WN_)                                                                                                  \
    rc = wgrad_x3_use_tr()                                                                                                       \
             ? launch_wgrad_x3t<BM_, BN_, WM_, WN_>(g, p, xs, x_lo_elems, x_scale_dev, dzs, dz_lo_elems, dz_scale_dev, part, M, K, want_bias, st) \
             : launch_wgrad_x3<BM_, BN_, WM_, WN_>(g, p, xs, x_lo_elems, x_scale_dev, dzs, dz_lo_elems, dz_scale_dev, part, M, K, want_bias, st)
#else
#define WGX(BM_, BN_, WM_, WN_) \
    rc = launch_wgrad_x3<BM_, BN_, WM_, WN_>(g, p, xs, x_lo_elems, x_scale_dev, dzs, dz_lo_elems, dz_scale_dev, part, M, K, want_bias, st)
#endif
    if (p.bm == 128 && p.bn == 128) WGX(128, 128, 64, 32);   // 8 waves
    else if (p.bm == 128 && p.bn == 64) WGX(128, 64, 64, 32);
    else if (p.bm == 64 && p.bn == 64) WGX(64, 64, 32, 32);
    else WGX(64, 128, 32, 64);
#undef WGX
    if (rc) return rc;
    const size_t nw = (size_t)g->Cout * K;
    const size_t n = nw + (want_bias ? g->Cout : 0);
    if (p.splits >= 128)
        hipLaunchKernelGGL(splitk_reduce_kernel<64>, dim3(cg_div_up(n * 64, 256)), dim3(256), 0, st, (const float*)part, dw,
                           dbias, nw, g->Cout, p.splits, accumulate);
    else if (p.splits >= 24)
        hipLaunchKernelGGL(splitk_reduce_kernel<8>, dim3(cg_div_up(n * 8, 256)), dim3(256), 0, st, (const float*)part, dw, dbias,
                           nw, g->Cout, p.splits, accumulate);
    else
        hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3(cg_div_up(n, 256)), dim3(256), 0, st, (const float*)part, dw, dbias,
                           nw, g->Cout, p.splits, accumulate);
    CG_LAUNCH_CHECK("splitk_reduce_kernel");
    return CG_OK;
}

extern "C" int cg_conv2d_wgrad_legacy(int on) {
    cg_wgrad_force_legacy = on != 0;
    return CG_OK;
}

static int launch_transpose(const float* w, float* out, int Cout, int T, int Cin, int ci0, int nci, const TransTable& tt,
                            int nz, hipStream_t st, bool split = false, float scale = 1.f, unsigned lo_elems = 0) {
    dim3 grid(cg_div_up(nci, 32), cg_div_up(Cout, 32), nz);
    if (split)
        hipLaunchKernelGGL(weight_transpose_kernel<true>, grid, dim3(256), 0, st, w, out, Cout, T, Cin, ci0, nci, tt, scale,
                           lo_elems);
    else
        hipLaunchKernelGGL(weight_transpose_kernel<false>, grid, dim3(256), 0, st, w, out, Cout, T, Cin, ci0, nci, tt, 1.f, 0u);
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
    return launch_transpose(w, out, Cout, T, Cin, ci0, nci, tt, Tc, cg_s(stream));
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
    size_t ws_floats;
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
}  // namespace

extern "C" size_t cg_conv2d_dgrad_workspace(const cg_conv_geom* g, int nci) {
    if (!g || g->T < 1 || g->T > CG_MAX_TAPS || nci < 1) return 0;
    return (size_t)g->T * nci * g->Cout * sizeof(float);   // every tap belongs to exactly one class
}

extern "C" int cg_conv2d_dgrad(const cg_conv_geom* g, const float* dz, const float* w, int ci0, int nci, float* dx,
                               void* ws, size_t ws_bytes, cg_stream_t stream) {
    int rc = validate_geom(g, "cg_conv2d_dgrad");
    if (rc) return rc;
    CG_CHECK_ARG(dz && w && dx, "cg_conv2d_dgrad: null pointer");
    const int Cin = g->C1 + g->C2;
    CG_CHECK_ARG(ci0 >= 0 && nci >= 1 && ci0 + nci <= Cin, "cg_conv2d_dgrad: channel range [%d, %d) outside %d", ci0, ci0 + nci, Cin);
    if (!ws || ws_bytes < cg_conv2d_dgrad_workspace(g, nci))
        return cg_set_error(CG_ERR_WORKSPACE, "cg_conv2d_dgrad: workspace too small");
    static thread_local DgradPlan p;
    rc = plan_dgrad(g, nci, p);
    if (rc) return rc;
    hipStream_t st = cg_s(stream);
    float* wt = (float*)ws;
    // one launch re-lays-out the weights of every class: wt_c[ci][tc][co]
    TransTable tt;
    memset(&tt, 0, sizeof(tt));
    int nz = 0;
    for (int c = 0; c < p.ncls; ++c)
        for (int tc = 0; tc < p.cg[c].T; ++tc, ++nz) {
            tt.src_tap[nz] = p.tap_src[c][tc];
            tt.dst_base[nz] = (int32_t)p.w_off[c];
            tt.dst_tc[nz] = tc;
            tt.dst_T[nz] = p.cg[c].T;
        }
    CG_CHECK_ARG(p.ws_floats < (size_t)0x7fffffff, "cg_conv2d_dgrad: weight tensor too large");
    rc = launch_transpose(w, wt, g->Cout, g->T, Cin, ci0, nci, tt, nz, st);
    if (rc) return rc;
    // batched pipelined launch when every class qualifies
    bool pipe = p.ncls <= 4;
    long m_total = 0;
    for (int c = 0; c < p.ncls && pipe; ++c) {
        pipe = pipe_ok(&p.cg[c], p.cg[c].T * p.cg[c].C1);
        m_total += (long)p.cg[c].N * p.cg[c].Ho * p.cg[c].Wo;
    }
    if (pipe) {
        PipeBatch b;
        for (int c = 0; c < p.ncls; ++c) fill_class(b.c[c], &p.cg[c], wt + p.w_off[c]);
        const int cfg = pick_fwd_cfg(&p.cg[0], (int)(m_total > 0x7fffffffL ? 0x7fffffffL : m_total), true);
        if (cfg >= 20)
            return launch_pipe_cfg(cfg, b, p.ncls, dz, nullptr, dx,
                                   (unsigned)((size_t)g->N * g->Ho * g->Wo * g->Cout * sizeof(float)), st);
    }
    for (int c = 0; c < p.ncls; ++c) {
        rc = cg_conv2d_fwd(&p.cg[c], dz, nullptr, wt + p.w_off[c], nullptr, dx, stream);
        if (rc) return rc;
    }
    return CG_OK;
}

extern "C" int cg_act_bwd_split(const float* dy, const float* y, size_t n, int act, void* out, size_t lo_elems,
                                float* state, float* dz, cg_stream_t stream) {
    CG_CHECK_ARG(dy && y && out && state && n > 0 && x3_lo_ok(lo_elems, n * 2), "cg_act_bwd_split: bad args");
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipStream_t st = cg_s(stream);
    const int nslots = amax_blocks(n);
    hipLaunchKernelGGL(act_bwd_amax_kernel, dim3(nslots), dim3(256), 0, st, dy, y, dz, n, act, state);
    CG_LAUNCH_CHECK("act_bwd_amax_kernel");
    hipLaunchKernelGGL(act_bwd_split_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dy, y, (_Float16*)out, n, lo_elems, act,
                       state, nslots);
    CG_LAUNCH_CHECK("act_bwd_split_kernel");
    return CG_OK;
}

// split-precision data-gradient: dz arrives as {hi, lo} fp16 planes with its device-side scale (cg_split_f16_dynamic);
// the weights are re-laid-out AND split (scaled by CG_X3_WSCALE) by one launch into `ws`
extern "C" int cg_conv2d_dgrad_x3(const cg_conv_geom* g, const void* dz_split, size_t dz_lo_elems,
                                  const float* dz_scale_dev, const float* w, int ci0, int nci, float* dx, void* ws,
                                  size_t ws_bytes, cg_stream_t stream) {
    int rc = validate_geom(g, "cg_conv2d_dgrad_x3");
    if (rc) return rc;
    CG_CHECK_ARG(dz_split && w && dx && dz_scale_dev, "cg_conv2d_dgrad_x3: null pointer");
    const int Cin = g->C1 + g->C2;
    CG_CHECK_ARG(ci0 >= 0 && nci >= 1 && ci0 + nci <= Cin, "cg_conv2d_dgrad_x3: channel range outside %d", Cin);
    CG_CHECK_ARG(g->Cout % BK == 0, "cg_conv2d_dgrad_x3: Cout %% 32 != 0");
    if (!ws || ws_bytes < cg_conv2d_dgrad_workspace(g, nci))
        return cg_set_error(CG_ERR_WORKSPACE, "cg_conv2d_dgrad_x3: workspace too small");
    static thread_local DgradPlan p;
    rc = plan_dgrad(g, nci, p);
    if (rc) return rc;
    CG_CHECK_ARG(p.ncls <= 4, "cg_conv2d_dgrad_x3: stride %d has more than 4 output classes", g->stride);
    hipStream_t st = cg_s(stream);
    TransTable tt;
    memset(&tt, 0, sizeof(tt));
    int nz = 0;
    for (int c = 0; c < p.ncls; ++c)
        for (int tc = 0; tc < p.cg[c].T; ++tc, ++nz) {
            tt.src_tap[nz] = p.tap_src[c][tc];
            tt.dst_base[nz] = (int32_t)p.w_off[c];
            tt.dst_tc[nz] = tc;
            tt.dst_T[nz] = p.cg[c].T;
        }
    const size_t wt_elems = p.ws_floats;          // one fp16 plane = as many elements as the fp32 layout had floats
    const size_t dz_plane = (size_t)g->N * g->Ho * g->Wo * g->Cout * 2;
    CG_CHECK_ARG(x3_lo_ok(dz_lo_elems, dz_plane) && x3_span(dz_lo_elems, dz_plane) < (size_t)CG_OOB && 4 * wt_elems < (size_t)CG_OOB,
                 "cg_conv2d_dgrad_x3: operand planes out of range / lo offset does not match the layout");
    const size_t wt_lo = CG_X3_INTERLEAVE ? (size_t)CG_X3_LO_ELEMS : wt_elems;      // lo offset of the re-laid-out weights
    rc = launch_transpose(w, (float*)ws, g->Cout, g->T, Cin, ci0, nci, tt, nz, st, true, CG_X3_WSCALE, (unsigned)wt_lo);
    if (rc) return rc;
    PipeBatch b;
    long m_total = 0;
    for (int c = 0; c < p.ncls; ++c) {
        fill_class(b.c[c], &p.cg[c], (const float*)((const _Float16*)ws + cg_il(p.w_off[c])));     // class sizes: multiples of 32
        b.c[c].w_bytes = (unsigned)(wt_lo * 2);                                                     // lo offset
        b.c[c].pad_ = (int32_t)(4 * wt_elems - 2 * cg_il(p.w_off[c]));                             // span from this class's base
        m_total += b.c[c].M;
    }
    return launch_x3_cfg(pick_x3_cfg(nci, m_total, g->Cout), b, p.ncls, dz_split, nullptr, dx, (unsigned)(dz_lo_elems * 2),
                         (unsigned)x3_span(dz_lo_elems, dz_plane), 1.0f / CG_X3_WSCALE, dz_scale_dev, st, nullptr, nullptr, 0);
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

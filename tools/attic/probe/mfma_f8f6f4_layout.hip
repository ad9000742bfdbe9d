// Operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 (gfx950) for fp8 e4m3 and fp6 e2m3 operands, checked against a CPU
// matmul with small integer matrices (A asymmetric to B).  Hypothesis under test:
//   A: lane l holds row i = l & 31, contraction indices k = 32 * (l >> 5) + j, j = 0..31, element j in byte j of the lane's
//      8 dwords (fp8) / in bits [6j, 6j + 6) of its first 6 dwords (fp6);  B: the same with column n = l & 31;
//   C: col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), r = 0..15;  scales: e8m0 byte 127 = x1.
// Build + run: hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_f8f6f4_layout.hip -o /tmp/f8 && /tmp/f8
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef int i8v __attribute__((ext_vector_type(8)));

template <int FMT>
__global__ void probe(const int* a_regs, const int* b_regs, float* c_out, int scale_a, int scale_b) {
    i8v a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = a_regs[threadIdx.x * 8 + i];
        b[i] = b_regs[threadIdx.x * 8 + i];
    }
    f16v c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, FMT, FMT, 0, scale_a, 0, scale_b);
    for (int i = 0; i < 16; ++i) c_out[threadIdx.x * 16 + i] = c[i];
}

static uint8_t enc8(int v) {   // e4m3fn of 0..4
    static const uint8_t t[5] = {0x00, 0x38, 0x40, 0x44, 0x48};
    return t[v];
}
static uint8_t enc6(int v) {   // e2m3 of 0..4
    static const uint8_t t[5] = {0, 8, 16, 20, 24};
    return t[v];
}

static void pack(int fmt, const int M[32][64], int regs[64 * 8]) {
    memset(regs, 0, sizeof(int) * 64 * 8);
    for (int l = 0; l < 64; ++l) {
        uint8_t bytes[32];
        memset(bytes, 0, sizeof(bytes));
        for (int j = 0; j < 32; ++j) {
            const int v = M[l & 31][32 * (l >> 5) + j];
            if (fmt == 0) {
                bytes[j] = enc8(v);
            } else {
                const unsigned code = enc6(v), bit = 6 * j;
                bytes[bit >> 3] |= (uint8_t)(code << (bit & 7));
                if ((bit & 7) > 2) bytes[(bit >> 3) + 1] |= (uint8_t)(code >> (8 - (bit & 7)));
            }
        }
        memcpy(&regs[l * 8], bytes, 32);
    }
}

int main() {
    static int A[32][64], Bt[32][64];          // Bt[n][k] = B[k][n]
    for (int i = 0; i < 32; ++i)
        for (int k = 0; k < 64; ++k) {
            A[i][k] = (i * 7 + k * 3 + (i * k) % 5) % 5;
            Bt[i][k] = (i * 5 + k * 11 + (i + 2 * k) % 3) % 4;
        }
    int *d_a, *d_b;
    float* d_c;
    hipMalloc(&d_a, 64 * 8 * 4);
    hipMalloc(&d_b, 64 * 8 * 4);
    hipMalloc(&d_c, 64 * 16 * 4);
    for (int fmt = 0; fmt <= 2; fmt += 2) {
        static int ra[64 * 8], rb[64 * 8];
        pack(fmt, A, ra);
        pack(fmt, Bt, rb);
        hipMemcpy(d_a, ra, sizeof(ra), hipMemcpyHostToDevice);
        hipMemcpy(d_b, rb, sizeof(rb), hipMemcpyHostToDevice);
        for (int sc = 0; sc < 2; ++sc) {
            const int sa = sc ? 0x80808080 : 0x7f7f7f7f;        // second run: A scaled by 2^1 in every byte
            if (fmt == 0)
                hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, d_a, d_b, d_c, sa, 0x7f7f7f7f);
            else
                hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, d_a, d_b, d_c, sa, 0x7f7f7f7f);
            static float c[64 * 16];
            hipMemcpy(c, d_c, sizeof(c), hipMemcpyDeviceToHost);
            int bad = 0;
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 16; ++r) {
                    const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                    long ref = 0;
                    for (int k = 0; k < 64; ++k) ref += (long)A[row][k] * Bt[col][k];
                    if (c[l * 16 + r] != (float)(ref * (sc ? 2 : 1))) {
                        if (bad < 4) printf("   lane %d reg %d: got %g want %ld\n", l, r, c[l * 16 + r], ref * (sc ? 2 : 1));
                        ++bad;
                    }
                }
            printf("%s, A scale byte %s: %d of 1024 results differ from the hypothesis\n", fmt == 0 ? "fp8 e4m3" : "fp6 e2m3",
                   sc ? "128 (x2)" : "127 (x1)", bad);
        }
    }
    return 0;
}

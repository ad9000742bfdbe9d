// What does ds_read_b64_tr_b16 deliver?  LDS halfword h holds the value h; every lane passes a byte address and gets
// four halfwords back.  Run: hipcc --offload-arch=gfx950 -O2 tools/probe/tr_probe.hip -o gpurun_out/tr_probe && ./tr_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void probe(const int* lane_addr, uint16_t* out) {
    __shared__ uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) uint16_t*)lds;
    const unsigned a = base + (unsigned)lane_addr[threadIdx.x];
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = (uint16_t)(v.x & 0xffff);
    out[threadIdx.x * 4 + 1] = (uint16_t)(v.x >> 16);
    out[threadIdx.x * 4 + 2] = (uint16_t)(v.y & 0xffff);
    out[threadIdx.x * 4 + 3] = (uint16_t)(v.y >> 16);
}

static void run(const char* name, int (*f)(int)) {
    int h_addr[64];
    for (int l = 0; l < 64; ++l) h_addr[l] = f(l);
    int* d_addr;
    uint16_t* d_out;
    uint16_t h_out[256];
    hipMalloc(&d_addr, sizeof(h_addr));
    hipMalloc(&d_out, sizeof(h_out));
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("== %s (lane: byte address -> 4 halfword indices)\n", name);
    for (int l = 0; l < 64; ++l)
        printf("%2d: %4d -> %4d %4d %4d %4d%s", l, h_addr[l], h_out[4 * l], h_out[4 * l + 1], h_out[4 * l + 2], h_out[4 * l + 3],
               (l & 3) == 3 ? "\n" : "   |  ");
    hipFree(d_addr);
    hipFree(d_out);
}

int main() {
    run("addr = lane*8 (64 consecutive 8-byte items)", [](int l) { return l * 8; });
    run("addr = (lane%16)*64 + (lane/16)*8  (row = lane%16, 64-byte rows; 8-byte column block = lane/16)", [](int l) { return (l % 16) * 64 + (l / 16) * 8; });
    run("addr = (lane%16)*8 + (lane/16)*512", [](int l) { return (l % 16) * 8 + (l / 16) * 512; });
    run("addr = 0 for all lanes", [](int) { return 0; });
    run("addr = (lane%4)*128 + (lane/4%4)*8 + (lane/16)*32", [](int l) { return (l % 4) * 128 + (l / 4 % 4) * 8 + (l / 16) * 32; });
    return 0;
}

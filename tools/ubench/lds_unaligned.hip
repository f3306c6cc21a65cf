// lds_unaligned.hip -- do ds_read_b128 / b64 / b32 work at any byte address on gfx950 (ROCm 7.2), and what do they cost?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(uint32_t* out, int off) {
    __shared__ __align__(16) uint8_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const uint32_t a = (uint32_t)(uintptr_t)lds + off + 147u * (threadIdx.x & 15) + 16u * (threadIdx.x >> 4);   // the stage-B operand pattern
    u32x4 v; uint32_t w; uint64_t d;
    asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a));
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(a));
    asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(a));
    out[threadIdx.x * 8 + 0] = v.x; out[threadIdx.x * 8 + 1] = v.y; out[threadIdx.x * 8 + 2] = v.z; out[threadIdx.x * 8 + 3] = v.w;
    out[threadIdx.x * 8 + 4] = w; out[threadIdx.x * 8 + 5] = (uint32_t)d; out[threadIdx.x * 8 + 6] = (uint32_t)(d >> 32);
}
template <int MODE>
__global__ void rate(unsigned long long* out, int off, int n) {
    __shared__ __align__(16) uint8_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (uint8_t)i;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t a = (uint32_t)(uintptr_t)lds + (MODE == 0 ? 16u * lane : off + 147u * (lane & 15) + 16u * (lane >> 4));
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        u32x4 v0, v1, v2, v3;
        asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:64\n ds_read_b128 %2, %4 offset:128\n ds_read_b128 %3, %4 offset:192\n s_waitcnt lgkmcnt(0)"
                     : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(a));
        acc += v0 ^ v1 ^ v2 ^ v3;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (acc.x == 0x12345) out[1] = acc.y;
}
int main() {
    uint32_t* d; hipMalloc(&d, 64 * 8 * 4);
    int bad = 0;
    for (int off = 0; off < 4; ++off) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, off);
        uint32_t h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int t = 0; t < 64; ++t) {
            const int a = off + 147 * (t & 15) + 16 * (t >> 4);
            auto byte = [&](int i) { return (uint32_t)(uint8_t)((a + i) * 7 + 3); };
            auto dw = [&](int k) { return byte(4 * k) | byte(4 * k + 1) << 8 | byte(4 * k + 2) << 16 | byte(4 * k + 3) << 24; };
            for (int k = 0; k < 4; ++k) if (h[t * 8 + k] != dw(k)) { if (bad < 5) printf("b128 off %d lane %d dword %d: got %08x want %08x\n", off, t, k, h[t * 8 + k], dw(k)); ++bad; }
            if (h[t * 8 + 4] != dw(0)) { if (bad < 5) printf("b32 off %d lane %d: got %08x want %08x\n", off, t, h[t * 8 + 4], dw(0)); ++bad; }
            if (h[t * 8 + 5] != dw(0) || h[t * 8 + 6] != dw(1)) { if (bad < 5) printf("b64 off %d lane %d wrong\n", off, t); ++bad; }
        }
    }
    printf("unaligned ds_read_b32/b64/b128 at byte addresses off + 147 c + 16 g: %d mismatches\n", bad);
    unsigned long long* o; hipMalloc(&o, 64);
    const int n = 20000;
    for (int waves = 1; waves <= 8; waves *= 2) {
        unsigned long long h0, h1;
        hipLaunchKernelGGL(rate<0>, dim3(256), dim3(64 * waves), 0, 0, o, 0, n); hipLaunchKernelGGL(rate<0>, dim3(256), dim3(64 * waves), 0, 0, o, 0, n);
        hipMemcpy(&h0, o, 8, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(rate<1>, dim3(256), dim3(64 * waves), 0, 0, o, 1, n); hipLaunchKernelGGL(rate<1>, dim3(256), dim3(64 * waves), 0, 0, o, 1, n);
        hipMemcpy(&h1, o, 8, hipMemcpyDeviceToHost);
        printf("waves/CU %d: ticks per ds_read_b128 per wave: aligned 16-byte slots %.1f, the unaligned stage-B pattern %.1f\n", waves, (double)h0 / (4.0 * n), (double)h1 / (4.0 * n));
    }
    return 0;
}

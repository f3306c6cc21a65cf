// micro-benchmark: how long a wave's burst of LDS stores takes to drain (s_waitcnt lgkmcnt(0)), 12 waves per CU
//   mode 0: 12 x ds_write_b32, lanes 16 bytes apart (+ one pad dword per 16: the FIR kernel's staging pattern)
//   mode 1: 3 x ds_write_b128, lanes 16 bytes apart, linear
//   mode 2: 12 x ds_write_b32, lanes 4 bytes apart (conflict-free)
//   mode 3: 6 x ds_write_b64, lanes 16 bytes apart + pad (8-byte aligned variant)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(768) void k(unsigned* out, int iters, int mode, unsigned long long* cyc) {
    extern __shared__ __align__(16) unsigned char smem[];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* wb = smem + wave * 4096;
    unsigned a0[3];
    for (int i = 0; i < 3; ++i) { const unsigned L = 4u * (lane + 64u * i); a0[i] = 4u * (L + (L >> 4)); }
    u32x4 v[3] = {{lane, 1, 2, 3}, {lane, 5, 6, 7}, {lane, 9, 10, 11}};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (mode == 0) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                unsigned* d = (unsigned*)(wb + a0[i]);
                d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
            }
        } else if (mode == 1) {
#pragma unroll
            for (int i = 0; i < 3; ++i) *(u32x4*)(wb + 16u * (lane + 64u * i)) = v[i];
        } else if (mode == 2) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                unsigned* d = (unsigned*)(wb + 4u * lane + 1024u * i);
                d[0] = v[i].x; d[64] = v[i].y; d[128] = v[i].z; d[192] = v[i].w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const unsigned L = 4u * (lane + 64u * i);
                u32x2* d = (u32x2*)(wb + 4u * (L + 2u * (L >> 4)));
                d[0] = u32x2{v[i].x, v[i].y}; d[1] = u32x2{v[i].z, v[i].w};
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
        asm volatile("" ::: "memory");
        v[0].x += 1; v[1].y ^= v[0].x; v[2].z += v[1].y;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) atomicAdd(cyc, t1 - t0);
    out[blockIdx.x * blockDim.x + threadIdx.x] = ((unsigned*)wb)[lane] + v[0].x;
}
int main() {
    unsigned* out; unsigned long long* cyc; (void)hipMalloc(&out, 1 << 24); (void)hipMalloc(&cyc, 8);
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int waves = 1; waves <= 12; waves += (waves == 1 ? 3 : 8))
    for (int mode = 0; mode < 4; ++mode) {
        (void)hipMemset(cyc, 0, 8);
        const int iters = 2000;
        hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 49152, 0, out, iters, mode, cyc);
        (void)hipDeviceSynchronize();
        unsigned long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("waves/CU %2d mode %d: %.1f ticks per burst per wave\n", waves, mode, (double)h / (256.0 * waves) / iters);
    }
    return 0;
}

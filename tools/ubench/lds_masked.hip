// lds_masked.hip -- LDS throughput of ds_read_b128 with all 64 lanes against 16 active lanes (one kgroup of a wave), eight waves per CU:
// does the LDS pipe skip the inactive lanes' passes?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void thr(unsigned long long* out, int n, int cls) {
    extern __shared__ __align__(16) uint8_t lds[];
    for (int i = threadIdx.x; i < 65536; i += blockDim.x) lds[i] = (uint8_t)i;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t a = (uint32_t)(uintptr_t)lds + wave * 8192u + (lane & 15) * 272u + 16u * (lane >> 4);
    u32x4 acc = {0, 0, 0, 0};
    u32x4 v[8];
    for (int k = 0; k < 8; ++k) v[k] = u32x4{0, 0, 0, 0};
    const bool on = MODE == 0 || (int)(lane >> 4) == cls;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        if (on) {
            asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:64\n ds_read_b128 %2, %8 offset:128\n ds_read_b128 %3, %8 offset:192\n"
                         "ds_read_b128 %4, %8 offset:4352\n ds_read_b128 %5, %8 offset:4416\n ds_read_b128 %6, %8 offset:4480\n ds_read_b128 %7, %8 offset:4544\n s_waitcnt lgkmcnt(0)"
                         : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]) : "v"(a));
        }
        for (int k = 0; k < 8; ++k) acc ^= v[k];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (acc.x == 0x12345) out[1] = acc.y;
}
int main() {
    unsigned long long* o; hipMalloc(&o, 64);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&thr<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&thr<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int n = 20000;
    for (int waves : {1, 2, 4, 8}) {
        unsigned long long h0, h1;
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(thr<0>, dim3(256), dim3(64 * waves), 65536, 0, o, n, 0);
        hipMemcpy(&h0, o, 8, hipMemcpyDeviceToHost);
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(thr<1>, dim3(256), dim3(64 * waves), 65536, 0, o, n, 2);
        hipMemcpy(&h1, o, 8, hipMemcpyDeviceToHost);
        printf("%d waves per CU: ticks per ds_read_b128 per wave: all lanes %.2f | 16 lanes %.2f\n", waves, (double)h0 / (8.0 * n), (double)h1 / (8.0 * n));
    }
    return 0;
}

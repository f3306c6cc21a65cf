// smfmac_rate.hip -- cycles per instruction of v_mfma_i32_32x32x32_i8 and v_smfmac_i32_32x32x64_i8, one wave per SIMD, back to back
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int KIND>
__global__ void k(unsigned long long* out, int* sink, int n) {
    v4i A = {(int)threadIdx.x, 2, 3, 4}; v8i B = {1, 2, 3, 4, 5, 6, 7, (int)threadIdx.x};
    v4i B4 = {1, 2, 3, (int)threadIdx.x};
    v16i C0 = {0}, C1 = {0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        if (KIND == 0) { C0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B4, C0, 0, 0, 0); C1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B4, C1, 0, 0, 0); }
        else { C0 = __builtin_amdgcn_smfmac_i32_32x32x64_i8(A, B, C0, 0x1B1B1B1B, 0, 0); C1 = __builtin_amdgcn_smfmac_i32_32x32x64_i8(A, B, C1, 0x1B1B1B1B, 0, 0); }
    }
    asm volatile("" :: "v"(C0), "v"(C1));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[KIND] = t1 - t0;
    if (C0[0] == 12345 && C1[3] == 7) sink[0] = 1;
}
int main() {
    unsigned long long* d; int* s; hipMalloc(&d, 64); hipMalloc(&s, 64);
    const int n = 20000;
    for (int waves = 1; waves <= 4; ++waves) {
        hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
        hipLaunchKernelGGL(k<0>, dim3(256), dim3(256 * waves), 0, 0, d, s, n);   // warm
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<0>, dim3(256), dim3(256 * waves), 0, 0, d, s, n);
        hipEventRecord(e1, 0);
        hipLaunchKernelGGL(k<1>, dim3(256), dim3(256 * waves), 0, 0, d, s, n);
        hipEventRecord(e2, 0);
        hipDeviceSynchronize();
        float ms0 = 0, ms1 = 0; hipEventElapsedTime(&ms0, e0, e1); hipEventElapsedTime(&ms1, e1, e2);
        printf("waves/SIMD %d: wall dense %.3f ms sparse %.3f ms -> %.2f / %.2f ns per MFMA per SIMD\n", waves, ms0, ms1, ms0 * 1e6 / (2.0 * n * waves), ms1 * 1e6 / (2.0 * n * waves));
        unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("waves/SIMD %d: dense 32x32x32 i8: %.1f ticks per MFMA per wave; sparse 32x32x64 i8: %.1f ticks per MFMA per wave\n", waves, (double)h[0] / (2.0 * n), (double)h[1] / (2.0 * n));
    }
    return 0;
}

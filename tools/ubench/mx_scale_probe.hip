// mx_scale_probe.hip -- what the scale operands of v_mfma_scale_f32_32x32x64_f8f6f4 do (fp6 e2m3 x fp4 e2m1):
// A = all 1.0, B = all 1.0 -> 64 per element; then with scale bytes e_a / e_b in every byte of the scale registers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
template <int OA, int OB>
__global__ void k(float* out, int sa, int sb) {
    // e2m3 1.0 = 0b001000; 32 of them = 192 bits: pattern of 6-bit fields 001000 -> bytes repeat every 3: 0x08, 0x82, 0x20
    v8i A = {(int)0x08208208u, (int)0x82082082u, (int)0x20820820u, (int)0x08208208u, (int)0x82082082u, (int)0x20820820u, 0, 0};
    v8i B = {0x22222222, 0x22222222, 0x22222222, 0x22222222, 0, 0, 0, 0};     // e2m1 1.0 = 0b0010
    v16f C = {0};
    C = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, C, 2, 4, OA, sa, OB, sb);
    if (threadIdx.x == 0) out[0] = C[0];
    if (threadIdx.x == 37) out[1] = C[7];
}
int main() {
    float* d; hipMalloc(&d, 64);
    auto run = [&](int ea, int eb) {
        const int sa = ea * 0x01010101, sb = eb * 0x01010101;
        float h[2];
        hipLaunchKernelGGL((k<0, 0>), dim3(1), dim3(64), 0, 0, d, sa, sb); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("scale bytes a=%d b=%d: D = %g, %g (64 = unscaled)\n", ea, eb, h[0], h[1]);
    };
    run(127, 127); run(127, 130); run(130, 127); run(124, 130); run(0, 0);
    // which byte does opsel pick?  bytes 127,128,129,130 from low to high
    float h[2];
    hipLaunchKernelGGL((k<0, 0>), dim3(1), dim3(64), 0, 0, d, 0x7f7f7f7f, (int)0x8281807fu); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost); printf("opsel_b 0: %g\n", h[0]);
    hipLaunchKernelGGL((k<0, 1>), dim3(1), dim3(64), 0, 0, d, 0x7f7f7f7f, (int)0x8281807fu); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost); printf("opsel_b 1: %g\n", h[0]);
    hipLaunchKernelGGL((k<0, 2>), dim3(1), dim3(64), 0, 0, d, 0x7f7f7f7f, (int)0x8281807fu); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost); printf("opsel_b 2: %g\n", h[0]);
    hipLaunchKernelGGL((k<0, 3>), dim3(1), dim3(64), 0, 0, d, 0x7f7f7f7f, (int)0x8281807fu); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost); printf("opsel_b 3: %g\n", h[0]);
    return 0;
}

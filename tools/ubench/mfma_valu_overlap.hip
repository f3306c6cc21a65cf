// micro-benchmark: can a wave's VALU stream overlap another wave's int8 MFMA stream on the same SIMD?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
// mode bit0: MFMA waves active, bit1: VALU waves active.  8 waves per block: waves 0-3 = MFMA role, 4-7 = VALU role
__global__ __launch_bounds__(512) void k(unsigned* out, int iters, int mode, int same_wave) {
    const unsigned wave = threadIdx.x >> 6;
    unsigned a = threadIdx.x, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, e = a * 5, f = a | 1, g = a + 11, h = a ^ 3;
    v16i acc0 = {0}, acc1 = {0};
    v4i A = {(int)a, (int)b, (int)c, (int)d}, B = {1, 2, 3, 4};
    const bool do_m = same_wave ? true : (wave < 4) && (mode & 1);
    const bool do_v = same_wave ? true : (wave >= 4) && (mode & 2);
    for (int i = 0; i < iters; ++i) {
        if (do_m) {
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(B, A, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(B, A, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(B, A, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(B, A, acc1, 0, 0, 0);
        }
        if (do_v) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {      // 8 independent chains: throughput-bound
                a = (a & 0x01010101u) + 0x11u; b = (b & 0x02020202u) + 0x22u; c = (c & 0x04040404u) + 0x33u; d = (d & 0x08080808u) + 0x44u;
                e = (e & 0x10101010u) + 0x55u; f = (f & 0x20202020u) + 0x66u; g = (g & 0x40404040u) + 0x77u; h = (h & 0x80808080u) + 0x88u;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + acc0[0] + acc1[3];
}
int main() {
    unsigned* out; (void)hipMalloc(&out, 1 << 24);
    const int iters = 20000;
    for (int same = 0; same < 2; ++same)
        for (int mode = 1; mode <= 3; ++mode) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, 100, mode, same);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode, same);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("same_wave=%d mode=%d (%s%s): %.3f ms; per iter %.1f ns = %.0f cycles@2.4GHz  [4 MFMA; 128 VALU]\n", same, mode,
                   mode & 1 ? "MFMA " : "", mode & 2 ? "VALU" : "", ms, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
            if (same) break;
        }
    return 0;
}

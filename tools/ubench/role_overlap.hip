// micro-benchmark: "MFMA-role" waves (int8 MFMA chain with 5 VALU of operand preparation per MFMA, like the
// FIR kernel's MFMA phase) next to "VALU-role" waves (an epilogue-like mix) on the same SIMD.
//   roles per block of 12 waves: waves 0-3 MFMA role (one per SIMD), waves 4-11 VALU role (two per SIMD)
// prints cycles per MFMA for the MFMA role alone / together, and VALU-role throughput alone / together
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(768) void k(unsigned* out, int iters, int mode, unsigned long long* cyc) {
    const unsigned wave = threadIdx.x >> 6;
    unsigned a = threadIdx.x, b = a * 3 + 1, c = a ^ 0x55, d = a + 7;
    double x = a * 1e-3, y = 0.5, z = 1.25;
    v16i acc0 = {0}, acc1 = {0};
    v4i A = {(int)a, (int)b, (int)c, (int)d};
    unsigned w0 = a * 2654435761u, w1 = b * 2246822519u;
    const bool do_m = (wave < 4) && (mode & 1);
    const bool do_v = (wave >= 4) && (mode & 2);
    unsigned long long t0 = 0, t1 = 0;
    if (do_m) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v4i B0 = {(int)(w0 & 0x01010101u), (int)(w0 & 0x02020202u), (int)(w0 & 0x04040404u), (int)(w0 & 0x08080808u)};
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B0, acc0, 0, 0, 0);
                v4i B1 = {(int)(w1 & 0x01010101u), (int)(w1 & 0x02020202u), (int)(w1 & 0x04040404u), (int)(w1 & 0x08080808u)};
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B1, acc1, 0, 0, 0);
                w0 += 0x9E3779B9u; w1 ^= w0 >> 3;           // 2 more VALU per pair
            }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
        if ((threadIdx.x & 63) == 0) atomicAdd(&cyc[0], t1 - t0);
    }
    if (do_v) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {               // per j: ~12 int + 5 f64 ops, epilogue-like
                unsigned h = a + b; h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16;
                const unsigned term = (h & 0xFFFFu) + (h >> 16) + 1u;
                const double dd = fma((double)term, 0x1p-16, -1.0);
                x = fma(x, z, y) + dd;
                a = h + (unsigned)(int)x; b += 0x01010101u;
            }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
        if ((threadIdx.x & 63) == 0) atomicAdd(&cyc[1], t1 - t0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + (unsigned)x + acc0[0] + acc1[3];
}
int main() {
    unsigned* out; (void)hipMalloc(&out, 1 << 24);
    unsigned long long* cyc; (void)hipMalloc(&cyc, 16);
    const int iters = 4000;
    for (int mode = 1; mode <= 3; ++mode) {
        (void)hipMemset(cyc, 0, 16);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(256), dim3(768), 0, 0, out, 10, mode, cyc);
        (void)hipDeviceSynchronize(); (void)hipMemset(cyc, 0, 16);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(768), 0, 0, out, iters, mode, cyc);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        // s_memtime counts at 100 MHz; report kernel time based numbers instead
        const double cycles = ms * 1e-3 * 2.4e9;
        printf("mode %d (%s%s): %.3f ms = %.0f cycles; per MFMA-role iteration (8 MFMA) %.1f cycles; per VALU-role iteration (4 samples) %.1f cycles\n",
               mode, mode & 1 ? "MFMA " : "", mode & 2 ? "VALU" : "", ms, cycles, cycles / iters, cycles / iters);
    }
    return 0;
}

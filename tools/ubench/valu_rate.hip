// micro-benchmark: VALU issue rate per SIMD vs waves per SIMD (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int KIND>
__global__ void k(unsigned* out, int iters) {
    unsigned a = threadIdx.x, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, e = a * 5, f = a | 1, g = a + 11, h = a ^ 3;
    double x = a, y = b;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {      // 8 independent v_and/v_xor-ish ops
            a = (a & 0x01010101u) ^ b; b = (b & 0x02020202u) ^ c; c = (c & 0x04040404u) ^ d; d = (d & 0x08080808u) ^ e;
            e = (e & 0x10101010u) ^ f; f = (f & 0x20202020u) ^ g; g = (g & 0x40404040u) ^ h; h = (h & 0x80808080u) ^ a;
        } else if (KIND == 1) { // mul_lo
            a = a * 0x7feb352du + b; b = b * 0x846ca68bu + c; c = c * 0x7feb352du + d; d = d * 0x846ca68bu + a;
        } else {              // f64 fma
            x = fma(x, 1.0000001, y); y = fma(y, 0.9999999, x);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + (unsigned)x + (unsigned)y;
}
template <int KIND>
void run(const char* name, int ops_per_iter) {
    unsigned* out; hipMalloc(&out, 1 << 24);
    for (int wps = 1; wps <= 8; wps *= 2) {
        int blocks = 256, threads = 64 * 4 * wps;    // one block per CU, wps waves per SIMD
        if (threads > 1024) { blocks = 256 * (threads / 1024); threads = 1024; }
        int iters = 20000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, 100);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double wave_instrs_per_simd = (double)iters * ops_per_iter * wps;
        printf("%s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, wps, ms,
               ms * 1e6 / wave_instrs_per_simd, ms * 1e6 / wave_instrs_per_simd * 2.4);
    }
}
int main() { run<0>("and/xor", 16); run<1>("mul_lo+add", 8); run<2>("fma_f64", 2); return 0; }

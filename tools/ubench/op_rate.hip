// micro-benchmark: issue cost of individual VALU ops (8 independent chains per lane, 4 waves per SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int OP>
__global__ __launch_bounds__(1024) void k(double* out, int iters) {
    double x[8]; int v[8]; unsigned u[8];
    for (int j = 0; j < 8; ++j) { x[j] = threadIdx.x * 1.5 + j; v[j] = threadIdx.x + j; u[j] = threadIdx.x * 7u + j; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == 0) x[j] = fma(x[j], 1.0000001, 0.5);
            else if (OP == 1) x[j] = x[j] + 1.25;
            else if (OP == 2) { v[j] = (int)x[j]; x[j] = __hiloint2double(__double2hiint(x[j]) ^ (v[j] & 1), __double2loint(x[j])); }      // cvt_i32_f64 (+2 cheap)
            else if (OP == 3) { x[j] = (double)v[j]; v[j] += __double2hiint(x[j]) & 1; }   // cvt_f64_i32 (+cheap)
            else if (OP == 4) x[j] = trunc(x[j]) + 0.0;                                   // trunc + add
            else if (OP == 5) x[j] = fmax(x[j], 3.0);
            else if (OP == 6) u[j] = u[j] * 0x7feb352du;
            else if (OP == 7) u[j] = (u[j] & 0x01010101u) + 0x11u;
            else if (OP == 8) x[j] = copysign(0.5, x[j]) + x[j];
            else if (OP == 9) { x[j] = (double)u[j]; u[j] += __double2hiint(x[j]) & 1; }  // cvt_f64_u32
        }
    }
    double s = 0; for (int j = 0; j < 8; ++j) s += x[j] + v[j] + u[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, double extra_simple) {
    double* out; (void)hipMalloc(&out, 1 << 24);
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, out, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double per = ms * 1e6 / ((double)iters * 8 * 4);   // ns per statement per SIMD (4 waves per SIMD)
    printf("%-28s %.2f ns per statement per SIMD = %.1f cycles@2.4GHz (incl. ~%.0f cheap ops)\n", name, per, per * 2.4, extra_simple);
    (void)hipFree(out);
}
int main() {
    run<7>("and+add (2 simple)", 2); run<0>("v_fma_f64", 0); run<1>("v_add_f64", 0); run<2>("cvt_i32_f64 (+xor,and)", 2);
    run<3>("cvt_f64_i32 (+and,add)", 2); run<9>("cvt_f64_u32 (+and,add)", 2); run<4>("trunc_f64 + add_f64", 0); run<5>("v_max_f64", 0);
    run<6>("v_mul_lo_u32", 0); run<8>("copysign(bfi)+add_f64", 0);
    return 0;
}

// micro-benchmark: what one SIMD of gfx950 issues per cycle.
//   part 1: a stream of independent VALU ops of one kind, on W waves per SIMD (W = 1, 2, 4)
//   part 2: one MFMA wave per SIMD (back-to-back v_mfma_i32_32x32x32_i8) beside V VALU waves per SIMD
//   part 3: one wave per SIMD, K VALU ops between consecutive MFMAs (K = 0..8); and the same with W waves per SIMD
// prints s_memtime ticks (shader clock) per instruction per wave and per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define S_(x) #x
#define S(x) S_(x)
// 512 instructions per loop trip: 64 x 8 independent destinations
#define VALU512(OP) \
    asm volatile(".rept 64\n\t" OP(%0) "\n\t" OP(%1) "\n\t" OP(%2) "\n\t" OP(%3) "\n\t" OP(%4) "\n\t" OP(%5) "\n\t" OP(%6) "\n\t" OP(%7) "\n\t.endr" \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(x), "s"(sk));
#define OP_AND_LIT(d) "v_and_b32 " #d ", 0x10101010, %8"
#define OP_AND_SGPR(d) "v_and_b32 " #d ", %9, %8"
#define OP_AND_INL(d) "v_and_b32 " #d ", 15, %8"
#define OP_ADD(d) "v_add_u32 " #d ", %8, " #d
#define OP_XOR(d) "v_xor_b32 " #d ", %8, " #d
#define OP_MUL_LO(d) "v_mul_lo_u32 " #d ", %8, " #d
#define OP_LSHL_ADD(d) "v_lshl_add_u32 " #d ", %8, 8, " #d
#define OP_PERM(d) "v_perm_b32 " #d ", %8, " #d ", %9"
#define OP_MAD24(d) "v_mad_u32_u24 " #d ", %8, " #d ", " #d
#define OP_SAD16(d) "v_sad_u16 " #d ", %8, 0, " #d
#define OP_BFE(d) "v_bfe_u32 " #d ", %8, 4, 8"
#define OP_MED3(d) "v_med3_i32 " #d ", %8, " #d ", %9"
#define OP_ADD3(d) "v_add3_u32 " #d ", %8, " #d ", %9"
#define OP_XORLIT(d) "v_xor_b32 " #d ", 0x7feb352d, " #d

template <int KIND>
__global__ __launch_bounds__(1024) void valu_kernel(unsigned* out, int iters, unsigned long long* cyc) {
    unsigned r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    unsigned x = threadIdx.x * 2654435761u;
    unsigned sk = __builtin_amdgcn_readfirstlane(0x04020100u);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { VALU512(OP_AND_LIT) }
        if (KIND == 1) { VALU512(OP_AND_SGPR) }
        if (KIND == 2) { VALU512(OP_AND_INL) }
        if (KIND == 3) { VALU512(OP_ADD) }
        if (KIND == 4) { VALU512(OP_XOR) }
        if (KIND == 5) { VALU512(OP_MUL_LO) }
        if (KIND == 6) { VALU512(OP_LSHL_ADD) }
        if (KIND == 7) { VALU512(OP_PERM) }
        if (KIND == 8) { VALU512(OP_MAD24) }
        if (KIND == 9) { VALU512(OP_SAD16) }
        if (KIND == 10) { VALU512(OP_BFE) }
        if (KIND == 11) { VALU512(OP_MED3) }
        if (KIND == 12) { VALU512(OP_ADD3) }
        if (KIND == 13) { VALU512(OP_XORLIT) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) atomicAdd(&cyc[0], t1 - t0);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
}

template <int KIND>
__global__ __launch_bounds__(1024) void valu64_kernel(double* out, int iters, unsigned long long* cyc) {
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    double m = 1.0000001, c = 0.5;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0)
            asm volatile(".rept 128\n\tv_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5\n\t.endr"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
        if (KIND == 1)
            asm volatile(".rept 128\n\tv_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4\n\t.endr"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));
        if (KIND == 2)
            asm volatile(".rept 128\n\tv_cvt_f64_i32 %0, %4\n\tv_cvt_f64_i32 %1, %5\n\tv_cvt_f64_i32 %2, %6\n\tv_cvt_f64_i32 %3, %7\n\t.endr"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(i0), "v"(i1), "v"(i2), "v"(i3));
        if (KIND == 3)
            asm volatile(".rept 128\n\tv_cvt_i32_f64 %0, %4\n\tv_cvt_i32_f64 %1, %5\n\tv_cvt_i32_f64 %2, %6\n\tv_cvt_i32_f64 %3, %7\n\t.endr"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) atomicAdd(&cyc[0], t1 - t0);
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + i0 + i1 + i2 + i3;
}

// roles: waves 0-3 of a block (one per SIMD) issue MFMAs back to back; the other waves VALU
template <int VK>
__global__ __launch_bounds__(1024) void mix_kernel(unsigned* out, int iters, int mode, unsigned long long* cyc) {
    const unsigned wave = threadIdx.x >> 6;
    unsigned r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    unsigned x = threadIdx.x * 2654435761u;
    unsigned sk = __builtin_amdgcn_readfirstlane(0x10101010u);
    v16i acc0 = {0};
    v4i A = {(int)r0, (int)r1, (int)r2, (int)r3}, B = {(int)r4, (int)r5, (int)r6, (int)r7};
    const bool do_m = wave < 4 && (mode & 1);
    const bool do_v = wave >= 4 && (mode & 2);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (do_m) {
        for (int i = 0; i < iters; ++i)
            asm volatile(".rept 64\n\tv_mfma_i32_32x32x32_i8 %0, %1, %2, %0\n\t.endr" : "+v"(acc0) : "v"(A), "v"(B));
    }
    if (do_v) {
        for (int i = 0; i < iters; ++i) {
            if (VK == 0) { VALU512(OP_AND_SGPR) }
            if (VK == 1) { VALU512(OP_AND_LIT) }
            if (VK == 2) { VALU512(OP_MUL_LO) }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && (do_m || do_v)) atomicAdd(&cyc[do_m ? 0 : 1], t1 - t0);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + acc0[0];
}

// every wave: K independent VALU (v_and with SGPR mask) after every MFMA; W waves per SIMD
#define GAPBODY(KSTR) \
    asm volatile(".rept 32\n\tv_mfma_i32_32x32x32_i8 %8, %9, %10, %8\n\t" KSTR ".endr" \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(acc0) : "v"(A), "v"(B), "v"(x), "s"(sk));
#define A1 "v_and_b32 %0, %12, %11\n\t"
#define A2 A1 "v_and_b32 %1, %12, %11\n\t"
#define A3 A2 "v_and_b32 %2, %12, %11\n\t"
#define A4 A3 "v_and_b32 %3, %12, %11\n\t"
#define A5 A4 "v_and_b32 %4, %12, %11\n\t"
#define A6 A5 "v_and_b32 %5, %12, %11\n\t"
#define A7 A6 "v_and_b32 %6, %12, %11\n\t"
#define A8 A7 "v_and_b32 %7, %12, %11\n\t"
#define A12 A8 "v_and_b32 %0, %12, %11\n\tv_and_b32 %1, %12, %11\n\tv_and_b32 %2, %12, %11\n\tv_and_b32 %3, %12, %11\n\t"
template <int K>
__global__ __launch_bounds__(1024) void gap_kernel(unsigned* out, int iters, unsigned long long* cyc) {
    unsigned r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    unsigned x = threadIdx.x * 2654435761u;
    unsigned sk = __builtin_amdgcn_readfirstlane(0x10101010u);
    v16i acc0 = {0};
    v4i A = {(int)r0, (int)r1, (int)r2, (int)r3}, B = {(int)r4, (int)r5, (int)r6, (int)r7};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (K == 0) { GAPBODY("") }
        if (K == 1) { GAPBODY(A1) }
        if (K == 2) { GAPBODY(A2) }
        if (K == 3) { GAPBODY(A3) }
        if (K == 4) { GAPBODY(A4) }
        if (K == 5) { GAPBODY(A5) }
        if (K == 6) { GAPBODY(A6) }
        if (K == 7) { GAPBODY(A7) }
        if (K == 8) { GAPBODY(A8) }
        if (K == 12) { GAPBODY(A12) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) atomicAdd(&cyc[0], t1 - t0);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + acc0[0];
}

static unsigned* out; static unsigned long long* cyc;
template <class F> static void timeit(const char* name, double n_inst_per_wave_iter, int waves_per_simd, int n0, int n1, F launch) {
    const int IT = 100;
    (void)hipMemset(cyc, 0, 16);
    launch(2);
    (void)hipDeviceSynchronize(); (void)hipMemset(cyc, 0, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    launch(IT);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    const double ninst = n_inst_per_wave_iter * IT;
    printf("%-46s ms %.3f | ticks/inst per wave: %6.2f", name, ms, n0 ? (double)h[0] / n0 / ninst : 0.0);
    if (n1) printf("  role2: %6.2f", (double)h[1] / n1 / ninst);
    printf("  | per SIMD (W=%d): %.2f\n", waves_per_simd, n0 ? (double)h[0] / n0 / ninst / waves_per_simd : 0.0);
}
int main() {
    (void)hipMalloc(&out, 1 << 26); (void)hipMalloc(&cyc, 16);
    for (int W = 1; W <= 4; W *= 2) {
#define VK(K, NAME) { char nm[96]; snprintf(nm, sizeof nm, "valu %-12s %d wave/SIMD", NAME, W); \
        timeit(nm, 512, W, 256 * 4 * W, 0, [&](int it) { hipLaunchKernelGGL(valu_kernel<K>, dim3(256), dim3(256 * W), 0, 0, out, it, cyc); }); }
        VK(0, "and lit") VK(1, "and sgpr") VK(2, "and inline") VK(3, "add_u32") VK(4, "xor") VK(5, "mul_lo_u32") VK(6, "lshl_add")
        VK(7, "perm") VK(8, "mad_u32_u24") VK(9, "sad_u16") VK(10, "bfe") VK(11, "med3") VK(12, "add3") VK(13, "xor lit")
#define VK64(K, NAME) { char nm[96]; snprintf(nm, sizeof nm, "valu %-12s %d wave/SIMD", NAME, W); \
        timeit(nm, 512, W, 256 * 4 * W, 0, [&](int it) { hipLaunchKernelGGL(valu64_kernel<K>, dim3(256), dim3(256 * W), 0, 0, (double*)out, it, cyc); }); }
        VK64(0, "fma_f64") VK64(1, "add_f64") VK64(2, "cvt_f64_i32") VK64(3, "cvt_i32_f64")
    }
    // MFMA wave (64 MFMA per trip) beside V VALU waves (512 VALU per trip): ticks per instruction of each role
    for (int V = 0; V <= 3; ++V)
        for (int mode = 1; mode <= 3; ++mode) {
            if (V == 0 && mode != 1) continue;
            char nm[96]; snprintf(nm, sizeof nm, "mix %s%s V=%d and-sgpr", mode & 1 ? "MFMA " : "", mode & 2 ? "VALU" : "", V);
            (void)hipMemset(cyc, 0, 16);
            const int IT = 100;
            hipLaunchKernelGGL(mix_kernel<0>, dim3(256), dim3(256 * (1 + V)), 0, 0, out, 2, mode, cyc);
            (void)hipDeviceSynchronize(); (void)hipMemset(cyc, 0, 16);
            hipLaunchKernelGGL(mix_kernel<0>, dim3(256), dim3(256 * (1 + V)), 0, 0, out, IT, mode, cyc);
            (void)hipDeviceSynchronize();
            unsigned long long h[2]; (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
            printf("%-46s ticks per MFMA %.2f | ticks per VALU per wave %.2f, per SIMD %.2f\n", nm, (double)h[0] / 1024 / (64.0 * IT),
                   V ? (double)h[1] / (1024.0 * V) / (512.0 * IT) : 0.0, V ? (double)h[1] / (1024.0 * V) / (512.0 * IT) / V : 0.0);
        }
    for (int W = 1; W <= 4; ++W) {
        if (W == 3) continue;
#define GAP(K) { char nm[96]; snprintf(nm, sizeof nm, "gap K=%d, %d wave/SIMD: ticks per (MFMA+K VALU)", K, W); \
        timeit(nm, 32, W, 256 * 4 * W, 0, [&](int it) { hipLaunchKernelGGL(gap_kernel<K>, dim3(256), dim3(256 * W), 0, 0, out, it, cyc); }); }
        GAP(0) GAP(1) GAP(2) GAP(3) GAP(4) GAP(5) GAP(6) GAP(7) GAP(8) GAP(12)
    }
    return 0;
}

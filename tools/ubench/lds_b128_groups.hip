// lds_b128_groups.hip -- which lanes of a wave does gfx950 serve together for ds_read_b128 / ds_read_b64 / ds_read_b32, i.e. which
// lanes' addresses must fall on different banks?  Method: all lanes read linear conflict-free slots (16 bytes x lane), except that
// lane j reads lane i's banks in another row (+ 4096 bytes: same banks); the extra time says whether i and j are served together.
// Then the candidate operand layouts of the stage-B kernel (rows of pitch RP, lane = column + 16 kgroup).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int W>
__global__ void rate(unsigned long long* out, const uint32_t* addr, int n) {
    extern __shared__ __align__(16) uint8_t lds[];
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = (uint8_t)i;
    __syncthreads();
    const uint32_t a = (uint32_t)(uintptr_t)lds + addr[threadIdx.x & 63];
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        if constexpr (W == 16) {
            u32x4 v0, v1, v2, v3;
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:8192\n ds_read_b128 %2, %4 offset:16384\n ds_read_b128 %3, %4 offset:24576\n s_waitcnt lgkmcnt(0)"
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(a));
            acc += v0 ^ v1 ^ v2 ^ v3;
        } else if constexpr (W == 8) {
            u32x2 v0, v1, v2, v3;
            asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8192\n ds_read_b64 %2, %4 offset:16384\n ds_read_b64 %3, %4 offset:24576\n s_waitcnt lgkmcnt(0)"
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(a));
            acc.x += v0.x ^ v1.x ^ v2.y ^ v3.y;
        } else {
            uint32_t v0, v1, v2, v3;
            asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:8192\n ds_read_b32 %2, %4 offset:16384\n ds_read_b32 %3, %4 offset:24576\n s_waitcnt lgkmcnt(0)"
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(a));
            acc.x += v0 ^ v1 ^ v2 ^ v3;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (acc.x == 0x12345) out[1] = acc.y;
}

static uint32_t* d_addr; static unsigned long long* d_out;
template <int W>
static double run(const std::vector<uint32_t>& a) {
    const int n = 4000;
    hipMemcpy(d_addr, a.data(), 256, hipMemcpyHostToDevice);
    unsigned long long h = 0;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(rate<W>, dim3(1), dim3(64), 32768, 0, d_out, d_addr, n);
    hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
    return (double)h / (4.0 * n);      // s_memtime ticks (100 MHz) per read... relative numbers are what matters
}

int main() {
    hipMalloc(&d_addr, 256); hipMalloc(&d_out, 64);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&rate<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&rate<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&rate<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
    std::vector<uint32_t> lin(64);
    for (int w : {16, 8, 4}) {
        for (int l = 0; l < 64; ++l) lin[l] = (uint32_t)(w * l);
        const double base = w == 16 ? run<16>(lin) : w == 8 ? run<8>(lin) : run<4>(lin);
        printf("width %2d: linear %.3f ticks per read; lanes that collide with lane i when they share its banks:\n", w, base);
        for (int i : {0, 5, 20, 37}) {
            printf("  i = %2d:", i);
            for (int j = 0; j < 64; ++j) {
                if (j == i) continue;
                std::vector<uint32_t> a = lin;
                a[j] = lin[i] + 4096u;                   // lane j on lane i's banks, another row
                const double t = w == 16 ? run<16>(a) : w == 8 ? run<8>(a) : run<4>(a);
                if (t > base * 1.05) printf(" %d", j);
            }
            printf("\n");
        }
    }
    // stage-B operand layouts: lane = col + 16 kg reads 16 bytes at row(col) * RP + 16 kg (+ a common offset)
    printf("ds_read_b128, lane = col + 16 kg:\n");
    for (int rp : {256, 272, 288, 304, 320, 336, 400, 208, 240}) {
        std::vector<uint32_t> a(64);
        for (int l = 0; l < 64; ++l) a[l] = (uint32_t)((l & 15) * rp + 16 * (l >> 4));
        printf("  pitch %3d: %.3f\n", rp, run<16>(a));
    }
    // rows permuted: row slot of column c = sigma(c)
    printf("pitch 272 with the kgroup slot rotated by the column: 16 ((kg + (col >> s)) & 3)\n");
    for (int s = 0; s < 4; ++s) {
        std::vector<uint32_t> a(64);
        for (int l = 0; l < 64; ++l) a[l] = (uint32_t)((l & 15) * 272 + 16 * (((l >> 4) + ((l & 15) >> s)) & 3));
        printf("  s = %d: %.3f\n", s, run<16>(a));
    }
    printf("pitch p, column c at row slot with an extra 16-byte shift for c >= 8:  c * p + 16 kg + 16 * x * (c >> 3)\n");
    for (int p : {256, 272, 288})
        for (int x : {1, 2, 3, 4, 5}) {
            std::vector<uint32_t> a(64);
            for (int l = 0; l < 64; ++l) a[l] = (uint32_t)((l & 15) * p + 16 * (l >> 4) + 16 * x * ((l & 15) >> 3));
            printf("  p = %d x = %d: %.3f\n", p, x, run<16>(a));
        }
    return 0;
}

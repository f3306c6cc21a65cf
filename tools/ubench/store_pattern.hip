// store_pattern.hip -- what the write path sustains for the frame-store patterns of the FIR kernels, against plain coalesced stores.
// A wave owns one tile per trip (512 frames) and writes it with the same instructions the kernels use:
//   0  coalesced: every instruction writes 1 KiB contiguous (16 bytes per lane)
//   1  float frames from registers: lane (n = l & 31, h = l >> 5) owns frames 16n + 4h + 8g + {0..3}: 4 dwordx4 stores, 16-byte pieces
//      32 bytes apart
//   2  24-bit frames from registers: the lane's 4 frames = 24 bytes: dwordx4 + dwordx2, per g
//   3  as 0 with one 16-byte load per 4 stores (the FIR kernels' read:write mix at M = 8)
// Optionally a read stream beside it (mode 3).  Prints GB/s for each.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void k(uint8_t* out, const uint8_t* in, uint32_t ntiles, uint32_t* sink) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t wv = blockIdx.x * 8 + wave, stride = gridDim.x * 8;
    const uint32_t n = lane & 31, h = lane >> 5;
    constexpr uint32_t TB = MODE == 2 ? 3072 : 4096;         // bytes per tile
    u32x4 v = {lane, wv, 3, 4};
    uint32_t acc = 0;
    for (uint32_t t = wv; t < ntiles; t += stride) {
        uint8_t* o = out + (size_t)t * TB;
        v.x += t;
        if (MODE == 0 || MODE == 3) {
            if (MODE == 3) { const u32x4 r = *reinterpret_cast<const u32x4*>(in + (size_t)t * 1024 + 16 * lane); acc += r.x; }
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(o + 1024 * j + 16 * lane) = v;
        } else if (MODE == 1) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                *reinterpret_cast<u32x4*>(o + 128 * n + 32 * h + 64 * g) = v;
                *reinterpret_cast<u32x4*>(o + 128 * n + 32 * h + 64 * g + 16) = v;
            }
        } else {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                *reinterpret_cast<u32x4*>(o + 96 * n + 24 * h + 48 * g) = v;
                *reinterpret_cast<u32x2*>(o + 96 * n + 24 * h + 48 * g + 16) = u32x2{v.x, v.y};
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE>
static void run(const char* name, uint8_t* out, const uint8_t* in, uint32_t* sink, size_t bytes, int gx) {
    const uint32_t TB = MODE == 2 ? 3072 : 4096;
    const uint32_t ntiles = (uint32_t)(bytes / TB);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(gx), dim3(512), 0, 0, out, in, ntiles, sink);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MODE>, dim3(gx), dim3(512), 0, 0, out, in, ntiles, sink);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double wr = (double)ntiles * TB, rd = MODE == 3 ? (double)ntiles * 1024 : 0;
    printf("%-44s grid %5d: %.3f ms, written %.2f TB/s, moved %.2f TB/s\n", name, gx, ms, wr / ms * 1e-9, (wr + rd) / ms * 1e-9);
}

int main() {
    const size_t bytes = (size_t)12 << 30;
    uint8_t *out, *in; uint32_t* sink;
    if (hipMalloc(&out, bytes) != hipSuccess || hipMalloc(&in, bytes / 4) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(in, 1, bytes / 4);
    for (int gx : {512, 1024, 4096}) {
        run<0>("coalesced 1 KiB per instruction", out, in, sink, bytes, gx);
        run<1>("float frames, 16-byte pieces 32 apart", out, in, sink, bytes, gx);
        run<2>("24-bit frames, 16+8-byte pieces 24 apart", out, in, sink, bytes, gx);
        run<3>("coalesced + one load per 4 stores", out, in, sink, bytes, gx);
    }
    return 0;
}

// d2d_mfma2_dev.h -- device-side pieces shared by the two-group matrix-core kernels (d2d_kernels_mfma2.hip and its
// software-pipelined stereo variant d2d_kernels_mfma3.hip): launch arguments, staging geometry, the byte-gather path.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "d2d_device.h"
#include "d2d_launch.h"
#include "d2d_mfma.h"

namespace d2d {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

struct Mfma2Args {
    FirArgs f;
    double c1, c0;        // x = fma(acc128, c1, -c0) == round(y*c0): c1 = 2^(1-S-7)*c0, c0 = scale | gain | 2^S
    double dmul, dadd;    // integer depths: d = fma(term, dmul, dadd)
    uint32_t dkind;       // 0: no dither, 1: triangular, 2: rectangular
    uint32_t qsh;         // 4 for 20-bit samples in a 24-bit container, else 0
    int32_t qmin_i, qmax_i;
    uint32_t wide;        // 1: limb sums may exceed 2^23, recombine in f64
    uint32_t off_waves;   // LDS: start of the per-wave regions (after the shared tap table)
    uint32_t wave_lds;    // LDS bytes per wave
    uint32_t off_out;     // the wave's output slice inside its region
    uint32_t nwaves;      // waves per block
    uint32_t ngroups;     // channel groups per file: 1 for mono/stereo, else one block row per channel PAIR
    uint32_t intq;        // 1: unit gain at an integer depth -- the all-integer requantiser applies
    uint32_t gainq;       // 1 (pipelined kernels): another level in dB -- the f64 requantiser inside the pipelined epilogue (KIND + 4)
    int32_t  fbits;       // intq: x = v * 2^-fbits LSB (v = sum q s), fbits = S - (bits - 1)
    uint32_t dbg;         // diagnostic ablation mask (make DIAG=1, env D2D_DBG): 1 no chain, 2 no epilogue, 4 no staging
    uint32_t npairs;      // fp6 kernel: channel pairs a wave converts per tile (1; 3: planar 5.1 frames -- whole frames from one wave, one block row per file)
};

#ifndef D2D_DIAG
#define D2D_DIAG 0
#endif

constexpr int M2_TILE = 512;          // outputs per wave-tile and channel

template <int MB>
struct M2Geom {
    static constexpr int RS = 4 * MB;                               // row stride in dwords (16 outputs)
    static constexpr int LSH = MB == 1 ? 2 : MB == 2 ? 3 : MB == 4 ? 4 : MB == 8 ? 5 : 6;
};

// plane 0 unmasked: a byte then weighs up to 128*128 + 254*128 in a limb sum; the int32 recombination needs the sums below 2^23
__host__ __device__ constexpr bool m2_unmask0(int NPG) { return (long long)NPG * 8 * (128 * 128 + 254 * 128) < (1 << 23); }
__host__ __device__ constexpr int m2_span_dw(int MB, int NPG) { return 31 * 4 * MB + 2 * (NPG + MB); }
__host__ __device__ constexpr int m2_chunks(int MB, int NPG) { return (m2_span_dw(MB, NPG) + 3 + 3) / 4; }   // + up to 3 dwords in front
__host__ __device__ constexpr int m2_pf(int MB, int NPG) { return (m2_chunks(MB, NPG) + 63) / 64; }
__host__ __device__ constexpr int m2_stream_bytes(int MB, int NPG) {
    const int dw = 4 * 64 * m2_pf(MB, NPG);
    const int lsh = MB == 1 ? 2 : MB == 2 ? 3 : MB == 4 ? 4 : MB == 8 ? 5 : 6;
    return (((dw + (dw >> lsh) + 4) * 4 + 15) & ~15) + 16;   // + a dummy slot for the dwords in front of the window
}

__device__ __forceinline__ void wave_sync2() {
    // LDS operations of one wave execute in order; this only stops the compiler from moving them.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 16 bytes of the channel's stream starting at call-relative byte j (any alignment): the slow,
// always-right path (history, ragged blocks, interleaved layouts, call edges)
static __device__ __noinline__ u32x4 gather_chunk(const StreamJob* job, uint32_t C, uint32_t B, uint32_t keep, int32_t j) {
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll 1
    for (int b = 0; b < 16; ++b) {
        const uint32_t x = stream_byte(*job, C, B, keep, j + b) << (8 * (b & 3));
        if ((b >> 2) == 0) w[0] |= x; else if ((b >> 2) == 1) w[1] |= x; else if ((b >> 2) == 2) w[2] |= x; else w[3] |= x;
    }
    return u32x4{w[0], w[1], w[2], w[3]};
}


// defined in d2d_kernels_mfma2.hip / d2d_kernels_mfma3.hip
hipError_t launch_fir_mfma3(Mfma2Args& m, int variant, int MB, int NPG, int NT, uint32_t nwt_max, uint32_t nrows, hipStream_t s);
bool mfma3_supported(int MB, int NPG, int NT);
bool mfma3_sparse_compiled(int MB, int NT);
bool mfma3_scr_supported(int MB, int NPG);

}  // namespace d2d

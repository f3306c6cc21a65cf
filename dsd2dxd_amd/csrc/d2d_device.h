// d2d_device.h -- device-side building blocks shared by the FIR kernels:
//   * channel addressing of the packed DSD bytes (planar / interleaved, carried history)
//   * cooperative staging of a channel's byte window into LDS with 16-byte global loads
//   * the epilogue: level, dither, requantise, pack (bit-exact against oracle/d2d_oracle.c)
// Compiled with -ffp-contract=off: every f64 operation is the single IEEE operation written.
#pragma once
#include <hip/hip_runtime.h>

#include "d2d_internal.h"

namespace d2d {

// Pointers read out of a StreamJob are generic to the compiler; these casts say "global memory"
// so that loads and stores become global_* instead of flat_*.
#define D2D_GLOBAL __attribute__((address_space(1)))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ D2D_GLOBAL T* as_global(T* p) { return (D2D_GLOBAL T*)(p); }

// a2: byte j of channel `ch` in a call that feeds L bytes per channel.
// planar = [ch0 blk][ch1 blk]..., interleaved = block size 1 (README.md:9 of the reference).
// 32-bit index math: one call feeds fewer than 2^31 bytes per channel.
__device__ __forceinline__ uint64_t layout_addr(uint32_t C, uint32_t B, uint32_t L, uint32_t ch, uint32_t j) {
    if (B == 1) return (uint64_t)j * C + ch;
    const uint32_t blk = (B & (B - 1)) == 0 ? j >> (31 - __builtin_clz(B)) : j / B;
    const uint32_t off = j - blk * B;
    uint32_t blen = L - blk * B;
    if (blen > B) blen = B;
    return (uint64_t)blk * B * C + (uint64_t)ch * blen + off;
}

// One raw byte of the channel's stream at call-relative index j (negative = history).
__device__ __noinline__ uint32_t stream_byte(const StreamJob& job, uint32_t C, uint32_t B, uint32_t keep, int32_t j) {
    if (j < 0) {
        const int32_t h = (int32_t)keep + j;
        return h >= 0 ? as_global(job.hist)[h] : 0u;
    }
    if ((uint32_t)j >= (uint32_t)job.L) return 0u;
    return as_global(job.in)[layout_addr(C, B, (uint32_t)job.L, job.ch, (uint32_t)j)];
}

// Stage the channel bytes [abeg, abeg + nbytes) (abeg % 16 == 0, nbytes % 16 == 0) into LDS.
// Fast path: one global_load_dwordx4 per 16 bytes when the chunk lies inside one planar block
// whose start is 16-byte aligned; otherwise byte gathers (history, short last block, interleaved).
__device__ __forceinline__ void stage_window(uint8_t* lds, const StreamJob& job, uint32_t C, uint32_t B,
                                             uint32_t keep, int64_t abeg64, uint32_t nbytes,
                                             uint32_t tid, uint32_t nthreads) {
    const uint32_t nchunks = nbytes >> 4;
    const bool aligned_layout = (B & 15u) == 0;
    const uint32_t L = (uint32_t)job.L;
    const int32_t abeg = (int32_t)abeg64;
    for (uint32_t q = tid; q < nchunks; q += nthreads) {
        const int32_t j0 = abeg + (int32_t)(q * 16);
        u32x4 v;
        bool fast = false;
        if (aligned_layout && j0 >= 0 && (uint32_t)j0 + 16 <= L) {
            const uint32_t j = (uint32_t)j0;
            const uint32_t blk = (B & (B - 1)) == 0 ? j >> (31 - __builtin_clz(B)) : j / B;
            const uint32_t off = j - blk * B;
            uint32_t blen = L - blk * B;
            if (blen > B) blen = B;
            if ((blen & 15u) == 0) {
                const uint8_t* p = job.in + (uint64_t)blk * B * C + (uint64_t)job.ch * blen + off;
                v = *reinterpret_cast<D2D_GLOBAL const u32x4*>(as_global(p));
                fast = true;
            }
        }
        if (!fast) {
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll 1
            for (int b = 0; b < 16; ++b) {
                const uint32_t x = stream_byte(job, C, B, keep, j0 + b) << (8 * (b & 3));
                if ((b >> 2) == 0) w[0] |= x; else if ((b >> 2) == 1) w[1] |= x; else if ((b >> 2) == 2) w[2] |= x; else w[3] |= x;
            }
            v = u32x4{w[0], w[1], w[2], w[3]};
        }
        *reinterpret_cast<u32x4*>(lds + (size_t)q * 16) = v;
    }
}

// [own] counter-based dither generator, identical to orc_rng() in oracle/d2d_oracle.c:
// word = lowbias32(lo32(n) + k32 + hi32(n)*kstep).  The host folds hi32(n0) into job.rng_key; a
// call spans fewer than 2^32 outputs, so lo32(n) wraps at most once inside it.
__device__ __forceinline__ uint32_t rng32(const StreamJob& job, uint64_t n) {
    const uint32_t lo = (uint32_t)n;
    uint32_t x = lo + job.rng_key + (lo < job.rng_lo0 ? job.rng_kstep : 0u);
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

// a5-a7 for the integer depths: level, dither, round half away from zero, clip.
// Returns the value as it sits in the container (20-bit samples already shifted into 24 bits).
__device__ __forceinline__ int32_t quantise_int(const Epilogue& ep, double y, uint32_t rnd) {
    const double x = y * ep.scale;
    double d = 0.0;
    if (ep.dither == 'T') d = (double)((rnd & 0xFFFFu) + (rnd >> 16) + 1u) * 0x1p-16 - 1.0;
    else if (ep.dither == 'R') d = (double)(2u * (rnd >> 16) + 1u) * 0x1p-17 - 0.5;
    const double q = x + d;
    const double r = trunc(q + copysign(0.5, q));      // == (q >= 0 ? floor(q + .5) : ceil(q - .5))
    const int32_t lim = 1 << (ep.bits - 1);
    int32_t iv = (int32_t)fmax(fmin(r, 2147483520.0), -2147483648.0);
    iv = min(max(iv, -lim), lim - 1);
    return ep.bits == 20 ? iv * 16 : iv;
}

// ... and for 32-bit float output (Airwindows "Dither Float" when dither == 'F', else a plain cast)
__device__ __forceinline__ float quantise_f32(const Epilogue& ep, double y, uint32_t rnd) {
    double x = y * ep.gain;
    if (ep.dither == 'F') {
        const uint32_t fb = __float_as_uint((float)x);
        const int e = (int)((fb >> 23) & 0xFF);
        const int expon = e ? e - 126 : 0;
        const double t = ((double)rnd - 2147483647.0) * 5.5e-36;
        x = x + ldexp(t, expon + 62);
    }
    return (float)x;
}

// The same two steps for callers that already hold x = round(y*scale) (integer depths) or
// x = round(y*gain) (float): identical arithmetic from `x` on.
__device__ __forceinline__ int32_t finish_int(const Epilogue& ep, double x, uint32_t rnd) {
    double d = 0.0;
    if (ep.dither == 'T') d = fma((double)((rnd & 0xFFFFu) + (rnd >> 16) + 1u), 0x1p-16, -1.0);   // exact either way
    else if (ep.dither == 'R') d = fma((double)(2u * (rnd >> 16) + 1u), 0x1p-17, -0.5);
    const double q = x + d;
    const double lim = (double)(1u << (ep.bits - 1));
    const double r = fmax(fmin(trunc(q + copysign(0.5, q)), lim - 1.0), -lim);
    const int32_t iv = (int32_t)r;
    return ep.bits == 20 ? iv * 16 : iv;
}

__device__ __forceinline__ float finish_f32(const Epilogue& ep, double x, uint32_t rnd) {
    if (ep.dither == 'F') {
        const uint32_t fb = __float_as_uint((float)x);
        const int e = (int)((fb >> 23) & 0xFF);
        const int expon = e ? e - 126 : 0;
        const double t = ((double)rnd - 2147483647.0) * 5.5e-36;
        x = x + ldexp(t, expon + 62);
    }
    return (float)x;
}

// One sample straight to memory (used by the LUT and resampler kernels); returns |y*gain|.
__device__ __forceinline__ double emit_sample(const Epilogue& ep, const StreamJob& job, double y, uint64_t n, uint8_t* dst) {
    const uint32_t rnd = rng32(job, n);
    if (ep.bits == 32) {
        *reinterpret_cast<float*>(dst) = quantise_f32(ep, y, rnd);   // frame stride is a multiple of 4 bytes
    } else {
        const int32_t iv = quantise_int(ep, y, rnd);
        if (ep.bits == 16) {
            *reinterpret_cast<uint16_t*>(dst) = (uint16_t)iv;
        } else {
            dst[0] = (uint8_t)iv; dst[1] = (uint8_t)(iv >> 8); dst[2] = (uint8_t)(iv >> 16);
        }
    }
    return fabs(y * ep.gain);
}

// The two samples of a channel PAIR into a wider frame (multichannel files are converted pair by pair): 6 / 4 / 8 bytes that start on an even
// byte of the frame -- one 4-byte and one 2-byte store for 24-bit samples (the three byte stores per sample this replaces were what stage B
// of an 8-channel conversion spent most of its time on), one store otherwise.  The frames' base is 16-byte aligned and the pair's offset
// even, so 2-byte alignment is all these stores have.
typedef uint32_t u32_a2 __attribute__((aligned(2)));
typedef uint32_t u32x2_a2 __attribute__((ext_vector_type(2), aligned(2)));
__device__ __forceinline__ void store_pair_in_frame(uint8_t* dst, uint32_t L, uint32_t R, uint32_t SBY) {
    if (SBY == 3) {
        *reinterpret_cast<D2D_GLOBAL u32_a2*>(as_global(dst)) = (L & 0x00FFFFFFu) | (R << 24);
        *reinterpret_cast<D2D_GLOBAL uint16_t*>(as_global(dst + 4)) = (uint16_t)(R >> 8);
    } else if (SBY == 2) {
        *reinterpret_cast<D2D_GLOBAL u32_a2*>(as_global(dst)) = (L & 0xFFFFu) | (R << 16);
    } else {
        *reinterpret_cast<D2D_GLOBAL u32x2_a2*>(as_global(dst)) = u32x2_a2{L, R};
    }
}

// Which (file, channel group) a block row converts.  The channel groups of one multichannel file write into the SAME frames (a pair's 6 bytes
// of every 24-byte frame of an 8-channel stream), so their partial line writes should meet in one L2: workgroups go to the 8 XCDs round
// robin by their flat index (x fastest), and with one block per row the groups of a file would land on different XCDs, whose L2s cannot
// merge them (stage B of the config-5 shape: 21 ms for 4.4 GB of frames).  Rows y = 8 k ng + 8 s + x (x < 8, s < ng) of whole batches of
// eight files are therefore dealt out as file 8 k + x, group s: the ng blocks of a file sit 8 apart -- same XCD, dispatched together.
// Grids with several blocks per row (gridDim.x a multiple of 8) already have that property and keep the plain order.
__device__ __forceinline__ void row_to_file_group(uint32_t y, uint32_t nrows, uint32_t ng, uint32_t gx, uint32_t& file, uint32_t& grp) {
    const uint32_t batch = 8u * ng;
    if (ng > 1 && (gx & 7u) != 0 && y < nrows / batch * batch) {
        const uint32_t k = y / batch, r = y - k * batch;
        file = 8u * k + (r & 7u); grp = r >> 3;
    } else {
        file = y / ng; grp = y - file * ng;
    }
}

// Non-negative doubles order like their bit patterns: one atomic per block for the peak meter.
__device__ __forceinline__ void block_peak_max(double pk, double* dst, double* lds_red /* [nwaves] */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pk = fmax(pk, __shfl_xor(pk, o));
    const uint32_t wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) lds_red[wave] = pk;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = 0.0;
        for (uint32_t i = 0; i < nw; ++i) m = fmax(m, lds_red[i]);
        if (m > 0.0) atomicMax(reinterpret_cast<unsigned long long*>(dst), (unsigned long long)__double_as_longlong(m));
    }
}

}  // namespace d2d

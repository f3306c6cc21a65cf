// d2d_device.h -- device-side building blocks shared by the FIR kernels:
//   * channel addressing of the packed DSD bytes (planar / interleaved, carried history)
//   * cooperative staging of a channel's byte window into LDS with 16-byte global loads
//   * the epilogue: level, dither, requantise, pack (bit-exact against oracle/d2d_oracle.c)
// Compiled with -ffp-contract=off: every f64 operation is the single IEEE operation written.
#pragma once
#include <hip/hip_runtime.h>

#include "d2d_internal.h"

namespace d2d {

// a2: byte j of channel `ch` in a call that feeds L bytes per channel.
// planar = [ch0 blk][ch1 blk]..., interleaved = block size 1 (README.md:9 of the reference).
__device__ __forceinline__ uint64_t layout_addr(uint32_t C, uint32_t B, uint64_t L, uint32_t ch, uint64_t j) {
    if (B == 1) return j * C + ch;
    uint64_t blk = j / B, off = j - blk * B;
    uint64_t blen = L - blk * B;
    if (blen > B) blen = B;
    return blk * (uint64_t)B * C + (uint64_t)ch * blen + off;
}

// One raw byte of the channel's stream at call-relative index j (negative = history).
__device__ __forceinline__ uint8_t stream_byte(const StreamJob& job, uint32_t C, uint32_t B, uint32_t keep, int64_t j) {
    if (j < 0) {
        int64_t h = (int64_t)keep + j;
        return h >= 0 ? job.hist[h] : (uint8_t)0;
    }
    if ((uint64_t)j >= job.L) return 0;
    return job.in[layout_addr(C, B, job.L, job.ch, (uint64_t)j)];
}

// Stage the channel bytes [abeg, abeg + nbytes) (abeg % 16 == 0, nbytes % 16 == 0) into LDS.
// Fast path: one global_load_dwordx4 per 16 bytes when the chunk lies inside one planar block
// whose start is 16-byte aligned; otherwise byte gathers (history, short last block, interleaved).
__device__ __forceinline__ void stage_window(uint8_t* lds, const StreamJob& job, uint32_t C, uint32_t B,
                                             uint32_t keep, int64_t abeg, uint32_t nbytes,
                                             uint32_t tid, uint32_t nthreads) {
    const uint32_t nchunks = nbytes >> 4;
    const bool aligned_layout = (B & 15u) == 0;
    for (uint32_t q = tid; q < nchunks; q += nthreads) {
        int64_t j0 = abeg + (int64_t)q * 16;
        uint4 v;
        bool fast = false;
        if (aligned_layout && j0 >= 0 && (uint64_t)j0 + 16 <= job.L) {
            uint64_t blk = (uint64_t)j0 / B, off = (uint64_t)j0 - blk * B;
            uint64_t blen = job.L - blk * B;
            if (blen > B) blen = B;
            if ((blen & 15u) == 0) {
                const uint8_t* p = job.in + blk * (uint64_t)B * C + (uint64_t)job.ch * blen + off;
                v = *reinterpret_cast<const uint4*>(p);
                fast = true;
            }
        }
        if (!fast) {
            uint32_t w[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                uint32_t x = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) x |= (uint32_t)stream_byte(job, C, B, keep, j0 + d * 4 + b) << (8 * b);
                w[d] = x;
            }
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        *reinterpret_cast<uint4*>(lds + (size_t)q * 16) = v;
    }
}

// [own] counter-based dither generator, identical to orc_rng() in oracle/d2d_oracle.c.
__device__ __forceinline__ uint64_t rng64(uint64_t seed, uint32_t channel, uint64_t n) {
    uint64_t z = (seed ^ ((uint64_t)channel * 0xD1B54A32D192ED03ull)) + (n + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// a5-a7: level, dither, requantise and pack one sample; returns |y*gain| for the peak meter.
// y = FIR (or cascade) output, exact multiple of 2^-S for the 44.1k family.
// dst = first byte of this sample in the interleaved little-endian frame.
__device__ __forceinline__ double emit_sample(const Epilogue& ep, double y, uint32_t ch, uint64_t n, uint8_t* dst) {
    const double v = y * ep.gain;
    if (ep.bits == 32) {
        double x = v;
        if (ep.dither == 'F') {
            // Airwindows "Dither Float": noise scaled to the f32 ulp at the sample's exponent
            uint32_t u1 = (uint32_t)(rng64(ep.seed, ch, n) >> 32);
            uint32_t fb = __float_as_uint((float)x);
            int e = (int)((fb >> 23) & 0xFF);
            int expon = e ? e - 126 : 0;
            double t = ((double)u1 - 2147483647.0) * 5.5e-36;
            x = x + ldexp(t, expon + 62);
        }
        float o = (float)x;
        *reinterpret_cast<float*>(dst) = o;   // frame stride is a multiple of 4 bytes
        return fabs(v);
    }
    double x = y * ep.scale;
    double d = 0.0;
    if (ep.dither == 'T') {
        uint64_t r = rng64(ep.seed, ch, n);
        d = ((double)(uint32_t)(r >> 32) + (double)(uint32_t)r) * 0x1p-32 - 1.0;
    } else if (ep.dither == 'R') {
        uint64_t r = rng64(ep.seed, ch, n);
        d = (double)(uint32_t)(r >> 32) * 0x1p-32 - 0.5;
    }
    double q = x + d;
    double r = q >= 0.0 ? floor(q + 0.5) : ceil(q - 0.5);
    double lim = (double)(1u << (ep.bits - 1));
    r = fmin(r, lim - 1.0);
    r = fmax(r, -lim);
    int32_t iv = (int32_t)r;
    if (ep.bits == 16) {
        *reinterpret_cast<uint16_t*>(dst) = (uint16_t)iv;
    } else {
        if (ep.bits == 20) iv *= 16;
        dst[0] = (uint8_t)iv; dst[1] = (uint8_t)(iv >> 8); dst[2] = (uint8_t)(iv >> 16);
    }
    return fabs(v);
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

// d2d_kernels.hip -- gfx950 device code of the DSD->PCM engine.
//
//   d2d_fir_lut_kernel<MB>   1-bit FIR decimator by M = 8*MB through per-nibble lookup tables in LDS
//                            (the per-bit-pattern LUT path; 16-entry f64 tables are bank-conflict free:
//                            16 entries x 8 B = 32 banks, equal indices broadcast)
//   (stage B of the 48k cascade lives in d2d_kernels_rs.hip: int8 matrix cores)
//   d2d_deinterleave_kernel  byte-interleaved multichannel input -> the planar 4096-byte-block layout (one LDS pass)
//   d2d_noise_shape_kernel   the 'N' dither extension: error-feedback requantiser, one lane per (8192-output segment, channel)
//   d2d_history_kernel       carries the last `keep` bytes per channel to the next call
//   d2d_xhist_kernel         carries the last P stage-A outputs per channel to the next call
//
// What they replace: the per-block translate loop inside Rdsd2Pcm::do_conversion
// (/root/reference/src/main.rs:345,429; the rdsd2pcm crate itself is absent from the reference).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "d2d_device.h"
#include "d2d_launch.h"

namespace d2d {

template <int MB>
struct LutGeo {
    static constexpr int R = MB >= 8 ? 1 : 8 / MB;   // outputs per lane
    static constexpr int LS = MB >= 8 ? MB : 8;       // byte stride between lanes' windows
};

// Each lane walks its window as 8-byte words read from LDS, realigned in registers by the
// block-uniform byte shift; every byte gives two nibble lookups per output the lane owns.
// The taps are q*2^-S, so every partial sum is exact in f64 and the order of adds is free.
template <int MB>
__global__ __launch_bounds__(LUT_THREADS) void d2d_fir_lut_kernel(FirArgs a) {
    constexpr int R = LutGeo<MB>::R, LS = LutGeo<MB>::LS;
    extern __shared__ __align__(16) unsigned char smem[];
    double* tt = reinterpret_cast<double*>(smem);
    double* red = tt + (size_t)a.ntab * 16;                       // 4 doubles for the peak reduction
    uint8_t* win = reinterpret_cast<uint8_t*>(red + 4);
    const StreamJob job = a.jobs[blockIdx.y];
    const uint32_t tid = threadIdx.x;

    {   // tables: L2 -> LDS once per block
        const double2* src = reinterpret_cast<const double2*>(a.tables);
        double2* dst = reinterpret_cast<double2*>(tt);
        for (uint32_t i = tid; i < a.ntab * 8; i += LUT_THREADS) dst[i] = src[i];
    }
    const uint32_t tile_out = LUT_THREADS * R;
    const uint32_t ntiles = (job.nout + tile_out - 1) / tile_out;
    const uint32_t span = (uint32_t)((255 * LS + 8 * (a.nq + 1) + 16 + 15) & ~15);
    const uint32_t sample_bytes = a.epi.sample_bytes;
    const uint32_t frame_bytes = sample_bytes * a.epi.channels;
    double pk = 0.0;

    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t tile_first = job.e0 - (int64_t)a.Wb + (int64_t)tile * (LUT_THREADS * LS);
        const int64_t abeg = tile_first & ~(int64_t)15;
        const uint32_t d = (uint32_t)(tile_first - abeg);
        __syncthreads();
        stage_window(win, job, a.in_channels, a.B, a.keep, abeg, span, tid, LUT_THREADS);
        __syncthreads();

        const uint8_t* lp = win + LS * tid + 8 * (d >> 3);
        const uint32_t sh = (d & 7) * 8;
        double acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.0;
        uint64_t cur = *reinterpret_cast<const uint64_t*>(lp);
        for (uint32_t k = 0; k < a.nq; ++k) {
            const uint64_t nxt = *reinterpret_cast<const uint64_t*>(lp + 8 * (k + 1));
            const uint64_t q = sh ? (cur >> sh) | (nxt << (64 - sh)) : cur;
            const double* tb = tt + (size_t)(a.pad + 16 * k) * 16;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const uint32_t v = (uint32_t)(q >> (8 * b)) & 0xFFu;
                const uint32_t hi = v >> 4, lo = v & 15u;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double* t0 = tb + (2 * (b - r * MB)) * 16;
                    acc[r] += t0[hi];
                    acc[r] += t0[16 + lo];
                }
            }
            cur = nxt;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t nl = (tile * LUT_THREADS + tid) * R + r;
            if (nl < job.nout) {
                if (a.to_scratch) {
                    job.xs[nl] = (int32_t)ldexp(acc[r], a.scale_bits);      // exact integer, |.| < 2^31
                } else {
                    uint8_t* dst = reinterpret_cast<uint8_t*>(job.out) + (size_t)nl * frame_bytes + job.och * sample_bytes;
                    pk = fmax(pk, emit_sample(a.epi, job, acc[r], job.n0 + nl, dst));
                }
            }
        }
    }
    if (!a.to_scratch) block_peak_max(pk, job.peak, red);
}

// Byte-interleaved input (DFF, `-f I`: c0 c1 c0 c1 ...) -> the planar 4096-byte-block layout the FIR
// kernels stream with 16-byte loads.  A block moves DI_TILE bytes per channel: the source range is
// contiguous (coalesced 16-byte loads into LDS), each thread then gathers one channel's 16 bytes from
// LDS and stores them as one 16-byte word (a wave writes 1 KiB contiguous per channel).  In LDS every
// group of 16 frames (16*C bytes) is followed by one pad dword: the threads of a wave gather from
// consecutive groups, 4C+1 dwords apart -- an odd stride, so their byte reads fall on distinct banks.
// The last (short) block group keeps the engine's layout rule: channels packed back to back with the
// short length.
constexpr uint32_t DI_TILE = 1024;
constexpr uint32_t DI_BLOCK = 4096;
__host__ __device__ constexpr uint32_t di_lds_bytes(uint32_t C) { return DI_TILE * C + (DI_TILE / 16) * 4; }

// (C = channels of the file; spf = streams per file in the job table: the channels the engine converts,
// jobs[file * spf].ch the first of them -- only those are written to the planar copy)
__global__ __launch_bounds__(256) void d2d_deinterleave_kernel(const StreamJob* jobs, uint32_t C, uint32_t spf) {
    extern __shared__ __align__(16) unsigned char smem[];
    const StreamJob job = jobs[blockIdx.y * spf];
    const uint32_t L = (uint32_t)job.L;
    const D2D_GLOBAL uint8_t* src = as_global(job.in_raw);
    D2D_GLOBAL uint8_t* dst = as_global(const_cast<uint8_t*>(job.in));
    const uint32_t ntiles = (L + DI_TILE - 1) / DI_TILE;
    const uint32_t gpitch = 16 * C + 4;                               // LDS bytes per group of 16 frames
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t j0 = tile * DI_TILE;
        const uint32_t nj = min(DI_TILE, L - j0);                 // bytes per channel in this tile
        const uint32_t nbytes = nj * C;
        const uint64_t s0 = (uint64_t)j0 * C;                     // a multiple of 16 (DI_TILE is)
        __syncthreads();
        for (uint32_t i = threadIdx.x * 16; i < nbytes; i += 256 * 16) {
            uint8_t* d = smem + i + 4 * (i / (16 * C));           // a 16-byte chunk never straddles a pad
            if (i + 16 <= nbytes && ((uintptr_t)(src + s0 + i) & 15) == 0) {
                const u32x4 v = *reinterpret_cast<const D2D_GLOBAL u32x4*>(src + s0 + i);
                uint32_t* dw = reinterpret_cast<uint32_t*>(d);
                dw[0] = v.x; dw[1] = v.y; dw[2] = v.z; dw[3] = v.w;
            } else {
                for (uint32_t b = i; b < min(i + 16, nbytes); ++b) d[b - i] = src[s0 + b];
            }
        }
        __syncthreads();
        const uint32_t blk = j0 / DI_BLOCK, off0 = j0 - blk * DI_BLOCK;   // DI_TILE divides DI_BLOCK
        const uint32_t blen = min(DI_BLOCK, L - blk * DI_BLOCK);
        const uint32_t nq = (nj + 15) / 16;
        for (uint32_t t = threadIdx.x; t < nq * spf; t += 256) {
            const uint32_t cl = t / nq, q = t - cl * nq, c = job.ch + cl;
            const uint32_t n = min(16u, nj - q * 16);
            const uint8_t* g = smem + q * gpitch + c;
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t b = 0; b < 16; ++b)
                if (b < n) w[b >> 2] |= (uint32_t)g[b * C] << (8 * (b & 3));
            const uint64_t d0 = (uint64_t)blk * DI_BLOCK * C + (uint64_t)c * blen + off0 + q * 16;
            if (n == 16 && ((uintptr_t)(dst + d0) & 15) == 0) {
                *reinterpret_cast<D2D_GLOBAL u32x4*>(dst + d0) = u32x4{w[0], w[1], w[2], w[3]};
            } else {
                for (uint32_t b = 0; b < n; ++b) dst[d0 + b] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
            }
        }
    }
}

// 'N' dither (an extension, see dsd2dxd_amd.h): TPDF dither inside the error-feedback loop
//   w = x - (2*e1 - e2);  r = round_half_away(w + d);  e = r - w;  e2 = e1;  e1 = e
// in exactly the oracle's operation order (emit_sample() in oracle/d2d_oracle.c).  The loop is a
// recurrence through a rounding, so it cannot be reassociated; by definition it restarts from
// e1 = e2 = 0 at every output index that is a multiple of NS_SEG = 8192 (0.093 s at 88.2 kHz), which is what
// makes segments independent.  One LANE walks one (segment, channel); the lanes of a file's channels sit
// side by side and share an LDS row per segment, in which eight whole frames are assembled and then
// stored with 16-byte stores by the segment's first lane (a lane storing its own three bytes per frame
// would touch 64 cache lines per store instruction: that, not the arithmetic, bounded the first version).
//   blockIdx.y = file; a wave holds 64/Cp segments (Cp = channels rounded up to a power of two).
constexpr uint32_t NS_SEG_BITS = 13;
constexpr uint32_t NS_FRAMES = 8;          // frames assembled per flush

__global__ __launch_bounds__(256) void d2d_noise_shape_kernel(NoiseShapeArgs a) {
    extern __shared__ __align__(16) unsigned char ns_smem[];
    const uint32_t C = a.epi.channels, sb = a.epi.sample_bytes, fb = sb * C;
    const uint32_t cp_bits = a.cp_bits, Cp = 1u << cp_bits;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t ch = lane & (Cp - 1), sw = lane >> cp_bits;            // channel, segment slot inside the wave
    const uint32_t spw = 64u >> cp_bits;                                  // segments per wave
    const uint32_t file = blockIdx.y;
    const StreamJob* jobs = a.jobs + (size_t)file * C;
    const StreamJob j0 = jobs[0];                                          // n0, nout, out are common to a file's channels
    // the index the pass runs on: the FIR outputs n (44.1k family) or stage B's outputs m (48k family)
    const uint64_t N0 = a.res ? j0.m0 : j0.n0;
    const uint32_t NOUT = a.res ? j0.nres : j0.nout;
    if (NOUT == 0) return;
    const uint64_t n_end = N0 + NOUT;
    const uint64_t k0 = N0 >> NS_SEG_BITS, k1 = (n_end - 1) >> NS_SEG_BITS;
    const uint64_t k = k0 + (uint64_t)(blockIdx.x * (blockDim.x >> 6) + wave) * spw + sw;
    const bool active = k <= k1 && ch < C;
    const uint32_t rb = NS_FRAMES * fb;                                    // row bytes (a multiple of 8)
    const uint32_t rstride = rb + 4;                                       // + one pad dword: rows on distinct banks
    uint8_t* row = ns_smem + (size_t)(wave * spw + sw) * rstride;
    // (every lane of a wave runs the same number of loop trips; inactive ones do nothing inside)
    const uint64_t seg_lo = k << NS_SEG_BITS, seg_hi = (k + 1) << NS_SEG_BITS;
    const uint32_t i0 = active ? (seg_lo > N0 ? (uint32_t)(seg_lo - N0) : 0u) : 0u;
    const uint32_t i1 = active ? (uint32_t)((seg_hi < n_end ? seg_hi : n_end) - N0) : 0u;
    const StreamJob job = jobs[ch < C ? ch : 0];
    const D2D_GLOBAL int32_t* xs = as_global(job.xs);
    const uint32_t sidx = file * C + (ch < C ? ch : 0);
    const D2D_GLOBAL double* ys = as_global(a.ys) + (size_t)sidx * a.ys_stride;      // (res only)
    const bool carried = active && seg_lo < N0;                         // begun in an earlier call
    double e1 = carried ? a.state[2 * sidx] : 0.0, e2 = carried ? a.state[2 * sidx + 1] : 0.0, pk = 0.0;
    const double lim = (double)(1u << (a.epi.bits - 1));
    const double xscale = ldexp(a.epi.scale, -a.scale_bits);
    uint32_t vmax = 0;                                                     // max |y * 2^S|: the peak, scaled back at the end
    uint8_t* gout = reinterpret_cast<uint8_t*>(j0.out);
    // dword stores need the segment's first byte on a dword boundary: true for a carried segment (the call's
    // buffer is 16-byte aligned) and for every later one when (first index of the segment) * fb is a multiple of 4
    const bool dw_ok = ((size_t)i0 * fb & 3u) == 0;
    // longest run of steps any lane of the wave has
    uint32_t len = i1 - i0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) len = max(len, (uint32_t)__shfl_xor((int)len, o));
    for (uint32_t t0 = 0; t0 < len; t0 += NS_FRAMES) {
        // this lane's integers of the next eight steps (two 16-byte loads: 4-byte aligned is enough)
        int32_t v[NS_FRAMES];
        double yv[NS_FRAMES];
        const uint32_t ib = i0 + t0;
        if (a.res) {
#pragma unroll
            for (uint32_t u = 0; u < NS_FRAMES; ++u) { v[u] = 0; yv[u] = (active && ib + u < i1) ? ys[ib + u] : 0.0; }
        } else if (active && ib + NS_FRAMES <= i1) {
            typedef int32_t i32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
            const i32x4_a4 lo4 = *reinterpret_cast<D2D_GLOBAL const i32x4_a4*>(xs + ib), hi4 = *reinterpret_cast<D2D_GLOBAL const i32x4_a4*>(xs + ib + 4);
            v[0] = lo4.x; v[1] = lo4.y; v[2] = lo4.z; v[3] = lo4.w; v[4] = hi4.x; v[5] = hi4.y; v[6] = hi4.z; v[7] = hi4.w;
        } else {
#pragma unroll
            for (uint32_t u = 0; u < NS_FRAMES; ++u) v[u] = (active && ib + u < i1) ? xs[ib + u] : 0;
        }
#pragma unroll
        for (uint32_t u = 0; u < NS_FRAMES; ++u) {
            const uint32_t i = ib + u;
            if (active && i < i1) {
                // y = v * 2^-S is exact and x = y * scale rounds once: v * (scale * 2^-S) is the same product, rounded the same
                vmax = max(vmax, (uint32_t)(v[u] < 0 ? -v[u] : v[u]));
                const double x = a.res ? yv[u] * a.epi.scale : (double)v[u] * xscale;      // (res: y is the resampler's f64, x = y * scale rounds once)
                const uint32_t rnd = rng32(job, N0 + i);
                const double d = (double)((rnd & 0xFFFFu) + (rnd >> 16) + 1u) * 0x1p-16 - 1.0;
                const double fbk = 2.0 * e1 - e2;
                const double w = x - fbk;
                const double q = w + d;
                const double r = trunc(q + copysign(0.5, q));
                e2 = e1;
                e1 = r - w;
                int32_t iv = (int32_t)fmax(fmin(r, lim - 1.0), -lim);
                if (a.epi.bits == 20) iv *= 16;
                uint8_t* dst = row + u * fb + ch * sb;
                if (sb == 2) { *reinterpret_cast<uint16_t*>(dst) = (uint16_t)iv; }
                else { dst[0] = (uint8_t)iv; dst[1] = (uint8_t)(iv >> 8); dst[2] = (uint8_t)(iv >> 16); }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the segment's first lane sends the assembled frames on their way
        if (active && ch == 0 && ib < i1) {
            const uint32_t nfr = min(NS_FRAMES, i1 - ib);
            uint8_t* g = gout + (size_t)ib * fb;
            const uint32_t nb = nfr * fb;
            if (dw_ok && nfr == NS_FRAMES) {
                uint32_t o = 0;
                typedef uint32_t u32x4_dw __attribute__((ext_vector_type(4), aligned(4)));      // dw_ok: dword-aligned, no more
                for (; o + 16 <= nb; o += 16)
                    *reinterpret_cast<D2D_GLOBAL u32x4_dw*>(as_global(g + o)) =
                        u32x4_dw{*reinterpret_cast<uint32_t*>(row + o), *reinterpret_cast<uint32_t*>(row + o + 4),
                              *reinterpret_cast<uint32_t*>(row + o + 8), *reinterpret_cast<uint32_t*>(row + o + 12)};
                for (; o < nb; o += 4) *reinterpret_cast<D2D_GLOBAL uint32_t*>(as_global(g + o)) = *reinterpret_cast<uint32_t*>(row + o);
            } else {
                for (uint32_t o = 0; o < nb; ++o) as_global(g)[o] = row[o];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // the open segment's state travels on -- into the OTHER state buffer: the lane that reads a stream's carried state and
    // the lane that writes its new one may sit in different blocks
    if (active && k == k1) { a.state_next[2 * sidx] = e1; a.state_next[2 * sidx + 1] = e2; }
    pk = fabs(ldexp((double)vmax, -a.scale_bits) * a.epi.gain);            // |y * gain| is monotonic in |y|
    if (active && pk > 0.0 && !a.res) atomicMax(reinterpret_cast<unsigned long long*>(job.peak), (unsigned long long)__double_as_longlong(pk));
}


// The same pass for STEREO frames of 16 or 24 bits (every configuration the bench names), built for the fact that it is bound by
// the instruction stream of single waves (a 64-file batch has 83 k chains: 1.3 waves per SIMD; PMC: 42 vector + 12 scalar + 4 LDS
// + 4 branch instructions per step in the general kernel above).  No LDS and no wave-level fences: the two channel lanes of a
// segment sit side by side, the left lane takes the right one's sample with one DPP move, packs whole frames with v_perm_b32 and
// stores a group's 48 (32) bytes itself -- one trip LATER, after the next group's integers have been requested, so that the wait
// for those never sits behind a fresh store.  Whole groups run without a test per step.  INTQ (unit gain): the recurrence in
// int32 -- x = v * 2^-F LSB, the dither has 16 fraction bits and the errors F, so every quantity of the loop is an exact
// fixed-point number and the f64 operations of the definition never round: same results.
template <int SB, bool INTQ>
__global__ __launch_bounds__(256) void d2d_noise_shape_stereo_kernel(NoiseShapeArgs a) {
    constexpr uint32_t FBY = 2u * SB;                                      // bytes per frame
    constexpr int NW = NS_FRAMES * FBY / 4;                                // dwords per group of eight frames: 12 (24-bit) or 8 (16-bit)
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t ch = lane & 1u, sw = lane >> 1;                         // channel, segment slot inside the wave
    const uint32_t file = blockIdx.y;
    const StreamJob* jobs = a.jobs + (size_t)file * 2;
    const StreamJob j0 = jobs[0];                                          // n0, nout, out are common to a file's channels
    if (j0.nout == 0) return;
    const uint64_t n_end = j0.n0 + j0.nout;
    const uint64_t k0 = j0.n0 >> NS_SEG_BITS, k1 = (n_end - 1) >> NS_SEG_BITS;
    const uint64_t k = k0 + (uint64_t)(blockIdx.x * (blockDim.x >> 6) + wave) * 32u + sw;
    const bool active = k <= k1;
    const uint64_t seg_lo = k << NS_SEG_BITS, seg_hi = (k + 1) << NS_SEG_BITS;
    const uint32_t i0 = active ? (seg_lo > j0.n0 ? (uint32_t)(seg_lo - j0.n0) : 0u) : 0u;
    const uint32_t i1 = active ? (uint32_t)((seg_hi < n_end ? seg_hi : n_end) - j0.n0) : 0u;
    const StreamJob job = jobs[ch];
    const D2D_GLOBAL int32_t* xs = as_global(job.xs);
    const uint32_t sidx = file * 2 + ch;
    const bool carried = active && seg_lo < j0.n0;                         // begun in an earlier call
    double e1 = carried ? a.state[2 * sidx] : 0.0, e2 = carried ? a.state[2 * sidx + 1] : 0.0;
    const int F = a.scale_bits - ((int)a.epi.bits - 1);                    // INTQ: fraction bits of x, 1..16
    const uint32_t kSh = 16u - (uint32_t)(INTQ ? F : 8);
    int32_t E1 = INTQ ? (int32_t)ldexp(e1, F) : 0, E2 = INTQ ? (int32_t)ldexp(e2, F) : 0;      // the errors scaled by 2^F (exact)
    const int32_t qmin_i = -(1 << (a.epi.bits - 1)), qmax_i = (1 << (a.epi.bits - 1)) - 1;
    const double lim = (double)(1u << (a.epi.bits - 1));
    const double xscale = ldexp(a.epi.scale, -a.scale_bits);
    uint32_t vmax = 0;                                                     // max |y * 2^S|: the peak, scaled back at the end
    uint8_t* gout = reinterpret_cast<uint8_t*>(j0.out);
    // the dither counter: lo32(n0 + i) + key, + kstep past the wrap of lo32 (at most once per call; tested per group)
    const uint32_t zbase = (uint32_t)job.n0 + job.rng_key;
    // Whole groups of eight frames first, in a loop whose memory operations are the same on every trip (two stores, two loads, no
    // test in front of any of them): the compiler can then count how many younger requests may stay in flight when a group's integers
    // are needed, and these are requested TWO groups ahead.  A lane whose segment is shorter than the longest of the wave (the call's
    // first and last segments) keeps requesting its own last group and re-storing its own last frames -- harmless -- while the
    // arithmetic is switched off for it.
    const uint32_t my_ngrp = active ? (i1 - i0) / NS_FRAMES : 0u;
    uint32_t ngrp_max = my_ngrp;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ngrp_max = max(ngrp_max, (uint32_t)__shfl_xor((int)ngrp_max, o));
    ngrp_max = __builtin_amdgcn_readfirstlane(ngrp_max);

    // eight steps of the loop on v[]; `checked`: only the steps below i1 (the segment's last, partial group)
    auto steps8 = [&](uint32_t ib, const int32_t (&v)[NS_FRAMES], int32_t (&r8)[NS_FRAMES], auto checked) {
        // this group's dither counter; the general form where lo32 wraps inside the group
        const uint32_t lo = (uint32_t)job.n0 + ib;
        const bool plain = lo + NS_FRAMES >= lo && (lo >= job.rng_lo0) == (lo + NS_FRAMES - 1 >= job.rng_lo0);
        const uint32_t zg = zbase + ib + (lo < job.rng_lo0 ? job.rng_kstep : 0u);
        auto step = [&](uint32_t u) {
            vmax = max(vmax, (uint32_t)(v[u] < 0 ? -v[u] : v[u]));
            uint32_t z = zg + u;
            if (!plain) { const uint32_t l = lo + u; z = l + job.rng_key + (l < job.rng_lo0 ? job.rng_kstep : 0u); }
            z ^= z >> 16; z *= 0x7feb352dU;
            z ^= z >> 15; z *= 0x846ca68bU;
            z ^= z >> 16;
            int32_t iv;
            if constexpr (INTQ) {
                // w = x - (2 e1 - e2), q = w + d, r = round half away (q), e = r - w: W, E in units of 2^-F LSB, T in 2^-16
                const int32_t W = v[u] - (2 * E1 - E2);
                const int32_t T = (int32_t)((z & 0xFFFFu) + (z >> 16)) - 32767;             // (d + 1/2) * 2^16
                int32_t r = (W + (T >> kSh)) >> F;                                          // floor(q + 1/2)
                const bool tie = ((((uint32_t)W << kSh) + (uint32_t)T) & 0xFFFFu) == 0;    // q + 1/2 an integer: then q = r - 1/2
                r -= (tie && r <= 0) ? 1 : 0;                                               // ... and a negative q rounds away from zero
                E2 = E1;
                E1 = (int32_t)((uint32_t)r << F) - W;
                iv = min(max(r, qmin_i), qmax_i);
            } else {
                // y = v * 2^-S is exact and x = y * scale rounds once: v * (scale * 2^-S) is the same product, rounded the same
                const double x = (double)v[u] * xscale;
                const double d = (double)((z & 0xFFFFu) + (z >> 16) + 1u) * 0x1p-16 - 1.0;
                const double fbk = 2.0 * e1 - e2;
                const double w = x - fbk;
                const double q = w + d;
                const double r = trunc(q + copysign(0.5, q));
                e2 = e1;
                e1 = r - w;
                iv = (int32_t)fmax(fmin(r, lim - 1.0), -lim);
            }
            r8[u] = iv;
        };
#pragma unroll
        for (uint32_t u = 0; u < NS_FRAMES; ++u) {
            if constexpr (decltype(checked)::value) { r8[u] = 0; if (ib + u < i1) step(u); } else step(u);
        }
    };

    typedef int32_t i32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
    // (the frames' byte offset inside a file is a multiple of the frame size only: 2-byte alignment is all these 16-byte stores may claim)
    typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(2)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    constexpr int NH = NW / 2;                                             // each lane of the pair packs half of the group: frames 0-3 (left lane) or 4-7
    // Rounds of four groups = one 128-byte line of a lane's integers, requested a round ahead and consumed from registers (a line is
    // fetched once and need not survive in L2 between trips).  The round's frames go through LDS -- a row per segment -- and leave
    // in 16-byte pieces that are consecutive along each segment: a store instruction then covers whole lines of five or six
    // segments instead of 24 bytes in each of 64 lines (the scattered form cost 0.55 ms of the pass's 2.15).  Every memory
    // operation of the loop is unconditional, so the compiler counts its waits: a lane whose segment is shorter keeps requesting
    // its last group (arithmetic switched off by exec), and a piece that belongs to no group of its segment goes to a dump line.
    constexpr uint32_t NR = 4;                                             // groups per round
    constexpr uint32_t CPG = NW / 4;                                       // 16-byte pieces per group: 3 (24-bit) or 2 (16-bit)
    constexpr uint32_t CPR = CPG * NR;                                     // ... per segment and round
    constexpr uint32_t ROW = CPR * 16 + 16;                                // LDS row pitch (one spare piece: rows start on different banks)
    constexpr uint32_t NI = 32 * CPR / 64;                                 // store instructions per round
    __shared__ __align__(16) unsigned char stage_all[4][32 * ROW];
    unsigned char* stage = stage_all[wave];
    if (ngrp_max) {
        const uint32_t g_last = my_ngrp ? my_ngrp - 1u : 0u;
        const uint32_t ld_max = j0.nout - NS_FRAMES;                       // (some lane of the wave has a whole group: nout >= 8)
        auto load8 = [&](uint32_t g, int32_t (&v)[NS_FRAMES]) {           // the lane's group min(g, last): always inside the stream
            const D2D_GLOBAL int32_t* p = xs + min(i0 + NS_FRAMES * min(g, g_last), ld_max);
            const i32x4_a4 lo4 = *reinterpret_cast<D2D_GLOBAL const i32x4_a4*>(p), hi4 = *reinterpret_cast<D2D_GLOBAL const i32x4_a4*>(p + 4);
            v[0] = lo4.x; v[1] = lo4.y; v[2] = lo4.z; v[3] = lo4.w; v[4] = hi4.x; v[5] = hi4.y; v[6] = hi4.z; v[7] = hi4.w;
        };
        // piece c = 64 i + lane of a round: segment slot c / CPR, piece c % CPR of that segment's CPR * 16 bytes
        uint32_t pc_lds[NI], pc_grp[NI], pc_ngrp[NI];
        uint64_t pc_out[NI];
#pragma unroll
        for (uint32_t i = 0; i < NI; ++i) {
            const uint32_t c = 64u * i + lane, sl = c / CPR, w = c - sl * CPR;
            pc_lds[i] = sl * ROW + w * 16u;
            pc_grp[i] = w / CPG;
            pc_ngrp[i] = (uint32_t)__shfl((int)my_ngrp, (int)(2u * sl));
            const uint32_t i0s = (uint32_t)__shfl((int)i0, (int)(2u * sl));
            pc_out[i] = (uint64_t)(uintptr_t)gout + (uint64_t)i0s * FBY + w * 16u;
        }
        const uint64_t dump = (uint64_t)(uintptr_t)a.dump + lane * 16u;
        auto process = [&](uint32_t g, uint32_t j, const int32_t (&v)[NS_FRAMES]) {
            const uint32_t ib = i0 + NS_FRAMES * g;
            int32_t r8[NS_FRAMES];
            steps8(ib, v, r8, std::false_type{});
            // the pair's lanes swap samples (quad_perm [1,0,3,2]); frame j of this lane's half: A = left channel, B = right channel
            uint32_t A[4], B[4];
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                const uint32_t nlo = (uint32_t)__builtin_amdgcn_mov_dpp(r8[q], 0xB1, 0xF, 0xF, true);
                const uint32_t nhi = (uint32_t)__builtin_amdgcn_mov_dpp(r8[4 + q], 0xB1, 0xF, 0xF, true);
                A[q] = ch ? nhi : (uint32_t)r8[q];
                B[q] = ch ? (uint32_t)r8[4 + q] : nlo;
            }
            unsigned char* row = stage + sw * ROW + j * (16u * CPG) + ch * (4u * NH);
            if constexpr (SB == 3) {
                // frames k, k+1 -> 12 bytes: [L0 L1 L2 R0 | R1 R2 L0' L1' | L2' R0' R1' R2']
                u32x2* d = reinterpret_cast<u32x2*>(row);                  // 8-byte aligned
                d[0] = u32x2{__builtin_amdgcn_perm(B[0], A[0], 0x04020100u), __builtin_amdgcn_perm(A[1], B[0], 0x05040201u)};
                d[1] = u32x2{__builtin_amdgcn_perm(B[1], A[1], 0x06050402u), __builtin_amdgcn_perm(B[2], A[2], 0x04020100u)};
                d[2] = u32x2{__builtin_amdgcn_perm(A[3], B[2], 0x05040201u), __builtin_amdgcn_perm(B[3], A[3], 0x06050402u)};
            } else {
                *reinterpret_cast<u32x4*>(row) = u32x4{__builtin_amdgcn_perm(B[0], A[0], 0x05040100u), __builtin_amdgcn_perm(B[1], A[1], 0x05040100u),
                                                       __builtin_amdgcn_perm(B[2], A[2], 0x05040100u), __builtin_amdgcn_perm(B[3], A[3], 0x05040100u)};
            }
        };
        auto wave_sync = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
        int32_t va[NR][NS_FRAMES], vb[NR][NS_FRAMES];
#pragma unroll
        for (uint32_t j = 0; j < NR; ++j) load8(j, va[j]);
        auto round = [&](uint32_t g0, const int32_t (&cur)[NR][NS_FRAMES], int32_t (&nxt)[NR][NS_FRAMES]) {
#pragma unroll
            for (uint32_t j = 0; j < NR; ++j) load8(g0 + NR + j, nxt[j]);
#pragma unroll
            for (uint32_t j = 0; j < NR; ++j)
                if (g0 + j < my_ngrp) process(g0 + j, j, cur[j]);
            wave_sync();
            const uint64_t rofs = (uint64_t)g0 * (NS_FRAMES * FBY);        // the round's first byte inside every segment
#pragma unroll
            for (uint32_t i = 0; i < NI; ++i) {
                const u32x4 piece = *reinterpret_cast<const u32x4*>(stage + pc_lds[i]);
                const uint64_t dst = g0 + pc_grp[i] < pc_ngrp[i] ? pc_out[i] + rofs : dump;
                *reinterpret_cast<D2D_GLOBAL u32x4_a4*>(as_global(reinterpret_cast<uint8_t*>((uintptr_t)dst))) = u32x4_a4{piece.x, piece.y, piece.z, piece.w};
            }
            wave_sync();
        };
        uint32_t g0 = 0;
        for (; g0 + NR < ngrp_max; g0 += 2 * NR) { round(g0, va, vb); round(g0 + NR, vb, va); }
        if (g0 < ngrp_max) round(g0, va, vb);
    }
    // the segment's last, partial group (the call ends inside it): step by step, the left lane stores what exists two bytes at a time
    if (active && ((i1 - i0) % NS_FRAMES)) {
        const uint32_t ib = i0 + NS_FRAMES * my_ngrp;
        int32_t v[NS_FRAMES], r8[NS_FRAMES];
#pragma unroll
        for (uint32_t u = 0; u < NS_FRAMES; ++u) v[u] = ib + u < i1 ? xs[ib + u] : 0;
        steps8(ib, v, r8, std::true_type{});
        uint32_t pk[NW];
        uint32_t L[NS_FRAMES], R[NS_FRAMES];
#pragma unroll
        for (uint32_t u = 0; u < NS_FRAMES; ++u) {
            L[u] = (uint32_t)r8[u];
            R[u] = (uint32_t)__builtin_amdgcn_mov_dpp(r8[u], 0xF5, 0xF, 0xF, true);     // quad_perm [1,1,3,3]: every even lane reads its right neighbour
        }
        if constexpr (SB == 3) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const uint32_t La = L[4 * g], Ra = R[4 * g], Lb = L[4 * g + 1], Rb = R[4 * g + 1];
                const uint32_t Lc = L[4 * g + 2], Rc = R[4 * g + 2], Ld = L[4 * g + 3], Rd = R[4 * g + 3];
                pk[6 * g + 0] = __builtin_amdgcn_perm(Ra, La, 0x04020100u); pk[6 * g + 1] = __builtin_amdgcn_perm(Lb, Ra, 0x05040201u);
                pk[6 * g + 2] = __builtin_amdgcn_perm(Rb, Lb, 0x06050402u); pk[6 * g + 3] = __builtin_amdgcn_perm(Rc, Lc, 0x04020100u);
                pk[6 * g + 4] = __builtin_amdgcn_perm(Ld, Rc, 0x05040201u); pk[6 * g + 5] = __builtin_amdgcn_perm(Rd, Ld, 0x06050402u);
            }
        } else {
#pragma unroll
            for (uint32_t u = 0; u < NS_FRAMES; ++u) pk[u] = __builtin_amdgcn_perm(R[u], L[u], 0x05040100u);
        }
        if (ch == 0) {
            uint8_t* g = gout + (size_t)ib * FBY;
            for (uint32_t b = 0; b < (i1 - ib) * FBY; b += 2) {
                uint32_t wv = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) wv = (b >> 2) == (uint32_t)w ? pk[w] : wv;
                *reinterpret_cast<D2D_GLOBAL uint16_t*>(as_global(g + b)) = (uint16_t)(wv >> (8 * (b & 2)));
            }
        }
    }
    // the open segment's state travels on -- into the OTHER state buffer: the lane that reads a stream's carried state and
    // the lane that writes its new one may sit in different blocks
    if (INTQ) { e1 = ldexp((double)E1, -F); e2 = ldexp((double)E2, -F); }
    if (active && k == k1) { a.state_next[2 * sidx] = e1; a.state_next[2 * sidx + 1] = e2; }
    const double pkv = fabs(ldexp((double)vmax, -a.scale_bits) * a.epi.gain);          // |y * gain| is monotonic in |y|
    if (active && pkv > 0.0) atomicMax(reinterpret_cast<unsigned long long*>(job.peak), (unsigned long long)__double_as_longlong(pkv));
}

// new_hist[j] = stream byte (L - keep + j), j in [0, keep)
__global__ void d2d_history_kernel(const StreamJob* jobs, uint32_t C, uint32_t B, uint32_t keep) {
    const StreamJob job = jobs[blockIdx.x];
    for (uint32_t j = threadIdx.x; j < keep; j += blockDim.x)
        job.hist_next[j] = (uint8_t)stream_byte(job, C, B, keep, (int32_t)job.L - (int32_t)keep + (int32_t)j);
}

// scratch layout per stream: [P history][nout new]; move the last P to the front (regions may overlap)
__global__ void d2d_xhist_kernel(const StreamJob* jobs, uint32_t P) {
    const StreamJob job = jobs[blockIdx.x];
    int32_t* s = job.xs - P;
    int32_t v = 0;
    if (threadIdx.x < P) v = s[job.nout + threadIdx.x];
    __syncthreads();
    if (threadIdx.x < P) s[threadIdx.x] = v;
}

// ---- launchers -------------------------------------------------------------------------------

size_t lut_smem_bytes(const FirArgs& a, int MB) {
    const int LS = MB >= 8 ? MB : 8;
    const size_t span = (size_t)((255 * LS + 8 * (a.nq + 1) + 16 + 15) & ~15);
    return (size_t)a.ntab * 128 + 4 * sizeof(double) + span;
}

template <int MB>
static hipError_t launch_lut_t(const FirArgs& a, dim3 grid, hipStream_t s) {
    const size_t smem = lut_smem_bytes(a, MB);
    static KernelPrep prep;
    hipError_t e = prep.max_dynamic_lds(reinterpret_cast<const void*>(&d2d_fir_lut_kernel<MB>), 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(d2d_fir_lut_kernel<MB>, grid, dim3(LUT_THREADS), smem, s, a);
    d2d_last_launched_kernel = launched_name<MB>("d2d_fir_lut_kernel");
    return hipGetLastError();
}

hipError_t launch_fir_lut(const FirArgs& a, int MB, uint32_t max_tiles, uint32_t nstreams, hipStream_t s) {
    if (nstreams == 0 || max_tiles == 0) return hipSuccess;
    // enough blocks to fill 256 CUs a few times over, the rest by the in-kernel tile loop
    uint32_t gx = max_tiles;
    const uint32_t cap = (4096 + nstreams - 1) / nstreams;
    if (gx > cap) gx = cap < 1 ? 1 : cap;
    dim3 grid(gx, nstreams);
    switch (MB) {
        case 1: return launch_lut_t<1>(a, grid, s);
        case 2: return launch_lut_t<2>(a, grid, s);
        case 4: return launch_lut_t<4>(a, grid, s);
        case 8: return launch_lut_t<8>(a, grid, s);
        case 16: return launch_lut_t<16>(a, grid, s);
        default: return hipErrorInvalidValue;
    }
}

uint32_t lut_outputs_per_tile(int MB) { return LUT_THREADS * (MB >= 8 ? 1 : 8 / MB); }

const char* lut_kernel_name(int MB) {
    switch (MB) {
        case 1: return "d2d_fir_lut_kernel<1>";
        case 2: return "d2d_fir_lut_kernel<2>";
        case 4: return "d2d_fir_lut_kernel<4>";
        case 8: return "d2d_fir_lut_kernel<8>";
        default: return "d2d_fir_lut_kernel<16>";
    }
}

hipError_t launch_deinterleave(const StreamJob* jobs, uint32_t nfiles, uint32_t C, uint32_t spf, uint32_t max_L, hipStream_t s) {
    if (nfiles == 0 || max_L == 0) return hipSuccess;
    uint32_t gx = (max_L + DI_TILE - 1) / DI_TILE;
    const uint32_t cap = (8192 + nfiles - 1) / nfiles;
    if (gx > cap) gx = cap;
    static KernelPrep prep;                                            // 64 channels need 66 KB of LDS
    hipError_t e = prep.max_dynamic_lds(reinterpret_cast<const void*>(&d2d_deinterleave_kernel), 80 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(d2d_deinterleave_kernel, dim3(gx, nfiles), dim3(256), (size_t)di_lds_bytes(C), s, jobs, C, spf);
    return hipGetLastError();
}

hipError_t launch_noise_shape(const NoiseShapeArgs& a0, hipStream_t s) {
    if (a0.nstreams == 0) return hipSuccess;
    NoiseShapeArgs a = a0;
    const uint32_t C = a.epi.channels;
    uint32_t cpb = 0;
    while ((1u << cpb) < C) ++cpb;
    if (cpb > 6) return hipErrorInvalidValue;                                  // at most 64 channels (d2d_create's limit)
    a.cp_bits = cpb;
    const uint32_t spw = 64u >> cpb;                                           // segments per wave
    const uint32_t max_seg = (a.max_nout >> NS_SEG_BITS) + 2;                 // segments one call can touch
    const uint32_t waves = 4;                                                  // per block
    const uint32_t fb = a.epi.sample_bytes * C;
    const size_t smem = (size_t)waves * spw * (NS_FRAMES * fb + 4);
    const uint32_t nfiles = a.nstreams / C;
    if (C == 2 && (a.epi.bits == 16 || a.epi.bits == 24) && !a.general && !a.res) {      // (general: D2D_DBG_NS_GENERAL, the general kernel for stereo too)
        const int F = a.scale_bits - ((int)a.epi.bits - 1);
        const bool intq = a.intq && a.epi.gain == 1.0 && F >= 1 && F <= 16;
        const dim3 grid((max_seg + waves * 32 - 1) / (waves * 32), nfiles);
        if (a.epi.bits == 24) {
            if (intq) hipLaunchKernelGGL((d2d_noise_shape_stereo_kernel<3, true>), grid, dim3(64 * waves), 0, s, a);
            else hipLaunchKernelGGL((d2d_noise_shape_stereo_kernel<3, false>), grid, dim3(64 * waves), 0, s, a);
        } else {
            if (intq) hipLaunchKernelGGL((d2d_noise_shape_stereo_kernel<2, true>), grid, dim3(64 * waves), 0, s, a);
            else hipLaunchKernelGGL((d2d_noise_shape_stereo_kernel<2, false>), grid, dim3(64 * waves), 0, s, a);
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL(d2d_noise_shape_kernel, dim3((max_seg + waves * spw - 1) / (waves * spw), nfiles), dim3(64 * waves), smem, s, a);
    return hipGetLastError();
}

// ---- tap_bits = 32 (d2d_params): the FIR ran twice into the scratch, with the 24-bit table q and with the residual table q32 - 256 q;
// v = sum q32 s = 256 v_hi + v_lo is an exact integer below 2^40, y = v * 2^-(S+8) exactly, and from there the f64 epilogue every
// other path uses (oracle: orc_use_fine_taps).  A block takes 256 frames of one file: a thread finishes its frame's samples channel by
// channel (the scratch reads of a channel are consecutive across the block), the frames meet in LDS and leave as 16-byte pieces.
constexpr uint32_t FC_FRAMES = 256;
template <int K, int CMAX>        // K chunks of 256 frames per trip (their scratch reads are all requested before the first is used), at most CMAX channels
__global__ __launch_bounds__(FC_FRAMES) void d2d_fine_combine_kernel(const StreamJob* jobs, size_t lo_off, long long lo_bias, int sbits, Epilogue epi) {
    extern __shared__ __align__(16) uint8_t fc_lds[];
    __shared__ unsigned long long pkl[64];                               // per channel: the block's peak (non-negative doubles order like their bits)
    const uint32_t C = epi.channels, sb = epi.sample_bytes, fb = C * sb;
    const StreamJob* fj = jobs + (size_t)blockIdx.y * C;
    const uint32_t nout = fj[0].nout;
    if (threadIdx.x < 64) pkl[threadIdx.x] = 0ull;
    __syncthreads();
    const double ys = ldexp(1.0, -sbits);
    constexpr uint32_t SPAN = FC_FRAMES * K;
    for (uint32_t n0 = blockIdx.x * SPAN; n0 < nout; n0 += gridDim.x * SPAN) {      // (block-uniform)
        int32_t xh[CMAX][K], xl[CMAX][K];
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if ((uint32_t)c < C) {
                const int32_t* xs = fj[c].xs;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const uint32_t n = min(n0 + threadIdx.x + FC_FRAMES * k, nout - 1u);
                    xh[c][k] = xs[n]; xl[c][k] = xs[lo_off + n];
                }
            }
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if ((uint32_t)c < C) {
                const StreamJob& job = fj[c];
                double pk = 0.0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const uint32_t fr = threadIdx.x + FC_FRAMES * k, n = n0 + fr;
                    if (n < nout) {
                        const long long v = ((long long)xh[c][k] << 8) + (long long)xl[c][k] + lo_bias;
                        pk = fmax(pk, emit_sample(epi, job, (double)v * ys, job.n0 + n, fc_lds + fr * fb + job.och * sb));
                    }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) pk = fmax(pk, __shfl_xor(pk, o));
                if ((threadIdx.x & 63) == 0 && pk > 0.0) atomicMax(&pkl[c], (unsigned long long)__double_as_longlong(pk));
            }
        __syncthreads();
        const uint32_t nbytes = min(SPAN, nout - n0) * fb;
        uint8_t* out = reinterpret_cast<uint8_t*>(fj[0].out) + (size_t)n0 * fb;     // 16-byte aligned: the buffer is, and 256 frames are a multiple of 16 bytes
        for (uint32_t i = 16u * threadIdx.x; i < nbytes; i += 16u * FC_FRAMES) {
            if (i + 16u <= nbytes) *reinterpret_cast<uint4*>(out + i) = *reinterpret_cast<const uint4*>(fc_lds + i);
            else for (uint32_t k = i; k < nbytes; ++k) out[k] = fc_lds[k];
        }
        __syncthreads();
    }
    if (threadIdx.x < C && pkl[threadIdx.x]) atomicMax(reinterpret_cast<unsigned long long*>(fj[threadIdx.x].peak), pkl[threadIdx.x]);
}

hipError_t launch_fine_combine(const StreamJob* jobs, uint32_t nstreams, uint32_t max_nout, size_t lo_off, int64_t lo_bias, int sbits,
                               const Epilogue& epi, hipStream_t s) {
    const uint32_t nfiles = nstreams / epi.channels;
    const bool few = epi.channels <= 2;
    const uint32_t span = FC_FRAMES * (few ? 4u : 1u);
    uint32_t gx = (max_nout + span - 1u) / span;
    const uint32_t cap = (8192u + nfiles - 1u) / nfiles;                            // about 8192 blocks in all, each walking its share of the chunks
    if (gx > cap) gx = cap;
    const size_t smem = (size_t)span * epi.channels * epi.sample_bytes;            // at most 64 channels x 4 bytes x 256 = 64 KiB
    if (few) hipLaunchKernelGGL((d2d_fine_combine_kernel<4, 2>), dim3(gx, nfiles), dim3(FC_FRAMES), smem, s, jobs, lo_off, (long long)lo_bias, sbits, epi);
    else hipLaunchKernelGGL((d2d_fine_combine_kernel<1, 64>), dim3(gx, nfiles), dim3(FC_FRAMES), smem, s, jobs, lo_off, (long long)lo_bias, sbits, epi);
    return hipGetLastError();
}

hipError_t launch_history(const StreamJob* jobs, uint32_t nstreams, uint32_t C, uint32_t B, uint32_t keep, hipStream_t s) {
    if (nstreams == 0) return hipSuccess;
    hipLaunchKernelGGL(d2d_history_kernel, dim3(nstreams), dim3(256), 0, s, jobs, C, B, keep);
    return hipGetLastError();
}

hipError_t launch_xhist(const StreamJob* jobs, uint32_t nstreams, uint32_t P, hipStream_t s) {
    if (nstreams == 0) return hipSuccess;
    hipLaunchKernelGGL(d2d_xhist_kernel, dim3(nstreams), dim3(128), 0, s, jobs, P);
    return hipGetLastError();
}

}  // namespace d2d

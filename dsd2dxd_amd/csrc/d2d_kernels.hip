// d2d_kernels.hip -- gfx950 device code of the DSD->PCM engine.
//
//   d2d_fir_lut_kernel<MB>   1-bit FIR decimator by M = 8*MB through per-nibble lookup tables in LDS
//                            (the per-bit-pattern LUT path; 16-entry f64 tables are bank-conflict free:
//                            16 entries x 8 B = 32 banks, equal indices broadcast)
//   d2d_resample_kernel      stage B of the 48k cascade: polyphase L/147 on f64, one fma per tap
//   d2d_history_kernel       carries the last `keep` bytes per channel to the next call
//   d2d_xhist_kernel         carries the last P stage-A outputs per channel to the next call
//
// What they replace: the per-block translate loop inside Rdsd2Pcm::do_conversion
// (/root/reference/src/main.rs:345,429; the rdsd2pcm crate itself is absent from the reference).
#include <hip/hip_runtime.h>

#include <algorithm>

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
        stage_window(win, job, a.epi.channels, a.B, a.keep, abeg, span, tid, LUT_THREADS);
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
                    job.xs[nl] = acc[r];
                } else {
                    uint8_t* dst = reinterpret_cast<uint8_t*>(job.out) + (size_t)nl * frame_bytes + job.ch * sample_bytes;
                    pk = fmax(pk, emit_sample(a.epi, job, acc[r], job.n0 + nl, dst));
                }
            }
        }
    }
    if (!a.to_scratch) block_peak_max(pk, job.peak, red);
}

// Stage B of the 48k cascade: y[m] = sum_k g[phi][k] * x[i_m - k],
// t = Mdn*m, i_m = t div L, phi = t mod L; acc = fma(g, x, acc) for k ascending (the oracle's order,
// so the f64 result is bit-identical).
//
// Write m = L*c + r ("cycle" c, "residue" r).  Then i_m = Mdn*c + (Mdn*r div L) and phi = Mdn*r mod L:
// the phase depends on r only and the window moves by exactly Mdn samples per cycle.  So
//   * a wave takes RS_R consecutive residues and its 64 lanes take 64 consecutive cycles: every
//     coefficient is wave-uniform (scalar loads from a table packed per task, no LDS traffic), and
//     the lanes' x reads are Mdn doubles apart -- Mdn = 147 is odd, so the 64-bit LDS reads are
//     bank-conflict free;
//   * the RS_R outputs of a lane have windows that start within a few samples of each other, so one
//     x read per step feeds all RS_R fma chains; an output whose window has not started / has ended
//     gets a zero coefficient (fma(0, x, acc) == acc exactly: acc is never -0 and x is finite), each
//     chain still sees its own taps in ascending k.
// A tile = 64 cycles of one stream: its Mdn*64 + nsteps stage-A samples are staged in LDS, the
// results go back through LDS so that the PCM stores run along consecutive frames.
constexpr int RS_R = 4;
constexpr int RS_WAVES = 10;                         // L/RS_R tasks per tile = RS_WAVES * NT
constexpr int RS_THREADS = RS_WAVES * 64;
constexpr int RS_SB = 8;                             // staging loads in flight per thread
typedef const __attribute__((address_space(4))) double* rs_const_ptr;

__device__ __forceinline__ uint32_t quantise_bits(const Epilogue& ep, const StreamJob& job, double y, uint64_t n, double& pk) {
    const uint32_t rnd = rng32(job, n);
    pk = fmax(pk, fabs(y * ep.gain));
    if (ep.bits == 32) return __float_as_uint(quantise_f32(ep, y, rnd));
    return (uint32_t)quantise_int(ep, y, rnd);
}

template <int NT>
__global__ __launch_bounds__(RS_THREADS, 2) void d2d_resample_kernel(ResampArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    double* xt = reinterpret_cast<double*>(smem);              // [Mdn*64 + nsteps], later the output tile
    uint32_t* ot = reinterpret_cast<uint32_t*>(smem);          // [64][L]
    __shared__ double red[RS_WAVES];
    const StreamJob job = a.jobs[blockIdx.y];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t L = a.L, Mdn = a.Mdn, nsteps = a.nsteps;
    const D2D_GLOBAL double* xs = as_global(job.xs);
    uint8_t* pcm = reinterpret_cast<uint8_t*>(job.out);
    const uint32_t sample_bytes = a.epi.sample_bytes;
    const uint32_t frame_bytes = sample_bytes * a.epi.channels;
    const uint64_t m_end = job.m0 + job.nres;
    const uint64_t c_first = job.m0 / L, c_last = (m_end - 1) / L;
    const uint32_t ntiles = job.nres ? (uint32_t)((c_last - c_first) / 64 + 1) : 0;
    const uint32_t nx = Mdn * 64 + nsteps;
    double pk = 0.0;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t cg0 = c_first + (uint64_t)tile * 64;
        // xt[i] = stage-A sample (Mdn*cg0 - nsteps + 1 + i); outside what exists it is 0 (never used
        // with a non-zero coefficient by an output that is stored)
        const int64_t rel0 = (int64_t)(Mdn * cg0) - (int64_t)nsteps + 1 - (int64_t)job.n0;
        __syncthreads();
        // loads are issued RS_SB at a time from clamped (always valid) addresses, then masked
        const int64_t jlo = -(int64_t)a.P, jhi = (int64_t)job.nout - 1;
        for (uint32_t base = 0; base < nx; base += RS_THREADS * RS_SB) {
            double v[RS_SB];
#pragma unroll
            for (int u = 0; u < RS_SB; ++u) {
                const int64_t j = rel0 + (int64_t)(base + u * RS_THREADS + tid);
                v[u] = xs[min(max(j, jlo), jhi)];
            }
#pragma unroll
            for (int u = 0; u < RS_SB; ++u) {
                const uint32_t i = base + u * RS_THREADS + tid;
                const int64_t j = rel0 + (int64_t)i;
                if (i < nx) xt[i] = (j >= jlo && j <= jhi) ? v[u] : 0.0;
            }
        }
        __syncthreads();
        double acc[NT][RS_R];
#pragma unroll
        for (int tix = 0; tix < NT; ++tix) {
            const uint32_t task = wave + RS_WAVES * tix;
            const uint32_t bmax = (Mdn * (RS_R * task + RS_R - 1)) / L;
            rs_const_ptr tab = (rs_const_ptr)(a.coef) + (size_t)task * nsteps * RS_R;
            const double* xp = xt + (Mdn * lane + bmax + nsteps - 1);
#pragma unroll
            for (int j = 0; j < RS_R; ++j) acc[tix][j] = 0.0;
#pragma unroll 8
            for (uint32_t s = 0; s < nsteps; ++s) {
                const double x = xp[-(int32_t)s];
#pragma unroll
                for (int j = 0; j < RS_R; ++j) acc[tix][j] = fma(tab[s * RS_R + j], x, acc[tix][j]);
            }
        }
        __syncthreads();                                       // all x reads done: reuse the tile for output
#pragma unroll
        for (int tix = 0; tix < NT; ++tix) {
            const uint32_t task = wave + RS_WAVES * tix;
#pragma unroll
            for (int j = 0; j < RS_R; ++j) {
                const uint32_t r = RS_R * task + j;
                const uint64_t m = (cg0 + lane) * L + r;
                uint32_t bits = 0;
                if (m >= job.m0 && m < m_end) bits = quantise_bits(a.epi, job, acc[tix][j], m, pk);
                ot[lane * L + r] = bits;
            }
        }
        __syncthreads();
        const uint64_t mb = cg0 * L;
        for (uint32_t i = tid; i < 64 * L; i += RS_THREADS) {
            const uint64_t m = mb + i;
            if (m < job.m0 || m >= m_end) continue;
            const uint32_t bits = ot[i];
            uint8_t* dst = pcm + (size_t)(m - job.m0) * frame_bytes + job.ch * sample_bytes;
            if (sample_bytes == 4) *reinterpret_cast<uint32_t*>(dst) = bits;
            else if (sample_bytes == 2) *reinterpret_cast<uint16_t*>(dst) = (uint16_t)bits;
            else { dst[0] = (uint8_t)bits; dst[1] = (uint8_t)(bits >> 8); dst[2] = (uint8_t)(bits >> 16); }
        }
    }
    block_peak_max(pk, job.peak, red);
}

// Byte-interleaved input (DFF, `-f I`: c0 c1 c0 c1 ...) -> the planar 4096-byte-block layout the FIR
// kernels stream with 16-byte loads.  A block moves DI_TILE bytes per channel: the source range is
// contiguous (coalesced 16-byte loads into LDS), each thread then gathers one channel's 16 bytes from
// LDS and stores them as one 16-byte word.  The last (short) block group keeps the engine's layout
// rule: channels packed back to back with the short length.
constexpr uint32_t DI_TILE = 256;
constexpr uint32_t DI_BLOCK = 4096;

__global__ __launch_bounds__(256) void d2d_deinterleave_kernel(const StreamJob* jobs, uint32_t C) {
    extern __shared__ __align__(16) unsigned char smem[];
    const StreamJob job = jobs[blockIdx.y * C];
    const uint32_t L = (uint32_t)job.L;
    const D2D_GLOBAL uint8_t* src = as_global(job.in_raw);
    D2D_GLOBAL uint8_t* dst = as_global(const_cast<uint8_t*>(job.in));
    const uint32_t ntiles = (L + DI_TILE - 1) / DI_TILE;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t j0 = tile * DI_TILE;
        const uint32_t nj = min(DI_TILE, L - j0);                 // bytes per channel in this tile
        const uint32_t nbytes = nj * C;
        const uint64_t s0 = (uint64_t)j0 * C;
        __syncthreads();
        for (uint32_t i = threadIdx.x * 16; i < nbytes; i += 256 * 16) {
            if (i + 16 <= nbytes && ((s0 + i) & 15) == 0) {
                *reinterpret_cast<u32x4*>(smem + i) = *reinterpret_cast<const D2D_GLOBAL u32x4*>(src + s0 + i);
            } else {
                for (uint32_t b = i; b < min(i + 16, nbytes); ++b) smem[b] = src[s0 + b];
            }
        }
        __syncthreads();
        const uint32_t blk = j0 / DI_BLOCK, off0 = j0 - blk * DI_BLOCK;   // DI_TILE divides DI_BLOCK
        const uint32_t blen = min(DI_BLOCK, L - blk * DI_BLOCK);
        const uint32_t nq = (nj + 15) / 16;
        for (uint32_t t = threadIdx.x; t < nq * C; t += 256) {
            const uint32_t c = t / nq, q = t - c * nq;
            const uint32_t n = min(16u, nj - q * 16);
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t b = 0; b < 16; ++b)
                if (b < n) w[b >> 2] |= (uint32_t)smem[(q * 16 + b) * C + c] << (8 * (b & 3));
            const uint64_t d0 = (uint64_t)blk * DI_BLOCK * C + (uint64_t)c * blen + off0 + q * 16;
            if (n == 16 && (d0 & 15) == 0) {
                *reinterpret_cast<D2D_GLOBAL u32x4*>(dst + d0) = u32x4{w[0], w[1], w[2], w[3]};
            } else {
                for (uint32_t b = 0; b < n; ++b) dst[d0 + b] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
            }
        }
    }
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
    double* s = job.xs - P;
    double v = 0.0;
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
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&d2d_fir_lut_kernel<MB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(d2d_fir_lut_kernel<MB>, grid, dim3(LUT_THREADS), smem, s, a);
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

template <int NT>
static hipError_t launch_resample_nt(const ResampArgs& a, uint32_t gx, uint32_t nstreams, size_t smem, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&d2d_resample_kernel<NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024 - 256);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(d2d_resample_kernel<NT>, dim3(gx, nstreams), dim3(RS_THREADS), smem, s, a);
    return hipGetLastError();
}

hipError_t launch_resample(const ResampArgs& a, uint32_t max_out, uint32_t nstreams, hipStream_t s) {
    if (nstreams == 0 || max_out == 0) return hipSuccess;
    const uint32_t ntask = a.L / RS_R;
    if (a.L % RS_R || ntask % RS_WAVES) return hipErrorInvalidValue;
    const size_t smem = std::max((size_t)(a.Mdn * 64 + a.nsteps) * sizeof(double), (size_t)64 * a.L * 4);
    if (smem > 80 * 1024 - 256) return hipErrorInvalidValue;
    uint32_t gx = max_out / (64 * a.L) + 2;                    // tiles follow absolute cycles: up to one extra
    const uint32_t cap = (4096 + nstreams - 1) / nstreams;
    if (gx > cap) gx = cap;
    switch (ntask / RS_WAVES) {
        case 1: return launch_resample_nt<1>(a, gx, nstreams, smem, s);
        case 2: return launch_resample_nt<2>(a, gx, nstreams, smem, s);
        case 4: return launch_resample_nt<4>(a, gx, nstreams, smem, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_deinterleave(const StreamJob* jobs, uint32_t nfiles, uint32_t C, uint32_t max_L, hipStream_t s) {
    if (nfiles == 0 || max_L == 0) return hipSuccess;
    uint32_t gx = (max_L + DI_TILE - 1) / DI_TILE;
    const uint32_t cap = (8192 + nfiles - 1) / nfiles;
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(d2d_deinterleave_kernel, dim3(gx, nfiles), dim3(256), (size_t)DI_TILE * C, s, jobs, C);
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

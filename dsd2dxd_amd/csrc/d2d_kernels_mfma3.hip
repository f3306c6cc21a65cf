// d2d_kernels_mfma3.hip -- the two-group int8 matrix-core FIR decimator, software-pipelined (gfx950), exact.
//
// Same arithmetic, tap tables, staging and output as d2d_kernels_mfma2.hip's stereo 24-bit flavour at 0 dB (EPI = 1 with
// the all-integer requantiser); what changes is WHEN a wave does the requantisation.  The two-group kernel alternates
// between a matrix-core chain and a vector-only epilogue, and with three waves per SIMD the two kinds of phase overlap
// only by chance (the parts of the kernel add up, profiles/r02_ablations.txt).  Here every chain carries the epilogue of
// the chain before it inside its own instruction stream:
//
//   region A (tile t):  chain of channel 0  ||  requantise channel 1 of tile t-1, then pack + store tile t-1
//   region B (tile t):  chain of channel 1  ||  requantise channel 0 of tile t
//
// so each wave keeps the matrix pipe and the vector issue port busy at the same time, whatever its neighbours do; two
// accumulator sets are live, the kernel runs two waves per SIMD (up to 256 VGPRs).  The epilogue interleaved with a chain
// is the branch-free fast form (whole tile, no clip possible, no exact rounding tie, dither counter not wrapping); a tile
// that fails one of those tests is redone after the region from its stream bytes, which are still in the wave's LDS buffer
// of that channel (one buffer per channel), by a plain chain and the general per-sample code -- identical results.
//
// The accumulators start from -2^S spread over the limb-3 rows (the MFMA's C operand) instead of zero, so a sample's limbs
// recombine to v = sum q s directly (no bias to carry through the epilogue), and the requantiser is
//   r = (v + (T >> (16 - F))) >> F,  T = lo16 + hi16 - 32767  (triangular; F = S - 23 fraction bits of x = v * 2^-F LSB)
// which is floor(x + d + 1/2) exactly: floor((a + y) / n) = floor((a + floor(y)) / n) for integers a, n.
//
// Replaces: the per-block translate loop inside Rdsd2Pcm::do_conversion
// (/root/reference/src/main.rs:345,429); the crate that holds it is absent from the reference.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "d2d_mfma2_dev.h"

namespace d2d {

#define D2D_M3_THREADS 512
#ifndef D2D_M3_STAGED
#define D2D_M3_STAGED 1     // M = 8: a full tile's frames go through LDS and leave as aligned 1-KiB rows (0: straight from registers, A/B builds)
#endif

#ifndef D2D_M3_ABL
#define D2D_M3_ABL 0
#endif
#ifndef D2D_M3_STAMPS
#define D2D_M3_STAMPS 0
#endif
#if D2D_M3_STAMPS
// per-wave s_memtime ticks (-DD2D_M3_STAMPS=1): [0] min, [1] max, [2] sum, [3] count of the waves' lifetimes; sums over all waves of
// [4] staging (waiting for the prefetch, LDS writes, next prefetch, stores), [5] the two regions (chain + epilogue), [6] what follows a region
__device__ unsigned long long d2d_m3_stamps[8] = {~0ull, 0, 0, 0, 0, 0, 0, 0};
#endif

__device__ __forceinline__ int32_t m3_lshl_add(int32_t x, uint32_t sh, int32_t y) {
    int32_t d;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(sh), "v"(y));
    return d;
}
__device__ __forceinline__ uint32_t m3_min3_u16(uint32_t x, uint32_t y, uint32_t z) {
    uint32_t d;
    asm("v_min3_u16 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z));
    return d;
}
__device__ __forceinline__ int32_t m3_min3(int32_t x, int32_t y, int32_t z) {
    int32_t d;
    asm("v_min3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z));
    return d;
}
__device__ __forceinline__ int32_t m3_max3(int32_t x, int32_t y, int32_t z) {
    int32_t d;
    asm("v_max3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z));
    return d;
}

typedef int v8i __attribute__((ext_vector_type(8)));

// KIND: 0 no dither, 1 triangular, 2 rectangular (unit gain, all-integer requantiser); 4, 5, 6: the same dithers at any level in dB (f64
// requantiser, M = 8 and 16).  Stereo; SBY = bytes per sample: 3 (24-bit packed frames), 2 (16-bit) or
// 4 (32-bit float, KIND 0 only: the sample is (float)v * 2^-S, one rounding like the oracle's (float)(double)).
// NT = 0 (the parameter once selected a structured-sparse chain, measured slower in round 2 and retired in round 4: profiles/r02_experiments.txt).
#ifndef D2D_M3_SCR_AF
#define D2D_M3_SCR_AF 0           // 1: the scratch flavour walks the call's inner tiles in the fixed-order loop too -- measured SLOWER (DSD64 -> 96 kHz 7.55-7.72
                                  // against 7.14-7.31 ms per step in one lease, profiles/r03_experiments.txt item 13): the general loop stays
#endif
template <int MB, int NPG, int NT, int KIND, int SBY>
__global__ __launch_bounds__(D2D_M3_THREADS) void d2d_fir_mfma3_kernel(Mfma2Args m) {
    using G = M2Geom<MB>;
    constexpr int RS = G::RS, LSH = G::LSH;
    static_assert(NT == 0, "the dense chain");
    // KIND = dither kind DK (0 none, 1 triangular, 2 rectangular), + 4 (GN) for any level in dB: the requantiser then follows the f64
    // definition operation by operation -- x = fl(v * (scale * 2^-S)), q = x + d, round half away, clip -- behind the same chains; it
    // has no careful path (nothing about it depends on the tile)
    constexpr int DK = KIND & 3;
    constexpr bool GN = KIND >= 4;
    constexpr uint32_t FB = 2u * (SBY ? SBY : 1);                   // bytes per stereo frame
    constexpr int TP = NPG + MB;                                    // steps of one chain
    constexpr int NCHK = m2_chunks(MB, NPG);
    constexpr int PF = m2_pf(MB, NPG);
    constexpr uint32_t SB = (uint32_t)m2_stream_bytes(MB, NPG);
    constexpr uint32_t TBL16 = 2u * NPG * 64u;                      // 16-byte units of one table variant
    const FirArgs& a = m.f;
    constexpr uint32_t dbg = D2D_M3_ABL;                  // compile-time ablation mask (tools/ab_build.sh <name> -DD2D_M3_ABL=<mask>): 1 no chain, 2 no epilogue, 4 no staging, 8 never slow, 16 zero taps, 64 no stores
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr bool SCR = SBY == 0;                          // the exact integers y * 2^S to the stage-A scratch (48k cascade, noise-shaping pass)
    const uint32_t Ct = a.in_channels;                     // channels of the file (input layout)
    // a block row = one file (stereo frames) or one channel PAIR of a file (SCR: any even channel count, each channel has its own scratch line)
    const uint32_t fidx = SCR ? blockIdx.y / m.ngroups : blockIdx.y;
    const uint32_t cbase = SCR ? (blockIdx.y - fidx * m.ngroups) * 2u : 0u;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint8_t* wbase = smem + m.off_waves + wave * m.wave_lds;       // [channel 0 stream buffer | channel 1 stream buffer]
    const StreamJob* jobs = a.jobs + (size_t)fidx * (SCR ? a.epi.channels : 2u) + cbase;
    const StreamJob j0 = jobs[0];          // in, L, e0, n0, nout are common to a file's channels

    const int64_t first0 = j0.e0 - (int64_t)a.Wb;          // first byte of output 0's window
    const uint32_t sh = (uint32_t)(first0 & 3);            // its misalignment inside the staged dword
    {   // tap fragments: L2 -> LDS once per block; the variant for this byte misalignment
        const uint4* s = reinterpret_cast<const uint4*>(a.tables) + (size_t)sh * TBL16;
        uint4* dl = reinterpret_cast<uint4*>(smem);
        for (uint32_t i = tid; i < TBL16; i += blockDim.x) dl[i] = (dbg & 16) ? uint4{0, 0, 0, 0} : s[i];      // (16: all-zero taps, a power experiment)
    }
    __syncthreads();

    const uint32_t nwt = (j0.nout + (M2_TILE - 1)) / M2_TILE;      // wave-tiles in this file
    const uint32_t wstride = gridDim.x * m.nwaves;
    const uint32_t r = lane & 31, h = lane >> 5;

    // ---- staging geometry: as in d2d_kernels_mfma2.hip ----
    const uint32_t X0 = (uint32_t)(first0 >> 2) & 3u;
    constexpr uint32_t DUMMY = SB - 16u;
    uint32_t wlo[PF], whi[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        const uint32_t q = lane + 64u * i;
        const uint32_t Lh = 4u * q;
        whi[i] = 4u * (Lh + (Lh >> LSH)) - 4u * X0;
        const uint32_t Ll = 4u * q - X0;
        wlo[i] = q == 0 ? DUMMY : 4u * (Ll + (Ll >> LSH));
    }
    // MONO2 (a.mono2, d2d_kernels_mx.hip has the long comment): a mono stream as a planar pair -- "channel" c = half c of the call's bytes; as
    // one "block" of 2^31 bytes the block arithmetic below degenerates to base + offset
    const bool mono2 = a.mono2 != 0;
    const uint32_t Bsz = mono2 ? 0x80000000u : a.B, Lcall = (uint32_t)j0.L;
    const bool pow2B = Bsz >= 16 && (Bsz & (Bsz - 1)) == 0;
    const uint32_t bshift = pow2B ? 31 - __builtin_clz(Bsz) : 0;
    // IL (a.il2: byte-interleaved stereo -- DFF files, the CLI's default -f I -- both channels converted, M < 64; the scratch flavour too): the
    // tile's frames come as they lie in memory, 2 NCHK pieces of 16 bytes = eight frames each, slot s of a lane = piece lane + 64 s;
    // one v_perm_b32 per channel and dword pair pulls a channel's eight bytes = its stream dwords 2 g, 2 g + 1 (run_loop below)
    constexpr bool ILK = MB < 8;
    const bool il = ILK && a.il2 != 0;
    const uint32_t full_bytes = il || mono2 ? Lcall : pow2B ? (Lcall >> bshift) << bshift : 0;
    uint32_t wil[ILK ? 2 * PF : 1][2];
    if constexpr (ILK) {
#pragma unroll
        for (int sl = 0; sl < 2 * PF; ++sl)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int32_t L = (int32_t)(2u * (lane + 64u * sl)) + kk - (int32_t)X0;
                wil[sl][kk] = L < 0 ? DUMMY + 4u * kk : 4u * ((uint32_t)L + ((uint32_t)L >> LSH));
            }
    }
    const uint32_t jump = (Ct - 1u) * Bsz;
    const bool fast_layout = mono2 || (pow2B && (uint64_t)full_bytes * Ct < (1ull << 32) && jump < (1u << 24));
    auto tile_ab16 = [&](uint32_t w) -> int32_t { return (int32_t)((first0 + (int64_t)w * (M2_TILE * MB)) & ~(int64_t)15); };

    // per-lane chunk offsets: lanes past the last chunk of a tile re-read it (their LDS writes are masked off), so the loads
    // need no predicate and their results no merge with older register contents
    uint32_t lofs[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) { const uint32_t q = lane + 64u * i; lofs[i] = 16u * (q < (uint32_t)NCHK ? q : (uint32_t)NCHK - 1u); }
    const uint32_t chf[2] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)jobs[0].ch), (uint32_t)__builtin_amdgcn_readfirstlane((int)jobs[1].ch)};
    const uint64_t chan_off[2] = {mono2 ? 0ull : (uint64_t)chf[0] << bshift, mono2 ? (uint64_t)Lcall : (uint64_t)chf[1] << bshift};
    uint8_t* const mono_out[2] = {reinterpret_cast<uint8_t*>(jobs[0].out), reinterpret_cast<uint8_t*>(jobs[1].out)};      // (MONO2: each half's own frames)
    // M = 32: one prefetch register set per channel, a tile's bytes are requested a whole tile ahead; M = 64 (five chunks per lane and
    // channel, no registers to spare): one set, a chain's bytes are requested one chain ahead
    constexpr int NPFSET = MB >= 8 ? 1 : 2;
    u32x4 pf[NPFSET][PF];
    // AF ("all fast"): the caller knows that the tile lies inside the call's full power-of-two blocks -- no test, and no byte-gather
    // call in the loop (a call makes the compiler wait for every outstanding load before the next LDS write)
    auto issue_loads = [&](uint32_t w, auto cc, auto af) {
        constexpr int c = decltype(cc)::value;
        constexpr bool AF = decltype(af)::value;
        const int32_t ab = tile_ab16(w);
        if (AF || (fast_layout && ab >= 0 && (uint32_t)ab + 16u * NCHK <= full_bytes)) {
            const uint32_t blk0 = (uint32_t)ab >> bshift, r0 = (uint32_t)ab & (Bsz - 1);
            const uint8_t* base = j0.in + ((uint64_t)(blk0 * Ct) << bshift) + chan_off[c];
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const uint32_t off = r0 + lofs[i];
                const uint32_t o = __umul24(off >> bshift, jump) + off;
                pf[c % NPFSET][i] = *reinterpret_cast<D2D_GLOBAL const u32x4*>(as_global(base) + o);
            }
        } else {
            if constexpr (!AF) {
#pragma unroll
                for (int i = 0; i < PF; ++i) pf[c % NPFSET][i] = gather_chunk(jobs + c, Ct, a.B, a.keep, ab + (int32_t)lofs[i]);
            }
        }
    };
    auto write_lds_x = [&](auto cc, auto xc) {
        constexpr int X = decltype(xc)::value;
        constexpr int c = decltype(cc)::value;
        uint8_t* buf = wbase + c * SB;
#pragma unroll
        for (int i = 0; i < PF; ++i)
            if (lane + 64u * i < (uint32_t)NCHK) {
                const uint32_t v[4] = {pf[c % NPFSET][i].x, pf[c % NPFSET][i].y, pf[c % NPFSET][i].z, pf[c % NPFSET][i].w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    *reinterpret_cast<uint32_t*>(buf + (k < X ? wlo[i] : whi[i]) + 4 * k) = v[k];
            }
    };
    auto write_lds = [&](auto cc) {
        if (X0 == 0) write_lds_x(cc, std::integral_constant<int, 0>{});
        else if (X0 == 1) write_lds_x(cc, std::integral_constant<int, 1>{});
        else if (X0 == 2) write_lds_x(cc, std::integral_constant<int, 2>{});
        else write_lds_x(cc, std::integral_constant<int, 3>{});
    };

    auto il_issue = [&](uint32_t w) {
        if constexpr (ILK) {
            const uint8_t* src = j0.in + 2u * (size_t)(uint32_t)tile_ab16(w);
#pragma unroll
            for (int sl = 0; sl < 2 * PF; ++sl) {
                uint32_t g = lane + 64u * (uint32_t)sl;
                g = g < 2u * (uint32_t)NCHK ? g : 2u * (uint32_t)NCHK - 1u;       // (slots past the last piece re-read it; their writes are masked)
                pf[sl / PF][sl % PF] = *reinterpret_cast<D2D_GLOBAL const u32x4*>(as_global(src) + 16u * g);
            }
        }
    };
    auto il_put = [&](uint32_t c, int sl, uint32_t x, uint32_t y) {
        if constexpr (ILK) {
            uint8_t* buf = wbase + c * SB;
            if (lane + 64u * (uint32_t)sl < 2u * (uint32_t)NCHK) {
                *reinterpret_cast<uint32_t*>(buf + wil[sl][0]) = x;
                *reinterpret_cast<uint32_t*>(buf + wil[sl][1]) = y;
            }
        }
    };

    const v4i* tp = reinterpret_cast<const v4i*>(smem) + lane;      // fragment f: tp[64 * f]
    uint32_t km[8];
#pragma unroll
    for (int p_ = 0; p_ < 8; ++p_) { km[p_] = 0x01010101u << p_; asm volatile("" : "+v"(km[p_])); }
    // accumulators start from -2^S: 128 * (-2^(S-1)) = -2^(S+6) = limb 3 (weight 2^24) times -2^(S-18)
    v16i cinit;
#pragma unroll
    for (int i = 0; i < 16; ++i) cinit[i] = (i & 3) == 3 ? -(1 << (a.scale_bits - 18)) : 0;
    asm volatile("" : "+v"(cinit));

    // One chain: TP pair steps, two groups of eight phases; the LDS reads of a step are issued one step ahead; `hook(u)` is
    // whatever else the wave does during step u.
    auto chain_dense = [&](const uint8_t* rbc, v16i& acc0, v16i& acc1, auto&& hook) {
        // M = 64: row block 1 reads its tap fragments again instead of holding block 0's for MB steps (64 registers the kernel does not
        // have there; LDS has the room)
        constexpr bool REREAD = MB >= 8;
        uint32_t W[TP];
        v4i F[2 * NPG], G[2 * NPG];
        auto rdW = [&](auto uc) { constexpr int u = decltype(uc)::value; W[u] = *reinterpret_cast<const uint32_t*>(rbc + 4 * (2 * u + ((2 * u) >> LSH))); };
        auto rdF = [&](auto uc) { constexpr int u = decltype(uc)::value; F[2 * u] = tp[64 * (2 * u)]; F[2 * u + 1] = tp[64 * (2 * u + 1)]; };
        auto rdG = [&](auto uc) { constexpr int u = decltype(uc)::value; G[2 * u] = tp[64 * (2 * u)]; G[2 * u + 1] = tp[64 * (2 * u + 1)]; };
        constexpr int AHEAD = 2;                                    // LDS reads run this many steps ahead of their use
        static_for<0, AHEAD>([&](auto uc) { rdW(uc); rdF(uc); });
        static_for<0, TP>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            if constexpr (u + AHEAD < TP) rdW(std::integral_constant<int, u + AHEAD>{});
            if constexpr (u + AHEAD < NPG) rdF(std::integral_constant<int, u + AHEAD>{});
            if constexpr (REREAD && u + AHEAD >= MB && u + AHEAD - MB < NPG) rdG(std::integral_constant<int, u + AHEAD - MB>{});
            const uint32_t w = W[u];
            const v4i lo = {(int)(w & km[0]), (int)(w & km[1]), (int)(w & km[2]), (int)(w & km[3])};   // every plane masked: a raw byte operand costs more power than its v_and saves
            const v4i hi = {(int)(w & km[4]), (int)(w & km[5]), (int)(w & km[6]), (int)(w & km[7])};
            if constexpr (u < NPG) {
                if constexpr (u == 0) acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[0], lo, cinit, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[2 * u], lo, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[2 * u + 1], hi, acc0, 0, 0, 0);
            }
            if constexpr (u >= MB && u - MB < NPG) {
                constexpr int pp = u - MB;
                const v4i A0 = REREAD ? G[2 * pp] : F[2 * pp], A1 = REREAD ? G[2 * pp + 1] : F[2 * pp + 1];
                if constexpr (pp == 0) acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A0, lo, cinit, 0, 0, 0);
                else acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A0, lo, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1, hi, acc1, 0, 0, 0);
            }
            hook(uc);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    // the channel's chain: c = 0 / 1 picks the stream buffer
    auto chain = [&](uint32_t c, v16i& acc0, v16i& acc1, auto&& hook) {
        chain_dense(wbase + c * SB + 4u * ((RS + 1) * r + h), acc0, acc1, hook);
    };
    auto no_hook = [](auto) {};

    // dither keys of the two channels (uniform)
    uint32_t rkey[2], rstep[2], rlo0[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) { rkey[c] = jobs[c].rng_key; rstep[c] = jobs[c].rng_kstep; rlo0[c] = jobs[c].rng_lo0; }
    double pk[2] = {0.0, 0.0};                              // peaks met on the slow path, in LSB
    int32_t vmn[2] = {0, 0}, vmx[2] = {0, 0};               // running extremes of v on the fast path

    // constants of the fast epilogue, parked in VGPRs
    const int F_ = SBY == 4 ? 1 : m.fbits;                  // 0 < F <= 16 (integer depths)
    float kFs = ldexpf(1.0f, -a.scale_bits);                // float output: y = v * 2^-S
    asm volatile("" : "+v"(kFs));
    uint32_t kF = (uint32_t)F_, kSh = 16u - (uint32_t)F_, kShR = 32u - (uint32_t)F_;
    uint32_t kC1 = 0x7feb352dU, kC2 = 0x846ca68bU, kTm = (uint32_t)-32767;
    uint32_t k2 = 2u, k10 = 10u, k18 = 18u;
    int32_t kHalf = 1 << (F_ - 1);
    asm volatile("" : "+v"(kF), "+v"(kSh), "+v"(kShR), "+v"(kC1), "+v"(kC2), "+v"(kTm), "+v"(k2), "+v"(k10), "+v"(k18), "+v"(kHalf));
    // |x| <= qmax - 2 LSB keeps x + d inside the range whatever the dither
    const int32_t kSafe = (int32_t)(((uint32_t)m.qmax_i - 2u) << F_);
    const uint32_t lane_fr = 16u * r + 4u * h;              // the lane's first frame inside a tile
    // GN: x = fl(v * kCg) is the oracle's y * scale (y = v * 2^-S exactly; the float flavour: y * gain)
    double kCg = ldexp(a.epi.bits == 32 ? a.epi.gain : a.epi.scale, -a.scale_bits);
    double kLim = a.epi.bits == 32 ? 1.0 : (double)(1u << (a.epi.bits - 1));
    if constexpr (GN) asm volatile("" : "+v"(kCg), "+v"(kLim));
    // the f64 requantiser from the hash word's dither term t (triangular: lo16 + hi16 + 1, rectangular: 2 hi16 + 1), as d2d_device.h: finish_int
    auto quant_gain = [&](int32_t v, uint32_t t) -> int32_t {
        const double x = (double)v * kCg;
        if constexpr (SBY == 4) {
            if constexpr (DK == 3) {
                // Airwindows "Dither Float" as d2d_device.h: quantise_f32 states it (t = the raw hash word)
                const uint32_t fb = __float_as_uint((float)x);
                const int e = (int)((fb >> 23) & 0xFFu);
                const int expon = e ? e - 126 : 0;
                const double tt = ((double)t - 2147483647.0) * 5.5e-36;
                return __float_as_int((float)(x + ldexp(tt, expon + 62)));
            }
            return __float_as_int((float)x);
        }
        double q = x;
        if constexpr (DK == 1) q = x + fma((double)t, 0x1p-16, -1.0);
        else if constexpr (DK == 2) q = x + fma((double)t, 0x1p-17, -0.5);
        const double rq = fmax(fmin(trunc(q + copysign(0.5, q)), kLim - 1.0), -kLim);
        return (int32_t)rq << m.qsh;                                   // (20-bit samples ride in 24 bits as r << 4)
    };

    // v = sum q s of sample k of a group's accumulators: (A0 >> 6) + 4*A1 + 2^10*A2 + 2^18*A3 (A0 is a multiple of 128; mod 2^32)
    auto recombine = [&](const v16i& A, int k) -> int32_t {
        return m3_lshl_add(A[4 * k + 3], k18, m3_lshl_add(A[4 * k + 2], k10, m3_lshl_add(A[4 * k + 1], k2, A[4 * k] >> 6)));
    };
    auto noise = [&](uint32_t c, uint32_t nl) -> uint32_t {
        const uint32_t nlo = (uint32_t)j0.n0 + nl;
        uint32_t z = nlo + rkey[c] + (nlo < rlo0[c] ? rstep[c] : 0u);
        z ^= z >> 16; z *= 0x7feb352dU;
        z ^= z >> 15; z *= 0x846ca68bU;
        z ^= z >> 16;
        return z;
    };
    // the general per-sample requantiser (any tile): x = v * 2^-F LSB, dither in 2^-16 (2^-17) LSB, round half away, clip
    auto quant_slow = [&](int32_t v, uint32_t c, uint32_t nl) -> int32_t {
        const int F = m.fbits;
        const int32_t vh = v >> F;
        const uint32_t vl = (uint32_t)v & ((1u << F) - 1u);
        int32_t rr;
        if constexpr (GN) {
            uint32_t t = 0;
            if constexpr (DK != 0) { const uint32_t z = noise(c, nl); t = DK == 1 ? (z & 0xFFFFu) + (z >> 16) + 1u : DK == 2 ? 2u * (z >> 16) + 1u : z; }
            return quant_gain(v, t);
        } else if constexpr (DK == 2) {
            const uint32_t z = noise(c, nl);
            const int32_t w = (int32_t)(vl << (17 - F)) + (int32_t)(2u * (z >> 16) + 1u) - 65536;
            const int32_t neg = (vh + (w >> 17)) >> 31;
            rr = vh + ((w + 65536 + neg) >> 17);
        } else {
            int32_t w = (int32_t)(vl << (16 - F));
            if constexpr (DK == 1) {
                const uint32_t z = noise(c, nl);
                w += (int32_t)((z & 0xFFFFu) + (z >> 16)) - 65535;
            }
            const int32_t neg = (vh + (w >> 16)) >> 31;
            rr = vh + ((w + 32768 + neg) >> 16);
        }
        return min(max(rr, m.qmin_i), m.qmax_i);
    };

    // ---- the fast epilogue of one (tile, channel), cut into jobs that ride on the steps of a chain ----
    struct Fast {
        uint32_t zb;            // hash input of the lane's first sample
        uint32_t T[8];          // per sample: the dither term
        int32_t res[8];
        int32_t vprev; uint32_t wprev;
        int32_t tmn, tmx; uint32_t tie;
    };
    auto fast_begin = [&](Fast& f, uint32_t tile, uint32_t c) {
        const uint32_t first = (uint32_t)j0.n0 + tile * (uint32_t)M2_TILE;
        const uint32_t key_eff = rkey[c] + (first < rlo0[c] ? rstep[c] : 0u);
        f.zb = first + key_eff + lane_fr;
        f.tmn = 0; f.tmx = 0; f.tie = 0xFFFFu;
    };
    constexpr int NJ = (DK == 0 ? 8 : 16);                  // jobs per epilogue
    auto fast_job = [&](Fast& f, const v16i& o0, const v16i& o1, auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int i = DK == 0 ? j : j >> 1;             // sample 0..7: group i >> 2, k = i & 3
        constexpr bool HASH = DK != 0 && (j & 1) == 0;
        if constexpr (HASH) {
            uint32_t z = f.zb + (uint32_t)(8 * (i >> 2) + (i & 3));
            z ^= z >> 16; z *= kC1;
            z ^= z >> 15; z *= kC2;
            z ^= z >> 16;
            if constexpr (GN) f.T[i] = DK == 1 ? __builtin_amdgcn_sad_u16(z, 0u, 1u) : DK == 2 ? ((z >> 15) | 1u) : z;     // lo16 + hi16 + 1; 2 hi16 + 1; the float dither's word
            else if constexpr (KIND == 1) f.T[i] = __builtin_amdgcn_sad_u16(z, 0u, kTm);      // lo16 + hi16 - 32767, units of 2^-16 LSB
            else f.T[i] = z >> kShR;                                                       // (2*hi16 + 1) >> (17 - F)
        } else {
            const v16i& A = (i >> 2) ? o1 : o0;
            const int32_t v = recombine(A, i & 3);
            int32_t s;
            if constexpr (GN) {
                s = 0;
            } else if constexpr (KIND == 1) {
                s = v + ((int32_t)f.T[i] >> kSh);
                const uint32_t w = (uint32_t)m3_lshl_add(v, kSh, (int32_t)f.T[i]);         // low 16 bits zero: an exact tie
                if constexpr (i & 1) f.tie = m3_min3_u16(f.tie, f.wprev, w); else f.wprev = w;
            } else if constexpr (KIND == 2) {
                s = v + (int32_t)f.T[i];
            } else if constexpr (SBY == 4 || SCR) {
                s = 0;
            } else {
                s = v + kHalf + (v >> 31);                                                 // round half away from zero
            }
            if constexpr (GN) f.res[i] = quant_gain(v, DK != 0 ? f.T[i] : 0u);
            else if constexpr (SBY == 4) f.res[i] = __float_as_int((float)v * kFs);
            else if constexpr (SCR) f.res[i] = v;
            else f.res[i] = s >> kF;
            asm volatile("" : "+v"(f.res[i]));         // keep the whole job on this step (the value is only used after the region)
            if constexpr (!SCR) { if constexpr (i & 1) { f.tmn = m3_min3(f.tmn, f.vprev, v); f.tmx = m3_max3(f.tmx, f.vprev, v); } else f.vprev = v; }
        }
    };
    // the jobs of step u: job j rides on step (j * TP) / NJ
    auto fast_hook = [&](Fast& f, const v16i& o0, const v16i& o1, auto uc) {
        constexpr int u = decltype(uc)::value;
        static_for<0, NJ>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr ((j * TP) / NJ == u) fast_job(f, o0, o1, jc);
        });
    };
    // after the region: did the fast form hold for this (tile, channel)?  (uniform)
    auto fast_failed = [&](const Fast& f, uint32_t tile) -> bool {
        const uint32_t first = (uint32_t)j0.n0 + tile * (uint32_t)M2_TILE;
        const bool full = tile * (uint32_t)M2_TILE + (uint32_t)M2_TILE <= j0.nout;
        if (SCR || (dbg & 8)) return false;                 // (SCR: the integers need no second look; 8: never take the slow path, for timing experiments)
        if (!full || first > 0xFFFFFFFFu - (uint32_t)M2_TILE) return true;
        if constexpr (SBY == 4 || GN) return false;          // float: nothing clips, nothing ties; any level: the f64 requantiser is the definition
        const bool bad = (KIND == 1 && (f.tie & 0xFFFFu) == 0) || f.tmx > kSafe || f.tmn < -kSafe;
        return __builtin_amdgcn_ballot_w64(bad) != 0;
    };
    // the careful way: the channel's chain again (its stream bytes are still in `rbc`'s buffer), then sample by sample
    auto redo = [&](uint32_t cbuf, uint32_t tile, uint32_t c, int32_t (&out)[8]) {
        v16i t0, t1;
        chain(cbuf, t0, t1, no_hook);
        const bool full = tile * (uint32_t)M2_TILE + (uint32_t)M2_TILE <= j0.nout;
        const uint32_t nl_base = tile * (uint32_t)M2_TILE + lane_fr;
        uint32_t vmax = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t nl = nl_base + 8u * (i >> 2) + (i & 3);
            const int32_t v = recombine((i >> 2) ? t1 : t0, i & 3);
            if constexpr (SBY == 4 && !GN) out[i] = __float_as_int((float)v * kFs); else out[i] = quant_slow(v, c, nl);
            const uint32_t va = (uint32_t)(v < 0 ? -v : v);
            vmax = max(vmax, full || nl < j0.nout ? va : 0u);
        }
        pk[c] = fmax(pk[c], ldexp((double)vmax, -m.fbits));   // |x| = |v| * 2^-F exactly
    };
    // a tile's frames: channel 0's samples in L[], channel 1's in R[].  A full tile is packed into registers here (per group
    // the lane owns 4 consecutive frames of both channels = 24 contiguous bytes) and stored by store_packed() AFTER the next
    // prefetch has been issued, so that nothing waits behind the stores; a partial tile (the file's last) goes out frame by
    // frame at once.
    auto tile_full = [&](uint32_t tile) -> bool { return tile * (uint32_t)M2_TILE + (uint32_t)M2_TILE <= j0.nout; };
    auto pack_tile = [&](uint32_t tile, const int32_t (&L)[8], const int32_t (&R)[8], u32x4 (&p4)[2], u32x4 (&p2)[2]) {
        if (tile_full(tile)) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const uint32_t La = L[4 * g], Ra = R[4 * g], Lb = L[4 * g + 1], Rb = R[4 * g + 1];
                const uint32_t Lc = L[4 * g + 2], Rc = R[4 * g + 2], Ld = L[4 * g + 3], Rd = R[4 * g + 3];
                if (mono2) {
                    // two mono streams: the lane's four consecutive samples of each half, 12 / 8 / 16 contiguous bytes per half
                    if constexpr (SBY == 3) {
                        p4[g] = u32x4{(La & 0x00FFFFFFu) | (Lb << 24), ((Lb >> 8) & 0xFFFFu) | (Lc << 16), ((Lc >> 16) & 0xFFu) | (Ld << 8), (Ra & 0x00FFFFFFu) | (Rb << 24)};
                        p2[g] = u32x4{((Rb >> 8) & 0xFFFFu) | (Rc << 16), ((Rc >> 16) & 0xFFu) | (Rd << 8), 0u, 0u};
                    } else if constexpr (SBY == 4) {
                        p4[g] = u32x4{La, Lb, Lc, Ld};
                        p2[g] = u32x4{Ra, Rb, Rc, Rd};
                    } else {
                        p4[g] = u32x4{(La & 0xFFFFu) | (Lb << 16), (Lc & 0xFFFFu) | (Ld << 16), (Ra & 0xFFFFu) | (Rb << 16), (Rc & 0xFFFFu) | (Rd << 16)};
                    }
                } else if constexpr (SBY == 3) {
                    // frames k, k+1 -> 12 bytes: [L0 L1 L2 R0 | R1 R2 L0' L1' | L2' R0' R1' R2']
                    p4[g] = u32x4{__builtin_amdgcn_perm(Ra, La, 0x04020100u), __builtin_amdgcn_perm(Lb, Ra, 0x05040201u),
                                  __builtin_amdgcn_perm(Rb, Lb, 0x06050402u), __builtin_amdgcn_perm(Rc, Lc, 0x04020100u)};
                    p2[g] = u32x4{__builtin_amdgcn_perm(Ld, Rc, 0x05040201u), __builtin_amdgcn_perm(Rd, Ld, 0x06050402u), 0u, 0u};
                } else if constexpr (SBY == 4) {
                    p4[g] = u32x4{La, Ra, Lb, Rb};
                    p2[g] = u32x4{Lc, Rc, Ld, Rd};
                } else {
                    // 16-bit: one dword per frame [L0 L1 R0 R1]
                    p4[g] = u32x4{__builtin_amdgcn_perm(Ra, La, 0x05040100u), __builtin_amdgcn_perm(Rb, Lb, 0x05040100u),
                                  __builtin_amdgcn_perm(Rc, Lc, 0x05040100u), __builtin_amdgcn_perm(Rd, Ld, 0x05040100u)};
                }
            }
        } else {
            // the file's last, partial tile: frame by frame (24-bit: three 2-byte stores each)
            uint8_t* gout = reinterpret_cast<uint8_t*>(j0.out) + (size_t)tile * (M2_TILE * FB) + FB * lane_fr;
            const uint32_t nl_base = tile * (uint32_t)M2_TILE + lane_fr;
#pragma unroll 1
            for (int i = 0; i < 8; ++i) {
                const uint32_t g = (uint32_t)i >> 2, k = (uint32_t)i & 3u;
                if (nl_base + 8u * g + k < j0.nout) {
                    uint32_t Lv = 0, Rv = 0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) { Lv = i == q ? (uint32_t)L[q] : Lv; Rv = i == q ? (uint32_t)R[q] : Rv; }
                    if (mono2) {
                        constexpr uint32_t SBm = SBY ? SBY : 1;
                        D2D_GLOBAL uint8_t* pl = as_global(mono_out[0] + (size_t)(nl_base + 8u * g + k) * SBm);
                        D2D_GLOBAL uint8_t* pr = as_global(mono_out[1] + (size_t)(nl_base + 8u * g + k) * SBm);
#pragma unroll
                        for (uint32_t b = 0; b < SBm; ++b) { pl[b] = (uint8_t)(Lv >> (8 * b)); pr[b] = (uint8_t)(Rv >> (8 * b)); }
                        continue;
                    }
                    D2D_GLOBAL uint16_t* p16 = reinterpret_cast<D2D_GLOBAL uint16_t*>(as_global(gout + 8u * FB * g + FB * k));
                    if constexpr (SBY == 3) { p16[0] = (uint16_t)Lv; p16[1] = (uint16_t)(((Lv >> 16) & 0xFFu) | (Rv << 8)); p16[2] = (uint16_t)(Rv >> 8); }
                    else if constexpr (SBY == 4) { p16[0] = (uint16_t)Lv; p16[1] = (uint16_t)(Lv >> 16); p16[2] = (uint16_t)Rv; p16[3] = (uint16_t)(Rv >> 16); }
                    else { p16[0] = (uint16_t)Lv; p16[1] = (uint16_t)Rv; }
                }
            }
        }
    };
    auto store_packed = [&](uint32_t tile, const u32x4 (&p4)[2], const u32x4 (&p2)[2], bool known_full = false) {
        if (!known_full && !tile_full(tile)) return;
        if constexpr (!SCR) {
            if (mono2) {
                // (straight from the registers at every M: staging the halves through LDS as the stereo frames of M = 8 are was measured, 4.12 against 4.05 ms)
                typedef uint32_t u32x3_a1 __attribute__((ext_vector_type(3), aligned(1)));
                typedef uint32_t u32x2_a1 __attribute__((ext_vector_type(2), aligned(1)));
                typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
                constexpr uint32_t SBm = SBY ? SBY : 1;
                const size_t at = ((size_t)tile * M2_TILE + lane_fr) * SBm;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    if (dbg & 64) { asm volatile("" :: "v"(p4[g]), "v"(p2[g])); continue; }
                    uint8_t* gl = mono_out[0] + at + 8u * SBm * g;
                    uint8_t* gr = mono_out[1] + at + 8u * SBm * g;
                    if constexpr (SBY == 3) {
                        *reinterpret_cast<D2D_GLOBAL u32x3_a1*>(as_global(gl)) = u32x3_a1{p4[g].x, p4[g].y, p4[g].z};
                        *reinterpret_cast<D2D_GLOBAL u32x3_a1*>(as_global(gr)) = u32x3_a1{p4[g].w, p2[g].x, p2[g].y};
                    } else if constexpr (SBY == 4) {
                        *reinterpret_cast<D2D_GLOBAL u32x4_a1*>(as_global(gl)) = u32x4_a1{p4[g].x, p4[g].y, p4[g].z, p4[g].w};
                        *reinterpret_cast<D2D_GLOBAL u32x4_a1*>(as_global(gr)) = u32x4_a1{p2[g].x, p2[g].y, p2[g].z, p2[g].w};
                    } else {
                        *reinterpret_cast<D2D_GLOBAL u32x2_a1*>(as_global(gl)) = u32x2_a1{p4[g].x, p4[g].y};
                        *reinterpret_cast<D2D_GLOBAL u32x2_a1*>(as_global(gr)) = u32x2_a1{p4[g].z, p4[g].w};
                    }
                }
                return;
            }
        }
        if constexpr (D2D_M3_STAGED && MB == 1 && !SCR) {
            // M = 8: the stores are what this shape waits for (profiles/r02_experiments.txt item 17): the tile's frames through
            // LDS, then 16 bytes per lane along the tile -- every store instruction writes eight whole, aligned lines
            uint8_t* ob = wbase + m.off_out;
            typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                u32x2* d = reinterpret_cast<u32x2*>(ob + FB * lane_fr + 8 * FB * g);      // 8-byte aligned
                d[0] = u32x2{p4[g].x, p4[g].y}; d[1] = u32x2{p4[g].z, p4[g].w};
                if constexpr (SBY >= 3) d[2] = u32x2{p2[g].x, p2[g].y};
                if constexpr (SBY == 4) d[3] = u32x2{p2[g].z, p2[g].w};
            }
            wave_sync2();
            uint8_t* gt = reinterpret_cast<uint8_t*>(j0.out) + (size_t)tile * (M2_TILE * FB) + 16u * lane;
#pragma unroll
            for (uint32_t j = 0; j < (uint32_t)M2_TILE * FB / 1024u; ++j) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(ob + 1024u * j + 16u * lane);
                if (dbg & 64) { asm volatile("" :: "v"(v)); continue; }
                *reinterpret_cast<D2D_GLOBAL u32x4_a4*>(as_global(gt + 1024u * j)) = u32x4_a4{v.x, v.y, v.z, v.w};
            }
            return;
        }
        uint8_t* gout = reinterpret_cast<uint8_t*>(j0.out) + (size_t)tile * (M2_TILE * FB) + FB * lane_fr;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (dbg & 64) { asm volatile("" :: "v"(p4[g]), "v"(p2[g])); continue; }
            *reinterpret_cast<D2D_GLOBAL u32x4*>(as_global(gout + 8 * FB * g)) = p4[g];
            if constexpr (SBY == 3) *reinterpret_cast<D2D_GLOBAL u32x2*>(as_global(gout + 8 * FB * g + 16)) = u32x2{p2[g].x, p2[g].y};
            if constexpr (SBY == 4) *reinterpret_cast<D2D_GLOBAL u32x4*>(as_global(gout + 8 * FB * g + 16)) = p2[g];
        }
    };
    // SCR: the lane's 2 x 4 consecutive integers of channel c go straight to that channel's scratch line
    auto store_scr = [&](uint32_t tile, uint32_t c, const int32_t (&v)[8], bool known_full = false) {
        D2D_GLOBAL int32_t* xs = as_global(jobs[c].xs) + (size_t)tile * M2_TILE + lane_fr;
        const uint32_t nl = tile * (uint32_t)M2_TILE + lane_fr;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (known_full || nl + 8u * g + 3u < j0.nout) *reinterpret_cast<D2D_GLOBAL i32x4*>(xs + 8 * g) = i32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
            else
#pragma unroll
                for (int k = 0; k < 4; ++k) if (nl + 8u * g + k < j0.nout) xs[8 * g + k] = v[4 * g + k];
        }
    };
    auto merge_extremes = [&](const Fast& f, uint32_t c) { vmn[c] = min(vmn[c], f.tmn); vmx[c] = max(vmx[c], f.tmx); };

    const uint32_t wv = blockIdx.x * m.nwaves + wave;       // this wave's index among the file's waves
#if D2D_M3_STAMPS
    const unsigned long long t_start = __builtin_amdgcn_s_memtime(), rt_start = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_sum[3] = {0, 0, 0}, st_last = t_start;
    auto stamp = [&](int slot) { const unsigned long long t = __builtin_amdgcn_s_memtime(); st_sum[slot] += t - st_last; st_last = t; };
#else
    auto stamp = [](int) {};
#endif
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    // The pipelined loop over the tiles t_begin + wv + k * wstride < t_end.
    //   IL (byte-interleaved stereo, two prefetch sets = all pieces of a tile; every tile of the range inside the call):
    //   A start: [pf = the pieces of tile t]  ch0 parts -> buf0, ch1 parts -> keep;  request the pieces of tile t+1 (a whole tile ahead)
    //   B start: keep -> buf1
    auto run_loop = [&](uint32_t t_begin, uint32_t t_end, auto af, auto ilc) {
        constexpr bool AF = decltype(af)::value;
        constexpr bool IL = decltype(ilc)::value;
        static_assert(!IL || (AF && ILK && NPFSET == 2), "the interleaved staging: the fixed-order loop with two prefetch sets");
        [[maybe_unused]] uint32_t keep[IL ? 4 * PF : 1];
        uint32_t wt = t_begin + wv;
        // the packed frames of the tile before: AF keeps them across trips (it stores on every trip), the general loop only from
        // the pack to the store
        u32x4 p4h[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}}, p2h[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
        if (wt < t_end) {
            if constexpr (IL) il_issue(wt);
            else {
                issue_loads(wt, C0{}, af);
                if constexpr (NPFSET == 2) issue_loads(wt, C1{}, af);
            }
            // AF: every trip issues the same loads and stores in the same order (the first trip stores zeros to its own tile, rewritten one
            // trip later; the last trip re-requests its own tile), so that the compiler can count exactly how many younger requests
            // may stay in flight at each LDS write -- with a conditional load or store in the loop it waits for all of them
            if constexpr (!SCR) { if (AF && !(dbg & 64)) store_packed(wt, p4h, p2h, true); }
        }
        v16i accA[2], accB[2];                                  // channel 0's / channel 1's accumulators
#pragma unroll
        for (int i = 0; i < 16; ++i) { accB[0][i] = 0; accB[1][i] = 0; }
        int32_t held[8];                                        // channel 0's samples of the tile in flight
        bool have_prev = false;
        uint32_t pw = wt;                                       // the tile whose channel 1 still waits for its epilogue
        for (; wt < t_end; wt += wstride) {
            const bool more = wt + wstride < t_end;
            const uint32_t nxt = more ? wt + wstride : wt;
            u32x4 p4l[2], p2l[2];
            auto& p4 = AF ? p4h : p4l;
            auto& p2 = AF ? p2h : p2l;
            // ---- region A: channel 0's chain of tile wt, channel 1's epilogue of tile pw ----
            stamp(2);
            wave_sync2();
            if constexpr (IL) {
#pragma unroll
                for (int sl = 0; sl < 2 * PF; ++sl) {
                    const u32x4 d = pf[sl / PF][sl % PF];
                    il_put(0u, sl, __builtin_amdgcn_perm(d.y, d.x, 0x06040200u), __builtin_amdgcn_perm(d.w, d.z, 0x06040200u));
                    keep[2 * sl] = __builtin_amdgcn_perm(d.y, d.x, 0x07050301u);
                    keep[2 * sl + 1] = __builtin_amdgcn_perm(d.w, d.z, 0x07050301u);
                }
                il_issue(nxt);
            } else if (!(dbg & 4)) {
                write_lds(C0{});
                if constexpr (NPFSET == 2) { if (AF || more) issue_loads(nxt, C0{}, af); }
                else issue_loads(wt, C1{}, af);
            }
            wave_sync2();
            stamp(0);
            {
                Fast f;
                fast_begin(f, pw, 1);
                if (dbg & 2) chain(0u, accA[0], accA[1], no_hook);
                else if (dbg & 1) { static_for<0, NJ>([&](auto jc) { fast_job(f, accB[0], accB[1], jc); }); accA[0] = cinit + (int)lane; accA[1] = cinit - (int)lane; }
                else chain(0u, accA[0], accA[1], [&](auto uc) { fast_hook(f, accB[0], accB[1], uc); });
                if (D2D_M3_STAMPS) asm volatile("" :: "v"(accA[0]), "v"(accA[1]));
                stamp(1);
                if constexpr (SCR && AF) store_scr(pw, 1, f.res, true);        // (first trip: its own tile, rewritten one trip later)
                else if (have_prev) {
                    if constexpr (SCR) store_scr(pw, 1, f.res);
                    else {
                        if (!(dbg & 3) && fast_failed(f, pw)) redo(1u, pw, 1, f.res); else merge_extremes(f, 1);
                        pack_tile(pw, held, f.res, p4, p2);
                    }
                }
            }
            // ---- region B: channel 1's chain of tile wt, channel 0's epilogue of tile wt ----
            stamp(2);
            wave_sync2();
            if constexpr (IL) {
#pragma unroll
                for (int sl = 0; sl < 2 * PF; ++sl) il_put(1u, sl, keep[2 * sl], keep[2 * sl + 1]);
            } else if (!(dbg & 4)) {
                write_lds(C1{});
                if (AF || more) { if constexpr (NPFSET == 2) issue_loads(nxt, C1{}, af); else issue_loads(nxt, C0{}, af); }
            }
            if constexpr (AF) { if constexpr (!SCR) store_packed(pw, p4, p2, true); }
            else if (have_prev && !SCR) store_packed(pw, p4, p2);
            wave_sync2();
            stamp(0);
            {
                Fast f;
                fast_begin(f, wt, 0);
                if (dbg & 2) chain(1u, accB[0], accB[1], no_hook);
                else if (dbg & 1) { static_for<0, NJ>([&](auto jc) { fast_job(f, accA[0], accA[1], jc); }); accB[0] = cinit - (int)lane; accB[1] = cinit + (int)lane; }
                else chain(1u, accB[0], accB[1], [&](auto uc) { fast_hook(f, accA[0], accA[1], uc); });
                if (D2D_M3_STAMPS) asm volatile("" :: "v"(accB[0]), "v"(accB[1]));
                stamp(1);
                if constexpr (SCR) store_scr(wt, 0, f.res, AF);
                else {
                    if (!(dbg & 3) && fast_failed(f, wt)) redo(0u, wt, 0, f.res); else merge_extremes(f, 0);
#pragma unroll
                    for (int i = 0; i < 8; ++i) held[i] = f.res[i];
                }
            }
            have_prev = true; pw = wt;
        }
        if (have_prev) {
            // drain: channel 1 of the wave's last tile
            Fast f;
            fast_begin(f, pw, 1);
            static_for<0, NJ>([&](auto jc) { fast_job(f, accB[0], accB[1], jc); });
            if constexpr (SCR) store_scr(pw, 1, f.res);
            else {
                if (fast_failed(f, pw)) redo(1u, pw, 1, f.res); else merge_extremes(f, 1);
                pack_tile(pw, held, f.res, p4h, p2h);
                store_packed(pw, p4h, p2h);
            }
        }
    };
    // One tile the careful way, start to finish (call edges: the window reaches into the carried history or past the call's full
    // blocks, so its bytes are gathered one by one).
    auto slow_tile = [&](uint32_t t) {
        wave_sync2();
        issue_loads(t, C0{}, std::false_type{});
        if constexpr (NPFSET == 2) issue_loads(t, C1{}, std::false_type{});
        write_lds(C0{});
        if constexpr (NPFSET == 1) issue_loads(t, C1{}, std::false_type{});
        write_lds(C1{});
        wave_sync2();
        if constexpr (SCR) {
            // the exact integers need no careful path: the chain, then every job of the epilogue at once
            v16i A0, A1;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                chain((uint32_t)c, A0, A1, no_hook);
                Fast f;
                fast_begin(f, t, (uint32_t)c);
                static_for<0, NJ>([&](auto jc) { fast_job(f, A0, A1, jc); });
                store_scr(t, (uint32_t)c, f.res);
            }
        } else {
            int32_t o0[8], o1[8];
            redo(0u, t, 0, o0);
            redo(1u, t, 1, o1);
            u32x4 p4[2]; u32x4 p2[2];
            pack_tile(t, o0, o1, p4, p2);
            store_packed(t, p4, p2);
        }
    };
    if ((fast_layout || il) && MB < 8 && (!SCR || D2D_M3_SCR_AF || il)) {       // (the scratch flavour: the fixed-order loop only where it saves the de-interleave pass)       // (M = 64: the general loop is faster there, 3.01 against 3.18 ms)
        // the tiles [t_lo, t_hi) lie inside the call's full blocks: the loop without the gather path; the few around them one by one
        const int64_t T = (int64_t)M2_TILE * MB;
        auto is_fast = [&](uint32_t w) { const int32_t ab = tile_ab16(w); return ab >= 0 && (uint32_t)ab + 16u * NCHK <= full_bytes; };
        uint32_t t_lo = first0 >= 0 ? 0u : (uint32_t)((-first0 + T - 1) / T);
        if (t_lo > nwt) t_lo = nwt;
        uint32_t t_hi = t_lo;
        {
            const int64_t room = (int64_t)full_bytes - 16 * NCHK - first0;
            if (room >= 0) { const int64_t e = room / T + 1; t_hi = e > (int64_t)nwt ? nwt : (uint32_t)e; if (t_hi < t_lo) t_hi = t_lo; }
            while (t_hi > t_lo && !is_fast(t_hi - 1)) --t_hi;
            while (t_hi < nwt && t_hi >= t_lo && is_fast(t_hi) && (t_hi > t_lo || is_fast(t_lo))) ++t_hi;
        }
        { const uint32_t nfull = j0.nout / (uint32_t)M2_TILE; if (t_hi > nfull) t_hi = nfull > t_lo ? nfull : t_lo; }     // whole tiles only
        if constexpr (ILK && NPFSET == 2) { if (il) run_loop(t_lo, t_hi, std::true_type{}, std::true_type{}); else run_loop(t_lo, t_hi, std::true_type{}, std::false_type{}); }
        else run_loop(t_lo, t_hi, std::true_type{}, std::false_type{});
        const uint32_t n_edge = t_lo + (nwt - t_hi);
        for (uint32_t i = wv; i < n_edge; i += wstride) slow_tile(i < t_lo ? i : t_hi + (i - t_lo));
    } else {
        run_loop(0u, nwt, std::false_type{}, std::false_type{});
    }

#if D2D_M3_STAMPS
    if (lane == 0) {
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_start;
        atomicMin(&d2d_m3_stamps[0], dt); atomicMax(&d2d_m3_stamps[1], dt); atomicAdd(&d2d_m3_stamps[2], dt); atomicAdd(&d2d_m3_stamps[3], 1ull);
        for (int i = 0; i < 3; ++i) atomicAdd(&d2d_m3_stamps[4 + i], st_sum[i]);
        atomicAdd(&d2d_m3_stamps[7], __builtin_amdgcn_s_memrealtime() - rt_start);      // constant 100 MHz: sum[2] / sum[7] = core clock / 100 MHz
    }
#endif
    if constexpr (SCR) return;                              // (stage B / the noise shaper keep the peaks)
    // peak meter: |x| in LSB; undo the power-of-two part exactly
    const double unscale = 1.0 / (double)(1u << (a.epi.bits - 1));   // (float: fbits = S - 31, so dev * 2^-fbits * 2^-31 = dev * 2^-S)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int32_t dev = max(vmx[c], -vmn[c]);
        double p = fmax(pk[c], ldexp((double)dev, -m.fbits)) * unscale;
        if constexpr (GN) p = p * a.epi.gain;                          // |y| is exact: one rounding, as the oracle's |y * gain| of the largest sample
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) p = fmax(p, __shfl_xor(p, o));
        if (lane == 0 && p > 0.0)
            atomicMax(reinterpret_cast<unsigned long long*>(jobs[c].peak), (unsigned long long)__double_as_longlong(p));
    }
}

// ---- host side -------------------------------------------------------------------------------
// The file is compiled twice (Makefile): D2D_M3_PART 0 = the 24-bit kernels, the table builder and the dispatcher; 1 = the 16-bit
// kernels (halves the compile time of the longest translation unit).
#ifndef D2D_M3_PART
#define D2D_M3_PART 0
#endif

// (MB, NPG, taps) of the filters this kernel serves: X_M32, C_M32, E_M32, C_M64, E_M64 (the stage-A filters only ever write the scratch)
#ifdef D2D_M2_DEV
#define D2D_M3_SHAPES(X) X(4, 13, 560)
#else
// (the M = 8 and 16 shapes are served by this kernel only: the two-group kernel loses to the one-group one there)
#define D2D_M3_SHAPES(X) X(1, 3, 96) X(1, 4, 144) X(2, 6, 192) X(2, 7, 256) X(2, 7, 288) X(4, 10, 384) X(4, 12, 512) X(4, 13, 560) X(8, 24, 1024) X(8, 25, 1104)
#endif

#if D2D_M3_PART == 0
// (MB, NPG) of every filter that can write the scratch: the 44.1k-family filters above (noise-shaping pass) and the stage-A filters of the
// 48k cascade (A_M8 96 taps, A_M16 176, A_M32 352, A_M64 688)
#define D2D_M3_SCR_SHAPES(X) X(1, 3) X(1, 4) X(2, 5) X(2, 6) X(2, 7) X(4, 10) X(4, 12) X(4, 13) X(8, 19) X(8, 24) X(8, 25)
bool mfma3_scr_supported(int MB, int NPG) {
#define X(mb, npg) if (MB == mb && NPG == npg) return true;
    D2D_M3_SCR_SHAPES(X)
#undef X
    return false;
}

bool mfma3_supported(int MB, int NPG, int NT) {
#define X(mb, npg, nt) if (MB == mb && NPG == npg && NT == nt) return true;
    D2D_M3_SHAPES(X)
#undef X
    return false;
}

static inline int8_t limb_of4(int64_t v, int l) {
    // balanced base-256 digits: v = d0 + d1*2^8 + d2*2^16 + d3*2^24, every d in [-128, 127]
    int8_t dgt = 0;
    for (int i = 0; i <= l; ++i) {
        int64_t dd = ((v + 128) & 255) - 128;
        dgt = (int8_t)dd;
        v = (v - dd) / 256;
    }
    return dgt;
}


#endif

template <int MB, int NPG, int NT, int KIND, int SBY>
static hipError_t launch_mfma3_t(Mfma2Args& m, uint32_t nwt_max, uint32_t nrows, hipStream_t s) {
    static KernelPrep prep;
    int dev = 0;
    const void* fn = reinterpret_cast<const void*>(&d2d_fir_mfma3_kernel<MB, NPG, NT, KIND, SBY>);
    hipError_t e = prep.max_dynamic_lds(fn, 160 * 1024, &dev);
    if (e != hipSuccess) return e;
    // LDS: the shared tap table, then two stream buffers per wave; eight waves per block = two per SIMD
    m.off_waves = (uint32_t)(2 * NPG) * 1024u;
    m.wave_lds = 2u * (uint32_t)m2_stream_bytes(MB, NPG);
    m.off_out = m.wave_lds;
    if (D2D_M3_STAGED && MB == 1 && SBY != 0) m.wave_lds += (uint32_t)M2_TILE * 2u * SBY;     // the tile's frames, staged for whole-line stores
    const uint32_t wdbg = (m.f.dbg_flags >> 8) & 0xFFu;   // diagnostic override (d2d_params.debug_flags bits 8..15)
    m.nwaves = wdbg ? wdbg : 8u;
    if (m.nwaves < 1 || m.nwaves > 8) m.nwaves = 8;
    while (m.nwaves > 1 && (size_t)m.off_waves + (size_t)m.nwaves * m.wave_lds > 160 * 1024) m.nwaves >>= 1;
    const size_t smem = (size_t)m.off_waves + (size_t)m.nwaves * m.wave_lds;
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    int blocks_per_cu, ncu;
    {
        std::lock_guard<std::mutex> g(prep.mu);
        if (prep.blocks_per_cu[dev] == 0 || smem != prep.smem_seen[dev] || m.nwaves != prep.nwaves_seen[dev]) {
            hipDeviceProp_t prop;
            if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
            int nb = 0;
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, d2d_fir_mfma3_kernel<MB, NPG, NT, KIND, SBY>, (int)(64 * m.nwaves), smem);
            if (e != hipSuccess) return e;
            prep.ncu[dev] = prop.multiProcessorCount;
            prep.blocks_per_cu[dev] = nb < 1 ? 1 : nb;
            prep.smem_seen[dev] = smem; prep.nwaves_seen[dev] = m.nwaves;
        }
        blocks_per_cu = prep.blocks_per_cu[dev]; ncu = prep.ncu[dev];
    }
    // every wave loops over its share of the wave-tiles: launch what is resident at once
    uint32_t gx = (uint32_t)(ncu * blocks_per_cu) / nrows;
    if (gx < 1) gx = 1;
    const uint32_t need = (nwt_max + m.nwaves - 1) / m.nwaves;
    if (gx > need) gx = need;
    hipLaunchKernelGGL((d2d_fir_mfma3_kernel<MB, NPG, NT, KIND, SBY>), dim3(gx, nrows), dim3(64 * m.nwaves), smem, s, m);
    d2d_last_launched_kernel = launched_name<MB, NPG, NT, KIND, SBY>("d2d_fir_mfma3_kernel");
    return hipGetLastError();
}

// any level in dB (Mfma2Args::gainq): KIND + 4, compiled for the shapes this kernel serves by default (M = 8, 16)
template <int MB, int NPG, int NT, int SBY>
static hipError_t launch_mfma3_gain(Mfma2Args& m, uint32_t nwt_max, uint32_t nrows, hipStream_t s) {
    if constexpr (MB < 4 && NT == 0 && SBY != 0) {
        if constexpr (SBY != 4) {
            if (m.dkind == 1) return launch_mfma3_t<MB, NPG, NT, 5, SBY>(m, nwt_max, nrows, s);
            if (m.dkind == 2) return launch_mfma3_t<MB, NPG, NT, 6, SBY>(m, nwt_max, nrows, s);
        } else {
            if (m.f.epi.dither == 'F') return launch_mfma3_t<MB, NPG, NT, 7, SBY>(m, nwt_max, nrows, s);      // the float dither
        }
        return launch_mfma3_t<MB, NPG, NT, 4, SBY>(m, nwt_max, nrows, s);
    } else return hipErrorInvalidValue;
}
#define K3(mb, npg, nt, sby)                                                                          \
    { if (m.gainq) return launch_mfma3_gain<mb, npg, nt, sby>(m, nwt_max, nrows, s);                    \
      if (m.dkind == 1) return launch_mfma3_t<mb, npg, nt, 1, sby>(m, nwt_max, nrows, s);              \
      if (m.dkind == 2) return launch_mfma3_t<mb, npg, nt, 2, sby>(m, nwt_max, nrows, s);              \
      return launch_mfma3_t<mb, npg, nt, 0, sby>(m, nwt_max, nrows, s); }
#if D2D_M3_PART == 1
#define D2D_M3_SCR_SHAPES(X) X(1, 3) X(1, 4) X(2, 5) X(2, 6) X(2, 7) X(4, 10) X(4, 12) X(4, 13) X(8, 19) X(8, 24) X(8, 25)
hipError_t launch_fir_mfma3_scr(Mfma2Args& m, int MB, int NPG, uint32_t nwt_max, uint32_t nrows, hipStream_t s) {
#define X(mb, npg) if (MB == mb && NPG == npg) return launch_mfma3_t<mb, npg, 0, 0, 0>(m, nwt_max, nrows, s);
    D2D_M3_SCR_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}
hipError_t launch_fir_mfma3_s16(Mfma2Args& m, int MB, int NPG, int NT, uint32_t nwt_max, uint32_t nrows, hipStream_t s) {
    if (m.f.epi.sample_bytes == 4) {       // float: no dither (the float dither 'F' stays with the two-group kernel)
#define X(mb, npg, nt) if (MB == mb && NPG == npg && NT == nt) return m.gainq ? launch_mfma3_gain<mb, npg, 0, 4>(m, nwt_max, nrows, s) : launch_mfma3_t<mb, npg, 0, 0, 4>(m, nwt_max, nrows, s);
        D2D_M3_SHAPES(X)
#undef X
        return hipErrorInvalidValue;
    }
#define X(mb, npg, nt) if (MB == mb && NPG == npg && NT == nt) K3(mb, npg, 0, 2)
    D2D_M3_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}
#else
hipError_t launch_fir_mfma3_s16(Mfma2Args& m, int MB, int NPG, int NT, uint32_t nwt_max, uint32_t nrows, hipStream_t s);
hipError_t launch_fir_mfma3_scr(Mfma2Args& m, int MB, int NPG, uint32_t nwt_max, uint32_t nrows, hipStream_t s);
hipError_t launch_fir_mfma3(Mfma2Args& m, int variant, int MB, int NPG, int NT, uint32_t nwt_max, uint32_t nrows, hipStream_t s) {
    if (m.f.to_scratch) return launch_fir_mfma3_scr(m, MB, NPG, nwt_max, nrows, s);
    if (m.f.epi.sample_bytes != 3) return launch_fir_mfma3_s16(m, MB, NPG, NT, nwt_max, nrows, s);     // 16-bit and float frames: part 1
    (void)variant;
#define X(mb, npg, nt) if (MB == mb && NPG == npg && NT == nt) K3(mb, npg, 0, 3)
    D2D_M3_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}
#endif
#undef K3

#if D2D_M3_PART == 0
#if D2D_M3_STAMPS
void mfma3_debug_stamps(unsigned long long out[8]) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(d2d_m3_stamps), sizeof(unsigned long long) * 8);
    unsigned long long z[8] = {~0ull, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(d2d_m3_stamps), z, sizeof(z));
}
#else
void mfma3_debug_stamps(unsigned long long out[8]) { for (int i = 0; i < 8; ++i) out[i] = 0; }
#endif
#endif

}  // namespace d2d

// d2d_kernels_px.hip -- DSD64 / DSD128 -> 96 / 192 / 384 kHz in ONE pass over the packed bits (gfx950), exact.
//
// The reference documents these rates as "cascaded FIR filters" (README.md:230, src/main.rs:88-89); until round 3 this engine ran them as
// two kernels that met in HBM (a decimator to 352.8 kHz writing int32, a polyphase L/147 resampler reading them back: 5.7 x the
// algorithmic traffic at DSD64 -> 96 kHz).  Here the two designs are composed into one polyphase filter on the bits (tools/design_filters.py:
// compose_polyphase),
//
//     y[m] = sum_j c[rho][j] s[q + D - j],   Mp m = Lp q + rho,   c = Q 2^-S,
//
// which is the decimators' arithmetic with a tap set per phase rho and windows that start at ANY bit: v = sum Q s is an exact integer
// (|v| < 2^31), y = v 2^-S, then level / dither / requantise as everywhere else.
//
//   d2d_fir_px_kernel<LP, MP, NP, G, KIND>   the fp6 x fp4 matrix-core form (v_mfma_scale_f32_32x32x64_f8f6f4, as d2d_kernels_mx.hip):
//       B operand = the bit stream, one nibble per bit (five vector instructions per 32 bits), A operand = the taps in five balanced
//       base-32 digits (e2m3), f32 accumulators holding exact digit sums.  Matrix row = (output of a GROUP of five, digit): 25 of 32
//       rows; a lane half owns outputs 0-2 / 3-4 of every group with all five digits of a sample in its own registers.  Matrix column
//       = G consecutive groups, a whole number of the filter's cycles, so every column sees the same taps at the same places; its
//       window starts at an arbitrary BIT of the stream: the lane reads the two dwords around it and one v_alignbit_b32 lines the
//       column up (the staged image is in time order, LSB first: MSB-first streams are bit-reversed per byte while they are staged).
//       Every (step, group) pair has its own tap fragment (the groups are 147 / 73.5 / 36.75 bits apart: no two share one).
//   d2d_poly_plain_kernel                    the same sums bit by bit, one output per lane: what D2D_KERNEL_LUT engines run at these rates
//                                            and the cross-check of the matrix-core form in every parity test.
//
// Replaces: the 48 kHz-family path inside Rdsd2Pcm::do_conversion (/root/reference/src/main.rs:345,429); the crate that holds it is
// absent from the reference.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>

#include "d2d_mfma2_dev.h"
#include "d2d_px.h"

namespace d2d {

typedef int px_v8i __attribute__((ext_vector_type(8)));
typedef float px_v16f __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4_pa2 __attribute__((ext_vector_type(4), aligned(2)));
typedef uint32_t u32x2_pa2 __attribute__((ext_vector_type(2), aligned(2)));

// time order inside every byte: bit 7 first -> bit 0 first
__device__ __forceinline__ uint32_t px_lsb_first(uint32_t w) { return __builtin_amdgcn_perm(0u, __builtin_bitreverse32(w), 0x00010203u); }

#ifndef D2D_PX_PART
#define D2D_PX_PART 0
#endif
#ifndef D2D_PX_ABL
#define D2D_PX_ABL 0        // compile-time ablation mask of A/B builds (tools/ab_build.sh px): 1 no chain, 2 no epilogue arithmetic, 4 no staging loads, 8 no stores
#endif

// KIND: 0 no dither, 1 triangular, 2 rectangular (unit gain, 16 / 24 bits: the all-integer requantiser); 3: every other format through the
// f64 epilogue of d2d_device.h; 4: the exact integers to the scratch (noise-shaped dither)
template <int LP, int MP, int NP, int G, int KIND>
__global__ __launch_bounds__(PX_THREADS) void d2d_fir_px_kernel(PxArgs a) {
    constexpr int TP = px_tp(LP, MP, NP, G), NSLOT = px_nslot(LP, MP, NP, G), SBITS = px_sbits(LP, MP, G);
    constexpr int OC = 5 * G, TILE = 32 * OC;
    constexpr int NCHK = px_chunks(LP, MP, NP, G), PF = (NCHK + 63) / 64;
    constexpr uint32_t SB = (uint32_t)px_stream_bytes(LP, MP, NP, G);
    static_assert((5 * G) % LP == 0, "a column is a whole number of cycles");
    constexpr uint32_t dbg = D2D_PX_ABL;
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {   // tap fragments: L2 -> LDS once per block
        const uint4* s = reinterpret_cast<const uint4*>(a.tables);
        uint4* dl = reinterpret_cast<uint4*>(smem);
        for (uint32_t i = tid; i < (uint32_t)NSLOT * (PX_FRAG_BYTES / 16); i += blockDim.x) dl[i] = s[i];
    }
    __syncthreads();

    const uint32_t C = a.epi.channels, Ct = a.in_channels;
    uint32_t file, grp;
    row_to_file_group(blockIdx.y, gridDim.y, a.ngroups, gridDim.x, file, grp);
    const uint32_t cbase = grp * a.cw;
    const uint32_t cwn = min(a.cw, C - cbase);                       // channels of this group (an odd count leaves a single)
    const StreamJob* jobs = a.jobs + (size_t)file * C + cbase;
    const StreamJob j0 = jobs[0];                                    // in, L, e0, n0, nout, out are common to a file's channels
    const uint32_t nout = j0.nout;
    if (nout == 0) return;
    const uint64_t m0 = j0.n0;
    const uint64_t T0 = m0 / (uint32_t)TILE, T1 = (m0 + nout - 1) / (uint32_t)TILE;
    const uint32_t ntiles = (uint32_t)(T1 - T0) + 1u;
    uint8_t* wbase = smem + a.off_waves + wave * a.wave_lds;         // [stream buffer per channel of the group | output slice [channel][TILE] dwords]
    int32_t* ob = reinterpret_cast<int32_t*>(wbase + a.off_out);
    const uint32_t n = lane & 31, kh = lane >> 5;

    // planar layouts with power-of-two blocks: 16-byte loads; anything else (history, ragged blocks, call edges): byte gathers
    const uint32_t Bsz = a.B, Lcall = (uint32_t)j0.L;
    const bool pow2B = Bsz >= 16 && (Bsz & (Bsz - 1)) == 0;
    const uint32_t bshift = pow2B ? 31 - __builtin_clz(Bsz) : 0;
    const uint32_t full_bytes = pow2B ? (Lcall >> bshift) << bshift : 0;
    const bool fast_layout = pow2B && (uint64_t)full_bytes * Ct < (1ull << 32);

    const uint8_t* tp16 = smem + 16u * lane;
    const uint8_t* tp8 = smem + 1024u + 8u * lane;
    uint32_t kmA = 0x11111111u, kmB = 0x22222222u;
    asm volatile("" : "+v"(kmA), "+v"(kmB));
    int scA = 0x7f7f7f7f, scB = (int)0x82828282u;                     // e8m0 scales: A x 1, B x 8 (every product becomes an integer)
    asm volatile("" : "+v"(scA), "+v"(scB));
    // the accumulators start from zero (an inline constant: no registers) and hold sum 2 Q b; the -2^S that makes it v = 2 sum Q b - 2^S = sum Q s
    // rides in the recombination: hi = S3 + 32 S4 - 2^(S-15), exact in f32 (px_exact)
    const px_v16f cinit = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float kNegBias = -(float)(1 << (a.S - 15));

    const int F = a.fbits;
    const float k32 = 32.0f, k1024 = 1024.0f;
    const double kCg = ldexp(a.epi.bits == 32 ? a.epi.gain : a.epi.scale, -a.S);     // x = fl(v * kCg): the oracle's y * scale (y = v 2^-S exactly)
    uint32_t vmax[2] = {0u, 0u};
    const uint32_t SBY = a.epi.sample_bytes, fb = SBY * C;

    // a tile's staged image starts at the 16-aligned byte a0 of the call, its column 0 at bit `obit` of the image
    auto tile_br = [&](uint32_t t) -> int64_t {          // first bit of column 0's window, call-relative
        const uint64_t mT = (T0 + t) * (uint32_t)TILE;    // first output of the tile: a multiple of LP
        return (int64_t)(mT / (uint32_t)LP * (uint32_t)MP) + a.D - (NP - 1) - 8 * j0.e0;
    };
    // staging: 16-byte chunks of every channel of the group, as they lie in the call's buffer, requested a whole tile ahead
    u32x4 pf[2][PF];
    const uint32_t chf[2] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)jobs[0].ch), (uint32_t)__builtin_amdgcn_readfirstlane((int)jobs[cwn - 1].ch)};
    // dither keys of the group's channels (uniform; read once: a load inside the tile loop would wait for the prefetch in front of it)
    uint32_t rkeys[2], rsteps[2], rlo0s[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const StreamJob* jc = jobs + ((uint32_t)c < cwn ? c : 0);
        rkeys[c] = (uint32_t)__builtin_amdgcn_readfirstlane((int)jc->rng_key);
        rsteps[c] = (uint32_t)__builtin_amdgcn_readfirstlane((int)jc->rng_kstep);
        rlo0s[c] = (uint32_t)__builtin_amdgcn_readfirstlane((int)jc->rng_lo0);
    }
    auto tile_a0 = [&](uint32_t t) -> int32_t { return (int32_t)((tile_br(t) >> 3) & ~(int64_t)15); };
    // FAST tiles: every chunk lies inside the call's full power-of-two blocks -- one straight block of 16-byte loads into the prefetch
    // registers, nothing else (a gather path that met this one at a join made the compiler copy the registers, i.e. wait, right there)
    // IL (a.il2: byte-interleaved stereo -- DFF files, the reference CLI's default -f I -- both channels converted): the tile's frames come
    // as they lie in memory, 2 NCHK pieces of 16 bytes = eight frames each, in the same registers; two v_perm_b32 per channel pull a piece apart
    const bool il = a.il2 != 0;
    auto is_fast = [&](uint32_t t) -> bool {
        const int32_t a0 = tile_a0(t);
        return a0 >= 0 && (il ? (uint32_t)a0 + 16u * (uint32_t)NCHK <= Lcall : fast_layout && (uint32_t)a0 + 16u * (uint32_t)NCHK <= full_bytes);
    };
    auto issue = [&](uint32_t t) {
        const uint32_t a0 = (uint32_t)tile_a0(t);
        if (il) {
            const D2D_GLOBAL uint8_t* src = as_global(j0.in) + 2u * (size_t)a0;
#pragma unroll
            for (int j = 0; j < 2 * PF; ++j) {
                uint32_t k = lane + 64u * j;
                k = k < 2u * (uint32_t)NCHK ? k : 2u * (uint32_t)NCHK - 1u;
                if (dbg & 4) pf[j / PF][j % PF] = u32x4{0u, 0u, 0u, 0u};
                else pf[j / PF][j % PF] = *reinterpret_cast<D2D_GLOBAL const u32x4*>(src + 16u * k);
            }
            return;
        }
        // (channel 1 of a mono group re-reads channel 0's bytes; nothing is written from them)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const uint32_t ck = lane + 64u * i;
                const uint32_t jb = a0 + 16u * (ck < (uint32_t)NCHK ? ck : (uint32_t)NCHK - 1u);      // (lanes past the last chunk re-read it; their writes are masked)
                const uint32_t blk = jb >> bshift, off = jb & (Bsz - 1);
                if (dbg & 4) pf[c][i] = u32x4{0u, 0u, 0u, 0u};
                else pf[c][i] = *reinterpret_cast<D2D_GLOBAL const u32x4*>(as_global(j0.in) + (((uint64_t)blk * Ct + chf[c]) << bshift) + off);
            }
    };
    auto put = [&](uint32_t c, uint32_t ck, u32x4 v) {
        if (a.msb) v = u32x4{px_lsb_first(v.x), px_lsb_first(v.y), px_lsb_first(v.z), px_lsb_first(v.w)};
        *reinterpret_cast<u32x4*>(wbase + c * SB + 16u * ck) = v;
    };
    auto commit = [&]() {
        if (il) {
#pragma unroll
            for (int j = 0; j < 2 * PF; ++j) {
                const uint32_t k = lane + 64u * j;
                const u32x4 p4 = pf[j / PF][j % PF];
                u32x2 c0 = {__builtin_amdgcn_perm(p4.y, p4.x, 0x06040200u), __builtin_amdgcn_perm(p4.w, p4.z, 0x06040200u)};
                u32x2 c1 = {__builtin_amdgcn_perm(p4.y, p4.x, 0x07050301u), __builtin_amdgcn_perm(p4.w, p4.z, 0x07050301u)};
                if (a.msb) { c0 = u32x2{px_lsb_first(c0.x), px_lsb_first(c0.y)}; c1 = u32x2{px_lsb_first(c1.x), px_lsb_first(c1.y)}; }
                if (k < 2u * (uint32_t)NCHK) { *reinterpret_cast<u32x2*>(wbase + 8u * k) = c0; *reinterpret_cast<u32x2*>(wbase + SB + 8u * k) = c1; }
            }
            return;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if ((uint32_t)c >= cwn) continue;
#pragma unroll
            for (int i = 0; i < PF; ++i) if (lane + 64u * i < (uint32_t)NCHK) put((uint32_t)c, lane + 64u * i, pf[c][i]);
        }
    };
    // any other tile (the carried history in front, ragged or odd-sized blocks, the call's last bytes): gathered byte by byte, straight to LDS
    auto gather_tile = [&](uint32_t t) {
        const int32_t a0 = tile_a0(t);
#pragma unroll 1
        for (uint32_t c = 0; c < cwn; ++c)
#pragma unroll 1
            for (uint32_t ck = lane; ck < (uint32_t)NCHK; ck += 64) put(c, ck, gather_chunk(jobs + c, Ct, a.B, a.keep, a0 + (int32_t)(16u * ck)));
    };

    // ---- the chain of channel c for the lane's column (it starts at bit `cbit` of the channel's image): TP steps of 64 stream bits, group g
    // takes part in steps u0(g) .. u1(g); `hook(k)` is whatever else the wave does behind its k-th matrix instruction ----
    auto chain = [&](uint32_t c, uint32_t cbit, px_v16f (&acc)[G], auto&& hook) {
        const uint8_t* rb = wbase + c * SB + 4u * ((cbit >> 5) + kh);
        const uint32_t shn = cbit & 31u;
        if constexpr ((dbg & 1) != 0) {
#pragma unroll
            for (int g = 0; g < G; ++g) { acc[g] = cinit + (float)(lane + g); asm volatile("" : "+v"(acc[g])); }
            static_for<0, NSLOT>([&](auto kc) { hook(kc); });
            return;
        }
        // LDS reads are issued ahead of their use (stream dwords AW steps, tap fragments AF matrix instructions) and every matrix
        // instruction is fenced, so that the compiler neither hoists all the fragment reads (258 registers) nor sinks them
        constexpr int AW = 2, AF = 3;
        uint32_t D0[TP], D1[TP];
        v4i F4[NSLOT]; u32x2 F2[NSLOT];
        auto rdW = [&](auto uc) {
            constexpr int u = decltype(uc)::value;
            D0[u] = *reinterpret_cast<const uint32_t*>(rb + 8 * u); D1[u] = *reinterpret_cast<const uint32_t*>(rb + 8 * u + 4);
        };
        auto rdF = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
            F4[k] = *reinterpret_cast<const v4i*>(tp16 + PX_FRAG_BYTES * k);
            F2[k] = *reinterpret_cast<const u32x2*>(tp8 + PX_FRAG_BYTES * k);
        };
        static_for<0, (AW < TP ? AW : TP)>([&](auto uc) { rdW(uc); });
        static_for<0, (AF < NSLOT ? AF : NSLOT)>([&](auto kc) { rdF(kc); });
        px_v8i Bv = {0, 0, 0, 0, 0, 0, 0, 0};
        static_for<0, NSLOT>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            constexpr int u = px_slot_u(LP, MP, NP, G, k), g = px_slot_g(LP, MP, NP, G, k);
            if constexpr (k == 0 || px_slot_u(LP, MP, NP, G, k > 0 ? k - 1 : 0) != u) {      // the step's first matrix instruction: its operand
                if constexpr (u + AW < TP) rdW(std::integral_constant<int, u + AW>{});
                const uint32_t w = __builtin_amdgcn_alignbit(D1[u], D0[u], shn), w2 = w >> 2;
                Bv = px_v8i{(int)(w & kmA), (int)(w & kmB), (int)(w2 & kmA), (int)(w2 & kmB), 0, 0, 0, 0};
            }
            if constexpr (k + AF < NSLOT) rdF(std::integral_constant<int, k + AF>{});
            const px_v8i Av = {F4[k].x, F4[k].y, F4[k].z, F4[k].w, (int)F2[k].x, (int)F2[k].y, 0, 0};
            if constexpr (u == px_u0(LP, MP, g)) acc[g] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Av, Bv, cinit, 2, 4, 0, scA, 0, scB);
            else acc[g] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Av, Bv, acc[g], 2, 4, 0, scA, 0, scB);
            // (what rides along sits BEHIND a matrix instruction: an in-order wave that issues two back to back sits out the first one's 32 cycles)
            hook(kc);
            __builtin_amdgcn_sched_barrier(0);
        });
        // the chain ends HERE (or the compiler sinks each group's matrix instructions into the block that uses its sums)
#pragma unroll
        for (int g = 0; g < G; ++g) asm volatile("" : "+v"(acc[g]));
    };
    auto no_hook = [](auto) {};
    // (The one inline-asm instruction of this kernel's epilogues writes its result over one of its own live operands, never into a fresh register: register
    // 15 of an accumulator set belongs to no output row and is dead when the chain ends, and a temporary that an inline-asm instruction writes there gets
    // no hazard wait states -- the chain's last MFMA, still in flight, would land its zero row on top of it (d2d_kernels_mx.hip: `hold_acc`; holding the
    // sets here instead cost the four-group shapes 42-62 spilled registers).)
    // v0 = v + 2^S = 2 sum Q b of sample i of a group's accumulators: the digits S0 .. S4 are registers 5 i .. 5 i + 4 (exact integers in f32)
    auto recombine0 = [&](const px_v16f& A, int i) -> int32_t {
        const float lo = __builtin_fmaf(A[5 * i + 2], k1024, __builtin_fmaf(A[5 * i + 1], k32, A[5 * i]));
        const float hi = __builtin_fmaf(A[5 * i + 4], k32, A[5 * i + 3]);
        return (int32_t)(((uint32_t)(int32_t)hi << 15) + (uint32_t)(int32_t)lo);
    };
    const int32_t kBias = 1 << a.S;

    // ---- the epilogue of (tile, channel c) sample by sample, every case exact (clipping, rounding ties, outputs outside the call): the
    // lane's samples are outputs 3 kh + i of every group (half 1 owns two) ----
    auto epilogue_exact = [&](uint32_t c, const px_v16f (&acc)[G], int64_t nl0) {
        const uint32_t rkey = rkeys[c], rstep = rsteps[c], rlo0 = rlo0s[c];
        uint32_t vm = 0;
        static_for<0, G>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const uint32_t o = (uint32_t)OC * n + 5u * g + 3u * kh + (uint32_t)i;       // output inside the tile
                const int64_t nl = nl0 + (int64_t)o;
                const bool live = (i < 2 || kh == 0) && (uint64_t)nl < (uint64_t)nout;
                const int32_t v = recombine0(acc[g], i) - kBias;
                const uint32_t va = (uint32_t)(v < 0 ? -v : v);
                vm = max(vm, live ? va : 0u);
                int32_t rv;
                if constexpr (KIND == 4 || (dbg & 2) != 0) {
                    rv = v;
                } else {
                    uint32_t z = 0;
                    if constexpr (KIND != 0) {
                        const uint32_t nlo = (uint32_t)m0 + (uint32_t)nl;                  // lo32 of the absolute output index
                        z = nlo + rkey + (nlo < rlo0 ? rstep : 0u);
                        z ^= z >> 16; z *= 0x7feb352dU;
                        z ^= z >> 15; z *= 0x846ca68bU;
                        z ^= z >> 16;
                    }
                    if constexpr (KIND == 3) {
                        const double x = (double)v * kCg;
                        rv = a.epi.bits == 32 ? __float_as_int(finish_f32(a.epi, x, z)) : finish_int(a.epi, x, z);
                    } else {
                        // x = v * 2^-F LSB, dither in 2^-16 (2^-17) LSB, round half away from zero, clip: all integers
                        const int32_t vh = v >> F;
                        const uint32_t vl = (uint32_t)v & ((1u << F) - 1u);
                        int32_t rr;
                        if constexpr (KIND == 2) {
                            const int32_t w = (int32_t)(vl << (17 - F)) + (int32_t)(2u * (z >> 16) + 1u) - 65536;
                            const int32_t neg = (vh + (w >> 17)) >> 31;
                            rr = vh + ((w + 65536 + neg) >> 17);
                        } else {
                            int32_t w = (int32_t)(vl << (16 - F));
                            if constexpr (KIND == 1) w += (int32_t)((z & 0xFFFFu) + (z >> 16)) - 65535;
                            const int32_t neg = (vh + (w >> 16)) >> 31;
                            rr = vh + ((w + 32768 + neg) >> 16);
                        }
                        rv = min(max(rr, a.qmin_i), a.qmax_i);
                    }
                }
                if (i < 2 || kh == 0) ob[c * TILE + o] = rv;
            }
        });
        vmax[c] = max(vmax[c], vm);
    };

    // ---- the tile's samples out of the slice: nl0 = the tile's first output relative to the call's (may be negative) ----
    auto store_tile = [&](int64_t nl0) {
        if constexpr ((dbg & 8) != 0) {
        } else if constexpr (KIND == 4) {
            for (uint32_t c = 0; c < cwn; ++c) {
                D2D_GLOBAL int32_t* xs = as_global(jobs[c].xs);
                for (uint32_t i = lane; i < (uint32_t)TILE; i += 64) {
                    const int64_t nl = nl0 + (int64_t)i;
                    if ((uint64_t)nl < (uint64_t)nout) xs[nl] = ob[c * TILE + i];
                }
            }
        } else {
            uint8_t* out = reinterpret_cast<uint8_t*>(j0.out) + (size_t)j0.och * SBY;
            if (C == 2 && cwn == 2 && (SBY == 3 || SBY == 2 || SBY == 4)) {
                // whole stereo frames: a lane takes groups of four consecutive frames (24 / 16 / 32 contiguous bytes)
                for (uint32_t q = lane; 4u * q < (uint32_t)TILE; q += 64) {
                    const int64_t nl = nl0 + (int64_t)(4u * q);
                    const u32x4 Lq = *reinterpret_cast<const u32x4*>(ob + 4u * q), Rq = *reinterpret_cast<const u32x4*>(ob + TILE + 4u * q);
                    if (nl >= 0 && (uint64_t)nl + 3u < (uint64_t)nout) {
                        uint8_t* g = out + (size_t)nl * fb;
                        if (SBY == 3) {
                            *reinterpret_cast<D2D_GLOBAL u32x4_pa2*>(as_global(g)) =
                                u32x4_pa2{__builtin_amdgcn_perm(Rq.x, Lq.x, 0x04020100u), __builtin_amdgcn_perm(Lq.y, Rq.x, 0x05040201u),
                                          __builtin_amdgcn_perm(Rq.y, Lq.y, 0x06050402u), __builtin_amdgcn_perm(Rq.z, Lq.z, 0x04020100u)};
                            *reinterpret_cast<D2D_GLOBAL u32x2_pa2*>(as_global(g + 16)) =
                                u32x2_pa2{__builtin_amdgcn_perm(Lq.w, Rq.z, 0x05040201u), __builtin_amdgcn_perm(Rq.w, Lq.w, 0x06050402u)};
                        } else if (SBY == 2) {
                            *reinterpret_cast<D2D_GLOBAL u32x4_pa2*>(as_global(g)) =
                                u32x4_pa2{__builtin_amdgcn_perm(Rq.x, Lq.x, 0x05040100u), __builtin_amdgcn_perm(Rq.y, Lq.y, 0x05040100u),
                                          __builtin_amdgcn_perm(Rq.z, Lq.z, 0x05040100u), __builtin_amdgcn_perm(Rq.w, Lq.w, 0x05040100u)};
                        } else {
                            *reinterpret_cast<D2D_GLOBAL u32x4_pa2*>(as_global(g)) = u32x4_pa2{Lq.x, Rq.x, Lq.y, Rq.y};
                            *reinterpret_cast<D2D_GLOBAL u32x4_pa2*>(as_global(g + 16)) = u32x4_pa2{Lq.z, Rq.z, Lq.w, Rq.w};
                        }
                    } else {
                        const uint32_t Ls[4] = {Lq.x, Lq.y, Lq.z, Lq.w}, Rs[4] = {Rq.x, Rq.y, Rq.z, Rq.w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if ((uint64_t)(nl + k) < (uint64_t)nout) {
                                D2D_GLOBAL uint16_t* p16 = reinterpret_cast<D2D_GLOBAL uint16_t*>(as_global(out + (size_t)(nl + k) * fb));
                                if (SBY == 3) { p16[0] = (uint16_t)Ls[k]; p16[1] = (uint16_t)(((Ls[k] >> 16) & 0xFFu) | (Rs[k] << 8)); p16[2] = (uint16_t)(Rs[k] >> 8); }
                                else if (SBY == 2) { p16[0] = (uint16_t)Ls[k]; p16[1] = (uint16_t)Rs[k]; }
                                else { p16[0] = (uint16_t)Ls[k]; p16[1] = (uint16_t)(Ls[k] >> 16); p16[2] = (uint16_t)Rs[k]; p16[3] = (uint16_t)(Rs[k] >> 16); }
                            }
                        }
                    }
                }
            } else {
                // the group's samples inside the file's wider (or mono) frames
                for (uint32_t i = lane; i < (uint32_t)TILE; i += 64) {
                    const int64_t nl = nl0 + (int64_t)i;
                    if ((uint64_t)nl >= (uint64_t)nout) continue;
                    if (cwn == 2) { store_pair_in_frame(out + (size_t)nl * fb, (uint32_t)ob[i], (uint32_t)ob[TILE + i], SBY); continue; }
                    for (uint32_t c = 0; c < cwn; ++c) {
                        const uint32_t w = (uint32_t)ob[c * TILE + i];
                        uint8_t* dst = out + (size_t)nl * fb + c * SBY;
                        if (SBY == 4) { D2D_GLOBAL uint16_t* p = reinterpret_cast<D2D_GLOBAL uint16_t*>(as_global(dst)); p[0] = (uint16_t)w; p[1] = (uint16_t)(w >> 16); }
                        else if (SBY == 2) *reinterpret_cast<D2D_GLOBAL uint16_t*>(as_global(dst)) = (uint16_t)w;
                        else { D2D_GLOBAL uint8_t* p = as_global(dst); p[0] = (uint8_t)w; p[1] = (uint8_t)(w >> 8); p[2] = (uint8_t)(w >> 16); }
                    }
                }
            }
        }
    };

    auto tile_nl0 = [&](uint32_t t) -> int64_t { return (int64_t)((T0 + t) * (uint32_t)TILE - m0); };
    auto tile_cbit = [&](uint32_t t) -> uint32_t {           // the lane's column starts at this bit of the tile's staged image
        return (uint32_t)(tile_br(t) - 8 * (int64_t)tile_a0(t)) + (uint32_t)SBITS * n;
    };
    // One tile from its staged image, one thing after the other: the chains of the group's channels, their epilogues, the frames
    auto convert_tile = [&](uint32_t t) {
        const int64_t nl0 = tile_nl0(t);
        const uint32_t cbit = tile_cbit(t);
        static_for<0, 2>([&](auto cc) {
            constexpr uint32_t c = (uint32_t)decltype(cc)::value;
            if (c >= cwn) return;
            px_v16f acc[G];
            chain(c, cbit, acc, no_hook);
            epilogue_exact(c, acc, nl0);
        });
        wave_sync2();
        store_tile(nl0);
    };

    const uint32_t wstride = gridDim.x * a.nwaves, wv = blockIdx.x * a.nwaves + wave;
    // the fast tiles are a range [t_lo, t_hi) (a tile's first byte grows with its index)
    uint32_t t_lo = 0, t_hi = 0;
    if (fast_layout || il) {
        const uint32_t lim = il ? Lcall : full_bytes;
        const int64_t c0 = tile_br(0), K = (int64_t)(TILE / LP) * MP;                 // stream bits between two tiles
        t_lo = c0 >= 0 ? 0u : (uint32_t)((-c0 + K - 1) / K);
        if (t_lo > ntiles) t_lo = ntiles;
        const int64_t room = 8 * ((int64_t)lim - 16 * NCHK + 15) - c0;
        t_hi = room < 0 ? t_lo : (uint32_t)std::min<int64_t>(room / K + 1, (int64_t)ntiles);
        if (t_hi < t_lo) t_hi = t_lo;
        while (t_hi > t_lo && !is_fast(t_hi - 1)) --t_hi;
        while (t_hi < ntiles && is_fast(t_hi)) ++t_hi;
        while (t_lo < t_hi && !is_fast(t_lo)) ++t_lo;
    }
    for (uint32_t t = wv; t < ntiles; t += wstride) {
        if (t >= t_lo && t < t_hi) continue;
        wave_sync2();
        gather_tile(t);
        wave_sync2();
        convert_tile(t);
    }
    uint32_t t = wv;
    while (t < t_lo) t += wstride;

    // PIPE (a channel pair, the all-integer requantisers): the epilogue of one chain rides on the next, cut into jobs behind its matrix
    // instructions --
    //     region A (tile t):  chain of channel 0  ||  requantise channel 1 of the tile before; then that tile's frames leave
    //     region B (tile t):  chain of channel 1  ||  requantise channel 0 of tile t
    // in a branch-free form that is valid for a tile in which nothing clips, no rounding is an exact tie, the dither counter does not wrap
    // and every output belongs to the call; per tile the lane keeps the extremes of v and the least tie distance, one ballot after the
    // region decides, and a tile that fails is redone by epilogue_exact from the accumulators the jobs just read (they are still live).
    constexpr bool PIPE = KIND <= 2 && (dbg & 3) == 0;
    if constexpr (PIPE) {
        if (cwn == 2) {
            constexpr int NS = 3 * G;                                  // sample slots per lane and channel (half 1's third slot of a group repeats its second: the table holds output 4 twice)
            constexpr int JPS = KIND == 0 ? 2 : 3;                     // jobs per sample: [hash,] recombine, finish
            constexpr int NJ = JPS * NS;
            uint32_t kF = (uint32_t)F, kSh = 16u - (uint32_t)F, kShR = 32u - (uint32_t)F, kC1 = 0x7feb352dU, kC2 = 0x846ca68bU, kTm = (uint32_t)-32767;
            int32_t kHalf = 1 << (F - 1), kNegB = -kBias;
            asm volatile("" : "+v"(kF), "+v"(kSh), "+v"(kShR), "+v"(kC1), "+v"(kC2), "+v"(kTm), "+v"(kHalf), "+v"(kNegB));
            const int32_t kSafe = (int32_t)(((uint32_t)a.qmax_i - 2u) << F);
            // where the lane's samples go in the slice: slot i of group g at base + 5 g + i; half 1's third slot goes to a dword nobody reads
            int32_t* const sl_real = ob + (uint32_t)OC * n + 3u * kh;
            int32_t* const sl_third[2] = {kh ? ob + 2 * TILE + n : sl_real, kh ? ob + 2 * TILE + n : sl_real + TILE};
            int32_t tmn[2] = {kBias, kBias}, tmx[2] = {kBias, kBias};    // running extremes of v0 = v + 2^S over the tiles the fast form served
            struct Fast { uint32_t zb, T; int32_t v0, mn, mx; uint32_t tie; };
            auto fast_begin = [&](Fast& f, uint32_t tt, uint32_t c) {
                const uint32_t first = (uint32_t)((T0 + tt) * (uint32_t)TILE);             // lo32 of the tile's first output index
                f.zb = first + rkeys[c] + (first < rlo0s[c] ? rsteps[c] : 0u) + (uint32_t)OC * n + 3u * kh;
                f.mn = kBias; f.mx = kBias; f.tie = 0xFFFFu;
            };
            auto fast_job = [&](Fast& f, auto cc, const px_v16f (&o)[G], auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int c = decltype(cc)::value;
                constexpr int i = j / JPS, g = i / 3, q = i % 3;
                constexpr int tk = j % JPS + (KIND == 0 ? 1 : 0);       // 0 hash, 1 recombine, 2 finish
                if constexpr (tk == 0) {
                    uint32_t z = f.zb + (uint32_t)(5 * g + q);
                    z ^= z >> 16; z *= kC1;
                    z ^= z >> 15; z *= kC2;
                    z ^= z >> 16;
                    if constexpr (KIND == 1) f.T = __builtin_amdgcn_sad_u16(z, 0u, kTm);    // lo16 + hi16 - 32767, units of 2^-16 LSB
                    else f.T = z >> kShR;                                                   // (2 hi16 + 1) >> (17 - F)
                    asm volatile("" : "+v"(f.T));
                } else if constexpr (tk == 1) {
                    f.v0 = recombine0(o[g], q);
                    asm volatile("" : "+v"(f.v0));
                } else {
                    const int32_t v0 = f.v0;
                    int32_t sres;
                    if constexpr (KIND == 1) {
                        // r = floor(x + d + 1/2) = (v + (T >> (16 - F))) >> F; an exact tie (the only case where round-half-away differs) has the low 16 bits of v 2^(16-F) + T zero
                        sres = v0 + ((int32_t)f.T >> kSh) + kNegB;
                        const uint32_t w = ((uint32_t)v0 << kSh) + f.T;
                        asm("v_min3_u16 %0, %0, %1, %1" : "+v"(f.tie) : "v"(w));               // (the result in the register that holds the running value: see the note above)
                    } else if constexpr (KIND == 2) {
                        sres = v0 + (int32_t)f.T + kNegB;                                  // never a tie
                    } else {
                        const int32_t v = v0 + kNegB;
                        sres = v + kHalf + (v >> 31);                                      // round half away from zero
                    }
                    if constexpr (q == 2) sl_third[c][5 * g + q] = sres >> kF;
                    else sl_real[c * TILE + 5 * g + q] = sres >> kF;
                    f.mn = min(f.mn, v0); f.mx = max(f.mx, v0);
                }
            };
            auto fast_hook = [&](Fast& f, auto c, const px_v16f (&o)[G], auto kc) {
                constexpr int k = decltype(kc)::value;
                static_for<0, NJ>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    if constexpr ((j * NSLOT) / NJ == k) fast_job(f, c, o, jc);
                });
            };
            auto fast_failed = [&](const Fast& f, uint32_t tt) -> bool {
                const int64_t nl0 = tile_nl0(tt);
                const uint32_t first = (uint32_t)((T0 + tt) * (uint32_t)TILE);
                if (nl0 < 0 || (uint64_t)nl0 + (uint32_t)TILE > (uint64_t)nout || first > 0xFFFFFFFFu - (uint32_t)TILE) return true;     // (uniform)
                const bool bad = (KIND == 1 && (f.tie & 0xFFFFu) == 0) || f.mx > kSafe + kBias || f.mn < kBias - kSafe;
                return __builtin_amdgcn_ballot_w64(bad) != 0;
            };
            auto finish_channel = [&](const Fast& f, uint32_t tt, auto cc, const px_v16f (&o)[G]) {
                constexpr int c = decltype(cc)::value;
                if (fast_failed(f, tt)) epilogue_exact((uint32_t)c, o, tile_nl0(tt));
                else { tmn[c] = min(tmn[c], f.mn); tmx[c] = max(tmx[c], f.mx); }
            };
            using C0 = std::integral_constant<int, 0>;
            using C1 = std::integral_constant<int, 1>;
            if (t < t_hi) issue(t);
            px_v16f accA[G], accB[G];                                  // channel 0's / channel 1's accumulators
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int i = 0; i < 16; ++i) accB[g][i] = 0.0f;
            bool have_prev = false;
            uint32_t pw = t;                                           // the tile whose channel 1 still waits for its epilogue
            for (; t < t_hi; t += wstride) {
                const uint32_t cbit = tile_cbit(t);
                wave_sync2();
                commit();
                if (t + wstride < t_hi) issue(t + wstride);            // the next tile's bytes are on their way while this one is converted
                wave_sync2();
                {   // ---- region A ----
                    Fast f;
                    fast_begin(f, pw, 1);
                    chain(0u, cbit, accA, [&](auto kc) { fast_hook(f, C1{}, accB, kc); });
                    if (have_prev) {
                        finish_channel(f, pw, C1{}, accB);
                        wave_sync2();
                        store_tile(tile_nl0(pw));
                        wave_sync2();
                    }
                }
                {   // ---- region B ----
                    Fast f;
                    fast_begin(f, t, 0);
                    chain(1u, cbit, accB, [&](auto kc) { fast_hook(f, C0{}, accA, kc); });
                    finish_channel(f, t, C0{}, accA);
                }
                have_prev = true; pw = t;
            }
            if (have_prev) {
                // drain: channel 1 of the wave's last tile
                Fast f;
                fast_begin(f, pw, 1);
                static_for<0, NJ>([&](auto jc) { fast_job(f, C1{}, accB, jc); });
                finish_channel(f, pw, C1{}, accB);
                wave_sync2();
                store_tile(tile_nl0(pw));
            }
            // the extremes the fast form met, as |v|
#pragma unroll
            for (int c = 0; c < 2; ++c) vmax[c] = max(vmax[c], (uint32_t)max(tmx[c] - kBias, kBias - tmn[c]));
            t = t_hi;                                                  // (nothing left for the plain loop below)
        }
    }
    if (t < t_hi) issue(t);
    for (; t < t_hi; t += wstride) {
        wave_sync2();
        commit();
        if (t + wstride < t_hi) issue(t + wstride);       // the next tile's bytes are on their way while this one is converted
        wave_sync2();
        convert_tile(t);
    }
    if constexpr (KIND == 4) return;                                   // (the noise-shaping pass keeps the peaks)
    // peak meter: |y * gain| of the largest |v| (y = v 2^-S exactly; the product rounds once, as the oracle's)
    for (uint32_t c = 0; c < cwn; ++c) {
        double p = fabs(ldexp((double)vmax[c], -a.S) * a.epi.gain);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) p = fmax(p, __shfl_xor(p, o));
        if (lane == 0 && p > 0.0)
            atomicMax(reinterpret_cast<unsigned long long*>(jobs[c].peak), (unsigned long long)__double_as_longlong(p));
    }
}

// (LP, MP, NP, G) of the tables this kernel serves (filters/filter_tables.inc: D2D_POLYS); one shape per object (Makefile: -DD2D_PX_PART=0..7),
// part 0 also holds the plain kernel, the table builder and the dispatcher
#define D2D_PX_SHAPE_0(X) X(5, 147, 751, 3)
#define D2D_PX_SHAPE_1(X) X(10, 147, 375, 4)
#define D2D_PX_SHAPE_2(X) X(20, 147, 269, 4)
#define D2D_PX_SHAPE_3(X) X(5, 294, 1501, 2)
#define D2D_PX_SHAPE_4(X) X(5, 147, 771, 3)
#define D2D_PX_SHAPE_5(X) X(10, 147, 539, 4)
#define D2D_PX_SHAPE_6(X) X(5, 294, 1541, 2)
#define D2D_PX_SHAPE_7(X) X(5, 147, 1079, 3)
#define D2D_PX_SHAPES(X) D2D_PX_SHAPE_0(X) D2D_PX_SHAPE_1(X) D2D_PX_SHAPE_2(X) D2D_PX_SHAPE_3(X) D2D_PX_SHAPE_4(X) D2D_PX_SHAPE_5(X) D2D_PX_SHAPE_6(X) D2D_PX_SHAPE_7(X)

#define D2D_PX_DECL(n) hipError_t launch_fir_px_part##n(PxArgs& a, const d2d_poly_def& p, uint32_t max_nout, uint32_t nfiles, hipStream_t s);
D2D_PX_DECL(0) D2D_PX_DECL(1) D2D_PX_DECL(2) D2D_PX_DECL(3) D2D_PX_DECL(4) D2D_PX_DECL(5) D2D_PX_DECL(6) D2D_PX_DECL(7)

template <int LP, int MP, int NP, int G, int KIND>
static hipError_t launch_px_t(PxArgs& a, uint32_t max_nout, uint32_t nfiles, hipStream_t s) {
    static KernelPrep prep;
    int dev = 0;
    const void* fn = reinterpret_cast<const void*>(&d2d_fir_px_kernel<LP, MP, NP, G, KIND>);
    hipError_t e = prep.max_dynamic_lds(fn, 160 * 1024, &dev);
    if (e != hipSuccess) return e;
    constexpr uint32_t TILE = 160u * G;
    const uint32_t C = a.epi.channels;
    a.cw = C == 1 ? 1u : 2u;
    a.ngroups = (C + a.cw - 1) / a.cw;
    a.off_waves = (uint32_t)px_nslot(LP, MP, NP, G) * PX_FRAG_BYTES;
    a.off_out = a.cw * (uint32_t)px_stream_bytes(LP, MP, NP, G);
    a.wave_lds = a.off_out + a.cw * TILE * 4u + 256u;                // (+ 64 dwords nobody reads: where the pipelined epilogue puts half 1's third slot)
    uint32_t nwaves = PX_THREADS / 64;
    while (nwaves > 1 && (size_t)a.off_waves + (size_t)nwaves * a.wave_lds > 160 * 1024) --nwaves;
    a.nwaves = nwaves;
    const size_t smem = (size_t)a.off_waves + (size_t)nwaves * a.wave_lds;
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    int ncu = 0;
    if ((e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
    const uint32_t nrows = nfiles * a.ngroups;
    const uint32_t ntiles = max_nout / TILE + 2;                     // (a call's outputs may straddle one tile more than their count fills)
    uint32_t gx = std::max(1u, (uint32_t)ncu / std::max(1u, nrows));
    gx = std::min(gx, (ntiles + nwaves - 1) / nwaves);
    hipLaunchKernelGGL((d2d_fir_px_kernel<LP, MP, NP, G, KIND>), dim3(gx, nrows), dim3(64 * nwaves), smem, s, a);
    d2d_last_launched_kernel = launched_name<LP, MP, NP, G, KIND>("d2d_fir_px_kernel");
    return hipGetLastError();
}

#define D2D_PX_LAUNCH(lp, mp, np, g)                                                                    \
    if (p.Lp == lp && p.Mp == mp && p.NP == np) {                                                       \
        if (a.to_scratch) return launch_px_t<lp, mp, np, g, 4>(a, max_nout, nfiles, s);                  \
        const bool intq = a.epi.gain == 1.0 && (a.epi.bits == 24 || a.epi.bits == 16) && a.epi.dither != 'F'; \
        if (!intq) return launch_px_t<lp, mp, np, g, 3>(a, max_nout, nfiles, s);                         \
        if (a.dkind == 1) return launch_px_t<lp, mp, np, g, 1>(a, max_nout, nfiles, s);                  \
        if (a.dkind == 2) return launch_px_t<lp, mp, np, g, 2>(a, max_nout, nfiles, s);                  \
        return launch_px_t<lp, mp, np, g, 0>(a, max_nout, nfiles, s);                                    \
    }
#define D2D_PX_PART_FN(n, shape)                                                                        \
    hipError_t launch_fir_px_part##n(PxArgs& a, const d2d_poly_def& p, uint32_t max_nout, uint32_t nfiles, hipStream_t s) { \
        shape(D2D_PX_LAUNCH)                                                                            \
        return hipErrorInvalidValue;                                                                    \
    }

#if D2D_PX_PART == 0
D2D_PX_PART_FN(0, D2D_PX_SHAPE_0)

// ---- the plain form: one output per lane, bit by bit ----
constexpr int PXP_THREADS = 256;
__global__ __launch_bounds__(PXP_THREADS) void d2d_poly_plain_kernel(PxArgs a, uint32_t span) {
    extern __shared__ __align__(16) unsigned char smem[];
    int32_t* tq = reinterpret_cast<int32_t*>(smem);                                  // Q[Lp][NP]
    double* red = reinterpret_cast<double*>(smem + (((size_t)a.Lp * a.NP * 4 + 15) & ~(size_t)15));
    uint8_t* win = reinterpret_cast<uint8_t*>(red + 4);
    const StreamJob job = a.jobs[blockIdx.y];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < a.Lp * a.NP; i += PXP_THREADS) tq[i] = reinterpret_cast<const int32_t*>(a.tables)[i];
    const uint32_t ntiles = (job.nout + PXP_THREADS - 1) / PXP_THREADS;
    const uint32_t sample_bytes = a.epi.sample_bytes, frame_bytes = sample_bytes * a.epi.channels;
    double pk = 0.0;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t mt = job.n0 + (uint64_t)tile * PXP_THREADS;
        // the tile's oldest bit: the first output's window start, call-relative
        const int64_t first_bit = (int64_t)(mt * a.Mp / a.Lp) + a.D - (int64_t)(a.NP - 1) - 8 * job.e0;
        const int64_t abeg = (first_bit >> 3) & ~(int64_t)15;
        __syncthreads();
        stage_window(win, job, a.in_channels, a.B, a.keep, abeg, span, tid, PXP_THREADS);
        __syncthreads();
        const uint32_t nl = tile * PXP_THREADS + tid;
        if (nl >= job.nout) continue;
        const uint64_t m = job.n0 + nl;
        const uint64_t tt = m * a.Mp;
        const int64_t q = (int64_t)(tt / a.Lp);
        const uint32_t rho = (uint32_t)(tt % a.Lp);
        const int32_t* g = tq + (size_t)rho * a.NP;
        const int32_t b0 = (int32_t)(q + a.D - 8 * job.e0 - 8 * abeg);             // the newest bit, relative to the staged window
        int64_t acc = 0;
        for (uint32_t j = 0; j < a.NP; ++j) {
            const uint32_t b = (uint32_t)(b0 - (int32_t)j);
            const uint32_t byte = win[b >> 3];
            const uint32_t bit = a.msb ? (byte >> (7u - (b & 7u))) & 1u : (byte >> (b & 7u)) & 1u;
            acc += bit ? (int64_t)g[j] : -(int64_t)g[j];
        }
        if (a.to_scratch) {
            job.xs[nl] = (int32_t)acc;
        } else {
            uint8_t* dst = reinterpret_cast<uint8_t*>(job.out) + (size_t)nl * frame_bytes + job.och * sample_bytes;
            pk = fmax(pk, emit_sample(a.epi, job, ldexp((double)acc, -a.S), m, dst));
        }
    }
    if (!a.to_scratch) block_peak_max(pk, job.peak, red);
}

hipError_t launch_poly_plain(PxArgs& a, const d2d_poly_def& p, uint32_t max_nout, uint32_t nstreams, hipStream_t s) {
    if (max_nout == 0 || nstreams == 0) return hipSuccess;
    a.Lp = (uint32_t)p.Lp; a.Mp = (uint32_t)p.Mp; a.NP = (uint32_t)p.NP; a.D = p.D; a.S = p.S;
    // bytes a tile's windows span: 255 outputs further on, the window itself, up to 15 bytes in front, 16 of slack
    const uint32_t span = (uint32_t)((((uint64_t)255 * p.Mp / p.Lp + p.NP + 7) / 8 + 15 + 16 + 15) & ~(uint64_t)15);
    const size_t smem = (((size_t)p.Lp * p.NP * 4 + 15) & ~(size_t)15) + 32 + span;
    static KernelPrep prep;
    hipError_t e = prep.max_dynamic_lds(reinterpret_cast<const void*>(&d2d_poly_plain_kernel), 160 * 1024);
    if (e != hipSuccess) return e;
    const uint32_t ntiles = (max_nout + PXP_THREADS - 1) / PXP_THREADS;
    const uint32_t gx = std::min(ntiles, std::max(1u, 2048u / nstreams));
    hipLaunchKernelGGL(d2d_poly_plain_kernel, dim3(gx, nstreams), dim3(PXP_THREADS), smem, s, a, span);
    d2d_last_launched_kernel = "d2d_poly_plain_kernel";
    return hipGetLastError();
}

bool px_supported(const d2d_poly_def& p) {
#define X(lp, mp, np, g) if (p.Lp == lp && p.Mp == mp && p.NP == np) return true;
    D2D_PX_SHAPES(X)
#undef X
    return false;
}
int px_groups(const d2d_poly_def& p) {
#define X(lp, mp, np, g) if (p.Lp == lp && p.Mp == mp && p.NP == np) return g;
    D2D_PX_SHAPES(X)
#undef X
    return 0;
}

// balanced base-32 digit l of v: v = sum d_l 32^l, every d in [-16, 15]
static int px_digit32(int64_t v, int l) {
    int dd = 0;
    for (int i = 0; i <= l; ++i) {
        dd = (int)(((v + 16) & 31) - 16);
        v = (v - dd) / 32;
    }
    return dd;
}
// e2m3 code of x (a multiple of 1/8 up to 2, of 1/4 up to 4)
static uint32_t px_e2m3(double x) {
    const uint32_t sg = x < 0 ? 32u : 0u;
    const double ax = fabs(x);
    for (uint32_t c = 0; c < 32; ++c) {
        const uint32_t e = c >> 3, mm = c & 7;
        const double v = e ? (1.0 + mm / 8.0) * (double)(1 << (e - 1)) : mm * 0.125;
        if (v == ax) return ax == 0 ? 0u : (sg | c);
    }
    fprintf(stderr, "d2d: %g is not an e2m3 number\n", x);
    abort();
}

// The kernel recombines v = lo + 2^15 hi with lo = S0 + 32 S1 + 2^10 S2 and hi = S3 + 32 S4 in f32: exact while every value that can occur
// stays below 2^24; a digit sum over any subset of a window's bits is bounded by the sum of the digits' magnitudes (per phase).
bool px_exact(const d2d_poly_def& p) {
    if (p.S < 20 || p.S > 30) return false;
    for (int ph = 0; ph < p.Lp; ++ph) {
        int64_t sa[5] = {0, 0, 0, 0, 0}, sq = 0;
        for (int j = 0; j < p.NP; ++j) {
            const int64_t q = p.q[(size_t)ph * p.NP + j], q2 = 2 * q;
            if (q2 > 16236247 || q2 < -17318416) return false;            // 2 Q has to fit five digits
            sq += q;
            for (int l = 0; l < 5; ++l) { const int d = px_digit32(q2, l); sa[l] += d < 0 ? -d : d; }
        }
        if (sq != ((int64_t)1 << p.S)) return false;                       // the accumulators' start value assumes unity DC gain per phase
        const int64_t lo = sa[0] + 32 * sa[1] + 1024 * sa[2], hi = sa[3] + 32 * (sa[4] + ((int64_t)1 << (p.S - 20)));
        if (lo >= (1 << 24) || hi >= (1 << 24)) return false;
    }
    return true;
}

// Tap fragments [slot (step u, group g) in issue order][64 lanes x 16 bytes | 64 lanes x 8 bytes].  A lane l = matrix row l & 31, K half
// l >> 5; its element e (a 6-bit e2m3 code at bits [6e, 6e+6) of the lane's 192) meets B register e >> 3, nibble e & 7 = bit 4 (e & 7) + (e >> 3)
// of the lane half's dword = bit x = 64 u + 32 (l >> 5) + that of the column's window, which arrives as 0.5 (even register) or 1.0 (odd).
// D row i lands in lane half (i >> 2) & 1, register 4 (i >> 3) + (i & 3) = 5 i' + digit: output 3 half + i' of the group.  Output o of the
// column (o = 5 g + 3 half + i') meets window bit x with tap j = q_o + NP - 1 - x of phase (o Mp) mod Lp.
std::vector<int8_t> build_px_tables(const d2d_poly_def& p) {
    const int G = px_groups(p), LP = p.Lp, MP = p.Mp, NP = p.NP;
    const int TP = px_tp(LP, MP, NP, G), NSLOT = px_nslot(LP, MP, NP, G);
    std::vector<int8_t> t((size_t)NSLOT * PX_FRAG_BYTES, 0);
    for (int u = 0; u < TP; ++u)
        for (int g = 0; g < G; ++g) {
            if (!px_active(LP, MP, NP, u, g)) continue;
            int8_t* fbp = &t[(size_t)px_slot(LP, MP, NP, G, u, g) * PX_FRAG_BYTES];
            for (int l = 0; l < 64; ++l) {
                const int row = l & 31, kh = l >> 5;
                const int half = (row >> 2) & 1, rr = 4 * (row >> 3) + (row & 3);
                uint32_t regs[6] = {0, 0, 0, 0, 0, 0};
                const int ii = rr / 5, dg = rr % 5;
                const int og = 3 * half + ii < 5 ? 3 * half + ii : 4;         // (half 1's third slot repeats output 4: a real sample for the pipelined epilogue's extremes, stored nowhere)
                if (rr < 15) {
                    const int o = 5 * g + og;
                    const int qo = px_q(LP, MP, o), ph = (int)(((long long)o * MP) % LP);
                    for (int e = 0; e < 32; ++e) {
                        const int x = 64 * u + 32 * kh + 4 * (e & 7) + (e >> 3);
                        const int j = qo + NP - 1 - x;
                        if (j < 0 || j >= NP) continue;
                        const int d = px_digit32(2 * (int64_t)p.q[(size_t)ph * NP + j], dg);
                        const uint32_t code = px_e2m3(((e >> 3) & 1) ? d * 0.125 : d * 0.25);
                        for (int b = 0; b < 6; ++b) if ((code >> b) & 1) regs[(6 * e + b) >> 5] |= 1u << ((6 * e + b) & 31);
                    }
                }
                memcpy(fbp + (size_t)l * 16, regs, 16);
                memcpy(fbp + 1024 + (size_t)l * 8, regs + 4, 8);
            }
        }
    return t;
}

hipError_t launch_fir_px(PxArgs& a, const d2d_poly_def& p, uint32_t max_nout, uint32_t nfiles, hipStream_t s) {
    if (max_nout == 0 || nfiles == 0) return hipSuccess;
    a.Lp = (uint32_t)p.Lp; a.Mp = (uint32_t)p.Mp; a.NP = (uint32_t)p.NP; a.D = p.D; a.S = p.S;
    a.dkind = a.epi.dither == 'T' ? 1u : (a.epi.dither == 'R' ? 2u : 0u);
    a.fbits = p.S - ((int)a.epi.bits - 1);
    a.qmin_i = a.epi.bits == 32 ? 0 : -(1 << (a.epi.bits - 1));
    a.qmax_i = a.epi.bits == 32 ? 0 : (1 << (a.epi.bits - 1)) - 1;
    a.qsh = a.epi.bits == 20 ? 4u : 0u;
#define R(n, shape) { auto hit = [&]() -> bool { shape(XT) return false; }; if (hit()) return launch_fir_px_part##n(a, p, max_nout, nfiles, s); }
#define XT(lp, mp, np, g) if (p.Lp == lp && p.Mp == mp && p.NP == np) return true;
    R(0, D2D_PX_SHAPE_0) R(1, D2D_PX_SHAPE_1) R(2, D2D_PX_SHAPE_2) R(3, D2D_PX_SHAPE_3) R(4, D2D_PX_SHAPE_4) R(5, D2D_PX_SHAPE_5) R(6, D2D_PX_SHAPE_6) R(7, D2D_PX_SHAPE_7)
#undef XT
#undef R
    return hipErrorInvalidValue;
}
#elif D2D_PX_PART == 1
D2D_PX_PART_FN(1, D2D_PX_SHAPE_1)
#elif D2D_PX_PART == 2
D2D_PX_PART_FN(2, D2D_PX_SHAPE_2)
#elif D2D_PX_PART == 3
D2D_PX_PART_FN(3, D2D_PX_SHAPE_3)
#elif D2D_PX_PART == 4
D2D_PX_PART_FN(4, D2D_PX_SHAPE_4)
#elif D2D_PX_PART == 5
D2D_PX_PART_FN(5, D2D_PX_SHAPE_5)
#elif D2D_PX_PART == 6
D2D_PX_PART_FN(6, D2D_PX_SHAPE_6)
#else
D2D_PX_PART_FN(7, D2D_PX_SHAPE_7)
#endif

}  // namespace d2d

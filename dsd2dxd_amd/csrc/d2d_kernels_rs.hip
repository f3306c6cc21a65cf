// d2d_kernels_rs.hip -- stage B of the 48k cascade on the int8 matrix cores (gfx950), exact.
//
//   y[m] = sum_k g[phi][k] * x[i_m - k],  t = 147 m,  i_m = t div L,  phi = t mod L        (README.md:230: cascaded FIRs)
//
// Both factors are dyadic: x = X 2^-S with the exact stage-A integers X (int32, in the scratch line of the
// stream) and g = G 2^-T on the grid of filters/filter_tables.inc, so the sum is the exact integer v = sum G X (|v| < 2^61).  Here it
// is formed from int8 limbs: X ^ 0x00808080 read as four signed bytes is X - 0x808080 in balanced base-256 digits (the constant
// comes back as 0x808080 * 2^T: every phase's G sum to 2^T exactly), G has four balanced digits, and v_mfma_i32_16x16x64_i8 gives
// the sixteen digit-pair sums of an output to ONE lane:
//
//   matrix row    = (phase p of a block of four consecutive residues r = 4 rho + p, G digit a): a lane group g = lane / 16 owns
//                   the four digits of phase g
//   matrix column = cycle c (m = L c + r): 16 consecutive cycles per wave-tile
//   K             = the X samples of the column's cycle.  Write the index as m = L c + r: i_m = 147 c + b_r (b_r = 147 r div L) and
//                   phi depends on r only, so every column reads the same coefficients against its own row of samples.  A row holds
//                   X[147 c - (P-1) .. + 255] of ONE limb, staged from the scratch with any alignment the loads like, and starts on
//                   a 16-byte boundary of LDS: the 16 bytes a lane feeds to an MFMA are one aligned ds_read_b128 (an unaligned one
//                   costs 64 LDS cycles, tools/ubench/lds_unaligned.hip), the block's offset inside the row (b_4rho mod 16) sits in
//                   the coefficient fragment.
//   four MFMA chains per block, one per X limb (same coefficient fragments), 2 K steps of 64 samples for L = 40 / 80, 1 for L = 160.
//
// A lane then recombines its sixteen sums into the 64-bit v and requantises ONE output: at unit gain and 24 / 16 bits in integers --
// r = (v + bias + dither * 2^(F-16) + 2^(F-1)) >> F with F = S + T - (bits - 1) -- unless the fraction comes within 2^-28 LSB of a
// rounding boundary (ties included), where, as for every other format, it takes the f64 definition: y = (double)v * 2^-(S+T) through
// the epilogue of d2d_device.h.  Frames leave through an LDS slice, both channels of a pair together.
//
// Replaces: the second stage of the 48 kHz-family path inside Rdsd2Pcm::do_conversion (/root/reference/src/main.rs:345,429;
// README.md:230 "cascaded FIR filters"); the crate that holds it is absent from the reference.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "d2d_device.h"
#include "d2d_filters.h"
#include "d2d_launch.h"

namespace d2d {

typedef int v4i_rs __attribute__((ext_vector_type(4)));
typedef int32_t i32x4_rs __attribute__((ext_vector_type(4)));
typedef int32_t i32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
typedef uint32_t u32x2_a2 __attribute__((ext_vector_type(2), aligned(2)));

constexpr int RS2_NCOL = 16;            // matrix columns per wave-tile: 16 cycles of one channel, or 8 cycles of a channel pair
// bytes between two rows of a limb plane: the samples a row needs (P + 146) rounded up to 16, plus 16 or 32 so that the pitch is an
// ODD number of 16-byte slots (the 64 lanes' 16-byte reads then fall on all banks)
__host__ __device__ constexpr int rs2_rp(int P) { return 16 * ((P + 146 + 15) / 16 + (((P + 146 + 15) / 16) % 2 ? 2 : 1)); }
constexpr uint32_t RS2_C0 = 0x00808080u;
constexpr int64_t RS2_GUARD = 8;        // numerator units: 2^-28 LSB at F = 31

__device__ __forceinline__ void rs2_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// DK: dither of the all-integer requantiser (0 none, 1 triangular, 2 rectangular); -1: no fast path (every output through the f64 epilogue)
// RP: row pitch of a limb plane (rs2_rp(P))
template <int NSTEP, int DK, int RP>
__global__ __launch_bounds__(512) void d2d_resample_mfma_kernel(Rs2Args a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int PLANE = RS2_NCOL * RP;
    const uint32_t L = a.L, P = a.P, NB = a.NB;
    {   // coefficient fragments and the blocks' row offsets: L2 -> LDS once per block
        const uint32_t n16 = NB * NSTEP * 64u + (NB * 4u + 15u) / 16u;
        const uint4* s = reinterpret_cast<const uint4*>(a.tables);
        uint4* dl = reinterpret_cast<uint4*>(smem);
        for (uint32_t i = tid; i < n16; i += blockDim.x) dl[i] = s[i];
    }
    __syncthreads();
    const uint8_t* frag = smem + 16u * lane;
    const uint32_t* rowoff = reinterpret_cast<const uint32_t*>(smem + (size_t)NB * NSTEP * 1024u);
    uint8_t* wbase = smem + a.off_waves + wave * a.wave_lds;      // [4 limb planes][16 rows][RP] | output slice [channel of the group][frames of the tile] dwords
    uint32_t* ob = reinterpret_cast<uint32_t*>(wbase + a.off_out);

    const uint32_t C = a.epi.channels, cw_n = a.cw;
    const uint32_t ngroups = C / cw_n;
    uint32_t file, grp;
    row_to_file_group(blockIdx.y, gridDim.y, ngroups, gridDim.x, file, grp);
    const StreamJob* jobs = a.jobs + (size_t)file * C + (size_t)grp * cw_n;
    const StreamJob j0 = jobs[0];                                  // m0, nres, n0, nout, out are common to a file's channels
    const uint32_t nres = j0.nres;
    const uint64_t c_first = j0.m0 / L;
    const uint32_t ncyc = nres ? (uint32_t)((j0.m0 + nres - 1) / L - c_first) + 1u : 0u;
    // a column = (cycle, channel of the wave's group): a channel pair shares a tile of eight cycles, a single channel takes sixteen
    const uint32_t cw_sh = cw_n == 2 ? 1u : 0u, CPT = RS2_NCOL >> cw_sh;
    const uint32_t ntiles = (ncyc + CPT - 1) / CPT;
    // everything below is 32-bit and relative to this call: stage-A indices to job.n0, outputs to m0
    const int32_t rs0 = (int32_t)((int64_t)(a.Mdn * c_first) - (int64_t)j0.n0) - (int32_t)(P - 1);     // row start of cycle c_first
    const int32_t o0 = (int32_t)((int64_t)(c_first * L) - (int64_t)j0.m0);                               // its first output, in (-L, 0]
    const int32_t jlo = -(int32_t)P, jhi = (int32_t)j0.nout - 1;
    const uint32_t col = lane & 15, kg = lane >> 4;
    const uint32_t cyc_l = col >> cw_sh, ch_l = col & (cw_n - 1u);     // the lane's cycle inside the tile and its channel inside the group
    const uint32_t tile_out = CPT * L;                                // frames per tile

    // epilogue constants
    const int F = a.fbits;
    const int64_t bias = (int64_t)RS2_C0 << a.T;
    const double yscale = ldexp(1.0, -(a.S + a.T));
    const int32_t qmin = a.epi.bits == 32 ? 0 : -(1 << (a.epi.bits - 1)), qmax = a.epi.bits == 32 ? 0 : (1 << (a.epi.bits - 1)) - 1;
    constexpr bool fast = DK >= 0;
    constexpr int dkind = DK;
    int64_t kconst = bias; int dsh = 0;
    if (fast) {
        if (dkind == 1) { kconst += (int64_t)1 << (F - 1); dsh = F - 16; }
        else if (dkind == 2) { dsh = F - 17; }
        else kconst += (int64_t)1 << (F - 1);
    }

    i32x4_rs pf[RS2_NCOL];
    const D2D_GLOBAL int32_t* xs0 = as_global(jobs[0].xs);
    const D2D_GLOBAL int32_t* xs1 = as_global(jobs[cw_n - 1u].xs);
    // row r of the tile = column r: cycle r >> cw_sh, channel r & (cw_n - 1)
    auto issue_rows = [&](uint32_t tile) {
        const int32_t rel_t = rs0 + (int32_t)(tile * CPT * a.Mdn) + 4 * (int32_t)lane;
        const bool inside = rel_t - 4 * (int32_t)lane >= jlo && rel_t - 4 * (int32_t)lane + (int32_t)((CPT - 1) * a.Mdn) + 255 <= jhi;
        if (inside) {
#pragma unroll
            for (int r = 0; r < RS2_NCOL; ++r) {
                const D2D_GLOBAL int32_t* xs = (r & 1) && cw_n == 2 ? xs1 : xs0;
                const i32x4_a4 v = *reinterpret_cast<const D2D_GLOBAL i32x4_a4*>(xs + (rel_t + (int32_t)(((uint32_t)r >> cw_sh) * a.Mdn)));
                pf[r] = i32x4_rs{v.x, v.y, v.z, v.w};
            }
        } else {
#pragma unroll 1
            for (int r = 0; r < RS2_NCOL; ++r) {
                const D2D_GLOBAL int32_t* xs = (r & 1) && cw_n == 2 ? xs1 : xs0;
                int32_t e[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int32_t j = rel_t + (int32_t)(((uint32_t)r >> cw_sh) * a.Mdn) + k;
                    const int32_t x = xs[min(max(j, jlo), jhi)];          // always a valid address
                    e[k] = (j >= jlo && j <= jhi) ? x : 0;                 // samples that do not exist only meet outputs that are not stored
                }
                const i32x4_rs v = {e[0], e[1], e[2], e[3]};
#pragma unroll
                for (int q = 0; q < RS2_NCOL; ++q) if (q == r) pf[q] = v;
            }
        }
    };
    auto write_rows = [&]() {
        if (4u * lane < (uint32_t)RP) {                                    // (a row is narrower than the 256 samples a wave fetches)
#pragma unroll
            for (int r = 0; r < RS2_NCOL; ++r) {
                const uint32_t y0 = (uint32_t)pf[r].x, y1 = (uint32_t)pf[r].y, y2 = (uint32_t)pf[r].z, y3 = (uint32_t)pf[r].w;
                // 4 x 4 byte transposition: limb b of four consecutive samples in one dword; the three low limbs then flip their top bit
                const uint32_t p01l = __builtin_amdgcn_perm(y1, y0, 0x05010400u), p01h = __builtin_amdgcn_perm(y1, y0, 0x07030602u);
                const uint32_t p23l = __builtin_amdgcn_perm(y3, y2, 0x05010400u), p23h = __builtin_amdgcn_perm(y3, y2, 0x07030602u);
                uint8_t* d = wbase + (uint32_t)r * RP + 4u * lane;
                *reinterpret_cast<uint32_t*>(d) = __builtin_amdgcn_perm(p23l, p01l, 0x05040100u) ^ 0x80808080u;
                *reinterpret_cast<uint32_t*>(d + PLANE) = __builtin_amdgcn_perm(p23l, p01l, 0x07060302u) ^ 0x80808080u;
                *reinterpret_cast<uint32_t*>(d + 2 * PLANE) = __builtin_amdgcn_perm(p23h, p01h, 0x05040100u) ^ 0x80808080u;
                *reinterpret_cast<uint32_t*>(d + 3 * PLANE) = __builtin_amdgcn_perm(p23h, p01h, 0x07060302u);
            }
        }
    };

    double pk = 0.0;                                                   // the lane's channel: max |v| over its outputs
    // dither key of the lane's channel
    const uint32_t rkey = ch_l ? jobs[cw_n - 1u].rng_key : jobs[0].rng_key, rstep = ch_l ? jobs[cw_n - 1u].rng_kstep : jobs[0].rng_kstep;
    const uint32_t rlo0 = j0.rng_lo0;                                 // (the index the counter runs on is common to a file's channels)
    const uint32_t wstride = gridDim.x * a.nwaves;
    uint32_t tile = blockIdx.x * a.nwaves + wave;
    if (tile < ntiles) issue_rows(tile);
    for (; tile < ntiles; tile += wstride) {
        const int32_t o_tile = o0 + (int32_t)(tile * tile_out);
        {
            rs2_wave_sync();
            write_rows();
            if (tile + wstride < ntiles) issue_rows(tile + wstride);     // the next tile's samples are on their way while this one's matrix work runs
            rs2_wave_sync();
            const uint8_t* xrow = wbase + col * RP + 16u * kg;
            uint32_t* oslot = ob + ch_l * tile_out + cyc_l * L + kg;
            const int32_t o_lane = o_tile + (int32_t)(cyc_l * L + kg);        // the lane's output of block 0; block rho: + 4 rho
            // one block of four residues: four MFMA chains, then the lane's sixteen digit-pair sums as the 64-bit v (mod 2^64)
            auto block_sum = [&](uint32_t rho) -> uint64_t {
                const uint32_t ro = rowoff[rho];
                v4i_rs acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
                for (int s = 0; s < NSTEP; ++s) {
                    const v4i_rs A = *reinterpret_cast<const v4i_rs*>(frag + (size_t)(rho * NSTEP + s) * 1024u);
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const v4i_rs B = *reinterpret_cast<const v4i_rs*>(xrow + b * PLANE + ro + 64u * s);
                        acc[b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, acc[b], 0, 0, 0);
                    }
                }
                // acc[b][a_] = sum over the window of (G digit a_) * (X limb b): weight 2^(8 (a_ + b))
                const int32_t P0 = acc[0][0];
                const int32_t P1 = acc[0][1] + acc[1][0];
                const int32_t P2 = acc[0][2] + acc[1][1] + acc[2][0];
                const int32_t P3 = acc[0][3] + acc[1][2] + acc[2][1] + acc[3][0];
                const int32_t P4 = acc[1][3] + acc[2][2] + acc[3][1];
                const int32_t P5 = acc[2][3] + acc[3][2];
                const int32_t P6 = acc[3][3];
                const int32_t A0 = P0 + (int32_t)((uint32_t)P1 << 8), A1 = P2 + (int32_t)((uint32_t)P3 << 8);
                const uint32_t A2 = (uint32_t)P4 + ((uint32_t)P5 << 8) + ((uint32_t)P6 << 16);      // the high dword's share, mod 2^32
                return (uint64_t)(int64_t)A0 + ((uint64_t)(int64_t)A1 << 16) + ((uint64_t)A2 << 32) + (uint64_t)bias;
            };
            auto hash = [&](int32_t o) -> uint32_t {
                const uint32_t nlo = rlo0 + (uint32_t)o;                      // lo32 of the absolute output index
                uint32_t z = nlo + rkey + (nlo < rlo0 ? rstep : 0u);
                z ^= z >> 16; z *= 0x7feb352dU;
                z ^= z >> 15; z *= 0x846ca68bU;
                z ^= z >> 16;
                return z;
            };
            // every other format, and the samples the guard band sends here: the f64 definition
            auto careful_bits = [&](int64_t v, uint32_t z) -> uint32_t {
                const double y = (double)v * yscale;
                return a.epi.bits == 32 ? __float_as_uint(quantise_f32(a.epi, y, z)) : (uint32_t)quantise_int(a.epi, y, z);
            };
            int64_t vmx = 0, vmn = 0;                                          // extremes of v over the outputs that exist
            if constexpr (fast) {
                // branch-free body (two blocks per trip, so that one block's requantiser overlaps the other's matrix work); the lanes
                // whose fraction came within the guard band of a rounding boundary are redone afterwards
                bool flagged = false;
                const uint32_t gsh = F < 32 ? 32u - (uint32_t)F : 0u;          // (F >= 32: the low dword alone is a superset test)
                // INNER: every output of the tile exists and the dither counter does not wrap inside it (all but the call's first and last
                // tiles): no per-output test, the hash key is the tile's
                const bool inner = o_tile >= 0 && (uint32_t)o_tile + tile_out <= nres && rlo0 + (uint32_t)o_tile <= 0xFFFFFFFFu - tile_out;
                const uint32_t key_t = rkey + (rlo0 + (uint32_t)(o_tile < 0 ? 0 : o_tile) < rlo0 ? rstep : 0u) + rlo0;
                auto fast_one = [&](uint32_t rho, auto inner_c) {
                    constexpr bool INNER = decltype(inner_c)::value;
                    const int64_t v = (int64_t)block_sum(rho);
                    const int32_t o = o_lane + (int32_t)(4u * rho);
                    const bool live = INNER || (uint32_t)o < nres;
                    uint32_t z;
                    if constexpr (INNER) {
                        z = (uint32_t)o + key_t;
                        z ^= z >> 16; z *= 0x7feb352dU;
                        z ^= z >> 15; z *= 0x846ca68bU;
                        z ^= z >> 16;
                    } else z = hash(o);
                    int64_t t = 0;
                    if constexpr (dkind == 1) t = (int64_t)(int32_t)((z & 0xFFFFu) + (z >> 16)) - 65535;
                    else if constexpr (dkind == 2) t = (int64_t)(2u * (z >> 16) + 1u);
                    const int64_t N = v + (kconst - bias) + (t << dsh);
                    const int32_t rr = (int32_t)(N >> F);                      // |N >> F| < 2^31: v / 2^F is below 2^(bits+1)
                    flagged |= live && (((uint32_t)N + (uint32_t)RS2_GUARD) << gsh) < ((2u * (uint32_t)RS2_GUARD) << gsh);
                    oslot[4u * rho] = (uint32_t)min(max(rr, qmin), qmax);
                    vmx = live && v > vmx ? v : vmx;
                    vmn = live && v < vmn ? v : vmn;
                };
                uint32_t rho = 0;
                if (inner) {
                    for (; rho + 4 <= NB; rho += 4) { fast_one(rho, std::true_type{}); fast_one(rho + 1, std::true_type{}); fast_one(rho + 2, std::true_type{}); fast_one(rho + 3, std::true_type{}); }
                    for (; rho < NB; ++rho) fast_one(rho, std::true_type{});
                } else {
                    for (; rho < NB; ++rho) fast_one(rho, std::false_type{});
                }
                if (__builtin_amdgcn_ballot_w64(flagged) != 0) {
                    for (rho = 0; rho < NB; ++rho) {
                        const int64_t v = (int64_t)block_sum(rho);
                        const int32_t o = o_lane + (int32_t)(4u * rho);
                        if ((uint32_t)o < nres) oslot[4u * rho] = careful_bits(v, hash(o));
                    }
                }
            } else {
                D2D_GLOBAL double* ysl = a.ys ? as_global(a.ys) + (size_t)(file * C + grp * cw_n + ch_l) * a.ys_stride : nullptr;
                for (uint32_t rho = 0; rho < NB; ++rho) {
                    const int64_t v = (int64_t)block_sum(rho);
                    const int32_t o = o_lane + (int32_t)(4u * rho);
                    const bool live = (uint32_t)o < nres;
                    if (ysl) { if (live) ysl[o] = (double)v * yscale; }          // the noise-shaping pass requantises
                    else oslot[4u * rho] = live ? careful_bits(v, hash(o)) : 0u;
                    vmx = live && v > vmx ? v : vmx;
                    vmn = live && v < vmn ? v : vmn;
                }
            }
            pk = fmax(pk, fmax((double)vmx, -(double)vmn));           // correctly rounded conversions: the oracle's |(double)isum|
        }
        rs2_wave_sync();
        if (a.ys) continue;
        // ---- the tile's frames: 16 L consecutive outputs, the wave's channels side by side ----
        const uint32_t SBY = a.epi.sample_bytes, fb = SBY * C;
        uint8_t* out = reinterpret_cast<uint8_t*>(j0.out) + (size_t)j0.och * SBY;
        if (C == 2 && cw_n == 2 && (SBY == 3 || SBY == 2)) {
            for (uint32_t q = lane; 4u * q < tile_out; q += 64) {
                const int32_t o = o_tile + (int32_t)(4u * q);
                const u32x4 Lq = *reinterpret_cast<const u32x4*>(ob + 4u * q), Rq = *reinterpret_cast<const u32x4*>(ob + tile_out + 4u * q);
                if (o >= 0 && (uint32_t)o + 3u < nres) {
                    uint8_t* g = out + (size_t)(uint32_t)o * fb;
                    if (SBY == 3) {
                        *reinterpret_cast<D2D_GLOBAL u32x4_a2*>(as_global(g)) =
                            u32x4_a2{__builtin_amdgcn_perm(Rq.x, Lq.x, 0x04020100u), __builtin_amdgcn_perm(Lq.y, Rq.x, 0x05040201u),
                                     __builtin_amdgcn_perm(Rq.y, Lq.y, 0x06050402u), __builtin_amdgcn_perm(Rq.z, Lq.z, 0x04020100u)};
                        *reinterpret_cast<D2D_GLOBAL u32x2_a2*>(as_global(g + 16)) =
                            u32x2_a2{__builtin_amdgcn_perm(Lq.w, Rq.z, 0x05040201u), __builtin_amdgcn_perm(Rq.w, Lq.w, 0x06050402u)};
                    } else {
                        *reinterpret_cast<D2D_GLOBAL u32x4_a2*>(as_global(g)) =
                            u32x4_a2{__builtin_amdgcn_perm(Rq.x, Lq.x, 0x05040100u), __builtin_amdgcn_perm(Rq.y, Lq.y, 0x05040100u),
                                     __builtin_amdgcn_perm(Rq.z, Lq.z, 0x05040100u), __builtin_amdgcn_perm(Rq.w, Lq.w, 0x05040100u)};
                    }
                } else {
                    const uint32_t Ls[4] = {Lq.x, Lq.y, Lq.z, Lq.w}, Rs[4] = {Rq.x, Rq.y, Rq.z, Rq.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if ((uint32_t)(o + k) < nres) {
                            D2D_GLOBAL uint16_t* p16 = reinterpret_cast<D2D_GLOBAL uint16_t*>(as_global(out + (size_t)(uint32_t)(o + k) * fb));
                            if (SBY == 3) { p16[0] = (uint16_t)Ls[k]; p16[1] = (uint16_t)(((Ls[k] >> 16) & 0xFFu) | (Rs[k] << 8)); p16[2] = (uint16_t)(Rs[k] >> 8); }
                            else { p16[0] = (uint16_t)Ls[k]; p16[1] = (uint16_t)Rs[k]; }
                        }
                    }
                }
            }
        } else {
            for (uint32_t i = lane; i < tile_out; i += 64) {
                const int32_t o = o_tile + (int32_t)i;
                if ((uint32_t)o >= nres) continue;
                if (cw_n == 2) { store_pair_in_frame(out + (size_t)(uint32_t)o * fb, ob[i], ob[tile_out + i], SBY); continue; }
                for (uint32_t cw = 0; cw < cw_n; ++cw) {
                    const uint32_t w = ob[cw * tile_out + i];
                    uint8_t* dst = out + (size_t)(uint32_t)o * fb + cw * SBY;
                    if (SBY == 4) { D2D_GLOBAL uint16_t* p = reinterpret_cast<D2D_GLOBAL uint16_t*>(as_global(dst)); p[0] = (uint16_t)w; p[1] = (uint16_t)(w >> 16); }
                    else if (SBY == 2) *reinterpret_cast<D2D_GLOBAL uint16_t*>(as_global(dst)) = (uint16_t)w;
                    else { D2D_GLOBAL uint8_t* p = as_global(dst); p[0] = (uint8_t)w; p[1] = (uint8_t)(w >> 8); p[2] = (uint8_t)(w >> 16); }
                }
            }
        }
    }
    // peak meter: |y * gain|, per channel of the group
    for (uint32_t cw = 0; cw < cw_n; ++cw) {
        double p = ch_l == cw ? pk * yscale * a.epi.gain : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) p = fmax(p, __shfl_xor(p, o));
        if (lane == 0 && p > 0.0)
            atomicMax(reinterpret_cast<unsigned long long*>(jobs[cw].peak), (unsigned long long)__double_as_longlong(p));
    }
}

// ---- host side -------------------------------------------------------------------------------
static inline int8_t rs2_limb(int64_t v, int l) {
    int8_t dgt = 0;
    for (int i = 0; i <= l; ++i) {
        const int64_t dd = ((v + 128) & 255) - 128;
        dgt = (int8_t)dd;
        v = (v - dd) / 256;
    }
    return dgt;
}

uint32_t resamp2_nstep(const d2d_resamp_def& r) {
    int need = 0;
    for (int rho = 0; rho < r.L / 4; ++rho) {
        const int b0 = (r.Mdn * 4 * rho) / r.L, b3 = (r.Mdn * (4 * rho + 3)) / r.L;
        need = std::max(need, b3 + r.P - (b0 & ~15));
    }
    return (uint32_t)((need + 63) / 64);
}

// [L/4 blocks][NSTEP][64 lanes][16 bytes], then the blocks' row offsets (uint32, padded to 16 bytes).  A lane l = matrix row l & 15
// = (phase p = row / 4, digit a = row % 4), K group l / 16; its byte j is K slot kappa = 64 s + 16 (l / 16) + j = the row's sample
// rowoff + kappa = X[147 c - (P-1) + rowoff + kappa], which tap k = b_r + (P-1) - rowoff - kappa of residue r = 4 rho + p multiplies.
std::vector<int8_t> build_resamp2_table(const d2d_resamp_def& r) {
    const int NB = r.L / 4, NSTEP = (int)resamp2_nstep(r);
    std::vector<int8_t> t((size_t)NB * NSTEP * 1024 + (((size_t)NB * 4 + 15) & ~(size_t)15), 0);
    uint32_t* ro = reinterpret_cast<uint32_t*>(t.data() + (size_t)NB * NSTEP * 1024);
    for (int rho = 0; rho < NB; ++rho) {
        const int rowoff = ((r.Mdn * 4 * rho) / r.L) & ~15;
        ro[rho] = (uint32_t)rowoff;
        for (int s = 0; s < NSTEP; ++s)
            for (int l = 0; l < 64; ++l) {
                const int row = l & 15, p = row >> 2, a_ = row & 3, kgp = l >> 4;
                const int res = 4 * rho + p, b = (r.Mdn * res) / r.L, phase = (r.Mdn * res) % r.L;
                for (int j = 0; j < 16; ++j) {
                    const int kappa = 64 * s + 16 * kgp + j;
                    const int k = b + (r.P - 1) - rowoff - kappa;
                    if (k < 0 || k >= r.P) continue;
                    t[((size_t)(rho * NSTEP + s) * 64 + l) * 16 + j] = rs2_limb((int64_t)r.q[(size_t)phase * r.P + k], a_);
                }
            }
    }
    return t;
}

hipError_t launch_resample2(Rs2Args& a, const d2d_resamp_def& r, uint32_t max_out, uint32_t nfiles, hipStream_t s) {
    if (nfiles == 0 || max_out == 0) return hipSuccess;
    const uint32_t NSTEP = resamp2_nstep(r);
    a.L = (uint32_t)r.L; a.Mdn = (uint32_t)r.Mdn; a.P = (uint32_t)r.P; a.NB = (uint32_t)r.L / 4; a.NSTEP = NSTEP; a.T = r.T;
    if (r.L % 4 || r.P + 146 > 256 || NSTEP > 2) return hipErrorInvalidValue;
    const uint32_t C = a.epi.channels;
    a.cw = (C % 2 == 0) ? 2u : 1u;
    a.fbits = a.S + a.T - ((int)a.epi.bits - 1);
    a.dkind = a.epi.dither == 'T' ? 1u : (a.epi.dither == 'R' ? 2u : 0u);
    a.fast = (!a.ys && a.epi.gain == 1.0 && (a.epi.bits == 24 || a.epi.bits == 16) && a.epi.dither != 'F' && a.fbits >= 20 && a.fbits <= 46) ? 1u : 0u;
    a.off_waves = a.NB * NSTEP * 1024u + ((a.NB * 4u + 15u) & ~15u);
    const uint32_t RP = (uint32_t)rs2_rp(r.P);
    a.off_out = 4u * RS2_NCOL * RP + 64u;                                   // (the last row's reads run past its samples, against zero coefficients)
    a.wave_lds = a.off_out + RS2_NCOL * a.L * 4u;                           // the tile's frames: 16 cycles of one channel or 8 of a pair
    uint32_t nwaves = 8;
    while (nwaves > 1 && (size_t)a.off_waves + (size_t)nwaves * a.wave_lds > 160 * 1024) --nwaves;
    a.nwaves = nwaves;
    const size_t smem = (size_t)a.off_waves + (size_t)nwaves * a.wave_lds;
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    int dev = 0, ncu = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if ((e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
    const uint32_t nrows = nfiles * (C / a.cw);
    const uint32_t cpt = RS2_NCOL / a.cw;
    const uint32_t ntiles = (max_out / a.L + 2 + cpt - 1) / cpt;
    uint32_t gx = std::max(1u, (uint32_t)ncu / std::max(1u, nrows));
    gx = std::min(gx, (ntiles + nwaves - 1) / nwaves);
    const int dk = a.fast ? (int)a.dkind : -1;
#define RS2_LAUNCH(ns, d, rp)                                                                                              \
    if (NSTEP == ns && dk == d && RP == rp) {                                                                              \
        static KernelPrep prep;                                                                                            \
        if ((e = prep.max_dynamic_lds(reinterpret_cast<const void*>(&d2d_resample_mfma_kernel<ns, d, rp>), 160 * 1024)) != hipSuccess) return e; \
        hipLaunchKernelGGL((d2d_resample_mfma_kernel<ns, d, rp>), dim3(gx, nrows), dim3(64 * nwaves), smem, s, a);           \
        return hipGetLastError();                                                                                          \
    }
#define RS2_SHAPE(ns, rp) RS2_LAUNCH(ns, -1, rp) RS2_LAUNCH(ns, 0, rp) RS2_LAUNCH(ns, 1, rp) RS2_LAUNCH(ns, 2, rp)
    RS2_SHAPE(2, rs2_rp(96)) RS2_SHAPE(2, rs2_rp(48)) RS2_SHAPE(1, rs2_rp(32))      // B_96000, B_192000, B_384000
#undef RS2_SHAPE
#undef RS2_LAUNCH
    return hipErrorInvalidValue;
}

}  // namespace d2d

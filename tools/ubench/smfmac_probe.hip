// smfmac_probe.hip -- discovers the operand layout of v_smfmac_i32_32x32x64_i8 on gfx950 by one-hot probing:
// which (lane, register, byte) of the compressed A operand multiplies which (lane, register, byte) of the dense B operand for a
// given index word, and where the product lands in D.  Build: hipcc --offload-arch=gfx950 -O2 -o smfmac_probe smfmac_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));

// block b: A one-hot at element e = b / NV (lane e / 16, byte e % 16), index word = idxs[b % NV]
// out[b][lane][reg] = (valid << 16) | id of the B element (lane * 32 + byte index) whose product reached D[lane][reg]
__global__ void probe(unsigned* out, const unsigned* idxs, int NV) {
    const int b = blockIdx.x, e = b / NV, lane = threadIdx.x;
    const unsigned idx = idxs[b % NV];
    v4i A = {0, 0, 0, 0};
    if (lane == e / 16) A[(e % 16) / 4] = 1 << (8 * (e % 4));
    unsigned code[16], valid[16];
    for (int i = 0; i < 16; ++i) { code[i] = 0; valid[i] = 0; }
    for (int j = 0; j <= 11; ++j) {
        v8i B;
        for (int r = 0; r < 8; ++r) {
            unsigned w = 0;
            for (int by = 0; by < 4; ++by) {
                const unsigned id = (unsigned)lane * 32u + (unsigned)r * 4u + (unsigned)by;
                const unsigned bit = j == 11 ? 1u : (id >> j) & 1u;
                w |= bit << (8 * by);
            }
            B[r] = (int)w;
        }
        v16i C = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        C = __builtin_amdgcn_smfmac_i32_32x32x64_i8(A, B, C, (int)idx, 0, 0);
        for (int i = 0; i < 16; ++i) { if (j == 11) valid[i] = (unsigned)C[i]; else code[i] |= ((unsigned)C[i] & 1u) << j; }
    }
    for (int i = 0; i < 16; ++i) out[((size_t)b * 64 + lane) * 16 + i] = (valid[i] << 16) | code[i];
}

int main() {
    const unsigned idxs[] = {0x00000000u, 0x55555555u, 0xAAAAAAAAu, 0xFFFFFFFFu, 0x44444444u, 0xEEEEEEEEu, 0x000000E4u, 0xE4000000u};
    const int NV = sizeof(idxs) / sizeof(idxs[0]);
    const int NB = 1024 * NV;
    unsigned *d_out, *d_idx;
    hipMalloc(&d_out, (size_t)NB * 64 * 16 * 4); hipMalloc(&d_idx, sizeof(idxs));
    hipMemcpy(d_idx, idxs, sizeof(idxs), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(NB), dim3(64), 0, 0, d_out, d_idx, NV);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<unsigned> h((size_t)NB * 64 * 16);
    hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost);
    // per (A element, idx variant): list of (D lane, D reg) -> (B lane, B byte); D col/row are inferred from the list
    for (int v = 0; v < NV; ++v) {
        printf("== idx 0x%08x\n", idxs[v]);
        for (int e = 0; e < 1024; ++e) {
            const int b = e * NV + v;
            int n = 0; int first_dl = -1, first_dr = -1, first_bl = -1, first_bb = -1; int dreg_set = 0; int same_reg = 1, bb_same = 1; long vsum = 0;
            for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) {
                const unsigned x = h[((size_t)b * 64 + l) * 16 + i];
                if (x >> 16) {
                    const int id = x & 0xFFFF; vsum += x >> 16;
                    if (n == 0) { first_dl = l; first_dr = i; first_bl = id / 32; first_bb = id % 32; }
                    else { if (i != first_dr) same_reg = 0; if (id % 32 != first_bb) bb_same = 0; }
                    dreg_set |= 1 << i; ++n;
                }
            }
            // compact line: A (lane, byte) -> hits n; D reg; first D lane; B byte index (reg*4+byte); first B lane
            if (e < 64 || e % 16 == 0 || n != 32)
                printf("A lane %2d byte %2d: hits %2d  D reg %2d%s first D lane %2d  | B byte %2d%s first B lane %2d  vsum %ld\n", e / 16, e % 16, n, first_dr, same_reg ? " " : "*", first_dl, first_bb, bb_same ? " " : "*", first_bl, vsum);
        }
    }
    // full detail for a few elements under idx 0xE4 variants
    for (int v = 6; v < NV; ++v) for (int e = 0; e < 16; ++e) {
        const int b = e * NV + v;
        printf("-- idx 0x%08x A lane %d byte %d:", idxs[v], e / 16, e % 16);
        for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) { const unsigned x = h[((size_t)b * 64 + l) * 16 + i]; if (x >> 16) printf(" D(l%d,r%d)<-B(l%d,b%d)", l, i, (x & 0xFFFF) / 32, (x & 0xFFFF) % 32); }
        printf("\n");
    }
    return 0;
}

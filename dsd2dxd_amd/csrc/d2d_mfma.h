// d2d_mfma.h -- host view of the int8-MFMA FIR kernel (d2d_kernels_mfma.hip): geometry, table
// construction and launcher.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "d2d_filters.h"
#include "d2d_internal.h"

namespace d2d {

struct MfmaLayout {
    int M = 0, N = 0;
    int ksteps = 0;       // K steps of 32 bits over the widened window
    int phases = 8;       // output phases per matrix row
    int limbs = 4;        // int8 limbs per 32-bit tap
};

bool mfma_supported(int M, int N);
MfmaLayout mfma_layout(int M, int N);
size_t mfma_smem_bytes(const MfmaLayout& g, uint32_t channels, uint32_t sample_bytes, uint32_t* waves_per_block);
std::vector<int8_t> build_mfma_tables(const d2d_filter_def& f, const MfmaLayout& g, bool msb_first);
hipError_t launch_fir_mfma(const FirArgs& a, const MfmaLayout& g, uint32_t max_nout, uint32_t nstreams, hipStream_t s);
const char* mfma_kernel_name(const MfmaLayout& g);
void mfma_debug_stamps(unsigned long long out[8]);   // diagnostic (D2D_DBG=16)

// second-generation kernel (d2d_kernels_mfma2.hip): two phase groups per matrix column
int mfma2_pairs(int M, int N);
bool mfma2_supported(int M, int N);
size_t mfma2_smem_bytes(int M, int N, uint32_t channels, uint32_t sample_bytes, uint32_t* waves_per_block);
// `unmask0`: plane 0 of a stream dword reaches the matrix cores unmasked where the limb sums allow it (the two-group kernel);
// false: every plane masked (the pipelined kernel)
std::vector<int8_t> build_mfma2_tables(const d2d_filter_def& f, bool msb_first, bool unmask0);
// does this launch shape go to the software-pipelined kernel (d2d_kernels_mfma3.hip)?  Fixed per engine: decides the table variant.
// 0: the two-group kernel itself; 3: the int8 pipelined kernel; 5: the fp6 x fp4 kernel (d2d_kernels_mx.hip)
int mfma2_pipelined(const FirArgs& a, int M, int N);
void mfma2_debug_stamps(unsigned long long out[8]);   // diagnostic (make DIAG=1, D2D_DBG & 256)
hipError_t launch_fir_mfma2(const FirArgs& a, int M, int N, uint32_t max_nout, uint32_t nstreams, hipStream_t s);
void mfma3_debug_stamps(unsigned long long out[8]);  // diagnostic (make DIAG=1, D2D_DBG & 256): per-wave lifetimes of the pipelined kernel

}  // namespace d2d

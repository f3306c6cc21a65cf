// d2d_kernels_mfma.hip -- int8 MFMA evaluation of the 1-bit FIR (placeholder until the kernel lands)
#include "d2d_mfma.h"

namespace d2d {
bool mfma_supported(int, int) { return false; }
MfmaLayout mfma_layout(int M, int N) { MfmaLayout g; g.M = M; g.N = N; return g; }
uint32_t mfma_keep_bytes(const MfmaLayout&, int) { return 0; }
std::vector<int8_t> build_mfma_tables(const d2d_filter_def&, const MfmaLayout&, bool) { return {}; }
hipError_t launch_fir_mfma(const FirArgs&, const MfmaLayout&, uint32_t, uint32_t, hipStream_t) { return hipErrorNotSupported; }
const char* mfma_kernel_name(const MfmaLayout&) { return "d2d_fir_mfma_kernel"; }
}  // namespace d2d

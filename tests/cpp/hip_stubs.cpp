// Host-only stand-ins for the device side of libflash_attention.so, for the SANITIZER build of the C ABI's
// validation / plan-selection code (tests/test_host_api.py::test_c_abi_under_address_and_ub_sanitizers).
// fa_capi.cpp is compiled with g++ -fsanitize=address,undefined and linked against these instead of the HIP
// kernels and the HIP runtime: every launcher records that it was reached and returns success, so the test can
// drive fa_fwd / fa_bwd through every argument check and through plan selection without a GPU.  (GPU
// AddressSanitizer is not available on the pool; sanitizers run on the CPU build only.)
#include "fa_params.h"

int g_stub_launches = 0;
char g_stub_last = '-';

extern "C" const char* hipGetErrorString(hipError_t) { return "stub"; }

namespace fa {
hipError_t launch_bwd(const BwdParams&, int, hipStream_t) { ++g_stub_launches; g_stub_last = 'B'; return hipSuccess; }
hipError_t launch_bwd_generic(const BwdParams&, int, hipStream_t) { ++g_stub_launches; g_stub_last = 'b'; return hipSuccess; }
hipError_t launch_bwd_f32(const BwdParams&, hipStream_t) { ++g_stub_launches; g_stub_last = 'F'; return hipSuccess; }
hipError_t launch_fwd_extra(const FwdParams&, int, hipStream_t) { ++g_stub_launches; g_stub_last = 'E'; return hipSuccess; }
hipError_t launch_bwd_extra(const BwdParams&, int, hipStream_t) { ++g_stub_launches; g_stub_last = 'e'; return hipSuccess; }
hipError_t launch_fwd_dma(const FwdParams&, int, int, int, int, int, hipStream_t) { ++g_stub_launches; g_stub_last = 'd'; return hipSuccess; }
hipError_t launch_fwd_generic(const FwdParams&, int, hipStream_t) { ++g_stub_launches; g_stub_last = 'g'; return hipSuccess; }
hipError_t launch_fwd_f32(const FwdParams&, hipStream_t) { ++g_stub_launches; g_stub_last = 'f'; return hipSuccess; }
hipError_t launch_fwd_pipe(const FwdParams&, int, int, int, hipStream_t) { ++g_stub_launches; g_stub_last = 'p'; return hipSuccess; }
hipError_t launch_fwd_pipe16(const FwdParams&, int, int, int, hipStream_t) { ++g_stub_launches; g_stub_last = 'x'; return hipSuccess; }
hipError_t launch_fwd_kh(const FwdParams&, int, int, hipStream_t) { ++g_stub_launches; g_stub_last = 'h'; return hipSuccess; }
double probe_mfma_peak(int, int, int) { return -1.0; }
hipError_t launch_clock_sampler(unsigned long long*, int, hipStream_t) { return hipSuccess; }
}  // namespace fa

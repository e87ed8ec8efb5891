#include "common.h"
thread_local char g_knnsvc_err[512] = "";
extern "C" int knnsvc_abi_version(void) { return KNNSVC_ABI_VERSION; }
extern "C" const char* knnsvc_last_error(void) { return g_knnsvc_err; }

// Stream-placement probe (knn_svc_amd/pipeline.py: streams_overlap): a launch of `blocks` one-wave workgroups that do nothing but
// spin for `spin` shader-clock ticks each — its duration is set by how fast the stream's hardware queue gets workgroups onto CUs
// beside whatever else is being dispatched, which is what two streams of a pipeline compete for.
namespace {
__global__ __launch_bounds__(64) void probe_dispatch_kernel(int spin, int* sink) {
    const long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < spin) {}
    if (sink && threadIdx.x == 4096) *sink = 1;
}
}  // namespace
extern "C" int knnsvc_probe_dispatch(int32_t blocks, int32_t spin, void* stream) {
    KN_REQUIRE(blocks > 0 && blocks <= (1 << 22) && spin >= 0, "probe_dispatch: bad sizes");
    hipLaunchKernelGGL(probe_dispatch_kernel, dim3((unsigned)blocks), dim3(64), 0, (hipStream_t)stream, spin, (int*)nullptr);
    return knnsvc_check_launch("probe_dispatch");
}

// Listing aid (tools/isa_loops.py, tools/kernel_regs.py): instantiates a few instances of the
// pipeline kernel only, so that `hipcc -S` of this file takes seconds instead of the library's
// 90 s:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -DMOD16_EXPERIMENTS
//        -DMOD16_NO_FUSED_FINAL [-DISA_MODES="kStreamRaw,kStreamTotals"] -o x.s tools/isa_instances.hip
#include <hip/hip_runtime.h>
#include "../mod16_amd/csrc/mod16_kernels.hpp"
#include "../mod16_amd/csrc/mod16_stream.hpp"
namespace mod16 {
template __global__ void et_stream_kernel<double, kStreamTotals, true, true>(const StreamArgs<double>);
template __global__ void et_stream_kernel<double, kStreamRaw, true, true>(const StreamArgs<double>);
template __global__ void et_stream_kernel<double, kStreamRawTotalHours, true, true>(const StreamArgs<double>);
#ifdef ISA_F32
template __global__ void et_stream_kernel<float, kStreamTotals, true, true>(const StreamArgs<float>);
template __global__ void et_stream_kernel<float, kStreamTotalsMixed, true, true>(const StreamArgs<float>);
template __global__ void et_stream_kernel<float, kStreamTotalsMixed, true, false>(const StreamArgs<float>);
template __global__ void et_stream_kernel<float, kStreamRawMixed, true, true>(const StreamArgs<float>);
template __global__ void et_stream_kernel<float, kStreamRawTotalHoursMixed, true, true>(const StreamArgs<float>);
template __global__ void et_stream_kernel<float, kStreamRawTotalHours, true, true>(const StreamArgs<float>);
template __global__ void et_stream_kernel<float, kStreamRaw, true, true>(const StreamArgs<float>);
#endif
}  // namespace mod16

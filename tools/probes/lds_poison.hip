// debug probe: fill the whole LDS of every CU with a value (a kernel that reads LDS it never wrote then shows it)
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(1024) void lds_poison_kernel(float v, float* sink) {
    extern __shared__ float l[];
    const int n = 160 * 1024 / 4;
    for (int i = threadIdx.x; i < n; i += 1024) l[i] = v;
    __syncthreads();
    if (sink && l[(threadIdx.x * 37) % n] == 12345.f) sink[0] = 1.f;   // keep the stores alive
}
extern "C" int lds_poison(float v, float* sink, void* stream) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(lds_poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(lds_poison_kernel, dim3(256 * 4), dim3(1024), 160 * 1024, (hipStream_t)stream, v, sink);
    return (int)hipGetLastError();
}

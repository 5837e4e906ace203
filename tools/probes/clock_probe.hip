// clock_probe.hip — what shader clock is the part running at RIGHT NOW?  One wave per CU spins ~20 us of dependent FMAs and
// reports (shader cycles, 100 MHz wall ticks); launched on the caller's stream right behind the kernel under study
// (tools/inframe_gap.py).  hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libclock_probe.so clock_probe.hip
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(64) void clock_probe_kernel(long long* out, int iters) {
    float a = (float)threadIdx.x;
    const long long t0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < iters; ++i) a = __builtin_fmaf(a, 1.0001f, 0.5f);
    const long long c1 = clock64(), t1 = wall_clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = t1 - t0; }
    if (a == 12345.f) out[0] = 0;
}
extern "C" int clock_probe(long long* out, int blocks, int iters, void* stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, out, iters);
    return (int)hipGetLastError();
}

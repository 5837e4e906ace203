// optim.hip — the Adam update of the training loop (train_KVNet.py:228-232: optim.Adam(model_KVnet.parameters(), lr, betas=(.9, .999));
// train_utils/train_KVNet.py:153 optimizer_KV.step()) as ONE kernel type over all 459 parameter tensors.
//
// torch.optim.Adam's default path runs six _foreach_ launches per 30-odd-tensor slab of the parameter list (~50 kernel launches,
// 0.47 ms per iteration at config 4 — the 21 MB of state is a 30 us stream).  Here a launch takes up to 48 tensors BY VALUE in its
// kernel arguments (pointers of parameter / gradient / both moments / step counter, element counts, chunk prefix sums — 2.3 KB of
// kernarg), so nothing is uploaded and the launches can be captured into a hipGraph as they are; a workgroup walks 2,048-element
// chunks of the slab.  459 tensors = 10 launches.
// Arithmetic = torch.optim.adam._single_tensor_adam (non-capturable form), operation for operation in fp32:
//     g' = g (+ weight_decay * p);  m += (g' - m) * (1 - beta1);  v = v * beta2 + (1 - beta2) * g' * g'
//     p -= (lr / (1 - beta1^t)) * (m / (sqrt(v) / sqrt(1 - beta2^t) + eps))
// with the scalar factors formed in double on the device from the tensor's own step counter t (a device float, incremented by
// a second tiny launch: parameters that got no gradient in a step — the K-Net on a first frame — are simply not in the list
// and keep their count, as in torch).
#include "common.hpp"

namespace nrgbd {

constexpr int kAdamSlab = 48, kAdamChunk = 2048;

struct AdamSlab {
    float* p[kAdamSlab];
    const float* g[kAdamSlab];
    float* m[kAdamSlab];
    float* v[kAdamSlab];
    float* step[kAdamSlab];
    int first_chunk[kAdamSlab + 1];
    int n[kAdamSlab];
    int nt;
};

__global__ __launch_bounds__(256) void adam_kernel(const AdamSlab a, double lr, double beta1, double beta2, double eps, double wd, int maximize) {
    const float w1 = (float)(1.0 - beta1), b2 = (float)beta2, w2 = (float)(1.0 - beta2), epsf = (float)eps, wdf = (float)wd;
    const int nchunks = a.first_chunk[a.nt];
    int ti = 0;
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        while (c >= a.first_chunk[ti + 1]) ++ti;            // uniform; chunks of a workgroup come in increasing order
        const double t = (double)a.step[ti][0] + 1.0;
        const float step_size = (float)(lr / (1.0 - pow(beta1, t)));
        const float bc2_sqrt = (float)sqrt(1.0 - pow(beta2, t));
        float* __restrict__ p = a.p[ti];
        const float* __restrict__ g = a.g[ti];
        float* __restrict__ m = a.m[ti];
        float* __restrict__ v = a.v[ti];
        const int n = a.n[ti], base = (c - a.first_chunk[ti]) * kAdamChunk;
#pragma unroll
        for (int j = 0; j < kAdamChunk / 256; ++j) {
            const int i = base + j * 256 + (int)threadIdx.x;
            if (i < n) {
                float gi = maximize ? -g[i] : g[i];
                const float pi = p[i];
                if (wdf != 0.f) gi = gi + wdf * pi;
                const float mi = m[i] + (gi - m[i]) * w1;
                const float vi = v[i] * b2 + w2 * gi * gi;
                m[i] = mi; v[i] = vi;
                const float denom = sqrtf(vi) / bc2_sqrt + epsf;
                p[i] = pi - step_size * (mi / denom);
            }
        }
    }
}

__global__ void adam_count_kernel(const AdamSlab a) {
    const int i = threadIdx.x;
    if (i < a.nt) a.step[i][0] += 1.f;
}

}  // namespace nrgbd

extern "C" int nrgbd_adam_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                               float* const* steps, const long* numel, int ntensors, double lr, double beta1, double beta2, double eps,
                               double weight_decay, int maximize, void* stream) {
    using namespace nrgbd;
    if (ntensors < 0) return NRGBD_E_SHAPE;
    if (ntensors == 0) return NRGBD_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !steps || !numel) return NRGBD_E_NULL;
    for (int t0 = 0; t0 < ntensors; t0 += kAdamSlab) {
        AdamSlab a;
        a.nt = ntensors - t0 < kAdamSlab ? ntensors - t0 : kAdamSlab;
        int chunks = 0;
        for (int i = 0; i < a.nt; ++i) {
            const long n = numel[t0 + i];
            if (!params[t0 + i] || !grads[t0 + i] || !exp_avg[t0 + i] || !exp_avg_sq[t0 + i] || !steps[t0 + i]) return NRGBD_E_NULL;
            if (n <= 0 || n > (1L << 30)) return NRGBD_E_SHAPE;
            a.p[i] = params[t0 + i]; a.g[i] = grads[t0 + i]; a.m[i] = exp_avg[t0 + i]; a.v[i] = exp_avg_sq[t0 + i];
            a.step[i] = steps[t0 + i];
            a.n[i] = (int)n;
            a.first_chunk[i] = chunks;
            chunks += (int)((n + kAdamChunk - 1) / kAdamChunk);
        }
        for (int i = a.nt; i <= kAdamSlab; ++i) a.first_chunk[i] = chunks;
        for (int i = a.nt; i < kAdamSlab; ++i) { a.p[i] = nullptr; a.g[i] = nullptr; a.m[i] = nullptr; a.v[i] = nullptr; a.step[i] = nullptr; a.n[i] = 0; }
        const int grid = chunks < 1024 ? chunks : 1024;
        hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, lr, beta1, beta2, eps, weight_decay, maximize);
        hipLaunchKernelGGL(adam_count_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
        NRGBD_CHECK_LAUNCH();
    }
    return NRGBD_OK;
}

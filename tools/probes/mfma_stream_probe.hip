// mfma_stream_probe.hip — how well can ONE wave per SIMD keep the fp32 matrix pipe busy through the Winograd consumers' operand
// stream, and what do TWO waves per SIMD get?    Stand-alone: hipcc --offload-arch=gfx950 -O3 -o probe ...
// Every workgroup owns a CU (100 KB of LDS); a "stage" = 16 transform points, each: A operands from LDS (ds_read_b128), one weight
// line from global memory (1 KB per wave, 7 points ahead in an 8-slot register ring), MFMAs v_mfma_f32_16x16x4_f32.
//   mode 1: 4 waves (one per SIMD), 2 row blocks per wave: 2 LDS reads + 8 MFMAs per point  (what wino_pc.hip's consumers do)
//   mode 2: 8 waves (two per SIMD), 1 row block per wave:  1 LDS read  + 4 MFMAs per point  (the proposed split)
// Caveat: every variant is its own compilation; some carry accumulator copies (v_accvgpr_mov) the real kernels do not have, so
// differences of a few per cent between variants are not significant — confirm on the real kernel (tools/bench_wino.py).
// Output: microseconds per stage and the fraction of the 128 x 32-cycle MFMA time per SIMD at the measured clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int RB, int ABL, int POS = 1, int SPLIT = 0>   // POS / SPLIT (with ABL & 128): MFMA gap of the weight load; second LDS read moved into gap 0
// ABL bits: 1 = no stage barrier, 2 = no weight loads (ring filled once), 4 = no LDS reads, 8 = no sched_barrier pin,
                             // 128 = the 16-byte weight load issued in the third MFMA gap of the point instead of at its top,
                             // 64 = the weight line from LDS (ds_read_b128) instead of global memory,
                             // 16 = the weight line as four 4-byte loads (lane-contiguous layout), one per MFMA gap; 32 = as two 8-byte loads
__global__ __launch_bounds__(RB == 2 ? 256 : 512) void probe(const f32x4* __restrict__ w, float* __restrict__ out, int stages, long long* clk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 24 * 1024; i += blockDim.x) lds[i] = 1.0f + (float)(i & 7);
    __syncthreads();
    f32x4 acc[16][RB];
#pragma unroll
    for (int x = 0; x < 16; ++x)
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[x][r] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4* wl = w + (size_t)(blockIdx.x * 8 + wave) * 64 + lane;     // this wave's weight lines: + point * 4096
    const float* a0 = lds + lane * 4 + (wave & 3) * 2048;
    const float* wf = reinterpret_cast<const float*>(w) + (size_t)(blockIdx.x * 8 + wave) * 256 + (ABL & 32 ? 2 * lane : lane);
    f32x4 Bn[8], An[4][RB];
#pragma unroll
    for (int b = 0; b < 7; ++b) Bn[b] = wl[b * 4096];
#pragma unroll
    for (int r = 0; r < RB; ++r) { An[0][r] = *reinterpret_cast<const f32x4*>(a0 + r * 256); An[1][r] = *reinterpret_cast<const f32x4*>(a0 + 512 + r * 256); }
#pragma unroll
    for (int r = 0; r < RB; ++r) { An[2][r] = An[0][r]; An[3][r] = An[1][r]; }
    Bn[7] = Bn[0];
    const long long t0 = wall_clock64();
    const long long c0 = clock64();
    for (int s = 0; s < stages; ++s) {
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            const int cur = xi & 3, nxt = (xi + 2) & 3;
#pragma unroll
            for (int r = 0; r < RB; ++r) if (!(ABL & 4) && !(SPLIT && r == 1)) An[nxt][r] = *reinterpret_cast<const f32x4*>(a0 + ((xi + 2) & 15) * 512 + r * 256);
            const size_t wpt = (size_t)(((s * 16 + xi + 7) & 63)) * 4096;
            if (!(ABL & (2 | 16 | 32 | 64 | 128))) Bn[(xi + 7) & 7] = wl[wpt];
            if (ABL & 64) Bn[(xi + 7) & 7] = *reinterpret_cast<const f32x4*>(lds + 16384 + ((xi + 7) & 7) * 1024 + (wave & 3) * 256 + lane * 4);   // weight line from an LDS ring
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int r = 0; r < RB; ++r)
                    acc[xi][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(An[cur][r][e], Bn[xi & 7][e], acc[xi][r], 0, 0, 0);
                if ((ABL & 128) && e == POS) Bn[(xi + 7) & 7] = wl[wpt];
                if (SPLIT && RB == 2 && e == 0) An[nxt][1] = *reinterpret_cast<const f32x4*>(a0 + ((xi + 2) & 15) * 512 + 256);      // the weight load two MFMA pairs after the LDS reads, not beside them
                if (ABL & 16) Bn[(xi + 7) & 7][e] = wf[wpt * 4 + e * 64];                     // 256 contiguous bytes per wave instruction
                if ((ABL & 32) && (e & 1) == 0) {
                    const float2 v = *reinterpret_cast<const float2*>(wf + wpt * 4 + e * 64);  // 512 contiguous bytes per wave instruction
                    Bn[(xi + 7) & 7][e] = v.x; Bn[(xi + 7) & 7][e + 1] = v.y;
                }
                if (!(ABL & 8)) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!(ABL & 1)) __syncthreads();
    }
    const long long c1 = clock64();
    const long long t1 = wall_clock64();
    f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int x = 0; x < 16; ++x)
#pragma unroll
        for (int r = 0; r < RB; ++r) s4 += acc[x][r];
    out[(size_t)blockIdx.x * blockDim.x + tid] = s4.x + s4.y + s4.z + s4.w;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = c1 - c0; }
}

// MFMAs only (operands loaded once, real registers), order pinned, GRP transform points interleaved k-step by k-step: a dependent
// MFMA is 2 * GRP issue slots behind its predecessor
template <int GRP>
__global__ __launch_bounds__(256) void probe_grp(const f32x4* __restrict__ w, float* __restrict__ out, int stages, long long* clk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 24 * 1024; i += blockDim.x) lds[i] = 1.0f + (float)(i & 7);
    __syncthreads();
    f32x4 acc[16][2], An[4][2], Bn[8];
#pragma unroll
    for (int x = 0; x < 16; ++x) { acc[x][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[x][1] = acc[x][0]; }
    const f32x4* wl = w + (size_t)(blockIdx.x * 8 + wave) * 64 + lane;
#pragma unroll
    for (int b = 0; b < 8; ++b) Bn[b] = wl[b * 4096];
#pragma unroll
    for (int q = 0; q < 4; ++q) { An[q][0] = *reinterpret_cast<const f32x4*>(lds + lane * 4 + q * 512); An[q][1] = *reinterpret_cast<const f32x4*>(lds + lane * 4 + q * 512 + 256); }
    const long long t0 = wall_clock64(), c0 = clock64();
    for (int s = 0; s < stages; ++s) {
#pragma unroll
        for (int xp = 0; xp < 16; xp += GRP)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int d = 0; d < GRP; ++d) {
                    const int xi = xp + d;
                    acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(An[xi & 3][0][e], Bn[xi & 7][e], acc[xi][0], 0, 0, 0);
                    acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(An[xi & 3][1][e], Bn[xi & 7][e], acc[xi][1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
        __syncthreads();
    }
    const long long c1 = clock64(), t1 = wall_clock64();
    f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int x = 0; x < 16; ++x) s4 += acc[x][0] + acc[x][1];
    out[(size_t)blockIdx.x * blockDim.x + tid] = s4.x + s4.y + s4.z + s4.w;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = c1 - c0; }
}

int main() {
    const int nwg = 256, stages = 400;
    f32x4* w; float* out; long long* clk;
    hipMalloc(&w, (size_t)64 * 4096 * sizeof(f32x4) + (size_t)nwg * 8 * 64 * sizeof(f32x4));
    hipMemset(w, 0, (size_t)64 * 4096 * sizeof(f32x4) + (size_t)nwg * 8 * 64 * sizeof(f32x4));
    hipMalloc(&out, (size_t)nwg * 512 * sizeof(float));
    hipMalloc(&clk, nwg * 2 * sizeof(long long));
    const size_t lds = 100 * 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, int threads, const char* name) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(nwg), dim3(threads), lds, 0, w, out, stages, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(nwg * 2); hipMemcpy(h.data(), clk, nwg * 2 * sizeof(long long), hipMemcpyDeviceToHost);
        double wall = 0, cyc = 0; for (int i = 0; i < nwg; ++i) { wall += h[2 * i]; cyc += h[2 * i + 1]; }
        const double ghz = cyc / (wall * 10.0), us_stage = ms * 1e3 / stages, ideal_us = 128.0 * 32.0 / (ghz * 1e3);
        printf("%-58s %.3f us per stage; clock %.2f GHz -> 128 MFMAs = %.3f us -> matrix pipe busy %5.1f %%\n", name, us_stage, ghz, ideal_us, 100.0 * ideal_us / us_stage);
    };
    run(&probe<2, 0>, 256, "1 wave/SIMD, as the consumers");
    run(&probe<2, 1>, 256, "1 wave/SIMD, no stage barrier");
    run(&probe<2, 2>, 256, "1 wave/SIMD, no weight loads");
    run(&probe<2, 4>, 256, "1 wave/SIMD, no LDS reads");
    run(&probe<2, 8>, 256, "1 wave/SIMD, scheduler free (no sched_barrier)");
    run(&probe<2, 7>, 256, "1 wave/SIMD, MFMAs only");
    run(&probe<2, 15>, 256, "1 wave/SIMD, MFMAs only, scheduler free");
    run(&probe<2, 128, 0>, 256, "1 wave/SIMD, weight load in gap 0");
    run(&probe<2, 128, 1>, 256, "1 wave/SIMD, weight load in gap 1");
    run(&probe<2, 128, 2>, 256, "1 wave/SIMD, weight load in gap 2");
    run(&probe<2, 128, 3>, 256, "1 wave/SIMD, weight load in gap 3");
    run(&probe<2, 128, 2, 1>, 256, "1 wave/SIMD, LDS reads in gaps top/0, weight load in gap 2");
    run(&probe<2, 128, 3, 1>, 256, "1 wave/SIMD, LDS reads in gaps top/0, weight load in gap 3");
    run(&probe<2, 128, 1, 1>, 256, "1 wave/SIMD, LDS reads in gaps top/0, weight load in gap 1");
    run(&probe_grp<1>, 256, "1 wave/SIMD, MFMAs only, pinned, dependency 2 slots");
    run(&probe_grp<2>, 256, "1 wave/SIMD, MFMAs only, pinned, dependency 4 slots");
    run(&probe_grp<4>, 256, "1 wave/SIMD, MFMAs only, pinned, dependency 8 slots");
    run(&probe_grp<8>, 256, "1 wave/SIMD, MFMAs only, pinned, dependency 16 slots");
    run(&probe_grp<16>, 256, "1 wave/SIMD, MFMAs only, pinned, dependency 32 slots");
    run(&probe<2, 64>, 256, "1 wave/SIMD, weight line read from LDS");
    run(&probe<1, 64>, 512, "2 waves/SIMD, weight line read from LDS");
    run(&probe<2, 16>, 256, "1 wave/SIMD, weight line = 4 x 4-byte loads in the gaps");
    run(&probe<2, 32>, 256, "1 wave/SIMD, weight line = 2 x 8-byte loads in the gaps");
    run(&probe<1, 0>, 512, "2 waves/SIMD, half the tiles each");
    run(&probe<1, 16>, 512, "2 waves/SIMD, weight line = 4 x 4-byte loads");
    run(&probe<1, 32>, 512, "2 waves/SIMD, weight line = 2 x 8-byte loads");
    run(&probe<1, 1>, 512, "2 waves/SIMD, no stage barrier");
    run(&probe<1, 2>, 512, "2 waves/SIMD, no weight loads");
    run(&probe<1, 7>, 512, "2 waves/SIMD, MFMAs only");
    return 0;
}

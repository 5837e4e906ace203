// wino_design_probe.hip — stand-alone model of the Winograd consumers' / producers' instruction streams (round 4):
// what does a stage cost when the weight lines reach the consumers through an LDS ring filled by the producers' LDS-DMA, with two
// consumer waves per SIMD (16 tiles each) and one producer wave per SIMD (12 waves, <= 168 registers each)?
//   hipcc --offload-arch=gfx950 -O3 -o wino_design_probe wino_design_probe.hip && ./wino_design_probe
// Layout 8 : waves 0-3 producers, 4-7 consumers (2 row blocks, weight line from global memory 7 points ahead) = wino_dw.hip today.
// Layout 12: waves 0-3 producers, 4-11 consumers (1 row block; A and B operands from LDS two points ahead); producers issue the
//            stage's 64 weight lines as global_load_lds_dwordx4 (16 per wave and stage), in two halves with a barrier each
//            (MID = 1) or in one batch (MID = 0, ring hazards ignored: timing only).
// PW bits: 1 = producer VALU + LDS work, 2 = producer global loads (refills), 4 = producer weight DMA.
// The producers' arithmetic is a stand-in with the real kernel's instruction mix per stage: 2 units x 5 16-byte words (8 packed
// VALU each + ReLU), 5 + 5 + 5 strip LDS accesses, a 12-read / 16-op / 8-write transform.  Values are meaningless.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define GLOAD(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
#define VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define LAUNDER(r) asm volatile("" : "+v"(r))
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

constexpr int kV = 8192, kRing = 16384, kStrip = 1280;   // floats: one V buffer, the weight ring (16 points x 4 KB), one strip

// VK: the producers' VALU instruction kind, same COUNT per stage (96): 0 = as compiled from the packed-fp32 source, 1 = v_xor_b32
// (integer), 2 = v_fma_f32, 3 = v_pk_fma_f32, 4 = v_pk_mul_f32, 5 = v_max_f32 — is it issue slots or the fp32 pipe they take?
// WSRC (layout 8): the consumers' weight lines 0 = global memory (today), 1 = none (ring filled once), 2 = an LDS ring that
// somebody filled for free (the upper bound of any LDS staging scheme)
template <int LAYOUT, int PW, int MID, int VK = 0, int WSRC = 0>
__global__ __launch_bounds__(LAYOUT == 12 ? 768 : 512) void probe(const f32x4* __restrict__ w, const f32x4* __restrict__ x, float* __restrict__ out,
                                                                   int stages, long long* clk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ring = lds;                     // [16][4][64][4] FIRST: the LDS-DMA destination (M0) stays below 64 KB
    float* Vb = lds + kRing;               // [2][kV]
    float* strips = Vb + 2 * kV;           // [4][kStrip]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * kV + kRing + 4 * kStrip; i += blockDim.x) lds[i] = 1.0f + (float)(i & 7);
    __syncthreads();
    long long t0 = 0, c0 = 0;
    float sink = 0.f;
    if (wave >= 4) {
        constexpr int RB = LAYOUT == 12 ? 1 : 2;
        const int cw = wave - 4, cg = cw & 3, th = LAYOUT == 12 ? cw >> 2 : 0;
        f32x4 acc[16][RB];
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int r = 0; r < RB; ++r) acc[i][r] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 An[4][RB], Bn[8];
        const float* a0 = Vb + th * 256 + lane * 4;
        const float* b0 = ring + cg * 256 + lane * 4;
        const f32x4* wl = w + (size_t)cg * 64 + lane;          // global weight lines: + point * 256 (f32x4 units), 64 points
        if (LAYOUT == 12) {
#pragma unroll
            for (int q = 0; q < 2; ++q) { An[q][0] = *reinterpret_cast<const f32x4*>(a0 + q * 512); Bn[q] = *reinterpret_cast<const f32x4*>(b0 + q * 1024); }
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < RB; ++r) An[q][r] = *reinterpret_cast<const f32x4*>(a0 + q * 512 + r * 256);
#pragma unroll
            for (int b = 0; b < 7; ++b) Bn[b] = wl[b * 256];
        }
        t0 = wall_clock64(); c0 = clock64();
        for (int s = 0; s < stages; ++s) {
            const float* Vc = a0 + (s & 1) * kV;
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) {
                if (LAYOUT == 12) {
                    An[(xi + 2) & 3][0] = *reinterpret_cast<const f32x4*>(Vc + ((xi + 2) & 15) * 512);
                    Bn[(xi + 2) & 3] = *reinterpret_cast<const f32x4*>(b0 + ((xi + 2) & 15) * 1024);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(An[xi & 3][0][e], Bn[xi & 3][e], acc[xi][0], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (MID && xi == 6) __syncthreads();
                } else {
#pragma unroll
                    for (int r = 0; r < RB; ++r) An[(xi + 2) & 3][r] = *reinterpret_cast<const f32x4*>(Vc + ((xi + 2) & 15) * 512 + r * 256);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
#pragma unroll
                        for (int r = 0; r < RB; ++r)
                            acc[xi][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(An[xi & 3][r][e], Bn[xi & 7][e], acc[xi][r], 0, 0, 0);
                        if (e == 1 && WSRC == 0) Bn[(xi + 7) & 7] = wl[(size_t)((s * 16 + xi + 7) & 63) * 256];
                        if (e == 1 && WSRC == 2) Bn[(xi + 7) & 7] = *reinterpret_cast<const f32x4*>(b0 + ((xi + 7) & 15) * 1024);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (xi == 14) __syncthreads();
            }
        }
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int r = 0; r < RB; ++r) s4 += acc[i][r];
        sink = s4.x + s4.y + s4.z + s4.w;
    } else {
        // ------------------------------------------------------------------ producer stand-in
        const int pw = wave;
        float* strip = strips + pw * kStrip;
        f32x4 setA[5], setB[5];
        const f32x4* xa = x + (size_t)(blockIdx.x * 4 + pw) * 4096 + lane;     // refill sources: + u * 64 + stage * 640 (wrapping)
        f32x2 sc = {1.0001f, 0.9999f}, sh = {0.001f, -0.001f};
        asm volatile("" : "+v"(sc), "+v"(sh));
#pragma unroll
        for (int u = 0; u < 5; ++u) { GLOAD(setA[u], xa + u * 64); GLOAD(setB[u], xa + (u + 5) * 64); }
        VMCNT(0);      // every load of the producers is hand-counted: the compiler must not see a pending one at the loop head
#pragma unroll
        for (int u = 0; u < 5; ++u) { LAUNDER(setA[u]); LAUNDER(setB[u]); }
        if (PW & 2) {  // the loop expects both sets in flight
#pragma unroll
            for (int u = 0; u < 5; ++u) GLOAD(setA[u], xa + u * 64);
#pragma unroll
            for (int u = 0; u < 5; ++u) GLOAD(setB[u], xa + (u + 5) * 64);
        }
        const unsigned ring_base = (unsigned)pw * 1024u;         // LDS byte address: the ring starts at the dynamic base (0)
        const char* wsrc = reinterpret_cast<const char*>(w) + (size_t)pw * 1024 + (size_t)lane * 16;
        auto unit = [&](f32x4 (&r)[5], bool combine) __attribute__((always_inline)) {
            if (!(PW & 1)) return;
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                if (VK) {
                    f32x4 v = r[u];
                    if (combine) v += *reinterpret_cast<const f32x4*>(strip + u * 256 + lane * 4);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        if (VK == 1) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(v[k & 3]) : "v"(sc.x));
                        if (VK == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k & 3]) : "v"(sc.x), "v"(sh.x));
                        if (VK == 3) { f32x2 t = (k & 1) ? v.hi : v.lo; asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(t) : "v"(sc), "v"(sh)); if (k & 1) v.hi = t; else v.lo = t; }
                        if (VK == 4) { f32x2 t = (k & 1) ? v.hi : v.lo; asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(t) : "v"(sc)); if (k & 1) v.hi = t; else v.lo = t; }
                        if (VK == 5) asm volatile("v_max_f32 %0, %1, %0" : "+v"(v[k & 3]) : "v"(sc.x));
                    }
                    *reinterpret_cast<f32x4*>(strip + u * 256 + lane * 4) = v;
                    continue;
                }
                f32x2 lo = __builtin_elementwise_fma(r[u].lo, sc, sh), hi = __builtin_elementwise_fma(r[u].hi, sc, sh);
                lo.x = fmaxf(lo.x, 0.f); lo.y = fmaxf(lo.y, 0.f); hi.x = fmaxf(hi.x, 0.f); hi.y = fmaxf(hi.y, 0.f);
                lo = lo * sc; hi = hi * sc;
                f32x4 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
                if (combine) {
                    const f32x4 old = *reinterpret_cast<const f32x4*>(strip + u * 256 + lane * 4);
                    const f32x2 cl = __builtin_elementwise_fma(lo, sh, old.lo), ch = __builtin_elementwise_fma(hi, sh, old.hi);
                    v = __builtin_shufflevector(cl, ch, 0, 1, 2, 3);
                }
                *reinterpret_cast<f32x4*>(strip + u * 256 + lane * 4) = v;
            }
        };
        auto refill = [&](f32x4 (&r)[5], int s, int off) __attribute__((always_inline)) {
            if (!(PW & 2)) return;
#pragma unroll
            for (int u = 0; u < 5; ++u) GLOAD(r[u], xa + (size_t)(((s * 10 + off + u) & 63)) * 64);
        };
        auto dma = [&](int s, int p0, int n) __attribute__((always_inline)) {
            if (!(PW & 4)) return;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < n) glds16(wsrc + (size_t)((s * 16 + p0 + j) & 63) * 4096, ring_base + (unsigned)(p0 + j) * 4096u);
        };
        auto transform = [&](int s) __attribute__((always_inline)) {
            if (!(PW & 1)) return;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            f32x4 ya[4], yb[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const f32x4 R0 = *reinterpret_cast<const f32x4*>(strip + (cc * 3 + 0) * 64 + (lane & 15) * 4);
                const f32x4 R1 = *reinterpret_cast<const f32x4*>(strip + (cc * 3 + 1) * 64 + (lane & 15) * 4 + 256);
                const f32x4 R2 = *reinterpret_cast<const f32x4*>(strip + (cc * 3 + 2) * 64 + (lane & 15) * 4 + 512);
                ya[cc] = R0 - R1; yb[cc] = R2 + R1;
            }
            float* Vq = Vb + ((s + 1) & 1) * kV + pw * 2048 + lane * 4;
            *reinterpret_cast<f32x4*>(Vq + 0 * 256) = ya[0] - ya[2];
            *reinterpret_cast<f32x4*>(Vq + 1 * 256) = ya[1] + ya[2];
            *reinterpret_cast<f32x4*>(Vq + 2 * 256) = ya[2] - ya[1];
            *reinterpret_cast<f32x4*>(Vq + 3 * 256) = ya[1] - ya[3];
            *reinterpret_cast<f32x4*>(Vq + 4 * 256) = yb[0] - yb[2];
            *reinterpret_cast<f32x4*>(Vq + 5 * 256) = yb[1] + yb[2];
            *reinterpret_cast<f32x4*>(Vq + 6 * 256) = yb[2] - yb[1];
            *reinterpret_cast<f32x4*>(Vq + 7 * 256) = yb[1] - yb[3];
        };
        for (int s = 0; s < stages; ++s) {
            if (LAYOUT == 12 && MID) {
                // half 1: DMA of this stage's points 8..15, unit A, refill A; everything but refill A has landed before the barrier
                dma(s, 8, 8);
                if (PW & 2) { if (PW & 4) VMCNT(13); else VMCNT(5); }
#pragma unroll
                for (int u = 0; u < 5; ++u) LAUNDER(setA[u]);
                unit(setA, false);
                refill(setA, s, 0);
                if (PW & 2) VMCNT(5); else VMCNT(0);
#pragma unroll
                for (int u = 0; u < 5; ++u) LAUNDER(setB[u]);
                __syncthreads();
                // half 2: DMA of the next stage's points 0..7, unit B (combined into the strip), refill B, transform
                dma(s + 1, 0, 8);
                unit(setB, true);
                refill(setB, s, 5);
                transform(s);
                if (PW & 2) VMCNT(5); else VMCNT(0);
#pragma unroll
                for (int u = 0; u < 5; ++u) LAUNDER(setA[u]);
                __syncthreads();
            } else {
                if (LAYOUT == 12) dma(s + 1, 0, 16);
                if (PW & 2) { if (LAYOUT == 12 && (PW & 4)) VMCNT(21); else VMCNT(5); }
#pragma unroll
                for (int u = 0; u < 5; ++u) LAUNDER(setA[u]);
                unit(setA, false);
                refill(setA, s, 0);
                if (PW & 2) VMCNT(5); else VMCNT(0);
#pragma unroll
                for (int u = 0; u < 5; ++u) LAUNDER(setB[u]);
                unit(setB, true);
                refill(setB, s, 5);
                transform(s);
                __syncthreads();
            }
        }
        VMCNT(0);
#pragma unroll
        for (int u = 0; u < 5; ++u) { LAUNDER(setA[u]); LAUNDER(setB[u]); sink += setA[u].x + setB[u].y; }
    }
    const long long c1 = clock64(), t1 = wall_clock64();
    out[(size_t)blockIdx.x * blockDim.x + tid] = sink;
    if (tid == 256) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = c1 - c0; }
}

int main() {
    const int nwg = 256, stages = 600;
    f32x4 *w, *x; float* out; long long* clk;
    hipMalloc(&w, (size_t)64 * 256 * sizeof(f32x4) + 65536);
    hipMemset(w, 0, (size_t)64 * 256 * sizeof(f32x4) + 65536);
    hipMalloc(&x, (size_t)(nwg * 4 + 2) * 4096 * sizeof(f32x4));
    hipMemset(x, 0, (size_t)(nwg * 4 + 2) * 4096 * sizeof(f32x4));
    hipMalloc(&out, (size_t)nwg * 768 * sizeof(float));
    hipMalloc(&clk, nwg * 2 * sizeof(long long));
    const size_t lds = (size_t)(2 * kV + kRing + 4 * kStrip) * sizeof(float);     // 148 KB
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, int threads, const char* name) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(nwg), dim3(threads), lds, 0, w, x, out, stages, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        hipError_t err = hipGetLastError();
        std::vector<long long> h(nwg * 2); hipMemcpy(h.data(), clk, nwg * 2 * sizeof(long long), hipMemcpyDeviceToHost);
        double wall = 0, cyc = 0; for (int i = 0; i < nwg; ++i) { wall += h[2 * i]; cyc += h[2 * i + 1]; }
        const double ghz = cyc / (wall * 10.0), us_stage = ms * 1e3 / stages, ideal_us = 128.0 * 32.0 / (ghz * 1e3);
        printf("%-74s %.3f us/stage  clock %.2f GHz  matrix pipe busy %5.1f %%  (%s)\n", name, us_stage, ghz, 100.0 * ideal_us / us_stage, hipGetErrorString(err));
        fflush(stdout);
    };
    run(&probe<8, 0, 0>, 512, "8 waves (today): consumers alone");
    run(&probe<8, 1, 0>, 512, "8 waves (today): + producer VALU/LDS");
    run(&probe<8, 3, 0>, 512, "8 waves (today): + producer VALU/LDS + refills");
    run(&probe<8, 3, 0, 0, 1>, 512, "8 waves: full producers, consumers WITHOUT weight loads");
    run(&probe<8, 3, 0, 0, 2>, 512, "8 waves: full producers, weight lines from a free LDS ring");
    run(&probe<8, 0, 0, 0, 2>, 512, "8 waves: consumers alone, weight lines from a free LDS ring");
    run(&probe<8, 1, 0, 1>, 512, "8 waves: + producer VALU as 96 v_xor_b32 / stage");
    run(&probe<8, 1, 0, 2>, 512, "8 waves: + producer VALU as 96 v_fma_f32 / stage");
    run(&probe<8, 1, 0, 3>, 512, "8 waves: + producer VALU as 96 v_pk_fma_f32 / stage");
    run(&probe<8, 1, 0, 4>, 512, "8 waves: + producer VALU as 96 v_pk_mul_f32 / stage");
    run(&probe<8, 1, 0, 5>, 512, "8 waves: + producer VALU as 96 v_max_f32 / stage");
    run(&probe<12, 1, 0, 1>, 768, "12 waves: + producer VALU as 96 v_xor_b32 / stage");
    run(&probe<12, 1, 0, 2>, 768, "12 waves: + producer VALU as 96 v_fma_f32 / stage");
    run(&probe<12, 1, 0, 3>, 768, "12 waves: + producer VALU as 96 v_pk_fma_f32 / stage");
    run(&probe<12, 0, 0>, 768, "12 waves: consumers alone, one barrier per stage");
    run(&probe<12, 0, 1>, 768, "12 waves: consumers alone, two barriers per stage");
    run(&probe<12, 1, 0>, 768, "12 waves: + producer VALU/LDS, one barrier");
    run(&probe<12, 1, 1>, 768, "12 waves: + producer VALU/LDS, two barriers");
    run(&probe<12, 3, 0>, 768, "12 waves: + producer VALU/LDS + refills, one barrier");
    run(&probe<12, 3, 1>, 768, "12 waves: + producer VALU/LDS + refills, two barriers");
    run(&probe<12, 4, 1>, 768, "12 waves: weight DMA only, two barriers");
    run(&probe<12, 7, 0>, 768, "12 waves: full producers (VALU/LDS + refills + weight DMA), one barrier");
    run(&probe<12, 7, 1>, 768, "12 waves: full producers (VALU/LDS + refills + weight DMA), two barriers");
    return 0;
}

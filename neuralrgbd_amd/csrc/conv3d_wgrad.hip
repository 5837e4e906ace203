// conv3d_wgrad.hip — weight gradient of the K-Net 3x3x3 convolution on the fp32 matrix cores (training), Winograd F(2x2, 3x3)
// in the image plane, direct along depth (round 4; rounds 2-3: direct in all three dimensions, 27 multiplies per voxel and
// (ci, co) — 785 us per 64 -> 64 layer at the 64x64x96 training grid = 71 % of the fp32 matrix peak, 21 % of the iteration).
//
//   dW[co][ci][kd][ky][kx] = sum_voxels dY[v][co] X[v + (kd, ky, kx) - 1][ci]                      (X zero outside the volume)
// For one depth tap kd and one 2x2 tile of output pixels this is the gradient of y = A^T [(G w G^T) . (B^T d B)] A w.r.t. w:
//   dU_kd[xi] += Z[xi] V_kd[xi]        Z = A dY_tile A^T (4x4 from 2x2),  V_kd = B^T d_kd B (4x4 patch of X at depth z + kd - 1)
//   dW_kd = G^T dU_kd G                (once, in the reduction kernel)
// i.e. 3 x 16 = 48 multiplies per 2x2 outputs and (ci, co) = 12 per voxel instead of 27: 2.25x fewer MFMAs; the transforms are
// adds on values each lane already holds (exact-algorithm fp32, rounding order only).
//
// Decomposition (persistent): grid (128 tile ranges, quadrants of the (co, ci) block), 4 waves of 256 registers per workgroup, TWO
// workgroups per CU; wave (cib, cob) owns the 16 x 16 block (co block 2*coh + cob, ci block 2*cih + cib) of ALL 48 points: 192
// accumulator registers.  (All 64 x 64 x 48 accumulators are 3,072 registers per lane, more than a CU has: hence quadrants.  A
// quadrant stages only ITS 32 input channels, so two workgroups fit a CU and one stages while the other computes — with one
// 8-wave workgroup per CU the staging, a third of a tile's time now that the MFMAs are 2.25x fewer, was exposed: 0.64 ms.)
// The workgroup walks 2 x 4 x 16-voxel tiles; the (4 x 6 x 18)-voxel halo of X is staged raw in LDS (voxel stride = channels + 8
// floats: the four tiles of a step are two voxels = 16 banks apart, so a ds_read_b32 of (tile, ci) hits 32 distinct banks per
// half-wave); a step contracts the 4 Winograd tiles (tx = 4 xg + k) of one tile row: A = Z (co x tile), B = V (tile x ci),
// both computed by the lane that feeds them — lane (i, k) transforms dY of (co = i, tile k) AND X of (ci = i, tile k).
// Partials [range][48][64][Cin] are reduced in a fixed order by a second kernel that also applies G^T . G (no atomics: bitwise
// reproducible).
#include "common.hpp"

namespace nrgbd {

typedef float f32x4w __attribute__((ext_vector_type(4)));

constexpr int kWD = 2, kWH = 4, kWW = 16;                       // voxel tile
constexpr int kWHD = kWD + 2, kWHH = kWH + 2, kWHW = kWW + 2;   // halo tile
constexpr int kWHalo = kWHD * kWHH * kWHW;                      // 432
constexpr int kWRanges = 128;                                   // tile ranges (grid.x)
constexpr int kWPts = 48;                                       // 3 depth taps x 16 Winograd points

struct WgradArgs {
    const float* x;    // [D][H][W][Cin]  conv input (activated)
    const float* gy;   // [D][H][W][64]   gradient w.r.t. the conv output
    float* partial;    // [gridDim.x][48][64][Cin]
    int D, H, W, Cin;
};

constexpr int kWThreads = 256;

template <int CSTAGE>     // input channels a workgroup stages: 32 (Cin = 64) or 16 (the first layer)
__global__ __launch_bounds__(kWThreads, 2) void conv3d_wgrad_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [kWHalo][kWSV]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nci = a.Cin >> 4;                       // ci blocks (1 or 4)
    constexpr int cstage = CSTAGE;
    constexpr int kWSV = CSTAGE + 8;                  // LDS voxel stride (floats)
    const int cib = wv & 1, cob = wv >> 1;            // this wave's ci block and co block inside the quadrant
    const int coh = blockIdx.y & 1, cih = blockIdx.y >> 1;
    const int ci0 = cih * 32;                         // first staged channel
    const bool wave_on = cih * 2 + cib < nci;
    const int i16 = lane & 15, k4 = lane >> 4;        // MFMA row/col (0..15) and k (0..3) = the tile of the step

    f32x4w acc[3][16];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[g][t] = f32x4w{0.f, 0.f, 0.f, 0.f};

    const int tiles_x = (a.W + kWW - 1) / kWW, tiles_y = (a.H + kWH - 1) / kWH, tiles_z = (a.D + kWD - 1) / kWD;
    const int ntiles = tiles_x * tiles_y * tiles_z;
    // Wave priority by progress (round 5, as in costvol_quad.hip): the launch is ONE round of two workgroups per CU, and a SIMD arbitrates
    // its two waves by age — the older workgroup ran ahead and the younger finished alone, its staging no longer hidden behind the
    // other's MFMAs.  Priority = 3 - the quarter of the workgroup's own tile list done: whoever is ahead yields.  0.600 -> 0.588 ms.
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    int it_ = 0, pq_ = -1;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it_) {
        {
            const int q = (4 * it_) / my_tiles;
            if (q != pq_) {
                pq_ = q;
                if (q <= 0) __builtin_amdgcn_s_setprio(3); else if (q == 1) __builtin_amdgcn_s_setprio(2);
                else if (q == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
            }
        }
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y; const int tz = t / tiles_y;
        const int x0 = tx * kWW, y0 = ty * kWH, z0 = tz * kWD;
        __syncthreads();  // previous tile's readers are done
        // ---- stage the halo tile of X, this quadrant's channels, zero outside the volume.  Batches of 4 UNCONDITIONAL loads
        // (clamped address, value masked afterwards): an `if (inside) load` is compiled into a branch with a wait per
        // element, i.e. 14 exposed memory latencies per tile — a third of the kernel's time once the MFMAs were 2.25x fewer ----
        constexpr int c4n = cstage >> 2, kItems = kWHalo * c4n, kBatch = 4;
        for (int b0 = 0; b0 < kItems; b0 += kBatch * kWThreads) {
            f32x4w v[kBatch];
            int keep[kBatch];
#pragma unroll
            for (int i = 0; i < kBatch; ++i) {
                int idx = min(b0 + i * kWThreads + tid, kItems - 1);
                asm volatile("" : "+v"(idx));      // decoded HERE, per tile: hoisted out of the tile loop the 14 decodes spill
                const int hv = idx / c4n, c4 = idx - hv * c4n;
                const int hz = hv / (kWHH * kWHW), rem = hv - hz * (kWHH * kWHW);
                const int hy = rem / kWHW, hx = rem - hy * kWHW;
                const int gz = z0 + hz - 1, gy_ = y0 + hy - 1, gx = x0 + hx - 1;
                keep[i] = (gz >= 0 && gz < a.D && gy_ >= 0 && gy_ < a.H && gx >= 0 && gx < a.W) ? -1 : 0;
                const int cz = min(max(gz, 0), a.D - 1), cy = min(max(gy_, 0), a.H - 1), cx = min(max(gx, 0), a.W - 1);
                v[i] = *reinterpret_cast<const f32x4w*>(a.x + (((size_t)cz * a.H + cy) * a.W + cx) * a.Cin + ci0 + c4 * 4);
            }
#pragma unroll
            for (int i = 0; i < kBatch; ++i) {
                int idx = b0 + i * kWThreads + tid;
                asm volatile("" : "+v"(idx));
                if (idx < kItems) {
                    const int hv = idx / c4n, c4 = idx - hv * c4n;
                    typedef int i32x4w __attribute__((ext_vector_type(4)));
                    const i32x4w bits = __builtin_bit_cast(i32x4w, v[i]) & keep[i];
                    *reinterpret_cast<f32x4w*>(lds + hv * kWSV + c4 * 4) = __builtin_bit_cast(f32x4w, bits);
                }
            }
        }
        __syncthreads();
        if (!wave_on) continue;
        // ---- 8 steps: (output slice vz, tile row tr, group of four tiles xg); this lane's tile of a step: column 4 xg + k4 ----
        // dY of this lane's (co, tile): 2 x 2 pixels, an unconditional load (pixels outside the volume read zeros)
        // The loads are UNCONDITIONAL (clamped address) and the value is masked afterwards with integer ANDs: a select between two
        // pointers (rounds 2-3) is compiled into a branch around every load with an s_waitcnt vmcnt(0) behind it once four
        // loads with different conditions sit in one step — the latency of every single load exposed.
        auto load_g = [&](int step, float (&g)[4], int (&gm)[4]) {
            const int vz = step >> 2, tr = (step >> 1) & 1, tc = (step & 1) * 4 + k4;
            const int gz = min(z0 + vz, a.D - 1);
            const float* base = a.gy + (size_t)(coh * 32 + cob * 16 + i16);
            const int zok = z0 + vz < a.D ? -1 : 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int py = y0 + 2 * tr + (q >> 1), px = x0 + 2 * tc + (q & 1);
                gm[q] = zok & (py < a.H ? -1 : 0) & (px < a.W ? -1 : 0);
                g[q] = base[(((size_t)gz * a.H + min(py, a.H - 1)) * a.W + min(px, a.W - 1)) * 64];
            }
        };
        auto step_mfma = [&](int step, const float (&gr)[4], const int (&gm)[4]) {
            const int vz = step >> 2, tr = (step >> 1) & 1, tc = (step & 1) * 4 + k4;
            float g[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] = __builtin_bit_cast(float, __builtin_bit_cast(int, gr[q]) & gm[q]);
            // Z = A g A^T, A = [1 0; 1 1; 1 -1; 0 -1]: rows over the tile's two pixel rows, then the same over its two columns
            float zr[4][2];
            zr[0][0] = g[0]; zr[0][1] = g[1];
            zr[1][0] = g[0] + g[2]; zr[1][1] = g[1] + g[3];
            zr[2][0] = g[0] - g[2]; zr[2][1] = g[1] - g[3];
            zr[3][0] = -g[2]; zr[3][1] = -g[3];
            float Z[16];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                Z[4 * r + 0] = zr[r][0];
                Z[4 * r + 1] = zr[r][0] + zr[r][1];
                Z[4 * r + 2] = zr[r][0] - zr[r][1];
                Z[4 * r + 3] = -zr[r][1];
            }
            // the 4 x 4 patch of X of this tile at halo depth vz + kd: rows 2 tr .. 2 tr + 3, columns 2 tc .. 2 tc + 3
            const float* pbase = lds + (((vz * kWHH + 2 * tr) * kWHW) + 2 * tc) * kWSV + cib * 16 + i16;
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
                const float* pk = pbase + kd * (kWHH * kWHW * kWSV);
                float u[4][4];        // B^T d: rows
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float d0 = pk[(0 * kWHW + c) * kWSV], d1 = pk[(1 * kWHW + c) * kWSV];
                    const float d2 = pk[(2 * kWHW + c) * kWSV], d3 = pk[(3 * kWHW + c) * kWSV];
                    u[0][c] = d0 - d2; u[1][c] = d1 + d2; u[2][c] = d2 - d1; u[3][c] = d1 - d3;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {      // (B^T d) B: columns, four points at a time straight into their MFMAs
                    const float v0 = u[r][0] - u[r][2], v1 = u[r][1] + u[r][2], v2 = u[r][2] - u[r][1], v3 = u[r][1] - u[r][3];
                    acc[kd][4 * r + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(Z[4 * r + 0], v0, acc[kd][4 * r + 0], 0, 0, 0);
                    acc[kd][4 * r + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(Z[4 * r + 1], v1, acc[kd][4 * r + 1], 0, 0, 0);
                    acc[kd][4 * r + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(Z[4 * r + 2], v2, acc[kd][4 * r + 2], 0, 0, 0);
                    acc[kd][4 * r + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(Z[4 * r + 3], v3, acc[kd][4 * r + 3], 0, 0, 0);
                }
            }
        };
        constexpr int NSTEP = 8;
        float g0[4], g1[4];
        int m0[4], m1[4];
        load_g(0, g0, m0);
#pragma unroll 1
        for (int step = 0; step < NSTEP; step += 2) {      // two steps per trip: the operand registers alternate, nothing is moved
            load_g(step + 1, g1, m1);
            __builtin_amdgcn_sched_barrier(0);             // the requests leave BEFORE the MFMAs of this step
            step_mfma(step, g0, m0);
            __builtin_amdgcn_sched_barrier(0);
            load_g(min(step + 2, NSTEP - 1), g0, m0);
            __builtin_amdgcn_sched_barrier(0);
            step_mfma(step + 1, g1, m1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (wave_on) {
        // C/D layout of 16x16x4: col = lane & 15 (j = ci), row = (lane >> 4) * 4 + reg (i = co)
        float* out = a.partial + (size_t)blockIdx.x * kWPts * 64 * a.Cin;
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int xi = 0; xi < 16; ++xi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = coh * 32 + cob * 16 + k4 * 4 + r, ci = ci0 + cib * 16 + i16;
                    out[((size_t)(kd * 16 + xi) * 64 + co) * a.Cin + ci] = acc[kd][xi][r];
                }
    }
}

// dW[co][ci][kd][ky][kx] (torch layout [64][Cin][27]) = G^T (sum over ranges of dU_kd) G.  Workgroup = 32 items (kd, co, 4
// consecutive ci: 16-byte loads) x 8 interleaved slices of the range list, combined through LDS in index order.
__global__ __launch_bounds__(256) void conv3d_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                  int nranges, int Cin) {
    __shared__ float4 part[8][32][16];                // 64 KB
    const int c4n = Cin >> 2, nitems = 3 * 64 * c4n;
    const int lane = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int item = blockIdx.x * 32 + lane;
    const bool live = item < nitems;
    const int it = live ? item : 0;
    const int c4 = it % c4n, co = (it / c4n) % 64, kd = it / (c4n * 64);
    float4 s[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) s[xi] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        for (int g = sl; g < nranges; g += 8) {
            const float* p = partial + ((size_t)g * kWPts + kd * 16) * 64 * Cin + (size_t)co * Cin + c4 * 4;
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) {
                const float4 q = *reinterpret_cast<const float4*>(p + (size_t)xi * 64 * Cin);
                s[xi].x += q.x; s[xi].y += q.y; s[xi].z += q.z; s[xi].w += q.w;
            }
        }
    }
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) part[sl][lane][xi] = s[xi];
    __syncthreads();
    if (sl == 0 && live) {
#pragma unroll
        for (int q = 1; q < 8; ++q)
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) {
                const float4 v = part[q][lane][xi];
                s[xi].x += v.x; s[xi].y += v.y; s[xi].z += v.z; s[xi].w += v.w;
            }
        // G^T U G with G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1] (4 x 3), per channel of the float4
        auto gt = [](float u0, float u1, float u2, float u3, float (&o)[3]) {
            o[0] = u0 + 0.5f * (u1 + u2); o[1] = 0.5f * (u1 - u2); o[2] = 0.5f * (u1 + u2) + u3;
        };
        float* o = dw + ((size_t)co * Cin + c4 * 4) * 27 + kd * 9;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            auto comp = [&](const float4& v) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); };
            float rowt[3][4];          // (G^T U)[ky][b]
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                float col3[3];
                gt(comp(s[0 * 4 + b]), comp(s[1 * 4 + b]), comp(s[2 * 4 + b]), comp(s[3 * 4 + b]), col3);
                rowt[0][b] = col3[0]; rowt[1][b] = col3[1]; rowt[2][b] = col3[2];
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                float w3[3];
                gt(rowt[ky][0], rowt[ky][1], rowt[ky][2], rowt[ky][3], w3);
                o[(size_t)e * 27 + ky * 3 + 0] = w3[0]; o[(size_t)e * 27 + ky * 3 + 1] = w3[1]; o[(size_t)e * 27 + ky * 3 + 2] = w3[2];
            }
        }
    }
}

}  // namespace nrgbd

// scratch of nrgbd_conv3d_wgrad_f32 in units of 27 * 64 * Cin floats: kWRanges x 48 points x 64 x Cin
extern "C" int nrgbd_conv3d_wgrad_workgroups(void) {
    return (nrgbd::kWRanges * nrgbd::kWPts * 64 + 27 * 64 - 1) / (27 * 64);
}

extern "C" int nrgbd_conv3d_wgrad_f32(const float* x, const float* gy, float* partial, float* dw, int D, int H,
                                      int W, int Cin, void* stream) {
    using namespace nrgbd;
    if (!x || !gy || !partial || !dw) return NRGBD_E_NULL;
    if (D <= 0 || H <= 0 || W <= 0 || (Cin != 16 && Cin != 64)) return NRGBD_E_SHAPE;
    WgradArgs a{x, gy, partial, D, H, W, Cin};
    const size_t lds = (size_t)kWHalo * ((Cin < 32 ? Cin : 32) + 8) * sizeof(float);  // 69,120 B (Cin = 64): two workgroups per CU
    hipError_t e;
    if (Cin >= 64) {
        e = set_max_dynamic_lds(reinterpret_cast<const void*>(&conv3d_wgrad_kernel<32>), (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(conv3d_wgrad_kernel<32>, dim3(kWRanges, 4), dim3(kWThreads), lds, (hipStream_t)stream, a);
    } else {
        e = set_max_dynamic_lds(reinterpret_cast<const void*>(&conv3d_wgrad_kernel<16>), (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(conv3d_wgrad_kernel<16>, dim3(kWRanges, 2), dim3(kWThreads), lds, (hipStream_t)stream, a);
    }
    hipLaunchKernelGGL(conv3d_wgrad_reduce_kernel, dim3(ceil_div(3 * 64 * (Cin / 4), 32)), dim3(256), 0, (hipStream_t)stream,
                       partial, dw, kWRanges, Cin);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

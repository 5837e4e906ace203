// conv3d.hip — K-Net 3x3x3 convolution (stride 1, pad 1, no bias) on the fp32 matrix cores of gfx950,
// with the BatchNorm3d work of the reference fused around it.
//
// Replaces, per K-Net layer (models/basic.py:71-94,127-132; psm_submodule.py:19-23):
//   nn.Conv3d(Cin, 64, 3, padding=1, bias=False)  ->  implicit GEMM on v_mfma_f32_32x32x2_f32
//   nn.BatchNorm3d (batch statistics: the reference never leaves train() mode)
//        statistics  -> per-workgroup (sum, sum of squares) partials written by the conv epilogue
//        normalise + affine + ReLU + residual add -> applied by the NEXT layer while it loads its
//        input tile (the activated tensor is only written when a later layer needs it as a residual)
// so one layer = one pass: read Z_in (+ residual), write Z_out.  The reference runs conv, BN-stat,
// BN-apply, ReLU and add as separate kernels over a 805 MB activation (192x256x64 grid).
//
// Exactness: f32 MFMA is a k-ordered fmaf chain (bit-identical to VALU fp32), so the result differs
// from MIOpen / the CPU oracle only by summation order.
//
// Data layout: activations are channels-last [D][H][W][C] (a voxel's channels are contiguous), weights
// are pre-packed per (tap, channel block, k-group, cout fragment) so that a wave's B-operand load is one
// contiguous 1 KB line (conv3d_pack_weights below).
//
// Tiling: workgroup = 256 threads = 4 waves = 2 x 8 x 16 output voxels x 64 output channels; a wave
// owns 4 rows x 16 x = 64 voxels = two 32-row MFMA tiles x two 32-column tiles (64 accumulator VGPRs).
// The K loop runs over channel blocks of 16: the (4 x 10 x 18)-voxel halo tile of the block is staged
// in LDS as [voxel][16] with the 16-B slots XOR-swizzled by the voxel index (conflict-free
// ds_read_b128 at a 64-B stride, no padding), then 27 taps x 2 k-groups, each = 2 A reads (LDS, b128) + 2 B loads (L2, 16 B per
// lane) + 16 MFMAs.  One b128 feeds FOUR k-steps: step e of a group contracts channel 4g+e (lanes
// 0-31) and channel 8+4g+e (lanes 32-63) — the contraction order is free, A and B just agree on it.
// MFMA row i <-> voxel: the 16 lanes of each ds_read_b128 lane group take 16 consecutive x of one row.
#include <cstdlib>

#include "conv_tile.hpp"

namespace nrgbd {

constexpr int kTD = 2, kTH = 8, kTW = 16;                   // output tile (voxels)
constexpr int kHD = kTD + 2, kHH = kTH + 2, kHW = kTW + 2;  // halo tile
constexpr int kHaloVox = kHD * kHH * kHW;                   // 720
constexpr int kCout = 64;

struct Conv3dArgs {
    const float* x;       // [D][H][W][Cin] raw input (pre-activation)
    const float* x_ss;    // [Cin][2] (scale, shift) applied to x, or null = identity
    const float* res;     // [D][H][W][Cin] second operand added after activation, or null
    const float* res_ss;  // [Cin][2] for res, or null = identity
    float* mat;           // [D][H][W][Cin]: materialised input act(x) + act(res), or null
    const float* wp;      // packed weights (conv3d_pack_weights)
    float* y;             // [D][H][W][64] raw convolution output
    float* stats;         // [num_workgroups][128]: per-channel sum (0..63) and sum of squares (64..127), or null
    int x_relu, res_relu;
    int D, H, W;
    int xcd;              // re-map workgroups so each XCD owns a contiguous run of tiles (conv_tile.hpp)
};

// tile id -> tile coordinates.  order 0/1: x fastest, then y, then z; order 2: z fastest, then x, then y — the z halo is
// the largest shared part of neighbouring tiles (4 input planes for 2 output planes), so consecutive ids then re-use it
// from the XCD's L2 (with the contiguous-run XCD mapping of conv_tile.hpp).
__device__ __forceinline__ void tile_coords(int t, int tiles_x, int tiles_y, int tiles_z, int order, int& tx, int& ty, int& tz) {
    if (order == 2) { tz = t % tiles_z; t /= tiles_z; tx = t % tiles_x; ty = t / tiles_x; }
    else            { tx = t % tiles_x; t /= tiles_x; ty = t % tiles_y; tz = t / tiles_y; }
}

// Stage the (4 x 10 x 18)-voxel halo tile of channel block `cblk` into LDS as [voxel][kSV]:
//   in = act(x*s+t) [+ act(res*s'+t')] inside the volume, 0 outside (zero padding applies to the
//   ACTIVATED tensor, exactly like F.conv3d(padding=1) on the materialised activation).
template <int CIN>
__device__ __forceinline__ void stage_halo(const Conv3dArgs& a, int cblk, float* lds, int tid, int x0, int y0, int z0) {
    for (int idx = tid; idx < kHaloVox * (kCB / 4); idx += 256) {
        const int hv = idx >> 2, c4 = idx & 3;
        const int hz = hv / (kHH * kHW), rem = hv - hz * (kHH * kHW);
        const int hy = rem / kHW, hx = rem - hy * kHW;
        const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gz >= 0 && gz < a.D && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
            const size_t vox = ((size_t)gz * a.H + gy) * a.W + gx;
            const int c = cblk * kCB + c4 * 4;
            v = *reinterpret_cast<const f32x4*>(a.x + vox * CIN + c);
            if (a.x_ss) {
                const float* ss = a.x_ss + 2 * c;
                v.x = __builtin_fmaf(v.x, ss[0], ss[1]); v.y = __builtin_fmaf(v.y, ss[2], ss[3]);
                v.z = __builtin_fmaf(v.z, ss[4], ss[5]); v.w = __builtin_fmaf(v.w, ss[6], ss[7]);
            }
            if (a.x_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (a.res) {
                f32x4 r = *reinterpret_cast<const f32x4*>(a.res + vox * CIN + c);
                if (a.res_ss) {
                    const float* ss = a.res_ss + 2 * c;
                    r.x = __builtin_fmaf(r.x, ss[0], ss[1]); r.y = __builtin_fmaf(r.y, ss[2], ss[3]);
                    r.z = __builtin_fmaf(r.z, ss[4], ss[5]); r.w = __builtin_fmaf(r.w, ss[6], ss[7]);
                }
                if (a.res_relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
                v = v + r;
            }
            // the activated input is written once, by the tile that owns the voxel
            if (a.mat && hz >= 1 && hz <= kTD && hy >= 1 && hy <= kTH && hx >= 1 && hx <= kTW)
                *reinterpret_cast<f32x4*>(a.mat + vox * CIN + c) = v;
        }
        *reinterpret_cast<f32x4*>(lds + lds_slot(hv, c4)) = v;
    }
}

// PF = true (layers without a residual operand, Cin = 64): the raw input words of channel block c+1 are
// fetched into registers WHILE the 864 MFMAs of block c run (one 16-B load per step, so each load's latency
// hides inside the one-step-ahead operand pipeline), and are normalised / activated / written to LDS after
// the loop — the matrix pipe no longer idles during staging (78.9 % busy without it, rocprofv3 PMC).
template <int CIN, bool PF, bool RES = false>
__global__ __launch_bounds__(256, PF ? 2 : 3) void conv3d_mfma_kernel(const Conv3dArgs a) {
    constexpr int NCBLK = CIN / kCB;
    constexpr int NPF = (kHaloVox * (kCB / 4) + 255) / 256;  // 16-B words per thread per block (12)
    constexpr int G4 = kCB / 8;  // k-groups per block (4 k-steps = 8 channels each)
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [kHaloVox][kSV] (+ 4*128 floats stats)

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tiles_x = (a.W + kTW - 1) / kTW, tiles_y = (a.H + kTH - 1) / kTH;
    int t = xcd_tile(blockIdx.x, gridDim.x, a.xcd);
    const int tile_id = t;
    int tx, ty, tz;
    tile_coords(t, tiles_x, tiles_y, (a.D + kTD - 1) / kTD, a.xcd, tx, ty, tz);
    const int x0 = tx * kTW, y0 = ty * kTH, z0 = tz * kTD;

    // this lane's A rows: wave -> (dz, 4-row band); tile m -> 2 rows of the band
    const int i = lane & 31, khalf = lane >> 5;
    int dy, px;
    row_to_yx(i, dy, px);
    const int wz = wv >> 1, wy = (wv & 1) * 4;
    // halo-relative voxel index of (wz, wy + 2m + dy, px) for tap (0,0,0); + tap offset per tap
    const int hv0 = ((wz * kHH) + (wy + dy)) * kHW + px;   // m = 0
    const int hv1 = hv0 + 2 * kHW;                         // m = 1: two rows further

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;

    // prefetch bookkeeping: word u of this thread = halo voxel (tid>>2) + 64u, 16-B word tid&3 of the block
    const int c4 = tid & 3;
    unsigned pf_off[PF ? NPF : 1];   // element offset of the voxel's channel c4*4 (a harmless in-tensor offset when outside)
    unsigned pf_ok = 0;              // bit u: the voxel is inside the volume (outside = zero padding)
    unsigned pf_own = 0;             // bit u: the voxel belongs to this tile's interior (materialise target)
    f32x4 pre[PF ? NPF : 1], prer[(PF && RES) ? NPF : 1];
    if constexpr (PF) {
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int hv = (tid >> 2) + 64 * u;
            const int hz = hv / (kHH * kHW), rem = hv - hz * (kHH * kHW);
            const int hy = rem / kHW, hx = rem - hy * kHW;
            const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
            const bool ok = hv < kHaloVox && gz >= 0 && gz < a.D && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            pf_off[u] = ok ? (unsigned)((((size_t)gz * a.H + gy) * a.W + gx) * CIN + c4 * 4) : (unsigned)(c4 * 4);
            if (ok) pf_ok |= 1u << u;
            if (ok && hz >= 1 && hz <= kTD && hy >= 1 && hy <= kTH && hx >= 1 && hx <= kTW) pf_own |= 1u << u;
        }
        // unconditional loads: lanes outside the volume read a valid dummy word that is zeroed when published
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            pre[u] = *reinterpret_cast<const f32x4*>(a.x + pf_off[u]);
            if constexpr (RES) prer[u] = *reinterpret_cast<const f32x4*>(a.res + pf_off[u]);
        }
    }

    // B operand ring: runs BD steps ahead of its use — vmcnt retires in order, so a B load also waits for the
    // (HBM-latency) prefetch words issued before it — and straight across channel-block boundaries, so the first steps
    // of a block never wait for their weights.  A (LDS) runs one step ahead.
    constexpr int NSTEP = 27 * G4;
    constexpr int WSTEP = NCBLK * G4 * 2 * 64;  // f32x4 per tap
    constexpr int BD = PF ? 3 : 1, NB = BD + 1;
    constexpr bool BCONT = !RES;   // the residual variant has no register left to carry the ring across the publish phase
    f32x4 Bn[NB][2];
    if constexpr (BCONT) {
        const f32x4* w0 = reinterpret_cast<const f32x4*>(a.wp) + lane;
#pragma unroll
        for (int b = 0; b < BD; ++b) {
            Bn[b][0] = w0[(size_t)(b / G4) * WSTEP + (b % G4) * (2 * 64)];
            Bn[b][1] = w0[(size_t)(b / G4) * WSTEP + (b % G4) * (2 * 64) + 64];
        }
    }

    for (int cblk = 0; cblk < NCBLK; ++cblk) {
        if constexpr (PF) {
            // normalise / activate the prefetched words of this block and publish them to LDS
            const int c = cblk * kCB + c4 * 4;
            float ss[8] = {1.f, 0.f, 1.f, 0.f, 1.f, 0.f, 1.f, 0.f}, rs[8] = {1.f, 0.f, 1.f, 0.f, 1.f, 0.f, 1.f, 0.f};
            if (a.x_ss) {
#pragma unroll
                for (int e = 0; e < 8; ++e) ss[e] = a.x_ss[2 * c + e];
            }
            if (RES && a.res_ss) {
#pragma unroll
                for (int e = 0; e < 8; ++e) rs[e] = a.res_ss[2 * c + e];
            }
#pragma unroll
            for (int u = 0; u < NPF; ++u) {
                const int hv = (tid >> 2) + 64 * u;
                if (hv >= kHaloVox) continue;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if ((pf_ok >> u) & 1u) {
                    v = pre[u];
                    v.x = __builtin_fmaf(v.x, ss[0], ss[1]); v.y = __builtin_fmaf(v.y, ss[2], ss[3]);
                    v.z = __builtin_fmaf(v.z, ss[4], ss[5]); v.w = __builtin_fmaf(v.w, ss[6], ss[7]);
                    if (a.x_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    if constexpr (RES) {
                        f32x4 r = prer[u];
                        if (a.res_ss) {
                            r.x = __builtin_fmaf(r.x, rs[0], rs[1]); r.y = __builtin_fmaf(r.y, rs[2], rs[3]);
                            r.z = __builtin_fmaf(r.z, rs[4], rs[5]); r.w = __builtin_fmaf(r.w, rs[6], rs[7]);
                        }
                        if (a.res_relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
                        v = v + r;
                    }
                    if (a.mat && ((pf_own >> u) & 1u))
                        *reinterpret_cast<f32x4*>(a.mat + pf_off[u] + cblk * kCB) = v;
                }
                *reinterpret_cast<f32x4*>(lds + lds_slot(hv, c4)) = v;
            }
        } else {
            stage_halo<CIN>(a, cblk, lds, tid, x0, y0, z0);
        }
        __syncthreads();

        // ---- 27 taps x G4 k-groups; B operand streamed from L2 (packed: one 1 KB line per wave load) ----
        const f32x4* wb = reinterpret_cast<const f32x4*>(a.wp) + (size_t)cblk * (G4 * 2 * 64) + lane;
        f32x4 An[2][2];
        // next block's words; in the last block every lane re-reads element 0 instead (one cached line, no branch)
        const unsigned nb = (cblk + 1) * kCB, live = cblk + 1 < NCBLK ? ~0u : 0u;
        const f32x4* wbn = wb + (cblk + 1 < NCBLK ? G4 * 2 * 64 : 0);  // next block's stream (last block: a harmless re-read)
        if constexpr (!BCONT) {
#pragma unroll
            for (int b = 0; b < BD; ++b) {
                Bn[b][0] = wb[(size_t)(b / G4) * WSTEP + (b % G4) * (2 * 64)];
                Bn[b][1] = wb[(size_t)(b / G4) * WSTEP + (b % G4) * (2 * 64) + 64];
            }
        }
        An[0][0] = *reinterpret_cast<const f32x4*>(lds + lds_slot(hv0, khalf * 2));
        An[0][1] = *reinterpret_cast<const f32x4*>(lds + lds_slot(hv1, khalf * 2));
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s + 1 < NSTEP) {  // A operands of step s+1
                const int tap = (s + 1) / G4, g = (s + 1) % G4;
                const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
                const int voff = (kd * kHH + kh) * kHW + kw;  // tap offset in halo voxels
                // opaque copy: the swizzled address is recomputed here (6 VALU beside 16 MFMAs) instead of all
                // 54 x 2 addresses being hoisted out of the unrolled loop into registers (that spills)
                int h0 = hv0;
                asm volatile("" : "+v"(h0));
                An[nxt][0] = *reinterpret_cast<const f32x4*>(lds + lds_slot(h0 + voff, khalf * 2 + g));
                An[nxt][1] = *reinterpret_cast<const f32x4*>(lds + lds_slot(h0 + 2 * kHW + voff, khalf * 2 + g));
            }
            if (BCONT || s + BD < NSTEP) {   // B operands of step s+BD (of the next channel block once this one's stream is exhausted)
                const int t = (s + BD) % NSTEP, tap = t / G4, g = t % G4;
                const f32x4* wn = (s + BD < NSTEP ? wb : wbn) + (size_t)tap * WSTEP + g * (2 * 64);
                Bn[(s + BD) % NB][0] = wn[0]; Bn[(s + BD) % NB][1] = wn[64];
            }
            if constexpr (PF) {  // one word of the NEXT channel block per step: x in steps 2 .. 2+NPF-1, then the residual
                constexpr int PF0 = 2;
                if constexpr (!RES) {   // registers to spare: a scalar branch skips the loads in the last block
                    if (s >= PF0 && s < PF0 + NPF && live) pre[s - PF0] = *reinterpret_cast<const f32x4*>(a.x + pf_off[s - PF0] + nb);
                } else {
                    if (s >= PF0 && s < PF0 + NPF) pre[s - PF0] = *reinterpret_cast<const f32x4*>(a.x + ((pf_off[s - PF0] + nb) & live));
                }
                if constexpr (RES) {
                    if (s >= PF0 + NPF && s < PF0 + 2 * NPF)
                        prer[s - PF0 - NPF] = *reinterpret_cast<const f32x4*>(a.res + ((pf_off[s - PF0 - NPF] + nb) & live));
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(An[cur][m][e], Bn[s % NB][n][e], acc[m][n], 0, 0, 0);
        }
        if constexpr (BCONT && NSTEP % NB != 0) {  // next block's steps 0..BD-1 sit in ring slots (NSTEP+b) % NB: move them to b
            f32x4 t[BD][2];
#pragma unroll
            for (int b = 0; b < BD; ++b) { t[b][0] = Bn[(NSTEP + b) % NB][0]; t[b][1] = Bn[(NSTEP + b) % NB][1]; }
#pragma unroll
            for (int b = 0; b < BD; ++b) { Bn[b][0] = t[b][0]; Bn[b][1] = t[b][1]; }
        }
        __syncthreads();
    }

    // ---- epilogue: raw output (channels-last) + per-channel partial statistics ----
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    const int gz = z0 + wz;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * khalf;  // MFMA C/D row held in register r
            int ry, rx;
            row_to_yx(row, ry, rx);
            const int gy = y0 + wy + 2 * m + ry, gx = x0 + rx;
            if (gz < a.D && gy < a.H && gx < a.W) {
                const size_t vox = ((size_t)gz * a.H + gy) * a.W + gx;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const float z = acc[m][n][r];
                    a.y[vox * kCout + n * 32 + i] = z;
                    s1[n] += z;
                    s2[n] = __builtin_fmaf(z, z, s2[n]);
                }
            }
        }
    }
    if (a.stats) {
        float* red = lds;  // reuse: [4 waves][128]
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            s1[n] += __shfl_xor(s1[n], 32, 64);
            s2[n] += __shfl_xor(s2[n], 32, 64);
        }
        if (khalf == 0) {
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                red[wv * 128 + n * 32 + i] = s1[n];
                red[wv * 128 + 64 + n * 32 + i] = s2[n];
            }
        }
        __syncthreads();
        if (tid < 128)
            a.stats[(size_t)tile_id * 128 + tid] = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
    }
}

// Last K-Net layer (models/basic.py:92-94: Conv3d(64, 1, 3, padding=1, bias=False), no BatchNorm).
// One output channel leaves the MFMA columns empty if the taps are the K dimension, so the sum is re-associated:
//     P[u][tap] = sum_c w[tap][c] * in[u][c]          a [halo voxels x 64] x [64 x 27(->32)] GEMM on the matrix cores
//     out[v]    = sum_tap P[v + tap][tap]             27 scalar LDS reads per output voxel
// i.e. every input voxel is projected onto the 27 tap weights once.  The layer is a pure stream over an 805 MB tensor, so
// the organisation is chosen for bytes, not for the (small) matrix work: a workgroup owns an 8x16-pixel column of `dz`
// output slices and MARCHES along the depth axis — every input slice of its column is loaded, activated and projected
// ONCE (the previous form projected a 2-slice tile's 4-slice halo: 2.8x the input per launch, 0.62 ms at the 192x256x64
// grid); what is left is the 10x18 / 8x16 halo in the plane (neighbours share it in their XCD's L2) and two slices per
// depth chunk.  Per slice: the 180 halo pixels x 64 channels of slice z+1 are in flight (12 16-B words per thread) while
// slice z is projected (v_mfma_f32_16x16x4_f32: wave w owns pixel rows 48w .. 48w+47 x 32 tap columns), P goes to LDS as
// [pixel][27] (odd stride: conflict-free along x) and 128 threads add the slice's three depth-tap sums to the running
// outputs z+1 (kd = 0), z (kd = 1), z-1 (kd = 2, which completes and stores it).
// w1 is [27][64] (tap-major): a lane's B operand is 16 contiguous bytes of it.
constexpr int kC1TH = 8, kC1TW = 16, kC1HW = kC1TW + 2, kC1Pix = (kC1TH + 2) * kC1HW;   // 180 halo pixels per slice
constexpr int kC1Img = kC1Pix * kSV + 16;      // floats of one channel block's LDS image (+16: the four images start in different banks)
constexpr int kC1PS = 27;                      // P row stride (floats)
constexpr int kC1NPF = 12;                     // 16-B words per thread and slice: pixel (tid >> 4) + 16 u, channels 4 (tid & 15) ..
constexpr size_t kC1Lds = (size_t)(4 * kC1Img + kC1Pix * kC1PS) * sizeof(float);

__global__ __launch_bounds__(256, 2) void conv3d_cout1_kernel(const Conv3dArgs a, const float* __restrict__ w1, int dz) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* stage = lds;                         // [4 channel blocks][180 pixels][16] (swizzled 16-B slots)
    float* P = lds + 4 * kC1Img;                // [180 pixels][27 taps]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tiles_x = (a.W + kC1TW - 1) / kC1TW, tiles_y = (a.H + kC1TH - 1) / kC1TH;
    int t = xcd_tile(blockIdx.x, gridDim.x, 1);  // an XCD works on a contiguous run of columns: in-plane halos meet in its L2
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y, cz = t / tiles_y;
    const int x0 = tx * kC1TW, y0 = ty * kC1TH, zc0 = cz * dz, zc1 = min(zc0 + dz, a.D);

    // ---- loader role: word u of this thread = halo pixel (tid >> 4) + 16 u, channels 4 c16 .. 4 c16 + 3
    const int c16 = tid & 15;
    unsigned pf_off[kC1NPF], pf_ok = 0;
#pragma unroll
    for (int u = 0; u < kC1NPF; ++u) {
        const int p = (tid >> 4) + 16 * u;
        const int hy = p / kC1HW, hx = p - hy * kC1HW;
        const int gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool ok = p < kC1Pix && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        pf_off[u] = (ok ? (unsigned)(gy * a.W + gx) * 64u : 0u) + (unsigned)(c16 * 4);   // an outside word reads a harmless in-tensor one
        if (ok) pf_ok |= 1u << u;
    }
    float ss[8] = {1.f, 0.f, 1.f, 0.f, 1.f, 0.f, 1.f, 0.f};
    if (a.x_ss) {
#pragma unroll
        for (int e = 0; e < 8; ++e) ss[e] = a.x_ss[8 * c16 + e];
    }
    const size_t plane = (size_t)a.H * a.W * 64;
    f32x4 pre[kC1NPF];
    auto fetch = [&](int z) __attribute__((always_inline)) {
        const float* xs = a.x + (size_t)z * plane;
#pragma unroll
        for (int u = 0; u < kC1NPF; ++u) pre[u] = *reinterpret_cast<const f32x4*>(xs + pf_off[u]);
    };

    // ---- matrix role: pixel rows 16 (3 wv + rt) + (lane & 15), channel word kq of a block, tap columns 16 ct + (lane & 15)
    const int kq = lane >> 4, n = lane & 15;
    f32x4 B[4][2];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int tap = 16 * ct + n;
            B[cb][ct] = tap < 27 ? *reinterpret_cast<const f32x4*>(w1 + tap * 64 + cb * kCB + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    int arow[3];
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) {
        const int p = min(16 * (3 * wv + rt) + n, kC1Pix - 1);   // rows 180..191 of the last tile are computed and dropped
        arow[rt] = p;
    }
    // ---- gather role (threads 0..127): output pixel (oy, ox) of the column
    const int ox = tid & 15, oy = (tid >> 4) & 7;
    const float* pg = P + (oy * kC1HW + ox) * kC1PS;
    const bool store_ok = tid < 128 && y0 + oy < a.H && x0 + ox < a.W;
    float* yo = a.y + (size_t)(y0 + oy) * a.W + (x0 + ox);
    float run_a = 0.f, run_b = 0.f;             // partial sums of outputs zz - 1 (kd = 0, 1 done) and zz (kd = 0 done)

    if (zc0 - 1 >= 0) fetch(zc0 - 1);
    for (int zz = zc0 - 1; zz <= zc1; ++zz) {
        const bool valid = zz >= 0 && zz < a.D;   // uniform; an outside slice is zero padding: it contributes nothing
        if (valid) {
#pragma unroll
            for (int u = 0; u < kC1NPF; ++u) {
                const int p = (tid >> 4) + 16 * u;
                if (p >= kC1Pix) continue;
                f32x4 v = pre[u];
                v.x = __builtin_fmaf(v.x, ss[0], ss[1]); v.y = __builtin_fmaf(v.y, ss[2], ss[3]);
                v.z = __builtin_fmaf(v.z, ss[4], ss[5]); v.w = __builtin_fmaf(v.w, ss[6], ss[7]);
                if (a.x_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (!((pf_ok >> u) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};   // zero padding applies to the ACTIVATED tensor
                *reinterpret_cast<f32x4*>(stage + (c16 >> 2) * kC1Img + lds_slot(p, c16 & 3)) = v;
            }
        }
        __syncthreads();
        if (zz + 1 <= zc1 && zz + 1 < a.D) fetch(zz + 1);   // lands while this slice is projected and summed
        if (valid) {
            f32x4 acc[3][2];
#pragma unroll
            for (int rt = 0; rt < 3; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
                for (int rt = 0; rt < 3; ++rt) {
                    const f32x4 A = *reinterpret_cast<const f32x4*>(stage + cb * kC1Img + lds_slot(arow[rt], kq));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[rt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[e], B[cb][0][e], acc[rt][0], 0, 0, 0);
                        acc[rt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[e], B[cb][1][e], acc[rt][1], 0, 0, 0);
                    }
                }
            }
            // accumulator register r of a 16x16 block = row 4 kq + r, column n
#pragma unroll
            for (int rt = 0; rt < 3; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int p = 16 * (3 * wv + rt) + 4 * kq + r;
                    if (p < kC1Pix) {
                        P[p * kC1PS + n] = acc[rt][0][r];
                        if (n < 27 - 16) P[p * kC1PS + 16 + n] = acc[rt][1][r];
                    }
                }
        }
        __syncthreads();
        if (tid < 128) {
            float g0 = 0.f, g1 = 0.f, g2 = 0.f;
            if (valid) {
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const float* q = pg + (kh * kC1HW + kw) * kC1PS + kh * 3 + kw;
                        g0 += q[0]; g1 += q[9]; g2 += q[18];
                    }
            }
            const float done = run_a + g2;        // output zz - 1: its kd = 2 tap reads slice zz
            if (store_ok && zz - 1 >= zc0 && zz - 1 < zc1) yo[(size_t)(zz - 1) * a.H * a.W] = done;
            run_a = run_b + g1;
            run_b = g0;
        }
    }
}

// weights [64][Cin][3][3][3] (torch layout) -> packed [tap][cblk][g][nfrag][lane = khalf*32 + j][4]
//   value = w[cout = nfrag*32 + j][cin = cblk*16 + khalf*8 + g*4 + e][tap]
__global__ void conv3d_pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin) {
    const int ncblk = Cin / kCB, G4 = kCB / 8;
    const int total = 27 * ncblk * G4 * 2 * 64 * 4;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    int t = idx;
    const int e = t & 3; t >>= 2;
    const int ln = t & 63; t >>= 6;
    const int nfrag = t & 1; t >>= 1;
    const int g = t % G4; t /= G4;
    const int cblk = t % ncblk; const int tap = t / ncblk;
    const int cout = nfrag * 32 + (ln & 31);
    const int cin = cblk * kCB + (ln >> 5) * (kCB / 2) + g * 4 + e;
    wp[idx] = w[((size_t)cout * Cin + cin) * 27 + tap];
}

// Reduce the per-workgroup partials to BatchNorm (scale, shift) and update the running statistics
// (nn.BatchNorm3d in train mode: biased variance normalises, unbiased variance feeds running_var).
// grid = 64 workgroups (one per channel) x 256 threads: each thread walks a strided slice of the
// workgroup list for its channel's (sum, sum of squares), tree-reduced in double.
__global__ __launch_bounds__(256) void bn3d_finalize_kernel(const float* __restrict__ stats, int nwg, double count,
                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                     float eps, float momentum, float* __restrict__ running_mean,
                                     float* __restrict__ running_var, float* __restrict__ ss, unsigned int* __restrict__ collapse_count, long long* __restrict__ batches_tracked) {
    __shared__ double sh[2][256];
    const int c = blockIdx.x, tid = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int g = tid; g < nwg; g += 256) {
        s1 += (double)stats[(size_t)g * 128 + c];
        s2 += (double)stats[(size_t)g * 128 + 64 + c];
    }
    sh[0][tid] = s1; sh[1][tid] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { sh[0][tid] += sh[0][tid + o]; sh[1][tid] += sh[1][tid + o]; }
        __syncthreads();
    }
    if (tid == 0) bn_finalize_channel(sh[0][0], sh[1][0], count, gamma[c], beta[c], eps, momentum, running_mean, running_var, ss, c, collapse_count);
    if (tid == 0 && c == 0 && batches_tracked) *batches_tracked += 1;      // nn.BatchNorm's num_batches_tracked side effect (one launch less per layer)
}

}  // namespace nrgbd

extern "C" int nrgbd_conv3d_workgroups(int D, int H, int W) {
    using namespace nrgbd;
    if (D <= 0 || H <= 0 || W <= 0) return NRGBD_E_SHAPE;
    return ceil_div(W, kTW) * ceil_div(H, kTH) * ceil_div(D, kTD);
}

extern "C" int nrgbd_conv3d_pack_weights(const float* w, float* wp, int Cin, void* stream) {
    using namespace nrgbd;
    if (!w || !wp) return NRGBD_E_NULL;
    if (Cin <= 0 || Cin % kCB) return NRGBD_E_SHAPE;
    const int total = 27 * Cin * kCout;
    hipLaunchKernelGGL(conv3d_pack_weights_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wp, Cin);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_conv3d_3x3x3_f32(const float* x, const float* x_ss, int x_relu, const float* res,
                                      const float* res_ss, int res_relu, float* materialized,
                                      const float* w_packed, float* y, float* stats, int D, int H, int W,
                                      int Cin, int Cout, void* stream) {
    using namespace nrgbd;
    if (!x || !w_packed || !y) return NRGBD_E_NULL;
    if (D <= 0 || H <= 0 || W <= 0) return NRGBD_E_SHAPE;
    if (Cout != kCout || (Cin != 16 && Cin != 64)) return NRGBD_E_SHAPE;
    Conv3dArgs a{x, x_ss, res, res_ss, materialized, w_packed, y, stats, x_relu, res_relu, D, H, W, dev_env_int("NRGBD_XCD")};
    const int nwg = ceil_div(W, kTW) * ceil_div(H, kTH) * ceil_div(D, kTD);
    const size_t lds = (size_t)kHaloVox * kSV * sizeof(float);  // 46,080 B (>= the 2 KB the statistics reuse)
    const bool prefetch = (Cin == 64) && ((long)D * H * W * Cin < (1L << 32)) && !dev_env_int("NRGBD_CONV3D_NOPF");
    const bool small = (long)D * H * W * Cin < (1L << 32);
    if (Cin == 16 && small && !res)   // single channel block: batched unconditional loads only
        hipLaunchKernelGGL((conv3d_mfma_kernel<16, true>), dim3(nwg), dim3(256), lds, (hipStream_t)stream, a);
    else if (Cin == 16)
        hipLaunchKernelGGL((conv3d_mfma_kernel<16, false>), dim3(nwg), dim3(256), lds, (hipStream_t)stream, a);
    else if (prefetch && res && !dev_env_int("NRGBD_CONV3D_NOPFRES"))
        hipLaunchKernelGGL((conv3d_mfma_kernel<64, true, true>), dim3(nwg), dim3(256), lds, (hipStream_t)stream, a);
    else if (prefetch && !res)
        hipLaunchKernelGGL((conv3d_mfma_kernel<64, true>), dim3(nwg), dim3(256), lds, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((conv3d_mfma_kernel<64, false>), dim3(nwg), dim3(256), lds, (hipStream_t)stream, a);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_conv3d_3x3x3_cout1_f32(const float* x, const float* x_ss, int x_relu, const float* res,
                                            const float* res_ss, int res_relu, const float* w_tap_major,
                                            float* y, int D, int H, int W, int Cin, void* stream) {
    using namespace nrgbd;
    if (!x || !w_tap_major || !y) return NRGBD_E_NULL;
    if (D <= 0 || H <= 0 || W <= 0 || Cin != 64) return NRGBD_E_SHAPE;
    // the tap-projection kernel has no residual operand (the reference's classify branch has none, basic.py:92-94):
    // refuse one instead of silently ignoring it
    if (res || res_ss || res_relu) return NRGBD_E_ARG;
    Conv3dArgs a{x, x_ss, res, res_ss, nullptr, nullptr, y, nullptr, x_relu, res_relu, D, H, W, 1};
    // depth chunks: enough workgroups for ~3 rounds of the 512 resident ones, at least 8 slices each (a chunk re-reads 2)
    const int cols = ceil_div(W, kC1TW) * ceil_div(H, kC1TH);
    int nz = dev_env_int("NRGBD_C1_NZ");
    if (nz <= 0) nz = ceil_div(1536, cols);
    nz = nz < 1 ? 1 : (nz > ceil_div(D, 8) ? ceil_div(D, 8) : nz);
    const int dz = ceil_div(D, nz);
    nz = ceil_div(D, dz);
    // > 64 KB of dynamic LDS needs the opt-in; it is idempotent and costs ~1 us, so it is simply repeated per call
    // (no process-global flag: re-entrant from any thread on any device)
    hipError_t ea = set_max_dynamic_lds(reinterpret_cast<const void*>(conv3d_cout1_kernel), (int)kC1Lds);
    if (ea != hipSuccess) return (int)ea;
    hipLaunchKernelGGL(conv3d_cout1_kernel, dim3(cols * nz), dim3(256), kC1Lds, (hipStream_t)stream, a, w_tap_major, dz);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

extern "C" int nrgbd_bn3d_finalize(const float* stats, int num_workgroups, long count, const float* gamma,
                                   const float* beta, float eps, float momentum, float* running_mean,
                                   float* running_var, float* scale_shift, unsigned int* collapse_count, long long* batches_tracked, void* stream) {
    using namespace nrgbd;
    if (!stats || !gamma || !beta || !scale_shift) return NRGBD_E_NULL;
    if (num_workgroups <= 0 || count <= 0) return NRGBD_E_SHAPE;
    if ((running_mean == nullptr) != (running_var == nullptr)) return NRGBD_E_NULL;
    hipLaunchKernelGGL(bn3d_finalize_kernel, dim3(kCout), dim3(256), 0, (hipStream_t)stream, stats, num_workgroups,
                       (double)count, gamma, beta, eps, momentum, running_mean, running_var, scale_shift, collapse_count, batches_tracked);
    NRGBD_CHECK_LAUNCH();
    return NRGBD_OK;
}

// wino_pc.hpp — declarations shared by the Winograd-domain producer/consumer kernels (wino_pc.hip: F(2x2,3x3) per depth tap;
// wino_dw.hip: the same with F(2,3) along the depth axis on top): tile geometry, the LDS image of the transformed input,
// the launch arguments and the packed-fp32 helpers of the producer waves.
#pragma once
#include <type_traits>

#include "conv_tile.hpp"

namespace nrgbd {

constexpr int kPcTH = 8, kPcTW = 16;            // output pixels of a tile, in units of the dilation lattice
constexpr int kPcTiles = 32;                    // Winograd tiles per workgroup tile: ty = tile >> 3, tx = tile & 7
constexpr int kPcRawW = 20;                     // raw strip row pitch in pixels (18 used; a multiple of 4 keeps a row's bank map)
constexpr int kPcRawWave = 4 * kPcRawW * kCB;   // floats of one producer wave's 4-row strip (5 KB)
constexpr int kPcV = 16 * kPcTiles * kCB;       // floats of one V buffer [16 xi][32 tiles][16] (32 KB)
constexpr int kPcNBuf = 3;
constexpr int kPcItems = 4 * 18 * 4;            // (row, column, 16-byte word) items of a strip
constexpr int kPcNPF = (kPcItems + 63) / 64;    // per producer lane and stage (5)
#ifndef NRGBD_WPOS
#define NRGBD_WPOS 1   // MFMA gap (0..3) of a transform point in which the consumers request the weight line 7 points ahead
#endif
constexpr int kPcBD = 7, kPcNB = 8;             // weight ring: distance / slots
// SHARED strips (see wino_dw.hip): the 10 x 18 halo of a stage is split once over the 256 producer lanes into a strip all four
// producer waves share (3 words per lane instead of 5 from four overlapping 4-row private strips), published one stage ahead of
// its transform; the stage barrier is the only synchronisation.
#ifndef NRGBD_PC_SHARED
#define NRGBD_PC_SHARED 1   // 0: the private-strip producers (experimental A/B builds only)
#endif
constexpr int kPcShRows = kPcTH + 2;                       // halo rows of a tile
constexpr int kPcShStrip = kPcShRows * kPcRawW * kCB;      // floats of one shared strip: [10 rows][20 pixels][16] = 12.8 KB
constexpr int kPcShItems = kPcShRows * 18 * 4;             // (row, column, 16-byte word) items of a stage: 720
constexpr int kPcNPFx = NRGBD_PC_SHARED ? 3 : kPcNPF;      // items per producer lane and stage in wino_pc.hip
constexpr int kPcStrips = NRGBD_PC_SHARED ? 2 * kPcShStrip : 4 * kPcRawWave;   // floats of wino_pc.hip's strip region

struct WinoPcArgs {
    const float* x;       // [N][H][W][Cin] raw input (pre-activation); N = depth slices when KD = 3
    const float* x_ss;    // [Cin][2] (scale, shift) applied to x, or null
    const float* res;     // second operand added after activation, or null
    const float* res_ss;  // [Cin][2] for res, or null
    float* mat;           // materialised input act(x) + act(res), or null
    const float* wp;      // Winograd-domain weights [Cout/64][stage = cb*KD + kd][16 xi][4 waves][64 lanes][4]
    float* y;             // [N][H][W][Cout] raw convolution output
    float* stats;         // [2*Cout][spatial tiles] (column-major): per-channel sum and sum of squares of y per tile, or null
    int x_relu, res_relu;
    int N, H, W, Cin, Cout;
    int ntiles;           // spatial tiles x Cout/64
    int rows;             // statistics rows: spatial tiles (x 2 in wino_pc.hip's HALF form: one per (tile, row block))
    const float* bias;    // EPI = 1 (R-Net form): [Cout] added to the output, then LeakyReLU(0.01) if out_lrelu; no statistics
    int out_lrelu;
    int ldy, ycoff, cout_valid;   // EPI = 1 only: pixel stride of y (0 = Cout), first output column, columns that exist (0 = Cout):
                                  // the R-Net writes into concat buffers and pads 67 / 96 outputs to the 64-column groups
    int abl;              // developer ablation bits, honoured by -DNRGBD_DEV builds only: 1 = producers only, 2 = consumers only,
                          // 4 = no transform, 8 = no publish, 16 / 32 = s_setprio 2 for the consumers / producers
    float x_unit;         // CLAMP instantiations only: 2^-k; (scale, shift) of x are multiplied by it and the ReLU is the [0, 1]
                          // clamp of the packed FMA; the weight stream carries the factor 2^k (see nrgbd_conv_wino_dw_unit_f32)
};

struct PcTile { int n, y0, x0, py, px, cg, row; };

template <int KD, int DIL>
__device__ __forceinline__ PcTile pc_decode(int t, const WinoPcArgs& a) {
    PcTile r;
    const int ncg = (a.Cout + 63) >> 6;          // Cout = 32 (wino_pc.hip's HALF form): one column group
    const int tiles_x = (a.W + kPcTW * DIL - 1) / (kPcTW * DIL), tiles_y = (a.H + kPcTH * DIL - 1) / (kPcTH * DIL);
    r.row = t / ncg;
    r.cg = t - r.row * ncg;
    t = r.row;
    r.n = 0;
    if (KD == 3) { r.n = t % a.N; t /= a.N; }   // depth fastest: the three workgroups that read one slice are list neighbours
    int par = 0;
    if (DIL > 1) { par = t % (DIL * DIL); t /= DIL * DIL; }
    const int tx = t % tiles_x; t /= tiles_x;
    int ty = t;
    if (KD != 3) { ty = t % tiles_y; r.n = t / tiles_y; }
    r.py = par / DIL; r.px = par - r.py * DIL;
    r.y0 = ty * kPcTH * DIL; r.x0 = tx * kPcTW * DIL;
    return r;
}

// LDS image of V: [xi*32 + tile][16 floats]; the 16-byte slot s of a tile is stored at slot (s + 2*((tile >> 3) & 1)) & 3, so
// that every ds_read_b128 service group of the 16x16x4 A-operand pattern hits 16 different bank quads (as in generation 1)
__device__ __forceinline__ int pc_slot(int xi, int tile, int slot) {
    return ((xi * kPcTiles + tile) << 4) + (((slot + 2 * ((tile >> 3) & 1)) & 3) << 2);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// The producers' raw words come in as BUFFER loads: a resource descriptor (uniform base of the stage's slice / channel block,
// in SGPRs) + the lane's 32-bit byte offset — one instruction per 16-byte word.  As flat-address loads (uniform 64-bit base +
// zero-extended lane offset) hipcc materialises every address with a v_lshl_add_u64 (+ a v_mov of the zero high half) inside the
// stage loop: ~10 VALU instructions per unit on SIMDs whose issue slots the matrix pipe needs.  The compiler counts buffer loads in
// vmcnt like any other load.  num_records = 4 GB - 1: the launchers bound a slice / tensor below that.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pc_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
}
__device__ __forceinline__ f32x4 pc_bload(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}

// Position of element i of a [C][2] (scale, shift) table in the producers' LDS copy: per 4 channels the words are stored
// (s0, s1, t0, t1 | s2, s3, t2, t3), i.e. as the operand PAIRS of the packed FMAs — a lane's two ds_read_b128 are its
// (scale pair, shift pair) x 2 with no re-pairing moves (6 v_mov_b32 per unit and stage otherwise).
__device__ __forceinline__ int pc_ss_slot(int i) {
    const int c = i >> 1, t = i & 1;
    return ((c >> 2) << 3) + (((c >> 1) & 1) << 2) + (t << 1) + (c & 1);
}

// Packed fp32 helpers (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32: two lanes of a register pair per instruction).  The
// producers' VALU instructions only get the issue slots the co-resident consumer's MFMA stream leaves (measured: about one
// per MFMA), so every instruction saved there is stage time saved.  a*s + c with a splat s; s = -1 is the exact c - a.
__device__ __forceinline__ f32x4 pk_fma_s(f32x4 a, float s, f32x4 c) {
    const f32x2 m = {s, s};
    const f32x2 lo = __builtin_elementwise_fma(a.lo, m, c.lo), hi = __builtin_elementwise_fma(a.hi, m, c.hi);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
// max(x, 0) as ONE v_max_f32: fmaxf() on a packed-FMA result costs two (the compiler quiets a possible signalling NaN first)
__device__ __forceinline__ float relu1(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}
// a * b + c clamped to [0, 1] in the same instruction (the VOP3P clamp bit): with operands pre-scaled so that the result cannot
// exceed 1 this IS the ReLU — no v_max_f32 per element
__device__ __forceinline__ f32x2 pk_fma_clamp01(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x4 pk_add(f32x4 a, f32x4 b) {
    const f32x2 lo = a.lo + b.lo, hi = a.hi + b.hi;
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

}  // namespace nrgbd

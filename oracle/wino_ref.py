"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy, float64) of the Winograd F(2x2, 3x3) algorithm that
neuralrgbd_amd/csrc/wino_pc.hip runs for the reference's 3x3 stride-1 convolutions (models/basic.py:71-94 Conv3d per depth tap,
models/psm_submodule.py:10-16 Conv2d): the same G, B^T, A^T matrices, the same (xi_y, xi_x) ordering and the same weight-stream
layout, so that the host-side packing logic and the kernel's index conventions are pinned on a machine without a GPU.
Nothing under neuralrgbd_amd/ imports this file.

    Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A      per 2x2 output tile, 4x4 input patch d, 3x3 filter g
"""
import numpy as np

G = np.array([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]])
BT = np.array([[1.0, 0.0, -1.0, 0.0], [0.0, 1.0, 1.0, 0.0], [0.0, -1.0, 1.0, 0.0], [0.0, 1.0, 0.0, -1.0]])
AT = np.array([[1.0, 1.0, 1.0, 0.0], [0.0, 1.0, -1.0, -1.0]])


def conv2d_wino(x, w):
    """x [Cin, H, W], w [Cout, Cin, 3, 3], padding 1 -> y [Cout, H, W] through F(2x2, 3x3) (H, W even)."""
    cin, H, W = x.shape
    cout = w.shape[0]
    U = np.einsum("ay,ocyx,bx->ocab", G, w, G)                     # [co, ci, xi_y, xi_x]
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    y = np.zeros((cout, H, W))
    for ty in range(H // 2):
        for tx in range(W // 2):
            d = xp[:, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]         # [ci, 4, 4]
            V = np.einsum("ay,cyx,bx->cab", BT, d, BT)              # [ci, xi_y, xi_x]
            M = np.einsum("ocab,cab->oab", U, V)                    # [co, xi_y, xi_x]
            y[:, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = np.einsum("ia,oab,jb->oij", AT, M, AT)
    return y


def packed_index(co, ci, kd, xi, Cin, KD):
    """Position of U[co][ci][kd][xi] in the kernel's weight stream [cg][stage = cb*KD + kd][xi][wave][lane = kq*16 + j][e]
    (co = cg*64 + 16*wave + j, ci = cb*16 + 4*kq + e) — the layout documented in include/nrgbd.h."""
    cg, wave, j = co // 64, (co % 64) // 16, co % 16
    cb, kq, e = ci // 16, (ci % 16) // 4, ci % 4
    stages = (Cin // 16) * KD
    return ((((cg * stages + cb * KD + kd) * 16 + xi) * 4 + wave) * 64 + kq * 16 + j) * 4 + e


def conv3d_wino_dw(x, w):
    """x [Cin, D, H, W], w [Cout, Cin, 3, 3, 3], padding 1 -> y [Cout, D, H, W] through F(2x2x2, 3x3x3) = the plane transform
    above plus F(2, 3) along depth (D, H, W even): neuralrgbd_amd/csrc/wino_dw.hip.
        D_t = sum_j BT[t][j] d_j;  V_t = B^T D_t B;  U_t = sum_kd G[t][kd] (G g_kd G^T);  M_t = sum_ci U_t .* V_t
        y[z0] = A^T (M_0 + M_1 + M_2) A;  y[z0 + 1] = A^T (M_1 - M_2 - M_3) A"""
    cin, D, H, W = x.shape
    cout = w.shape[0]
    U = np.einsum("tz,ay,oczyx,bx->octab", G, G, w, G)             # [co, ci, t, xi_y, xi_x]
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (1, 1)))
    y = np.zeros((cout, D, H, W))
    for tz in range(D // 2):
        for ty in range(H // 2):
            for tx in range(W // 2):
                d = xp[:, 2 * tz:2 * tz + 4, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]       # [ci, 4, 4, 4]
                V = np.einsum("tz,ay,czyx,bx->ctab", BT, BT, d, BT)
                M = np.einsum("octab,ctab->otab", U, V)
                y[:, 2 * tz:2 * tz + 2, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = np.einsum("kt,ia,otab,jb->okij", AT, AT, M, AT)
    return y


def packed_index_dw(co, ci, t, xi, Cin):
    """Position of U_t[co][ci][xi] in the stream of nrgbd_conv_wino_dw_f32: [cg][stage = t*(Cin/16) + cb][xi][wave][lane][e]."""
    cg, wave, j = co // 64, (co % 64) // 16, co % 16
    cb, kq, e = ci // 16, (ci % 16) // 4, ci % 4
    ncb = Cin // 16
    return ((((cg * 4 * ncb + t * ncb + cb) * 16 + xi) * 4 + wave) * 64 + kq * 16 + j) * 4 + e

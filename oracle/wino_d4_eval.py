"""TEST INFRASTRUCTURE ONLY — numerics of F(4,3) along the DEPTH axis only for the K-Net (round 6), on the CPU.

    python -m oracle.wino_d4_eval [full]

csrc/wino_dw.hip runs the K-Net's 3x3x3 layers as F(2x2,3x3) in the plane x F(2,3) along depth: 8 multiplies per output voxel.
oracle/wino_f4_eval.py (round 5) rejected F(4x4,3x3) in the plane on numerics (3.9x further from float64 than the direct
convolution).  The smaller step — F(4,3) along depth ONLY, 6 multiplies per output voxel, 1.33x fewer matrix-core flops — applies the
ill-conditioned transform once instead of twice.  This script emulates it in float32 (input transform, channel contraction, inverse
transform each in fp32; weights transformed in float64 and rounded once) for several interpolation-point sets (Cook-Toom matrices built
exactly with sympy) on one 64->64 layer and, with `full`, on the whole K-Net of the config-S update frame like wino_f4_eval.py.
Result (profiles/r6_wino_d4_probe.txt): with the points (0, 1/2, -1/2, 3/2, -3/2) DPV sits 1.26e-5 (mean) from float64 against 1.01e-5
for the direct convolution and 0.93e-5 for today's form: ratio 1.25 — numerically viable.  It was not built because four output slices
per tile do not fit the workgroup's LDS beside the V buffers.
"""
import sys

import numpy as np
import torch
import torch.nn.functional as F

from oracle.wino_f4_eval import G2, BT2, AT2, G4, BT4, AT4, knet, conv3d_wino
from neuralrgbd_amd import camera, synth
from oracle import cpu_oracle as co, kvnet_oracle as ko
import fractions

def winograd_matrices(points):
    """Cook-Toom F(4,3) with 5 finite points + infinity, exact rationals -> float64 (AT [4x6], G [6x3], BT [6x6])."""
    import sympy
    n=6; m=4; r=3
    pts=[sympy.Rational(p) for p in points]
    # Following wincnn (Lavin): 
    from sympy import Matrix, eye, zeros, symbols, Poly
    a=pts
    def At(a,m,n): return Matrix(m, n, lambda i,j: a[j]**i if j<n-1 else (1 if i==m-1 else 0))
    def A(a,m,n): return At(a,m,n).T  # n x m
    x=symbols('x')
    def fdiag(a):
        f=[]
        for i in range(len(a)):
            p=1
            for j in range(len(a)):
                if i!=j: p*= (a[i]-a[j])
            f.append(p)
        return f
    na=len(a)  # n-1 =5 finite points
    f=fdiag(a)
    AT=At(a,m,n)
    Gm=Matrix(n, r, lambda i,j: (a[i]**j)/f[i] if i<n-1 else (1 if j==r-1 else 0))
    # B^T from polynomial: 
    # T = lagrange basis; use wincnn construction
    def T(a,n):
        return Matrix(n, n, lambda i,j: 0)  # placeholder
    # Build BT via solving: Y = AT [(G g) .* (BT d)] must equal conv for all g,d -> use known identity BT = (V^{-T}) construction:
    # B^T = inverse of A-like vandermonde extended. Use: for points a_0..a_{n-2}, inf: BT = inv(Vfull)^T scaled by f where Vfull = Vandermonde n x n
    V=Matrix(n, n, lambda i,j: a[i]**j if i<n-1 else (1 if j==n-1 else 0))
    Vinv=V.inv()
    # BT_i = f_i * (row i of Vinv^T)  for finite points; last row = row for infinity
    BT=Matrix(n, n, lambda i,j: (Vinv.T[i,j]*f[i]) if i<n-1 else Vinv.T[i,j])
    return (np.array(AT.tolist(),dtype=np.float64), np.array(Gm.tolist(),dtype=np.float64), np.array(BT.tolist(),dtype=np.float64))

def check(AT,G,BT):
    rng=np.random.RandomState(0)
    d=rng.randn(6); g=rng.randn(3)
    y=AT@((G@g)*(BT@d))
    ref=np.array([sum(d[i+k]*g[k] for k in range(3)) for i in range(4)])
    return np.abs(y-ref).max()

def conv3d_wino_d4(x, w, GT, BTd, ATd):
    """in-plane F(2x2,3x3), depth F(4,3) with the given matrices; every stage in fp32."""
    dt=x.dtype
    _,cin,D,H,W=x.shape; cout=w.shape[0]
    assert D%4==0
    U=torch.einsum("tz,ay,oczyx,bx->octab", GT, G2, w.double(), G2).to(dt)     # [co,ci,6,4,4]
    xp=F.pad(x[0],(1,1,1,1,1,1))
    p=xp.unfold(2,4,2).unfold(3,4,2)
    BTf,ATf=BT2.to(dt),AT2.to(dt)
    V=torch.einsum("ay,cdijyx->cdijax",BTf,p)
    V=torch.einsum("bx,cdijax->cdijab",BTf,V)
    Vd=V.unfold(1,6,4)                                                           # [ci, D/4, nty, ntx, 4,4, 6]
    Vd=torch.einsum("tz,cpijabz->cptijab",BTd.to(dt),Vd)
    M=torch.einsum("octab,cptijab->optijab",U,Vd)
    Y=torch.einsum("na,optijab->optijnb",ATf,M)
    Y=torch.einsum("eb,optijnb->optijne",ATf,Y)
    Y=torch.einsum("kt,optijne->opkijne",ATd.to(dt),Y)                            # [co, D/4, 4, nty,ntx,2,2]
    return Y.permute(0,1,2,3,5,4,6).reshape(1,cout,D,H,W)

def main():
    variants={}
    variants["lavin(0,1,-1,2,-2)"]=(torch.tensor(AT4.numpy()),G4,BT4)
    for name,pts in (("(0,1,-1,1/2,-1/2)",[0,1,-1,fractions.Fraction(1,2),fractions.Fraction(-1,2)]),("(0,1,-1,2,-1/2)",[0,1,-1,2,fractions.Fraction(-1,2)]), ("(0,1/2,-1/2,3/2,-3/2)",[0,fractions.Fraction(1,2),fractions.Fraction(-1,2),fractions.Fraction(3,2),fractions.Fraction(-3,2)])):
        AT,G,BT=winograd_matrices(pts)
        print(name,"identity check",check(AT,G,BT))
        variants[name]=(torch.from_numpy(AT),torch.from_numpy(G),torch.from_numpy(BT))
    # single layer
    import neuralrgbd_amd
    H,W,D=256,384,64
    cam=camera.scannet_intrinsics(W//4,H//4); d_candi=np.linspace(0.1,5.0,D)
    model=neuralrgbd_amd.KVNET(64,cam,d_candi,10.0,64,None,if_refined=True,refineNet_name="DPV",t_win_r=2)
    sd=synth.seeded_state_dict(model,0)
    torch.manual_seed(0)
    x=torch.relu(torch.randn(1,64,16,32,32)); wt=sd["kv_net.dres1.0.0.weight"]
    y64=F.conv3d(x.double(),wt.double(),None,1,1)
    def rep(name,y):
        e=(y.double()-y64).abs(); print("  one layer %-26s max|d|/max|y| %.2e mean|d|/mean|y| %.2e"%(name,e.max().item()/y64.abs().max().item(), e.mean().item()/y64.abs().mean().item()))
    rep("direct",F.conv3d(x,wt,None,1,1)); rep("F2",conv3d_wino(x,wt,2))
    for name,(AT,G,BT) in variants.items():
        rep("d4 "+name, conv3d_wino_d4(x,wt,G,BT,AT))
    if len(sys.argv)>1:
        w1,w2=(synth.noise_window(s,H,W) for s in (101,102))
        torch.set_num_threads(8)
        with torch.no_grad():
            o1=ko.step(sd,*w1,cam,d_candi,10.0,None); pred=o1[3]
            ref,src,poses=w2
            BV_cur,feats,full=ko.dnet(sd,ref,src,poses,cam,d_candi,10.0)
            V=src.shape[1]; rgb=full[:,-3:]
            KR,Kt=ko._terms(cam,poses[0])
            warped=co.warp_volume(rgb[:V].numpy(),KR,Kt,cam["unit_ray_array_2D"].numpy(),d_candi,cam["intrinsic_M"][0,2],cam["intrinsic_M"][1,2])
            h,w=rgb.shape[2:]
            vol=torch.cat((torch.from_numpy(warped).reshape(V*3,D,h,w),rgb[V][:,None].expand(3,D,h,w),(BV_cur-pred)),0)[None]
            direct=lambda x,wt: F.conv3d(x,wt,None,1,1)
            first=lambda f:(lambda x,wt: f(x,wt) if wt.shape[1]==64 else F.conv3d(x,wt,None,1,1))
            sd64={k:v.double() for k,v in sd.items()}
            g64=knet(sd64,vol.double(),direct)
            d64=torch.log_softmax(g64[0,0]+pred[0].double(),dim=0)
            res={}
            todo=[("direct",direct),("F2",first(lambda x,wt: conv3d_wino(x,wt,2)))]
            for name,(AT,G,BT) in variants.items():
                todo.append(("d4 "+name, first(lambda x,wt,AT=AT,G=G,BT=BT: conv3d_wino_d4(x,wt,G,BT,AT))))
            for name,conv in todo:
                g=knet(sd,vol,conv); a=torch.log_softmax(g[0,0]+pred[0],dim=0); res[name]=a
                e64=(a.double()-d64).abs(); eo=(a-res["direct"]).abs()
                print("  %-28s DPV vs oracle L1 %.3e max %.3e flips %d | vs fp64 mean %.3e max %.3e"%(name,eo.mean().item(),eo.max().item(),int((a.argmax(0)!=res["direct"].argmax(0)).sum()),e64.mean().item(),e64.max().item()))

if __name__ == "__main__":
    main()

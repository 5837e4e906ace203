import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from neuralrgbd_amd import ops
torch.manual_seed(0)
dev="cuda:0"
for (D,H,W) in ((4,12,24),(8,24,48),(16,64,96)):
    x=torch.randn(D,H,W,64,device=dev); gy=torch.randn(D,H,W,64,device=dev); w=torch.randn(64,64,3,3,3,device=dev)*0.05
    wp=ops.conv3d_pack_weights(w)
    ref_w=ops.conv3d_wgrad(x,gy); ref_y,_,_=ops.conv3d(x,wp,want_stats=False)
    ss=torch.randn(64,2,device=dev); res=torch.randn(D,H,W,64,device=dev)
    ref_y2,_,_=ops.conv3d(x,wp,x_ss=ss,res=res,want_stats=False)
    bad_w=bad_y=bad_y2=0
    for i in range(30):
        # interleave other work to perturb timing
        junk=torch.randn(1<<20,device=dev).sum()
        dw=ops.conv3d_wgrad(x,gy); y,_,_=ops.conv3d(x,wp,want_stats=False); y2,_,_=ops.conv3d(x,wp,x_ss=ss,res=res,want_stats=False)
        bad_w+=int(not torch.equal(dw,ref_w)); bad_y+=int(not torch.equal(y,ref_y)); bad_y2+=int(not torch.equal(y2,ref_y2))
        if not torch.equal(dw,ref_w): print('  wgrad diff max', (dw-ref_w).abs().max().item(), 'count', int((dw!=ref_w).sum()))
    print((D,H,W),'nondeterministic runs: wgrad',bad_w,'conv PF',bad_y,'conv res',bad_y2)

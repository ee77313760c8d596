import torch, os
from hd_yolo_amd import _lib, ops
from hd_yolo_amd.bench_util import time_record
DEV='cuda:0'
N,H,W,K=64,640,640,32
img=torch.rand(N,3,H,W,device=DEV); prep=torch.zeros((N,H+4,W+4,4),dtype=torch.bfloat16,device=DEV)
ops.run([ops.rec_stem_prep(img,prep)])
y=torch.randn(N,H//2,W//2,K,device=DEV).bfloat16(); dz=(torch.randn(N,H//2,W//2,K,device=DEV)*0.1).bfloat16()
sc=torch.ones(K,device=DEV); sh=torch.zeros(K,device=DEV); mu=torch.zeros(K,device=DEV); isd=torch.ones(K,device=DEV)
dg=torch.zeros(K,device=DEV); db=torch.zeros(K,device=DEV); ga=torch.zeros(K,3,6,6,device=DEV)
ws=torch.empty(ops.stem_bwd_onepass_ws_floats(N,H,W,K),dtype=torch.float32,device=DEV)
rec=ops.rec_stem_bwd_onepass(prep,dz,y,sc,sh,mu,isd,(H,W),dg,db,ga,None,ws)
for v in (0,1,2,3,4,8,12,15):
    with _lib.option('HDY_DEEP_DEBUG',v):
        print('dbg',v, round(time_record(rec,reps=10),1),'us (4 launches: pass + finalize + reduce + combine)',flush=True)
M=N*(H//2)*(W//2)
ws_bn=torch.empty(ops.bn_bwd_ws_floats(M,K),dtype=torch.float32,device=DEV)
r1=ops.rec_bn_act_bwd(dz,y,sc,sh,mu,isd,None,dg,db,ws_bn)
c1,c2=ops.bn_bwd_coeffs(ws_bn,M,K)
ws2=torch.empty(ops.wgrad_ws_bytes(N,H,W,3,K,6,6,2,2,torch.bfloat16,stem=True)//4+16,dtype=torch.float32,device=DEV)
r2=ops.rec_conv_wgrad_stem_fused(prep,dz,y,sc,sh,mu,isd,c1,c2,(H,W),ga,None,ws2)
print('statistics pass', round(time_record(r1,reps=10),1),'stem wgrad fused', round(time_record(r2,reps=10),1))

import torch
from hd_yolo_amd import _lib, ops
from hd_yolo_amd.bench_util import time_record
DEV='cuda:0'
print('M K ld_y pair | U=2(default) U=1 U=3 U=4  (us; bytes at U=2 in TB/s)')
for M,K,Ka,pitch in [(6553600,32,None,32),(1638400,64,None,64),(1638400,64,32,64),(1638400,32,None,32),(409600,128,64,128),(409600,64,None,64),(102400,256,128,256),(102400,128,None,128),(25600,256,None,256),(25600,512,256,512),(1638400,48,None,48),(1638400,96,48,96),(409600,96,None,96)]:
    y=torch.randn(1,1,M,pitch,device=DEV).bfloat16()[...,:K]
    sc=torch.ones(K,device=DEV); sh=torch.zeros(K,device=DEV)
    if Ka:
        za=torch.empty(1,1,M,Ka,dtype=torch.bfloat16,device=DEV)
        zb=torch.empty(1,1,M,2*(K-Ka),dtype=torch.bfloat16,device=DEV)[...,:K-Ka]
        rec=ops.rec_bn_act_fwd_pair(y,sc,sh,za,zb)
    else:
        z=torch.empty(1,1,M,K,dtype=torch.bfloat16,device=DEV)
        rec=ops.rec_bn_act_fwd(y,sc,sh,z)
    row=[]
    for v in (0,1,2,3):
        with _lib.option('HDY_DEEP_DEBUG',v):
            row.append(time_record(rec,reps=20))
    print(f'{M:8d} {K:4d} {str(Ka):>4} | '+' '.join(f'{t:7.1f}' for t in row)+f'   {4.0*M*K/row[0]/1e6:.2f} TB/s',flush=True)

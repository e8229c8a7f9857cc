import torch, sys, os
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from dynavsr_amd import engine, synth
from test_gpu_edvr import make_net
from conftest import relerr
net = make_net(0)
k,h,w=2,96,128
base=[p.detach().clone() for p in net.ordered_parameters()]
g=torch.Generator(device="cuda").manual_seed(5)
stacked=[]
for p in base:
    s_=p.unsqueeze(0).repeat((k,)+(1,)*p.dim()); s_=s_*(1.0+0.05*torch.randn(s_.shape,device="cuda",generator=g)); stacked.append(s_.contiguous())
x=synth.clip(63,k,5,h,w).cuda()
cfg=net._cfg()
pk=engine.get_plan(cfg,k,h,w,grad_groups=k,weight_sets=k)
wsk=torch.empty(pk.workspace_bytes(True),dtype=torch.uint8,device='cuda')
outk=torch.empty(k,3,4*h,4*w,device='cuda')
pk.forward(stacked,x,outk,wsk)
p1=engine.get_plan(cfg,1,h,w)
ws1=torch.empty(p1.workspace_bytes(True),dtype=torch.uint8,device='cuda')
names=[]
for (kind,nm,fl,by) in pk.op_info():
    n=nm.split('[')[0]
    if n not in names: names.append(n)
info1={nm.split('[')[0]:nm for (_k,nm,_f,_b) in p1.op_info()}
infok={nm.split('[')[0]:nm for (_k,nm,_f,_b) in pk.op_info()}
for i in range(k):
    out1=torch.empty(1,3,4*h,4*w,device='cuda')
    p1.forward([s_[i].contiguous() for s_ in stacked],x[i:i+1].contiguous(),out1,ws1)
    print("slice",i,"final",relerr(outk[i:i+1],out1))
    shown=0
    for n in names:
        try:
            tk=pk.tensor(wsk,n); t1=p1.tensor(ws1,n)
        except Exception as e:
            continue
        per=tk.numel()//k
        if per!=t1.numel(): continue
        e=relerr(tk[i*per:(i+1)*per],t1)
        if e>1e-5 and shown<6:
            print("   ",n,infok.get(n),info1.get(n),e); shown+=1

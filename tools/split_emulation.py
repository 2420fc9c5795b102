import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import numpy as np, torch
from clair_amd import synth, weights
from oracle import model_np
torch.set_num_threads(8)
def f16(x): return x.to(torch.float16).to(torch.float32)
def split2(x):
    a1=f16(x); a2=f16(x-a1); return a1,a2
def mm_split(A,B,terms=3):
    a1,a2=split2(A); b1,b2=split2(B)
    out=a2@b1
    out=out+a1@b2
    if terms==4: out=out+a2@b2
    out=out+a1@b1
    return out
def bf(x): return x.to(torch.bfloat16).to(torch.float32)
def mm_bf3(A,B):
    def s3(x):
        p1=bf(x); r=x-p1; p2=bf(r); r=r-p2; p3=bf(r); return p1,p2,p3
    a=s3(A); b=s3(B)
    out=a[2]@b[0]+a[0]@b[2]; out=out+a[1]@b[1]; out=out+a[1]@b[0]; out=out+a[0]@b[1]; out=out+a[0]@b[0]; return out
def forward(w,x,mm_rec,mm_proj):
    T=33;H=128
    n=x.shape[0]
    s=torch.from_numpy(x).reshape(n,T,32).transpose(0,1).contiguous()
    def layer(inp,name,D):
        outs=[]
        for d,rev in (("fw",False),("bw",True)):
            K=torch.from_numpy(w[f"{name}_{d}_kernel"]); b=torch.from_numpy(w[f"{name}_{d}_bias"])
            Wx,Wh=K[:D],K[D:]
            zx=mm_proj(inp.reshape(T*n,D),Wx).reshape(T,n,512)+b
            h=torch.zeros(n,H); c=torch.zeros(n,H); out=torch.empty(T,n,H)
            for t in (range(T-1,-1,-1) if rev else range(T)):
                z=zx[t]+ (mm_rec(h,Wh) if t!=(T-1 if rev else 0) else 0)
                i,g,f,o=z[:,:H],z[:,H:2*H],z[:,2*H:3*H],z[:,3*H:]
                c=torch.sigmoid(f)*c+torch.sigmoid(i)*torch.tanh(g); h=torch.sigmoid(o)*torch.tanh(c); out[t]=h
            outs.append(out)
        return torch.cat(outs,2)
    a1=layer(s,"lstm1",32); a2=layer(a1,"lstm2",256)
    a2b=a2.transpose(0,1)
    l3=torch.selu(torch.einsum("ntc,ctu->nuc",a2b,torch.from_numpy(w["l3_kernel"]))+torch.from_numpy(w["l3_bias"]).t()[None])
    l4=torch.selu(l3.reshape(n,7680)@torch.from_numpy(w["l4_kernel"])+torch.from_numpy(w["l4_bias"]))
    outs=[]
    for k,nm in enumerate(("gt21","genotype","len1","len2")):
        l5=torch.selu(l4@torch.from_numpy(w["l5_kernel"][k])+torch.from_numpy(w["l5_bias"][k]))
        lg=torch.selu(l5@torch.from_numpy(w[f"head_{nm}_kernel"])+torch.from_numpy(w[f"head_{nm}_bias"]))
        outs.append(torch.softmax(lg,1).numpy())
    return outs
w=weights.synthetic_weights(seed=20250928, head_gain=4.0)
x,_=synth.synthetic_input(512,'ont',seed=5)
ref=model_np.forward(w,x,dtype=np.float64)
f32=lambda A,B:A@B
with torch.no_grad():
    for name,mr,mp in (("fp32",f32,f32),("rec fp16x3",lambda A,B:mm_split(A,B),f32),("rec+proj fp16x3",lambda A,B:mm_split(A,B),lambda A,B:mm_split(A,B)),("rec fp16x4",lambda A,B:mm_split(A,B,4),f32),("rec bf16x6",mm_bf3,f32)):
        o=forward(w,x,mr,mp)
        print(name, [float(np.abs(a-b).max()) for a,b in zip(o,ref)])

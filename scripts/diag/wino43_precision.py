"""F(4,3) vs F(2,3) along y with f16x3 operands, emulated on the CPU in torch (fp32 / fp64 accumulation) against an fp64 convolution:\nthe precision study behind csrc/conv_wino4_f16x3.hip (run: python scripts/diag/wino43_precision.py)."""
import torch, math
torch.manual_seed(0)
N,H,W,Cin,Cout=2,32,32,256,64
x=torch.randn(N,Cin,H,W,dtype=torch.float64)
x=torch.nn.functional.silu(x*1.5)            # post GN+SiLU like
w=torch.randn(Cout,Cin,3,3,dtype=torch.float64)/math.sqrt(9*Cin)
ref=torch.nn.functional.conv2d(x,w,padding=1)
x32=x.float()
def split(v,scale):
    s=(v.float()*scale)
    hi=s.half(); lo=(s-hi.float()).half()
    return hi.float(),lo.float()
def conv1x3(v,u):   # v [N,C,R,W], u [Cout,C,3] along x, fp32 accumulate
    return torch.nn.functional.conv2d(v.unsqueeze(2),u.unsqueeze(2),padding=(0,1)).squeeze(2)
def wino(BT,G,AT,m,r,vscale,acc=torch.float32):
    n=m+r-1
    BT=torch.tensor(BT,dtype=torch.float64);G=torch.tensor(G,dtype=torch.float64);AT=torch.tensor(AT,dtype=torch.float64)
    xp=torch.nn.functional.pad(x32,(0,0,1,1+ (m - H%m)%m))
    out=torch.zeros(N,Cout,H,W,dtype=torch.float64)
    # U_f = sum_k G[f,k] g_k  (fp64 then scaled per out row)
    U=torch.einsum('fk,ockx->focx',G,w)       # [n,Cout,Cin,3]
    sc=U.abs().amax(dim=(0,2,3))               # per cout
    e=torch.floor(torch.log2(16384.0/sc)); rs=2.0**e
    Us=U*rs.view(1,-1,1,1)
    uh=Us.float().half().float(); ul=(Us.float()-uh).half().float()
    maxV=0
    for j in range(H//m):
        d=xp[:,:,j*m:j*m+n,:]                  # rows j*m-1 .. (fp32)
        V=torch.einsum('fk,nckw->nfcw',BT.float(),d)   # fp32 transform (approx order)
        maxV=max(maxV,V.abs().max().item())
        vh,vl=split(V,vscale)
        M=[]
        for f in range(n):
            a=conv1x3(vh[:,f].to(acc),uh[f].to(acc))+conv1x3(vl[:,f].to(acc),uh[f].to(acc))+conv1x3(vh[:,f].to(acc),ul[f].to(acc))
            M.append(a.float() if acc==torch.float32 else a)
        M=torch.stack(M,1)                    # [N,n,Cout,W]
        Y=torch.einsum('rf,nfow->norw',AT.to(M.dtype),M).double()/(vscale*rs.view(1,-1,1,1))
        out[:,:,j*m:j*m+m,:]=Y
    return out,maxV
F23=([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],[[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],[[1,1,1,0],[0,1,-1,-1]])
F43=([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]],
     [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]],
     [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]])
sc=ref.abs().max().item()
d32=torch.nn.functional.conv2d(x32,w.float(),padding=1).double()
print("fp32 direct (cpu):",((d32-ref).abs().max()/sc).item(), ((d32-ref).pow(2).mean().sqrt()/ref.pow(2).mean().sqrt()).item())
for name,F,m,vs in (("F23 f16x3",F23,2,32.0),("F43 f16x3",F43,4,8.0)):
    for acc in (torch.float32,torch.float64):
        o,mv=wino(*F,m,3,vs,acc)
        print(name,acc,"max rel",((o-ref).abs().max()/sc).item(),"rms rel",((o-ref).pow(2).mean().sqrt()/ref.pow(2).mean().sqrt()).item(),"max|V|",mv)

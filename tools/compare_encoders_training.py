"""Free-running 300-step training runs of the same SAE on the same batches with the three fp32-accurate encoder modes.

    python tools/compare_encoders_training.py

Prints reconstruction MSE every 20 steps for f32 (exact fp32 MFMA), f16x3 and f16r.  Trajectories separate at the 1e-5..
1e-4 level (rounding-level changes flip near-tied TopK selections, and training amplifies them) -- equally for both fast
modes, with no systematic offset."""
import torch, math
from saev_amd.engine import EngineConfig, SaeEngine
torch.manual_seed(0)
D,S,B,K=256,8192,4096,32
g=torch.Generator(device="cuda").manual_seed(1)
A=torch.randn(D,4*D,device="cuda",generator=g); A/=A.norm(dim=0,keepdim=True)
def batch(i):
    gg=torch.Generator(device="cuda").manual_seed(100+i)
    s=torch.zeros(B,4*D,device="cuda")
    idx=torch.randint(0,4*D,(B,16),device="cuda",generator=gg)
    s.scatter_(1,idx,torch.empty(B,16,device="cuda").exponential_(1.0,generator=gg))
    return s@A.T+0.1*torch.randn(B,D,device="cuda",generator=gg)
W=(torch.rand(S,D,device="cuda",generator=g)*2-1)*(6.0/D)**0.5; W/=W.norm(dim=1,keepdim=True)
res={}
for enc in ("f32","f16x3","f16r"):
    eng=SaeEngine(EngineConfig(d_model=D,d_sae=S,top_k=K,k_aux=256,dead_threshold_tokens=40000,max_batch=B,encoder=enc), torch.device("cuda:0"))
    eng.view("W_dec").copy_(W); eng.view("W_enc").copy_(W.t())
    log=[]
    for i in range(300):
        lr=4e-4*min(1.0,i/50)
        eng.train_step(batch(i), lr, 1.0)
        if i%20==19:
            st=eng.read_stats(); log.append((st.mse, st.aux, st.n_dead))
    res[enc]=log
for i in range(len(res["f32"])):
    a,b,c=res["f32"][i],res["f16x3"][i],res["f16r"][i]
    print(f"step {20*i+19:3d}  mse f32 {a[0]:.6f} f16x3 {b[0]:.6f} f16r {c[0]:.6f} | rel {abs(b[0]-a[0])/a[0]:.1e} {abs(c[0]-a[0])/a[0]:.1e} | n_dead {a[2]} {b[2]} {c[2]} | aux {a[1]:.4e} {c[1]:.4e}")

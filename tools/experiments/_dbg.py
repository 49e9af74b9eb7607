import torch, math, sys
sys.path.insert(0, ".")
from saev_amd.engine import EngineConfig, SaeEngine
mode = sys.argv[1]
d, s, k, b = 256, 4096, 16, 700
eng = SaeEngine(EngineConfig(d_model=d, d_sae=s, top_k=k, max_batch=b, k_aux=0, encoder=mode))
g = torch.Generator(device='cuda').manual_seed(3)
W = (torch.rand(s, d, device='cuda', generator=g) * 2 - 1) * math.sqrt(6.0 / d)
eng.view('W_dec').copy_(W); eng.view('W_enc').copy_(W.t())
x = torch.randn(b, d, device='cuda', generator=g)
idx, val = eng.encode_topk(x)
torch.cuda.synchronize()
print("ok", mode, idx[0, :4].tolist())

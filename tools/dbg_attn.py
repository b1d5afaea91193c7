import os, sys, math, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "thinkdiff-mlre_amd"))
from thinkdiff import _hip
torch.manual_seed(0)
S=64
def run(q,k,v):
    out = torch.zeros(1,S,128,dtype=torch.bfloat16,device="cuda")
    _hip.attention(q.bfloat16().cuda().contiguous(),k.bfloat16().cuda().contiguous(),v.bfloat16().cuda().contiguous(),out,1,1)
    torch.cuda.synchronize(); return out.float().cpu()
def ref(q,k,v):
    q,k,v=[t.bfloat16().float() for t in (q,k,v)]
    return torch.softmax(q@k.transpose(-1,-2)/math.sqrt(128),-1)@v
z=torch.zeros(1,S,128)
# (a) uniform softmax, V[key][d] = key  -> out = mean(key) = 31.5 ; V[key][d]=d -> out[d]=d
v=torch.arange(S).float()[None,:,None].expand(1,S,128).clone()
o=run(z,z,v); print("a1 uniform,V=key: expect 31.5 got", o[0,0,:4], o[0,5,:4], (o-31.5).abs().max())
v=torch.arange(128).float()[None,None,:].expand(1,S,128).clone()
o=run(z,z,v); print("a2 uniform,V=d: max err", (o-v).abs().max(), o[0,0,:8], o[0,0,32:40])
# (b) one-hot attention: q_i strongly matches k_i -> out_i = v_i
q=torch.zeros(1,S,128); k=torch.zeros(1,S,128)
for i in range(S): q[0,i,i]=40.; k[0,i,i]=40.
v=torch.randn(1,S,128)
o=run(q,k,v); r=ref(q,k,v); print("b onehot: err", (o-r).abs().max())
bad=(o-r).abs().amax(-1)[0]; print("bad rows", (bad>0.05).nonzero().flatten().tolist()[:20])
# which key does each row pick?
vk=torch.arange(S).float()[None,:,None].expand(1,S,128).clone()
o=run(q,k,vk); print("b2 picked key per row:", o[0,:,0].tolist())
# (c) random
q=torch.randn(1,S,128);k=torch.randn(1,S,128);v=torch.randn(1,S,128)
o=run(q,k,v); r=ref(q,k,v); print("c random: err", (o-r).abs().max())
o=run(q,k,vk); r=ref(q,k,vk); print("c2 random q,k, V=key: err", (o-r).abs().max(), o[0,:4,0], r[0,:4,0])

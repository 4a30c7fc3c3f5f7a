import torch, time
q=torch.randn(320,12,128,64,device='cuda',dtype=torch.bfloat16); k=torch.randn_like(q); v=torch.randn_like(q)
def t(fn,n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)*1e3/n
F=torch.nn.functional
print("default sdpa us", t(lambda: F.scaled_dot_product_attention(q,k,v)))
try:
    print("preferred lib:", torch.backends.cuda.preferred_rocm_fa_library())
    torch.backends.cuda.preferred_rocm_fa_library("ck")
    print("ck sdpa us", t(lambda: F.scaled_dot_product_attention(q,k,v)))
except Exception as e: print("ck not available:", repr(e)[:200])
from torch.nn.attention import sdpa_kernel, SDPBackend
for be in (SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION, SDPBackend.MATH):
    try:
        with sdpa_kernel(be):
            print(be, t(lambda: F.scaled_dot_product_attention(q,k,v)))
    except Exception as e: print(be, "failed", repr(e)[:120])
# layout as HF produces: [B,T,H,D] transposed views
qq=torch.randn(320,128,12,64,device='cuda',dtype=torch.bfloat16).transpose(1,2); kk=torch.randn(320,128,12,64,device='cuda',dtype=torch.bfloat16).transpose(1,2); vv=torch.randn(320,128,12,64,device='cuda',dtype=torch.bfloat16).transpose(1,2)
print("strided BTHD views", t(lambda: F.scaled_dot_product_attention(qq,kk,vv)))
m=torch.zeros(320,1,1,128,device='cuda',dtype=torch.bfloat16)
print("with additive mask", t(lambda: F.scaled_dot_product_attention(qq,kk,vv,attn_mask=m)))

import os, sys
sys.path.insert(0, os.getcwd())
from bayeformers_amd import ops
print("bf_attention_fwd (BTHD views)", t(lambda: ops.attention_forward(qq, kk, vv, None, 0.125)))
mk = torch.zeros(320, 128, device='cuda')
print("bf_attention_fwd + key mask  ", t(lambda: ops.attention_forward(qq, kk, vv, mk, 0.125)))
q3=torch.randn(160,384,16,64,device='cuda',dtype=torch.float16).transpose(1,2); k3=torch.randn(160,384,16,64,device='cuda',dtype=torch.float16).transpose(1,2); v3=torch.randn(160,384,16,64,device='cuda',dtype=torch.float16).transpose(1,2)
print("BERT-large shape sdpa", t(lambda: F.scaled_dot_product_attention(q3,k3,v3)), " bf", t(lambda: ops.attention_forward(q3,k3,v3,None,0.125)))

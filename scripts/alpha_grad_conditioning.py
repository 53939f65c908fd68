"""CPU analysis of the AlphaBlender gradient d(alpha) = sum dy * (x_spatial - x_temporal) on the small test configuration:
its conditioning (sum|terms| / |sum|) and what bf16 rounding of the inputs alone does to it (fp32 oracle autograd as the truth).

    python scripts/alpha_grad_conditioning.py rowwise|pointwise
"""
import torch, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import ctsd_oracle as O
from tests.common import small_config, small_inputs
bf16=torch.bfloat16
cfg = small_config(temporal_attention_type=sys.argv[1])
sd = {k: v.to(bf16).float() for k, v in O.make_state_dict(cfg, 0).items()}
inp = small_inputs(cfg, 0)
inp = {k: (v.to(bf16).float() if v.is_floating_point() and k not in ("timestep", "added_time_ids") else v) for k, v in inp.items()}
g = torch.Generator().manual_seed(11)
wgt = torch.randn(inp["sample"].shape, generator=g)
caps=[]
orig=O.alpha_blender
def hooked(sd_, p, a, b, image_only):
    out = orig(sd_, p, a, b, image_only)
    rec={"p":p,"a":a.detach(),"b":b.detach()}
    out.register_hook(lambda gr, rec=rec: rec.__setitem__("dy", gr.detach()))
    caps.append(rec)
    return out
O.alpha_blender=hooked
sdo = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
ref = O.dit_forward(sdo, cfg, **inp)
(ref*wgt).sum().backward()
r=lambda x: x.to(bf16).float()
for rec in caps:
    a,b,dy=rec["a"],rec["b"],rec["dy"]
    true=(dy*(a-b)).sum().item()
    cond=(dy*(a-b)).abs().sum().item()/abs(true)
    e1=(r(dy)*(r(a)-r(b))).sum().item()
    e2=(r(dy)*(a-b)).sum().item()
    e3=(r(dy)*r(a)).sum().item()-(r(dy)*r(b)).sum().item()
    print(rec["p"], "true %.4e cond %.1f | bf16 a,b,dy relerr %.3e | bf16 dy only %.3e | ratio |a|/|a-b| %.2f"%(true,cond,abs(e1-true)/abs(true),abs(e2-true)/abs(true),a.norm()/(a-b).norm()), "mix grad", sdo[rec["p"]+".mix_factor"].grad.item())

"""Diagnostic: eager-cond / eager-uncond / folded attention at the tiny-UNet shapes vs the CPU oracle."""
import os, sys, math
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "paint-with-words-sd_amd"), REPO, os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch, warnings
warnings.simplefilter("ignore")
import pww_hip
from pww_hip.sampler import ROW_GATE
from oracle import pww_oracle as O
from sd_standin import CrossAttention
dev = torch.device("cuda:0")
wf = lambda w, sigma, qk: 0.4 * w * math.log(1 + sigma) * qk.max()
for (N, C, H, ctxd) in ((4096, 32, 4, 64), (1024, 64, 4, 64), (256, 128, 4, 64), (64, 128, 4, 64), (4096, 320, 8, 768)):
    for dtype in (torch.float32, torch.float16):
        torch.manual_seed(0)
        mod = CrossAttention(C, ctxd, H, C // H).requires_grad_(False)
        mod.to_q.weight.mul_(2.0)
        hid = torch.randn(2, N, C)
        ctx_c, ctx_u = torch.randn(1, 77, ctxd), torch.randn(1, 77, ctxd)
        w = (torch.rand(N, 77) < 0.15).float() * torch.rand(N, 77) * 1.5
        sig = torch.tensor(7.84)
        def ref(i, cond):
            m = mod if dtype == torch.float32 else mod.to(torch.float32)
            d = {"CONTEXT_TENSOR": ctx_c if cond else ctx_u, f"CROSS_ATTENTION_WEIGHT_{N}": w if cond else 0, "SIGMA": sig,
                 "WEIGHT_FUNCTION": wf if cond else (lambda w, s, qk: 0.0)}
            return O.inj_forward(m, hid[i:i+1], d)
        md = CrossAttention(C, ctxd, H, C // H).requires_grad_(False)
        md.load_state_dict(mod.state_dict()); md = md.to(dev, dtype)
        hd = hid.to(dev, dtype)
        def hip(i, cond):
            d = {"CONTEXT_TENSOR": (ctx_c if cond else ctx_u).to(dev, dtype), f"CROSS_ATTENTION_WEIGHT_{N}": w.to(dev) if cond else 0,
                 "SIGMA": sig, "WEIGHT_FUNCTION": wf if cond else (lambda w, s, qk: 0.0)}
            return pww_hip.inj_forward(md, hd[i:i+1], d).float().cpu()
        fold = {"CONTEXT_TENSOR": torch.cat([ctx_c.expand(2, -1, -1), ctx_u.expand(2, -1, -1)]).to(dev, dtype).contiguous(),
                f"CROSS_ATTENTION_WEIGHT_{N}": w.to(dev), "SIGMA": sig, "WEIGHT_FUNCTION": wf,
                ROW_GATE: torch.tensor([1., 1., 0., 0.], device=dev)}
        yf = pww_hip.inj_forward(md, torch.cat([hd, hd]), fold).float().cpu()
        out = []
        for i in range(2):
            rc, ru = ref(i, True), ref(i, False)
            sc = rc.abs().max().item()
            out.append("img%d: eager-cond %.2e eager-uncond %.2e | folded-cond %.2e folded-uncond %.2e (max|O| %.2f, cond-uncond %.2e)" % (
                i, (hip(i, True) - rc).abs().max() / sc, (hip(i, False) - ru).abs().max() / sc,
                (yf[i:i+1] - rc).abs().max() / sc, (yf[2+i:3+i] - ru).abs().max() / sc, sc, (rc - ru).abs().max() / sc))
        print(N, C, H, dtype, "\n   " + "\n   ".join(out), flush=True)

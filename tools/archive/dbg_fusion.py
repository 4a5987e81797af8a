import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch, torch.nn.functional as F
from common import *
from applecider_amd.models.applecider import AppleCider
from applecider_amd.synthetic import make_batch
from applecider_amd import hipops as H
from oracle import functional as O
dev = torch.device('cuda')
fc = {"mode": "all", "p_d_model": 128, "p_n_heads": 8, "p_n_layers": 4, "p_dropout": 0.0, "max_len": 257, "num_classes": 5, "hidden_dim": 64, "fusion": "avg", "lr": 1e-3}
m = AppleCider(fc); sd = closed_form_sd(m); m.load_state_dict(sd); m = m.to(dev).eval()
b = make_batch(4, seed=7)
ocfg = {"p_n_heads": 8, "p_n_layers": 4, "fusion": "avg", "kernel_sizes_per_stage": cfg_default()["model"]["SpectraNet"]["kernel_sizes_per_stage"]}
args = [T(b[k]) for k in ("photometry", "pad_mask", "metadata", "image", "spectra")]
osd = {k: v.clone().double().requires_grad_() for k, v in sd.items()}
a64 = [a.double() if a.dtype.is_floating_point else a for a in args]
sub = O._sub
p_emb = O.baselinecls_forward(sub(osd, "photometry_encoder."), a64[0], a64[1], 8, 4, classification=False)
s_emb = O.spectranet_forward(sub(osd, "spectra_encoder."), a64[4], ocfg["kernel_sizes_per_stage"])
im_emb = O.astrominn_forward(sub(osd, "img_metadata_encoder."), a64[2], a64[3])
ref = O.fusion_head(osd, p_emb, s_emb, im_emb, "avg")
loss = F.cross_entropy(ref, T(b["label"])); loss.backward()
dargs = [a.to(dev) for a in args]
pe = m.photometry_encoder((dargs[0], dargs[1], None)); se = m.spectra_encoder((dargs[4], None, None)); ie = m.img_metadata_encoder((dargs[2], dargs[3], None))
print("p_emb", relerr(pe.detach().cpu().numpy(), p_emb.detach().numpy()))
print("s_emb", relerr(se.detach().cpu().numpy(), s_emb.detach().numpy()))
print("im_emb", relerr(ie.detach().cpu().numpy(), im_emb.detach().numpy()))
logits = m(*dargs)
print("logits", relerr(logits.detach().cpu().numpy(), ref.detach().numpy()))
l = H.cross_entropy_index(logits, T(b["label"]).to(dev)); m.optimizer.zero_grad(); l.backward()
gr = grads_by_ref_name(m)
for k in sorted(gr):
    if osd[k].grad is None: continue
    e = relerr(gr[k].detach().cpu().numpy(), osd[k].grad.numpy())
    if e > 5e-4: print(f"{e:.2e}", k)

import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch, torch.nn.functional as F
from common import *
from applecider_amd.models.spectranet import SpectraNet
from applecider_amd.synthetic import make_batch
from applecider_amd import hipops as H
from oracle import functional as O
dev = torch.device('cuda')
cfg = cfg_default()
m = SpectraNet(cfg); sd = closed_form_sd(m); m.load_state_dict(sd); m = m.to(dev).eval()
b = make_batch(4, seed=7)
ks = cfg["model"]["SpectraNet"]["kernel_sizes_per_stage"]
x = T(b["spectra"])
print("zero rows:", [bool((x[i]==0).all()) for i in range(4)])
osd = {k: v.clone().double().requires_grad_() for k, v in sd.items()}
logits, stages = O.spectranet_forward(osd, x.double(), ks, return_stages=True)
for s in stages: s.retain_grad()
loss = F.cross_entropy(logits, T(b["label"])); loss.backward()
# GPU
h = x.to(dev).reshape(4, 4096, 1)
gst = []
for st in m.stages:
    h = st(h); h.retain_grad(); gst.append(h)
z = H.global_max(h)
head = m.classifier
out = head[4](head[3](head[1](head[0](z), act="gelu")))
print("logits", relerr(out.detach().cpu().numpy(), logits.detach().numpy()))
l = H.cross_entropy_index(out, T(b["label"]).to(dev)); l.backward()
for i,(g, o) in enumerate(zip(gst, stages)):
    a = g.detach().permute(0,2,1).cpu().numpy(); r = o.detach().numpy()
    ga = g.grad.permute(0,2,1).cpu().numpy(); gr = o.grad.numpy()
    d = np.abs(ga-gr); thr = 1e-4*np.abs(gr).max()
    print(f"stage{i}: fwd err {relerr(a,r):.2e}  grad err {relerr(ga,gr):.2e}  n(|d|>1e-4max)={(d>thr).sum()} of {d.size}")
gr = grads_by_ref_name(m)
for k in sorted(gr):
    e = relerr(gr[k].detach().cpu().numpy(), osd[k].grad.numpy())
    if e > 2e-4: print(f"{e:.2e}", k)
print("---- zero row (b=1) at stage1 output grad")
ga = gst[1].grad.permute(0,2,1).cpu().numpy()[1]; gr_ = stages[1].grad.numpy()[1]
print("per-channel sums match:", relerr(ga.sum(1), gr_.sum(1)))
print("gpu ch0 first 12:", np.round(ga[0,:12]*1e6,3))
print("cpu ch0 first 12:", np.round(gr_[0,:12]*1e6,3))
print("gpu ch0 pos 100..112:", np.round(ga[0,100:112]*1e6,3))
print("cpu ch0 pos 100..112:", np.round(gr_[0,100:112]*1e6,3))
# forward pre-pool uniqueness on GPU for the zero row
blk = m.all_stages[1][0]
xin = gst[0].detach()
y = H.conv_group1d(xin, blk.kernel_sizes, [c.weight for c in blk.convs], [c.bias for c in blk.convs])
print("conv bank interior distinct (ch0..3):", [len(torch.unique(y[1,300:700,c])) for c in (0,1,130,300)])
y2 = blk.norm(y, act="gelu"); print("ln distinct:", [len(torch.unique(y2[1,300:700,c])) for c in (0,1,130,300)])
y3 = blk.downsample(y2); print("1x1 distinct:", [len(torch.unique(y3[1,300:700,c])) for c in (0,1,100)])
print("stage0 out distinct for zero row:", [len(torch.unique(xin[1,:,c])) for c in (0,1,2)])
print("---- top mismatches at stage1 output grad")
ga = gst[1].grad.permute(0,2,1).cpu().numpy(); gr_ = stages[1].grad.numpy()
d = np.abs(ga-gr_); idx = np.argsort(d.reshape(-1))[::-1][:8]
for i in idx:
    b_, c_, l_ = np.unravel_index(i, d.shape); print(b_, c_, l_, ga[b_,c_,l_], gr_[b_,c_,l_])
print("max|ref| per sample:", [float(np.abs(gr_[i]).max()) for i in range(4)])
for bb in range(4):
    print("sample", bb, "relerr", relerr(ga[bb], gr_[bb]), "n>thr", int((np.abs(ga[bb]-gr_[bb]) > 1e-4*np.abs(gr_).max()).sum()))

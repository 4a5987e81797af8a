"""Functional CPU restatement of the hot path (TEST INFRASTRUCTURE ONLY — see oracle/__init__.py).

Every function takes a state_dict `sd` keyed like the reference module's own state_dict plus a
key prefix, and plain CPU fp32 tensors.  Dropout is the identity here (eval / p = 0): the
reference's training-mode dropout cannot be matched bit for bit (SURVEY.md §7 hard parts).
"""

from __future__ import annotations

import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------- helpers


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


# Tests only (the ReLU counterpart of spectranet_forward's `routing`): ReLU' is discontinuous at 0, so a unit whose
# pre-activation is within two implementations' rounding of zero may be open in one and closed in the other, and its
# whole upstream gradient row differs by one token's contribution.  RELU_GATES = {site: [bool masks in call order]}
# makes the ReLUs of that site use THOSE gates (y = x * gate); every call is recorded under "_seen"[site] as
# (own gate, |x| / max |x|) so that a test can show the differing gates were all near zero.  None = plain F.relu.
RELU_GATES = None


def _relu(x, site):
    if RELU_GATES is None:
        return F.relu(x)
    seen = RELU_GATES.setdefault("_seen", {}).setdefault(site, [])
    xd = x.detach()
    seen.append((xd > 0, xd.abs() / xd.abs().max().clamp_min(1e-30)))
    given = RELU_GATES.get(site)
    if given is None:
        return F.relu(x)
    return x * given[len(seen) - 1].reshape(x.shape).to(x.dtype)


def _ln(sd, p, x, eps=1e-5):
    w = sd[p + ".weight"]
    return F.layer_norm(x, (w.numel(),), w, sd[p + ".bias"], eps)


# ----------------------------------------------------------------------------- A2: ConvNeXt-Tiny
CONVNEXT_DEPTHS = (3, 3, 9, 3)
CONVNEXT_DIMS = (96, 192, 384, 768)


def _ln2d(sd, p, x, eps=1e-6):
    """LayerNorm over channels of an NCHW map (timm LayerNorm2d)."""
    return _ln(sd, p, x.permute(0, 2, 3, 1), eps).permute(0, 3, 1, 2)


def convnext_block(sd, p, x):
    """timm ConvNeXtBlock (conv_mlp=False): dw7x7 -> LN(C, 1e-6) -> fc1 -> GELU(erf) -> fc2
    -> * gamma -> + shortcut.  Call site of the backbone: astrominn.py:12-17."""
    C = x.shape[1]
    h = F.conv2d(x, sd[p + ".conv_dw.weight"], sd[p + ".conv_dw.bias"], padding=3, groups=C)
    h = h.permute(0, 2, 3, 1)
    h = _ln(sd, p + ".norm", h, 1e-6)
    h = _lin(sd, p + ".mlp.fc2", F.gelu(_lin(sd, p + ".mlp.fc1", h)))
    h = h.permute(0, 3, 1, 2) * sd[p + ".gamma"].reshape(1, -1, 1, 1)
    return x + h


def convnext_tiny_features(sd, p, x):
    """create_model('convnext_tiny', in_chans, num_classes=0): [B,Cin,H,W] -> [B,768]
    (stem 4x4/s4 + LN2d; stages with LN2d + 2x2/s2 downsample; avg-pool; head.norm; flatten)."""
    h = F.conv2d(x, sd[p + ".stem.0.weight"], sd[p + ".stem.0.bias"], stride=4)
    h = _ln2d(sd, p + ".stem.1", h)
    for i, depth in enumerate(CONVNEXT_DEPTHS):
        sp = f"{p}.stages.{i}"
        if i > 0:
            h = _ln2d(sd, sp + ".downsample.0", h)
            h = F.conv2d(h, sd[sp + ".downsample.1.weight"], sd[sp + ".downsample.1.bias"], stride=2)
        for j in range(depth):
            h = convnext_block(sd, f"{sp}.blocks.{j}", h)
    h = h.mean((-2, -1), keepdim=True)
    h = _ln2d(sd, p + ".head.norm", h)
    return h.flatten(1)


def convnext_tiny_shapes(p, in_chans=3):
    """Parameter names/shapes of timm convnext_tiny(num_classes=0) (naming per timm 1.0.x)."""
    s = {f"{p}.stem.0.weight": (96, in_chans, 4, 4), f"{p}.stem.0.bias": (96,),
         f"{p}.stem.1.weight": (96,), f"{p}.stem.1.bias": (96,)}
    prev = 96
    for i, (depth, dim) in enumerate(zip(CONVNEXT_DEPTHS, CONVNEXT_DIMS)):
        sp = f"{p}.stages.{i}"
        if i > 0:
            s[sp + ".downsample.0.weight"] = (prev,)
            s[sp + ".downsample.0.bias"] = (prev,)
            s[sp + ".downsample.1.weight"] = (dim, prev, 2, 2)
            s[sp + ".downsample.1.bias"] = (dim,)
        for j in range(depth):
            bp = f"{sp}.blocks.{j}"
            s[bp + ".gamma"] = (dim,)
            s[bp + ".conv_dw.weight"] = (dim, 1, 7, 7)
            s[bp + ".conv_dw.bias"] = (dim,)
            s[bp + ".norm.weight"] = (dim,)
            s[bp + ".norm.bias"] = (dim,)
            s[bp + ".mlp.fc1.weight"] = (4 * dim, dim)
            s[bp + ".mlp.fc1.bias"] = (4 * dim,)
            s[bp + ".mlp.fc2.weight"] = (dim, 4 * dim)
            s[bp + ".mlp.fc2.bias"] = (dim,)
        prev = dim
    s[p + ".head.norm.weight"] = (768,)
    s[p + ".head.norm.bias"] = (768,)
    return s


# ----------------------------------------------------------------------------- A1, A3, A4: AstroMiNN
def split_head_convnext(sd, p, x, features=None):
    """SplitHeadConvNeXt.forward (astrominn.py:8-41): main(f) * aux(f)."""
    f = convnext_tiny_features(sd, p + ".backbone", x) if features is None else features
    m = _ln(sd, p + ".head_main.1", F.gelu(f))
    m = _relu(_lin(sd, p + ".head_main.2", m), "image_head")
    m = _lin(sd, p + ".head_main.6", _lin(sd, p + ".head_main.5", m))
    a = torch.tanh(_lin(sd, p + ".head_aux.1", _ln(sd, p + ".head_aux.0", f)))
    return m * a


def residual_tower(sd, p, x):
    """ResidualTowerBlock.forward (astrominn.py:44-64)."""
    h = F.gelu(_lin(sd, p + ".start_path.0", x))
    gate = torch.sigmoid(_lin(sd, p + ".activation.2", _ln(sd, p + ".activation.0", h)))
    main = _lin(sd, p + ".main_path.2", _ln(sd, p + ".main_path.0", h))
    skip = _lin(sd, p + ".skip_path", x) if (p + ".skip_path.weight") in sd else x
    return main * gate + skip


ASTRO_COLS = {  # astrominn.py:249-261
    "nst1_tower": [0, 2], "nst2_tower": [1, 3], "spatial_tower": [2, 3, 4], "psf_tower": [5, 14],
    "mag_tower": [6, 9, 10, 13, 15, 17, 18], "coord_tower": [7, 8],
    "mega_tower": list(range(19)), "lc_tower": [6, 9, 10, 13, 15, 17, 18, 19, 20, 21, 22, 23],
}
ASTRO_CAT_ORDER = ["nst1_tower", "nst2_tower", "spatial_tower", "psf_tower", "mag_tower",
                   "coord_tower", "mega_tower", "image", "lc_tower"]  # astrominn.py:264-267


def astrominn_forward(sd, metadata, image, num_experts=4, use_probabilities=False,
                      image_features=None, return_aux=False):
    """AstroMiNN.forward (astrominn.py:220-300).  Dense restatement of the top-2 routing loop:
    experts not in a sample's top-2 contribute exactly zero, as in the masked reference loop."""
    feats = {}
    for name, cols in ASTRO_COLS.items():
        feats[name] = residual_tower(sd, name, metadata[:, cols])
    feats["image"] = split_head_convnext(sd, "image_tower", image, image_features)
    allf = torch.cat([feats[k] for k in ASTRO_CAT_ORDER], 1)
    r = torch.tanh(_lin(sd, "fusion_router.0", allf))
    scores = torch.sigmoid(_lin(sd, "fusion_router.3", r))
    tw, ti = torch.topk(scores, k=2, dim=-1)
    out = torch.zeros(metadata.shape[0], 5)
    for e in range(num_experts):
        sel = ti == e  # [B,2]
        w = (tw * sel).sum(-1, keepdim=True)  # weight if selected else 0
        out = out + w * residual_tower(sd, f"fusion_experts.{e}", allf)
    if use_probabilities:
        out = F.softmax(out, -1)
    if return_aux:
        return out, scores, ti, allf
    return out


# ----------------------------------------------------------------------------- B1, B2, B3
def time2vec(sd, p, t):
    """Time2Vec.forward (Time2Vec.py:62-72)."""
    v0 = sd[p + ".w0"] * t + sd[p + ".b0"]
    vp = torch.sin(t.unsqueeze(-1) * sd[p + ".w"] + sd[p + ".b"])
    return torch.cat([v0.unsqueeze(-1), vp], -1)


def encoder_layer(sd, p, x, key_pad, n_heads):
    """nn.TransformerEncoderLayer defaults (post-LN, ReLU, eps 1e-5, batch_first) with
    src_key_padding_mask, as built at HyraxBaselineCLS.py:24-31."""
    B, T, D = x.shape
    dh = D // n_heads
    qkv = F.linear(x, sd[p + ".self_attn.in_proj_weight"], sd[p + ".self_attn.in_proj_bias"])
    q, k, v = (t.reshape(B, T, n_heads, dh).transpose(1, 2) for t in qkv.split(D, -1))
    s = (q / math.sqrt(dh)) @ k.transpose(-1, -2)
    s = s.masked_fill(key_pad[:, None, None, :], float("-inf"))
    a = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, T, D)
    x = _ln(sd, p + ".norm1", x + _lin(sd, p + ".self_attn.out_proj", a))
    ff = _lin(sd, p + ".linear2", _relu(_lin(sd, p + ".linear1", x), "encoder_ff"))
    return _ln(sd, p + ".norm2", x + ff)


def baselinecls_forward(sd, data, pad, n_heads=8, n_layers=4, classification=True,
                        use_probabilities=False, out_key="fc"):
    """HyraxBaselineCLS.forward (HyraxBaselineCLS.py:49-86).  out_key = "head": the legacy non-Hyrax `BaselineCLS`
    (Time2Vec.py:80-124), which is the same network classified through `head` instead of `fc`."""
    B = data.shape[0]
    h = _lin(sd, "in_proj", data) + time2vec(sd, "time2vec", data[..., 0])
    h = torch.cat([sd["cls_tok"].expand(B, -1, -1), h], 1)
    pad_ext = F.pad(pad, (1, 0), value=False)
    for i in range(n_layers):
        h = encoder_layer(sd, f"encoder.layers.{i}", h, pad_ext, n_heads)
    out = _ln(sd, "norm", h[:, 0])
    if classification:
        out = _lin(sd, out_key, out)
    if use_probabilities:
        out = F.softmax(out, 1)
    return out


def mpt_heads(sd, z):
    """Legacy MPTModel(base_enc).forward (Time2Vec.py:128-142): the three heads over an encoder output."""
    return _lin(sd, "head_flux", z), _lin(sd, "head_band", z), _lin(sd, "head_dt", z)


def pretrained_handoff(own_sd, checkpoint_sd):
    """`self.load_state_dict(torch.load(path), strict=False)` (HyraxBaselineCLS.py:43-47) on state_dicts: every key
    of the model that the checkpoint also holds takes the checkpoint's value, the others keep the model's own; keys
    only the checkpoint holds (an MPTModel's head_flux / head_band / head_dt) are ignored.
    Returns (merged, taken keys, kept keys, ignored keys)."""
    taken = sorted(k for k in own_sd if k in checkpoint_sd)
    kept = sorted(k for k in own_sd if k not in checkpoint_sd)
    ignored = sorted(k for k in checkpoint_sd if k not in own_sd)
    merged = {k: (checkpoint_sd[k] if k in checkpoint_sd else own_sd[k]) for k in own_sd}
    return merged, taken, kept, ignored


def mpt_apply_mask(data, masked):
    """The in-place effect of MPTModel._mask_batch for a GIVEN selection (HyraxBaselineCLS.py:316-318):
    channels 2..6 (logf, logfe, band one-hot) of the selected tokens are zeroed."""
    data = data.clone()
    data[masked] = torch.cat([data[masked][:, :2], torch.zeros_like(data[masked][:, 2:])], 1)
    return data


def mpt_loss(sd, data, pad, masked, n_heads=8, n_layers=4, lambdas=(1.0, 1.0, 1.0)):
    """MPTModel.train_step up to the loss (HyraxBaselineCLS.py:226-278) for a given mask and dropout 0.
    `data` is the tensor AFTER masking — the reference reads its regression / band targets from the
    tensor it has just zeroed (:262-266), so at the masked tokens true_f = 0 and true_b = argmax(0,0,0)
    = 0; that behaviour is reproduced, not corrected.  Returns (loss, loss_f, loss_b, loss_dt)."""
    B = data.shape[0]
    h = _lin(sd, "in_proj", data) + time2vec(sd, "time2vec", data[..., 0])
    h = torch.cat([sd["cls_tok"].expand(B, -1, -1), h], 1)
    pad_ext = torch.cat([pad.new_zeros((B, 1)), pad], 1)
    for i in range(n_layers):
        h = encoder_layer(sd, f"encoder.layers.{i}", h, pad_ext, n_heads)
    z = h[:, 1:, :]
    f_hat, b_hat, dt_hat = _lin(sd, "head_flux", z), _lin(sd, "head_band", z), _lin(sd, "head_dt", z)
    mf = masked.reshape(-1)
    loss_f = F.mse_loss(f_hat.reshape(-1)[mf], data[..., 2].reshape(-1)[mf])
    true_b = data[..., 4:7].argmax(-1).reshape(-1)
    loss_b = F.cross_entropy(b_hat.reshape(-1, 3)[mf], true_b[mf])
    dt_gt = torch.roll(data[..., 1], -1, dims=1).clone()
    dt_gt[:, -1] = 0.0
    loss_dt = F.mse_loss(dt_hat[..., 0].reshape(-1)[mf], dt_gt.reshape(-1)[mf])
    lf, lb, ldt = lambdas
    return lf * loss_f * lb * loss_b * ldt * loss_dt, loss_f, loss_b, loss_dt


def focal_loss(logits, target, gamma=2.0, alpha=None, eps=0.0):
    """FocalLoss.forward, reduction='mean' (HyraxBaselineCLS.py:169-191)."""
    C = logits.shape[1]
    logp = F.log_softmax(logits, 1)
    p = logp.exp()
    if eps > 0:
        y = torch.full_like(logp, eps / (C - 1))
        y.scatter_(1, target.unsqueeze(1), 1.0 - eps)
    else:
        y = F.one_hot(target, C).float()
    fw = (1.0 - p).pow(gamma)
    if alpha is not None:
        fw = fw * alpha.view(1, C)
    return -(y * fw * logp).sum(1).mean()


# ----------------------------------------------------------------------------- C1, C2: SpectraNet
def spectranet_block(sd, p, x, ksizes, do_pool, use_ln=True, training=False, routing=None):
    """SpectraNetBlock.forward (spectranet.py:28-41); x is [B,C,L].  use_ln=False: nn.BatchNorm1d over
    the channels (spectranet.py:21,33) — batch statistics and in-place running-statistics update when
    `training`, running statistics otherwise."""
    y = torch.cat([F.conv1d(x, sd[f"{p}.convs.{i}.weight"], sd[f"{p}.convs.{i}.bias"], padding=k // 2)
                   for i, k in enumerate(ksizes)], 1)
    if use_ln:
        y = F.gelu(_ln(sd, p + ".norm", y.permute(0, 2, 1)).permute(0, 2, 1))
    else:
        y = F.gelu(F.batch_norm(y, sd[p + ".norm.running_mean"], sd[p + ".norm.running_var"],
                                sd[p + ".norm.weight"], sd[p + ".norm.bias"], training, 0.1, 1e-5))
    if do_pool:
        y = F.conv1d(y, sd[p + ".downsample.weight"], sd[p + ".downsample.bias"])
        y = _pool4(y, routing)
    return y


def _pool4(y, routing):
    """MaxPool1d(4) (spectranet.py:40).  `routing` (tests only, see spectranet_forward): records this pool's window
    margins and, when it carries indices, selects THOSE positions instead of the arg-max."""
    if routing is None:
        return F.max_pool1d(y, 4)
    B, C, L = y.shape
    w = y.reshape(B, C, L // 4, 4)
    top = w.detach().topk(2, dim=-1)
    routing.setdefault("margins", []).append((top.values[..., 0] - top.values[..., 1]) / w.detach().abs().max())
    # the position torch's own max_pool1d routes to (first maximum on exact ties - e.g. the constant rows behind an
    # all-zero spectrum), as window positions 0..3
    own = F.max_pool1d(y.detach(), 4, return_indices=True)[1] - 4 * torch.arange(L // 4).view(1, 1, -1)
    routing.setdefault("argmax", []).append(own)
    given = routing.get("pool")
    if given is None:
        return w.max(-1).values
    idx = given[len(routing["argmax"]) - 1].long()
    return w.gather(-1, idx.unsqueeze(-1)).squeeze(-1)


def _global_max(x, routing):
    """adaptive_max_pool1d(., 1) (spectranet.py:161) with the same routing hook as _pool4."""
    if routing is None:
        return F.adaptive_max_pool1d(x, 1).squeeze(-1)
    if x.shape[-1] < 2:
        routing["global_margin"] = torch.full(x.shape[:-1], float("inf"))
        routing["global_argmax"] = torch.zeros(x.shape[:-1], dtype=torch.long)
    else:
        top = x.detach().topk(2, dim=-1)
        routing["global_margin"] = (top.values[..., 0] - top.values[..., 1]) / x.detach().abs().max()
        routing["global_argmax"] = F.adaptive_max_pool1d(x.detach(), 1, return_indices=True)[1].squeeze(-1)
    if routing.get("global") is None:
        return x.max(-1).values
    return x.gather(-1, routing["global"].long().unsqueeze(-1)).squeeze(-1)


def spectranet_forward(sd, x, kernel_sizes_per_stage, depths=None, head="classifier",
                       return_stages=False, use_ln_stages=None, training=False, routing=None):
    """SpectraNet.forward (spectranet.py:157-170); x is [B,1,L].
    `routing` (a dict, tests only): the max-pools record their windows' top-2 margins (relative to the tensor's
    max-abs) and arg-max positions into it; if it holds "pool" (list of [B, C, L/4] window positions 0..3, one per
    pooled stage) / "global" ([B, C] positions), the pools select THOSE positions — the gradient of the reference's
    network under another implementation's routing of near-tie windows, which is how a GPU gradient is compared
    tightly although max-pooling makes the gradient discontinuous."""
    n = len(kernel_sizes_per_stage)
    depths = depths or [1] * n
    use_ln_stages = use_ln_stages or [True] * n
    stages = []
    for i in range(n):
        for j in range(depths[i]):
            x = spectranet_block(sd, f"all_stages.{i}.{j}", x, kernel_sizes_per_stage[i],
                                 do_pool=(i < n - 1 and j == depths[i] - 1), use_ln=use_ln_stages[i],
                                 training=training, routing=routing)
        stages.append(x)
    z = _global_max(x, routing)
    z = F.gelu(_ln(sd, head + ".1", _lin(sd, head + ".0", z)))
    out = _lin(sd, head + ".4", z)
    if head == "regressor":
        out = out.squeeze(1)
    return (out, stages) if return_stages else out


def spectranet_train_steps(sd, flux, labels, kernel_sizes_per_stage, n_steps=1, lr=0.01, momentum=0.9,
                           redshifts=None):
    """SpectraNet.train_step (spectranet.py:172-184) under the optimizer / criterion Hyrax injects
    (torch.optim.SGD(lr, momentum), CrossEntropyLoss; spectranet_testing.ipynb cell 14), restated on a
    state_dict: zero_grad -> forward -> CE -> backward -> buf = momentum*buf + g ; p -= lr*buf
    (torch.optim.SGD: the first step initialises buf = g).  Labels may be any integer dtype (the
    reference's to_tensor emits int16, spectranet.py:204).  `redshifts` given = the `redshift = True` model
    (spectranet.py:139-147,167-168,178-179): regressor head, squeezed, MSELoss against the redshifts.
    Returns (losses, updated state_dict)."""
    sd = {k: v.detach().clone().requires_grad_() for k, v in sd.items()}
    buf, losses = {}, []
    for _ in range(n_steps):
        if redshifts is not None:
            loss = F.mse_loss(spectranet_forward(sd, flux, kernel_sizes_per_stage, head="regressor"), redshifts)
        else:
            logits = spectranet_forward(sd, flux, kernel_sizes_per_stage)
            loss = F.cross_entropy(logits, labels.long())
        grads = torch.autograd.grad(loss, list(sd.values()))
        losses.append(float(loss.detach()))
        with torch.no_grad():
            for (k, p), g in zip(sd.items(), grads):
                buf[k] = g.clone() if k not in buf else buf[k].mul_(momentum).add_(g)
                p.sub_(lr * buf[k])
    return losses, {k: v.detach() for k, v in sd.items()}


# ----------------------------------------------------------------------------- F1: fusion
def fusion_head(sd, p_emb, s_emb, im_emb, fusion="avg"):
    """AppleCider.get_embeddings/forward after the encoders
    (_archive/notebooks/brew_cider.py:834-862; older variant _archive/AppleCider/core/model.py:40-67):
    project each branch, L2-normalise, avg | concat (photometry, image+meta, spectra), fc."""
    p = _lin(sd, "photometry_proj", p_emb)
    s = _lin(sd, "spectra_proj", s_emb)
    im = _lin(sd, "img_metadata_proj", im_emb)
    p = p / p.norm(dim=-1, keepdim=True)
    im = im / im.norm(dim=-1, keepdim=True)
    s = s / s.norm(dim=-1, keepdim=True)
    emb = torch.cat((p, im, s), 1) if fusion == "concat" else (p + im + s) / 3
    return _lin(sd, "fc", emb)


def _sub(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


def applecider_forward(sd, photometry, photo_mask, metadata, images, spectra, cfg):
    """4-modality forward with the build's documented definition:
    photometry_encoder = HyraxBaselineCLS embedding (mode != 'photo'), spectra_encoder = src
    SpectraNet logits, img_metadata_encoder = AstroMiNN logits."""
    p_emb = baselinecls_forward(_sub(sd, "photometry_encoder."), photometry, photo_mask,
                                cfg["p_n_heads"], cfg["p_n_layers"], classification=False)
    s_emb = spectranet_forward(_sub(sd, "spectra_encoder."), spectra, cfg["kernel_sizes_per_stage"],
                               routing=cfg.get("routing"))
    im_emb = astrominn_forward(_sub(sd, "img_metadata_encoder."), metadata, images)
    return fusion_head(sd, p_emb, s_emb, im_emb, cfg["fusion"])

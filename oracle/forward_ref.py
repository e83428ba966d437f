"""ORACLE (test infrastructure — never imported by the product path).

CPU restatement, in plain PyTorch fp32 functional ops, of the reference CellViT forward pass:

  * ViT-S encoder  — /root/reference/models/encoders/VIT/vits_histo.py:106-247, 377-415 and the
    skip-extracting wrapper .../cell_segmentation/utils.py:149-174
  * SAM ViTDet encoder — /root/reference/models/encoders/VIT/SAM/image_encoder.py:128-392 and
    .../cell_segmentation/utils.py:218-233, SAM/utils.py:20-50
  * U-Net decoder — .../cell_segmentation/cellvit.py:116-131, 212-330 and utils.py:11-86
  * CellViT.forward / CellViTSAM.forward — cellvit.py:153-210, 586-644

It works from a ``state_dict`` with the reference's key names and a :class:`CellViTConfig`;
it does not import the reference.  Parity pin: ``tools/make_golden_forward.py`` ran the imported
reference in the development container on the seeded weights/inputs and committed its outputs
under ``tests/golden/``; ``tests/test_oracle_forward.py`` checks this restatement against them.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from cellvit_amd.spec import ARCH_SAM, ARCH_VIT, CellViTConfig

LN_EPS = 1e-6   # cellvit.py:99, 559 ; SAM/utils.py:39
BN_EPS = 1e-5   # torch default, utils.py:37, 80


def _ln(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[f"{p}.weight"], sd[f"{p}.bias"], LN_EPS)


def _lin(x, sd, p):
    return F.linear(x, sd[f"{p}.weight"], sd.get(f"{p}.bias"))


# ------------------------------------------------------------------------------------------
# encoders
# ------------------------------------------------------------------------------------------
def vit_pos_embed(sd, cfg: CellViTConfig, w: int, h: int) -> torch.Tensor:
    """interpolate_pos_encoding — vits_histo.py:377-402 (scale_factor semantics, bicubic)."""
    pos = sd["encoder.pos_embed"]
    N = pos.shape[1] - 1
    npatch = (w // cfg.patch_size) * (h // cfg.patch_size)
    if npatch == N and w == h:
        return pos
    cls_pos, patch_pos = pos[:, 0], pos[:, 1:]
    dim = pos.shape[-1]
    w0 = w // cfg.patch_size + 0.1
    h0 = h // cfg.patch_size + 0.1
    g = int(math.sqrt(N))
    patch_pos = F.interpolate(patch_pos.reshape(1, g, g, dim).permute(0, 3, 1, 2),
                              scale_factor=(w0 / math.sqrt(N), h0 / math.sqrt(N)), mode="bicubic")
    assert int(w0) == patch_pos.shape[-2] and int(h0) == patch_pos.shape[-1]
    patch_pos = patch_pos.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((cls_pos.unsqueeze(0), patch_pos), dim=1)


def encoder_vit(x, sd, cfg: CellViTConfig, taps: Optional[dict] = None):
    B, _, H, W = x.shape
    D, nh = cfg.embed_dim, cfg.num_heads
    t = F.conv2d(x, sd["encoder.patch_embed.proj.weight"], sd["encoder.patch_embed.proj.bias"],
                 stride=cfg.patch_size).flatten(2).transpose(1, 2)
    t = torch.cat((sd["encoder.cls_token"].expand(B, -1, -1), t), dim=1)
    t = t + vit_pos_embed(sd, cfg, H, W)   # prepare_tokens passes (w, h) = x.shape[2:]
    if taps is not None:
        taps["tokens0"] = t
    scale = cfg.head_dim ** -0.5
    z: List[torch.Tensor] = []
    for i in range(cfg.depth):
        p = f"encoder.blocks.{i}"
        y = _ln(t, sd, f"{p}.norm1")
        N = y.shape[1]
        qkv = _lin(y, sd, f"{p}.attn.qkv").reshape(B, N, 3, nh, D // nh).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = ((q @ k.transpose(-2, -1)) * scale).softmax(dim=-1)      # vits_histo.py:181-182
        y = (attn @ v).transpose(1, 2).reshape(B, N, D)
        t = t + _lin(y, sd, f"{p}.attn.proj")
        y = _ln(t, sd, f"{p}.norm2")
        y = _lin(F.gelu(_lin(y, sd, f"{p}.mlp.fc1")), sd, f"{p}.mlp.fc2")
        t = t + y
        if taps is not None:
            taps[f"block{i}"] = t
        if i + 1 in cfg.extract_layers:
            z.append(t)
    cls = _ln(t, sd, "encoder.norm")[:, 0]
    logits = _lin(cls, sd, "encoder.head") if cfg.num_tissue_classes > 0 else cls
    g = (H // cfg.patch_size, W // cfg.patch_size)
    zs = [zi[:, 1:, :].transpose(-1, -2).reshape(B, D, *g) for zi in z]    # cellvit.py:185-189
    return logits, zs


def _get_rel_pos(q_size: int, k_size: int, rel_pos: torch.Tensor) -> torch.Tensor:
    """image_encoder.py:321-351."""
    max_rel = int(2 * max(q_size, k_size) - 1)
    if rel_pos.shape[0] != max_rel:
        r = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1),
                          size=max_rel, mode="linear")
        r = r.reshape(-1, max_rel).permute(1, 0)
    else:
        r = rel_pos
    qc = torch.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    kc = torch.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    rel = (qc - kc) + (k_size - 1) * max(q_size / k_size, 1.0)
    return r[rel.long()]


def _sam_attention(x, sd, p, nh):
    """image_encoder.py:235-260 + add_decomposed_rel_pos :354-392.  x: [B', H, W, D]."""
    Bp, H, W, D = x.shape
    hd = D // nh
    qkv = _lin(x, sd, f"{p}.qkv").reshape(Bp, H * W, 3, nh, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, Bp * nh, H * W, -1).unbind(0)
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    Rh = _get_rel_pos(H, H, sd[f"{p}.rel_pos_h"])
    Rw = _get_rel_pos(W, W, sd[f"{p}.rel_pos_w"])
    r_q = q.reshape(Bp * nh, H, W, hd)
    rel_h = torch.einsum("bhwc,hkc->bhwk", r_q, Rh)
    rel_w = torch.einsum("bhwc,wkc->bhwk", r_q, Rw)
    attn = (attn.view(Bp * nh, H, W, H, W) + rel_h[..., :, None] + rel_w[..., None, :]
            ).view(Bp * nh, H * W, H * W)
    attn = attn.softmax(dim=-1)
    y = (attn @ v).view(Bp, nh, H, W, -1).permute(0, 2, 3, 1, 4).reshape(Bp, H, W, -1)
    return _lin(y, sd, f"{p}.proj")


def encoder_sam(x, sd, cfg: CellViTConfig, taps: Optional[dict] = None):
    B = x.shape[0]
    D, nh, ws = cfg.embed_dim, cfg.num_heads, cfg.window_size
    t = F.conv2d(x, sd["encoder.patch_embed.proj.weight"], sd["encoder.patch_embed.proj.bias"],
                 stride=cfg.patch_size).permute(0, 2, 3, 1)
    g = t.shape[1]
    t = t + sd["encoder.pos_embed"][:, :g, :g, :]              # utils.py:222-224 (square slice)
    if taps is not None:
        taps["tokens0"] = t
    z = []
    for i in range(cfg.depth):
        p = f"encoder.blocks.{i}"
        short = t
        y = _ln(t, sd, f"{p}.norm1")
        if i not in cfg.global_attn_indexes:                    # window blocks, :181-188
            H, W = y.shape[1], y.shape[2]
            ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
            y = F.pad(y, (0, 0, 0, pw, 0, ph))
            Hp, Wp = H + ph, W + pw
            y = y.view(B, Hp // ws, ws, Wp // ws, ws, D).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, D)
            y = _sam_attention(y, sd, f"{p}.attn", nh)
            y = y.view(B, Hp // ws, Wp // ws, ws, ws, D).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, D)
            y = y[:, :H, :W, :]
        else:
            y = _sam_attention(y, sd, f"{p}.attn", nh)
        t = short + y
        y = _ln(t, sd, f"{p}.norm2")
        t = t + _lin(F.gelu(_lin(y, sd, f"{p}.mlp.lin1")), sd, f"{p}.mlp.lin2")
        if taps is not None:
            taps[f"block{i}"] = t
        if i + 1 in cfg.extract_layers:
            z.append(t)

    def ln2d(v, p):                                              # SAM/utils.py:38-50
        u = v.mean(1, keepdim=True)
        s = (v - u).pow(2).mean(1, keepdim=True)
        v = (v - u) / torch.sqrt(s + LN_EPS)
        return sd[f"{p}.weight"][:, None, None] * v + sd[f"{p}.bias"][:, None, None]

    n = F.conv2d(t.permute(0, 3, 1, 2), sd["encoder.neck.0.weight"])
    n = ln2d(n, "encoder.neck.1")
    n = F.conv2d(n, sd["encoder.neck.2.weight"], padding=1)
    n = ln2d(n, "encoder.neck.3")
    pooled = n.flatten(2).mean(-1)                              # utils.py:231-233
    logits = _lin(pooled, sd, "classifier_head") if cfg.num_tissue_classes > 0 else pooled
    zs = [zi.permute(0, 3, 1, 2) for zi in z]                   # cellvit.py:618-621
    return logits, zs


# ------------------------------------------------------------------------------------------
# decoder
# ------------------------------------------------------------------------------------------
def _conv_block(x, sd, p):
    """Conv2DBlock in eval mode — utils.py:29-40."""
    x = F.conv2d(x, sd[f"{p}.block.0.weight"], sd[f"{p}.block.0.bias"], padding=1)
    x = F.batch_norm(x, sd[f"{p}.block.1.running_mean"], sd[f"{p}.block.1.running_var"],
                     sd[f"{p}.block.1.weight"], sd[f"{p}.block.1.bias"], False, 0.0, BN_EPS)
    return F.relu(x)


def _deconv_block(x, sd, p):
    """Deconv2DBlock in eval mode — utils.py:64-83."""
    x = F.conv_transpose2d(x, sd[f"{p}.block.0.weight"], sd[f"{p}.block.0.bias"], stride=2)
    x = F.conv2d(x, sd[f"{p}.block.1.weight"], sd[f"{p}.block.1.bias"], padding=1)
    x = F.batch_norm(x, sd[f"{p}.block.2.running_mean"], sd[f"{p}.block.2.running_var"],
                     sd[f"{p}.block.2.weight"], sd[f"{p}.block.2.bias"], False, 0.0, BN_EPS)
    return F.relu(x)


def _convT(x, sd, p):
    return F.conv_transpose2d(x, sd[f"{p}.weight"], sd[f"{p}.bias"], stride=2)


def shared_skips(z0, z1, z2, z3, sd):
    """decoder0..3 (cellvit.py:116-131), evaluated ONCE (they are branch-independent in eval)."""
    b0 = _conv_block(_conv_block(z0, sd, "decoder0.0"), sd, "decoder0.1")
    b1 = _deconv_block(_deconv_block(_deconv_block(z1, sd, "decoder1.0"), sd, "decoder1.1"), sd, "decoder1.2")
    b2 = _deconv_block(_deconv_block(z2, sd, "decoder2.0"), sd, "decoder2.1")
    b3 = _deconv_block(z3, sd, "decoder3.0")
    return b0, b1, b2, b3


def branch(z4, skips, sd, p, taps: Optional[dict] = None):
    """_forward_upsample — cellvit.py:212-244 (concat order [skip, upsampled])."""
    b0, b1, b2, b3 = skips
    x = _convT(z4, sd, f"{p}.bottleneck_upsampler")
    x = torch.cat([b3, x], dim=1)
    for j in range(3):
        x = _conv_block(x, sd, f"{p}.decoder3_upsampler.{j}")
    x = _convT(x, sd, f"{p}.decoder3_upsampler.3")
    if taps is not None:
        taps[f"{p}.up3"] = x
    x = torch.cat([b2, x], dim=1)
    for j in range(2):
        x = _conv_block(x, sd, f"{p}.decoder2_upsampler.{j}")
    x = _convT(x, sd, f"{p}.decoder2_upsampler.2")
    x = torch.cat([b1, x], dim=1)
    for j in range(2):
        x = _conv_block(x, sd, f"{p}.decoder1_upsampler.{j}")
    x = _convT(x, sd, f"{p}.decoder1_upsampler.2")
    x = torch.cat([b0, x], dim=1)
    for j in range(2):
        x = _conv_block(x, sd, f"{p}.decoder0_header.{j}")
    return F.conv2d(x, sd[f"{p}.decoder0_header.2.weight"], sd[f"{p}.decoder0_header.2.bias"])


@torch.no_grad()
def forward(x: torch.Tensor, sd: Dict[str, torch.Tensor], cfg: CellViTConfig,
            retrieve_tokens: bool = False, taps: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    """CellViT.forward (cellvit.py:153-210) / CellViTSAM.forward (:586-644), fp32 on CPU."""
    assert x.shape[-2] % cfg.patch_size == 0 and x.shape[-1] % cfg.patch_size == 0, \
        "Img must have a shape of that is divisible by patch_size (token_size)"
    x = x.float()
    if cfg.arch == ARCH_VIT:
        logits, (z1, z2, z3, z4) = encoder_vit(x, sd, cfg, taps)
    elif cfg.arch == ARCH_SAM:
        logits, (z1, z2, z3, z4) = encoder_sam(x, sd, cfg, taps)
    else:
        raise ValueError(cfg.arch)
    out = {"tissue_types": logits}
    skips = shared_skips(x, z1, z2, z3, sd)
    if taps is not None:
        for n, s in zip(("skip0", "skip1", "skip2", "skip3"), skips):
            taps[n] = s
    nb = branch(z4, skips, sd, "nuclei_binary_map_decoder", taps)
    if cfg.regression_loss:
        out["nuclei_binary_map"], out["regression_map"] = nb[:, :2], nb[:, 2:]
    else:
        out["nuclei_binary_map"] = nb
    out["hv_map"] = branch(z4, skips, sd, "hv_map_decoder", taps)
    out["nuclei_type_map"] = branch(z4, skips, sd, "nuclei_type_maps_decoder", taps)
    if retrieve_tokens:
        out["tokens"] = z4
    return out

"""Architecture description of the CellViT hot path and the reference checkpoint layout.

The reference builds its networks as nested ``nn.Module`` trees
(/root/reference/models/segmentation/cell_segmentation/cellvit.py:57-151, 444-479, 514-572 and
 .../utils.py:11-86); what a drop-in replacement has to honour is the *state_dict contract*:
the key names, shapes and order of the tensors in ``ckpt["model_state_dict"]``
(/root/reference/base_ml/base_trainer.py:229-245).  This module derives that contract from a
small config record, table-driven, so that the C-ABI loader, the seeded weight generator and the
CPU oracle all agree on one list.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple

ARCH_VIT = 0  # DINO/timm style ViT with cls token  (vits_histo.py:283-415)
ARCH_SAM = 1  # SAM ViTDet encoder, window + global attention (SAM/image_encoder.py:24-125)


@dataclass
class CellViTConfig:
    arch: int
    embed_dim: int
    depth: int
    num_heads: int
    extract_layers: Tuple[int, int, int, int]
    num_nuclei_classes: int = 6
    num_tissue_classes: int = 19
    mlp_ratio: int = 4
    regression_loss: bool = False
    patch_size: int = 16
    # ViT: pre-training grid of the learned 1-D pos-embed (224/16 = 14, utils.py:93)
    # SAM: pre-training grid of the 2-D pos-embed (1024/16 = 64, utils.py:181)
    pos_grid: int = 14
    # SAM only
    window_size: int = 0
    global_attn_indexes: Tuple[int, ...] = ()
    neck_chans: int = 256
    name: str = "CellViT"

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads

    @property
    def skip_dims(self) -> Tuple[int, int, int]:
        """(skip_dim_11, skip_dim_12, bottleneck_dim) — cellvit.py:106-113."""
        if self.embed_dim < 512:
            return 256, 128, 312
        return 512, 256, 512

    @property
    def branch_out(self) -> Tuple[int, int, int]:
        """Output channels of the (binary, hv, type) branches — cellvit.py:133-151."""
        off = 2 if self.regression_loss else 0
        return 2 + off, 2, self.num_nuclei_classes


def cellvit256_config(num_nuclei_classes=6, num_tissue_classes=19, regression_loss=False):
    """cellvit.py:454-463."""
    return CellViTConfig(arch=ARCH_VIT, embed_dim=384, depth=12, num_heads=6,
                         extract_layers=(3, 6, 9, 12), num_nuclei_classes=num_nuclei_classes,
                         num_tissue_classes=num_tissue_classes, regression_loss=regression_loss,
                         pos_grid=14, name="CellViT256")


def cellvit_generic_config(num_nuclei_classes, num_tissue_classes, embed_dim, depth, num_heads, extract_layers,
                           mlp_ratio=4, regression_loss=False):
    """The generic class ``CellViT(...)`` (cellvit.py:57-75): a ViT with cls token, pos-embed grid 14 (utils.py:93)."""
    return CellViTConfig(arch=ARCH_VIT, embed_dim=embed_dim, depth=depth, num_heads=num_heads,
                         extract_layers=tuple(extract_layers), num_nuclei_classes=num_nuclei_classes,
                         num_tissue_classes=num_tissue_classes, mlp_ratio=int(mlp_ratio),
                         regression_loss=regression_loss, pos_grid=14, name="CellViT")


_SAM = {  # cellvit.py:646-665
    "SAM-B": dict(embed_dim=768, depth=12, num_heads=12, global_attn_indexes=(2, 5, 8, 11),
                  extract_layers=(3, 6, 9, 12)),
    "SAM-L": dict(embed_dim=1024, depth=24, num_heads=16, global_attn_indexes=(5, 11, 17, 23),
                  extract_layers=(6, 12, 18, 24)),
    "SAM-H": dict(embed_dim=1280, depth=32, num_heads=16, global_attn_indexes=(7, 15, 23, 31),
                  extract_layers=(8, 16, 24, 32)),
}


def cellvit_sam_config(vit_structure="SAM-H", num_nuclei_classes=6, num_tissue_classes=19,
                       regression_loss=False):
    key = vit_structure.upper()
    if key not in _SAM:
        raise NotImplementedError("Unknown ViT-SAM backbone structure")  # cellvit.py:530
    return CellViTConfig(arch=ARCH_SAM, num_nuclei_classes=num_nuclei_classes,
                         num_tissue_classes=num_tissue_classes, regression_loss=regression_loss,
                         pos_grid=64, window_size=14, neck_chans=256, name="CellViTSAM",
                         **_SAM[key])


# ------------------------------------------------------------------------------------------
# state_dict contract
# ------------------------------------------------------------------------------------------
ParamSpec = Tuple[str, Tuple[int, ...], str]  # (key, shape, kind)
# kind ∈ {"w" dense weight, "b" bias, "ln_w", "ln_b", "bn_w", "bn_b", "bn_mean", "bn_var",
#         "bn_count", "emb" (pos_embed / cls_token / rel_pos)}


def _conv_block(prefix: str, cin: int, cout: int, k: int = 3) -> List[ParamSpec]:
    """Conv2DBlock = Conv2d(bias) + BatchNorm2d + ReLU (+Dropout) — utils.py:29-40."""
    return [
        (f"{prefix}.block.0.weight", (cout, cin, k, k), "w"),
        (f"{prefix}.block.0.bias", (cout,), "b"),
        (f"{prefix}.block.1.weight", (cout,), "bn_w"),
        (f"{prefix}.block.1.bias", (cout,), "bn_b"),
        (f"{prefix}.block.1.running_mean", (cout,), "bn_mean"),
        (f"{prefix}.block.1.running_var", (cout,), "bn_var"),
        (f"{prefix}.block.1.num_batches_tracked", (), "bn_count"),
    ]


def _deconv_block(prefix: str, cin: int, cout: int) -> List[ParamSpec]:
    """Deconv2DBlock = ConvTranspose2d(k2,s2,bias) + Conv2d + BN + ReLU — utils.py:64-83."""
    return [
        (f"{prefix}.block.0.weight", (cin, cout, 2, 2), "wt"),
        (f"{prefix}.block.0.bias", (cout,), "b"),
        (f"{prefix}.block.1.weight", (cout, cout, 3, 3), "w"),
        (f"{prefix}.block.1.bias", (cout,), "b"),
        (f"{prefix}.block.2.weight", (cout,), "bn_w"),
        (f"{prefix}.block.2.bias", (cout,), "bn_b"),
        (f"{prefix}.block.2.running_mean", (cout,), "bn_mean"),
        (f"{prefix}.block.2.running_var", (cout,), "bn_var"),
        (f"{prefix}.block.2.num_batches_tracked", (), "bn_count"),
    ]


def _convT(prefix: str, cin: int, cout: int) -> List[ParamSpec]:
    return [(f"{prefix}.weight", (cin, cout, 2, 2), "wt"), (f"{prefix}.bias", (cout,), "b")]


def _linear(prefix: str, cin: int, cout: int, bias: bool = True) -> List[ParamSpec]:
    out = [(f"{prefix}.weight", (cout, cin), "w")]
    if bias:
        out.append((f"{prefix}.bias", (cout,), "b"))
    return out


def _ln(prefix: str, c: int) -> List[ParamSpec]:
    return [(f"{prefix}.weight", (c,), "ln_w"), (f"{prefix}.bias", (c,), "ln_b")]


def _encoder_vit(cfg: CellViTConfig) -> List[ParamSpec]:
    D, hid = cfg.embed_dim, cfg.embed_dim * cfg.mlp_ratio
    p: List[ParamSpec] = [
        ("encoder.cls_token", (1, 1, D), "emb"),
        ("encoder.pos_embed", (1, cfg.pos_grid * cfg.pos_grid + 1, D), "emb"),
        ("encoder.patch_embed.proj.weight", (D, 3, cfg.patch_size, cfg.patch_size), "w"),
        ("encoder.patch_embed.proj.bias", (D,), "b"),
    ]
    for i in range(cfg.depth):
        b = f"encoder.blocks.{i}"
        p += _ln(f"{b}.norm1", D)
        p += _linear(f"{b}.attn.qkv", D, 3 * D)
        p += _linear(f"{b}.attn.proj", D, D)
        p += _ln(f"{b}.norm2", D)
        p += _linear(f"{b}.mlp.fc1", D, hid)
        p += _linear(f"{b}.mlp.fc2", hid, D)
    p += _ln("encoder.norm", D)
    if cfg.num_tissue_classes > 0:
        p += _linear("encoder.head", D, cfg.num_tissue_classes)
    return p


def _encoder_sam(cfg: CellViTConfig) -> List[ParamSpec]:
    D, hid, hd = cfg.embed_dim, cfg.embed_dim * cfg.mlp_ratio, cfg.head_dim
    p: List[ParamSpec] = [
        ("encoder.pos_embed", (1, cfg.pos_grid, cfg.pos_grid, D), "emb"),
        ("encoder.patch_embed.proj.weight", (D, 3, cfg.patch_size, cfg.patch_size), "w"),
        ("encoder.patch_embed.proj.bias", (D,), "b"),
    ]
    for i in range(cfg.depth):
        b = f"encoder.blocks.{i}"
        side = cfg.pos_grid if i in cfg.global_attn_indexes else cfg.window_size
        p += _ln(f"{b}.norm1", D)
        p += [(f"{b}.attn.rel_pos_h", (2 * side - 1, hd), "emb"),
              (f"{b}.attn.rel_pos_w", (2 * side - 1, hd), "emb")]
        p += _linear(f"{b}.attn.qkv", D, 3 * D)
        p += _linear(f"{b}.attn.proj", D, D)
        p += _ln(f"{b}.norm2", D)
        p += _linear(f"{b}.mlp.lin1", D, hid)
        p += _linear(f"{b}.mlp.lin2", hid, D)
    C = cfg.neck_chans
    p += [("encoder.neck.0.weight", (C, D, 1, 1), "w")]
    p += _ln("encoder.neck.1", C)
    p += [("encoder.neck.2.weight", (C, C, 3, 3), "w")]
    p += _ln("encoder.neck.3", C)
    return p


def _branch(prefix: str, cfg: CellViTConfig, n_out: int) -> List[ParamSpec]:
    """create_upsampling_branch — cellvit.py:246-330."""
    D = cfg.embed_dim
    _, _, bott = cfg.skip_dims
    p = _convT(f"{prefix}.bottleneck_upsampler", D, bott)
    p += _conv_block(f"{prefix}.decoder3_upsampler.0", 2 * bott, bott)
    p += _conv_block(f"{prefix}.decoder3_upsampler.1", bott, bott)
    p += _conv_block(f"{prefix}.decoder3_upsampler.2", bott, bott)
    p += _convT(f"{prefix}.decoder3_upsampler.3", bott, 256)
    p += _conv_block(f"{prefix}.decoder2_upsampler.0", 512, 256)
    p += _conv_block(f"{prefix}.decoder2_upsampler.1", 256, 256)
    p += _convT(f"{prefix}.decoder2_upsampler.2", 256, 128)
    p += _conv_block(f"{prefix}.decoder1_upsampler.0", 256, 128)
    p += _conv_block(f"{prefix}.decoder1_upsampler.1", 128, 128)
    p += _convT(f"{prefix}.decoder1_upsampler.2", 128, 64)
    p += _conv_block(f"{prefix}.decoder0_header.0", 128, 64)
    p += _conv_block(f"{prefix}.decoder0_header.1", 64, 64)
    p += [(f"{prefix}.decoder0_header.2.weight", (n_out, 64, 1, 1), "w"),
          (f"{prefix}.decoder0_header.2.bias", (n_out,), "b")]
    return p


def param_specs(cfg: CellViTConfig) -> List[ParamSpec]:
    """Ordered (key, shape, kind) list == ``list(reference_model.state_dict().items())``."""
    D = cfg.embed_dim
    s11, s12, bott = cfg.skip_dims
    p = _encoder_vit(cfg) if cfg.arch == ARCH_VIT else _encoder_sam(cfg)
    p += _conv_block("decoder0.0", 3, 32) + _conv_block("decoder0.1", 32, 64)
    p += (_deconv_block("decoder1.0", D, s11) + _deconv_block("decoder1.1", s11, s12)
          + _deconv_block("decoder1.2", s12, 128))
    p += _deconv_block("decoder2.0", D, s11) + _deconv_block("decoder2.1", s11, 256)
    p += _deconv_block("decoder3.0", D, bott)
    nb, nh, nt = cfg.branch_out
    p += _branch("nuclei_binary_map_decoder", cfg, nb)
    p += _branch("hv_map_decoder", cfg, nh)
    p += _branch("nuclei_type_maps_decoder", cfg, nt)
    if cfg.arch == ARCH_SAM and cfg.num_tissue_classes > 0:
        p += _linear("classifier_head", cfg.neck_chans, cfg.num_tissue_classes)  # cellvit.py:568
    return p


def num_parameters(cfg: CellViTConfig) -> int:
    """Trainable + BN-buffer element count as torchinfo reports it (params only)."""
    n = 0
    for _, shape, kind in param_specs(cfg):
        if kind in ("bn_mean", "bn_var", "bn_count"):
            continue
        k = 1
        for s in shape:
            k *= s
        n += k
    return n

"""Drop-in host-side mirror of the reference model classes, backed by the HIP C-ABI library.

Reference surface reproduced here (same names, argument meaning and error behaviour):
  * ``CellViT`` / ``CellViT256`` / ``CellViTSAM`` constructors
      — models/segmentation/cell_segmentation/cellvit.py:57-75, 444-453, 514-522
  * ``.forward(x, retrieve_tokens=False) -> dict``            — cellvit.py:153-210, 586-644
  * ``.calculate_instance_map(predictions, magnification)``    — cellvit.py:332-383
  * ``.generate_instance_nuclei_map(instance_maps, type_preds)`` — cellvit.py:385-414
  * ``state_dict()`` / ``load_state_dict()`` with the reference key names (checkpoint layout of
    base_ml/base_trainer.py:229-245), ``.patch_size``, ``.num_nuclei_classes``, ``.embed_dim``.

The classes are ``torch.nn.Module`` containers holding the fp32 master parameters only; none of
their computation is done by torch.  ``forward`` hands device pointers of torch-owned tensors to
``libcellvit_amd.so`` (ctypes, no torch types across the boundary) on the caller's current HIP
stream.  There is no CPU fallback: a CPU tensor or a missing library raises.
"""
from __future__ import annotations

import ctypes as C
import math
import weakref
from collections import OrderedDict
from pathlib import Path
from typing import Dict, List, Literal, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .spec import (ARCH_SAM, ARCH_VIT, CellViTConfig, cellvit256_config, cellvit_sam_config, param_specs)


class _Node(nn.Module):
    """Parameter container; never executed."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("cellvit_amd parameter containers are not executable; call the model")


def _build_tree(root: nn.Module, cfg: CellViTConfig) -> None:
    for key, shape, kind in param_specs(cfg):
        parts = key.split(".")
        node = root
        for p in parts[:-1]:
            if p not in node._modules:
                node.add_module(p, _Node())
            node = node._modules[p]
        leaf = parts[-1]
        if kind == "bn_count":
            node.register_buffer(leaf, torch.tensor(0, dtype=torch.long))
        elif kind in ("bn_mean", "bn_var"):
            node.register_buffer(leaf, torch.zeros(shape) if kind == "bn_mean" else torch.ones(shape))
        else:
            if kind in ("ln_w", "bn_w"):
                t = torch.ones(shape)
            elif kind in ("b", "ln_b", "bn_b"):
                t = torch.zeros(shape)
            elif kind == "emb":
                t = torch.zeros(shape)
            else:
                fan_in = int(np.prod(shape[1:])) if kind == "w" else int(shape[0])
                t = torch.randn(shape) / math.sqrt(max(fan_in, 1))
            node.register_parameter(leaf, nn.Parameter(t))


class _Engine:
    """One finalized C handle (weights packed for one compute dtype) + its geometry cache."""

    def __init__(self, cfg: CellViTConfig, dtype: int, state: "OrderedDict[str, torch.Tensor]", debug: bool,
                 options: Optional[Dict[str, int]] = None):
        self.lib = _lib.load()
        self.cfg = cfg
        self.dtype = dtype
        self.geom: Optional[Tuple[int, int, int]] = None
        c = _lib.cv_config()
        c.arch = cfg.arch
        c.embed_dim, c.depth, c.num_heads, c.mlp_ratio = cfg.embed_dim, cfg.depth, cfg.num_heads, int(cfg.mlp_ratio)
        for i, e in enumerate(cfg.extract_layers):
            c.extract_layers[i] = int(e)
        c.num_nuclei_classes, c.num_tissue_classes = cfg.num_nuclei_classes, cfg.num_tissue_classes
        c.regression_loss = int(cfg.regression_loss)
        c.patch_size, c.window_size = cfg.patch_size, cfg.window_size
        c.n_global = len(cfg.global_attn_indexes)
        for i, g in enumerate(cfg.global_attn_indexes):
            c.global_attn_indexes[i] = int(g)
        c.neck_chans = cfg.neck_chans
        c.compute_dtype = dtype
        h = C.c_void_p()
        _lib.check(self.lib.cv_create(C.byref(c), C.byref(h)))
        self.h = h
        try:
            if debug:
                _lib.check(self.lib.cv_set_debug(self.h, 1))
            for name, value in (options or {}).items():
                _lib.check(self.lib.cv_set_option(self.h, name.encode(), int(value)))
            for key, t in state.items():
                a = t.detach().to("cpu")
                if a.dtype == torch.long:
                    arr, code = np.ascontiguousarray(a.numpy().astype(np.int64)), 2
                else:
                    arr, code = np.ascontiguousarray(a.float().numpy()), 1
                shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
                _lib.check(self.lib.cv_load_weight(self.h, key.encode(), arr.ctypes.data_as(C.c_void_p), code,
                                                   shape, arr.ndim))
            _lib.check(self.lib.cv_finalize(self.h))
        except Exception:
            self.close()
            raise

    def close(self):
        if getattr(self, "h", None):
            self.lib.cv_destroy(self.h)
            self.h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def set_derived(self, name: str, t: torch.Tensor) -> None:
        arr = np.ascontiguousarray(t.detach().float().cpu().numpy())
        shape = (C.c_int64 * 2)(*arr.shape)
        _lib.check(self.lib.cv_set_derived(self.h, name.encode(), arr.ctypes.data_as(C.c_void_p), shape, 2))


# ---------------------------------------------------------------------------------------------
# input-size dependent tables — computed with the same torch ops and arguments as the reference
# ---------------------------------------------------------------------------------------------
def _vit_pos_table(pos: torch.Tensor, patch: int, w: int, h: int) -> torch.Tensor:
    """vits_histo.py:377-402 (called as interpolate_pos_encoding(x, w=H_img, h=W_img))."""
    N = pos.shape[1] - 1
    npatch = (w // patch) * (h // patch)
    if npatch == N and w == h:
        return pos[0]
    dim = pos.shape[-1]
    w0, h0 = w // patch + 0.1, h // patch + 0.1
    g = int(math.sqrt(N))
    pp = F.interpolate(pos[:, 1:].reshape(1, g, g, dim).permute(0, 3, 1, 2),
                       scale_factor=(w0 / math.sqrt(N), h0 / math.sqrt(N)), mode="bicubic")
    assert int(w0) == pp.shape[-2] and int(h0) == pp.shape[-1]
    pp = pp.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((pos[:, 0].unsqueeze(0), pp), dim=1)[0]


def _rel_table(rel_pos: torch.Tensor, side: int) -> torch.Tensor:
    """Resize step of get_rel_pos (SAM/image_encoder.py:333-344) for q_size == k_size == side."""
    L = 2 * side - 1
    if rel_pos.shape[0] == L:
        return rel_pos
    r = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=L, mode="linear")
    return r.reshape(-1, L).permute(1, 0)


class CellViT(nn.Module):
    """CellViT with a generic ViT backbone (cellvit.py:26-151) on the MI355X-native engine.

    Args mirror the reference constructor; the dropout arguments are accepted and ignored (they are
    identity in eval mode, the only mode of this inference path).
    ``compute_dtype``: "auto" (fp16 under ``torch.autocast``, fp32 otherwise — the reference's
    ``mixed_precision`` switch, cell_detection.py:306-318), "fp16", "fp32", or "fp8" (SAM encoders only: the fp16 engine
    with OCP MX-fp8 qkv / fc1 / fc2 contractions on the CDNA4 block-scaled MFMA, BASELINE.json configs[4]).
    """

    def __init__(self, num_nuclei_classes: int, num_tissue_classes: int, embed_dim: int, input_channels: int,
                 depth: int, num_heads: int, extract_layers: List, mlp_ratio: float = 4, qkv_bias: bool = True,
                 drop_rate: float = 0, attn_drop_rate: float = 0, drop_path_rate: float = 0,
                 regression_loss: bool = False, *, _cfg: Optional[CellViTConfig] = None,
                 compute_dtype: str = "auto"):
        super().__init__()
        assert len(extract_layers) == 4, "Please provide 4 layers for skip connections"
        if _cfg is None:
            if input_channels != 3 or not qkv_bias:
                raise NotImplementedError("cellvit_amd builds the RGB / qkv_bias=True configuration only")
            _cfg = CellViTConfig(arch=ARCH_VIT, embed_dim=embed_dim, depth=depth, num_heads=num_heads,
                                 extract_layers=tuple(extract_layers), num_nuclei_classes=num_nuclei_classes,
                                 num_tissue_classes=num_tissue_classes, mlp_ratio=int(mlp_ratio),
                                 regression_loss=regression_loss, pos_grid=14, name="CellViT")
        self.cfg = _cfg
        self.patch_size = 16
        self.num_tissue_classes = num_tissue_classes
        self.num_nuclei_classes = num_nuclei_classes
        self.embed_dim = _cfg.embed_dim
        self.input_channels = 3
        self.depth = _cfg.depth
        self.num_heads = _cfg.num_heads
        self.mlp_ratio = mlp_ratio
        self.qkv_bias = True
        self.extract_layers = list(_cfg.extract_layers)
        self.drop_rate, self.attn_drop_rate, self.drop_path_rate = drop_rate, attn_drop_rate, drop_path_rate
        self.regression_loss = regression_loss
        self.skip_dim_11, self.skip_dim_12, self.bottleneck_dim = _cfg.skip_dims
        nb, nh, nt = _cfg.branch_out
        self.branches_output = {"nuclei_binary_map": nb, "hv_map": nh, "nuclei_type_maps": nt}
        self.compute_dtype = compute_dtype
        # fp8 engine only: False keeps attn.proj on fp16 (qkv / fc1 / fc2 on MX-fp8; the tighter accuracy bounds of DESIGN §4) — cv_set_option
        self.fp8_proj = True
        self.debug_taps = False
        _build_tree(self, _cfg)
        self._engines: Dict[Tuple[int, int, int], _Engine] = {}     # (device index, compute dtype, options) -> packed weights + workspace
        self._last_argmax = None
        self._last_maps = None
        self.eval()

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        msg = super().load_state_dict(state_dict, strict=strict)
        self.invalidate()
        return msg

    def invalidate(self) -> None:
        """Drop packed device weights (call after mutating parameters in place)."""
        for e in self._engines.values():
            e.close()
        self._engines = {}

    def freeze_encoder(self):
        for name, p in self._modules["encoder"].named_parameters():
            if name.split(".")[0] != "head":
                p.requires_grad = False

    def unfreeze_encoder(self):
        for p in self._modules["encoder"].parameters():
            p.requires_grad = True

    # ------------------------------------------------------------------ engine plumbing
    def _dtype_code(self) -> int:
        cd = self.compute_dtype
        if cd == "auto":
            return _lib.DTYPE_F16 if torch.is_autocast_enabled() else _lib.DTYPE_F32
        if cd in ("fp16", "f16", "half"):
            return _lib.DTYPE_F16
        if cd in ("fp32", "f32", "float"):
            return _lib.DTYPE_F32
        if cd in ("fp8", "f8", "mxfp8"):
            return _lib.DTYPE_F8
        raise ValueError(f"unknown compute_dtype {cd!r}")

    def _engine(self, dtype: int, device: torch.device) -> _Engine:
        """One handle per (device, dtype): weights and workspace live on the device that was current at creation."""
        idx = device.index if device.index is not None else torch.cuda.current_device()
        opt = int(bool(self.fp8_proj)) if dtype == _lib.DTYPE_F8 else 1
        e = self._engines.get((idx, dtype, opt))
        if e is None:
            e = _Engine(self.cfg, dtype, self.state_dict(), self.debug_taps, {"fp8_proj": opt} if dtype == _lib.DTYPE_F8 else None)
            self._engines[(idx, dtype, opt)] = e
        return e

    def engine_flags(self) -> int:
        """Engine choices of the last forward's geometry (`cv_geometry_flags`): bit 0 = window blocks keep V row-major, bit 1 = fp8
        engine with proj on MX-fp8.  0 before the first forward."""
        e = getattr(self, "_last_engine", None)
        return int(e.lib.cv_geometry_flags(e.h)) if e is not None else 0

    def _ensure_geometry(self, e: _Engine, B: int, H: int, W: int) -> None:
        if e.geom is not None and e.geom[1:] == (H, W) and B <= e.geom[0]:
            return
        cfg = self.cfg
        gh, gw = H // 16, W // 16
        if cfg.arch == ARCH_SAM and gh != gw:
            # reference: x + pos_embed[:, :gh, :gh, :] cannot broadcast (cell_segmentation/utils.py:222-224)
            raise RuntimeError(f"The size of tensor a ({gw}) must match the size of tensor b ({gh}) at "
                               "non-singleton dimension 2")
        maxb = max(B, e.geom[0] if (e.geom and e.geom[1:] == (H, W)) else 0)
        _lib.check(e.lib.cv_set_geometry(e.h, maxb, H, W))
        sd = {k: v.detach().float().cpu() for k, v in self.state_dict().items()
              if "pos_embed" in k or "rel_pos" in k}
        with torch.no_grad():
            if cfg.arch == ARCH_VIT:
                e.set_derived("pos_table", _vit_pos_table(sd["encoder.pos_embed"], 16, H, W))
            else:
                e.set_derived("pos_table", sd["encoder.pos_embed"][0, :gh, :gw, :].reshape(gh * gw, -1))
                for i in range(cfg.depth):
                    glob = i in cfg.global_attn_indexes
                    e.set_derived(f"rel_h.{i}", _rel_table(sd[f"encoder.blocks.{i}.attn.rel_pos_h"],
                                                           gh if glob else cfg.window_size))
                    e.set_derived(f"rel_w.{i}", _rel_table(sd[f"encoder.blocks.{i}.attn.rel_pos_w"],
                                                           gw if glob else cfg.window_size))
        e.geom = (maxb, H, W)

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, x: torch.Tensor, retrieve_tokens: bool = False) -> dict:
        """Forward pass (cellvit.py:153-210 / 586-644): BCHW float images -> dict of raw branch outputs."""
        assert x.shape[-2] % self.patch_size == 0, \
            "Img must have a shape of that is divisible by patch_size (token_size)"
        assert x.shape[-1] % self.patch_size == 0, \
            "Img must have a shape of that is divisible by patch_size (token_size)"
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("expected a BCHW batch with 3 channels")
        if not x.is_cuda:
            raise RuntimeError("cellvit_amd runs on the MI355X only: move the batch to a cuda device "
                               "(there is no CPU fallback; the CPU oracle lives under oracle/ for tests)")
        B, _, H, W = x.shape
        return self._run(x.contiguous().float(), None, B, H, W, retrieve_tokens)

    @torch.no_grad()
    def forward_u8(self, tiles_u8: torch.Tensor, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5),
                   retrieve_tokens: bool = False) -> dict:
        """forward() on RAW tiles: uint8 [B, H, W, 3] on the device.  The inference transform of the reference CLI
        (T.ToTensor + T.Normalize(mean, std), cell_detection.py:214-227) is evaluated inside the kernels that read the
        image, so `forward_u8(t) == forward(((t / 255 - mean) / std).permute(0, 3, 1, 2))` bit for bit."""
        if tiles_u8.dim() != 4 or tiles_u8.shape[-1] != 3 or tiles_u8.dtype != torch.uint8:
            raise ValueError("expected a uint8 batch [B, H, W, 3]")
        assert tiles_u8.shape[1] % self.patch_size == 0 and tiles_u8.shape[2] % self.patch_size == 0, \
            "Img must have a shape of that is divisible by patch_size (token_size)"
        if not tiles_u8.is_cuda:
            raise RuntimeError("cellvit_amd runs on the MI355X only: move the batch to a cuda device")
        B, H, W, _ = tiles_u8.shape
        nm = ((C.c_float * 3)(*[float(v) for v in mean]), (C.c_float * 3)(*[float(v) for v in std]))
        return self._run(tiles_u8.contiguous(), nm, B, H, W, retrieve_tokens)

    def _run(self, x: torch.Tensor, u8_norm, B: int, H: int, W: int, retrieve_tokens: bool) -> dict:
        with torch.cuda.device(x.device):
            e = self._engine(self._dtype_code(), x.device)
            self._ensure_geometry(e, B, H, W)
            cfg = self.cfg
            dev = x.device
            f32 = dict(device=dev, dtype=torch.float32)
            out_t = {
                # num_tissue_classes == 0: the head is nn.Identity, `tissue_types` = the pooled embedding
                # (vits_histo.py:359-362: [B, embed_dim]; cellvit.py:568-572: [B, neck channels])
                "tissue_types": torch.empty((B, cfg.num_tissue_classes if cfg.num_tissue_classes > 0 else
                                             (cfg.embed_dim if cfg.arch == ARCH_VIT else cfg.neck_chans)), **f32),
                "nuclei_binary_map": torch.empty((B, 2, H, W), **f32),
                "hv_map": torch.empty((B, 2, H, W), **f32),
                "nuclei_type_map": torch.empty((B, cfg.num_nuclei_classes, H, W), **f32),
            }
            if cfg.regression_loss:
                out_t["regression_map"] = torch.empty((B, 2, H, W), **f32)
            tokens = torch.empty((B, H // 16, W // 16, cfg.embed_dim), **f32) if retrieve_tokens else None
            bin_am = torch.empty((B, H, W), device=dev, dtype=torch.uint8)
            typ_am = torch.empty((B, H, W), device=dev, dtype=torch.uint8)
            o = _lib.cv_outputs()
            o.tissue_types = out_t["tissue_types"].data_ptr()
            o.nuclei_binary_map = out_t["nuclei_binary_map"].data_ptr()
            o.hv_map = out_t["hv_map"].data_ptr()
            o.nuclei_type_map = out_t["nuclei_type_map"].data_ptr()
            o.regression_map = out_t["regression_map"].data_ptr() if cfg.regression_loss else None
            o.tokens_nhwc = tokens.data_ptr() if tokens is not None else None
            o.binary_argmax = bin_am.data_ptr()
            o.type_argmax = typ_am.data_ptr()
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            if u8_norm is None:
                _lib.check(e.lib.cv_forward(e.h, x.data_ptr(), B, H, W, C.byref(o), stream))
            else:
                _lib.check(e.lib.cv_forward_u8(e.h, x.data_ptr(), u8_norm[0], u8_norm[1], B, H, W, C.byref(o), stream))
        out_dict = {"tissue_types": out_t["tissue_types"]}
        out_dict["nuclei_binary_map"] = out_t["nuclei_binary_map"]
        if cfg.regression_loss:
            out_dict["regression_map"] = out_t["regression_map"]
        out_dict["hv_map"] = out_t["hv_map"]
        out_dict["nuclei_type_map"] = out_t["nuclei_type_map"]
        if retrieve_tokens:
            out_dict["tokens"] = tokens.permute(0, 3, 1, 2)
        # u8 argmax planes of the two classification maps, written by the kernels that produce the logits; valid for
        # calculate_instance_map as long as the caller hands back these very tensors, unmodified (see there)
        self._last_argmax = (bin_am, typ_am)
        self._last_maps = tuple((weakref.ref(t), t._version) for t in (out_t["nuclei_binary_map"], out_t["nuclei_type_map"]))
        self._last_engine = e
        return out_dict

    def debug_tap(self, name: str, numel_hint: int) -> np.ndarray:
        """Read a named intermediate of the last forward (needs ``debug_taps = True`` before forward)."""
        e = self._last_engine
        buf = np.empty(numel_hint, dtype=np.float32)
        n = C.c_size_t()
        _lib.check(e.lib.cv_debug_read(e.h, name.encode(), buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(n)))
        return buf[: n.value]

    # ------------------------------------------------------------------ post-processing
    def calculate_instance_map(self, predictions: OrderedDict, magnification: Literal[20, 40] = 40
                               ) -> Tuple[torch.Tensor, List[dict]]:
        """cellvit.py:332-383 — instance map [B,H,W] float32 + per-image dict of nuclei.

        The reference takes torch.argmax of the two classification maps (:366-374).  When `predictions` still holds
        the tensors the last forward() returned, untouched (same objects, same in-place version), the u8 argmax planes
        the forward kernels wrote next to the logits are used; for any other maps (e.g. after the CLI's softmax,
        cell_detection.py:500-505) the channel argmax runs as a HIP kernel on the maps given."""
        from .postproc import calculate_instance_map as _cim
        planes = None
        if self._last_maps is not None and self._last_argmax is not None:
            (rb, vb), (rt, vt) = self._last_maps
            tb, tt = rb(), rt()
            if tb is not None and tt is not None and predictions.get("nuclei_binary_map") is tb \
                    and predictions.get("nuclei_type_map") is tt and tb._version == vb and tt._version == vt:
                planes = self._last_argmax
        return _cim(predictions, self.num_nuclei_classes, magnification, argmax_planes=planes)

    def generate_instance_nuclei_map(self, instance_maps: torch.Tensor, type_preds: List[dict]) -> torch.Tensor:
        """cellvit.py:385-414 — [B,H,W] ids + dicts -> [B, num_nuclei_classes, H, W] per-class id maps (float32, on
        the host like the reference).  Instances that are not in the dict (quirk 4: failed the contour test,
        post_proc:113-116) stay unpainted.  One id -> class lookup table per image instead of the reference's
        per-instance full-image comparisons; runs on whatever device `instance_maps` lives on."""
        batch_size, h, w = instance_maps.shape
        dev = instance_maps.device
        out = torch.zeros((batch_size, self.num_nuclei_classes, h, w), dtype=torch.float32, device=dev)
        for i in range(batch_size):
            im = instance_maps[i].to(torch.int64)
            n = int(im.max().item()) + 1 if im.numel() else 1
            lut = torch.full((max(n, 1),), -1, dtype=torch.int64, device=dev)
            ids = [int(k) for k in type_preds[i].keys() if 0 <= int(k) < n]
            if ids:
                lut[torch.tensor(ids, device=dev)] = torch.tensor([int(type_preds[i][k]["type"]) for k in ids], device=dev)
            cls = lut[im.clamp(min=0)]                      # class of every pixel's instance, -1 = not painted
            cls = torch.where(im > 0, cls, torch.full_like(cls, -1)) if 0 not in type_preds[i] else cls
            valid = cls >= 0
            out[i].view(self.num_nuclei_classes, -1)[cls[valid], valid.view(-1).nonzero(as_tuple=True)[0]] = \
                instance_maps[i][valid].to(torch.float32)
        return out.cpu()


class CellViT256(CellViT):
    """CellViT with the ViT-256 (ViT-S/16, HIPT) backbone settings — cellvit.py:428-493."""

    def __init__(self, model256_path: Union[Path, str, None], num_nuclei_classes: int, num_tissue_classes: int,
                 drop_rate: float = 0, attn_drop_rate: float = 0, drop_path_rate: float = 0,
                 regression_loss: bool = False, compute_dtype: str = "auto"):
        cfg = cellvit256_config(num_nuclei_classes, num_tissue_classes, regression_loss)
        super().__init__(num_nuclei_classes, num_tissue_classes, cfg.embed_dim, 3, cfg.depth, cfg.num_heads,
                         list(cfg.extract_layers), 4, True, drop_rate, attn_drop_rate, drop_path_rate,
                         regression_loss, _cfg=cfg, compute_dtype=compute_dtype)
        self.model256_path = model256_path

    def load_pretrained_encoder(self, model256_path: str):
        """cellvit.py:483-493: DINO teacher checkpoint -> encoder weights (non-strict)."""
        state_dict = torch.load(str(model256_path), map_location="cpu")["teacher"]
        state_dict = {k.replace("module.", ""): v for k, v in state_dict.items()}
        state_dict = {k.replace("backbone.", ""): v for k, v in state_dict.items()}
        msg = self._modules["encoder"].load_state_dict(state_dict, strict=False)
        self.invalidate()
        print(f"Loading checkpoint: {msg}")


class CellViTSAM(CellViT):
    """CellViT with a SAM ViTDet backbone (SAM-B / SAM-L / SAM-H) — cellvit.py:496-665."""

    def __init__(self, model_path: Union[Path, str, None], num_nuclei_classes: int, num_tissue_classes: int,
                 vit_structure: Literal["SAM-B", "SAM-L", "SAM-H"], drop_rate: float = 0,
                 regression_loss: bool = False, compute_dtype: str = "auto"):
        cfg = cellvit_sam_config(vit_structure, num_nuclei_classes, num_tissue_classes, regression_loss)
        super().__init__(num_nuclei_classes, num_tissue_classes, cfg.embed_dim, 3, cfg.depth, cfg.num_heads,
                         list(cfg.extract_layers), 4, True, drop_rate, 0, 0, regression_loss, _cfg=cfg,
                         compute_dtype=compute_dtype)
        self.model_path = model_path
        self.prompt_embed_dim = 256
        self.encoder_global_attn_indexes = list(cfg.global_attn_indexes)

    def load_pretrained_encoder(self, model_path):
        """cellvit.py:574-584."""
        state_dict = torch.load(str(model_path), map_location="cpu")
        msg = self._modules["encoder"].load_state_dict(state_dict, strict=False)
        self.invalidate()
        print(f"Loading checkpoint: {msg}")


def build_model(arch: str, run_conf: dict, compute_dtype: str = "auto") -> CellViT:
    """Model factory of the inference CLI (cell_detection.py:142-212): checkpoint `arch` + config."""
    data, model = run_conf["data"], run_conf["model"]
    if arch == "CellViT":
        return CellViT(num_nuclei_classes=data["num_nuclei_classes"], num_tissue_classes=data["num_tissue_classes"],
                       embed_dim=model["embed_dim"], input_channels=model.get("input_channels", 3),
                       depth=model["depth"], num_heads=model["num_heads"], extract_layers=model["extract_layers"],
                       regression_loss=model.get("regression_loss", False), compute_dtype=compute_dtype)
    if arch == "CellViT256":
        return CellViT256(None, data["num_nuclei_classes"], data["num_tissue_classes"],
                          regression_loss=model.get("regression_loss", False), compute_dtype=compute_dtype)
    if arch == "CellViTSAM":
        return CellViTSAM(None, data["num_nuclei_classes"], data["num_tissue_classes"], model["backbone"],
                          regression_loss=model.get("regression_loss", False), compute_dtype=compute_dtype)
    raise NotImplementedError(f"Unknown model type: {arch} (shared-decoder variants are out of scope)")

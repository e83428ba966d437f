"""GPU: the PRODUCT route of the inference CLI, bit-exact where it is index / byte work:
  forward (raw u8 tiles, fused inference transform) -> u8 argmax planes written by the forward kernels ->
  on-device post-processing on exactly those planes -> on-device cell-token pooling.
Checked against torch.argmax of the logits, the CPU oracle (oracle/postproc_ref.c) on the same maps, and the
reference's token-pooling formula (cell_detection.py:396-409)."""
import numpy as np
import pytest
import torch

from helpers import load_case
from test_gpu_forward import _model
from test_gpu_postproc import _assert_dicts_equal

pytestmark = pytest.mark.gpu


def _oracle_dicts(type_am, bin_am, hv, mag=40):
    from oracle import postproc_ref as P
    pm = np.stack([type_am.astype(np.float32), bin_am.astype(np.float32), hv[0], hv[1]], -1)
    return P.postprocess_tile(pm, 6, mag)


def _check_planes_and_route(m, x, tag, mag=40):
    """argmax planes == torch.argmax(logits) (both branches); post-processing of the planes == oracle on the same maps."""
    from cellvit_amd.postproc import _params, postprocess_device, records_to_dicts
    out = m(x.cuda(), retrieve_tokens=True)
    bin_am, typ_am = m._last_argmax
    torch.cuda.synchronize()
    tb = out["nuclei_binary_map"].argmax(1).to(torch.uint8)
    tt = out["nuclei_type_map"].argmax(1).to(torch.uint8)
    assert torch.equal(bin_am, tb), f"{tag}: binary argmax plane differs in {(bin_am != tb).sum().item()} pixels"
    assert torch.equal(typ_am, tt), f"{tag}: type argmax plane differs in {(typ_am != tt).sum().item()} pixels"
    obj, ks = _params(mag)
    inst, recs, n_recs, contours, n_pts = postprocess_device(bin_am, typ_am, out["hv_map"], 6, obj, ks)   # as the CLI does
    torch.cuda.synchronize()
    dicts = records_to_dicts(recs, n_recs, contours, n_pts)
    hv = out["hv_map"].cpu().numpy()
    n_cells = 0
    for b in range(x.shape[0]):
        o_inst, o_d = _oracle_dicts(typ_am[b].cpu().numpy(), bin_am[b].cpu().numpy(), hv[b], mag)
        assert np.array_equal(inst[b].cpu().numpy(), o_inst), f"{tag}: instance map of tile {b} differs from the oracle"
        _assert_dicts_equal(dicts[b], o_d)
        n_cells += len(o_d)
    # calculate_instance_map on the dict forward just returned takes the same planes and must agree
    inst2, dicts2 = m.calculate_instance_map(out, magnification=mag)
    assert np.array_equal(inst2.numpy(), inst.float().cpu().numpy())
    for a, b_ in zip(dicts2, dicts):
        _assert_dicts_equal(a, b_)
    print(f"\n[{tag}] argmax planes exact; post-processing of the planes == oracle; {n_cells} instances")
    return out


@pytest.mark.parametrize("name,dtype", [("vit256_256", "fp32"), ("vit256_256", "fp16"), ("samh_256", "fp32"), ("samh_256", "fp16")])
def test_argmax_planes_and_postproc_route(name, dtype):
    cfg, sd, x, _ = load_case(name)
    _check_planes_and_route(_model(cfg, sd, dtype), x, f"{name} {dtype}")


@pytest.mark.parametrize("dtype", ["fp16", "fp32"])
def test_argmax_planes_and_postproc_route_samh_1024(dtype):
    """BASELINE.json configs[2] size: one full 1024^2 SAM-H tile through forward -> planes -> post-processing."""
    cfg, sd, x, _ = load_case("samh_1024")
    _check_planes_and_route(_model(cfg, sd, dtype), x, f"samh_1024 {dtype}")


@pytest.mark.parametrize("dtype", ["fp16", "fp32"])
def test_argmax_ties_resolve_to_the_first_maximum(dtype):
    """Deliberately tied logits: (a) a zeroed type head (all six channels equal the zero bias) must give class 0
    everywhere; (b) channels 1 and 3 made identical must never report 3; (c) binary head with equal rows -> 0."""
    cfg, sd, x, _ = load_case("vit256_256")
    sd = {k: v.clone() for k, v in sd.items()}
    wk, bk = "nuclei_type_maps_decoder.decoder0_header.2.weight", "nuclei_type_maps_decoder.decoder0_header.2.bias"
    bw, bb = "nuclei_binary_map_decoder.decoder0_header.2.weight", "nuclei_binary_map_decoder.decoder0_header.2.bias"
    sd2 = dict(sd)
    sd2[wk] = torch.zeros_like(sd[wk]); sd2[bk] = torch.zeros_like(sd[bk])
    sd2[bw] = sd[bw].clone(); sd2[bw][1] = sd2[bw][0]; sd2[bb] = sd[bb].clone(); sd2[bb][1] = sd2[bb][0]
    m = _model(cfg, sd2, dtype)
    out = m(x.cuda())
    bin_am, typ_am = m._last_argmax
    torch.cuda.synchronize()
    assert torch.equal(out["nuclei_type_map"], torch.zeros_like(out["nuclei_type_map"]))
    assert int(typ_am.max()) == 0, "all-tied type logits must resolve to class 0"
    assert torch.equal(out["nuclei_binary_map"][:, 0], out["nuclei_binary_map"][:, 1])
    assert int(bin_am.max()) == 0, "tied binary logits must resolve to class 0"
    sd3 = dict(sd)
    sd3[wk] = sd[wk].clone(); sd3[wk][3] = sd3[wk][1]; sd3[bk] = sd[bk].clone(); sd3[bk][3] = sd3[bk][1]
    m3 = _model(cfg, sd3, dtype)
    out3 = m3(x.cuda())
    _, typ3 = m3._last_argmax
    torch.cuda.synchronize()
    assert torch.equal(out3["nuclei_type_map"][:, 1], out3["nuclei_type_map"][:, 3])
    assert not bool((typ3 == 3).any()), "channel 3 duplicates channel 1: the first maximum wins"
    assert torch.equal(typ3, out3["nuclei_type_map"].argmax(1).to(torch.uint8))
    assert bool((typ3 == 1).any())


def test_calculate_instance_map_after_softmax_uses_the_argmax_kernel():
    """The reference CLI overwrites the two maps with their softmax before calculate_instance_map
    (cell_detection.py:500-505): the planes no longer belong to the dict's tensors, the HIP channel-argmax runs."""
    from cellvit_amd.postproc import argmax_channels
    cfg, sd, x, _ = load_case("vit256_256")
    m = _model(cfg, sd, "fp32")
    out = m(x.cuda())
    inst_a, d_a = m.calculate_instance_map(out, 40)
    out["nuclei_binary_map"] = torch.softmax(out["nuclei_binary_map"], dim=1)
    out["nuclei_type_map"] = torch.softmax(out["nuclei_type_map"], dim=1)
    for k in ("nuclei_binary_map", "nuclei_type_map"):
        assert torch.equal(argmax_channels(out[k]), out[k].argmax(1).to(torch.uint8)), k
    inst_b, d_b = m.calculate_instance_map(out, 40)
    # softmax is monotone; rounding can only create ties, which both argmax implementations resolve alike
    tb = out["nuclei_binary_map"].argmax(1).cpu().numpy().astype(np.uint8)
    tt = out["nuclei_type_map"].argmax(1).cpu().numpy().astype(np.uint8)
    o_inst, o_d = _oracle_dicts(tt[0], tb[0], out["hv_map"][0].cpu().numpy())
    assert np.array_equal(inst_b[0].numpy(), o_inst)
    _assert_dicts_equal(d_b[0], o_d)
    # in-place modification of a returned tensor also invalidates the planes
    out2 = m(x.cuda())
    out2["nuclei_type_map"].mul_(-1.0)
    inst_c, _ = m.calculate_instance_map(out2, 40)
    tt2 = out2["nuclei_type_map"].argmax(1).cpu().numpy().astype(np.uint8)
    tb2 = out2["nuclei_binary_map"].argmax(1).cpu().numpy().astype(np.uint8)
    o_inst2, _ = _oracle_dicts(tt2[0], tb2[0], out2["hv_map"][0].cpu().numpy())
    assert np.array_equal(inst_c[0].numpy(), o_inst2)


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_forward_u8_equals_forward_of_the_normalised_tensor(dtype):
    """F0 fused: raw u8 tiles through cv_forward_u8 == forward(ToTensor + Normalize) bit for bit, and the stand-alone
    normalise op equals the torch restatement of the reference transform (cell_detection.py:214-227) exactly."""
    import ctypes as C
    from cellvit_amd import _lib
    from cellvit_amd.weights import synthetic_tile_u8
    cfg, sd, _, _ = load_case("vit256_256")
    m = _model(cfg, sd, dtype)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    u8 = torch.from_numpy(np.stack([synthetic_tile_u8(i, size=256, he_like=bool(i)) for i in range(2)])).cuda()
    # T.ToTensor: uint8 HWC -> float CHW / 255 ; T.Normalize: (x - mean) / std — torch CPU ops, where the reference's
    # DataLoader workers run them (IEEE division; torch's GPU kernels may multiply by a reciprocal instead)
    ref = u8.cpu().permute(0, 3, 1, 2).float().div(255.0)
    ref = ((ref - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)).contiguous().cuda()
    nrm = torch.empty((2, 3, 256, 256), device="cuda", dtype=torch.float32)
    _lib.check(_lib.load().cv_op_normalize_u8(u8.data_ptr(), (C.c_float * 3)(*mean), (C.c_float * 3)(*std), nrm.data_ptr(), 2, 256,
                                              256, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert torch.equal(nrm, ref), f"normalise op differs: max {(nrm - ref).abs().max().item()}"
    a = m.forward_u8(u8, mean, std, retrieve_tokens=True)
    am_a = [t.clone() for t in m._last_argmax]
    a = {k: v.clone() for k, v in a.items()}
    b = m(ref, retrieve_tokens=True)
    torch.cuda.synchronize()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    for p, q in zip(am_a, m._last_argmax):
        assert torch.equal(p, q)


def test_pool_cell_tokens_matches_the_reference_formula():
    """F12: cv_pool_tokens on device records == mean over tokens[idx, :, r0:r1, c0:c1] with the reference's
    floor / ceil / uint8 index arithmetic (cell_detection.py:396-409), for every record of a batch."""
    from cellvit_amd.postproc import _params, pool_cell_tokens, postprocess_device, records_to_dicts
    from cellvit_amd.synth import synth_nuclei_maps
    B, T, D = 3, 512, 384
    maps = [synth_nuclei_maps(40 + i, T, 800 + 200 * i) for i in range(B)]
    dev = torch.device("cuda", 0)
    tm = torch.from_numpy(np.stack([m[0] for m in maps])).to(dev)
    bm = torch.from_numpy(np.stack([m[1] for m in maps])).to(dev)
    hv = torch.from_numpy(np.stack([m[2] for m in maps])).to(dev)
    obj, ks = _params(40)
    inst, recs, n_recs, contours, n_pts = postprocess_device(bm, tm, hv, 6, obj, ks)
    g = torch.Generator(device="cpu").manual_seed(0)
    tokens_nhwc = torch.randn((B, T // 16, T // 16, D), generator=g).to(dev)
    tokens = tokens_nhwc.permute(0, 3, 1, 2)                      # what forward returns under "tokens"
    pooled, off = pool_cell_tokens(tokens, recs, n_recs, 16)
    torch.cuda.synchronize()
    dicts, index = records_to_dicts(recs, n_recs, contours, n_pts, return_index=True)
    tok_cpu = tokens.cpu()
    pooled = pooled.cpu()
    n = 0
    for b in range(B):
        assert len(dicts[b]) > 50
        for cell, slot in zip(dicts[b].values(), index[b]):
            bb_index = cell["bbox"] / 16
            bb_index[0, :] = np.floor(bb_index[0, :])
            bb_index[1, :] = np.ceil(bb_index[1, :])
            bb_index = bb_index.astype(np.uint8)
            ct = tok_cpu[b, :, bb_index[0, 0]:bb_index[1, 0], bb_index[0, 1]:bb_index[1, 1]]
            want = torch.mean(ct.reshape(D, -1).T, dim=0)
            got = pooled[off[b] + slot]
            assert torch.allclose(got, want, rtol=1e-6, atol=1e-6), (b, slot, (got - want).abs().max().item())
            n += 1
    print(f"\n[pool_tokens] {n} cells pooled on the device == reference formula")


def test_capacity_overflow_is_reported():
    """more instances / contour points than the handle holds must raise, not return truncated arrays."""
    from cellvit_amd import postproc as PP
    from cellvit_amd.synth import synth_nuclei_maps
    tm, bm, hv, _ = synth_nuclei_maps(3, 512, 2400)              # ~500 instances on the 512^2 tile
    dev = torch.device("cuda", 0)
    args = (torch.from_numpy(bm)[None].to(dev), torch.from_numpy(tm)[None].to(dev), torch.from_numpy(hv)[None].to(dev), 6, 10, 21)
    try:
        PP.set_capacity(inst_div=512 * 512 // 256, pts_div=8)            # 256 record slots < ~500 cells
        inst, recs, n_recs, contours, n_pts = PP.postprocess_device(*args)
        assert int(n_recs[0]) > recs.shape[1]
        with pytest.raises(PP.CapacityError):
            PP.records_to_dicts(recs, n_recs, contours, n_pts)
        PP.set_capacity(inst_div=128, pts_div=512 * 512 // 4096)         # 4096 contour points < what ~500 cells need
        inst2, recs2, n_recs2, contours2, n_pts2 = PP.postprocess_device(*args)
        assert int(n_pts2[0]) > contours2.shape[1]
        with pytest.raises(PP.CapacityError):
            PP.records_to_dicts(recs2, n_recs2, contours2, n_pts2)
        # the instance map itself does not depend on the record capacity
        PP.set_capacity()
        inst3, recs3, n_recs3, contours3, n_pts3 = PP.postprocess_device(*args)
        assert torch.equal(inst, inst3) and torch.equal(inst2, inst3)
        assert 0 < len(PP.records_to_dicts(recs3, n_recs3, contours3, n_pts3)[0]) <= int(n_recs[0])
    finally:
        PP.set_capacity()


def test_stage_events_order_a_second_stream_behind_the_forward():
    """cv_stream_wait_stage (the tile loop's release point for the previous batch's post-processing): not before a forward has recorded the
    stage; once armed, work enqueued on a second stream behind stage 2 / stage 1 sees everything the forward wrote BEFORE that stage (the tokens
    come out of the encoder, ahead of both stages), and the forward's results do not depend on the events."""
    import ctypes as C
    from cellvit_amd import _lib
    from cellvit_amd.weights import synthetic_tile_u8
    cfg, sd, _, _ = load_case("vit256_256")
    m = _model(cfg, sd, "fp16")
    u8 = torch.from_numpy(np.stack([synthetic_tile_u8(i, size=256, he_like=True) for i in range(2)])).cuda()
    ref = {k: v.clone() for k, v in m.forward_u8(u8, retrieve_tokens=True).items()}
    e = m._last_engine
    s2 = torch.cuda.Stream()
    with pytest.raises(ValueError):
        _lib.check(e.lib.cv_stream_wait_stage(e.h, 3, C.c_void_p(s2.cuda_stream)))
    _lib.check(e.lib.cv_stream_wait_stage(e.h, 0, None))                  # arm
    with pytest.raises(RuntimeError):                                     # armed, but no forward has recorded the stages yet
        _lib.check(e.lib.cv_stream_wait_stage(e.h, 2, C.c_void_p(s2.cuda_stream)))
    for stage in (1, 2):
        out = m.forward_u8(u8, retrieve_tokens=True)
        with torch.cuda.stream(s2):
            _lib.check(e.lib.cv_stream_wait_stage(e.h, stage, C.c_void_p(s2.cuda_stream)))
            tok = out["tokens"].clone()                                    # on s2, ordered behind the stage event only
            out["tokens"].record_stream(s2)
        s2.synchronize()
        assert torch.equal(tok, ref["tokens"])
        torch.cuda.synchronize()
        for k in ref:
            assert torch.equal(out[k], ref[k]), k

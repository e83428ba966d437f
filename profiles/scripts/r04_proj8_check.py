"""fp8 engine with proj on MX-fp8 against the same engine with the fp16 proj (CVA_NO_PROJ8=1, ablation flavour), SAM-H 1024^2 golden case.
    CVA_LIB=abl [CVA_NO_PROJ8=1] python tools/experiments/r04_proj8_check.py out.npz"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import load_case
from test_gpu_forward import _model
cfg, sd, x, gold = load_case("samh_1024")
m = _model(cfg, sd, "fp8")
out = m(x.cuda(), retrieve_tokens=True)
torch.cuda.synchronize()
print("engine flags", m.engine_flags())
np.savez(sys.argv[1], **{k: out[k].float().cpu().numpy() for k in ("nuclei_binary_map", "hv_map", "nuclei_type_map", "tokens")})
if len(sys.argv) > 2:
    a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
    for k in a.files:
        print(k, "max abs diff between the two runs", float(np.abs(a[k] - b[k]).max()), "abs max", float(np.abs(a[k]).max()))
st = {}
for k in ("nuclei_binary_map", "hv_map", "nuclei_type_map"):
    a = out[k].float().cpu().numpy()
    c = gold[k + "_center"].shape[-1]; H = a.shape[-1]; y0 = (H - c) // 2
    gk = np.concatenate([gold[k + "_center"], gold[k + "_corner"]], 0)
    ak = np.concatenate([a[..., y0:y0 + c, y0:y0 + c], a[..., :c, :c]], 0)
    st[k] = (float(np.abs(ak - gk).max()), float(np.abs(ak - gk).mean()), float((ak.argmax(1) == gk.argmax(1)).mean()))
print("stats vs golden crops (max abs, mean abs, argmax agreement):", st)

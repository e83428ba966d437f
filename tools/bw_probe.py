import torch, time
x = torch.randn(65536, 1280, device="cuda")
y = torch.empty_like(x)
h = torch.empty(65536, 1280, device="cuda", dtype=torch.float16)
def t(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
ms = t(lambda: y.copy_(x)); print("fp32 copy  %.1f us  %.2f TB/s" % (ms * 1e3, 2 * x.numel() * 4 / ms / 1e9))
ms = t(lambda: h.copy_(x)); print("fp32->fp16 %.1f us  %.2f TB/s" % (ms * 1e3, x.numel() * 6 / ms / 1e9))
ms = t(lambda: torch.nn.functional.layer_norm(x, (1280,))); print("torch LN fp32->fp32 %.1f us  %.2f TB/s" % (ms * 1e3, x.numel() * 8 / ms / 1e9))
ms = t(lambda: x.sum()); print("read-only sum %.1f us  %.2f TB/s" % (ms * 1e3, x.numel() * 4 / ms / 1e9))

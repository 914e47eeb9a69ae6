import torch, time
x = torch.empty(256*1024*1024, device="cuda", dtype=torch.float32)  # 1 GiB
y = torch.empty_like(x)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n
ms=t(lambda: x.fill_(1.0)); print("fill 1GiB   %.1f us  %.2f TB/s" % (ms*1e3, 1.0737/ms*1e-3*1e3/1))
ms=t(lambda: y.copy_(x));   print("copy 1GiB   %.1f us  %.2f TB/s (r+w)" % (ms*1e3, 2*1.0737/ms))
ms=t(lambda: torch.mul(x, 2.0, out=y)); print("scale 1GiB  %.1f us  %.2f TB/s (r+w)" % (ms*1e3, 2*1.0737/ms))
ms=t(lambda: x.sum());      print("sum 1GiB    %.1f us  %.2f TB/s (r)" % (ms*1e3, 1.0737/ms))

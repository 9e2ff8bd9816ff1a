import torch
dev = torch.device("cuda:0")
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for mb in (100, 403, 1600):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, device=dev); b = torch.randn(n, device=dev)
    us = t(lambda: a.fill_(1.0)); print("fill %d MB: %.1f us %.2f TB/s write" % (mb, us, mb / us))
    us = t(lambda: a.copy_(b)); print("copy %d MB: %.1f us %.2f TB/s read+write" % (mb, us, 2 * mb / us))
    us = t(lambda: torch.add(a, b, out=a)); print("add  %d MB: %.1f us %.2f TB/s 2r+1w" % (mb, us, 3 * mb / us))

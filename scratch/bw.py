import torch, time
n = 1 << 30
a = torch.empty(n, dtype=torch.uint8, device='cuda'); b = torch.empty(n, dtype=torch.uint8, device='cuda')
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
ms = t(lambda: a.view(torch.int32).fill_(7)); print("fill 1GiB: %.3f ms  %.2f TB/s written" % (ms, n / ms / 1e9))
ms = t(lambda: b.copy_(a)); print("copy 1GiB: %.3f ms  %.2f TB/s total" % (ms, 2 * n / ms / 1e9))
ms = t(lambda: a.view(torch.int32).sum()); print("read 1GiB (sum): %.3f ms  %.2f TB/s" % (ms, n / ms / 1e9))
a4 = torch.empty(n // 4 // 4, dtype=torch.int32, device='cuda')
idx = torch.randint(0, 256, (n // 4,), device='cuda', dtype=torch.int64)
tab = torch.arange(256, device='cuda', dtype=torch.int32)
out = torch.empty(n // 4, dtype=torch.int32, device='cuda')
ms = t(lambda: torch.index_select(tab, 0, idx, out=out)); print("torch gather 256Mi int32 (int64 idx): %.3f ms" % ms)

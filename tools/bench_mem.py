import torch, time
n = 120_000_000  # doubles = 960 MB
a = torch.empty(n, dtype=torch.float64, device="cuda"); b = torch.empty(n, dtype=torch.float64, device="cuda")
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: a.fill_(1.5)); print(f"fill 960MB: {ms:.4f} ms  {0.96/ms*1e3:.0f} GB/s write")
ms = t(lambda: b.copy_(a)); print(f"copy 960MB: {ms:.4f} ms  {2*0.96/ms*1e3:.0f} GB/s read+write")
ms = t(lambda: a.sum()); print(f"sum 960MB: {ms:.4f} ms  {0.96/ms*1e3:.0f} GB/s read")
ms = t(lambda: torch.add(a, 1.0, out=b)); print(f"add 960MB: {ms:.4f} ms  {2*0.96/ms*1e3:.0f} GB/s read+write")

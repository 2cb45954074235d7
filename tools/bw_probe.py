import torch, time
n = 32*120*160*256
a = torch.empty(n, device="cuda"); b = torch.randn(n, device="cuda"); c = torch.randn(n, device="cuda")
def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
gb = n * 4 / 1e9
x = t(lambda: a.fill_(1.0)); print(f"fill  {x:.3f} ms  {gb / x * 1e3 / 1e3:.2f} TB/s (write {gb:.2f} GB)")
x = t(lambda: a.copy_(b)); print(f"copy  {x:.3f} ms  {2 * gb / x:.2f} TB/s")
x = t(lambda: torch.add(b, c, out=a)); print(f"add   {x:.3f} ms  {3 * gb / x:.2f} TB/s")
x = t(lambda: torch.relu_(b)); print(f"relu_ {x:.3f} ms  {2 * gb / x:.2f} TB/s")

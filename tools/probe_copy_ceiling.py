import torch, time
x = torch.empty(2 << 30, dtype=torch.uint8, device="cuda").random_(0, 256)
y = torch.empty_like(x)
for n in (2 << 30, 1 << 30):
    a, b = x[:n], y[:n]
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"device copy of {n>>20} MiB: {ms:.3f} ms, {2*n/ms/1e6:.0f} GB/s read+write")

import time, torch
dev = torch.device("cuda:0")
x = torch.randn(4096, 4096, device=dev)
def busy():
    for _ in range(3):
        torch.mm(x, x)
shape = (1, 64, 64, 16)
def probe(name, fn, reps=8):
    ts = []
    for _ in range(reps):
        busy()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t = fn()
        y = t.float().sum()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"{name:50s} median {sorted(ts)[len(ts)//2]:8.1f} us   min {min(ts):8.1f}")
g = torch.Generator().manual_seed(1)
probe("device randn (reference, no copy)", lambda: torch.randn(shape, device=dev, dtype=torch.bfloat16))
probe("pageable .to(dev)", lambda: torch.randn(shape, generator=g, dtype=torch.bfloat16).to(dev))
probe("pinned randn(pin_memory) .to(non_blocking)", lambda: torch.randn(shape, generator=g, dtype=torch.bfloat16, pin_memory=True).to(dev, non_blocking=True))
pin = torch.empty(shape, dtype=torch.bfloat16, pin_memory=True)
def f3():
    torch.randn(shape, generator=g, dtype=torch.bfloat16, out=pin)
    return pin.to(dev, non_blocking=True)
probe("persistent pinned buffer .to(non_blocking)", f3)
dbuf = torch.empty(shape, dtype=torch.bfloat16, device=dev)
def f4():
    torch.randn(shape, generator=g, dtype=torch.bfloat16, out=pin)
    dbuf.copy_(pin, non_blocking=True)
    return dbuf
probe("persistent pinned -> persistent device copy_", f4)
big = (6, 135, 240, 16)
pinb = torch.empty(big, dtype=torch.bfloat16, pin_memory=True)
probe("4K-size pageable .to(dev)", lambda: torch.randn(big, generator=g, dtype=torch.bfloat16).to(dev), reps=3)
def f5():
    torch.randn(big, generator=g, dtype=torch.bfloat16, out=pinb)
    return pinb.to(dev, non_blocking=True)
probe("4K-size pinned .to(non_blocking)", f5, reps=3)
# CPU cost of the draws
t0 = time.perf_counter(); torch.randn(shape, generator=g, dtype=torch.bfloat16); t1 = time.perf_counter()
torch.randn(big, generator=g, dtype=torch.bfloat16); t2 = time.perf_counter()
print(f"host randn: {1e3*(t1-t0):.2f} ms (512^2 latent), {1e3*(t2-t1):.1f} ms (4K latent)")

import torch, time
x = torch.empty(1<<30, dtype=torch.uint8).pin_memory()
d = torch.empty(1<<30, dtype=torch.uint8, device="cuda")
for name, f in (("H2D", lambda: d.copy_(x, non_blocking=True)), ("D2H", lambda: x.copy_(d, non_blocking=True))):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize()
    print(name, "%.1f GB/s" % (5 * (1<<30) / (time.perf_counter() - t0) / 1e9))
# both directions at once
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
x2 = torch.empty(1<<30, dtype=torch.uint8).pin_memory(); d2 = torch.empty(1<<30, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    with torch.cuda.stream(s1): d.copy_(x, non_blocking=True)
    with torch.cuda.stream(s2): x2.copy_(d2, non_blocking=True)
torch.cuda.synchronize()
print("both", "%.1f GB/s each" % (5 * (1<<30) / (time.perf_counter() - t0) / 1e9))

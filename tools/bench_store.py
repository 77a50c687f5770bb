"""How fast does the chip take plain writes of a (m, n) bf16 matrix? (torch fill / copy, for calibration)"""
import sys, time, torch
dev = torch.device("cuda:0")
def timeit(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(it): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / it * 1e6
for m, n in [(56504, 1024), (56504, 256), (120000, 64)]:
    y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    x = torch.randn(m, n, device=dev).bfloat16()
    t1 = timeit(lambda: y.fill_(1.0)); t2 = timeit(lambda: y.copy_(x))
    print(f"m={m} n={n} {m*n*2/1e6:.0f} MB: fill {t1:.1f} us = {m*n*2/t1/1e6:.2f} TB/s; copy {t2:.1f} us = {2*m*n*2/t2/1e6:.2f} TB/s (r+w)")

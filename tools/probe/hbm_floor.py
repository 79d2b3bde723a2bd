"""What HBM rates do plain streaming kernels reach on this box?  (floor for the epilogue-bound GEMMs)

fill = write only, sum = read only, copy = read + write, add3 = 2 reads + 1 write; sizes as the stage-1/2 MLP tensors.
"""
import torch

dev = torch.device("cuda")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for mb in (19, 77, 154, 308, 616):
    n = mb * 1024 * 1024 // 2
    x = torch.randn(n, device=dev).bfloat16()
    y = torch.empty_like(x)
    z = torch.empty_like(x)
    t_fill = timeit(lambda: y.fill_(1.0))
    t_sum = timeit(lambda: x.float().sum()) if False else None
    t_copy = timeit(lambda: y.copy_(x))
    t_add = timeit(lambda: torch.add(x, y, out=z))
    b = n * 2
    print(f"{mb:4d} MB  fill {t_fill:7.1f} us {b / t_fill / 1e6:6.2f} TB/s | copy {t_copy:7.1f} us {2 * b / t_copy / 1e6:6.2f} TB/s"
          f" | add3 {t_add:7.1f} us {3 * b / t_add / 1e6:6.2f} TB/s")

"""Attainable HBM rates on this GPU with plain streaming kernels (calibration for DESIGN.md §4)."""
import time, torch
dev = "cuda:0"
n = 1536 * 1000 * 1000          # 6.144 GB of fp32, the size of E on BASELINE config 3
x = torch.rand(n, device=dev)
y = torch.empty_like(x)


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - a) / reps


gb = n * 4 / 1e9
ms = t(lambda: y.copy_(x)); print("copy  (r+w) %.3f ms  %.2f TB/s" % (ms * 1e3, 2 * gb / ms / 1e3))
ms = t(lambda: x.sum());     print("read        %.3f ms  %.2f TB/s" % (ms * 1e3, gb / ms / 1e3))
ms = t(lambda: y.fill_(1.0)); print("write       %.3f ms  %.2f TB/s" % (ms * 1e3, gb / ms / 1e3))
ms = t(lambda: torch.add(x, 1.0, out=y)); print("add   (r+w) %.3f ms  %.2f TB/s" % (ms * 1e3, 2 * gb / ms / 1e3))

"""HBM bandwidth reference points on this device (torch kernels): fill (pure write), copy (read + write), sum (pure read)."""
import torch, time
dev = torch.device("cuda", 0)
n = 680 * 1024 * 1024 // 8
x = torch.empty(n, dtype=torch.float64, device=dev); y = torch.empty_like(x)
def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
gb = n * 8 / 1e9
ms = t(lambda: x.fill_(1.0)); print("fill  %.0f MB: %.3f ms = %.2f TB/s written" % (gb * 1e3, ms, gb / ms))
ms = t(lambda: y.copy_(x)); print("copy  %.0f MB: %.3f ms = %.2f TB/s read + %.2f TB/s written" % (gb * 1e3, ms, gb / ms, gb / ms))
ms = t(lambda: x.sum()); print("sum   %.0f MB: %.3f ms = %.2f TB/s read" % (gb * 1e3, ms, gb / ms))

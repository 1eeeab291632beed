"""What the device's own copy achieves for the 4:2:0 decode's traffic shape (1.6 GB read + 1.6 GB written): torch's
device-to-device copy (a tuned copy kernel / SDMA-free blit) and a fill, timed with events.  python tools/memcpy_rate.py"""
import torch
dev = torch.device("cuda:0")
n = 1_598_423_040  # bytes: the coefficient arena of 256 x 1080p 4:2:0 (the pixels are 1,592,524,800)
src = torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255)
dst = torch.empty(n, dtype=torch.uint8, device=dev)
def timed(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
t = timed(lambda: dst.copy_(src))
print(f"copy_  {n/1e9:.2f} GB -> {n/1e9:.2f} GB: {t:.4f} ms = {2*n/t/1e9:.0f} GB/s (read + write)")
s4, d4 = src.view(torch.int32), dst.view(torch.int32)
t = timed(lambda: torch.add(s4, 1, out=d4))
print(f"add 1  (int32 elementwise, same bytes):   {t:.4f} ms = {2*n/t/1e9:.0f} GB/s (read + write)")
t = timed(lambda: dst.fill_(7))
print(f"fill_  {n/1e9:.2f} GB: {t:.4f} ms = {n/t/1e9:.0f} GB/s (write only)")
t = timed(lambda: s4.sum())
print(f"sum    {n/1e9:.2f} GB: {t:.4f} ms = {n/t/1e9:.0f} GB/s (read only)")

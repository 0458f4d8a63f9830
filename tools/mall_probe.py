"""Does a producer -> consumer pair of streaming kernels run faster when the tensor between them fits the 256 MiB last-level
(Infinity) cache?  y = a + 1 (write y) followed by z = y * 2 (read y), per size; and a chunked walk over a 692 MB tensor."""
import torch
dev = "cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for mb in (16, 32, 64, 128, 192, 256, 384, 692, 1384):
    n = mb * (1 << 20) // 4
    a = torch.randn(n, device=dev); y = torch.empty_like(a); z = torch.empty_like(a)
    def pair():
        torch.add(a, 1.0, out=y)
        torch.mul(y, 2.0, out=z)
    us = t(pair)
    print(f"{mb:5d} MB: pair {us:8.1f} us = {4 * mb / 1024 / us * 1e6 / 1e3:6.2f} TB/s over 4 passes")
    del a, y, z
# chunked: the same two passes over 692 MB in chunks of c MB (producer chunk then consumer chunk)
n = 692 * (1 << 20) // 4
a = torch.randn(n, device=dev); y = torch.empty_like(a); z = torch.empty_like(a)
for c in (692, 346, 173, 87, 43):
    k = c * (1 << 20) // 4
    def chunked():
        for o in range(0, n, k):
            torch.add(a[o:o + k], 1.0, out=y[o:o + k])
            torch.mul(y[o:o + k], 2.0, out=z[o:o + k])
    print(f"692 MB in chunks of {c:4d} MB: {t(chunked):8.1f} us")

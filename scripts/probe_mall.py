"""Does the 256 MiB Infinity Cache (MALL) serve a read that follows a write of the same buffer?
(What a grouped two-pass FFT would rely on.)  For sizes 16 MiB ... 2 GiB: write W (fill), then read it (sum),
timed separately with events; the same read again (read-after-read); and a copy W -> W2.
usage: python scripts/probe_mall.py"""
import torch

dev = torch.device("cuda:0")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2] * 1e-3


big = torch.empty(1 << 30, dtype=torch.float32, device=dev)  # 4 GiB flush buffer
for mb in (16, 32, 64, 128, 192, 256, 512, 1024, 2048):
    n = mb * (1 << 20) // 4
    w = torch.empty(n, dtype=torch.float32, device=dev)
    w2 = torch.empty_like(w)
    acc = torch.zeros(1, device=dev)

    def write():
        w.fill_(1.0)

    def read():
        acc.copy_(w.sum())

    def write_then_read():
        w.fill_(1.0)
        acc.copy_(w.sum())

    def flush_then_read():
        big.add_(1.0)
        acc.copy_(w.sum())

    def flush():
        big.add_(1.0)

    t_w, t_r, t_wr, t_f, t_fr = timed(write), timed(read), timed(write_then_read), timed(flush, 5), timed(flush_then_read, 5)
    t_c = timed(lambda: w2.copy_(w))
    gb = n * 4 / 1e9
    print("%5d MiB: write %.0f GB/s | read again (warm) %.0f GB/s | read right after write %.0f GB/s | read after a 4 GiB flush %.0f GB/s | copy %.0f GB/s (r+w)"
          % (mb, gb / t_w, gb / t_r, gb / max(t_wr - t_w, 1e-9), gb / max(t_fr - t_f, 1e-9), 2 * gb / t_c), flush=True)
    del w, w2

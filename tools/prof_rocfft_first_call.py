import time, torch
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for n in (1 << 17, 1 << 17, 1 << 15, 1 << 20):
    x = torch.randn((256, n), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f = torch.fft.rfft(x, n=2 * n, dim=1); y = torch.fft.irfft(f * f.conj(), n=2 * n, dim=1)
    torch.cuda.synchronize(); print(n, "%.3f s" % (time.perf_counter() - t0))

"""rocFFT throughput probe through torch.fft (decides whether the constant-coefficient FFT preconditioner is affordable)."""
import time, torch
dev = 'cuda'
def bench(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for shape in [(2, 4096, 4096), (3, 8192, 8192), (2, 512, 512, 512), (2, 512, 512), (2, 1024, 1024)]:
    for dt in (torch.float32, torch.float64):
        x = torch.randn(*shape, device=dev, dtype=dt)
        nd = len(shape) - 1
        dims = tuple(range(1, nd + 1))
        t0 = time.perf_counter(); y = torch.fft.rfftn(x, dim=dims); torch.cuda.synchronize(); t_first = time.perf_counter() - t0
        t0 = time.perf_counter(); z = torch.fft.irfftn(y, s=shape[1:], dim=dims); torch.cuda.synchronize(); t_first_i = time.perf_counter() - t0
        f = bench(lambda: torch.fft.rfftn(x, dim=dims))
        i = bench(lambda: torch.fft.irfftn(y, s=shape[1:], dim=dims))
        nbytes = x.numel() * x.element_size()
        print('%s %s: rfftn %.3f ms (%.0f GB/s of in+out), irfftn %.3f ms; first call %.2f s / %.2f s' %
              (shape, dt, f, 2 * nbytes / f / 1e6, i, t_first, t_first_i), flush=True)
        del x, y, z

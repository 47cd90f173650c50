import torch, time
dev = "cuda:0"
def bench(fn, flops, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, flops / ms / 1e9
M = 25600
torch.backends.cuda.matmul.allow_tf32 = False
for (n, k) in [(600, 200), (800, 200), (200, 800), (200, 200)]:
    a = torch.randn(M, k, device=dev); w = torch.randn(n, k, device=dev)
    ms, tf = bench(lambda: a @ w.t(), 2 * M * n * k)
    print(f"NT  M={M} N={n} K={k}: {ms*1e3:.1f} us  {tf:.1f} TFLOP/s")
for (n, k) in [(200, 600), (200, 800), (800, 200)]:
    a = torch.randn(M, k, device=dev); w = torch.randn(k, n, device=dev)
    ms, tf = bench(lambda: a @ w, 2 * M * n * k)
    print(f"NN  M={M} N={n} K={k}: {ms*1e3:.1f} us  {tf:.1f} TFLOP/s")
for (m, n) in [(600, 200), (800, 200), (200, 800)]:
    dy = torch.randn(M, m, device=dev); x = torch.randn(M, n, device=dev)
    ms, tf = bench(lambda: dy.t() @ x, 2 * M * n * m)
    print(f"TN  M={m} N={n} K={M}: {ms*1e3:.1f} us  {tf:.1f} TFLOP/s")

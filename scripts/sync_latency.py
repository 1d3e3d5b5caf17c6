"""Latency of a host<-device scalar read (`.item()`) on an idle stream and behind queued work."""
import os, sys, time
import torch
dev = torch.device('cuda:0')
x = torch.zeros(1, device=dev)
a = torch.randn(4096, 4096, device=dev)
torch.cuda.synchronize()
def timeit(fn, n):
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6
print('env ROC_ACTIVE_WAIT_TIMEOUT=%s' % os.environ.get('ROC_ACTIVE_WAIT_TIMEOUT'))
print('idle .item(): %.1f us' % timeit(lambda: x.item(), 200))
def behind(k):
    for _ in range(k): a.mul_(1.0)
    x.item()
for k in (1, 10, 50):
    # a.mul_ on 64 MB ~ 25 us each
    t = timeit(lambda: behind(k), 50)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        for _ in range(k): a.mul_(1.0)
    torch.cuda.synchronize(); base = (time.perf_counter() - t0) / 50 * 1e6
    print('%d kernels then .item(): %.1f us  (same kernels, one final sync: %.1f us/iter)' % (k, t, base))
ev = torch.cuda.Event()
def ev_sync():
    a.mul_(1.0); ev.record(); ev.synchronize()
print('kernel + event.synchronize(): %.1f us' % timeit(ev_sync, 100))
def st_sync():
    a.mul_(1.0); torch.cuda.current_stream().synchronize()
print('kernel + stream.synchronize(): %.1f us' % timeit(st_sync, 100))

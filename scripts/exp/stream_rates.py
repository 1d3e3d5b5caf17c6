"""GPU experiment: what the memory system gives plain streaming kernels (torch's fill / copy / add) at the sizes of the
BatchNorm passes -- the write rate against the read rate."""
import torch

dev = torch.device('cuda')


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for n, c in ((396662, 96), (396662, 256), (226469, 96), (105363, 128)):
    x = torch.randn(n, c, device=dev).bfloat16()
    y = torch.empty_like(x)
    z = torch.empty_like(x)
    mb = x.numel() * 2 / 1e6
    t_fill = timeit(lambda: y.zero_())
    t_copy = timeit(lambda: y.copy_(x))
    t_add = timeit(lambda: torch.add(x, z, out=y))
    t_sum = timeit(lambda: x.sum())
    t_relu = timeit(lambda: torch.relu(x, out=y) if False else torch.clamp_min(x, 0, out=y))
    print('%7d x %3d (%6.1f MB)  fill %6.1f us (%.2f TB/s written)   copy %6.1f us (%.2f TB/s r+w)   add %6.1f us (%.2f TB/s)   '
          'sum %6.1f us (%.2f TB/s read)   clamp %6.1f us' % (n, c, mb, t_fill, mb / t_fill, t_copy, 2 * mb / t_copy, t_add,
                                                                 3 * mb / t_add, t_sum, mb / t_sum, t_relu), flush=True)

"""What the HBM gives plain streaming kernels on this box: fill (write only), copy (read + write), reduction (read only).
Context for the write-bound kernels (K5 strings, transpose, merge): their roofline fraction is quoted against the 8 TB/s
spec peak, the rates below are what a trivially coalesced kernel reaches."""
import json
import torch

def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3

out = {}
for gb in (0.25, 2, 16):
    n = int(gb * (1 << 30)) // 8
    x = torch.empty(n, dtype=torch.int64, device="cuda")
    y = torch.empty(n, dtype=torch.int64, device="cuda")
    x.fill_(3)
    t_fill = timed(lambda: y.fill_(1))
    t_zero = timed(lambda: y.zero_())
    t_copy = timed(lambda: y.copy_(x))
    t_sum = timed(lambda: x.sum())
    B = n * 8
    out["%g GiB" % gb] = {"fill_TBps": B / t_fill / 1e12, "memset_TBps": B / t_zero / 1e12, "copy_TBps_rw": 2 * B / t_copy / 1e12, "sum_TBps": B / t_sum / 1e12}
    del x, y
print(json.dumps(out, indent=1))

"""HBM stream-copy microbenchmark (quote the MEASURED peak next to the 8 TB/s spec): device-to-device copies and a read-only reduction of
buffers far larger than the 256 MB MALL / L2, timed with events on the current stream.  Prints GB/s (read + written bytes / time)."""
import json
import torch

assert torch.cuda.is_available()
out = {}
for gib in (1, 4):
    n = gib * (1 << 30) // 4
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_(); b = torch.empty_like(a)
    for name, fn, bytes_moved in (("copy", lambda: b.copy_(a), 2 * n * 4), ("read_sum", lambda: a.sum(), n * 4), ("fill", lambda: b.fill_(1.0), n * 4)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out[f"{name}_{gib}GiB_GBs"] = round(bytes_moved / ms / 1e6, 1)
    del a, b
print(json.dumps(out))

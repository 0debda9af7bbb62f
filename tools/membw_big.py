"""fill_ / sum / copy_ rate vs footprint (TLB reach check)."""
import torch
dev = torch.device('cuda:0')
for gib in (4, 32, 96):
    n = gib << 28
    x = torch.empty(n, device=dev)
    def t(fn, reps=5):
        fn(); torch.cuda.synchronize(); ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e-3)
        return sorted(ts)[len(ts) // 2]
    w = 4 * n / t(lambda: x.fill_(1.0))
    r = 4 * n / t(lambda: x.sum())
    print(f'{gib:3d} GiB  fill {w/1e12:.3f} TB/s   sum {r/1e12:.3f} TB/s', flush=True)
    del x

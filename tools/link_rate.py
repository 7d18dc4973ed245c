"""Developer measurement: what the host link of this box does -- H2D alone, D2H alone, both at once (page-locked buffers, two
streams) -- the ceiling for hare_shoot_batch from host buffers (104 bytes per ray cross it: 48 up, 56 down)."""
import torch
N = 256 << 20
h_up = torch.empty(N, dtype=torch.uint8).pin_memory(); h_dn = torch.empty(N, dtype=torch.uint8).pin_memory()
d_up = torch.empty(N, dtype=torch.uint8, device="cuda"); d_dn = torch.empty(N, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    s1.synchronize(); s2.synchronize(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def up():
    with torch.cuda.stream(s1): d_up.copy_(h_up, non_blocking=True)
    s1.synchronize()


def dn():
    with torch.cuda.stream(s2): h_dn.copy_(d_dn, non_blocking=True)
    s2.synchronize()


def both():
    with torch.cuda.stream(s1): d_up.copy_(h_up, non_blocking=True)
    with torch.cuda.stream(s2): h_dn.copy_(d_dn, non_blocking=True)
    s1.synchronize(); s2.synchronize()


tu, td, tb = timed(up), timed(dn), timed(both)
print("H2D alone %.1f GB/s | D2H alone %.1f GB/s | both at once %.1f GB/s aggregate (%.1f each way)" % (N / tu / 1e9, N / td / 1e9, 2 * N / tb / 1e9, N / tb / 1e9))

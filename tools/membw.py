"""Ceilings on this MI355X for the streaming kernels: fill / copy at cache-resident and HBM-resident sizes."""
import torch, time
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e-3
for mb in (82, 164, 1640):
    n = mb * 1000 * 1000
    a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty(n, dtype=torch.uint8, device="cuda"); c = torch.empty(n, dtype=torch.uint8, device="cuda")
    tf = t(lambda: a.fill_(1)); tc = t(lambda: b.copy_(a))
    a32, b32, c32 = a.view(torch.int32), b.view(torch.int32), c.view(torch.int32)
    tadd = t(lambda: torch.add(a32, b32, out=c32))  # 2 reads + 1 write, like a cross
    print(f"{mb:5d} MB: fill {n/tf/1e12:.2f} TB/s ({tf*1e6:.1f} us) | copy {2*n/tc/1e12:.2f} TB/s ({tc*1e6:.1f} us) | 2R+1W {3*n/tadd/1e12:.2f} TB/s ({tadd*1e6:.1f} us)")

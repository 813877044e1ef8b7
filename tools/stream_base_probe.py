import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
dev = torch.device("cuda", 0)
def timeit(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for name, n in (("925MB (r21d 144ch)", 32*16*56*56*144), ("411MB (64ch)", 32*16*56*56*64), ("205MB", 32*8*28*28*256)):
    a = torch.randn(n, device=dev); b = torch.randn(n, device=dev); c = torch.empty(n, device=dev)
    t1 = timeit(lambda: torch.mul(a, 2.0, out=c))            # 1R + 1W
    t2 = timeit(lambda: torch.add(a, b, out=c))              # 2R + 1W
    t3 = timeit(lambda: c.copy_(a))
    t4 = timeit(lambda: torch.sum(a))
    print(f"{name:20s} mul(1R1W) {2*n*4/t1/1e9:6.2f} TB/s  add(2R1W) {3*n*4/t2/1e9:6.2f} TB/s  copy {2*n*4/t3/1e9:6.2f} TB/s  sum(1R) {n*4/t4/1e9:6.2f} TB/s", flush=True)

import time, torch
x = torch.randn(64, 3, 3, 224, 224)              # 115.6 MB pageable
p = torch.empty_like(x).pin_memory()
d = torch.empty_like(x, device="cuda")
torch.cuda.synchronize()
for name, fn in (("host->pinned copy_", lambda: p.copy_(x)),
                 ("pinned->device (non_blocking)+sync", lambda: (d.copy_(p, non_blocking=True), torch.cuda.synchronize())),
                 ("pageable->device .to()", lambda: (x.to("cuda"), torch.cuda.synchronize())),
                 ("host->host pageable copy_", lambda: x.clone())):
    fn(); t0 = time.perf_counter()
    for _ in range(5): fn()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name:40s} {dt*1e3:7.2f} ms  {x.numel()*4/dt/1e9:6.1f} GB/s")
print("threads", torch.get_num_threads())

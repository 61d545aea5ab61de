import torch, time
x = torch.rand(256,4,96,96).pin_memory(); y = torch.empty_like(x, device='cuda')
for name, src in (("pinned", x), ("pageable", torch.rand(256,4,96,96))):
    for _ in range(3): y.copy_(src, non_blocking=True)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): y.copy_(src, non_blocking=True)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
    print(name, "%.2f ms  %.1f GB/s" % (dt*1e3, x.numel()*4/dt/1e9))

import torch, time
for mb in (2.5, 16, 256):
    n=int(mb*1e6)
    h=torch.empty(n,dtype=torch.uint8).pin_memory()
    d=torch.empty(n,dtype=torch.uint8,device='cuda')
    reps=max(4,int(2e9/n))
    for s in (1,2):
        streams=[torch.cuda.Stream() for _ in range(s)]
        torch.cuda.synchronize()
        t0=time.perf_counter()
        for i in range(reps):
            with torch.cuda.stream(streams[i%s]):
                d.copy_(h,non_blocking=True)
        torch.cuda.synchronize()
        dt=time.perf_counter()-t0
        print("H2D %.1f MB x%d on %d stream(s): %.1f GB/s"%(mb,reps,s,n*reps/dt/1e9))
for mb in (16.8, 256):
    n = int(mb * 1e6)
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device='cuda')
    reps = max(4, int(2e9 / n))
    for s in (1, 2, 3):
        streams = [torch.cuda.Stream() for _ in range(s)]
        hs = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(s)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            with torch.cuda.stream(streams[i % s]):
                hs[i % s].copy_(d, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("D2H %.1f MB x%d on %d stream(s): %.1f GB/s" % (mb, reps, s, n * reps / dt / 1e9))

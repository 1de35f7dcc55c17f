import torch, time
for mb in (46, 369):
    n = mb * 1024 * 1024 // 8
    d = torch.empty(n, dtype=torch.float64, device='cuda'); h = torch.empty(n, dtype=torch.float64, pin_memory=True)
    h.copy_(d, non_blocking=True); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): h.copy_(d, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print(f"D2H {mb} MB pinned: {mb/1024/dt:.1f} GiB/s")
    hp = torch.empty(n, dtype=torch.float64)
    t = time.perf_counter(); hp.copy_(d); torch.cuda.synchronize(); print(f"D2H {mb} MB pageable: {mb/1024/(time.perf_counter()-t):.1f} GiB/s")
# two halves of one buffer on two streams at once (two SDMA engines): does the link go faster than one copy stream?
n = 46 * 1024 * 1024 // 8
d = torch.empty(n, dtype=torch.float64, device='cuda'); h = torch.empty(n, dtype=torch.float64, pin_memory=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
h.copy_(d, non_blocking=True); torch.cuda.synchronize()
for streams in (1, 2):
    t = time.perf_counter()
    for _ in range(20):
        if streams == 1:
            with torch.cuda.stream(s1): h.copy_(d, non_blocking=True)
        else:
            with torch.cuda.stream(s1): h[: n // 2].copy_(d[: n // 2], non_blocking=True)
            with torch.cuda.stream(s2): h[n // 2:].copy_(d[n // 2:], non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print(f"D2H 46 MB pinned on {streams} stream(s): {46/1024/dt:.1f} GiB/s")

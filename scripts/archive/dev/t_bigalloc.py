import time, torch
torch.cuda.init()
for gb in (4, 27, 27, 55, 27):
    t0 = time.time(); x = torch.empty(int(gb * 1e9), dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); t1 = time.time()
    x.fill_(1); torch.cuda.synchronize(); t2 = time.time()
    del x; torch.cuda.empty_cache(); torch.cuda.synchronize(); t3 = time.time()
    print("hipMalloc %2d GB: alloc %.4f s, first touch %.4f s, free %.4f s" % (gb, t1 - t0, t2 - t1, t3 - t2), flush=True)

import time, torch, numpy as np
torch.cuda.init(); x = torch.zeros(10, device="cuda"); torch.cuda.synchronize()
t=time.perf_counter(); a = torch.empty(60000, dtype=torch.float64).pin_memory(); print("first pin 480KB %.2f ms" % (1e3*(time.perf_counter()-t)))
t=time.perf_counter(); b = torch.empty(940000, dtype=torch.float64).pin_memory(); print("pin 7.5MB %.2f ms" % (1e3*(time.perf_counter()-t)))
t=time.perf_counter(); c = torch.empty(940000, dtype=torch.float64).pin_memory(); print("pin 7.5MB again %.2f ms" % (1e3*(time.perf_counter()-t)))
d = torch.empty(940000, dtype=torch.float64, device="cuda"); h = torch.empty(940000, dtype=torch.float64)
for name, dst in (("pageable", h), ("pinned", b)):
    for rep in range(3):
        torch.cuda.synchronize(); t=time.perf_counter(); dst.copy_(d, non_blocking=True); torch.cuda.synchronize(); print("download 7.5MB %s %.2f ms" % (name, 1e3*(time.perf_counter()-t)))
xs = np.random.rand(60000); xd = torch.empty(60000, dtype=torch.float64, device="cuda")
for rep in range(3):
    torch.cuda.synchronize(); t=time.perf_counter(); xd.copy_(torch.from_numpy(xs)); torch.cuda.synchronize(); print("upload 480KB pageable %.3f ms" % (1e3*(time.perf_counter()-t)))

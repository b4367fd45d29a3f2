import time, numpy as np, torch
rows, c, nch = 131072, 16, 64
C = np.random.default_rng(0).standard_normal((rows * nch, c), dtype=np.float32); Ch = torch.from_numpy(C)
cpin = [torch.empty((rows, c), pin_memory=True) for _ in range(2)]; cdev = [torch.empty((rows, c), device="cuda") for _ in range(2)]
pn = [t.numpy() for t in cpin]
def run(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(nch): fn(k, k * rows)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-44s host %.3f ms/chunk  total %.3f ms/chunk" % (label, (t1 - t0) / nch * 1e3, (t2 - t0) / nch * 1e3), flush=True)
print("torch threads", torch.get_num_threads())
for rep in range(2):
    run("torch copy_ pageable->pinned", lambda k, lo: cpin[k % 2].copy_(Ch[lo:lo + rows]))
    run("np.copyto pageable->pinned", lambda k, lo: np.copyto(pn[k % 2], C[lo:lo + rows]))
    run("direct H2D from pageable slice", lambda k, lo: cdev[k % 2].copy_(Ch[lo:lo + rows], non_blocking=True))
    run("np.copyto + H2D pinned", lambda k, lo: (np.copyto(pn[k % 2], C[lo:lo + rows]), cdev[k % 2].copy_(cpin[k % 2], non_blocking=True)))
t0 = time.perf_counter(); x = Ch.to("cuda"); torch.cuda.synchronize(); print("whole C (%.0f MB) .to(cuda): %.1f ms" % (C.nbytes / 1e6, (time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); x = Ch.to("cuda"); torch.cuda.synchronize(); print("whole C again: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
torch.set_num_threads(8)
run("torch copy_ pageable->pinned, 8 threads", lambda k, lo: cpin[k % 2].copy_(Ch[lo:lo + rows]))

"""Random shapes through the differentiable seam (rnvp_backward_cond, rnvp_inverse_backward) against torch autograd over the float64
eager restatement (oracle/torch_cpu.py::EagerFlow) on the same weights.  GPU box only:  python scripts/autograd_fuzz.py [cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.torch_cpu import EagerFlow
from probaforms_amd.models import NormalizingFlow, RealNVPLayer, StandardNormalPrior

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = {"dC": 0.0, "dX": 0.0, "dP": 0.0, "dz": 0.0, "dCi": 0.0, "dPi": 0.0}
bad = 0
for it in range(cases):
    L = int(rng.integers(1, 6)); d = int(rng.choice([1, 2, 3, 5, 8, 16, 17, 31, 40, 70])); c = int(rng.choice([0, 1, 3, 4, 9, 17]))
    nh = int(rng.integers(1, 4)); hidden = tuple(int(rng.choice([3, 10, 16, 24, 33, 64])) for _ in range(nh))
    act = str(rng.choice(["tanh", "relu"])); n = int(rng.choice([1, 15, 16, 17, 100, 257, 1000]))
    if os.environ.get("FUZZ_WIDE") == "1":      # round 6: nets whose tile image the any-shape MFMA kernel cannot hold -> the VALU kernel
        L = int(rng.integers(1, 4)); d = int(rng.choice([2, 5, 16, 17, 40])); c = int(rng.choice([0, 1, 4, 9]))
        hidden = (int(rng.choice([300, 512, 700])),) if rng.random() < 0.7 else (int(rng.choice([300, 512])), int(rng.choice([16, 64])))
        n = int(rng.choice([1, 15, 17, 100, 257]))
    if d == 1:
        masks = [torch.tensor([i % 2]) for i in range(L)]
    elif rng.random() < 0.5:
        masks = [(torch.arange(d) + i) % 2 for i in range(L)]
    else:
        masks = [torch.from_numpy(rng.integers(0, 2, size=d).astype(np.int64)) for _ in range(L)]
    torch.manual_seed(1000 + it)
    layers = [RealNVPLayer(d, c, masks[i], hidden, act) for i in range(L)]
    nf = NormalizingFlow(layers, StandardNormalPrior(d, "cuda"))
    try:
        nf.engine()
    except Exception as e:
        print("case %d skipped at engine(): %s" % (it, e)); continue
    flat = torch.cat([p.detach().reshape(-1) for p in nf.parameters()]).cpu().double().numpy()
    ref = EagerFlow(L, d, c, hidden, act).double(); ref.load_flat(flat); ref.masks = [m.clone() for m in masks]
    ref.prior = torch.distributions.MultivariateNormal(torch.zeros(d, dtype=torch.float64), torch.eye(d, dtype=torch.float64))
    g = torch.Generator().manual_seed(it)
    X0 = torch.randn(n, d, generator=g); C0 = torch.randn(n, c, generator=g) if c else None; w = torch.rand(n, generator=g) + 0.5
    A = torch.randn(n, d, generator=g)

    def rel(a, b):
        b = b.double().cpu(); return float((a.double().cpu() - b).abs().max()) / max(float(b.abs().max()), 1e-30)

    def pgrads():
        gs = []
        for t, s_ in zip(ref.nets_t, ref.nets_s):
            gs += [p.grad for p in list(t.parameters()) + list(s_.parameters())]
        sc = max(float(q.abs().max()) for q in gs) or 1e-30
        return max(float((p.grad.double().cpu() - q).abs().max()) for p, q in zip(nf.parameters(), gs)) / sc
    try:
        # forward direction
        X = X0.cuda().requires_grad_(True); C = C0.cuda().requires_grad_(True) if c else None
        (-(w.cuda() * nf.log_prob_samples(X, C)).sum() / n).backward()
        Xr = X0.double().requires_grad_(True); Cr = C0.double().requires_grad_(True) if c else None
        (-(w.double() * ref.log_prob_rows(Xr, Cr)[0]).sum() / n).backward()
        e = {"dX": rel(X.grad, Xr.grad), "dP": pgrads(), "dC": rel(C.grad, Cr.grad) if c else 0.0}
        for p in nf.parameters(): p.grad = None
        for p in ref.parameters(): p.grad = None
        # inverse direction
        Z = X0.cuda().requires_grad_(True); C = C0.cuda().requires_grad_(True) if c else None
        x = nf.engine().inverse_autograd(Z, C)
        ((A.cuda() * x).sum() / n).backward()
        Zr = X0.double().requires_grad_(True); Cr = C0.double().requires_grad_(True) if c else None
        ((A.double() * ref.inverse_rows(Zr, Cr)).sum() / n).backward()
        e.update({"dz": rel(Z.grad, Zr.grad), "dPi": pgrads(), "dCi": rel(C.grad, Cr.grad) if c else 0.0})
    except RuntimeError as ex:
        print("case %d L=%d d=%d c=%d hidden=%s %s n=%d: %s" % (it, L, d, c, hidden, act, n, str(ex)[:120])); continue
    tol = 2e-5 if act == "relu" else 1e-5          # a float32 ReLU net can sit on the other side of a kink from the float64 one
    flag = any((not np.isfinite(v)) or v > tol for v in e.values())
    if flag and act == "relu":
        # is it a kink?  the same eager restatement in FLOAT32 against the float64 one: if that differs as much, a pre-activation
        # sits within rounding of zero and the two arithmetics took different branches -- no implementation can agree with both
        ref32 = EagerFlow(L, d, c, hidden, act); ref32.load_flat(flat.astype(np.float32)); ref32.masks = [m.clone() for m in masks]
        Z32 = X0.clone().requires_grad_(True); C32 = C0.clone().requires_grad_(True) if c else None
        ((A * ref32.inverse_rows(Z32, C32)).sum() / n).backward()
        k32 = rel(Z32.grad, Zr.grad)
        print("      float32 eager vs float64 eager, d loss / d z through the inverse: %.1e%s" % (k32, "  (a ReLU kink: not counted)" if k32 > tol else ""))
        if k32 > tol:
            flag = False
        # how many ROWS carry the disagreement, and does the sample itself agree?  One or two rows out of n with an exact sample
        # is a kink the float64 run took on the other side (a different summation order is enough); an implementation error shows in
        # every row
        rowerr = (Z.grad.double().cpu() - Zr.grad).abs().max(dim=1).values / max(float(Zr.grad.abs().max()), 1e-30)
        nbad = int((rowerr > tol).sum())
        xerr = float((x.detach().double().cpu() - ref.inverse_rows(X0.double(), C0.double() if c else None).detach()).abs().max())
        print("      rows beyond tolerance in d loss / d z: %d of %d; max |x - x_ref| %.1e" % (nbad, n, xerr))
        if nbad <= max(2, n // 200) and xerr < 1e-4:
            flag = False
            print("      -> isolated rows on a ReLU kink: not counted")
    bad += flag
    for k, v in e.items(): worst[k] = max(worst[k], v if np.isfinite(v) else 1e9)
    if flag or it % 20 == 0:
        print("case %3d L=%d d=%2d c=%2d hidden=%-14s %s n=%4d %s %s" % (it, L, d, c, hidden, act, n, " ".join("%s=%.1e" % kv for kv in e.items()), "<-- BAD" if flag else ""))
print("cases %d, beyond tolerance %d, worst %s" % (cases, bad, {k: "%.1e" % v for k, v in worst.items()}))
sys.exit(1 if bad else 0)

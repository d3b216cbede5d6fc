import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
for r in range(a.shape[0]):
    fa, fb = torch.isfinite(a[r]), torch.isfinite(b[r])
    ok = fa & fb
    d = (a[r][ok] - b[r][ok]).abs()
    print("rep", r, "nonfinite a/b:", int((~fa).sum()), int((~fb).sum()), "max abs diff", float(d.max()), "at", int(d.argmax()), "scale", float(b[r][ok].abs().max()),
          "cos", float((a[r][ok] * b[r][ok]).sum() / (a[r][ok].norm() * b[r][ok].norm())))
print("a rep0 vs rep1 max diff", float((a[0] - a[1]).abs().max()), " b rep0 vs rep1", float((b[0] - b[1]).abs().max()))

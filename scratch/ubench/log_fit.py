import numpy as np
ld = np.longdouble
zmax = ((np.sqrt(ld(2)) - 1) / (np.sqrt(ld(2)) + 1))**2 * ld(1.001)
def g(z):
    acc = ld(0) * z
    for k in range(40, 0, -1):
        acc = acc * z + ld(2) / ld(2*k + 1)
    return acc
def solve(A, b):
    A = A.copy(); b = b.copy(); n = len(b)
    for i in range(n):
        p = i + int(np.argmax(np.abs(A[i:, i])))
        A[[i, p]] = A[[p, i]]; b[[i, p]] = b[[p, i]]
        for r in range(i + 1, n):
            f = A[r, i] / A[i, i]; A[r] -= f * A[i]; b[r] -= f * b[i]
    x = np.zeros(n, dtype=ld)
    for i in range(n - 1, -1, -1):
        x[i] = (b[i] - np.dot(A[i, i + 1:], x[i + 1:])) / A[i, i]
    return x
for ncoef in (6, 7, 8):
    j = np.arange(ncoef, dtype=ld)
    x = np.cos(np.pi * (j + ld(0.5)) / ncoef)
    t = (x + 1) / 2                      # in (0,1)
    z = t * zmax
    A = np.stack([t**k for k in range(ncoef)], 1)
    ct = solve(A, g(z))
    cz = ct / np.array([zmax**k for k in range(ncoef)], dtype=ld)
    coef = np.asarray(cz, dtype=np.float64)
    zz = np.linspace(0, float(zmax), 20001).astype(ld)
    approx = ld(0) * zz
    for k in range(ncoef - 1, -1, -1):
        approx = approx * zz + ld(coef[k])
    err = np.max(np.abs(approx - g(zz)))
    print(ncoef, "max abs err of P:", float(err), " -> rel err of log part ~", float(err * zmax / 2))
    print(", ".join(float(v).hex() for v in coef)); print([repr(float(v)) for v in coef])

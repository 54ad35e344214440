"""Two in-situ GEMM tables (tools/gemm_insitu.py) side by side, per logical operation (row-split pieces of one GEMM summed)."""
import sys
def load(p):
    d = {}
    for l in open(p):
        f = l.split()
        if len(f) < 19 or not f[0].isdigit():
            continue
        M, N, K, ta, tb, act, f32, res, ain, aout, cs, acc, grp = f[:13]
        big = int(M) >= 40000 or (int(M) < 2000 and int(K) == 768 and int(N) == 3072) or int(M) == 6912
        key = ("vid" if (big or ta == "1" and grp == "1") else "grp" if grp != "1" else "one", N, K, ta, tb, act, f32, res)
        d[key] = d.get(key, 0) + float(f[17])
    return d
A, B = load(sys.argv[1]), load(sys.argv[2])
print("op (lane N K tA tB act f32 resid)          A ms/step  B ms/step")
for k in sorted(A, key=lambda k: -A[k]):
    b = B.get(k, float("nan"))
    print(" ".join(k).ljust(40), "%8.2f" % A[k], "%8.2f" % b, "  <-- B" if b < A[k] * 0.97 else ("  <-- A" if b > A[k] * 1.03 else ""))
print("total %.2f %.2f" % (sum(A.values()), sum(B.values())))

"""Error of the f32 expansion |a|^2 + |b|^2 - 2 a.b (accumulated the way v_mfma_f32_32x32x2_f32 accumulates: pairs of products, k ascending)
against the exact sum of (a - b)^2, relative to |a|^2 + |b|^2, on unit-norm ReID-like (non-negative) and centred random vectors.
The matrix-core euclidean path (sa_gemm.hip, visual_cell<EU>) flags a cell for direct recomputation when d^2 < rho (|a|^2 + |b|^2) with
rho = 5e-3 sqrt(D): twice the largest error below over the 2e-5 that 1e-5 on d leaves for d^2.  python scripts/euclid_error_model.py"""
import numpy as np
rng=np.random.default_rng(0)
def sim(D, n=20000, nonneg=True):
    a=rng.standard_normal((n,D)).astype(np.float32); b=rng.standard_normal((n,D)).astype(np.float32)
    if nonneg: a=np.abs(a); b=np.abs(b)
    a/=np.linalg.norm(a,axis=1,keepdims=True); b/=np.linalg.norm(b,axis=1,keepdims=True)
    a=a.astype(np.float32); b=b.astype(np.float32)
    # MFMA 32x32x2: acc += a0*b0 + a1*b1 sequentially over k pairs (products exact-ish in f32, pair add then acc add)
    def seqdot(x,y):
        p=(x*y).astype(np.float32)
        pair=(p[:,0::2]+p[:,1::2]).astype(np.float32)
        acc=np.zeros(len(x),np.float32)
        for k in range(pair.shape[1]): acc=(acc+pair[:,k]).astype(np.float32)
        return acc
    na=seqdot(a,a); nb=seqdot(b,b); dot=seqdot(a,b)
    d2=(na+nb).astype(np.float32)-(np.float32(2)*dot)
    ex=((a.astype(np.float64)-b.astype(np.float64))**2).sum(1)
    err=np.abs(d2.astype(np.float64)-ex)/(na+nb).astype(np.float64)
    ratio=ex/(na+nb)
    return err.max(), np.percentile(err,99.9), err.mean(), ratio.min(), ratio.mean()
for D in (64,128,512,2048,4096):
    for nn in (True,False):
        print(D, nn, ["%.2e"%x for x in sim(D, 20000 if D<=512 else 5000, nn)])

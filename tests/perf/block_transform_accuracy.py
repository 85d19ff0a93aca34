"""Accuracy of the in-place block Gauss transforms of k_chol_rr4 (T <- T + G_k T_k): unit-lower blocks against Cholesky-scaled ones.
With Cholesky scaling, W + I rounds W away where 1 / L_cc is small and T_k - (W + I) T_k cancels; the unit-lower form is exact there."""
import numpy as np, scipy.linalg as sl
rng = np.random.default_rng(1)
def trial(scales, mode):
    n=16
    B = rng.standard_normal((n, 40)); D0 = B@B.T/40 + 0.1*np.eye(n)
    sc = np.array(scales); D = D0*sc[:,None]*sc[None,:]
    L = np.linalg.cholesky(D)
    A = rng.standard_normal((n,16))*sc[:,None]
    Uex = sl.solve_triangular(L, A, lower=True)
    T = -A.copy()
    for k in range(4):
        c0=4*k; blk = slice(c0,c0+4)
        if mode=='chol':
            W = sl.solve_triangular(L[blk,blk], np.eye(4), lower=True)
            G = np.zeros((n,4)); G[blk]=-(W+np.eye(4)); G[c0+4:] = -(L[c0+4:,blk]@W)
        else:
            d = np.diag(L)[blk]; Lt = L[:,blk]/d[None,:]
            W = sl.solve_triangular(Lt[blk], np.eye(4), lower=True, unit_diagonal=True)
            G = np.zeros((n,4)); G[blk]=-(W+np.eye(4)); G[c0+4:] = -(Lt[c0+4:]@W)
        T = T + G@T[blk]
    if mode!='chol': T = T/np.diag(L)[:,None]
    return np.abs(T-Uex).max()/np.abs(Uex).max(), np.abs(L@T - A).max()/np.abs(A).max()
for name, scales in [("uniform", np.ones(16)), ("wide", 10.0**rng.uniform(-3,5,16)), ("descending", 10.0**np.linspace(5,-3,16)), ("ascending", 10.0**np.linspace(-3,5,16))]:
    for mode in ('chol','unit'):
        print(name, mode, trial(scales, mode))

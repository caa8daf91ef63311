"""Lane-level numpy model of the register-resident fused LocalAttention kernels (csrc/attention_reg.hip).

Development aid (no GPU in the build container): the kernels chain v_mfma_f32_16x16x4_f32 results from one product into the
next as operands without leaving registers, which works only if every index map is right.  This script restates the kernels'
data flow with explicit 64-lane arrays -- `mfma16` below is the instruction's lane map (MI355X guide: lane l supplies
A[m = l & 15][k = l >> 4] and B[k = l >> 4][n = l & 15]; accumulator register r of lane l is D[4 * (l >> 4) + r][l & 15]) --
and checks forward and backward against a plain numpy LocalAttention (enhanced_generator.py:13-47 on one 4x4 window).

    python tools/sim_attn_layout.py        # prints max errors, exits non-zero on a mismatch
"""
import sys

import numpy as np

L = np.arange(64)
I, G = L & 15, L >> 4


def mfma16(a, b, c):
    """a, b: (64,) per-lane operands; c: (64, 4) accumulator.  Returns c + A @ B in the accumulator layout."""
    A = np.zeros((16, 4)); A[I, G] = a
    B = np.zeros((4, 16)); B[G, I] = b
    D = A @ B
    out = c.copy()
    for r in range(4):
        out[:, r] += D[4 * G + r, I]
    return out


def zeros():
    return np.zeros((64, 4))


# A "tile" in layout L(a|b): rows a on (fragment, g, r), columns b on (fragment, lane i).  regs[fa][fb] is a (64, 4) array with
# regs[fa][fb][l][r] = M[16 fa + 4 g + r][16 fb + i].
def to_tiles(M):
    na, nb = M.shape[0] // 16, M.shape[1] // 16
    return [[np.stack([M[16 * fa + 4 * G + r, 16 * fb + I] for r in range(4)], axis=1) for fb in range(nb)] for fa in range(na)]


def from_tiles(t):
    na, nb = len(t), len(t[0])
    M = np.zeros((16 * na, 16 * nb))
    for fa in range(na):
        for fb in range(nb):
            for r in range(4):
                M[16 * fa + 4 * G + r, 16 * fb + I] = t[fa][fb][:, r]
    return M


def contract(T, U):
    """T in L(a|b), U in L(a|c)  ->  L(b|c):  out[b][c] = sum_a T[a][b] U[a][c]  (T supplies A operands, U supplies B operands)."""
    na, nb, nc = len(T), len(T[0]), len(U[0])
    out = [[zeros() for _ in range(nc)] for _ in range(nb)]
    for fb in range(nb):
        for fc in range(nc):
            for fa in range(na):
                for r in range(4):
                    out[fb][fc] = mfma16(T[fa][fb][:, r], U[fa][fc][:, r], out[fb][fc])
    return out


def row16_sum(v):
    """sum over the 16 lanes sharing lane >> 4 (DPP row reduction), result in every lane"""
    out = np.zeros_like(v)
    for g in range(4):
        out[G == g] = v[G == g].sum(axis=0)
    return out


def xg_sum(v):
    """sum over the 4 lanes sharing lane & 15 (permlane16/32 swap + add, twice), result in every lane"""
    out = np.zeros_like(v)
    for i in range(16):
        out[I == i] = v[I == i].sum(axis=0)
    return out


def lds_transpose(T):
    """L(a|b) -> L(b|a) through an LDS image (modelled as the dense matrix)."""
    return to_tiles(from_tiles(T).T)


# ---------------------------------------------------------------------------------------------------------------------
def ref_forward(X, Wqkv, bqkv, Wp, bp):
    C = X.shape[1]
    qkv = X @ Wqkv.T + bqkv
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    nq = np.maximum(np.sqrt((q * q).sum(1, keepdims=True)), 1e-12)
    nk = np.maximum(np.sqrt((k * k).sum(1, keepdims=True)), 1e-12)
    qh, kh = q / nq, k / nk
    S = qh.T @ kh                                  # [c1][c2]
    E = np.exp(S - S.max(1, keepdims=True))
    P = E / E.sum(1, keepdims=True)
    O = (P @ v.T).T                                # [p][c1]
    Y = O @ Wp.T + bp
    return Y, dict(q=q, k=k, v=v, nq=nq, nk=nk, qh=qh, kh=kh, P=P, O=O)


def ref_backward(X, Wqkv, bqkv, Wp, bp, dY):
    C = X.shape[1]
    Y, t = ref_forward(X, Wqkv, bqkv, Wp, bp)
    dO = dY @ Wp
    dWp = dY.T @ t["O"]
    dbp = dY.sum(0)
    dP = dO.T @ t["v"]                             # [c1][c2]
    dV = dO @ t["P"]                               # [p][c2]
    dS = t["P"] * (dP - (dP * t["P"]).sum(1, keepdims=True))
    dqh = t["kh"] @ dS.T                           # [p][c1]
    dkh = t["qh"] @ dS                             # [p][c2]
    dq = (dqh - t["qh"] * (t["qh"] * dqh).sum(1, keepdims=True)) / t["nq"]
    dk = (dkh - t["kh"] * (t["kh"] * dkh).sum(1, keepdims=True)) / t["nk"]
    dQKV = np.concatenate([dq, dk, dV], 1)
    dX = dQKV @ Wqkv
    dW = dQKV.T @ X
    db = dQKV.sum(0)
    return dX, dW, db, dWp, dbp


# ---------------------------------------------------------------------------------------------------------------------
def x_frag(X):
    """What a lane holds after its 16-byte global loads: lane (i = pixel, g) has X[p = i][16 h + 4 g + e], e = 0..3, h < C/16:
    the tile L(ci|p) with the K order (h, g, e) the MFMA chain uses."""
    return to_tiles(X.T)  # rows ci = 16 h + 4 g + e, cols p = i   (one column fragment)


def sim_forward(X, Wqkv, bqkv, Wp, bp, keep=None):
    C = X.shape[1]
    NF = C // 16
    Xt = x_frag(X)                                             # L(ci|p)
    Wt = to_tiles(Wqkv.T)                                      # L(ci|j): lane i = j, regs ci: 16-byte loads of Wqkv[j][16h+4g..]
    Wq = [[Wt[h][f] for f in range(0, 2 * NF)] for h in range(NF)]
    Wv = [[Wt[h][f] for f in range(2 * NF, 3 * NF)] for h in range(NF)]
    qk = contract(Xt, Wq)                                      # L(p|j): rows p, cols j (q | k)
    for f in range(2 * NF):
        qk[0][f] = qk[0][f] + bqkv[16 * f + I][:, None]        # bias per lane (column j)
    vt = contract(Wv, Xt)                                      # L(c|p)
    for f in range(NF):
        vt[f][0] = vt[f][0] + np.stack([bqkv[2 * C + 16 * f + 4 * G + r] for r in range(4)], 1)
    q, k = [qk[0][f] for f in range(NF)], [qk[0][NF + f] for f in range(NF)]
    sq = row16_sum(sum(t * t for t in q))                      # (64, 4): per row (g, r)
    sk = row16_sum(sum(t * t for t in k))
    iq, ik = 1.0 / np.maximum(np.sqrt(sq), 1e-12), 1.0 / np.maximum(np.sqrt(sk), 1e-12)
    qh, kh = [[t * iq for t in q]], [[t * ik for t in k]]      # L(p|c)
    st = contract(kh, qh)                                      # S^T in L(c2|c1)
    e = [[np.exp(st[m][n]) for n in range(NF)] for m in range(NF)]
    pt = [[None] * NF for _ in range(NF)]
    for n in range(NF):
        z = xg_sum(sum(e[m][n].sum(1) for m in range(NF)))     # per column c1 (lane i, fragment n)
        for m in range(NF):
            pt[m][n] = e[m][n] / z[:, None]
    ot = contract(pt, vt)                                      # L(c2|c1) x L(c2|p) -> O^T in L(c1|p)
    Wpt = to_tiles(Wp.T)                                       # L(c1|co): lane i = co, regs c1: 16-byte loads of Wp[co][16n+4g..]
    yt = contract(Wpt, ot)                                     # L(co|p)
    for f in range(NF):
        yt[f][0] = yt[f][0] + np.stack([bp[16 * f + 4 * G + r] for r in range(4)], 1)
    if keep is not None:
        keep.update(Xt=Xt, Wt=Wt, q=q, k=k, iq=iq, ik=ik, qh=qh, kh=kh, vt=vt, pt=pt, ot=ot)
    return from_tiles(yt).T                                    # lane (i = p, g) stores 4 consecutive channels: [p][co]


def sim_backward(X, Wqkv, bqkv, Wp, bp, dY):
    C = X.shape[1]
    NF = C // 16
    kp = {}
    sim_forward(X, Wqkv, bqkv, Wp, bp, kp)
    Xt, qh, kh, vt, pt, iq, ik = kp["Xt"], kp["qh"], kp["kh"], kp["vt"], kp["pt"], kp["iq"], kp["ik"]
    # O in L(p|c) for dWp: contract over c2 of V L(c2|p) and P^T L(c2|c1)
    o_pc = contract(vt, pt)                                    # L(p|c1)
    dYt = x_frag(dY)                                           # L(co|p) (16-byte loads)
    dY_pc = to_tiles(dY)                                       # L(p|co) (4-byte loads: lane i = co, regs p = 4g + r)
    WpT = to_tiles(Wp)                                         # L(co|c): lane i = c, regs co = Wp[16f+4g+r][c]
    dO = contract(dYt, WpT)                                    # L(p|c1)
    dWp = contract(dY_pc, o_pc)                                # L(co|c)
    dbp = np.zeros(C)
    for f in range(NF):
        col = xg_sum(dY_pc[0][f].sum(1))
        dbp[16 * f + I] = col
    v_pc = lds_transpose(vt)                                   # L(p|c2)
    dPt = contract(v_pc, dO)                                   # L(c2|c1)
    dSt = [[None] * NF for _ in range(NF)]
    for n in range(NF):
        d = xg_sum(sum((dPt[m][n] * pt[m][n]).sum(1) for m in range(NF)))
        for m in range(NF):
            dSt[m][n] = pt[m][n] * (dPt[m][n] - d[:, None])
    p_c1c2 = lds_transpose(pt)                                 # P in L(c1|c2)
    dOt = lds_transpose(dO)                                    # L(c1|p)
    dV = contract(p_c1c2, dOt)                                 # L(c2|p)
    # q, k (un-normalised) into L(c|p); norms again in that orientation (registers + cross-g)
    q_cp, k_cp = lds_transpose([kp["q"]]), lds_transpose([kp["k"]])
    sq = xg_sum(sum((t[0] * t[0]).sum(1) for t in q_cp))
    sk = xg_sum(sum((t[0] * t[0]).sum(1) for t in k_cp))
    iq2, ik2 = 1.0 / np.maximum(np.sqrt(sq), 1e-12), 1.0 / np.maximum(np.sqrt(sk), 1e-12)   # per lane (pixel i)
    qh_cp = [[t[0] * iq2[:, None]] for t in q_cp]
    kh_cp = [[t[0] * ik2[:, None]] for t in k_cp]
    dqh = contract(dSt, kh_cp)                                 # L(c2|c1) x L(c2|p) -> L(c1|p)
    dS = lds_transpose(dSt)                                    # L(c1|c2)
    dkh = contract(dS, qh_cp)                                  # L(c2|p)
    dotq = xg_sum(sum((qh_cp[f][0] * dqh[f][0]).sum(1) for f in range(NF)))
    dotk = xg_sum(sum((kh_cp[f][0] * dkh[f][0]).sum(1) for f in range(NF)))
    dq = [[(dqh[f][0] - qh_cp[f][0] * dotq[:, None]) * iq2[:, None]] for f in range(NF)]
    dk = [[(dkh[f][0] - kh_cp[f][0] * dotk[:, None]) * ik2[:, None]] for f in range(NF)]
    dqkv_jp = dq + dk + [[dV[f][0]] for f in range(NF)]        # L(j|p), 3 NF row fragments
    W2 = to_tiles(Wqkv)                                        # L(j|ci): lane i = ci, regs j
    dXt = contract(W2, dqkv_jp)                                # L(ci|p)
    dqkv_pj = lds_transpose(dqkv_jp)                           # L(p|j)
    X_pc = to_tiles(X)                                         # L(p|ci) (4-byte loads)
    dW = contract(dqkv_pj, X_pc)                               # L(j|ci)
    db = np.zeros(3 * C)
    for f in range(3 * NF):
        db[16 * f + I] = xg_sum(dqkv_pj[0][f].sum(1))
    return from_tiles(dXt).T, from_tiles(dW), db, from_tiles(dWp), dbp


def main():
    rng = np.random.default_rng(0)
    worst = 0.0
    for C in (16, 32):
        X = rng.standard_normal((16, C))
        Wqkv, bqkv = rng.standard_normal((3 * C, C)) / np.sqrt(C), rng.standard_normal(3 * C) * 0.1
        Wp, bp = rng.standard_normal((C, C)) / np.sqrt(C), rng.standard_normal(C) * 0.1
        dY = rng.standard_normal((16, C))
        Y, _ = ref_forward(X, Wqkv, bqkv, Wp, bp)
        err = np.abs(sim_forward(X, Wqkv, bqkv, Wp, bp) - Y).max()
        print(f"C={C} forward max err {err:.2e}")
        worst = max(worst, err)
        for name, a, b in zip(("dX", "dWqkv", "dbqkv", "dWp", "dbp"), sim_backward(X, Wqkv, bqkv, Wp, bp, dY),
                              ref_backward(X, Wqkv, bqkv, Wp, bp, dY)):
            err = np.abs(a - b).max()
            print(f"C={C} backward {name:6s} max err {err:.2e}")
            worst = max(worst, err)
    if worst > 1e-10:
        sys.exit("layout mismatch")
    print("ok")


if __name__ == "__main__":
    main()

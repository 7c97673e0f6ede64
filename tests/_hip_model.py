"""numpy model of the formulas used by sfft_amd/csrc/sfft_amd.hip (design validation only).

Each function mirrors one kernel's index arithmetic so that the algebra (Stockham stage indexing,
Bluestein, two-rows-per-transform packing, pruned Greek transforms on the half spectrum, fill
symmetries, bordered Cholesky) can be checked on CPU against the oracle.  Not product code and not
the oracle: tests/test_hip_model.py is its only user.
"""
import numpy as np


def ilog2(v):
    l = 0
    while (1 << l) < v:
        l += 1
    return l


def lds_fft(s):
    """Stockham autosort radix-4 (+ one radix-2 stage) exactly as lds_fft(); s: [nb, M] complex."""
    s = s.copy()
    nb, M = s.shape
    logM = ilog2(M)
    tw = np.exp(-2j * np.pi * np.arange(M) / M)
    p, logp = 1, 0
    if logM & 1:
        T = M >> 1
        i = np.arange(T)
        u0, u1 = s[:, i].copy(), s[:, i + T].copy()
        s[:, 2 * i] = u0 + u1
        s[:, 2 * i + 1] = u0 - u1
        p, logp = 2, 1
    T = M >> 2
    while p < M:
        tshift = logM - logp - 2
        i = np.arange(T)
        k = i & (p - 1)
        u0, u1, u2, u3 = s[:, i].copy(), s[:, i + T].copy(), s[:, i + 2 * T].copy(), s[:, i + 3 * T].copy()
        if p > 1:
            q = k << tshift
            u1 = u1 * tw[q]; u2 = u2 * tw[2 * q]; u3 = u3 * tw[3 * q]
        a02, s02, a13, s13 = u0 + u2, u0 - u2, u1 + u3, u1 - u3
        y0, y2 = a02 + a13, a02 - a13
        y1 = s02 - 1j * s13
        y3 = s02 + 1j * s13
        j = ((i - k) << 2) + k
        s[:, j] = y0; s[:, j + p] = y1; s[:, j + 2 * p] = y2; s[:, j + 3 * p] = y3
        p <<= 2; logp += 2
    return s


class Axis:
    def __init__(self, N):
        self.N = N
        if N & (N - 1) == 0:
            self.M, self.blue = N, False
        else:
            M = 1
            while M < 2 * N - 1:
                M <<= 1
            self.M, self.blue = M, True
            k = np.arange(N)
            q = (k * k) % (2 * N)
            self.chirp = np.exp(-1j * np.pi * q / N)
            f = np.zeros(M, complex)
            f[:N] = np.conj(self.chirp)
            f[M - k[1:]] = np.conj(self.chirp[1:])
            self.bf = np.fft.fft(f) / M
        self.root = np.exp(-2j * np.pi * np.arange(N) / N)


def lds_dft(s, ax):
    """s: [nb, M] with zeros beyond N; returns [nb, M] whose first N entries are the DFT."""
    if not ax.blue:
        return lds_fft(s)
    s = s.copy()
    s[:, :ax.N] *= ax.chirp
    s = lds_fft(s)
    s = np.conj(s * ax.bf)
    s = lds_fft(s)
    s[:, :ax.N] = ax.chirp * np.conj(s[:, :ax.N])
    return s


def forward_plane(img, ei, ej, ax0, ax1):
    """rows_r2c + cols_c2c: SCALE * DFT2(img * cx^ei * cy^ej) on the half spectrum [N0][Nh]."""
    N0, N1 = img.shape
    Nh = N1 // 2 + 1
    scale = 1.0 / (N0 * N1)
    cx = (np.arange(N0) + 1.0) / N0
    cy = (np.arange(N1) + 1.0) / N1
    out = np.zeros((N0, Nh), complex)
    for l0 in range(0, N0, 2):
        l1 = l0 + 1
        s = np.zeros((1, ax1.M), complex)
        v0 = img[l0] * (cx[l0] ** ei * cy ** ej)
        v1 = img[l1] * (cx[l1] ** ei * cy ** ej) if l1 < N0 else 0.0
        s[0, :N1] = v0 + 1j * v1
        s = lds_dft(s, ax1)[0]
        m = np.arange(Nh)
        z = s[m]
        zc = np.conj(s[np.where(m == 0, 0, N1 - m)])
        out[l0] = 0.5 * scale * (z + zc)
        if l1 < N0:
            d = z - zc
            out[l1] = 0.5 * scale * (d.imag - 1j * d.real)
    # columns
    s = np.zeros((Nh, ax0.M), complex)
    s[:, :N0] = out.T
    s = lds_dft(s, ax0)
    return s[:, :N0].T.copy()


def inverse_diff(FD, J, bpq, ref_pq, ax0, ax1):
    """cols_c2c(inverse) + rows_c2r_diff."""
    N0, N1 = J.shape
    Nh = N1 // 2 + 1
    s = np.zeros((Nh, ax0.M), complex)
    s[:, :N0] = np.conj(FD.T)
    s = lds_dft(s, ax0)
    Y = np.conj(s[:, :N0]).T          # [N0][Nh]
    cx = (np.arange(N0) + 1.0) / N0
    cy = (np.arange(N1) + 1.0) / N1
    DIFF = np.zeros((N0, N1))
    even = N1 % 2 == 0
    for l0 in range(0, N0, 2):
        l1 = l0 + 1
        has1 = l1 < N0
        m = np.arange(N1)
        mir = m >= Nh
        mm = np.where(mir, N1 - m, m)
        x0 = Y[l0][mm].copy()
        x1 = Y[l1][mm].copy() if has1 else np.zeros(N1, complex)
        selfc = (mm == 0) | (even & (mm == N1 // 2))
        x0[selfc] = x0[selfc].real
        x1[selfc] = x1[selfc].real
        x0 = np.where(mir, np.conj(x0), x0)
        x1 = np.where(mir, np.conj(x1), x1)
        Z = x0 + 1j * x1
        s = np.zeros((1, ax1.M), complex)
        s[0, :N1] = np.conj(Z)
        z = lds_dft(s, ax1)[0][:N1]
        B0 = sum(b * cx[l0] ** p * cy ** q for b, (p, q) in zip(bpq, ref_pq))
        DIFF[l0] = J[l0] - B0 - z.real
        if has1:
            B1 = sum(b * cx[l1] ** p * cy ** q for b, (p, q) in zip(bpq, ref_pq))
            DIFF[l1] = J[l1] - B1 + z.imag
    return DIFF


def greek_patch(A, B, h, scale, ax0, ax1, S=2):
    """greek_g1 + greek_g2 for one pair: patch[r+h][e+h] = scale * Re DFT2(A conj B)[r, e]."""
    N0, Nh = A.shape
    N1 = ax1.N
    rpc = (N0 + S - 1) // S
    G = np.zeros((2 * h + 1, Nh), complex)
    for c in range(S):
        lb, le = c * rpc, min(N0, (c + 1) * rpc)
        H = A[lb:le] * np.conj(B[lb:le])
        ls = np.arange(lb, le)
        G[h] += H.sum(axis=0)
        for r in range(1, h + 1):
            w = ax0.root[(ls * r) % N0][:, None]
            S1 = (H.real * w.real).sum(0); S2 = (H.imag * w.imag).sum(0)
            S3 = (H.real * w.imag).sum(0); S4 = (H.imag * w.real).sum(0)
            G[h + r] += (S1 - S2) + 1j * (S3 + S4)
            G[h - r] += (S1 + S2) + 1j * (S4 - S3)
    m = np.arange(Nh)
    wgt = np.where((m == 0) | ((N1 % 2 == 0) & (m == N1 // 2)), 1.0, 2.0)
    out = np.zeros((2 * h + 1, 2 * h + 1))
    for r in range(2 * h + 1):
        gx, gy = wgt * G[r].real, wgt * G[r].imag
        out[r, h] = scale * gx.sum()
        for e in range(1, h + 1):
            w = ax1.root[(m * e) % N1]
            U, V = (gx * w.real).sum(), (gy * w.imag).sum()
            out[r, h + e] = scale * (U - V)
            out[r, h - e] = scale * (U + V)
    return out


def poly_axis_dft(N, e, nout):
    v = ((np.arange(N) + 1.0) / N) ** e
    return np.fft.fft(v)[:nout]


def build_system(mI, mJ, p, T):
    """Patches -> bordered system exactly as sys_element(); returns full LHMAT, RHb."""
    N0, N1 = p['N0'], p['N1']
    w, Fij, Fpq, Fab, Fijab, NEQ, L = p['w0'], p['Fij'], p['Fpq'], p['Fab'], p['Fijab'], p['NEQ'], p['L1']
    scale = 1.0 / (N0 * N1)
    ax0, ax1 = Axis(N0), Axis(N1)
    Nh = N1 // 2 + 1
    spec = [forward_plane(mI, i, j, ax0, ax1) for (i, j) in T['REF_ij']]
    spec.append(forward_plane(mJ, 0, 0, ax0, ax1))
    hO, hG = 2 * w, w
    omg = {}
    for a in range(Fij):
        for b in range(a, Fij):
            omg[(a, b)] = greek_patch(spec[a], spec[b], hO, scale * scale, ax0, ax1)
    gam = {}
    for a in range(Fij):
        for q, (pp, qq) in enumerate(T['REF_pq']):
            FT = scale * np.outer(poly_axis_dft(N0, pp, N0), poly_axis_dft(N1, qq, Nh))
            gam[(a, q)] = greek_patch(spec[a], FT, hG, scale, ax0, ax1)
    the = [greek_patch(spec[a], spec[Fij], hG, scale, ax0, ax1) for a in range(Fij)]
    cx = (np.arange(N0) + 1.0) / N0
    cy = (np.arange(N1) + 1.0) / N1
    rowmom = np.stack([(mJ * cy ** q).sum(axis=1) for q in range(4)], axis=1)
    delta = [scale * (cx ** pp * rowmom[:, qq]).sum() for (pp, qq) in T['REF_pq']]
    Sx = [np.sum(cx ** e) for e in range(7)]
    Sy = [np.sum(cy ** e) for e in range(7)]
    phi = np.array([[scale * Sx[a[0] + b[0]] * Sy[a[1] + b[1]] for b in T['REF_pq']] for a in T['REF_pq']])

    def omg_at(i8, ij, r0, r1):
        if i8 > ij:
            return omg[(ij, i8)][-r0 + hO, -r1 + hO]
        return omg[(i8, ij)][r0 + hO, r1 + hO]

    def elem(R, C):
        if C == NEQ:
            if R < Fijab:
                i8, ab8 = divmod(R, Fab)
                a8, b8 = ab8 // L - w, ab8 % L - w
                t0 = the[i8][hG, hG]
                return t0 if (a8 == 0 and b8 == 0) else the[i8][a8 + hG, b8 + hG] - t0
            return delta[R - Fijab]
        if R < Fijab and C < Fijab:
            i8, ab8 = divmod(R, Fab)
            ij, ab = divmod(C, Fab)
            a8, b8, a, b = ab8 // L - w, ab8 % L - w, ab // L - w, ab % L - w
            c8, c = (a8 == 0 and b8 == 0), (a == 0 and b == 0)
            o00 = omg_at(i8, ij, 0, 0)
            if c8 and c:
                return o00
            if c8:
                return omg_at(i8, ij, -a, -b) - o00
            if c:
                return omg_at(i8, ij, a8, b8) - o00
            return -omg_at(i8, ij, a8, b8) - omg_at(i8, ij, -a, -b) + omg_at(i8, ij, a8 - a, b8 - b) + o00
        if R < Fijab:
            pq = C - Fijab
            i8, ab8 = divmod(R, Fab)
            a8, b8 = ab8 // L - w, ab8 % L - w
            G = gam[(i8, pq)]
            return G[hG, hG] if (a8 == 0 and b8 == 0) else G[a8 + hG, b8 + hG] - G[hG, hG]
        if C < Fijab:
            pq = R - Fijab
            ij, ab = divmod(C, Fab)
            a, b = ab // L - w, ab % L - w
            G = gam[(ij, pq)]
            return G[hG, hG] if (a == 0 and b == 0) else G[a + hG, b + hG] - G[hG, hG]
        return phi[R - Fijab, C - Fijab]

    LH = np.array([[elem(R, C) for C in range(NEQ)] for R in range(NEQ)])
    rhs = np.array([elem(R, NEQ) for R in range(NEQ)])
    return LH, rhs


def bordered_cholesky_solve(LH, rhs, CB=8):
    """chol_panel / chol_update / chol_backsolve on the bordered lower triangle."""
    n = LH.shape[0]
    A = np.zeros((n + 1, n + 1))
    A[:n, :n] = LH
    A[n, :n] = rhs
    for k in range(0, n, CB):
        nb = min(CB, n - k)
        D = A[k:k + nb, k:k + nb].copy()
        for j in range(nb):
            for i in range(j, nb):
                D[i, j] -= np.dot(D[i, :j], D[j, :j])
            rj = np.sqrt(D[j, j])
            D[j + 1:, j] /= rj
            D[j, j] = rj
        A[k:k + nb, k:k + nb] = np.tril(D)
        for i in range(k + nb, n + 1):
            for j in range(nb):
                A[i, k + j] = (A[i, k + j] - np.dot(A[i, k:k + j], D[j, :j])) / D[j, j]
        Lp = A[k + nb:, k:k + nb]
        upd = Lp @ Lp[:n - (k + nb)].T
        for i in range(k + nb, n + 1):
            jmax = min(i, n - 1)
            A[i, k + nb:jmax + 1] -= upd[i - (k + nb), :jmax + 1 - (k + nb)]
    y = A[n, :n].copy()
    for kb in range(((n - 1) // CB) * CB, -1, -CB):
        nb = min(CB, n - kb)
        for j in range(nb - 1, -1, -1):
            y[kb + j] /= A[kb + j, kb + j]
            y[kb:kb + j] -= A[kb + j, kb:kb + j] * y[kb + j]
        y[:kb] -= A[kb:kb + nb, :kb].T @ y[kb:kb + nb]
    return y


def construct_fd(specI, sol, p, ax0, ax1):
    """kernel_ctab + construct_fd on the half spectrum."""
    N0, N1 = p['N0'], p['N1']
    w, Fij, Fab, L = p['w0'], p['Fij'], p['Fab'], p['L1']
    Nh = N1 // 2 + 1
    scale = 1.0 / (N0 * N1)
    m = np.arange(Nh)
    l = np.arange(N0)
    FD = np.zeros((N0, Nh), complex)
    cen = w * L + w
    for ij in range(Fij):
        a = sol[ij * Fab:(ij + 1) * Fab]
        soff = a.sum() - a[cen]
        A2 = a.reshape(L, L)
        C = np.zeros((L, Nh), complex)
        for aa in range(L):
            for bb in range(L):
                C[aa] += A2[aa, bb] * ax1.root[(m * (bb - w)) % N1]
        K = np.zeros((N0, Nh), complex)
        for aa in range(L):
            K += ax0.root[(l * (aa - w)) % N0][:, None] * C[aa][None, :]
        FD += specI[ij] * (scale * (K - soff))
    return FD


# ------------------------------------------------------------------------------------------------
# Round 6: index arithmetic of the pair-per-workgroup column pass (cols_fwd_weighted_4096_z) and of the 2048-point
# network of scripts/micro/fft_h2048.hpp
# ------------------------------------------------------------------------------------------------
def pair_major_offset(l, m, pstride):
    """Element (row l, spectrum column m) of a stage plane with pair-major lines: 4-column panels, every 128-byte line (rows 2p, 2p + 1 x
    columns 0..3) stored [column pair h][row parity][column c]."""
    return (m >> 2) * pstride + (l >> 1) * 8 + ((m >> 1) & 1) * 4 + (l & 1) * 2 + (m & 1)


def rows_pm_store_offsets(l0, pstride):
    """What rows_r2c_4096 (pm = 1) stores where for the row pair (l0, l0 + 1): {offset: (row, column)}.  Thread j, step sx holds X0[m], X1[m]
    (rows l0, l0 + 1) of column m = j + 256 sx; lanes 0, 1 of a quad send X1 and receive X0 from lanes 2, 3; instruction A stores at
    line + (j & 3), instruction B at line + 4 + (j & 3)."""
    out = {}
    for sx in range(9):
        for j in range(256 if sx < 8 else 4):
            m = j + 256 * sx
            t = j & 3
            even = t < 2
            partner_m = m ^ 2                       # lane ^ 2 holds column m ^ 2
            base = (l0 >> 1) * 8 + t + (m >> 2) * pstride
            # A: even ? X0[m] : got = X1 of the partner (the partner is even and sent X1);  B: even ? got = X0 of the partner : X1[m]
            a_val = (l0, m) if even else (l0 + 1, partner_m)
            b_val = (l0, partner_m) if even else (l0 + 1, m)
            assert base not in out and base + 4 not in out
            out[base] = a_val
            out[base + 4] = b_val
    return out


def cols_z_load_offset(tid, r, cp, pstride):
    """Element offset of the r-th load of lane tid = 2 j + c in cols_fwd_weighted_4096_z for column pair cp, and the (row, column) it must be."""
    off = (cp >> 1) * pstride + (cp & 1) * 4 + (tid >> 2) * 8 + (tid & 3) + 1024 * r
    return off, ((tid >> 1) + 256 * r, 2 * cp + (tid & 1))


def h2048_forward(z):
    """The 2048-point network of fft_h2048.hpp (fft2048_fwd), thread by thread: returns (A, B, C, Cp) with A[s4][j] = Z[C + 512 s4], B = Z[Cp + 512 s4]."""
    tw = np.exp(-2j * np.pi * np.arange(4096) / 4096)
    j = np.arange(256)
    u = np.fft.fft(np.stack([z[j + 256 * r] for r in range(8)]), axis=0)
    for s in range(8):
        u[s] = u[s] * tw[2 * j * s]
    lds = np.zeros(2304, complex)
    for s in range(8):
        lds[256 * s + j] = u[s]
    a, s1 = j & 31, j >> 5
    u = np.fft.fft(np.stack([lds[256 * s1 + a + 32 * b] for b in range(8)]), axis=0)
    for s in range(8):
        u[s] = u[s] * tw[16 * a * s]
    lds[:] = 0
    for s2 in range(8):
        lds[a + 36 * (s1 + 8 * s2)] = u[s2]
    c, Q = j & 3, j >> 2
    u = np.fft.fft(np.stack([lds[c + 4 * d + 36 * Q] for d in range(8)]), axis=0)
    for s in range(8):
        u[s] = u[s] * tw[128 * c * s]
    lds[:] = 0
    for s3 in range(8):
        lds[514 * c + Q + 64 * s3] = u[s3]
    C, Cp = j, np.where(j > 0, 512 - j, 256)
    A = np.fft.fft(np.stack([lds[514 * cc + C] for cc in range(4)]), axis=0)
    B = np.fft.fft(np.stack([lds[514 * cc + Cp] for cc in range(4)]), axis=0)
    return A, B, C, Cp


def h2048_untangle(Zk, Zp, wk, h=0.5):
    """untangle2 of fft_h2048.hpp: (X[k], X[2048 - k]) from Z[k], Z[2048 - k], w = exp(-2 pi i k / 4096)."""
    S, D = Zk + np.conj(Zp), Zk - np.conj(Zp)
    wd = wk * D
    return h * ((S.real + wd.imag) + 1j * (S.imag - wd.real)), h * ((S.real - wd.imag) + 1j * (-S.imag - wd.real))

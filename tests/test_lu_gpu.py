"""GPU: the blocked LU with partial pivoting (sfft_amd/csrc/lu.hpp) -- the reference's solver semantics
(np.linalg.solve / cupy.linalg.solve = getrf + getrs, sfft/sfftcore/SFFTSubtract.py:15-23, 398-403, 743-747) -- on its own, through
the C ABI's sfft_dbg_solve_dense, against numpy.linalg.solve (LAPACK's pivoted LU) on seeded general matrices, and by residual at
the sizes numpy needs minutes for.  Tolerances: relative solution error <= 1e-9 on matrices whose condition number is <= ~1e5
(eps * cond * growth), residual ||A x - b|| <= 1e-12 ||A|| ||x|| everywhere (backward stability, size independent)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


_PLANS = {}


def _plan(dev, w, DK, DB, cpr=False):
    """a plan is only the solver's workspace here: n = SOLVER_N follows from the kernel geometry"""
    from sfft_amd.plan import Plan
    key = (w, DK, DB, cpr)
    if key not in _PLANS:
        side = max(64, 4 * (2 * w + 1))
        _PLANS[key] = Plan(side, side, w, DK, DB, cpr, device=dev.index)
    return _PLANS[key]


def _rel(x, y):
    return float(np.max(np.abs(x - y)) / np.max(np.abs(y)))


# (w, DK, DB, ConstPhotRatio) -> n:  10, 156, 300, 820, 1735 (config 2's system), 2900; each panel shape of lu_panel<W, R> up to 3072 rows
GEOMS = [(1, 0, 0, False), (2, 2, 2, False), (3, 2, 2, False), (4, 3, 3, False), (8, 2, 2, True), (8, 3, 3, False)]


@pytest.mark.parametrize("geom", GEOMS, ids=lambda g: "w%d_k%d_b%d%s" % (g[0], g[1], g[2], "_cpr" if g[3] else ""))
def test_general_matrices_match_numpy(dev, geom):
    plan = _plan(dev, *geom)
    n = plan.query("SOLVER_N")
    rng = np.random.default_rng(100 + n)
    for trial in range(3):
        A = rng.standard_normal((n, n))
        if trial == 1:          # rows of wildly different scale: the pivot order is far from the identity
            A *= 10.0 ** rng.uniform(-3, 3, size=(n, 1))
        if trial == 2:          # a zero diagonal: nothing works without row exchanges
            np.fill_diagonal(A, 0.0)
        b = rng.standard_normal(n)
        x = plan.solve_dense(torch.from_numpy(A).to(dev), torch.from_numpy(b).to(dev), use_lu=True).cpu().numpy()
        x_np = np.linalg.solve(A, b)
        resid = np.max(np.abs(A @ x - b)) / (np.max(np.abs(A)) * np.max(np.abs(x)) * n)
        assert resid <= 1e-14, (trial, resid)
        resid_np = np.max(np.abs(A @ x_np - b)) / (np.max(np.abs(A)) * np.max(np.abs(x_np)) * n)
        assert resid <= 20 * resid_np + 1e-18, (trial, resid, resid_np)           # as backward stable as LAPACK's
        cond = np.linalg.cond(A) if n <= 900 else None
        assert _rel(x, x_np) <= (1e-13 * cond if cond else 1e-8), (trial, _rel(x, x_np), cond)
    assert plan.query("LAST_SOLVER") == 2


def test_exact_arithmetic_cases(dev):
    """Matrices on which every operation is exact: the result must equal the exact solution bit for bit.
    (i) a row permutation of a diagonal of powers of two (one nonzero per column: the pivot search must find it);
    (ii) 2 x 2 anti-diagonal blocks; (iii) a column whose largest entries tie: the FIRST such row is the pivot (idamax), checked
    through a case where choosing a later row still solves the system -- so only exactness is asserted there."""
    plan = _plan(dev, 3, 2, 2, False)
    n = plan.query("SOLVER_N")
    rng = np.random.default_rng(5)
    perm = rng.permutation(n)
    d = 2.0 ** rng.integers(-8, 9, size=n)
    A = np.zeros((n, n))
    A[perm, np.arange(n)] = d                   # column c has its only nonzero in row perm[c]
    xt = rng.integers(-64, 65, size=n).astype(np.float64)
    b = A @ xt
    x = plan.solve_dense(torch.from_numpy(A).to(dev), torch.from_numpy(b).to(dev)).cpu().numpy()
    assert np.array_equal(x, xt)
    A2 = np.zeros((n, n))
    for k in range(0, n - 1, 2):
        A2[k, k + 1] = 2.0
        A2[k + 1, k] = -4.0
    if n % 2:
        A2[n - 1, n - 1] = 1.0
    b2 = A2 @ xt
    x2 = plan.solve_dense(torch.from_numpy(A2).to(dev), torch.from_numpy(b2).to(dev)).cpu().numpy()
    assert np.array_equal(x2, xt)
    A3 = np.eye(n)
    A3[:, 0] = 1.0                              # every entry of column 0 ties at 1: row 0 is the pivot, the update is exact
    b3 = A3 @ xt
    x3 = plan.solve_dense(torch.from_numpy(A3).to(dev), torch.from_numpy(b3).to(dev)).cpu().numpy()
    assert np.array_equal(x3, xt)


def test_singular_matrix_raises_linalgerror(dev):
    plan = _plan(dev, 2, 2, 2, False)
    n = plan.query("SOLVER_N")
    rng = np.random.default_rng(9)
    A = rng.standard_normal((n, n))
    A[:, 17] = 0.0                              # an exactly zero column: no pivot in column 17
    b = rng.standard_normal(n)
    with pytest.raises(np.linalg.LinAlgError):  # numpy.linalg.solve raises the same
        plan.solve_dense(torch.from_numpy(A).to(dev), torch.from_numpy(b).to(dev))
    with pytest.raises(np.linalg.LinAlgError):
        np.linalg.solve(A, b)
    A[:, 17] = rng.standard_normal(n)           # the plan is fine afterwards
    x = plan.solve_dense(torch.from_numpy(A).to(dev), torch.from_numpy(b).to(dev)).cpu().numpy()
    assert _rel(x, np.linalg.solve(A, b)) <= 1e-8
    An = A.copy()
    An[5, 7] = np.nan                           # NaN input: reported, never a hang or a silent number
    try:
        xn = plan.solve_dense(torch.from_numpy(An).to(dev), torch.from_numpy(b).to(dev)).cpu().numpy()
        assert not np.isfinite(xn).all()
    except np.linalg.LinAlgError:
        pass


def test_spd_system_cholesky_and_lu_agree(dev):
    plan = _plan(dev, 8, 2, 2, True)
    n = plan.query("SOLVER_N")
    rng = np.random.default_rng(3)
    G = rng.standard_normal((n, n + 40))
    A = G @ G.T / n + 0.5 * np.eye(n)
    b = rng.standard_normal(n)
    At, bt = torch.from_numpy(A).to(dev), torch.from_numpy(b).to(dev)
    x_lu = plan.solve_dense(At, bt, use_lu=True).cpu().numpy()
    x_ch = plan.solve_dense(At, bt, use_lu=False).cpu().numpy()
    x_np = np.linalg.solve(A, b)
    assert _rel(x_lu, x_np) <= 1e-11 and _rel(x_ch, x_np) <= 1e-11


@pytest.mark.parametrize("geom", [(3, 2, 2, False), (8, 2, 2, True), (9, 3, 3, False)], ids=["n300", "n1735_dataflow", "n3620_outer_blocks"])
def test_cholesky_path_reports_a_non_positive_pivot_wherever_it_is(dev, geom):
    """The Cholesky path checks the pivots of a 64 x 64 diagonal block once per block (a non-positive pivot leaves NaN in the block's
    reciprocal diagonal, solver.hpp chol_factor_diag<true>): an indefinite matrix must come back as LinAlgError from it -- first column,
    inside a block, on block boundaries, in the last (partial) block -- and be solved by the LU; the plan works afterwards."""
    plan = _plan(dev, *geom)
    n = plan.query("SOLVER_N")
    rng = np.random.default_rng(11 + n)
    G = rng.standard_normal((n, n + 30))
    A0 = G @ G.T / n + 0.5 * np.eye(n)
    b = rng.standard_normal(n)
    bt = torch.from_numpy(b).to(dev)
    for k in sorted({0, 37, 63, 64, n // 2, 64 * ((n - 1) // 64), n - 1}):
        A = A0.copy()
        A[k, k] = -A[k, k] - 1.0                # indefinite: the Schur complement's pivot k is negative whatever came before
        At = torch.from_numpy(A).to(dev)
        with pytest.raises(np.linalg.LinAlgError):
            plan.solve_dense(At, bt, use_lu=False)
        assert plan.query("CHOL_STATUS") & 1, (k, plan.query("CHOL_STATUS"))
        if k in (0, n - 1):                     # ... and the reference's solver takes it
            x = plan.solve_dense(At, bt, use_lu=True).cpu().numpy()
            assert np.max(np.abs(A @ x - b)) <= 1e-11 * np.max(np.abs(A)) * np.max(np.abs(x)) * n
    x = plan.solve_dense(torch.from_numpy(A0).to(dev), bt, use_lu=False).cpu().numpy()
    assert plan.query("CHOL_STATUS") == 0
    assert np.max(np.abs(A0 @ x - b)) <= 1e-12 * np.max(np.abs(A0)) * np.max(np.abs(x)) * n


@pytest.mark.parametrize("geom", [(12, 3, 3, True), (16, 3, 3, False), (20, 3, 3, False)], ids=["n6251", "n10900", "n16820"])
def test_large_systems_by_residual(dev, geom):
    """n = 6251 (config 5's system), 10 900 and 16 820 unknowns: panels of 8 / 4 / 2 / 1 column sub-panels with 8 .. 64 rows per thread.
    numpy would need minutes; backward stability is checked on the device with torch.matmul (fp64): ||A x - b|| <= 1e-12 ||A|| ||x||,
    and the solution of a system with a KNOWN solution is recovered to 1e-7."""
    plan = _plan(dev, *geom)
    n = plan.query("SOLVER_N")
    g = torch.Generator(device=dev)
    g.manual_seed(n)
    A = torch.randn((n, n), dtype=torch.float64, device=dev, generator=g)
    A += torch.diag(0.25 * torch.randn(n, dtype=torch.float64, device=dev, generator=g))
    xt = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
    b = A @ xt
    x = plan.solve_dense(A, b, use_lu=True)
    r = (A @ x - b).abs().max() / (A.abs().max() * x.abs().max() * n)
    assert float(r) <= 1e-14, float(r)
    assert float((x - xt).abs().max() / xt.abs().max()) <= 1e-7
    _PLANS.pop(geom, None)
    plan.close()


def test_two_plans_factor_at_the_same_time(dev):
    """Two plans (own hand-off slots, own streams, own host threads) run the multi-workgroup panel side by side: the spinning workgroups of one
    launch must neither starve nor read the other's slots.  Each result must equal the one the same plan gives alone, bit for bit (the pivot
    agreement is deterministic: largest magnitude, lowest workgroup among ties)."""
    import threading
    from sfft_amd.plan import Plan
    geom = (8, 3, 3, False)                      # n = 2900: three multi-workgroup panel shapes per solve
    plans = [Plan(68, 68, *geom, device=dev.index) for _ in range(2)]
    n = plans[0].query("SOLVER_N")
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    As = [torch.randn((n, n), dtype=torch.float64, device=dev, generator=g) for _ in range(2)]
    bs = [torch.randn(n, dtype=torch.float64, device=dev, generator=g) for _ in range(2)]
    alone = [plans[k].solve_dense(As[k], bs[k], use_lu=True).clone() for k in range(2)]
    streams = [torch.cuda.Stream(dev) for _ in range(2)]
    out = [[None] * 6 for _ in range(2)]

    def worker(k):
        torch.cuda.set_device(dev.index)
        with torch.cuda.stream(streams[k]):
            for r in range(6):
                out[k][r] = plans[k].solve_dense(As[k], bs[k], use_lu=True).clone()
    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize(dev)
    for k in range(2):
        r = (As[k] @ alone[k] - bs[k]).abs().max() / (As[k].abs().max() * alone[k].abs().max() * n)
        assert float(r) <= 1e-14
        for rr in range(6):
            assert torch.equal(out[k][rr], alone[k]), (k, rr)
    for p in plans:
        p.close()

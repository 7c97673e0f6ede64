"""BASELINE config 3: B-spline spatially-varying kernel on one 6144 x 6144 pair (KerHW 8, B-spline degree 2 with 2x2
internal knots -> Fij = 25, polynomial background degree 2, constant scaling).  Times one GSS and checks sanity."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from sfft_amd.BSplineSFFT import SingleSFFTConfigure, GeneralSFFTSubtract_PureCupy
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
dev = torch.device('cuda', 0)
rng = np.random.default_rng(3)
x = np.linspace(0, 60 * np.pi, N)[:, None]; y = np.linspace(0, 56 * np.pi, N)[None, :]
base = 80.0 * (np.sin(x) * np.cos(y)) ** 8
REF = base + rng.normal(0, 1.0, (N, N))
gain = 1.0 + 0.0 * x                     # constant photometric ratio
SCI = 1.25 * gain * (0.6 * base + 0.2 * np.roll(base, 1, 0) + 0.2 * np.roll(base, -1, 1)) + 2.0 + rng.normal(0, 1.0, (N, N))
t0 = time.time()
knots = [N / 3 + 0.5, 2 * N / 3 + 0.5]
cfg = SingleSFFTConfigure.SSC(N, N, KerHW=8, KerSpType='B-Spline', KerSpDegree=2, KerIntKnotX=knots, KerIntKnotY=knots,
                              SEPARATE_SCALING=True, ScaSpDegree=0, BkgSpType='Polynomial', BkgSpDegree=2, VERBOSE_LEVEL=0)
torch.cuda.synchronize(); print("plan: Fij=%d NEQ=%d  create %.2f s" % (cfg[0]['Fij'], cfg[0]['NEQ'], time.time() - t0))
R, S = torch.from_numpy(REF).to(dev), torch.from_numpy(SCI).to(dev)
# masked pair: the full pair with the faint background zeroed (distinct tensors, as in a real packet), or the pair itself (argv[2] = same)
same = len(sys.argv) > 2 and sys.argv[2] == 'same'
keep = torch.from_numpy(base > 0.5).to(dev)
mR, mS = (R, S) if same else (torch.where(keep, R, torch.zeros_like(R)).contiguous(), torch.where(keep, S, torch.zeros_like(S)).contiguous())
plan = cfg[1]['plan']
plan.set_timing(True)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    sol, diff, _ = GeneralSFFTSubtract_PureCupy.GSS(R, S, mR, mS, cfg, VERBOSE_LEVEL=0)
    torch.cuda.synchronize(); dt = time.time() - t0
    print("GSS %d: %.1f ms  stages %s solver %d" % (it, dt * 1e3, {k: round(v, 1) for k, v in plan.stage_ms().items()}, plan.query("LAST_SOLVER")))
d = diff.cpu().numpy()
L = 17; ij00 = np.arange(8 * L + 8, 25 * L * L, L * L)
print("rms(DIFF)=%.4f finite=%s  scaling=%.4f (tied: %s)" % (np.sqrt(np.mean(d * d)), np.isfinite(d).all(),
      float(sol[ij00[0]]) / N / N * 1.0, bool((sol[ij00] == sol[ij00[0]]).all())))

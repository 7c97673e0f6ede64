"""Development aid: per-wave start / end stamps of the grouped Omega launch (SFFT_G1_TRACE): how far apart do the sibling groups of a tile run?"""
import os, sys, subprocess
import numpy as np
if len(sys.argv) < 2:
    env = dict(os.environ, SFFT_G1_TRACE="/tmp/g1trace.txt")
    subprocess.run([sys.executable, "scripts/df_trace.py", "child"], env=env, capture_output=True)
    t = np.loadtxt("/tmp/g1trace.txt", dtype=np.uint64)
    t = t[t[:, 0] > 0]
    ng = 7
    t0 = t[:, 0].min()
    st, en, lg = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, t[:, 2].astype(np.int64)
    print("waves %d, launch span %.1f us, wave duration mean %.1f us (min %.1f, max %.1f)" % (len(t), en.max(), (en - st).mean(), (en - st).min(), (en - st).max()))
    tile = lg // ng
    order = np.argsort(lg)
    st, en, tile = st[order], en[order], tile[order]
    spread_s, spread_e = [], []
    for k in range(0, len(st) - ng + 1, ng):
        if tile[k] == tile[k + ng - 1]:
            spread_s.append(st[k:k + ng].max() - st[k:k + ng].min()); spread_e.append(en[k:k + ng].max() - en[k:k + ng].min())
    spread_s, spread_e = np.array(spread_s), np.array(spread_e)
    for name, v in (("start", spread_s), ("end", spread_e)):
        print("sibling %s spread (us): median %.1f  mean %.1f  p90 %.1f  max %.1f" % (name, np.median(v), v.mean(), np.percentile(v, 90), v.max()))
    print("tiles whose siblings start within 5 us: %.0f %%" % (100.0 * np.mean(spread_s < 5.0)))

import sys; sys.path.insert(0, "/root/repo")
import numpy as np
import rome_jl_amd as R
import oracle as ro
rng = np.random.default_rng(2024)
worst = 0.0; worstm = 0.0; bad = 0
for trial in range(120):
    N = int(rng.choice([2, 3, 5, 17, 63, 64, 65, 100, 127, 128, 129, 255, 256, 257, 400, 511, 512]))
    V = 6
    kind = trial % 6
    bel = np.empty((V, 3, N))
    for v in range(V):
        for k in range(3):
            if kind == 0: x = rng.normal(rng.normal(0, 10), 10 ** rng.uniform(-3, 2), N)
            elif kind == 1: x = np.where(rng.random(N) < 0.5, rng.normal(-1, 0.05, N), rng.normal(1, 0.3, N))
            elif kind == 2: x = np.round(rng.normal(0, 1, N), 1)                      # many duplicates
            elif kind == 3: x = np.concatenate([np.zeros(N - 1), [rng.normal(0, 5)]])  # constant + one outlier
            elif kind == 4: x = rng.standard_cauchy(N)                                 # heavy tails
            else: x = rng.uniform(-3.14, 3.14, N)
            bel[v, k] = x
    bel[:, 2] = np.arctan2(np.sin(bel[:, 2]), np.cos(bel[:, 2]))
    h = R.kde_bandwidth(bel, 0b100, 1e-2, 1e-5)
    ho = ro.kde_bandwidths(bel, 0b100, 1e-2, 1e-5)
    assert np.isfinite(h).all() and (h > 0).all(), (trial, N, kind)
    rel = np.abs(h / ho - 1)
    lim = np.array([3e-2, 3e-2, 3e-5])
    if (rel > lim).any():
        bad += 1; print("bw mismatch", trial, N, kind, rel.max(0))
    worst = max(worst, rel[:, :2].max())
    m = R.kde_max(bel, h); mo = ro.kde_max(bel, h)
    step = 1.2 * (bel.max(2) - bel.min(2)) / 199
    dm = np.abs(m - mo)
    if not (dm <= 1.0001 * step + 1e-12).all():
        bad += 1; print("max mismatch", trial, N, kind, (dm / np.maximum(step, 1e-300)).max())
    worstm = max(worstm, (dm > 1e-9 * (1 + np.abs(mo))).mean())
print("done: bad", bad, "worst bw rel (x,y)", worst, "fraction of kde_max points differing", worstm)

"""Distribution checks for the device noise generator (the stand-in for
`torch.randn_like`, glow_tts/models.py:348): moments, a Kolmogorov-Smirnov test
against N(0,1), independence across rows / channels / frames / seeds."""
import numpy as np


def check_gauss_noise(engine, B, C, T, ks_bound):
    from scipy import stats

    x = engine.gauss_noise(1234, B, C, T)
    n = x.size
    assert np.isfinite(x).all()
    flat = x.reshape(-1).astype(np.float64)
    se = 1.0 / np.sqrt(n)
    assert abs(flat.mean()) < 5 * se, flat.mean()
    assert abs(flat.var() - 1.0) < 5 * np.sqrt(2.0) * se, flat.var()
    assert abs(stats.skew(flat)) < 5 * np.sqrt(6.0) * se
    assert abs(stats.kurtosis(flat)) < 5 * np.sqrt(24.0) * se
    d = stats.kstest(flat, "norm").statistic
    assert d < ks_bound, d  # K-S critical value at alpha = 0.001 is 1.95/sqrt(n)
    assert np.abs(flat).max() < 6.5 and np.abs(flat).max() > 3.0  # tails present, nothing absurd
    # independence: neighbouring frames, neighbouring channels, neighbouring rows, and the same
    # (row, channel, frame) under another seed — sample correlations within 5 sigma of 0
    def corr(a, b):
        a = a.reshape(-1).astype(np.float64)
        b = b.reshape(-1).astype(np.float64)
        return float(np.corrcoef(a, b)[0, 1]), 5.0 / np.sqrt(a.size)

    for a, b in ((x[:, :, 1:], x[:, :, :-1]), (x[:, 1:], x[:, :-1]), (x[:, :, 2:], x[:, :, :-2])):
        r, lim = corr(a, b)
        assert abs(r) < lim, r
    if B > 1:
        r, lim = corr(x[1:], x[:-1])
        assert abs(r) < lim, r
    y = engine.gauss_noise(1235, B, C, T)
    r, lim = corr(x, y)
    assert abs(r) < lim and not np.array_equal(x, y)
    assert np.array_equal(x, engine.gauss_noise(1234, B, C, T))  # counter based: same key, same draw
    # a (row, channel, frame) draw does not depend on the launch geometry around it
    z = engine.gauss_noise(1234, 1, C, max(1, T // 2))
    assert np.array_equal(z[0], x[0, :, : max(1, T // 2)])
    # every channel / row by itself is N(0,1) too (a per-stream bias would hide in the pooled test)
    per = x.reshape(B * C, T).astype(np.float64)
    assert np.abs(per.mean(axis=1)).max() < 5.5 / np.sqrt(T)

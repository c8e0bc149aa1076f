"""Host-side check of the rounding STRACK's threshold slots rely on (csrc/grid_common.hpp st_theta).

The sign form of the f16 matrix-core filter carries -theta in two f16 pieces against the constants 2^15 and 2^6; the search is exact as long as
the value the pieces represent, theta_hat, satisfies   theta_hat >= theta + 16.2 u |theta_hat| + eps   (u = 2^-24) for the exact
theta = thr scale^2 - Rs — then a pair at or below its query's threshold cannot come out of the MFMA without its sign, whatever the (assumed, and
on the device measured) accumulation error does.  This file restates st_theta operation by operation in numpy — f32 arithmetic with one rounding
per operation, f16 conversions toward zero including subnormals — and checks that inequality over the magnitudes a search produces and far beyond
(the GPU-side twin: tests/test_gpu_parity.py::test_sign_form_never_misses_a_pair_at_or_below_its_threshold, through the kernel's own code)."""
import numpy as np

U = 2.0 ** -24
LIM = 2129920000.0          # 65 000 x 2^15


def f32(x):
    return np.asarray(x, np.float64).astype(np.float32)


def rtz16(x32):
    """float32 -> float16 rounded toward zero (v_cvt_pkrtz_f16_f32), as float64 values; finite input beyond the f16 range -> +-65504"""
    x = np.asarray(x32, np.float32)
    with np.errstate(over="ignore"):
        h = x.astype(np.float16)
    h = np.where(np.isinf(h) & np.isfinite(x), np.copysign(np.float16(65504.0), x.astype(np.float16)), h).astype(np.float16)
    too_big = np.abs(h.astype(np.float64)) > np.abs(x.astype(np.float64))
    h = np.where(too_big, np.nextafter(h, np.float16(0.0)), h).astype(np.float16)
    return h.astype(np.float64)


def st_theta(thr, sc2, Rs):
    """the device function, one rounding per operation; returns (hi, lo) as float64 values of the two f16 pieces"""
    thr, sc2, Rs = (np.asarray(a, np.float32) for a in (thr, sc2, Rs))
    with np.errstate(over="ignore", invalid="ignore"):
        th = f32(thr.astype(np.float64) * sc2.astype(np.float64) - Rs.astype(np.float64))            # v_fma_f32: one rounding
    th = np.clip(th, np.float32(-LIM), np.float32(LIM))                                                 # v_med3_f32 (+-inf land on the limits)
    th = f32(np.abs(th).astype(np.float64) * 2.0 ** -19 + th.astype(np.float64))                       # fma(|th|, 2^-19, th)
    hi = rtz16(f32(th.astype(np.float64) * 2.0 ** -15))
    rem = f32(th.astype(np.float64) - hi * 32768.0)                                                     # fma(-hi, 2^15, th): exact
    lo = f32(rem.astype(np.float64) * 2.0 ** -6)
    lo = f32(np.abs(lo).astype(np.float64) * 2.0 ** -9 + lo.astype(np.float64))                        # fma(|lo|, 2^-9, lo)
    lo = f32(lo.astype(np.float64) + 2.0 ** -23)
    return hi, rtz16(lo), rem


def check(thr, sc2, Rs):
    hi, lo, rem = st_theta(thr, sc2, Rs)
    theta_hat = hi * 32768.0 + lo * 64.0
    theta = np.asarray(thr, np.float64) * np.asarray(sc2, np.float64) - np.asarray(Rs, np.float64)     # exact in f64 for f32 inputs of sane exponents
    assert np.all(np.isfinite(theta_hat)) and np.all(np.abs(hi) <= 65504.0) and np.all(np.abs(lo) <= 65504.0)
    inside = np.abs(theta) <= LIM * 0.999
    need = theta + 16.2 * U * np.abs(theta_hat) + 2.0 ** -24                                            # eps: the kernel wants E <= -(something positive)
    assert np.all(theta_hat[inside] >= need[inside]), float(np.min((theta_hat - need)[inside]))
    # ... and it is not wasteful: the slack stays within 2^-17 |theta| + the absolute term of the second piece
    assert np.all((theta_hat - theta)[inside] <= 2.0 ** -17 * np.abs(theta[inside]) + 64.0 * 2.0 ** -13)
    # beyond the limits the sign of the clamped value decides: a huge threshold flags everything (|S| <= 2.5e7), a hugely negative one nothing
    assert np.all(theta_hat[theta > LIM] >= LIM * 0.999) and np.all(theta_hat[theta < -LIM] <= -LIM * 0.999 + 1.0)
    assert np.all(np.abs(rem) <= 2.0 ** -10 * 65504.0 * 32768.0 * 1.001)
    return theta_hat, theta


def test_threshold_pieces_round_up_over_search_magnitudes():
    rng = np.random.default_rng(20261004)
    n = 400_000
    # scaled units: |r| up to 32 000 per coordinate, thresholds from (1 mm)^2 to (100 m)^2 at scales 2^-10 .. 2^10 per unit
    Rs = (rng.uniform(0, 1, n) ** 4 * 3.0 * 32000.0 ** 2).astype(np.float32)
    sc2 = (2.0 ** (2 * rng.integers(-10, 11, n))).astype(np.float32)
    thr = (10.0 ** rng.uniform(-6, 4, n)).astype(np.float32)
    check(thr, sc2, Rs)
    # thresholds right at |r|^2 (theta ~ 0: a candidate about as far as the super-tile's centre), both signs of the difference
    thr2 = (Rs.astype(np.float64) / sc2.astype(np.float64) * (1.0 + rng.uniform(-1e-5, 1e-5, n))).astype(np.float32)
    check(thr2, sc2, Rs)


def test_threshold_pieces_small_subnormal_and_clamped_values():
    rng = np.random.default_rng(7)
    n = 200_000
    # tiny thetas: second piece in f16's subnormal range (spacing 2^-24 x 2^6), first piece zero
    Rs = (rng.uniform(0, 1, n) * 1e-3).astype(np.float32)
    thr = (rng.uniform(0, 1, n) * 1e-3).astype(np.float32)
    check(thr, np.ones(n, np.float32), Rs)
    check(np.zeros(n, np.float32), np.ones(n, np.float32), np.zeros(n, np.float32))
    # enormous thresholds / scales: the products overflow f32 or the limit
    thr = (10.0 ** rng.uniform(0, 38, n)).astype(np.float32)
    sc2 = (2.0 ** (2 * rng.integers(-60, 61, n))).astype(np.float32)
    Rs = (rng.uniform(0, 1, n) * 3.0e9).astype(np.float32)
    check(thr, sc2, Rs)
    # "never": thr = -inf gives theta at its lower limit
    hi, lo, _ = st_theta(np.full(8, -np.inf, np.float32), np.ones(8, np.float32), np.linspace(0, 3e9, 8).astype(np.float32))
    assert np.all(hi * 32768.0 + lo * 64.0 <= -LIM * 0.999)


def test_threshold_pieces_exhaustive_near_a_binade_edge():
    # every f32 theta in a few binades around where the first piece changes its exponent and where it becomes subnormal
    for centre in (2.0 ** 15 * 2.0 ** -14, 2.0, 32768.0, 2.0 ** 15 * 1024.0):
        base = np.float32(centre)
        vals = base.view(np.uint32).astype(np.int64) + np.arange(-200_000, 200_000, 7)
        th = vals.astype(np.uint32).view(np.float32)
        for sign in (1.0, -1.0):
            t = (th * np.float32(sign)).astype(np.float32)
            # thr = t + Rs with Rs = 0: theta = t exactly
            check(t, np.ones_like(t), np.zeros_like(t))

"""The footprint test of the fast blend kernels (csrc/gsx_raster_fast.hip: footprint / footprint_hits), restated in numpy and checked
against brute force: the minimum of N(d) = (l00 du + l01 dv)^2 + (l11 dv)^2 over a rectangle lies on one of the two sides facing the
centre (or is 0 inside), so two clamped 1-D minima give it exactly; a culled rectangle can therefore hold no point with N <= rad2,
which is what makes skipping the (wave, Gaussian) / (block, Gaussian) pair result-neutral (alpha < 1/255 on every pixel of it)."""
import numpy as np


def footprint_hits(u0, v0, rad2, k2, l00, l01, l11, b0, b1, b2, b3):
    """Line-by-line restatement of the device function (float32 arithmetic)."""
    f = np.float32
    xa, xb, ya, yb = f(b0 - u0), f(b1 - u0), f(b2 - v0), f(b3 - v0)
    med3 = lambda a, b, c: np.float32(sorted([a, b, c])[1])  # noqa: E731
    xc, yc = med3(f(0), xa, xb), med3(f(0), ya, yb)
    m = f(l01 * yc)
    t = f(med3(f(-m), f(l00 * xa), f(l00 * xb)) + m)
    e1 = f(l11 * yc)
    n1 = f(t * t + e1 * e1)
    ys = med3(f(k2 * xc), ya, yb)
    t2 = f(l01 * ys + f(l00 * xc))
    e2 = f(l11 * ys)
    n2 = f(t2 * t2 + e2 * e2)
    return not (min(n1, n2) > rad2), float(min(n1, n2))


def test_two_side_minimum_is_the_minimum_over_the_rectangle():
    rng = np.random.default_rng(0)
    worst = 0.0
    for _ in range(400):
        l00, l11 = np.float32(rng.uniform(0.2, 30.0)), np.float32(rng.uniform(0.2, 30.0))
        l01 = np.float32(rng.uniform(-20.0, 20.0))
        k2 = np.float32(-(l00 * l01) / (l01 * l01 + l11 * l11))
        u0, v0 = np.float32(rng.uniform(-1, 1)), np.float32(rng.uniform(-1, 1))
        w, h = rng.uniform(0.003, 0.2, 2)
        b0, b2 = np.float32(rng.uniform(-1.2, 1.0)), np.float32(rng.uniform(-1.2, 1.0))
        b1, b3 = np.float32(b0 + w), np.float32(b2 + h)
        _, nmin = footprint_hits(u0, v0, np.float32(1.0), k2, l00, l01, l11, b0, b1, b2, b3)
        # brute force in float64: dense grid over the rectangle incl. its border
        x = np.linspace(float(b0), float(b1), 401) - float(u0)
        y = np.linspace(float(b2), float(b3), 401) - float(v0)
        X, Y = np.meshgrid(x, y)
        N = (float(l00) * X + float(l01) * Y) ** 2 + (float(l11) * Y) ** 2
        brute = N.min()
        # the closed form can only be BELOW the sampled minimum (finer than any grid) and must agree with it to the grid's resolution
        assert nmin <= brute * (1 + 1e-4) + 1e-6, (nmin, brute)
        scale = max(brute, 1e-3)
        worst = max(worst, (brute - nmin) / scale)
    assert worst < 0.05   # 401 x 401 samples of a quadratic: the grid minimum overshoots the true one by a few per cent at most


def test_a_culled_rectangle_contains_no_contributing_point():
    """Random Gaussians in the kernel's own parametrisation: rad2 = tau2 * max den over the tile corners * 1.0021 + 1e-6; whenever the test
    says 'no hit' for a block of pixel centres, no pixel centre of that block satisfies alpha >= 1/255."""
    rng = np.random.default_rng(1)
    LOG2_255 = 7.994353436858858
    culled = kept = 0
    for _ in range(600):
        l00, l11 = rng.uniform(60.0, 2000.0), rng.uniform(60.0, 2000.0)
        l01 = rng.uniform(-1500.0, 1500.0)
        d1, d2 = rng.uniform(-0.3, 0.3, 2)
        d3, d5 = rng.uniform(0.5, 1.5, 2)
        d4 = rng.uniform(-0.5, 0.5)
        lo = np.log2(rng.uniform(0.01, 0.999))
        u0, v0 = rng.uniform(-0.012, 0.028, 2)
        # a 16x16 tile of pixel centres, pitch 1e-3, origin 0
        px = (np.arange(16) + 0.5) * 1e-3
        U, V = np.meshgrid(px, px)
        du, dv = U - u0, V - v0
        N = (l00 * du + l01 * dv) ** 2 + (l11 * dv) ** 2
        den = 1 + du * (d1 + d3 * du + d4 * dv) + dv * (d2 + d5 * dv)
        alpha = np.minimum(0.999, np.exp2(lo - N / den))
        tau2 = lo + LOG2_255
        if tau2 <= 0:
            continue
        dmax = 0.0
        for cu in (px[0], px[-1]):
            for cv in (px[0], px[-1]):
                a, b = cu - u0, cv - v0
                dmax = max(dmax, 1 + a * (d1 + d3 * a + d4 * b) + b * (d2 + d5 * b))
        rad2 = np.float32(tau2 * dmax * 1.0021 + 1e-6)
        k2 = np.float32(-(l00 * l01) / (l01 * l01 + l11 * l11))
        for by in range(4):
            for bx in range(4):
                hit, _ = footprint_hits(np.float32(u0), np.float32(v0), rad2, k2, np.float32(l00), np.float32(l01), np.float32(l11),
                                        np.float32(px[bx * 4]), np.float32(px[bx * 4 + 3]), np.float32(px[by * 4]), np.float32(px[by * 4 + 3]))
                contrib = (alpha[by * 4:by * 4 + 4, bx * 4:bx * 4 + 4] >= 1.0 / 255.0).any()
                if not hit:
                    culled += 1
                    assert not contrib
                else:
                    kept += 1
    assert culled > 1000 and kept > 500   # the test is exercised on both sides

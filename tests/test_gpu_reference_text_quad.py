"""The HIP path of the §8-f4 consumers against vectors produced by EXECUTING THE REFERENCE'S TEXT (tests/golden/reference_text_quad_vectors.json;
generator tests/golden/make_reference_text_quad_vectors.py, interpreter oracle/nim_subset_quad.py): the function forms and the discrete forms of
cumtrapz / cumsimpson (integrate.nim:121-175, 330-400), newHermiteSpline with and without derivatives and eval / derivEval with every
ExtrapolateKind (interpolate.nim:187-264, 299-390) — bit for bit, through the C ABI."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VEC = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_text_quad_vectors.json")))
POLY_SRC = "for (int c = 0; c < {d}; ++c) dy[c] = ((p[0] * t + p[1]) * t) * (1.0 + (double)c) + p[2];"


def fh(xs):
    return np.array([float.fromhex(x) for x in xs], dtype=np.float64)


def test_function_forms_equal_the_reference_text(nn, dev):
    fs = {d: nn.Rhs.custom(d, POLY_SRC.format(d=d), keys=("a", "b", "c"), name=f"poly{d}") for d in (1, 3)}
    for c in VEC["cumquad_fn"]:
        p, d = fh(c["params"]), max(c["dim"], 1)
        fn = nn.cumtrapz if c["rule"] == "trapz" else nn.cumsimpson
        got = fn(fs[d], fh(c["X"]), ctx=nn.newNumContext({"a": p[0], "b": p[1], "c": p[2]}), dx=float.fromhex(c["dx"]), n=3).cpu().numpy()
        assert got.shape[0] == c["rows"], c["name"]
        for i in (0, 2):
            g = got[:, i] if d == 1 else got[:, :, i]
            assert np.array_equal(g.ravel(), fh(c["out"])), c["name"]


@pytest.mark.parametrize("c", VEC["cumquad_discrete"], ids=lambda c: c["name"])
def test_discrete_forms_equal_the_reference_text(nn, dev, c):
    """The entries get the CALLER's X and Y, exactly what the text got — strictly ascending, unsorted, with pure duplicates, descending, -0.0 next to 0.0, a
    sorted X with a repeated maximum: sortAndTrimDataset (utils.nim:404-413) runs below the boundary (round 6).  cumtrapz returns one row per distinct abscissa,
    ascending; cumsimpson the rows hermiteInterpolate yields at the caller's abscissae."""
    import torch
    X = fh(c["X"])
    Y = np.stack([fh(y) for y in c["Y"]], axis=1)                      # [n, 3]: every series its own column
    Yb = np.ascontiguousarray(np.tile(Y, (1, 50)))                     # ... repeated: more than one wave of series
    got = nn.cumtrapz(torch.from_numpy(Yb).to(dev), X).cpu().numpy()
    want = np.stack([fh(v) for v in c["cumtrapz"]], axis=1)
    assert got.shape == (want.shape[0], 150) and np.array_equal(got, np.tile(want, (1, 50)))
    assert np.array_equal(nn.cumtrapz(Yb, X), got)                     # the host-pointer entry
    xs, (ys,) = nn.sortAndTrimDataset(X, torch.from_numpy(Yb).to(dev))
    assert np.array_equal(xs, fh(c["X_sorted_trimmed"])) and np.array_equal(np.signbit(xs), np.signbit(fh(c["X_sorted_trimmed"])))
    assert np.array_equal(ys.cpu().numpy(), np.tile(np.stack([fh(v) for v in c["Y_sorted_trimmed"]], axis=1), (1, 50)))
    if isinstance(c["cumsimpson"], dict):
        with pytest.raises(ValueError):
            nn.cumsimpson(torch.from_numpy(Yb).to(dev), X)
    else:
        got = nn.cumsimpson(torch.from_numpy(Yb).to(dev), X).cpu().numpy()
        want = np.stack([fh(v) for v in c["cumsimpson"]], axis=1)
        assert got.shape == (want.shape[0], 150) and np.array_equal(got, np.tile(want, (1, 50)))
        assert np.array_equal(nn.cumsimpson(Yb, X), got)


@pytest.mark.parametrize("c", VEC["impure"], ids=lambda c: c["name"])
def test_impure_duplicates_are_refused_like_the_reference_text(nn, dev, c):
    """the same x with different y — in ONE series of the batch, NaN duplicates included (NaN != NaN) —: ValueError (utils.nim:372) from every discrete consumer"""
    import torch
    X, y = fh(c["X"]), fh(c["Y"])
    clean = np.cos(X)
    Yb = np.ascontiguousarray(np.stack([clean] * 69 + [y] + [clean] * 5, axis=1))
    Yt = torch.from_numpy(Yb).to(dev)
    for call in (lambda: nn.cumtrapz(Yt, X), lambda: nn.cumsimpson(Yt, X), lambda: nn.newHermiteSpline(X, Yt), lambda: nn.newHermiteSpline(X, Yt, Yt.clone()),
                 lambda: nn.cumtrapz(Yb, X), lambda: nn.sortAndTrimDataset(X, Yt)):
        with pytest.raises(ValueError):
            call()
    ok = np.ascontiguousarray(np.tile(clean[:, None], (1, 75)))      # the same abscissae with pure duplicates everywhere: accepted
    assert nn.cumtrapz(torch.from_numpy(ok).to(dev), X).shape[0] == 3
    with pytest.raises(ValueError):                                    # NaN in X: no defined order in the reference's sort — refused
        nn.cumtrapz(Yt[:3], np.array([0.0, np.nan, 1.0]))


def test_sort_and_trim_of_a_long_series_equals_the_oracle(nn, oracle, dev):
    """100 000 abscissae in random order with ~1 % pure duplicates (more rows than one grid dimension holds; 700 on the ISA-backed fake node), 64 series: cumtrapz / cumsimpson / the sorted
    dataset through the C ABI == the oracle's restatement of sortAndTrimDataset + the rules, series by series."""
    import torch
    rng = np.random.default_rng(11)
    on_isa_node = bool(os.environ.get("FAKE_HIP_LIB"))   # (scripts/run_gpu_suite_on_isa_node.py: the interpreter takes tens of microseconds per wave-instruction)
    n, M = (700, 64) if on_isa_node else (100_000, 64)
    X = rng.uniform(-3.0, 7.0, n)
    dup = rng.integers(0, n, n // 100)
    X[dup] = X[(dup * 7919 + 13) % n]
    Y = np.cos(X)[:, None] * (1.0 + np.arange(M))[None, :] + X[:, None] * 0.125           # a function of x: every duplicate is pure
    Yt = torch.from_numpy(np.ascontiguousarray(Y)).to(dev)
    got_t = nn.cumtrapz(Yt, X).cpu().numpy()
    got_s = nn.cumsimpson(Yt, X).cpu().numpy()
    xs, (ys,) = nn.sortAndTrimDataset(X, Yt)
    assert len(xs) < n and got_t.shape == (len(xs), M) and got_s.shape == (n, M)
    for m in (0, 17, M - 1):
        ox, (oy,) = oracle.sort_and_trim(X, Y[:, m])
        assert np.array_equal(xs, ox) and np.array_equal(ys[:, m].cpu().numpy(), oy)
        assert np.array_equal(got_t[:, m], oracle.cumtrapz(Y[:, m], X))
        assert np.array_equal(got_s[:, m], oracle.cumsimpson(Y[:, m], X))


@pytest.mark.parametrize("c", VEC["hermite"], ids=lambda c: c["name"])
def test_hermite_spline_equals_the_reference_text(nn, dev, c):
    import torch
    X, Y, dY, xq, val = fh(c["X"]), fh(c["Y"]), fh(c["dY"]), fh(c["xq"]), float.fromhex(c["extrap_value"])    # knots in the CALLER's order (two cases unsorted)
    Yb, dYb = (torch.from_numpy(np.ascontiguousarray(np.tile(a[:, None], (1, 70)))).to(dev) for a in (Y, dY))
    with_dy = nn.newHermiteSpline(X, Yb, dYb)                            # the constructor sorts and trims (interpolate.nim:231)
    estimated = nn.newHermiteSpline(X, Yb)                              # ... (:244), then three-point slopes on the device (:245-253)
    assert np.array_equal(with_dy.X, fh(c["X_sorted_trimmed"])) and np.array_equal(estimated.X, fh(c["X_sorted_trimmed"]))
    assert np.array_equal(estimated.dY.cpu().numpy(), np.tile(fh(c["slopes_from_text"])[:, None], (1, 70)))
    # the C entries given the caller's arrays directly: the device-pointer eval entry sorts per call, the host-pointer one too (dY = NULL: estimates the slopes)
    host = nn.newHermiteSpline(X, np.ascontiguousarray(np.tile(Y[:, None], (1, 70))))
    assert np.array_equal(host.eval(xq), np.tile(fh(c["estimated_slopes"]["Native"]["eval"])[:, None], (1, 70)))
    raw = nn.HermiteSpline.__new__(nn.HermiteSpline)
    raw.X, raw.Y, raw.dY, raw.M, raw.host = X, Yb, dYb, 70, False
    assert np.array_equal(raw.derivEval(xq, extrap="Edge").cpu().numpy(), np.tile(fh(c["with_dY"]["Edge"]["derivEval"])[:, None], (1, 70)))
    for key, spl in (("with_dY", with_dy), ("estimated_slopes", estimated)):
        for ex, rec in c[key].items():
            v = val if ex == "Constant" else None
            e = spl.eval(xq, extrap=ex, extrapValue=v).cpu().numpy()
            d = spl.derivEval(xq, extrap=ex, extrapValue=v).cpu().numpy()
            assert np.array_equal(e, np.tile(fh(rec["eval"])[:, None], (1, 70))), (c["name"], key, ex)
            assert np.array_equal(d, np.tile(fh(rec["derivEval"])[:, None], (1, 70))), (c["name"], key, ex)
    with pytest.raises(ValueError):
        with_dy.eval(xq[:1], extrap="Error")

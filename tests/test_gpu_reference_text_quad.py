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


@pytest.mark.parametrize("c", [c for c in VEC["cumquad_discrete"] if c["strictly_ascending"]], ids=lambda c: c["name"])
def test_discrete_forms_equal_the_reference_text(nn, dev, c):
    import torch
    X = fh(c["X"])
    Y = np.stack([fh(y) for y in c["Y"]], axis=1)                      # [n, 3]: every series its own column
    Yb = np.ascontiguousarray(np.tile(Y, (1, 50)))                     # ... repeated: more than one wave of series
    got = nn.cumtrapz(torch.from_numpy(Yb).to(dev), X).cpu().numpy()
    want = np.stack([fh(v) for v in c["cumtrapz"]], axis=1)
    assert np.array_equal(got, np.tile(want, (1, 50)))
    assert np.array_equal(nn.cumtrapz(Yb, X), got)                     # the host-pointer entry
    if isinstance(c["cumsimpson"], dict):
        with pytest.raises(ValueError):
            nn.cumsimpson(torch.from_numpy(Yb).to(dev), X)
    else:
        got = nn.cumsimpson(torch.from_numpy(Yb).to(dev), X).cpu().numpy()
        want = np.stack([fh(v) for v in c["cumsimpson"]], axis=1)
        assert np.array_equal(got, np.tile(want, (1, 50)))
        assert np.array_equal(nn.cumsimpson(Yb, X), got)


@pytest.mark.parametrize("c", VEC["hermite"], ids=lambda c: c["name"])
def test_hermite_spline_equals_the_reference_text(nn, dev, c):
    import torch
    X, Y, dY, xq, val = fh(c["X"]), fh(c["Y"]), fh(c["dY"]), fh(c["xq"]), float.fromhex(c["extrap_value"])
    Yb, dYb = (torch.from_numpy(np.ascontiguousarray(np.tile(a[:, None], (1, 70)))).to(dev) for a in (Y, dY))
    with_dy = nn.newHermiteSpline(X, Yb, dYb)
    estimated = nn.newHermiteSpline(X, Yb)                              # three-point slopes on the device (interpolate.nim:241-253)
    assert np.array_equal(estimated.dY.cpu().numpy(), np.tile(fh(c["slopes_from_text"])[:, None], (1, 70)))
    for key, spl in (("with_dY", with_dy), ("estimated_slopes", estimated)):
        for ex, rec in c[key].items():
            v = val if ex == "Constant" else None
            e = spl.eval(xq, extrap=ex, extrapValue=v).cpu().numpy()
            d = spl.derivEval(xq, extrap=ex, extrapValue=v).cpu().numpy()
            assert np.array_equal(e, np.tile(fh(rec["eval"])[:, None], (1, 70))), (c["name"], key, ex)
            assert np.array_equal(d, np.tile(fh(rec["derivEval"])[:, None], (1, 70))), (c["name"], key, ex)
    with pytest.raises(ValueError):
        with_dy.eval(xq[:1], extrap="Error")

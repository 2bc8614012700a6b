"""Static checks of the repository's Python that still apply to code no CPU test run executes — bench.py's GPU legs, the -m gpu tests, the A/B scripts (a round
without GPU access leaves all of that unexecuted; a typo in it shows on the GPU box, under `pytest -x` or inside the driver's bench run): names read but never
bound (tools/undefined_names.py), calls of C-ABI entries against the signature table the library is loaded with (tools/check_capi_calls.py: entry exists, argument
count), and attribute chains on the package (`nn.solveODE`, `nn.Rhs.lorenz`, `nn._lib.NNHIP_EVALUE` ...) against the imported package."""
import ast
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _files():
    pats = ["bench.py", "__graft_entry__.py", "numericalnim_amd/*.py", "scripts/*.py", "tests/*.py", "tests/golden/*.py", "oracle/*.py", "tools/*.py", "examples/*.py"]
    out = []
    for p in pats:
        out += sorted(glob.glob(os.path.join(ROOT, p)))
    assert len(out) > 80
    return out


def test_no_name_is_read_that_is_never_bound():
    import undefined_names
    bad = [(os.path.relpath(p, ROOT),) + b for p in _files() for b in undefined_names.check(p)]
    assert not bad, bad


def test_c_abi_calls_match_the_signature_table():
    import check_capi_calls
    from numericalnim_amd._lib import SIGNATURES
    bad, total = [], 0
    for p in _files():
        b, seen = check_capi_calls.check(p, SIGNATURES)
        total += seen
        bad += [(os.path.relpath(p, ROOT),) + x for x in b]
    assert not bad, bad
    assert total > 200


def test_attribute_chains_on_the_package_exist():
    """`nn.<a>.<b>...` wherever `nn` is the package (the fixture of tests/conftest.py, `import numericalnim_amd as nn` elsewhere)."""
    import numericalnim_amd as nn
    bad, total = [], 0
    for p in _files():
        src = open(p).read()
        if "numericalnim_amd" not in src and "def test_" not in src:
            continue
        for node in ast.walk(ast.parse(src, p)):
            if not isinstance(node, ast.Attribute):
                continue
            chain, cur = [], node
            while isinstance(cur, ast.Attribute):
                chain.append(cur.attr)
                cur = cur.value
            if not (isinstance(cur, ast.Name) and cur.id == "nn"):
                continue
            obj, ok = nn, True
            for a in reversed(chain):
                if not hasattr(obj, a):
                    ok = False
                    break
                obj = getattr(obj, a)
                if callable(obj) and not isinstance(obj, type) and not hasattr(obj, "__wrapped__") and type(obj).__name__ in ("function", "builtin_function_or_method", "method"):
                    break  # what a call returns is not checked
                if not (isinstance(obj, type) or type(obj).__name__ == "module"):
                    break  # an instance attribute: only its existence on the class / module is checked
            total += 1
            if not ok:
                bad.append((os.path.relpath(p, ROOT), node.lineno, "nn." + ".".join(reversed(chain))))
    assert not bad, sorted(set(bad))
    assert total > 500


def test_calls_of_package_functions_bind_to_their_signatures():
    """`nn.f(a, b, key=...)`: the positional count and every keyword must bind to f's signature (inspect) — a misspelt keyword in a GPU-only test is a TypeError
    on the GPU box.  Calls with * / ** arguments are skipped."""
    import inspect
    import numericalnim_amd as nn
    bad, total = [], 0
    for p in _files():
        src = open(p).read()
        if "numericalnim_amd" not in src and "def test_" not in src:
            continue
        for node in ast.walk(ast.parse(src, p)):
            if not (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute)):
                continue
            chain, cur = [], node.func
            while isinstance(cur, ast.Attribute):
                chain.append(cur.attr)
                cur = cur.value
            if not (isinstance(cur, ast.Name) and cur.id == "nn"):
                continue
            obj = nn
            try:
                for a in reversed(chain):
                    obj = getattr(obj, a)
            except AttributeError:
                continue  # reported by the test above
            if any(isinstance(a, ast.Starred) for a in node.args) or any(k.arg is None for k in node.keywords):
                continue
            try:
                sig = inspect.signature(obj)
            except (TypeError, ValueError):
                continue
            total += 1
            try:
                sig.bind(*[None] * len(node.args), **{k.arg: None for k in node.keywords})
            except TypeError as e:
                bad.append((os.path.relpath(p, ROOT), node.lineno, "nn." + ".".join(reversed(chain)), str(e)))
    assert not bad, bad
    assert total > 300

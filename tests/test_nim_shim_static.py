"""The Nim shim has never met a compiler (no Nim toolchain in the build image or on the GPU boxes), and the C ABI it binds
has changed in every round.  What CAN be checked without one: every call the shim makes into the raw bindings is compared, argument by
argument, with the signature generated from include/nnhip_ode.h (nim/nnhip_ode_bindings.nim, itself checked against the header by
tests/test_abi_and_host_logic.py) — the number of arguments, and for every argument whose kind can be read off the source text (`addr x`,
`nil`, `.cint` / `.int64` / `.cdouble` / `.cstring` conversions, literals, locals with a visible declaration) that it fits the parameter: a
pointer where the C side takes a pointer, the element type of the seq whose first element is passed, a C integer of the right width where it
takes a scalar.  The class of mistake this catches — an entry that gained or reordered a parameter while the shim kept the old call — is the
likeliest one in a file that is only ever read.  The calls rhs_macro.nim makes into the shim are checked against the shim's own proc headers
the same way (arity within the required..total range, named arguments must exist)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NIM = os.path.join(ROOT, "nim")


def _blank_strings_and_comments(src):
    """Same length, same line structure: comment text and the inside of string / char literals become spaces (so brackets and commas inside
    them never count)."""
    out, i, n = [], 0, len(src)
    while i < n:
        ch = src[i]
        if src.startswith('"""', i):
            j = src.find('"""', i + 3)
            j = n if j < 0 else j + 3
            out.append('"' + "".join(c if c == "\n" else " " for c in src[i + 1:j - 1]) + '"')
            i = j
        elif ch == '"':
            j = i + 1
            raw = i > 0 and src[i - 1].isalpha() and not src[i - 1] == "&"   # r"..." style: no escapes (not used by the shim, kept for safety)
            while j < n and src[j] != '"' and src[j] != "\n":
                j += 2 if (src[j] == "\\" and not raw) else 1
            out.append('"' + " " * (j - i - 1) + '"')
            i = j + 1
        elif ch == "'" and i + 2 < n and (src[i + 2] == "'" or (src[i + 1] == "\\" and src.find("'", i + 2) - i <= 5)):
            j = src.find("'", i + 2)
            out.append("'" + " " * (j - i - 1) + "'")
            i = j + 1
        elif ch == "#":
            j = src.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i))
            i = j
        else:
            out.append(ch)
            i += 1
    return "".join(out)


def _split_top(s, sep=","):
    parts, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            parts.append("".join(cur).strip()); cur = []
        else:
            cur.append(ch)
    tail = "".join(cur).strip()
    if tail or parts:
        parts.append(tail)
    return parts


def _balanced(src, open_idx):
    depth = 0
    for j in range(open_idx, len(src)):
        if src[j] in "([{":
            depth += 1
        elif src[j] in ")]}":
            depth -= 1
            if depth == 0:
                return j
    raise AssertionError("unbalanced brackets")


def _bindings():
    """name -> ([(param, type)], return type) from the generated raw bindings."""
    text = open(os.path.join(NIM, "nnhip_ode_bindings.nim")).read()
    sigs = {}
    for m in re.finditer(r"^proc (\w+)\*\((.*?)\)(?:: ([\w ]+?))? \{\.importc", text, re.M):
        params = []
        for p in filter(None, (x.strip() for x in m.group(2).split(";"))):
            name, typ = p.split(":", 1)
            params.append((name.strip(), typ.strip()))
        sigs[m.group(1)] = (params, (m.group(3) or "void").strip())
    assert len(sigs) >= 60, len(sigs)
    return sigs


_INT_TYPES = {"cint", "int64", "int32", "cuint", "uint32", "uint64", "csize_t"}
_SAME = {"float": "cdouble", "float64": "cdouble", "cdouble": "cdouble", "int": "int64", "int64": "int64", "int32": "int32", "cint": "cint"}


def _decl_of(name, before):
    """The text of the last `let/var NAME ...` declaration in `before` (the blanked source up to the call), or of a proc parameter NAME."""
    best = None
    for m in re.finditer(r"\b(?:let|var)\s+" + re.escape(name) + r"\b\s*(:[^=\n]*)?(=\s*[^\n]*)?", before):
        best = ("local", (m.group(1) or "").lstrip(":").strip(), (m.group(2) or "").lstrip("=").strip())
    if best:
        return best
    heads = list(re.finditer(r"^proc\s+[\w`*]+\s*\(", before, re.M))
    if heads:
        h = heads[-1]
        end = _balanced(before + ")" * 50, h.end() - 1)
        for p in _split_top(before[h.end():end].replace(";", ",")):
            mm = re.match(r"([\w, ]+?)\s*:\s*([^=]+?)\s*(=.*)?$", p) or re.match(r"(\w+)\s*()(=.*)$", p)
            if mm and name in [x.strip() for x in mm.group(1).split(",")]:
                return ("param", mm.group(2).strip(), (mm.group(3) or "").lstrip("=").strip())
    return None


def _elem_type(container, before):
    """element type of the seq / array variable `container` where its declaration shows it"""
    d = _decl_of(container, before)
    if not d:
        return None
    _, typ, init = d
    for text in (typ, init):
        m = re.search(r"(?:seq|openArray|newSeq)\[(\w+)\]", text)
        if m:
            return _SAME.get(m.group(1), m.group(1))
    m = re.match(r"@(\w+)$", init)             # var ts = @tspan : a copy of an openArray parameter
    if m:
        return _elem_type(m.group(1), before)
    m = re.match(r"(\w+)\.(\w+)$", init)       # var y0d = y0.data / var xs = s.X : a field — resolved through the object types of the shim
    if m:
        return _FIELD_ELEM.get(m.group(2))
    m = re.match(r"(\w+)\(", init)              # var params = paramsOf(f, ctx) / flatten(Y): the helper's declared return type
    if m and m.group(1) in _HELPER_RET:
        return _HELPER_RET[m.group(1)]
    m = re.match(r"(\w+)$", init)              # var s = shared : a copy of another variable
    if m:
        return _elem_type(m.group(1), before)
    return None


_FIELD_ELEM = {"data": "cdouble", "X": "cdouble", "Y": "cdouble", "dY": "cdouble"}   # OdeBatch.data: seq[float]; BatchHermiteSpline.X / .Y / .dY
_HELPER_RET = {"paramsOf": "cdouble", "flatten": "cdouble"}


def _kind(arg, before, sigs):
    """(kind, detail): kind in pointer / nil / cstring / int / float / unknown; detail = pointee or C scalar type where known."""
    a = arg.strip()
    while a.startswith("(") and _balanced(a, 0) == len(a) - 1:
        a = a[1:-1].strip()
    if a == "nil":
        return ("nil", None)
    m = re.match(r"(?:unsafeAddr|addr)\s+(\w+)(\[0\])?$", a)
    if m:
        if m.group(2):
            return ("pointer", _elem_type(m.group(1), before))
        d = _decl_of(m.group(1), before)
        typ = None
        if d:
            typ = d[1] or None
            if not typ and re.search(r"\.toC\b", d[2]):
                typ = "NnhipOptions"
        return ("pointer", _SAME.get(typ, typ))
    m = re.search(r"\.(cint|int64|int32|cdouble|cstring)$", a)
    if m:
        t = m.group(1)
        return ("cstring", None) if t == "cstring" else (("float", t) if t == "cdouble" else ("int", t))
    if re.match(r"-?\d+$", a):
        return ("int", "literal")
    if re.match(r"-?\d+\.\d*(e-?\d+)?$", a):
        return ("float", "literal")
    if re.match(r"if\b.*\belse\s*:", a):           # if c: X else: Y  — both branches must agree
        mm = re.match(r"if\b.*?:\s*(.*)\s+else\s*:\s*(.*)$", a)
        k1, k2 = _kind(mm.group(1), before, sigs), _kind(mm.group(2), before, sigs)
        if k1[0] == "nil":
            return k2 if k2[0] != "nil" else k1
        if k2[0] == "nil":
            return k1
        return k1 if k1 == k2 else ("unknown", None)
    if re.match(r"\w+$", a):
        d = _decl_of(a, before)
        if d:
            _, typ, init = d
            if typ:
                t = typ.split("=")[0].strip()
                if t in _INT_TYPES:
                    return ("int", t)
                if t in ("cdouble", "float"):
                    return ("float", "cdouble")
                if t == "cstring":
                    return ("cstring", None)
                if t.startswith("ptr ") or t in ("pointer", "cstringArray"):
                    return ("pointer", t[4:] if t.startswith("ptr ") else t)
                return ("unknown", None)
            m = re.match(r"(nnhip_\w+)\(", init)
            if m and m.group(1) in sigs:
                rt = sigs[m.group(1)][1]
                return ("int", rt) if rt in _INT_TYPES else ("unknown", None)
            if init.startswith("allocCStringArray"):
                return ("pointer", "cstringArray")
            if init and init != a:
                return _kind(init, before, sigs)
    return ("unknown", None)


def _fits(kind, detail, ptype):
    is_ptr = ptype.startswith("ptr ") or ptype in ("pointer", "cstringArray")
    if kind == "nil":
        return is_ptr or ptype == "cstring"
    if kind == "pointer":
        if not is_ptr:
            return False
        if detail is None or ptype == "pointer":
            return True
        want = ptype[4:].strip() if ptype.startswith("ptr ") else ptype
        if detail == "cstringArray":
            return want in ("cstring", "cstringArray")
        return _SAME.get(detail, detail) == _SAME.get(want, want)
    if kind == "cstring":
        return ptype == "cstring"
    if kind == "int":
        if detail == "literal":
            return ptype in _INT_TYPES or ptype == "cdouble"
        return ptype == detail
    if kind == "float":
        return ptype == "cdouble"
    return True


def _calls(blanked, prefix=r"nnhip_\w+"):
    for m in re.finditer(r"\b(" + prefix + r")\s*\(", blanked):
        end = _balanced(blanked, m.end() - 1)
        yield m.group(1), m.start(), _split_top(blanked[m.end():end]), blanked[:m.start()].count("\n") + 1


def test_every_raw_call_of_the_shim_matches_the_generated_signature():
    sigs = _bindings()
    checked = known = 0
    used = set()
    for fn in ("numericalnim_hip.nim", "rhs_macro.nim"):
        src = _blank_strings_and_comments(open(os.path.join(NIM, fn)).read())
        for name, pos, args, line in _calls(src):
            assert name in sigs, f"{fn}:{line}: {name} is not declared in include/nnhip_ode.h"
            params, _ = sigs[name]
            used.add(name)
            assert len(args) == len(params), f"{fn}:{line}: {name} takes {len(params)} arguments ({', '.join(p for p, _ in params)}), the shim passes {len(args)}"
            for (pname, ptype), arg in zip(params, args):
                kind, detail = _kind(arg, src[:pos], sigs)
                checked += 1
                known += kind != "unknown"
                assert _fits(kind, detail, ptype), f"{fn}:{line}: {name}: argument `{arg}` ({kind} {detail or ''}) does not fit parameter {pname}: {ptype}"
    assert len(used) >= 18, sorted(used)                    # the three solveODE overloads, the RHS compiler, ctx binding, the consumers
    assert known >= 0.9 * checked, (known, checked)         # the reader above understands (nearly) every argument the shim writes


def test_the_reader_catches_the_mistakes_it_is_there_for():
    sigs = _bindings()
    src = _blank_strings_and_comments('''
proc p(y0: OdeBatch, tspan: openArray[float]) =
  var ts = @tspan
  var ny = newSeq[int32](y0.n)
  var opt = options.toC
  discard nnhip_ode_time_grid(addr opt, addr ts[0], ts.len.cint, nil, addr ny[0])
''')
    (name, pos, args, line), = list(_calls(src))
    params, _ = sigs[name]
    assert len(args) == len(params) == 5
    kinds = [_kind(a, src[:pos], sigs) for a in args]
    assert kinds[0] == ("pointer", "NnhipOptions") and kinds[1] == ("pointer", "cdouble") and kinds[2] == ("int", "cint") and kinds[3] == ("nil", None)
    assert kinds[4] == ("pointer", "int32") and not _fits(*kinds[4], params[4][1])          # n_t_out is ptr cint: an int32 seq does not fit
    assert not _fits("int", "int64", "cint") and not _fits("cstring", None, "ptr cdouble") and not _fits("nil", None, "cint")
    assert _fits("nil", None, "ptr cdouble") and _fits("int", "literal", "int64") and _fits("pointer", None, "ptr cint")


def _violations(src, sigs):
    bad = []
    for name, pos, args, line in _calls(src):
        params, _ = sigs[name]
        if len(args) != len(params):
            bad.append((line, name, "arity")); continue
        for (pname, ptype), arg in zip(params, args):
            if not _fits(*_kind(arg, src[:pos], sigs), ptype):
                bad.append((line, name, pname))
    return bad


def test_mutations_of_the_real_shim_are_caught():
    """The shim as it is has no violation; with the last argument of a call dropped, or two neighbouring arguments of different kinds swapped, it has."""
    sigs = _bindings()
    src = _blank_strings_and_comments(open(os.path.join(NIM, "numericalnim_hip.nim")).read())
    assert _violations(src, sigs) == []
    dropped = swapped = 0
    for name, pos, args, line in list(_calls(src)):
        if len(args) < 3:
            continue
        open_idx = src.index("(", pos)
        end = _balanced(src, open_idx)
        inner = src[open_idx + 1:end]
        cut = inner.rstrip()
        last = len(cut) - len(args[-1])
        mutated = src[:open_idx + 1] + cut[:last].rstrip().rstrip(",") + src[end:]
        assert any(v[1] == name and v[2] == "arity" for v in _violations(mutated, sigs)), (name, line)
        dropped += 1
        kinds = [_kind(a, src[:pos], sigs)[0] for a in args]
        for k in range(len(args) - 1):
            pair = {kinds[k], kinds[k + 1]}
            if pair in ({"pointer", "int"}, {"cstring", "int"}, {"pointer", "float"}):
                sw = list(args); sw[k], sw[k + 1] = sw[k + 1], sw[k]
                mutated = src[:open_idx + 1] + ", ".join(sw) + src[end:]
                assert any(v[1] == name for v in _violations(mutated, sigs)), (name, line, args[k], args[k + 1])
                swapped += 1
                break
    assert dropped >= 15 and swapped >= 12, (dropped, swapped)


def _shim_procs():
    """exported procs of numericalnim_hip.nim: name -> list of (required, total, parameter names) per overload"""
    src = _blank_strings_and_comments(open(os.path.join(NIM, "numericalnim_hip.nim")).read())
    procs = {}
    for m in re.finditer(r"^proc\s+(\w+)\*?\s*\(", src, re.M):
        end = _balanced(src, m.end() - 1)
        names, required = [], 0
        for p in _split_top(src[m.end():end].replace(";", ",")):
            mm = re.match(r"([\w, ]+?)\s*(?::\s*([^=]+?))?\s*(=.*)?$", p, re.S)
            group = [x.strip() for x in mm.group(1).split(",")]
            names += group
            if not mm.group(3):
                required += len(group)
        procs.setdefault(m.group(1), []).append((required, len(names), names))
    return procs


def test_the_macro_file_calls_the_shim_with_arguments_the_shim_declares():
    procs = _shim_procs()
    assert {"solveODE", "rhsFromSource", "rhsFromSourceCtx", "bindCtx", "cumtrapz", "newHermiteSpline"} <= set(procs)
    assert len(procs["solveODE"]) == 3                                                  # batch, per-IVP tEnd, per-IVP tspans
    src = _blank_strings_and_comments(open(os.path.join(NIM, "rhs_macro.nim")).read())
    seen = 0
    for name, pos, args, line in _calls(src, prefix="|".join(sorted(procs, key=len, reverse=True))):
        if re.search(r"\bproc\s+$|\bmacro\s+$|\btemplate\s+$", src[max(0, pos - 12):pos]):
            continue                                                                     # a definition, not a call
        positional = [a for a in args if not re.match(r"\w+\s*=[^=]", a)]
        named = [re.match(r"(\w+)\s*=", a).group(1) for a in args if re.match(r"\w+\s*=[^=]", a)]
        ok = any(req <= len(positional) + len([n for n in named if n in names]) and len(args) <= total and all(n in names for n in named)
                 for req, total, names in procs[name])
        assert ok, f"rhs_macro.nim:{line}: {name}({', '.join(args)}) matches none of {procs[name]}"
        seen += 1
    assert seen >= 2

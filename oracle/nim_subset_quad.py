"""nim_subset_quad.py — TEST INFRASTRUCTURE ONLY (like everything under oracle/).

Extends the Nim-subset interpreter of nim_subset.py by what the §8-f4 consumers of the ODE path are written in, so that THEIR text runs too:

  integrate.nim    cumtrapz (discrete :121-135, function form :138-175), cumsimpson (discrete :330-379, function form :381-400)
  utils.nim        hermiteInterpolate :282-312, hermiteSpline :273-279, sortAndTrimDataset / sortDataset / removeDuplicates / findDuplicates /
                   getIndexTable / delete :332-420, linspace
  interpolate.nim  newHermiteSpline (both forms :218-264), eval_hermitespline / derivEval_hermitespline :187-216, findInterval :114-115,
                   eval / derivEval with every ExtrapolateKind :299-390 (scalar and openArray forms)

read from /root/reference at run time; nothing of the reference is copied here — this module knows more of the LANGUAGE, not the program.

Language added (Nim manual): `block NAME:` / `break NAME`; `for i, x in s`; command-call syntax (`s.add x`, `assert cond, msg`); `if` as an
expression, also as the indented right-hand side of `let (a, b) =` and of `return`; `^n` backwards indices and `a .. ^b` slices; `mod` / `div`;
assignment to object fields and to tuples of targets; `tuple[x: ..., y: ...]` results whose fields are assigned one by one; a `when` branch whose
body is only a pragma; generic parameters bound to the argument's run-time type (`when U is Missing`); seq value semantics where the path
relies on them (`@s` copies, nested seqs included).

Restated instead of interpreted, with citations (std lib / third party, none of it numericalnim's): algorithm.sort / sorted / isSorted /
lowerBound / SortOrder, sequtils.zip / unzip / toSeq, system.delete / clamp / newSeq / newSeqOfCap / toInt, tables (a dict: the path never
depends on iteration order), and arraymancer's Tensor as far as the Hermite spline touches it (a rows x cols store: newTensorUninit, `[i, _] =`,
`[i, j]`, toTensor, reshape).
"""
import os

from oracle.nim_subset import (Alias, Env, Interp, NimError, NimObj, Parser, Routine, Vec, _BIN_PREC, _BUILTINS, _Break, _Return, fdiv, is_vector,
                               norm_ident, pick_overload, tokenize, REFERENCE_ROOT)

_BIN_PREC.setdefault("mod", 9)  # system.`mod` / `div`: multiplicative precedence (operators that are keywords)
_BIN_PREC.setdefault("div", 9)


class TWord(str):
    """the leading word of a type as the base parser keeps it, plus the whole type text (`full`) and, for tuple types, the field names"""
    full = ""
    fields = ()


def quad_tokenize(text):
    return [("kw", t[1], t[2]) if (t[0] == "id" and t[1] in ("mod", "div", "block")) else t for t in tokenize(text)]


class BackIndex:
    def __init__(self, n):
        self.n = n


class Tensor2:
    """arraymancer Tensor[T] of rank 2, as far as newHermiteSpline / eval_hermitespline use it (interpolate.nim:203-206, 236-238)"""
    def __init__(self, rows, cols):
        self.rows, self.cols, self.d = rows, cols, [[None] * cols for _ in range(rows)]


class _NamedBreak(Exception):
    def __init__(self, label):
        self.label = label


class QuadParser(Parser):
    def skip_type(self, stop_ops):
        start = self.p
        first = Parser.skip_type(self, stop_ops)
        if first is None:
            return None
        toks = self.t[start:self.p]
        if first == "var" and len(toks) > 1 and toks[1][0] in ("id", "kw"):  # `s: var seq[T]`: overloads are told apart by what follows `var`
            first = toks[1][1]
        w = TWord(first)
        w.full = "".join(str(t[1]) for t in toks if t[1] is not None)
        if first == "tuple":  # tuple[x: seq[Tx], y: seq[seq[Ty]]]: the names in front of a ':' at bracket depth 1
            depth, names = 0, []
            for k, t in enumerate(toks):
                if t[0] == "op" and t[1] in "[(": depth += 1
                elif t[0] == "op" and t[1] in "])": depth -= 1
                elif t[0] == "id" and depth == 1 and k + 1 < len(toks) and toks[k + 1][1] == ":": names.append(t[1])
            w.fields = tuple(names)
        return w

    def block_or_stmt(self):
        if self.accept("nl"):
            if not self.at("indent"):
                return []  # a branch whose body was only a pragma ({.error: ...}): nothing left of it
            self.next()
            return self.block()
        return [self.stmt()]

    def value_expr(self):
        """an expression, possibly written on the following, indented lines (`let x =` NL INDENT if ... else ... DEDENT)"""
        if self.at("nl") and self.peek(1)[0] == "indent":
            self.next(); self.next()
            e = self.expr()
            while self.accept("nl"): pass
            self.expect("dedent")
            return e
        return self.expr()

    def branch_value(self):
        """after `cond:` / `else:` of an if-expression: the value on the same line, or alone on the next, indented one"""
        if self.at("nl") and self.peek(1)[0] == "indent":
            self.next(); self.next()
            e = self.expr()
            while self.accept("nl"): pass
            self.expect("dedent")
            return e
        return self.expr()

    def if_expr(self):
        branches, other = [], None
        cond = self.expr()
        self.expect("op", ":")
        branches.append((cond, self.branch_value()))
        while True:
            save = self.p
            while self.accept("nl"): pass
            if self.accept("kw", "elif"):
                cond = self.expr(); self.expect("op", ":")
                branches.append((cond, self.branch_value()))
            elif self.accept("kw", "else"):
                self.expect("op", ":")
                other = self.branch_value(); break
            else:
                self.p = save; break
        return ("ifexpr", branches, other)

    def stmt(self):
        tok = self.peek()
        if tok[0] == "kw" and tok[1] == "block" and self.peek(1)[0] == "id":
            self.next(); label = self.next()[1]
            return ("blockstmt", label, self.colon_block())
        if tok[0] == "kw" and tok[1] == "break" and self.peek(1)[0] == "id":
            self.next(); return ("break", self.next()[1])
        if tok[0] == "kw" and tok[1] == "for" and self.peek(2)[0] == "op" and self.peek(2)[1] == ",":
            self.next()
            a = self.expect("id")[1]; self.expect("op", ","); b = self.expect("id")[1]
            self.expect("kw", "in")
            return ("for2", a, b, self.expr(), self.colon_block())
        if tok[0] == "kw" and tok[1] == "return" and self.peek(1)[0] == "nl" and self.peek(2)[0] == "indent":
            self.next()
            return ("return", self.value_expr())
        if tok[0] == "kw" and tok[1] in ("let", "var") and self.peek(1)[0] == "op" and self.peek(1)[1] == "(":
            self.next()
            lhs = self.primary()
            self.expect("op", "=")
            return ("destructure_decl", lhs, self.value_expr())
        if tok[0] == "id" and tok[1] == "assert" and not (self.peek(1)[0] == "op" and self.peek(1)[1] == "("):
            self.next()
            cond = self.expr()
            msg = self.expr() if self.accept("op", ",") else None
            return ("assert", cond, msg)
        st = Parser.stmt(self)
        if st[0] == "expr" and st[1][0] in ("dot", "id") and self.peek()[0] in ("id", "num", "str") :  # command call: `s.add x`
            arg = self.expr()
            return ("expr", ("call", st[1], [(None, None, arg)]))
        return st

    def unary(self):
        tok = self.peek()
        if tok[0] == "op" and tok[1] == "^":
            self.next(); return ("backidx", self.unary())
        return Parser.unary(self)

    def primary(self):
        tok = self.peek()
        if tok[0] == "kw" and tok[1] == "if":
            self.next(); return self.if_expr()
        return Parser.primary(self)


def _nim_cmp(a, b):
    """system.cmp: -1 / 0 / 1 from `<` and `==`; tuples field by field (the (x, index) pairs of sortDataset, utils.nim:394-395)"""
    if isinstance(a, tuple):
        for x, y in zip(a, b):
            c = _nim_cmp(x, y)
            if c != 0: return c
        return 0
    if a < b: return -1
    if a == b: return 0
    return 1


def _merge_sort(s, order):
    """algorithm.sort: a stable merge sort; SortOrder.Descending flips the comparison's sign"""
    import functools
    sign = -1 if order == "Descending" else 1
    return sorted(s, key=functools.cmp_to_key(lambda a, b: sign * _nim_cmp(a, b)))  # Python's sort is stable too


def _sort_in_place(s, order="Ascending"):
    s[:] = _merge_sort(s, order)


def _is_sorted(s, order="Ascending"):  # algorithm.isSorted: cmp(a[i], a[i+1]) * order <= 0 for every neighbour pair
    sign = -1 if order == "Descending" else 1
    return all(sign * _nim_cmp(s[i], s[i + 1]) <= 0 for i in range(len(s) - 1))


def _lower_bound(a, key):  # algorithm.lowerBound: the first index whose element is not < key (binary search with cmp)
    lo, count = 0, len(a)
    while count != 0:
        step = count >> 1
        mid = lo + step
        if _nim_cmp(a[mid], key) < 0:
            lo = mid + 1
            count -= step + 1
        else:
            count = step
    return lo


def _to_int(x):  # system.toInt(float): "rounds half away from zero" — implemented as `if f >= 0: int(f + 0.5) else: int(f - 0.5)`
    return int(x + 0.5) if x >= 0 else int(x - 0.5)


def _deep(v):
    return [_deep(x) for x in v] if isinstance(v, list) else v


def _delete(s, i):  # system.delete(s, i) for one index (utils.nim:332-339 is the reference's own seq-of-indices form, interpreted)
    del s[i]


def _to_tensor(v):
    t = Tensor2(1, len(v))
    t.d[0] = list(v)
    return t


def _reshape(t, r, c):
    flat = [x for row in t.d for x in row]
    assert r * c == len(flat)
    out = Tensor2(r, c)
    out.d = [flat[i * c:(i + 1) * c] for i in range(r)]
    return out


_QUAD_BUILTINS = {
    "zip": lambda a, b: [(x, y) for x, y in zip(a, b)],               # sequtils.zip: min length
    "unzip": lambda s: ([p[0] for p in s], [p[1] for p in s]),         # sequtils.unzip
    "toSeq": lambda r: list(r),
    "sort": _sort_in_place, "sorted": lambda s, order="Ascending": _merge_sort(list(s), order),
    "isSorted": _is_sorted, "lowerBound": _lower_bound,
    "clamp": lambda x, a, b: a if x < a else (b if x > b else x),     # system.clamp
    "Ascending": "Ascending", "Descending": "Descending",
    "Constant": "Constant", "Edge": "Edge", "Linear": "Linear", "Native": "Native", "Error": "Error",  # ExtrapolateKind (interpolate.nim:85-86): enum values
    "delete": _delete,
    "newSeq": lambda n=0: [0.0] * int(n),                             # (zero-initialised; every element the path reads is written first)
    "newSeqOfCap": lambda n=0: [],
    "values": lambda d: list(d.values()),
    "toInt": _to_int,
    "newTensorUninit": lambda r, c: Tensor2(int(r), int(c)),
    "toTensor": _to_tensor, "reshape": _reshape,
    "initTable": lambda: {},
}


class QuadInterp(Interp):
    parser_class = QuadParser
    tokenizer = staticmethod(quad_tokenize)

    def __init__(self):
        Interp.__init__(self)
        self.extra = {norm_ident(k): v for k, v in _QUAD_BUILTINS.items()}

    # ---- names ----
    def resolve_callable(self, name, first_arg, env):
        try:
            found = Interp.resolve_callable(self, name, first_arg, env)
        except NimError:
            found = None
        if name in self.extra:
            user = found if (isinstance(found, list) and found and isinstance(found[0], Routine)) else None
            if user is None: return self.extra[name]          # the std-lib proc (also where the base has a simpler stand-in: sorted, toInt, newSeq)
            return user                                       # the reference's own overload set; call_value falls back to the std-lib one
        if found is None: raise NimError(f"undeclared identifier: {name}")
        return found

    def call_value(self, fn, args, kwargs, env, arg_nodes=None, block=None):
        if isinstance(fn, list) and fn and isinstance(fn[0], Routine):
            r = pick_overload(fn, args)
            if r is None and fn[0].name in self.extra:
                return self.extra[fn[0].name](*args, **kwargs)
            if r is not None:
                return self.invoke(r, args, kwargs, caller_env=env, block=block, arg_nodes=arg_nodes)
        return Interp.call_value(self, fn, args, kwargs, env, arg_nodes=arg_nodes, block=block)

    def invoke(self, r, args, kwargs, caller_env=None, block=None, arg_nodes=None):
        # generic parameters named in `when ... is ...` (interpolate.nim:314, :357): bound to the run-time type of the argument they type
        bind = {}
        vals = list(args) + [None] * (len(r.params) - len(args))
        for i, (pname, tword, default) in enumerate(r.params):
            if tword is not None and len(str(tword)) == 1 and str(tword).isupper():
                v = vals[i] if i < len(args) else kwargs.get(pname, "__default__")
                if v == "__default__":
                    v = self.eval(default, r.env) if default is not None else None
                bind[str(tword)] = "Missing" if v is None else ("float" if isinstance(v, float) else ("int" if isinstance(v, int) else type(v).__name__))
        if "U" in bind and "T" not in bind and bind["U"] != "Missing":
            bind["T"] = bind["U"]  # `when not T is U: {.error.}` (interpolate.nim:317-318): a call that compiles has T == U
        if not bind:
            return Interp.invoke(self, r, args, kwargs, caller_env=caller_env, block=block, arg_nodes=arg_nodes)
        # (the base invoke creates the scope itself: the bindings go into a scope between the routine's defining one and that)
        shim = Routine(r.kind, r.name, r.params, r.rtype, r.body, Env(r.env))
        shim.env.vars.update(bind)
        return Interp.invoke(self, shim, args, kwargs, caller_env=caller_env, block=block, arg_nodes=arg_nodes)

    # ---- statements ----
    def exec_stmt(self, st, env):
        k = st[0]
        if k == "blockstmt":
            try:
                self.exec_block(st[2], env)
            except _NamedBreak as b:
                if b.label != st[1]: raise
            return None
        if k == "break" and len(st) > 1:
            raise _NamedBreak(st[1])
        if k == "for2":
            try:
                for i, v in enumerate(self.eval(st[3], env)):
                    scope = Env(env); scope.vars[st[1]] = i; scope.vars[st[2]] = v
                    self.exec_block(st[4], scope, new_scope=False)
            except _Break:
                pass
            return None
        if k == "assert":
            if not self.eval(st[1], env):
                raise NimError("AssertionDefect: " + (str(self.eval(st[2], env)) if st[2] is not None else ""))
            return None
        if k == "decl":  # seq value semantics: `var a = b` copies
            for names, tword, init in st[1]:
                for n in names:
                    env.vars[n] = _deep(self.eval(init, env)) if init is not None else self.default_of(tword)
            return None
        return Interp.exec_stmt(self, st, env)

    @staticmethod
    def default_of(tword):
        from oracle.nim_subset import _default_for
        if tword is not None and str(tword) == "Table": return {}
        return _default_for(tword)

    def assign(self, lhs, value, env):
        if lhs[0] == "dot":
            obj = self.eval(lhs[1], env)
            if not (isinstance(obj, NimObj) and obj.has(lhs[2])): raise NimError(f"no field {lhs[2]} to assign to")
            obj.values[obj.names.index(lhs[2])] = value
            return
        if lhs[0] == "idx":
            base = self.eval(lhs[1], env)
            if isinstance(base, Tensor2):
                i, j = lhs[2][0][2], lhs[2][1][2]
                if j == ("id", "_"):  # coeffs[i, _] = a 1 x cols tensor
                    base.d[self.eval(i, env)] = list(value.d[0])
                else:
                    base.d[self.eval(i, env)][self.eval(j, env)] = value
                return
            if base is None and lhs[1] == ("id", "result"):  # `result[key] = ...` on a Table result (utils.nim:341-346)
                base = {}
                Interp.assign(self, lhs[1], base, env)
            base[self.index_of(base, self.eval(lhs[2][0][2], env))] = value
            return
        return Interp.assign(self, lhs, value, env)

    @staticmethod
    def index_of(base, i):
        return len(base) - i.n if isinstance(i, BackIndex) else i

    # ---- expressions ----
    def binop(self, op, a, b, env):
        if op == "mod": return a % b if (a >= 0 and b > 0) else int(a - b * int(a / b))   # system.`mod`: truncated (sign of the dividend)
        if op == "div": return a // b if (a >= 0 and b > 0) else int(a / b)
        if op in ("..", "..<") and isinstance(b, BackIndex): return ("slice", a, b, op)
        if op in ("in", "notin") and b is None: return op == "notin"                          # an empty Table result not yet written to
        if op in ("==", "!=") and (isinstance(a, str) or isinstance(b, str)): return (a == b) == (op == "==")
        return Interp.binop(self, op, a, b, env)

    def eval(self, node, env):
        k = node[0]
        if k == "id" and node[1] in self.extra and env.find(node[1]) is None and node[1] not in self.lazy_consts:
            return self.extra[node[1]]
        if k == "backidx":
            return BackIndex(self.eval(node[1], env))
        if k == "ifexpr":
            for cond, val in node[1]:
                if self.eval(cond, env): return self.eval(val, env)
            return self.eval(node[2], env)
        if k == "un" and node[1] == "@":
            v = self.eval(node[2], env)
            if isinstance(v, Tensor2): return v          # `@[a, b].toTensor.reshape(1, 2)`: the parser applied `@` last; it belongs to the array literal
            if isinstance(v, list): return _deep(v)      # `@s` is a copy, nested seqs included (utils.nim:380-384 deletes from the copy)
            return Interp.eval(self, node, env)
        if k == "idx":
            base = self.eval(node[1], env)
            if isinstance(base, Tensor2):
                return base.d[self.eval(node[2][0][2], env)][self.eval(node[2][1][2], env)]
            if isinstance(base, (list, tuple)) and not (base and isinstance(base[0], Routine)):
                i = self.eval(node[2][0][2], env)
                if isinstance(i, BackIndex): return base[len(base) - i.n]
                if isinstance(i, tuple) and i and i[0] == "slice":
                    hi = len(base) - i[2].n
                    return list(base[i[1]:hi + (1 if i[3] == ".." else 0)])
                if isinstance(i, range): return list(base[i.start:i.stop])
                return base[i]
            if isinstance(base, dict):
                return base[self.eval(node[2][0][2], env)]
        if k == "dot":
            recv = self.eval(node[1], env)
            if isinstance(recv, NimObj) and recv.has(node[2]): return recv.get(node[2])
            return self.call_value(self.resolve_callable(node[2], recv, env), [recv], {}, env)
        return Interp.eval(self, node, env)


SRC = os.path.join("src", "numericalnim")
UTILS_PROCS = ["hermiteSpline", "hermiteInterpolate", "sortAndTrimDataset", "sortDataset", "removeDuplicates", "findDuplicates", "getIndexTable", "delete", "linspace"]
INTEGRATE_PROCS = ["cumtrapz", "cumsimpson"]
INTERPOLATE_PROCS = ["findInterval", "eval_hermitespline", "derivEval_hermitespline", "newHermiteSpline", "eval", "derivEval", "missing"]


def load_reference_quad(root=REFERENCE_ROOT):
    """An interpreter holding the f4 consumers' procs, parsed from the reference's text."""
    it = QuadInterp()
    src = os.path.join(root, SRC)
    # utils.nim's sortDataset has a third overload returning seq[(float, T)] (:422) that nothing on this path calls
    it.load(os.path.join(src, "utils.nim"), names=UTILS_PROCS, accept=lambda head: "seq[(float, T)]" not in head)
    it.load(os.path.join(src, "integrate.nim"), names=INTEGRATE_PROCS)
    # interpolate.nim: `eval` / `derivEval` also exist for the 2D / 3D / unstructured interpolators — only the InterpolatorType[T] ones are on the path
    it.load(os.path.join(src, "interpolate.nim"), names=INTERPOLATE_PROCS,
            accept=lambda head: ("eval" not in head.lower().split("(")[0]) or "InterpolatorType[T]" in head or "hermitespline" in head.lower())
    return it

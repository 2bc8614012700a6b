"""nim_subset.py — TEST INFRASTRUCTURE ONLY (like everything under oracle/).

A small tree-walking interpreter for the subset of Nim that numericalnim's explicit-RK ODE path is written in.  It EXECUTES THE
REFERENCE'S OWN SOURCE TEXT — src/numericalnim/ode.nim and the few procs of utils.nim the path calls — read from /root/reference
at run time.  Nothing of the reference is copied into this repository: the interpreter knows the language, not the program.

Why it exists: the reference is pure Nim and this image has no Nim compiler, so `oracle/_ref` cannot be built and the C++ oracle
(oracle/ode_oracle.cpp) would otherwise be pinned only by the reference's analytic known-answer tests.  With this module the procs the
oracle restates (the 14 `*_step` procs incl. their `const` tableaux, `commonAdaptiveMethodCode`, `ODESolver`, `solveODE`,
`newODEoptions`, `hermiteSpline`, `linspace`) are run from the reference's text on the same inputs, and the oracle — and, through the
committed vectors of tests/golden/reference_text_vectors.json, the HIP path on the GPU box — is compared with the result bit for bit.

`Vector[float]` states are the reference's own object too: `newVector`, `[]`, `@`, `+ - * /`, `+. -. *. /.`, unary `-`, `abs`, `sum` -> `norm`,
`size`, `checkVectorSizes` are utils.nim's procs, interpreted (overloads resolved by the arguments' run-time types); the Vec class below is a 20 x
faster stand-in for runs where the state type is not what is being checked (load_reference_ode(interpret_vector=False)) — the two agree bit for bit.

What is NOT interpreted (restated here instead, with citations):
  * the handful of std-lib procs the path calls (system.min/max/abs, math.sqrt/pow/`^`, sequtils.filter/concat, algorithm.sorted/
    reversed, strutils.toLower): `_BUILTINS`.
Floating point: every Nim `float` operation is one Python float operation = one IEEE-754 binary64 operation of the C double CPython is
built on (x86-64 SSE2, no contraction across byte codes), `pow` is this process's libm pow — the arithmetic the reference's C backend
emits.

Language notes that matter for parity (Nim manual): identifiers are compared with the first character case-sensitive and the rest
case- and underscore-insensitive (ode.nim writes both `tNegative` and `tnegative`); unary operators bind tighter than any binary one
(`-1/3` is `(-1)/3`); binary precedence is decided by the operator's first character (`+.` binds like `+`, `*.` and `/.` like `*`);
`^` is right-associative and binds tighter than `*`; `/` on two integer literals is a float division; templates with `untyped`
parameters are substituted in the caller's scope.
"""
import math
import os

REFERENCE_ROOT = "/root/reference"


class NimError(Exception):
    pass


# ----------------------------------------------------------------------------------------------------------------------------------
# lexer
# ----------------------------------------------------------------------------------------------------------------------------------
_KEYWORDS = {"let", "var", "const", "proc", "template", "func", "if", "elif", "else", "while", "break", "return", "case", "of", "raise",
             "discard", "in", "notin", "and", "or", "not", "for", "true", "false", "nil", "when", "is"}
_OPS = ["+.=", "-.=", "*.=", "/.=", "+.", "-.", "*.", "/.", "+=", "-=", "*=", "/=", "==", "<=", ">=", "!=", "..<", "..", "+", "-", "*", "/", "<", ">",
        "=", "^", ".", ",", ":", ";", "(", ")", "[", "]", "{", "}", "@", "&", "$"]


def norm_ident(s):
    return s[0] + s[1:].replace("_", "").lower()


def _strip_comment(line):
    out, i, in_str = [], 0, False
    while i < len(line):
        ch = line[i]
        if in_str:
            out.append(ch)
            if ch == "\\" and i + 1 < len(line):
                out.append(line[i + 1]); i += 1
            elif ch == '"':
                in_str = False
        elif ch == '"':
            in_str = True; out.append(ch)
        elif ch == "#":
            break
        else:
            out.append(ch)
        i += 1
    return "".join(out).rstrip()


def tokenize(text):
    """-> list of (kind, value, line).  kinds: num, str, id, kw, op, nl, indent, dedent, eof"""
    toks, indents, depth = [], [0], 0
    for lineno, raw in enumerate(text.split("\n"), 1):
        line = _strip_comment(raw)
        if not line.strip():
            continue
        col = len(line) - len(line.lstrip(" "))
        if depth == 0:
            if toks:
                last = toks[-1]
                continued = last[0] == "op" and last[1] in ("+", "-", "*", "/", ",", "+.", "*.", "/.")
            else:
                continued = False
            if not continued:
                if toks:
                    toks.append(("nl", None, lineno))
                if col > indents[-1]:
                    indents.append(col); toks.append(("indent", None, lineno))
                else:
                    while col < indents[-1]:
                        indents.pop(); toks.append(("dedent", None, lineno))
        i = col
        while i < len(line):
            ch = line[i]
            if ch == " ":
                i += 1; continue
            if ch.isdigit():
                j, is_float = i, False
                while j < len(line) and (line[j].isdigit() or line[j] == "_"): j += 1
                if j + 1 < len(line) and line[j] == "." and line[j + 1].isdigit():
                    is_float = True; j += 1
                    while j < len(line) and (line[j].isdigit() or line[j] == "_"): j += 1
                if j < len(line) and line[j] in "eE" and (line[j + 1].isdigit() or (line[j + 1] in "+-" and line[j + 2].isdigit())):
                    is_float = True; j += 2
                    while j < len(line) and line[j].isdigit(): j += 1
                lit = line[i:j].replace("_", "")
                if j < len(line) and line[j] == "'":  # type suffix ('f64 ...)
                    k = j + 1
                    while k < len(line) and line[k].isalnum(): k += 1
                    if line[j + 1] == "f": is_float = True
                    j = k
                toks.append(("num", float(lit) if is_float else int(lit), lineno)); i = j; continue
            if ch.isalpha() or ch == "_":
                j = i
                while j < len(line) and (line[j].isalnum() or line[j] == "_"): j += 1
                word = line[i:j]
                toks.append(("kw", word, lineno) if word in _KEYWORDS else ("id", norm_ident(word), lineno)); i = j; continue
            if ch == "`":
                j = line.index("`", i + 1)
                word = line[i + 1:j]
                toks.append(("id", norm_ident(word) if (word[0].isalpha()) else word, lineno)); i = j + 1; continue
            if ch == '"':
                j, buf = i + 1, []
                while line[j] != '"':
                    if line[j] == "\\": buf.append(line[j + 1]); j += 2
                    else: buf.append(line[j]); j += 1
                toks.append(("str", "".join(buf), lineno)); i = j + 1; continue
            if line.startswith("{.", i):  # pragma: ignored
                i = line.index(".}", i) + 2; continue
            for op in _OPS:
                if line.startswith(op, i):
                    if op in "([{": depth += 1
                    elif op in ")]}": depth -= 1
                    toks.append(("op", op, lineno)); i += len(op); break
            else:
                raise NimError(f"line {lineno}: cannot tokenize {line[i:]!r}")
    toks.append(("nl", None, 0))
    while len(indents) > 1:
        indents.pop(); toks.append(("dedent", None, 0))
    toks.append(("eof", None, 0))
    return toks


# ----------------------------------------------------------------------------------------------------------------------------------
# parser -> tuples
# ----------------------------------------------------------------------------------------------------------------------------------
_BIN_PREC = {"^": 10, "*": 9, "/": 9, "*.": 9, "/.": 9, "+": 8, "-": 8, "+.": 8, "-.": 8, "&": 7, "..": 6, "..<": 6,
             "==": 5, "<=": 5, "<": 5, ">=": 5, ">": 5, "!=": 5, "in": 5, "notin": 5, "is": 5, "and": 4, "or": 3}


class Parser:
    def __init__(self, toks):
        self.t, self.p = toks, 0

    def peek(self, k=0):
        return self.t[self.p + k]

    def next(self):
        tok = self.t[self.p]; self.p += 1; return tok

    def at(self, kind, val=None):
        tok = self.t[self.p]
        return tok[0] == kind and (val is None or tok[1] == val)

    def accept(self, kind, val=None):
        if self.at(kind, val):
            return self.next()
        return None

    def expect(self, kind, val=None):
        if not self.at(kind, val):
            tok = self.t[self.p]
            raise NimError(f"line {tok[2]}: expected {kind} {val!r}, got {tok[0]} {tok[1]!r}")
        return self.next()

    # ---- types are skipped; only the leading word is kept (for default initialisation) ----
    def skip_type(self, stop_ops):
        first, depth = None, 0
        while True:
            tok = self.peek()
            if tok[0] in ("nl", "eof", "indent", "dedent") and depth == 0:
                break
            if tok[0] == "op":
                if depth == 0 and tok[1] in stop_ops: break
                if tok[1] in "([{": depth += 1
                elif tok[1] in ")]}":
                    if depth == 0: break
                    depth -= 1
            if first is None and tok[0] in ("id", "kw"): first = tok[1]
            elif first is None and tok[0] == "op" and tok[1] == "(": first = "tuple"
            self.next()
        return first

    def params(self):
        """after '(' ... consumes ')'.  -> [(name, type_word, default_expr|None)]"""
        out = []
        while not self.at("op", ")"):
            names = [self.expect("id")[1]]
            while self.accept("op", ","):
                names.append(self.expect("id")[1])
            tword, default = None, None
            if self.accept("op", ":"):
                tword = self.skip_type({",", ";", "=", ")"})
            if self.accept("op", "="):
                default = self.expr()
            for n in names:
                out.append((n, tword, default))
            if not (self.accept("op", ",") or self.accept("op", ";")):
                break
        self.expect("op", ")")
        return out

    def routine(self, kind):
        """after the proc/template keyword"""
        name = self.next()[1]
        self.accept("op", "*")
        if self.at("op", "["):  # generic parameters
            depth = 0
            while True:
                tok = self.next()
                if tok[1] == "[": depth += 1
                elif tok[1] == "]":
                    depth -= 1
                    if depth == 0: break
        self.expect("op", "(")
        params = self.params()
        rtype = None
        if self.accept("op", ":"):
            rtype = self.skip_type({"="})
        self.expect("op", "=")
        body = self.block_or_stmt()
        return (kind, name, params, rtype, body)

    def block_or_stmt(self):
        if self.accept("nl"):
            self.expect("indent")
            return self.block()
        return [self.stmt()]

    def block(self):
        out = []
        while not self.at("dedent") and not self.at("eof"):
            if self.accept("nl"): continue
            out.append(self.stmt())
        self.accept("dedent")
        return out

    def colon_block(self):
        self.expect("op", ":")
        return self.block_or_stmt()

    def decl_line(self):
        names = [self.expect("id")[1]]
        self.accept("op", "*")
        while self.accept("op", ","):
            names.append(self.expect("id")[1]); self.accept("op", "*")
        tword, init = None, None
        if self.accept("op", ":"):
            tword = self.skip_type({"="})
        if self.accept("op", "="):
            init = self.expr()
        return (names, tword, init)

    def stmt(self):
        tok = self.peek()
        if tok[0] == "kw":
            kw = tok[1]
            if kw in ("let", "var", "const"):
                self.next()
                decls = []
                if self.accept("nl"):
                    self.expect("indent")
                    while not self.at("dedent"):
                        if self.accept("nl"): continue
                        decls.append(self.decl_line())
                    self.expect("dedent")
                elif self.at("op", "("):  # var (a, b) = ...
                    lhs = self.primary()
                    self.expect("op", "=")
                    return ("destructure_decl", lhs, self.expr())
                else:
                    decls.append(self.decl_line())
                return ("decl", decls, kw)
            if kw == "if" or kw == "when":
                self.next()
                branches, other = [(self.expr(), self.colon_block())], None
                while True:
                    save = self.p
                    while self.accept("nl"): pass
                    if self.accept("kw", "elif"):
                        branches.append((self.expr(), self.colon_block()))
                    elif self.accept("kw", "else"):
                        other = self.colon_block(); break
                    else:
                        self.p = save; break
                return ("if", branches, other)
            if kw == "while":
                self.next()
                return ("while", self.expr(), self.colon_block())
            if kw == "for":
                self.next()
                var = self.expect("id")[1]
                self.expect("kw", "in")
                return ("for", var, self.expr(), self.colon_block())
            if kw == "break":
                self.next(); return ("break",)
            if kw == "discard":
                self.next()
                return ("expr", self.expr()) if not self.at("nl") else ("pass",)
            if kw == "return":
                self.next()
                return ("return", None if self.at("nl") or self.at("dedent") else self.expr())
            if kw == "raise":
                self.next(); return ("raise", self.expr())
            if kw == "case":
                self.next()
                subject = self.expr()
                self.accept("op", ":")
                self.expect("nl")
                indented = self.accept("indent") is not None
                branches, other = [], None
                while True:
                    while self.accept("nl"): pass
                    if self.accept("kw", "of"):
                        vals = [self.expr()]
                        while self.accept("op", ","): vals.append(self.expr())
                        branches.append((vals, self.colon_block()))
                    elif self.accept("kw", "else"):
                        other = self.colon_block()
                    else:
                        break
                if indented: self.expect("dedent")
                return ("case", subject, branches, other)
            if kw in ("proc", "template", "func") and self.peek(1)[0] == "id":
                self.next(); return self.routine("proc" if kw != "template" else "template")
        e = self.expr()
        tok = self.peek()
        if tok[0] == "op" and tok[1] in ("=", "+=", "-=", "*=", "/="):
            self.next()
            return ("assign", tok[1], e, self.expr())
        if tok[0] == "op" and tok[1] == ":" and e[0] == "call":  # template call with a trailing block
            return ("blockcall", e, self.colon_block())
        return ("expr", e)

    # ---- expressions ----
    def expr(self, min_prec=0):
        lhs = self.unary()
        while True:
            tok = self.peek()
            op = tok[1] if tok[0] in ("op", "kw") else None
            prec = _BIN_PREC.get(op)
            if prec is None or prec < min_prec: return lhs
            self.next()
            rhs = self.expr(prec if op == "^" else prec + 1)  # ^ is right-associative
            lhs = ("bin", op, lhs, rhs)

    def unary(self):
        tok = self.peek()
        if tok[0] == "op" and tok[1] in ("-", "+"):
            self.next(); return ("un", tok[1], self.unary())
        if tok[0] == "kw" and tok[1] == "not":
            self.next(); return ("un", "not", self.unary())
        if tok[0] == "op" and tok[1] == "@":
            self.next(); return ("un", "@", self.unary())
        if tok[0] == "op" and tok[1] == "&":
            self.next(); return self.unary()  # strformat: only reached on the error path; the string is kept as written
        return self.primary()

    def call_args(self, closer):
        args = []
        while not self.at("op", closer):
            if self.peek()[0] == "id" and self.peek(1)[0] == "op" and self.peek(1)[1] in ("=", ":") and not (self.peek(1)[1] == "=" and self.peek(2)[1] == "="):
                name = self.next()[1]; sep = self.next()[1]
                args.append((name, sep, self.expr()))
            else:
                args.append((None, None, self.expr()))
            if not self.accept("op", ","): break
        self.expect("op", closer)
        return args

    def primary(self):
        tok = self.next()
        if tok[0] == "num": node = ("num", tok[1])
        elif tok[0] == "str": node = ("str", tok[1])
        elif tok[0] == "id": node = ("id", tok[1])
        elif tok[0] == "kw" and tok[1] in ("true", "false"): node = ("num", tok[1] == "true")
        elif tok[0] == "kw" and tok[1] == "nil": node = ("nil",)
        elif tok[0] == "kw" and tok[1] == "proc":  # anonymous proc: proc(params): T = expr
            self.expect("op", "(")
            params = self.params()
            if self.accept("op", ":"): self.skip_type({"="})
            self.expect("op", "=")
            node = ("lambda", params, self.expr())
        elif tok[0] == "op" and tok[1] == "(":
            items = self.call_args(")")
            if len(items) == 1 and items[0][0] is None: node = items[0][2]
            else: node = ("tuple", items)
        elif tok[0] == "op" and tok[1] == "[":
            node = ("seq", [a[2] for a in self.call_args("]")])
        else:
            raise NimError(f"line {tok[2]}: unexpected {tok[0]} {tok[1]!r}")
        while True:
            tok = self.peek()
            if tok[0] == "op" and tok[1] == ".":
                self.next(); node = ("dot", node, self.next()[1])
            elif tok[0] == "op" and tok[1] == "(":
                self.next(); node = ("call", node, self.call_args(")"))
            elif tok[0] == "op" and tok[1] == "[":
                self.next(); node = ("idx", node, self.call_args("]"))
            else:
                return node


# ----------------------------------------------------------------------------------------------------------------------------------
# values
# ----------------------------------------------------------------------------------------------------------------------------------
def fdiv(a, b):
    try:
        return a / b
    except ZeroDivisionError:  # IEEE semantics
        a, b = float(a), float(b)
        if a != a or a == 0.0: return float("nan")
        return math.copysign(float("inf"), a) * math.copysign(1.0, b)


class Vec:
    """utils.nim Vector[float]: immutable here; every operator is the element-wise loop of utils.nim:59-224 (operand order as written there)."""
    __slots__ = ("c",)

    def __init__(self, comps):
        self.c = [float(x) for x in comps]

    def _zip(self, o, f):
        if len(self.c) != len(o.c): raise NimError("Vectors must have the same size.")  # utils.nim:22-26
        return Vec([f(a, b) for a, b in zip(self.c, o.c)])

    def binop(self, op, other, swapped):
        if isinstance(other, Vec):
            l, r = (other, self) if swapped else (self, other)
            if op in ("+",): return l._zip(r, lambda a, b: a + b)          # :59-64
            if op in ("-",): return l._zip(r, lambda a, b: a - b)          # :113-118
            if op == "*.": return l._zip(r, lambda a, b: a * b)            # :186-191
            if op == "/.": return l._zip(r, lambda a, b: fdiv(a, b))       # :192-197
            raise NimError(f"Vector {op} Vector is not on the path")
        d = other
        if op in ("+", "+."): return Vec([a + d for a in self.c])          # :66-83 (both operand orders compute v1[i] + d)
        if op == "*": return Vec([a * d for a in self.c])                  # :171-180 (both orders compute v1[i] * d)
        if op == "/" and not swapped: return Vec([fdiv(a, d) for a in self.c])  # :166-170
        if op in ("-", "-."): return Vec([(d - a) if swapped else (a - d) for a in self.c])  # :120-131
        raise NimError(f"Vector {op} float (swapped={swapped}) is not on the path")

    def neg(self):
        return Vec([-a for a in self.c])  # utils.nim:214-218

    def __eq__(self, o):
        return isinstance(o, Vec) and self.c == o.c

    def __repr__(self):
        return f"Vec({self.c})"


class NimObj:
    def __init__(self, names, values, tname=None):
        self.names, self.values, self.tname = list(names), list(values), tname

    def get(self, n):
        return self.values[self.names.index(n)]

    def has(self, n):
        return n in self.names


def is_vector(v):
    """an object built by the reference's own `Vector[T](components: ..., len: ...)` (utils.nim:14-20), interpreted"""
    return isinstance(v, NimObj) and v.tname == "Vector"


def _arg_fits(tword, a):
    """How well a run-time value fits a parameter whose type starts with `tword` (overload resolution, reduced to what utils.nim / ode.nim need):
    0 = not at all, higher = more specific."""
    if tword in (None, "T", "auto", "untyped", "typed"): return 1
    if is_vector(a): return 3 if tword == "Vector" else (2 if tword == "var" else 0)
    if isinstance(a, bool): return 3 if tword == "bool" else 0
    if isinstance(a, int): return 3 if tword in ("int", "Natural") else (2 if tword in ("float", "float64") else 0)
    if isinstance(a, float): return 3 if tword in ("float", "float64") else 0
    if isinstance(a, list):
        if tword not in ("seq", "openArray", "openarray"): return 0
        full = getattr(tword, "full", "")  # seq[seq[Ty]] next to seq[Ty] (utils.nim:385,406): told apart by the first element
        if full.replace(" ", "").lower().startswith(("seq[seq[", "openarray[seq[")): return 4 if (a and isinstance(a[0], list)) else 0
        if full and a and isinstance(a[0], list) and "T" in full: return 2
        return 3
    return 1 if tword not in ("Vector", "float", "float64", "int", "seq", "openArray", "openarray") else 0


def pick_overload(routines, args):
    """The routine of an overload set that accepts `args` (positional), most specific first, declaration order on ties; None if none does."""
    best, best_score = None, -1
    for r in routines:
        if len(args) > len(r.params) or any(p[2] is None for p in r.params[len(args):]): continue
        score = 0
        for a, (pname, tword, _d) in zip(args, r.params):
            f = _arg_fits(tword, a)
            if f == 0: score = -1; break
            score += f
        if score > best_score: best, best_score = r, score
    return best


class Routine:
    def __init__(self, kind, name, params, rtype, body, env):
        self.kind, self.name, self.params, self.rtype, self.body, self.env = kind, name, params, rtype, body, env


class Alias:  # an `untyped` template argument: an AST evaluated where the template was invoked
    def __init__(self, node, env):
        self.node, self.env = node, env


class Env:
    def __init__(self, parent=None):
        self.vars, self.parent = {}, parent

    def find(self, n):
        e = self
        while e is not None:
            if n in e.vars: return e
            e = e.parent
        return None


class _Break(Exception):
    pass


class _Return(Exception):
    pass


def _default_for(tword):
    if getattr(tword, "fields", None):  # `tuple[x: ..., y: ...]` result (a parser that keeps the field names): fields assigned one by one
        return NimObj(list(tword.fields), [None] * len(tword.fields))
    return {"seq": lambda: [], "float": lambda: 0.0, "float64": lambda: 0.0, "int": lambda: 0, "bool": lambda: False}.get(tword, lambda: None)()


def nim_min(*a):
    if len(a) == 1:  # system.min(openArray)
        r = a[0][0]
        for x in a[0][1:]:
            if x < r: r = x
        return r
    x, y = a
    return x if x <= y else y  # system.min: `if x <= y: x else: y`


def nim_max(*a):
    if len(a) == 1:
        r = a[0][0]
        for x in a[0][1:]:
            if r < x: r = x
        return r
    x, y = a
    return x if y <= x else y  # system.max: `if y <= x: x else: y`


def nim_pow_int(x, n):  # math.`^`(x, y: Natural): 0 -> 1, 1 -> x, 2 -> x*x, 3 -> x*x*x, else square-and-multiply
    if n == 0: return 1.0 if isinstance(x, float) else 1
    if n == 1: return x
    if n == 2: return x * x
    if n == 3: return x * x * x
    result, y = (1.0 if isinstance(x, float) else 1), n
    while True:
        if y & 1: result = result * x
        y >>= 1
        if y == 0: break
        x = x * x
    return result


def nim_pow(x, y):
    try:
        return math.pow(x, y)
    except OverflowError:
        return float("inf")
    except ValueError:
        return float("nan")


def nim_sum(v):
    if isinstance(v, Vec):  # utils.nim:243-250 -> norm(v, 1) :233-235 -> math.sum(@v): left to right from 0.0
        r = 0.0
        for x in v.c: r = r + x
        return r
    r = 0.0
    for x in v: r = r + x
    return r


_BUILTINS = {
    "abs": lambda x: Vec([abs(a) for a in x.c]) if isinstance(x, Vec) else abs(x),   # utils.nim:219-223 / system.abs
    "sqrt": lambda x: math.sqrt(x) if x >= 0 else float("nan"),
    "pow": nim_pow, "min": nim_min, "max": nim_max,
    "toFloat": float, "float": float, "float64": float, "toInt": lambda x: int(round(x)), "floor": math.floor,
    "len": lambda s: len(s.c) if isinstance(s, Vec) else len(s), "high": lambda s: len(s) - 1,
    "filter": lambda s, pred: [x for x in s if pred(x)],          # sequtils.filter keeps order
    "reversed": lambda s: list(reversed(s)), "sorted": lambda s: sorted(s),
    "concat": lambda *ss: [x for s in ss for x in s],
    "add": lambda s, x: s.append(x),
    "clone": lambda x: x,                                          # utils.nim:269
    "isNil": lambda x: x is None, "toLower": lambda s: s.lower(),
    "newException": lambda kind, msg: NimError(f"{kind}: {msg}"),
    "newNumContext": lambda: NimObj([], []),
    "ValueError": "ValueError",
    "newVector": lambda comps: Vec(comps),                         # utils.nim:19-20
    "sgn": lambda x: (x > 0) - (x < 0),
    "sum": nim_sum,                                                # math.sum(openArray): `for i in items(x): result = result + i` from 0.0
    "newSeq": lambda n=0: [0.0] * int(n),                          # system.newSeq[T](n): zero-initialised
}
_BUILTINS = {norm_ident(k): v for k, v in _BUILTINS.items()}
_VEC_FIRST = {"size": lambda v: len(v.c), "sum": nim_sum}          # utils.nim:57, :243-250 (overloads chosen by the argument's type)


# ----------------------------------------------------------------------------------------------------------------------------------
# interpreter
# ----------------------------------------------------------------------------------------------------------------------------------
class Interp:
    parser_class = None  # set below (Parser); a subclass that reads more of the language brings its own (nim_subset_quad.py)
    tokenizer = None

    def __init__(self):
        self.globals = Env()
        self.lazy_consts = {}

    # ---- loading: only column-0 `proc` / `template` / `const NAME = expr` declarations with the wanted names are parsed ----
    def load(self, path, names=None, accept=None):
        """accept: optional predicate on a declaration's first line (tells overloads of one name apart: only the accepted ones are parsed)"""
        text = open(path).read()
        chunks, cur = [], None
        for lineno, line in enumerate(text.split("\n"), 1):
            if line and not line[0].isspace() and not line.startswith("#"):
                cur = [lineno, []]; chunks.append(cur)
            if cur is not None: cur[1].append(line)
        wanted = None if names is None else {norm_ident(n) if n[0].isalpha() else n for n in names}
        for start, lines in chunks:
            head = lines[0].split()
            if not head or head[0] not in ("proc", "template", "const", "func") or len(head) < 2: continue
            raw = head[1].split("*")[0].split("[")[0].split("(")[0].strip("`")
            if head[1].startswith("`"): raw = head[1][1:head[1].index("`", 1)]
            key = norm_ident(raw) if raw[:1].isalpha() else raw
            if wanted is not None and key not in wanted: continue
            if accept is not None and not accept(lines[0]): continue
            toks = (self.tokenizer or tokenize)("\n" * (start - 1) + "\n".join(lines))
            ps = (self.parser_class or Parser)(toks)
            while ps.accept("nl"): pass
            node = ps.stmt()
            if node[0] in ("proc", "template"):
                self.globals.vars.setdefault(node[1], []).append(Routine(node[0], node[1], node[2], node[3], node[4], self.globals))
            elif node[0] == "decl":
                for nms, _t, init in node[1]:
                    self.lazy_consts[nms[0]] = init

    def routine(self, name):
        key = norm_ident(name) if name[0].isalpha() else name
        r = self.globals.vars.get(key)
        if not r: raise NimError(f"{name} is not loaded")
        return r[0]

    def call(self, name, *args, **kw):
        key = norm_ident(name) if name[0].isalpha() else name
        cands = self.globals.vars.get(key) or []
        r = (pick_overload(cands, list(args)) if len(cands) > 1 else None) or self.routine(name)
        return self.invoke(r, list(args), {norm_ident(k): v for k, v in kw.items()})

    def consts_of(self, name):
        """The `const` section of a loaded proc, evaluated: {identifier as normalised: value} in declaration order."""
        out = Env(self.globals)
        for st in self.routine(name).body:
            if st[0] == "decl" and st[2] == "const":
                self.exec_stmt(st, out)
            else:
                break
        return dict(out.vars)

    # ---- calls ----
    def invoke(self, r, args, kwargs, caller_env=None, block=None, arg_nodes=None):
        env = Env(r.env)
        if r.kind == "template" and any(p[1] == "untyped" for p in r.params):
            env = Env(caller_env)  # substituted in the caller's scope
        pos = list(args)
        for a in pos:  # the generic parameter T of `proc f*[T](v: Vector[T], ...)`: the element type, for `when T is Vector` (utils.nim:244)
            if is_vector(a):
                comps = a.get("components")
                env.vars["T"] = "Vector" if (comps and is_vector(comps[0])) else "float"
                break
        for i, (pname, tword, default) in enumerate(r.params):
            if tword == "untyped":
                if i < len(pos): env.vars[pname] = Alias(arg_nodes[i], caller_env)
                elif pname in kwargs: env.vars[pname] = kwargs[pname]
                else: env.vars[pname] = Alias(("block", block), caller_env)
                continue
            if i < len(pos): env.vars[pname] = pos[i]
            elif pname in kwargs: env.vars[pname] = kwargs[pname]
            elif default is not None: env.vars[pname] = self.eval(default, r.env)
            else: raise NimError(f"{r.name}: missing argument {pname}")
        if r.kind == "proc":
            if len(r.body) == 1 and r.body[0][0] == "expr":  # expression-bodied proc (`proc size(d: float): int = 1`, anonymous procs)
                return self.eval(r.body[0][1], env)
            env.vars["result"] = _default_for(r.rtype)
            try:
                self.exec_block(r.body, env, new_scope=False)
            except _Return:
                pass
            return env.vars["result"]
        return self.exec_block(r.body, env, new_scope=False, want_value=True)  # template: its last expression is its value

    def call_value(self, fn, args, kwargs, env, arg_nodes=None, block=None):
        if isinstance(fn, list) and fn and isinstance(fn[0], Routine):
            if len(fn) == 1 and fn[0].name not in _BUILTINS:
                fn = fn[0]
            else:
                r = pick_overload(fn, args)
                if r is None:  # no user overload takes these: the std-lib proc of that name (abs(float) next to utils.nim's abs(Vector))
                    if fn[0].name in _BUILTINS: return _BUILTINS[fn[0].name](*args, **kwargs)
                    raise NimError(f"no overload of {fn[0].name} accepts {[type(a).__name__ for a in args]}")
                fn = r
        if isinstance(fn, Routine):
            return self.invoke(fn, args, kwargs, caller_env=env, block=block, arg_nodes=arg_nodes)
        if callable(fn):
            return fn(*args, **kwargs)
        raise NimError(f"not callable: {fn!r}")

    def resolve_callable(self, name, first_arg, env):
        if isinstance(first_arg, Vec) and name in _VEC_FIRST: return _VEC_FIRST[name]
        e = env.find(name)
        if e is not None:
            v = e.vars[name]
            if isinstance(v, Alias): v = self.eval(v.node, v.env)
            return v
        if name in self.lazy_consts: return self.const_value(name)
        if name in _BUILTINS: return _BUILTINS[name]
        raise NimError(f"undeclared identifier: {name}")

    def const_value(self, name):
        init = self.lazy_consts[name]
        if not isinstance(init, tuple) or init[0] != "__value__":
            self.lazy_consts[name] = ("__value__", self.eval(init, self.globals))
        return self.lazy_consts[name][1]

    # ---- statements ----
    def exec_block(self, stmts, env, new_scope=True, want_value=False):
        scope = Env(env) if new_scope else env
        last = None
        for st in stmts:
            last = self.exec_stmt(st, scope)
        return last if want_value else None

    def exec_stmt(self, st, env):
        k = st[0]
        if k == "expr":
            node = st[1]
            if node[0] == "id":  # a bare identifier may be an `untyped` block parameter: spliced into this scope
                e = env.find(node[1])
                if e is not None and isinstance(e.vars[node[1]], Alias) and e.vars[node[1]].node[0] == "block":
                    a = e.vars[node[1]]
                    for s2 in a.node[1]: self.exec_stmt(s2, env)  # same scope: `let error_y` of the body is visible to the template's next lines
                    return None
            return self.eval(node, env)
        if k == "decl":
            for names, tword, init in st[1]:
                for n in names:
                    env.vars[n] = self.eval(init, env) if init is not None else _default_for(tword)
            return None
        if k == "destructure_decl":
            vals = self.eval(st[2], env)
            vals = vals.values if isinstance(vals, NimObj) else vals
            for (_n, _s, target), v in zip(st[1][1], vals): env.vars[target[1]] = v
            return None
        if k == "assign":
            op, lhs, rhs = st[1], st[2], self.eval(st[3], env)
            if op != "=":
                rhs = self.binop(op[:-1], self.eval(lhs, env), rhs, env)
            self.assign(lhs, rhs, env)
            return None
        if k == "if":
            for cond, body in st[1]:
                if self.eval(cond, env):
                    self.exec_block(body, env); return None
            if st[2] is not None: self.exec_block(st[2], env)
            return None
        if k == "while":
            try:
                while self.eval(st[1], env):
                    self.exec_block(st[2], env)
            except _Break:
                pass
            return None
        if k == "for":
            try:
                for v in self.eval(st[2], env):
                    scope = Env(env); scope.vars[st[1]] = v
                    self.exec_block(st[3], scope, new_scope=False)
            except _Break:
                pass
            return None
        if k == "break": raise _Break()
        if k == "return":
            if st[1] is not None:
                self.assign(("id", "result"), self.eval(st[1], env), env)
            raise _Return()
        if k == "raise":
            raise self.eval(st[1], env)
        if k == "case":
            subject = self.eval(st[1], env)
            for vals, body in st[2]:
                if any(self.eval(v, env) == subject for v in vals):
                    self.exec_block(body, env); return None
            if st[3] is not None: self.exec_block(st[3], env)
            return None
        if k == "blockcall":
            call = st[1]
            fn = self.eval(call[1], env)
            args, kwargs, nodes = [], {}, []
            for name, _sep, node in call[2]:
                if name is None:
                    nodes.append(node); args.append(None)  # evaluated lazily (untyped) or below
                else:
                    kwargs[name] = self.eval(node, env)
            r = fn[0] if isinstance(fn, list) else fn
            for i, node in enumerate(nodes):
                if r.params[i][1] != "untyped": args[i] = self.eval(node, env)
            return self.invoke(r, args, kwargs, caller_env=env, block=st[2], arg_nodes=nodes)
        if k in ("proc", "template"):
            env.vars.setdefault(st[1], []).append(Routine(k, st[1], st[2], st[3], st[4], env)); return None
        if k == "pass": return None
        raise NimError(f"statement {k} not supported")

    def assign(self, lhs, value, env):
        if lhs[0] == "id":
            e = env.find(lhs[1])
            if e is None: raise NimError(f"assignment to undeclared {lhs[1]}")
            if isinstance(e.vars[lhs[1]], Alias):
                a = e.vars[lhs[1]]; return self.assign(a.node, value, a.env)
            e.vars[lhs[1]] = value
        elif lhs[0] == "tuple":
            vals = value.values if isinstance(value, NimObj) else value
            for (_n, _s, target), v in zip(lhs[1], vals): self.assign(target, v, env)
        elif lhs[0] == "idx":
            self.eval(lhs[1], env)[self.eval(lhs[2][0][2], env)] = value
        else:
            raise NimError(f"cannot assign to {lhs[0]}")

    # ---- expressions ----
    def user_op(self, op, args):
        """utils.nim's operator procs / templates on its Vector type, from the text: the overload that takes these operands"""
        cands = self.globals.vars.get(op)
        r = pick_overload(cands, args) if cands else None
        if r is None: raise NimError(f"the reference's text (as loaded) has no `{op}` for {[('Vector' if is_vector(a) else type(a).__name__) for a in args]}")
        return self.invoke(r, list(args), {})

    def binop(self, op, a, b, env):
        if is_vector(a) or is_vector(b): return self.user_op(op, [a, b])
        if hasattr(a, "binop"): return a.binop(op, b, False)  # Vec, or a caller-supplied state type (symbolic execution of a step proc)
        if hasattr(b, "binop"): return b.binop(op, a, True)
        if op in ("+.", "*.", "/.", "-."):  # ode.nim:45-52: templates on floats, interpreted from the reference's text when loaded
            user = self.globals.vars.get(op)
            if user: return self.invoke(user[0], [a, b], {})
            op = op[0]
        if op == "+": return a + b
        if op == "-": return a - b
        if op == "*": return a * b
        if op == "/": return fdiv(a, b)
        if op == "^": return nim_pow_int(a, b)
        if op == "==": return a == b
        if op == "!=": return a != b
        if op == "<": return a < b
        if op == "<=": return a <= b
        if op == ">": return a > b
        if op == ">=": return a >= b
        if op == "in": return any(x == a for x in b)
        if op == "notin": return not any(x == a for x in b)
        if op == "..": return range(a, b + 1)
        if op == "..<": return range(a, b)
        raise NimError(f"operator {op} not supported")

    def eval(self, node, env):
        k = node[0]
        if k == "num" or k == "str": return node[1]
        if k == "nil": return None
        if k == "id":
            e = env.find(node[1])
            if e is not None:
                v = e.vars[node[1]]
                if isinstance(v, Alias): return self.eval(v.node, v.env)
                return v
            if node[1] in self.lazy_consts: return self.const_value(node[1])
            if node[1] in _BUILTINS: return _BUILTINS[node[1]]
            raise NimError(f"undeclared identifier: {node[1]}")
        if k == "bin":
            op = node[1]
            if op == "and": return self.eval(node[2], env) and self.eval(node[3], env)
            if op == "or": return self.eval(node[2], env) or self.eval(node[3], env)
            if op == "is":  # `when T is Vector`: T was bound to the element type's name when the generic proc was entered
                return self.eval(node[2], env) == (node[3][1] if node[3][0] == "id" else None)
            return self.binop(op, self.eval(node[2], env), self.eval(node[3], env), env)
        if k == "un":
            v = self.eval(node[2], env)
            if is_vector(v) and node[1] in ("-", "@"): return self.user_op(node[1], [v])  # utils.nim:214-218 / :43
            if node[1] == "-": return v.neg() if hasattr(v, "neg") else -v
            if node[1] == "not": return not v
            if node[1] == "@": return list(v.c) if isinstance(v, Vec) else list(v)
            return v
        if k == "tuple":
            vals = [self.eval(a[2], env) for a in node[1]]
            if node[1] and node[1][0][0] is not None: return NimObj([a[0] for a in node[1]], vals)
            return tuple(vals)
        if k == "seq": return [self.eval(a, env) for a in node[1]]
        if k == "lambda":
            r = Routine("proc", "<anonymous>", node[1], None, [("expr", node[2])], env)
            return lambda *args: self.invoke(r, list(args), {})
        if k == "dot":
            recv = self.eval(node[1], env)
            if isinstance(recv, NimObj) and recv.has(node[2]): return recv.get(node[2])
            return self.call_value(self.resolve_callable(node[2], recv, env), [recv], {}, env)
        if k == "idx":
            base = self.eval(node[1], env)
            if isinstance(base, (Routine,)) or (isinstance(base, list) and base and isinstance(base[0], Routine)) or callable(base):
                return base  # generic instantiation: DOPRI54_step[T], newNumContext[T, float]
            i = self.eval(node[2][0][2], env)
            if is_vector(base): return self.user_op("[]", [base, i])  # utils.nim:29
            return base.c[i] if isinstance(base, Vec) else base[i]
        if k == "call":
            fnode, arglist = node[1], node[2]
            if arglist and arglist[0][1] == ":":  # object construction: ODEoptions(dt: ..., ...), Vector[T](components: ..., len: ...)
                tnode = fnode[1] if fnode[0] == "idx" else fnode
                return NimObj([a[0] for a in arglist], [self.eval(a[2], env) for a in arglist], tname=tnode[1] if tnode[0] == "id" else None)
            args, kwargs, nodes = [], {}, []
            recv_first = None
            if fnode[0] == "dot":  # method-call syntax a.f(x) = f(a, x), unless a has a field f
                recv = self.eval(fnode[1], env)
                if isinstance(recv, NimObj) and recv.has(fnode[2]):
                    fn = recv.get(fnode[2])
                else:
                    recv_first = recv
                    fn = self.resolve_callable(fnode[2], recv, env)
                    args.append(recv); nodes.append(fnode[1])
            elif fnode[0] == "id":
                fn = None
            else:
                fn = self.eval(fnode, env)
            for name, _sep, a in arglist:
                if name is None:
                    args.append(self.eval(a, env)); nodes.append(a)
                else:
                    kwargs[name] = self.eval(a, env)
            if fn is None:
                fn = self.resolve_callable(fnode[1], args[0] if args else None, env)
            return self.call_value(fn, args, kwargs, env, arg_nodes=nodes)
        if k == "block":
            return self.exec_block(node[1], env, new_scope=False)
        raise NimError(f"expression {k} not supported")


# ----------------------------------------------------------------------------------------------------------------------------------
# the reference's ODE path, loaded from its text
# ----------------------------------------------------------------------------------------------------------------------------------
STEP_PROCS = {"heun2": "HEUN2_step", "ralston2": "RALSTON2_step", "kutta3": "KUTTA3_step", "heun3": "HEUN3_step", "ralston3": "RALSTON3_step",
              "ssprk3": "SSPRK3_step", "ralston4": "RALSTON4_step", "kutta4": "KUTTA4_step", "rk4": "RK4_step", "rk21": "RK21_step",
              "bs32": "BS32_step", "dopri54": "DOPRI54_step", "tsit54": "TSIT54_step", "vern65": "VERN65_step"}


def reference_available(root=REFERENCE_ROOT):
    return os.path.exists(os.path.join(root, "src", "numericalnim", "ode.nim"))


VECTOR_PROCS = ["newVector", "checkVectorSizes", "[]", "@", "size", "+", "-", "*", "/", "+.", "-.", "*.", "/.", "abs", "norm", "sum", "clone"]


def load_reference_ode(root=REFERENCE_ROOT, interpret_vector=True):
    """An interpreter holding ode.nim's solver procs and the utils.nim procs they call.  interpret_vector: `Vector[float]` states are the
    reference's own object and every operator on them is utils.nim's proc, interpreted (build them with `vector(it, [...])`); False: the Vec
    class above stands in (several times faster; used where the state type is not what is being checked)."""
    it = Interp()
    src = os.path.join(root, "src", "numericalnim")
    it.load(os.path.join(src, "ode.nim"), names=list(STEP_PROCS.values()) + ["+.", "/.", "*.", "size", "sum", "commonAdaptiveMethodCode", "newODEoptions",
                                                                          "DEFAULT_ODEoptions", "ODESolver", "solveODE"])
    it.load(os.path.join(src, "utils.nim"), names=["hermiteSpline", "linspace"] + (VECTOR_PROCS if interpret_vector else []))
    it.interpret_vector = interpret_vector
    return it


def vector(it, comps):
    """A Vector[float] state for the interpreter `it`: newVector(@[...]) of the reference's text (utils.nim:19-20), or the stand-in class."""
    comps = [float(x) for x in comps]
    return it.call("newVector", comps) if getattr(it, "interpret_vector", False) else Vec(comps)


def components(v):
    return list(v.get("components")) if is_vector(v) else list(v.c)

"""Runs nim/numericalnim_hip.nim — the reference-side binding, which has never met a Nim compiler here — THROUGH AN INTERPRETER against the
built libnnhip_ode.so: the Nim-subset interpreter that executes the reference's own ode.nim for the oracle pin (oracle/nim_subset.py), extended
by what a binding module needs on top of a numerical one — `type` sections (objects, enums), object construction with default fields,
seq / string / table values with Nim's value semantics, `addr`, `if` expressions, command-call syntax, slices, C conversions (`.cint`,
`.cdouble`, `.cstring` ...), and `{.importc.}` procs bound with ctypes from nim/nnhip_ode_bindings.nim.  `addr x[0]` becomes a C array built
from the seq and copied back after the call, `addr v` a by-reference scalar or struct.

What this shows and what it does not: the shim's LOGIC — which entry it calls with which arguments in which order, how it sizes and slices its
buffers, how it assembles what it returns — is executed against the real library, so its results can be compared bit for bit with the Python
mirror's (tests/test_gpu_nim_shim_interpreted.py).  It is not a type checker: a program this interpreter runs may still be rejected by `nim c`
(tests/test_nim_shim_static.py checks the raw calls' argument kinds against the generated signatures; tests/test_gpu_nim_shim.py compiles where
a toolchain exists).  The four names the shim imports from the numericalnim package (ODEoptions / newODEoptions, NumContext / newNumContext)
are stand-ins here, field for field (ode.nim:26-34, 78-104; commonTypes.nim:4-39): /root/reference does not travel to the GPU box."""
import ctypes
import os
import re

from oracle.nim_subset import (Alias, Env, Interp, NimError, NimObj, Parser, Routine, _BIN_PREC, _BUILTINS, _Return, norm_ident, tokenize)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NIM_DIR = os.path.join(ROOT, "nim")
LIB_PATH = os.environ.get("NNHIP_LIB", os.path.join(ROOT, "numericalnim_amd", "csrc", "libnnhip_ode.so"))


class TWord(str):
    """the leading word of a type, as the base parser keeps it, with the whole type expression attached"""
    full = ""


# ---- type expressions: "seq[seq[float]]", "(seq[float], seq[Odebatch])", "ptr cdouble", "Numcontext[Odebatch, float]" -----------------
def parse_type(text):
    text = (text or "").strip()
    if not text:
        return ("name", "")
    if text.startswith("("):
        inner, depth, parts, cur = text[1:text.rindex(")")], 0, [], []
        for ch in inner:
            if ch in "([": depth += 1
            elif ch in ")]": depth -= 1
            if ch == "," and depth == 0:
                parts.append("".join(cur)); cur = []
            else:
                cur.append(ch)
        parts.append("".join(cur))
        return ("tuple", [parse_type(p.split(":")[-1] if ":" in p and "[" not in p.split(":")[0] else p) for p in parts])
    for kw in ("ptr ", "ref ", "var "):
        if text.startswith(kw):
            return (kw.strip(), parse_type(text[len(kw):]))
    m = re.match(r"(\w+)\[(.*)\]$", text)
    if m:
        head = m.group(1)
        if head in ("seq", "openarray", "openArray", "array"):
            return ("seq", parse_type(m.group(2).split(",")[-1] if head == "array" else m.group(2)))
        return ("name", head)
    return ("name", text)


_INTS = {"int", "cint", "int64", "int32", "int16", "int8", "uint32", "uint64", "cuint", "csizet", "natural"}
_FLOATS = {"float", "float64", "cdouble", "cfloat", "float32"}


class CStringArray(list):
    """allocCStringArray(seq[string])"""


class Ptr:
    """`addr x[i]` (container + index) or `addr v` (a variable: environment + name)"""
    def __init__(self, container=None, index=0, env=None, name=None):
        self.container, self.index, self.env, self.name = container, index, env, name


class ShimParser(Parser):
    BIN = dict(_BIN_PREC); BIN.update({"div": 9, "mod": 9})

    def skip_type(self, stop_ops):
        start = self.p
        first = super().skip_type(stop_ops)
        w = TWord(first if first is not None else "")
        w.full = "".join((" " + str(t[1]) + " ") if (t[0] in ("kw",) or (t[0] == "id" and t[1] in ("ptr", "ref", "var"))) else str(t[1]) for t in self.t[start:self.p]).strip()
        w.full = re.sub(r"\s+", " ", w.full)
        return w if first is not None else None

    def routine(self, kind):
        """as the base parser's, plus forward declarations (a header without `=`) — skipped, the definition follows"""
        save = self.p
        name = self.next()[1]
        self.accept("op", "*")
        if self.at("op", "["):
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
        if not self.accept("op", "="):
            return ("pass",)
        return (kind, name, params, rtype, self.block_or_stmt())

    # -- statements ---------------------------------------------------------------------------------------------------------------
    LINES = {}   # id(statement node) -> source line (the nodes live as long as the loaded module): statement coverage of an interpreted run

    def stmt(self):
        line = self.peek()[2]
        node = self._stmt()
        if node[0] not in ("proc", "template", "typedef", "pass"): ShimParser.LINES[id(node)] = (getattr(self, "unit", "?"), line, node)
        return node

    def _stmt(self):
        tok = self.peek()
        if tok[0] == "id" and tok[1] == "macro" and self.peek(1)[0] == "id":
            self.next(); return self.routine("proc")
        if tok[0] == "id" and tok[1] == "block" and self.peek(1)[0] == "op" and self.peek(1)[1] == ":":
            self.next(); return ("blockstmt", self.colon_block())
        if tok[0] == "kw" and tok[1] == "for" and self.peek(2)[0] == "op" and self.peek(2)[1] == ",":
            self.next()
            v1 = self.expect("id")[1]; self.expect("op", ","); v2 = self.expect("id")[1]
            self.expect("kw", "in")
            return ("for2", v1, v2, self.expr(), self.colon_block())
        if tok[0] == "id" and tok[1] == "import":
            while not self.at("nl"): self.next()
            return ("pass",)
        if tok[0] == "id" and tok[1] == "type":
            self.next()
            return self.type_section()
        if tok[0] == "kw" and tok[1] in ("let", "var", "const", "if", "when", "while", "for", "break", "discard", "return", "raise", "case", "proc", "template", "func"):
            return super().stmt()
        e = self.expr()
        tok = self.peek()
        if tok[0] == "op" and tok[1] in ("=", "+=", "-=", "*=", "/="):
            self.next()
            return ("assign", tok[1], e, self.expr())
        if e[0] in ("id", "dot") and self.starts_expr(tok):  # command-call syntax: `check f(x)`, `result.add y`
            args = [(None, None, self.expr())]
            while self.accept("op", ","):
                args.append((None, None, self.expr()))
            return ("expr", ("call", e, args))
        return ("expr", e)

    @staticmethod
    def starts_expr(tok):
        return tok[0] in ("id", "num", "str") or (tok[0] == "kw" and tok[1] in ("if", "not", "nil", "true", "false")) or (tok[0] == "op" and tok[1] in ("@", "$", "&"))

    def type_section(self):
        defs = []
        indented = False
        if self.accept("nl"):
            self.expect("indent"); indented = True
        while True:
            while self.accept("nl"): pass
            if indented and self.at("dedent"):
                self.next(); break
            if self.at("eof"): break
            name = self.expect("id")[1]
            self.accept("op", "*")
            if self.at("op", "["):  # generic parameters
                depth = 0
                while True:
                    t = self.next()
                    if t[1] == "[": depth += 1
                    elif t[1] == "]":
                        depth -= 1
                        if depth == 0: break
            self.expect("op", "=")
            kind = self.next()[1]
            if kind == "ref": kind = self.next()[1]
            if kind == "enum":
                members, value = [], 0
                def member():
                    nonlocal value
                    n = self.expect("id")[1]
                    if self.accept("op", "="):
                        value = self.expr()[1]
                    members.append((n, value)); value += 1
                if self.at("nl"):
                    self.next(); self.expect("indent")
                    while not self.at("dedent"):
                        if self.accept("nl") or self.accept("op", ","): continue
                        member()
                    self.expect("dedent")
                else:
                    member()
                    while self.accept("op", ","): member()
                defs.append((name, "enum", members))
            elif kind == "object":
                fields = []
                self.expect("nl"); self.expect("indent")
                while not self.at("dedent"):
                    if self.accept("nl"): continue
                    names = [self.expect("id")[1]]; self.accept("op", "*")
                    while self.accept("op", ","):
                        names.append(self.expect("id")[1]); self.accept("op", "*")
                    self.expect("op", ":")
                    t = self.skip_type(set())
                    fields += [(n, t.full) for n in names]
                self.expect("dedent")
                defs.append((name, "object", fields))
            else:
                raise NimError(f"type {name}: `{kind}` types are not supported")
            if not indented: break
        return ("typedef", defs)

    # -- expressions --------------------------------------------------------------------------------------------------------------
    def expr(self, min_prec=0):
        lhs = self.unary()
        while True:
            tok = self.peek()
            op = tok[1] if tok[0] in ("op", "kw", "id") else None
            prec = self.BIN.get(op) if (tok[0] != "id" or op in ("div", "mod")) else None
            if prec is None or prec < min_prec: return lhs
            self.next()
            rhs = self.expr(prec if op == "^" else prec + 1)
            lhs = ("bin", op, lhs, rhs)

    def unary(self):
        tok = self.peek()
        if tok[0] == "op" and tok[1] == "$":
            self.next(); return ("un", "$", self.unary())
        if tok[0] == "op" and tok[1] == "&" and self.peek(1)[0] == "str":
            self.next(); return ("fmt", self.next()[1])
        if tok[0] == "id" and tok[1] in ("addr", "unsafeaddr") and not (self.peek(1)[0] == "op" and self.peek(1)[1] in (".", "=", ",", ")", ":")):
            self.next()
            if self.at("op", "("):
                self.next(); e = self.expr(); self.expect("op", ")")
            else:
                e = self.primary()
            return ("addr", e)
        return super().unary()

    def primary(self):
        tok = self.peek()
        if tok[0] == "op" and tok[1] == "{":   # set literal
            self.next()
            return ("set", [a[2] for a in self.call_args("}")])
        if tok[0] == "kw" and tok[1] == "if":  # if-expression
            self.next()
            branches = []
            cond = self.expr(); self.expect("op", ":"); branches.append((cond, self.expr()))
            while self.accept("kw", "elif"):
                cond = self.expr(); self.expect("op", ":"); branches.append((cond, self.expr()))
            self.expect("kw", "else"); self.expect("op", ":")
            return ("ifexpr", branches, self.expr())
        return super().primary()


_ESC = {"n": "\n", "t": "\t", "r": "\r", "0": "\0"}


def _tokenize(text):
    """the base tokenizer keeps the character after a backslash as it is (the reference's numerical code has no escapes); a binding module
    writes \\n in the text it emits: escapes are carried through it as private-use characters and put back in the string tokens"""
    def protect(m):
        body = re.sub(r"\\([ntr0])", lambda e: chr(0xE000 + ord(_ESC[e.group(1)])), m.group(0))
        return body
    marked = re.sub(r'"(?:[^"\\\n]|\\.)*"', protect, text)
    toks = tokenize(marked)
    return [(k, "".join(chr(ord(c) - 0xE000) if 0xE000 <= ord(c) < 0xE100 else c for c in v), ln) if k == "str" else (k, v, ln) for k, v, ln in toks]


class NimNode:
    """macros.NimNode, as far as a translating macro reads it: kind, children, the literal values, `$`, repr"""
    KINDS = ["nnkNone", "nnkEmpty", "nnkIdent", "nnkSym", "nnkType", "nnkCharLit", "nnkIntLit", "nnkInt8Lit", "nnkInt16Lit", "nnkInt32Lit", "nnkInt64Lit", "nnkUIntLit",
             "nnkUInt8Lit", "nnkUInt16Lit", "nnkUInt32Lit", "nnkUInt64Lit", "nnkFloatLit", "nnkFloat32Lit", "nnkFloat64Lit", "nnkFloat128Lit", "nnkStrLit", "nnkRStrLit",
             "nnkTripleStrLit", "nnkNilLit", "nnkCommand", "nnkCall", "nnkInfix", "nnkPrefix", "nnkPar", "nnkBracketExpr", "nnkDotExpr", "nnkAsgn", "nnkStmtList",
             "nnkLetSection", "nnkIdentDefs", "nnkForStmt", "nnkDiscardStmt", "nnkCommentStmt", "nnkTupleConstr", "nnkIfStmt", "nnkWhileStmt", "nnkVarSection"]   # (the order of system's NimNodeKind up to nnkNilLit: the ranges a macro writes rely on it)
    FIELDS = {"kind", "intval", "floatval", "strval", "repr"}

    def __init__(self, kind, sons=(), value=None, text=""):
        self.kind, self.sons, self.value, self.text = NimNode.KINDS.index(kind), list(sons), value, text

    def field(self, name):
        if name == "kind": return self.kind
        if name == "repr": return self.text or str(self)
        return self.value

    def __len__(self): return len(self.sons)
    def __iter__(self): return iter(self.sons)

    def __str__(self):
        return str(self.value) if self.value is not None else (self.text or NimNode.KINDS[self.kind])


class BodyParser(ShimParser):
    """parses the body a user hands to `deviceRhs` into the tree the Nim compiler would give the macro: parentheses are nodes of their own (nnkPar)"""
    def primary(self):
        tok = self.peek()
        if tok[0] == "op" and tok[1] == "(":
            self.next()
            items = self.call_args(")")
            if len(items) != 1 or items[0][0] is not None: raise NimError("tuples are not part of a right-hand side")
            node = ("par", items[0][2])
            while True:
                tok = self.peek()
                if tok[0] == "op" and tok[1] == ".":
                    self.next(); node = ("dot", node, self.next()[1])
                elif tok[0] == "op" and tok[1] == "[":
                    self.next(); node = ("idx", node, self.call_args("]"))
                else:
                    return node
        return super().primary()


def nim_ast(body_text):
    """the body of a `deviceRhs(...): body` block as a NimNode tree (identifiers keep the spelling of the source: the macro compares them)"""
    spelled = {}
    for w in re.findall(r"[A-Za-z_]\w*", body_text):
        spelled.setdefault(norm_ident(w), w)
    ident = lambda n: NimNode("nnkIdent", value=spelled.get(n, n))

    def conv(a):
        k = a[0]
        if k == "num": return NimNode("nnkFloatLit" if isinstance(a[1], float) else "nnkIntLit", value=a[1])
        if k == "str": return NimNode("nnkStrLit", value=a[1])
        if k == "id": return ident(a[1])
        if k == "par": return NimNode("nnkPar", [conv(a[1])])
        if k == "bin": return NimNode("nnkInfix", [ident(a[1]), conv(a[2]), conv(a[3])])
        if k == "un": return NimNode("nnkPrefix", [ident(a[1]), conv(a[2])])
        if k == "idx": return NimNode("nnkBracketExpr", [conv(a[1])] + [conv(x[2]) for x in a[2]])
        if k == "dot": return NimNode("nnkDotExpr", [conv(a[1]), ident(a[2])])
        if k == "call": return NimNode("nnkCall", [conv(a[1])] + [conv(x[2]) for x in a[2]])
        raise NimError(f"expression {k} has no NimNode form here")

    def stmt(st):
        k = st[0]
        if k == "assign" and st[1] == "=": return NimNode("nnkAsgn", [conv(st[2]), conv(st[3])])
        if k == "decl" and st[2] == "let":
            return NimNode("nnkLetSection", [NimNode("nnkIdentDefs", [ident(names[0]), NimNode("nnkEmpty"), conv(init)]) for names, _t, init in st[1]])
        if k == "decl": return NimNode("nnkVarSection")
        if k == "for": return NimNode("nnkForStmt", [ident(st[1]), conv(st[2]), NimNode("nnkStmtList", [stmt(x) for x in st[3]])])
        if k == "pass": return NimNode("nnkDiscardStmt")
        if k == "if": return NimNode("nnkIfStmt")
        if k == "while": return NimNode("nnkWhileStmt")
        if k == "expr": return conv(st[1])
        raise NimError(f"statement {k} has no NimNode form here")

    ps = BodyParser(_tokenize(body_text))
    out = []
    while not ps.at("eof"):
        if ps.accept("nl") or ps.accept("dedent") or ps.accept("indent"): continue
        out.append(stmt(ps.stmt()))
    return NimNode("nnkStmtList", out)


class FFIProc:
    def __init__(self, interp, cname, params, ret):
        self.interp, self.cname, self.params, self.ret = interp, cname, params, ret  # params: [(name, type text)]
        self.fn = getattr(interp.lib, cname)
        self.fn.restype = {"cint": ctypes.c_int, "int64": ctypes.c_int64, "cstring": ctypes.c_char_p, "cdouble": ctypes.c_double, "void": None}[ret]
        self.calls = 0

    def __call__(self, *args):
        it = self.interp
        if len(args) != len(self.params):
            raise NimError(f"{self.cname}: {len(self.params)} arguments expected, {len(args)} given")
        cargs, backs = [], []
        for (pname, ptype), a in zip(self.params, args):
            cargs.append(it.to_c(ptype, a, backs, f"{self.cname}({pname})"))
        self.calls += 1
        it.ffi_log.append(self.cname)
        r = self.fn(*cargs)
        for back in backs: back()
        if self.ret == "cstring":
            return None if r is None else r.decode()
        return r


class ShimInterp(Interp):
    def __init__(self, lib_path=LIB_PATH):
        super().__init__()
        self.types = {}      # normalised name -> ("object", [(field, type text)]) | ("enum", [(member, value)])
        self.structs = {}    # normalised name -> ctypes.Structure subclass
        self.ffi_log = []
        self.lib = ctypes.CDLL(lib_path)
        g = self.globals.vars
        g["newodeoptions"] = self._new_ode_options
        g["newnumcontext"] = lambda: NimObj(["fvalues", "tvalues"], [{}, {}], tname="Numcontext")
        g[norm_ident("allocCStringArray")] = lambda s: CStringArray(s)
        g[norm_ident("deallocCStringArray")] = lambda a: None
        g[norm_ident("IOError")] = "IOError"
        for i, kname in enumerate(NimNode.KINDS): g[norm_ident(kname)] = i
        g["chr"] = chr
        g["contains"] = lambda s, sub: sub in s
        g["find"] = lambda s, sub, start=0: (s.find(sub, start) if isinstance(s, str) else (s.index(sub) if sub in s else -1))
        g["join"] = lambda parts, sep="": sep.join(parts)
        g["delete"] = lambda s, i: s.pop(i) and None
        g["error"] = self._macro_error
        g[norm_ident("BiggestFloat")] = float
        g[norm_ident("isMainModule")] = False
        g[norm_ident("newLit")] = lambda v: v
        g[norm_ident("newTree")] = lambda kind, *sons: list(sons)
        self.types["Odeoptions"] = ("object", [(n, "float") for n in ("dt", "dtmax", "dtmin", "tstart", "abstol", "reltol", "scalemax", "scalemin")])

    @staticmethod
    def _macro_error(msg, node=None):
        raise NimError(f"compile-time error: {msg}")

    # stand-in for the reference's constructor (ode.nim:78-104): defaults, abs() of every field but tStart, the two checks it makes
    @staticmethod
    def _new_ode_options(dt=1e-4, absTol=1e-4, relTol=1e-4, dtMax=1e-2, dtMin=1e-4, scaleMax=4.0, scaleMin=0.1, tStart=0.0, **kw):
        v = dict(dt=dt, abstol=absTol, reltol=relTol, dtmax=dtMax, dtmin=dtMin, scalemax=scaleMax, scalemin=scaleMin, tstart=tStart)
        v.update(kw)
        if abs(v["dtmax"]) < abs(v["dtmin"]): raise NimError("ValueError: dtMin must be less than dtMax")
        names = ["dt", "dtmax", "dtmin", "tstart", "abstol", "reltol", "scalemax", "scalemin"]
        return NimObj(names, [float(v[n]) if n == "tstart" else abs(float(v[n])) for n in names], tname="Odeoptions")

    # ---- loading -------------------------------------------------------------------------------------------------------------------
    def load_bindings(self, path=os.path.join(NIM_DIR, "nnhip_ode_bindings.nim")):
        text = open(path).read()
        head = text[:text.index("\nproc ")]
        self.exec_toplevel(head)
        for name, (kind, fields) in list(self.types.items()):
            if kind == "object" and name.startswith("Nnhip"):
                self.structs[name] = type(name, (ctypes.Structure,), {"_fields_": [(f, self.ctype_of(t)) for f, t in fields]})
        for m in re.finditer(r"^proc (\w+)\*\((.*?)\)(?:: ([\w ]+?))? \{\.importc", text, re.M):
            params = []
            for p in filter(None, (x.strip() for x in m.group(2).split(";"))):
                n, t = p.split(":", 1)
                params.append((norm_ident(n.strip()), " ".join(norm_ident(w) if w not in ("ptr",) else w for w in t.split())))
            self.globals.vars[norm_ident(m.group(1))] = FFIProc(self, m.group(1), params, (m.group(3) or "void").strip())

    def exec_toplevel(self, text, unit="?"):
        ps = ShimParser(_tokenize(text))
        ps.unit = unit
        while not ps.at("eof"):
            if ps.accept("nl") or ps.accept("dedent"): continue
            self.exec_stmt(ps.stmt(), self.globals)

    def load_shim(self, path=os.path.join(NIM_DIR, "numericalnim_hip.nim")):
        self.exec_toplevel(open(path).read(), unit=os.path.basename(path))

    @staticmethod
    def ctype_of(t):
        t = t.strip()
        return {"cdouble": ctypes.c_double, "float": ctypes.c_double, "int64": ctypes.c_int64, "int32": ctypes.c_int32, "cint": ctypes.c_int, "int": ctypes.c_int64}[t]

    # ---- values --------------------------------------------------------------------------------------------------------------------
    def default_of(self, ttext):
        t = parse_type(ttext if not isinstance(ttext, TWord) else ttext.full)
        return self._default(t)

    def _default(self, t):
        if t[0] == "tuple": return [self._default(x) for x in t[1]]
        if t[0] == "seq": return []
        if t[0] in ("ptr", "ref"): return None
        name = t[1]
        low = name.lower()
        if low in _INTS: return 0
        if low in _FLOATS: return 0.0
        if low == "bool": return False
        if low == "string": return ""
        if low in ("cstring", "pointer", "cstringarray"): return None
        d = self.types.get(name)
        if d and d[0] == "object":
            return NimObj([f for f, _ in d[1]], [self.default_of(ft) for _, ft in d[1]], tname=name)
        if d and d[0] == "enum": return d[1][0][1]
        return None

    def fits(self, t, v):
        """does the run-time value fit the declared parameter type (overload resolution of the three solveODE procs)"""
        if t[0] == "tuple": return isinstance(v, (list, tuple)) and len(v) == len(t[1])
        if t[0] == "seq":
            return isinstance(v, list) and (not v or self.fits(t[1], v[0]))
        if t[0] in ("ptr", "ref", "var"): return True
        low = t[1].lower()
        if low in _INTS: return isinstance(v, int) and not isinstance(v, bool)
        if low in _FLOATS: return isinstance(v, (int, float)) and not isinstance(v, bool)
        if low == "bool": return isinstance(v, bool)
        if low in ("string", "cstring"): return isinstance(v, str) or (low == "cstring" and v is None)
        if low in ("nimnode", "static", "untyped", "typed", "auto", "biggestfloat", "openarray", "seq"): return True
        d = self.types.get(t[1])
        if d and d[0] == "object": return isinstance(v, NimObj) and v.tname == t[1]
        if d and d[0] == "enum": return isinstance(v, int)
        if t[1] == "Numcontext": return v is None or (isinstance(v, NimObj) and v.tname == "Numcontext")
        return True

    # ---- FFI marshalling -----------------------------------------------------------------------------------------------------------
    def to_c(self, ptype, a, backs, where):
        if ptype in ("cint", "int32"): return ctypes.c_int(int(a))
        if ptype == "int64": return ctypes.c_int64(int(a))
        if ptype == "cdouble": return ctypes.c_double(float(a))
        if ptype == "cstring": return None if a is None else a.encode()
        if ptype == "cstringarray":
            if a is None: return None
            if not isinstance(a, CStringArray): raise NimError(f"{where}: a cstringArray is expected")
            return (ctypes.c_char_p * max(len(a), 1))(*[s.encode() for s in a])
        if ptype == "pointer": raise NimError(f"{where}: raw pointers are not modelled")
        if not ptype.startswith("ptr "): raise NimError(f"{where}: parameter type {ptype} is not modelled")
        if a is None: return None
        pointee = ptype[4:]
        if pointee == "cstring":
            if not isinstance(a, CStringArray): raise NimError(f"{where}: a cstringArray is expected")
            arr = (ctypes.c_char_p * max(len(a), 1))(*[s.encode() for s in a])
            return arr
        if not isinstance(a, Ptr): raise NimError(f"{where}: an address is expected, got {type(a).__name__}")
        if pointee in self.structs:
            S = self.structs[pointee]
            objs = a.container[a.index:] if a.container is not None else [a.env.vars[a.name]]
            arr = (S * len(objs))()
            for k, o in enumerate(objs):
                if not (isinstance(o, NimObj) and o.tname == pointee): raise NimError(f"{where}: {pointee} expected, got {getattr(o, 'tname', type(o).__name__)}")
                for f, _ in S._fields_: setattr(arr[k], f, o.get(f))
            def back(arr=arr, objs=objs):
                for k, o in enumerate(objs):
                    for f, _ in S._fields_: o.values[o.names.index(f)] = getattr(arr[k], f)
            backs.append(back)
            return arr
        ct = self.ctype_of(pointee)
        conv = float if ct is ctypes.c_double else int
        if a.container is not None:
            vals = a.container[a.index:]
            arr = (ct * max(len(vals), 1))(*[conv(v) for v in vals])
            def back(arr=arr, a=a, n=len(vals)):
                a.container[a.index:a.index + n] = [conv(arr[k]) for k in range(n)]
            backs.append(back)
            return arr
        cell = ct(conv(a.env.vars[a.name]))
        def back(cell=cell, a=a):
            a.env.vars[a.name] = conv(cell.value)
        backs.append(back)
        return ctypes.byref(cell)

    # ---- statements ----------------------------------------------------------------------------------------------------------------
    def exec_stmt(self, st, env):
        k = st[0]
        rec = ShimParser.LINES.get(id(st))
        if rec is not None and rec[2] is st: EXECUTED.add((rec[0], rec[1]))
        if k == "case":   # as the base interpreter's, plus `of a .. b` ranges
            subject = self.eval(st[1], env)
            for vals, body in st[2]:
                for v in vals:
                    x = self.eval(v, env)
                    if (subject in x) if isinstance(x, range) else (x == subject):
                        self.exec_block(body, env); return None
            if st[3] is not None: self.exec_block(st[3], env)
            return None
        if k == "blockstmt":
            from oracle.nim_subset import _Break
            try:
                return self.exec_block(st[1], env, want_value=True)   # a block is an expression: its last statement's value
            except _Break:
                pass
            return None
        if k == "for2":
            from oracle.nim_subset import _Break
            try:
                for i, v in enumerate(self.eval(st[3], env)):
                    scope = Env(env); scope.vars[st[1]] = i; scope.vars[st[2]] = v
                    self.exec_block(st[4], scope, new_scope=False)
            except _Break:
                pass
            return None
        if k == "typedef":
            for name, kind, payload in st[1]:
                self.types[name] = (kind, payload)
                if kind == "enum":
                    for member, value in payload: self.globals.vars[member] = value
            return None
        if k == "decl":
            for names, tword, init in st[1]:
                for n in names:
                    if init is not None:
                        v = self.eval(init, env)
                        if st[2] == "var" and isinstance(v, list) and not isinstance(v, CStringArray): v = list(v)   # seqs have value semantics
                        if st[2] == "var" and isinstance(v, NimObj) and v.tname != "Numcontext": v = NimObj(v.names, v.values, v.tname)
                        if tword is not None and not self.fits(parse_type(tword.full if isinstance(tword, TWord) else tword), v):
                            raise NimError(f"type mismatch: {st[2]} {n}: {getattr(tword, 'full', tword)} = {getattr(v, 'tname', type(v).__name__)}")
                    else:
                        v = self.default_of(tword) if tword is not None else None
                    env.vars[n] = v
            return None
        if k == "proc" and env is self.globals:
            self.globals.vars.setdefault(st[1], [])
            if not isinstance(self.globals.vars[st[1]], list): self.globals.vars[st[1]] = []
            self.globals.vars[st[1]].append(Routine(k, st[1], st[2], st[3], st[4], env)); return None
        return super().exec_stmt(st, env)

    def assign(self, lhs, value, env):
        if lhs[0] == "dot":
            obj = self.eval(lhs[1], env)
            if not (isinstance(obj, NimObj) and obj.has(lhs[2])): raise NimError(f"no field {lhs[2]} to assign to")
            obj.values[obj.names.index(lhs[2])] = value
            return
        return super().assign(lhs, value, env)

    # ---- calls ---------------------------------------------------------------------------------------------------------------------
    def pick(self, routines, args, kwargs):
        best, best_score = None, -1
        for r in routines:
            names = [p[0] for p in r.params]
            if len(args) > len(r.params) or any(k not in names for k in kwargs): continue
            if any(p[2] is None and p[0] not in kwargs for p in r.params[len(args):]): continue
            score = 0
            for a, (pname, tword, _d) in zip(args, r.params):
                if tword is None: score += 1; continue
                if not self.fits(parse_type(tword.full if isinstance(tword, TWord) else tword), a): score = -1; break
                score += 2
            if score > best_score: best, best_score = r, score
        return best

    def call_value(self, fn, args, kwargs, env, arg_nodes=None, block=None):
        if isinstance(fn, list) and fn and isinstance(fn[0], Routine):
            r = self.pick(fn, args, kwargs)
            if r is None:
                if fn[0].name in _BUILTINS: return _BUILTINS[fn[0].name](*args, **kwargs)
                raise NimError(f"no overload of {fn[0].name} accepts {[getattr(a, 'tname', type(a).__name__) for a in args]} {sorted(kwargs)}")
            return self.invoke(r, args, kwargs, caller_env=env)
        return super().call_value(fn, args, kwargs, env, arg_nodes=arg_nodes, block=block)

    def invoke(self, r, args, kwargs, caller_env=None, block=None, arg_nodes=None):
        if r.kind != "proc": return super().invoke(r, args, kwargs, caller_env, block, arg_nodes)
        env = Env(r.env)
        for i, (pname, tword, default) in enumerate(r.params):
            if i < len(args): v = args[i]
            elif pname in kwargs: v = kwargs[pname]
            elif default is not None: v = self.eval(default, r.env)
            else: raise NimError(f"{r.name}: missing argument {pname}")
            if tword is None and default is not None and default[0] in ("str", "num") and not isinstance(v, (Alias, NimNode)):   # `integrator = "dopri54"`: the type of the default
                want = type(default[1])
                if not (isinstance(v, want) or (want is float and isinstance(v, int) and not isinstance(v, bool))) or (isinstance(v, bool) != (want is bool)):
                    raise NimError(f"type mismatch: {r.name}({pname} = {default[1]!r}) got {type(v).__name__}")
            if tword is not None and tword not in ("untyped", "typed", "auto") and not isinstance(v, (Alias, NimNode)):
                t = parse_type(tword.full if isinstance(tword, TWord) else tword)
                if not self.fits(t, v):   # a run-time stand-in for the compiler's check: every argument is of the kind its parameter declares
                    raise NimError(f"type mismatch: {r.name}({pname}: {getattr(tword, 'full', tword)}) got {getattr(v, 'tname', type(v).__name__)}")
            env.vars[pname] = v
        if len(r.body) == 1 and r.body[0][0] == "expr" and not (r.body[0][1][0] == "call" and r.rtype is None):
            rec = ShimParser.LINES.get(id(r.body[0]))      # an expression-bodied proc: its one statement runs here
            if rec is not None and rec[2] is r.body[0]: EXECUTED.add((rec[0], rec[1]))
            return self.eval(r.body[0][1], env)
        env.vars["result"] = self.default_of(r.rtype) if r.rtype is not None else None
        last = None
        try:
            for st in r.body: last = self.exec_stmt(st, env)
        except _Return:
            return env.vars["result"]
        # a proc whose last statement is an expression returns it (`RhsSpec(...)` at the end of rhsFromSource)
        if r.rtype is not None and r.body and r.body[-1][0] == "expr" and last is not None: return last
        return env.vars["result"]

    def call(self, name, *args, **kw):
        """entry point for the tests: a proc of the shim (overloads resolved by the arguments), a raw binding, or a stand-in, by its Nim name"""
        key = norm_ident(name)
        if key not in self.globals.vars: raise NimError(f"{name} is not loaded")
        return self.call_value(self.globals.vars[key], list(args), {norm_ident(k): v for k, v in kw.items()}, self.globals)

    def expr(self, text, **local):
        """evaluates one Nim expression (object constructors, enum members ...) in the module's scope"""
        env = Env(self.globals)
        env.vars.update({norm_ident(k): v for k, v in local.items()})
        return self.eval(ShimParser(_tokenize(text)).expr(), env)

    # ---- expressions ---------------------------------------------------------------------------------------------------------------
    _CONV = {"cint": int, "int": int, "int64": int, "int32": int, "cdouble": float, "float": float, "float64": float, "cstring": lambda s: s, "string": str}

    def eval(self, node, env):
        k = node[0]
        if k == "ifexpr":
            for cond, val in node[1]:
                if self.eval(cond, env): return self.eval(val, env)
            return self.eval(node[2], env)
        if k == "fmt":
            return re.sub(r"\{(\w+)\}", lambda m: str(self.eval(("id", norm_ident(m.group(1))), env)), node[1])
        if k == "addr":
            target = node[1]
            if target[0] == "idx":
                return Ptr(container=self.eval(target[1], env), index=self.eval(target[2][0][2], env))
            if target[0] == "id":
                e = env.find(target[1])
                if e is None: raise NimError(f"addr of undeclared {target[1]}")
                return Ptr(env=e, name=target[1])
            raise NimError("addr of this expression is not modelled")
        if k == "un" and node[1] == "$":
            v = self.eval(node[2], env)
            if isinstance(v, float): return repr(v)           # Nim's `$` on a float: shortest text that reads back as the same double
            return "" if v is None else str(v)
        if k == "set": return [self.eval(a, env) for a in node[1]]
        if k == "par": return self.eval(node[1], env)
        if k == "bin" and node[1] == "&":
            return str(self.eval(node[2], env)) + str(self.eval(node[3], env))
        if k == "bin" and node[1] in ("div", "mod"):
            a, b = self.eval(node[2], env), self.eval(node[3], env)
            q = abs(a) // abs(b) * (1 if (a < 0) == (b < 0) else -1)      # Nim's div truncates
            return q if node[1] == "div" else a - q * b
        if k == "dot":
            recv = self.eval(node[1], env)
            if isinstance(recv, NimObj) and recv.has(node[2]): return recv.get(node[2])
            if node[2] in self._CONV: return self._CONV[node[2]](recv)
            if isinstance(recv, NimNode) and node[2] in NimNode.FIELDS: return recv.field(node[2])
            if node[2] == "len": return len(recv)
            if node[2] == "isnil" or node[2] == "isNil": return recv is None
            return self.call_value(self.resolve_callable(node[2], recv, env), [recv], {}, env)
        if k == "idx":
            head = node[1]
            if head[0] == "id" and (head[1] in ("newseq",) or head[1] in self.types or head[1] in ("newnumcontext", "Numcontext", "inittable")):
                return ("generic", head[1], node[2])
            base = self.eval(node[1], env)
            i = self.eval(node[2][0][2], env)
            if isinstance(i, range): return base[i.start:i.stop] if isinstance(base, str) else list(base[i.start:i.stop])
            if isinstance(base, NimNode): return base.sons[i]
            if isinstance(base, dict):
                if i not in base: raise NimError(f"KeyError: key not found: {i}")
                return base[i]
            return base[i]
        if k == "call":
            fnode, arglist = node[1], node[2]
            if fnode[0] == "idx" and fnode[1][0] == "id" and fnode[1][1] == "newseq":   # newSeq[T](n): zero-initialised
                t = fnode[2][0][2]
                n = self.eval(arglist[0][2], env) if arglist else 0
                return [self._default(("name", t[1] if t[0] == "id" else "float"))] * int(n)
            if fnode[0] == "idx" and fnode[1][0] == "id" and fnode[1][1] == "newnumcontext":
                return self.globals.vars["newnumcontext"]()
            tname = fnode[1] if fnode[0] == "id" else None
            if tname in self.types:
                kind, payload = self.types[tname]
                if kind == "enum": return int(self.eval(arglist[0][2], env))
                obj = self._default(("name", tname))
                for name, sep, a in arglist:
                    if sep != ":" or not obj.has(name): raise NimError(f"{tname} has no field {name}")
                    obj.values[obj.names.index(name)] = self.eval(a, env)
                return obj
            if tname in self._CONV and len(arglist) == 1: return self._CONV[tname](self.eval(arglist[0][2], env))
            if fnode[0] == "dot" and fnode[2] == "add":   # s.add(x) / result[1].add x
                recv = self.eval(fnode[1], env)
                if isinstance(recv, list):
                    recv.append(self.eval(arglist[0][2], env)); return None
                if isinstance(recv, str):                  # strings are values: the variable gets the longer string
                    self.assign(fnode[1], recv + str(self.eval(arglist[0][2], env)), env); return None
        return super().eval(node, env)


def load(lib_path=LIB_PATH, macros=False):
    it = ShimInterp(lib_path)
    it.load_bindings()
    it.load_shim()
    if macros:   # nim/rhs_macro.nim: its translating procs, macros and templates (not its `when isMainModule` self-test, which needs the compiler's own macro expansion)
        text = open(os.path.join(NIM_DIR, "rhs_macro.nim")).read()
        it.exec_toplevel(text[:text.index("when isMainModule")], unit="rhs_macro.nim")
    return it


EXECUTED = set()   # (unit, line) of every statement any interpreter instance of this process has executed


def coverage(unit):
    """(statement lines of `unit` inside proc bodies, those executed so far)"""
    lines = {ln for u, ln, _n in ShimParser.LINES.values() if u == unit}
    return lines, {ln for u, ln in EXECUTED if u == unit}

## rhs_macro.nim — write the right-hand side f(t, y, ctx) in NIM, run it on the MI355X.
##
## The reference's `f` is a Nim closure (ODEProc[T], src/numericalnim/ode.nim:36).  A closure cannot run on the device; the backend takes a
## compiled-in kind or HIP C++ source text (`rhsFromSource*` in numericalnim_hip.nim, nnhip_ode_rhs_compile* in include/nnhip_ode.h).  So that a
## numericalnim user does not have to write C++ in a string, `deviceRhs` translates a RESTRICTED Nim body into that source at compile time:
##
##   let lorenz = deviceRhs(3, ["sigma", "rho", "beta"]):
##     dy[0] = ctx.fValues["sigma"] * (y[1] - y[0])
##     dy[1] = y[0] * (ctx.fValues["rho"] - y[2]) - y[1]
##     dy[2] = y[0] * y[1] - ctx.fValues["beta"] * y[2]
##   let (t, ys) = solveODE(lorenz, batch, tspan, ctx = ctx, integrator = "dopri54")     # numericalnim_hip.solveODE
##
## What the body may contain (anything else is a compile-time error naming the construct):
##   assignments   dy[i] = expr            (i an integer literal or a `for` variable)
##   locals        let x = expr            (-> const double x)
##   loops         for i in a ..< b / a .. b   with integer-literal bounds (emitted as a C loop; the backend unrolls it)
##   expressions   + - * /  unary -  parentheses  float and integer literals (integers become doubles: Nim's `1/3` is a float division)
##                 t   y[i]   ctx.fValues["key"] (-> p[k], k = position of "key" in `keys`)   ctx.tValues["name"][j] (-> name[j], see below)
##                 sqrt abs sin cos exp ln pow min max   (-> sqrt fabs sin cos exp log pow fmin-like selects; see NOTE)
## Evaluation order: every Nim infix node becomes ONE parenthesised C operation, so `a * b * c` is emitted as ((a * b) * c) — the order Nim
## evaluates it in — and the backend compiles user source with -ffp-contract=off: + - * / and sqrt give the bits the Nim closure gives on the CPU.
## NOTE  sin/cos/exp/ln/pow are the device's, not glibc's: a body that calls them agrees with the CPU to ~1 ulp per call, not bit for bit.
##       min/max follow system.min/max (`if a <= b: a else: b` / `if b <= a: a else: b`), NaN behaviour included.
## ctx.tValues: names used as ctx.tValues["name"][j] must be declared to the backend (`vectors`, rhsFromSourceCtx); `deviceRhsCtx` below does
## that from the same body.
##
## NOT compiled in this repository's build image (no Nim toolchain there) — like numericalnim_hip.nim; tests/test_gpu_nim_shim.py probes for a
## compiler on every GPU run and compiles an example through this macro where it finds one.
import std/[macros, strutils, sequtils]
import ./numericalnim_hip

const
  brOpen = $chr(1)    ## brackets of a ctx-vector access in the emitted text until its layout (shared: NAME[j], per IVP: NAME(j)) is known
  brClose = $chr(2)
  brName = $chr(3)    ## precedes the NAME of such an access: a vector "b" must not be found at the end of a vector "ab"

proc fail(n: NimNode, what: string) {.compileTime.} =
  error("deviceRhs: " & what & " is not in the translatable subset: " & n.repr, n)

proc cDouble(f: BiggestFloat): string {.compileTime.} =
  # shortest text that reads back as the same double (Nim's `$` on a float is round-trip exact), always with a decimal point or exponent
  result = $f
  if not (result.contains(".") or result.contains("e") or result.contains("inf") or result.contains("nan")): result.add ".0"

proc tr(n: NimNode, keys: seq[string], ints: seq[string], vectors: var seq[string]): string {.compileTime.}

proc trIndex(n: NimNode, ints: seq[string]): string {.compileTime.} =
  ## an index expression: integer literals, loop variables, + - * of those
  case n.kind
  of nnkIntLit..nnkUInt64Lit: result = $n.intVal
  of nnkIdent, nnkSym:
    if $n notin ints: fail(n, "index variable (only `for` variables are integers)")
    result = $n
  of nnkInfix:
    let op = $n[0]
    if op notin ["+", "-", "*", "mod", "div"]: fail(n, "index operator")
    result = "(" & trIndex(n[1], ints) & " " & (if op == "mod": "%" elif op == "div": "/" else: op) & " " & trIndex(n[2], ints) & ")"
  of nnkPar: result = "(" & trIndex(n[0], ints) & ")"
  else: fail(n, "index expression")

proc tr(n: NimNode, keys: seq[string], ints: seq[string], vectors: var seq[string]): string {.compileTime.} =
  case n.kind
  of nnkFloatLit..nnkFloat64Lit: result = cDouble(n.floatVal)
  of nnkIntLit..nnkUInt64Lit: result = cDouble(BiggestFloat(n.intVal))        # Nim converts integer literals in float context
  of nnkPar: result = "(" & tr(n[0], keys, ints, vectors) & ")"
  of nnkIdent, nnkSym:
    let s = $n
    if s == "t": result = "t"
    elif s in ints: result = "(double)" & s
    else: result = s                                                           # a `let` local of the body
  of nnkPrefix:
    if $n[0] != "-": fail(n, "prefix operator")
    result = "(-" & tr(n[1], keys, ints, vectors) & ")"
  of nnkInfix:
    let op = $n[0]
    if op notin ["+", "-", "*", "/"]: fail(n, "operator `" & op & "`")
    result = "(" & tr(n[1], keys, ints, vectors) & " " & op & " " & tr(n[2], keys, ints, vectors) & ")"
  of nnkBracketExpr:
    # y[i] | ctx.fValues["key"] | ctx.tValues["name"][j]
    if n[0].kind in {nnkIdent, nnkSym} and $n[0] == "y": return "y[" & trIndex(n[1], ints) & "]"
    if n[0].kind == nnkDotExpr and $n[0][0] == "ctx" and $n[0][1] == "fValues":
      if n[1].kind notin {nnkStrLit, nnkRStrLit}: fail(n, "a non-literal fValues key")
      let k = keys.find(n[1].strVal)
      if k < 0: error("deviceRhs: ctx.fValues[\"" & n[1].strVal & "\"] is not in `keys`", n)
      return "p[" & $k & "]"
    if n[0].kind == nnkBracketExpr and n[0][0].kind == nnkDotExpr and $n[0][0][0] == "ctx" and $n[0][0][1] == "tValues":
      if n[0][1].kind notin {nnkStrLit, nnkRStrLit}: fail(n, "a non-literal tValues key")
      let name = n[0][1].strVal
      if name notin vectors: vectors.add name
      # brackets of a ctx vector are emitted as brOpen .. brClose: whether NAME is shared (NAME[j]) or per IVP (NAME(j)) is decided where the layout is known
      return brName & name & brOpen & trIndex(n[1], ints) & brClose
    fail(n, "indexing")
  of nnkCall, nnkCommand:
    let f = $n[0]
    template arg(i: int): string = tr(n[i], keys, ints, vectors)
    case f
    of "sqrt", "sin", "cos", "exp": result = f & "(" & arg(1) & ")"
    of "ln": result = "log(" & arg(1) & ")"
    of "abs": result = "fabs(" & arg(1) & ")"
    of "pow": result = "pow(" & arg(1) & ", " & arg(2) & ")"
    of "min": result = "nnhip::nmin(" & arg(1) & ", " & arg(2) & ")"          # system.min: x <= y ? x : y (ode_device.hpp)
    of "max": result = "nnhip::nmax(" & arg(1) & ", " & arg(2) & ")"
    of "float", "toFloat": result = "(double)(" & (if n[1].kind in {nnkIdent, nnkSym} and $n[1] in ints: $n[1] else: arg(1)) & ")"
    else: fail(n, "call of `" & f & "`")
  else: fail(n, "expression of kind " & $n.kind)

proc trStmt(n: NimNode, keys: seq[string], ints: var seq[string], vectors: var seq[string], indent: string): string {.compileTime.} =
  case n.kind
  of nnkStmtList:
    for c in n: result.add trStmt(c, keys, ints, vectors, indent)
  of nnkAsgn:
    if n[0].kind != nnkBracketExpr or $n[0][0] != "dy": fail(n, "assignment (only `dy[i] = ...`)")
    result = indent & "dy[" & trIndex(n[0][1], ints) & "] = " & tr(n[1], keys, ints, vectors) & ";\n"
  of nnkLetSection:
    for d in n:
      if d.len != 3 or d[0].kind notin {nnkIdent, nnkSym}: fail(d, "let section entry")
      result.add indent & "const double " & $d[0] & " = " & tr(d[2], keys, ints, vectors) & ";\n"
  of nnkForStmt:
    if n.len != 3 or n[1].kind != nnkInfix or $n[1][0] notin ["..", "..<"]: fail(n, "for loop (only `for i in a .. b` / `a ..< b`)")
    let v = $n[0]
    ints.add v
    let lo = trIndex(n[1][1], ints)
    let hi = trIndex(n[1][2], ints)
    result = indent & "for (int " & v & " = " & lo & "; " & v & (if $n[1][0] == "..": " <= " else: " < ") & hi & "; ++" & v & ") {\n"
    result.add trStmt(n[2], keys, ints, vectors, indent & "  ")
    result.add indent & "}\n"
    ints.delete(ints.find(v))
  of nnkDiscardStmt, nnkCommentStmt: discard
  else: fail(n, "statement of kind " & $n.kind)

proc translate(body: NimNode, keys: seq[string], vectors: var seq[string]): string {.compileTime.} =
  var ints: seq[string]
  trStmt(body, keys, ints, vectors, "")

macro deviceRhsSource*(keys: static[openArray[string]], body: untyped): untyped =
  ## The HIP source text of `body` (the body of `rhs(double t, const double* y, double* dy, const double* p)`), as a string literal.
  var vectors: seq[string]
  let src = translate(body, @keys, vectors)
  if vectors.len > 0: error("deviceRhsSource: the body reads ctx.tValues (" & vectors.join(", ") & "): use deviceRhsCtx", body)
  result = newLit(src)

template deviceRhs*(dim: int, keys: static[openArray[string]], body: untyped): RhsSpec =
  ## f(t, y, ctx) written in Nim, compiled for the device once (here, at first use): stands in for ODEProc[T] in numericalnim_hip.solveODE.
  rhsFromSource(dim, deviceRhsSource(keys, body), @keys)

macro deviceRhsCtxSource*(keys: static[openArray[string]], body: untyped): untyped =
  ## (source, names of the ctx.tValues vectors the body reads, in order of first use)
  var vectors: seq[string]
  let src = translate(body, @keys, vectors)
  result = newTree(nnkTupleConstr, newLit(src), newLit(vectors))

template deviceRhsCtx*(dim: int, keys: static[openArray[string]], lens: openArray[int], perIvp: openArray[bool], body: untyped): RhsSpec =
  ## The same for a body that reads ctx.tValues["name"][j]: `lens[k]` / `perIvp[k]` describe the k-th vector the body uses (order of first
  ## use); a per-IVP vector is read as NAME(j) by the backend, so its uses are rewritten.  Bind the values with `bindCtx` before solving.
  block:
    const st = deviceRhsCtxSource(keys, body)
    var vs: seq[CtxVector]
    var src = st[0]
    for k, nm in st[1]:
      vs.add CtxVector(name: nm, len: lens[k], perIvp: perIvp[k])
      # this vector's accesses: NAME(j) for a per-IVP vector (function-like on the device), NAME[j] for a shared one
      var res = ""
      var i = 0
      while i < src.len:
        let at = src.find(brName & nm & brOpen, i)
        if at < 0:
          res.add src[i ..< src.len]
          break
        let close = src.find(brClose, at)
        res.add src[i ..< at] & nm & (if perIvp[k]: "(" else: "[") & src[at + nm.len + 2 ..< close] & (if perIvp[k]: ")" else: "]")
        i = close + 1
      src = res
    rhsFromSourceCtx(dim, src, @keys, vs)

when isMainModule:
  # what the macro emits for the Lorenz system — compare with the compiled-in kind NNHIP_RHS_LORENZ (include/nnhip_ode.h) term by term
  const src = deviceRhsSource(["sigma", "rho", "beta"]):
    dy[0] = ctx.fValues["sigma"] * (y[1] - y[0])
    dy[1] = y[0] * (ctx.fValues["rho"] - y[2]) - y[1]
    dy[2] = y[0] * y[1] - ctx.fValues["beta"] * y[2]
  doAssert src == "dy[0] = (p[0] * ((y[1] - y[0])));\ndy[1] = ((y[0] * ((p[1] - y[2]))) - y[1]);\ndy[2] = ((y[0] * y[1]) - (p[2] * y[2]));\n"
  const ring = deviceRhsSource(["c"]):
    for i in 0 ..< 16:
      dy[i] = -(float(i + 1) / 16.0) * y[i] + ctx.fValues["c"] * y[(i + 1) mod 16]
  echo src, ring

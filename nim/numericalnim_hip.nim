## numericalnim_hip.nim — the reference-side binding a numericalnim maintainer would add to route batches of
## independent IVPs to the MI355X backend.  It keeps the `solveODE` / `ODEoptions` / `NumContext` signatures
## (src/numericalnim/ode.nim:589-591, :26-34, common/commonTypes.nim:4-39) so it drops in behind the existing
## generic-T dispatch: `T` becomes `OdeBatch` (a batch of float / Vector[float] states) and `f` an `RhsKind`
## whose parameters are read from `ctx.fValues`.
##
## NOT compiled in this repository's build image (no Nim toolchain there); it is the binding shown in
## INTEGRATION.md.  Build:  nim c -d:release --passL:"-L<repo>/numericalnim_amd/csrc -lnnhip_ode" yourprog.nim
## (Nim's {.compile.} only drives the configured C compiler, so the hipcc-built library is linked, not compiled.)
import std/[strformat, tables, algorithm]
import numericalnim            # ODEoptions, newODEoptions, NumContext, newNumContext stay the reference's own

import ./nnhip_ode_bindings   # raw {.importc.} procs + NnhipOptions / NnhipStats, generated from include/nnhip_ode.h

type
  RhsKind* = enum                           ## include/nnhip_ode.h: enum nnhip_rhs_kind
    rhsNegY = 0, rhsLinear = 1, rhsLorenz = 2, rhsRing = 3, rhsAffineT = 4, rhsVanDerPol = 5
  RhsSpec* = object                         ## stands in for ODEProc[T] (ode.nim:36)
    kind*: RhsKind
    keys*: seq[string]                      ## ctx.fValues keys, in rhs_params order
    userKind*: int                          ## > 0: handle of a run-time compiled RHS (rhsFromSource); overrides `kind`
  BatchLayout* = enum layoutSoA = 0, layoutAoS = 1
  OdeBatch* = object                        ## N states of `dim` float64 components
    n*: int
    dim*: int
    layout*: BatchLayout
    data*: seq[float]                       ## [dim][N] (SoA) or [N][dim] (AoS)

proc rhsFromSource*(dim: int, body: string, keys: seq[string] = @[], name = "user"): RhsSpec =
  ## An arbitrary right-hand side given as HIP C++ source (compiled on the fly by the backend): `body` is the body of
  ## `rhs(double t, const double* y, double* dy, const double* p)`, p = ctx.fValues[keys[i]].
  var kind: cint
  let rc = nnhip_ode_rhs_compile(name.cstring, dim.cint, keys.len.cint, body.cstring, addr kind)
  if rc != 0: raise newException(ValueError, $nnhip_last_error())
  RhsSpec(kind: RhsKind(0), keys: keys, userKind: kind.int)

proc rhsFromSourcePerComponent*(dim: int, body: string, keys: seq[string] = @[], name = "user", haloLo = -1, haloHi = -1): RhsSpec =
  ## A right-hand side given per component: `body` returns dy_c for the component index `c`.  haloLo / haloHi >= 0: it reads components
  ## c - haloLo .. c + haloHi of its system only (cyclically) — neighbours then come from the adjacent lanes (nnhip_ode_rhs_set_halo).
  var kind: cint
  var rc = nnhip_ode_rhs_compile_comp(name.cstring, dim.cint, keys.len.cint, body.cstring, addr kind)
  if rc == 0 and haloLo >= 0 and haloHi >= 0: rc = nnhip_ode_rhs_set_halo(kind, haloLo.cint, haloHi.cint)
  if rc != 0: raise newException(ValueError, $nnhip_last_error())
  RhsSpec(kind: RhsKind(0), keys: keys, userKind: kind.int)

type CtxVector* = object                  ## one ctx.tValues entry a right-hand side from source reads
  name*: string
  len*: int
  perIvp*: bool                            ## false: NAME[j], shared by the batch; true: NAME(j), the IVP's own vector ([len][N])

proc rhsFromSourceCtx*(dim: int, body: string, keys: seq[string], vectors: seq[CtxVector], nAux = 0, name = "user",
                       perComponent = false): RhsSpec =
  ## NumContext in full (commonTypes.nim:4-27, nnhip_ode_rhs_compile_ctx): the body also sees the named ctx.tValues entries, any number
  ## of fValues (p[k]) and `aux(j)`, per-IVP doubles it may mutate (ode.nim:599).  Bind the values with `bindCtx` before solving.
  var kind: cint
  var vnames: seq[string]
  var lens: seq[int64]
  var per: seq[cint]
  for v in vectors:
    vnames.add(v.name)
    lens.add(v.len.int64)
    per.add((if v.perIvp: 1 else: 0).cint)
  var names = allocCStringArray(vnames)
  let rc = nnhip_ode_rhs_compile_ctx(name.cstring, dim.cint, keys.len.cint, body.cstring, (if perComponent: 1 else: 0).cint, vectors.len.cint, names,
                                     (if lens.len > 0: addr lens[0] else: nil), (if per.len > 0: addr per[0] else: nil), nAux.cint, addr kind)
  deallocCStringArray(names)
  if rc != 0: raise newException(ValueError, $nnhip_last_error())
  RhsSpec(kind: RhsKind(0), keys: (if keys.len > 8: @[] else: keys), userKind: kind.int)   # more than 8 scalars lead the shared block

proc bindCtx*(f: RhsSpec, shared, perIvp, auxInit: seq[float], nAux: int, n: int, device = 0) =
  ## shared = [the scalars, when the right-hand side has more than 8][the shared vectors in declaration order]; perIvp [rows][n];
  ## auxInit [nAux][n].  Host arrays; the backend keeps device copies until the next bindCtx / release.
  var s = shared
  var p = perIvp
  var a = auxInit
  let rc = nnhip_ode_rhs_bind_ctx_f64(f.userKind.cint, (if s.len > 0: addr s[0] else: nil), s.len.int64, (if p.len > 0: addr p[0] else: nil),
                                      (if n > 0: p.len div n else: 0).int64, (if a.len > 0: addr a[0] else: nil), nAux.cint, n.int64, device.cint)
  if rc != 0: raise newException(ValueError, $nnhip_last_error())

proc readAux*(f: RhsSpec, nAux: int, n: int): seq[float] =
  result = newSeq[float](nAux * n)
  if result.len > 0 and nnhip_ode_rhs_read_aux_f64(f.userKind.cint, addr result[0]) != 0: raise newException(ValueError, $nnhip_last_error())

proc toC(o: ODEoptions): NnhipOptions =
  NnhipOptions(dt: o.dt, dtMax: o.dtMax, dtMin: o.dtMin, tStart: o.tStart, absTol: o.absTol, relTol: o.relTol,
               scaleMax: o.scaleMax, scaleMin: o.scaleMin)

proc check(rc: cint) =
  if rc == 0: return
  let msg = $nnhip_last_error()
  if rc == -1 or rc == -2: raise newException(ValueError, msg)      # as ode.nim:95-100, :651
  raise newException(IOError, &"nnhip error {rc}: {msg}")

proc solveODE*(f: RhsSpec, y0: OdeBatch, tspan: openArray[float],
               options: ODEoptions = newODEoptions(), ctx: NumContext[OdeBatch, float] = nil,
               integrator = "dopri54", nGpus = 1, sweep: seq[seq[float]] = @[],
               sortBy: seq[float] = @[], autoSort = false): (seq[float], seq[OdeBatch]) =
  ## Batched drop-in for ode.nim:589-651: same parameter names, order and defaults; returns (t, y) where
  ## y[j] is the whole batch at t[j].  sweep[k][i] = value of RHS parameter k for IVP i (every IVP its own ctx).
  ## sortBy (one key per IVP) / autoSort: integrate heterogeneous batches in a divergence-friendly order below the C ABI
  ## (nnhip_ode_solve_batch_sorted_f64); results stay in the caller's order and are bit-identical.
  var ctx = ctx
  if ctx.isNil: ctx = newNumContext[OdeBatch, float]()               # ode.nim:604-606
  let integ = nnhip_ode_integrator_id(integrator.cstring)           # toLower + dispatch, ode.nim:607-651
  if integ < 0: raise newException(ValueError, &"{integrator} is not a valid integrator")
  var params: seq[cdouble]
  for k in f.keys: params.add(ctx.fValues[k].cdouble)               # ctx is the parameter channel (ode.nim:599)
  var opt = options.toC
  var ts = @tspan
  var tOut = newSeq[cdouble](max(ts.len, 1))
  var yOut = newSeq[cdouble](ts.len * y0.n * y0.dim)
  var ny = newSeq[int32](y0.n)
  var stats: NnhipStats
  var y0d = y0.data
  let pp = if params.len > 0: addr params[0] else: nil
  let rhsKind = (if f.userKind > 0: f.userKind else: f.kind.int).cint
  if nGpus > 1 and (sortBy.len > 0 or autoSort):
    raise newException(ValueError, "sortBy / autoSort order ONE device's batch: not available together with nGpus > 1")
  if nGpus > 1:                                                      # contiguous shards of the batch (and of the sweep table) per device
    var flat: seq[cdouble]
    for row in sweep:
      for v in row: flat.add(v.cdouble)
    let sp = if flat.len > 0: addr flat[0] else: nil
    check nnhip_ode_solve_batch_multi_gpu_sweep_f64(addr opt, integ, rhsKind, pp, params.len.cint, sp, sweep.len.cint, addr y0d[0], y0.n.int64,
                                                    y0.dim.cint, y0.layout.cint, addr ts[0], ts.len.cint, addr tOut[0],
                                                    addr yOut[0], addr ny[0], nil, nil, 0, addr stats, nGpus.cint)
  elif sortBy.len > 0 or autoSort:
    if sortBy.len > 0 and sortBy.len != y0.n: raise newException(ValueError, "sortBy needs one key per IVP")
    var flat: seq[cdouble]
    for row in sweep:
      for v in row: flat.add(v.cdouble)
    let sp = if flat.len > 0: addr flat[0] else: nil
    var keys: seq[cdouble]
    for v in sortBy: keys.add(v.cdouble)
    let kp = if keys.len > 0: addr keys[0] else: nil
    check nnhip_ode_solve_batch_sorted_f64(addr opt, integ, rhsKind, pp, params.len.cint, sp, sweep.len.cint, addr y0d[0], y0.n.int64,
                                           y0.dim.cint, y0.layout.cint, addr ts[0], ts.len.cint, addr tOut[0], addr yOut[0],
                                           addr ny[0], nil, nil, 0, kp, 0, 0)
    var nt: cint
    check nnhip_ode_time_grid(addr opt, addr ts[0], ts.len.cint, nil, addr nt)
    stats.nTOut = nt
  else:
    var flat: seq[cdouble]
    for row in sweep:
      for v in row: flat.add(v.cdouble)
    let sp = if flat.len > 0: addr flat[0] else: nil
    check nnhip_ode_solve_batch_sweep_f64(addr opt, integ, rhsKind, pp, params.len.cint, sp, sweep.len.cint, addr y0d[0], y0.n.int64,
                                          y0.dim.cint, y0.layout.cint, addr ts[0], ts.len.cint, addr tOut[0], addr yOut[0],
                                          addr ny[0], nil, nil, 0, addr stats, 0)
  result[0] = tOut[0 ..< stats.nTOut.int]
  let row = y0.n * y0.dim
  for j in 0 ..< ts.len:
    result[1].add OdeBatch(n: y0.n, dim: y0.dim, layout: y0.layout, data: yOut[j*row ..< (j+1)*row])

proc solveODE*(f: RhsSpec, y0: OdeBatch, tEnd: openArray[float], options: openArray[ODEoptions],
               ctx: NumContext[OdeBatch, float] = nil, integrator = "dopri54",
               sweep: seq[seq[float]] = @[]): (seq[OdeBatch], seq[int32]) =
  ## N separate reference calls in one launch: IVP i is `solveODE(f, y0_i, [options[i].tStart, tEnd[i]], options[i])`
  ## (ode.nim:589-591: every call owns its tspan and its options).  `options` holds one object per IVP, or a single one for all.
  ## Returns the two rows the reference returns per call (y0 first when tEnd > tStart, last when tEnd < tStart) and ny:
  ## 2, 1 when the span is empty, -1 for a call the reference would refuse (rows NaN); the other calls are unaffected.
  if tEnd.len != y0.n: raise newException(ValueError, "tEnd needs one value per IVP")
  if options.len != 1 and options.len != y0.n: raise newException(ValueError, "options: one object, or one per IVP")
  var ctx = ctx
  if ctx.isNil: ctx = newNumContext[OdeBatch, float]()
  let integ = nnhip_ode_integrator_id(integrator.cstring)
  if integ < 0: raise newException(ValueError, &"{integrator} is not a valid integrator")
  var params: seq[cdouble]
  for k in f.keys: params.add(ctx.fValues[k].cdouble)
  var base = (if options.len > 0: options[0] else: newODEoptions()).toC   # an empty batch may come with no options at all
  var each: seq[NnhipOptions]
  if options.len > 1:
    for o in options: each.add(o.toC)
  var flat: seq[cdouble]
  for row in sweep:
    for v in row: flat.add(v.cdouble)
  var te = @tEnd
  var y0d = y0.data
  var yOut = newSeq[cdouble](2 * y0.n * y0.dim)
  var ny = newSeq[int32](y0.n)
  let pp = if params.len > 0: addr params[0] else: nil
  let sp = if flat.len > 0: addr flat[0] else: nil
  let ep = if each.len > 0: addr each[0] else: nil
  let rhsKind = (if f.userKind > 0: f.userKind else: f.kind.int).cint
  check nnhip_ode_solve_batch_calls_f64(addr base, ep, integ, rhsKind, pp, params.len.cint, sp, sweep.len.cint, addr y0d[0], y0.n.int64,
                                        y0.dim.cint, y0.layout.cint, addr te[0], addr yOut[0], addr ny[0], nil, nil, 0, 0)
  let row = y0.n * y0.dim
  for j in 0 ..< 2:
    result[0].add OdeBatch(n: y0.n, dim: y0.dim, layout: y0.layout, data: yOut[j*row ..< (j+1)*row])
  result[1] = ny

proc solveODE*(f: RhsSpec, y0: OdeBatch, tspans: seq[seq[float]], options: openArray[ODEoptions],
               ctx: NumContext[OdeBatch, float] = nil, integrator = "dopri54"): (seq[seq[float]], seq[OdeBatch], seq[int32]) =
  ## N separate reference calls with n_t-point tspans in one launch: IVP i is `solveODE(f, y0_i, tspans[i], options[i])` — any order,
  ## both sides of tStart, duplicates (ode.nim:589-591, 476-487, 609).  Returns (t, y, ny): t[i] = the times the reference returns for
  ## call i, y[j] = the batch at output slot j, ny[i] = rows call i returns (-1: a call the reference would refuse).
  if tspans.len != y0.n: raise newException(ValueError, "tspans needs one row per IVP")
  if options.len != 1 and options.len != y0.n: raise newException(ValueError, "options: one object, or one per IVP")
  let nT = (if tspans.len > 0: tspans[0].len else: 0)
  var ctx = ctx
  if ctx.isNil: ctx = newNumContext[OdeBatch, float]()
  let integ = nnhip_ode_integrator_id(integrator.cstring)
  if integ < 0: raise newException(ValueError, &"{integrator} is not a valid integrator")
  var params: seq[cdouble]
  for k in f.keys: params.add(ctx.fValues[k].cdouble)
  var base = (if options.len > 0: options[0] else: newODEoptions()).toC   # an empty batch may come with no options at all
  var each: seq[NnhipOptions]
  if options.len > 1:
    for o in options: each.add(o.toC)
  var flat: seq[cdouble]
  for row in tspans:
    if row.len != nT: raise newException(ValueError, "every tspan needs the same number of points")
    for v in row: flat.add(v.cdouble)
  var y0d = y0.data
  var tFlat = newSeq[cdouble](max(y0.n * nT, 1))
  var yOut = newSeq[cdouble](max(nT * y0.n * y0.dim, 1))
  var ny = newSeq[int32](y0.n)
  let pp = if params.len > 0: addr params[0] else: nil
  let ep = if each.len > 0: addr each[0] else: nil
  let fp = if flat.len > 0: addr flat[0] else: nil
  let rhsKind = (if f.userKind > 0: f.userKind else: f.kind.int).cint
  check nnhip_ode_solve_batch_tspans_f64(addr base, ep, integ, rhsKind, pp, params.len.cint, nil, 0, addr y0d[0], y0.n.int64, y0.dim.cint,
                                         y0.layout.cint, fp, nT.cint, addr tFlat[0], addr yOut[0], addr ny[0], nil, nil, 0, 0)
  for i in 0 ..< y0.n:
    var ti: seq[float]
    for j in 0 ..< nT:
      let v = tFlat[i*nT + j]
      if v == v: ti.add(v)                                            # NaN = beyond the returned times
    result[0].add ti
  let row = y0.n * y0.dim
  for j in 0 ..< nT:
    result[1].add OdeBatch(n: y0.n, dim: y0.dim, layout: y0.layout, data: yOut[j*row ..< (j+1)*row])
  result[2] = ny

# ---- the consumers on either side of the solver (SURVEY §8 f4), same names as the reference's procs -------------------------
proc paramsOf(f: RhsSpec, ctx: NumContext[OdeBatch, float]): seq[cdouble] =
  for k in f.keys: result.add(ctx.fValues[k].cdouble)

proc cumQuadFn(simpson: bool, f: RhsSpec, X: openArray[float], ctx: NumContext[OdeBatch, float], dx: float,
               sweep: seq[seq[float]], dim: int): seq[OdeBatch] =
  ## cumtrapz(f, X, ctx, dx) / cumsimpson(f, X, ctx, dx) (integrate.nim:138-175, 377-400) for N parameter sets at once:
  ## sweep[k][i] overrides parameter k for item i (empty sweep: one item with ctx's parameters).  result[j] = the batch at X[j].
  var ctx = ctx
  if ctx.isNil: ctx = newNumContext[OdeBatch, float]()
  var params = paramsOf(f, ctx)
  let n = (if sweep.len > 0: sweep[0].len else: 1)
  var flat: seq[cdouble]
  for row in sweep:
    for v in row: flat.add(v.cdouble)
  var xs = @X
  var outBuf = newSeq[cdouble](xs.len * dim * n)
  var rows: cint
  let rhsKind = (if f.userKind > 0: f.userKind else: f.kind.int).cint
  let pp = if params.len > 0: addr params[0] else: nil
  let sp = if flat.len > 0: addr flat[0] else: nil
  if simpson:
    check nnhip_cumsimpson_fn_batch_f64(rhsKind, pp, params.len.cint, sp, sweep.len.cint, n.int64, dim.cint, 0, addr xs[0], xs.len.cint,
                                        dx.cdouble, addr outBuf[0], addr rows, 0)
  else:
    check nnhip_cumtrapz_fn_batch_f64(rhsKind, pp, params.len.cint, sp, sweep.len.cint, n.int64, dim.cint, 0, addr xs[0], xs.len.cint,
                                      dx.cdouble, addr outBuf[0], addr rows, 0)
  let row = dim * n
  for j in 0 ..< rows.int:
    result.add OdeBatch(n: n, dim: dim, layout: layoutSoA, data: outBuf[j*row ..< (j+1)*row])

proc cumtrapz*(f: RhsSpec, X: openArray[float], ctx: NumContext[OdeBatch, float] = nil, dx = 1e-5,
               sweep: seq[seq[float]] = @[], dim = 1): seq[OdeBatch] = cumQuadFn(false, f, X, ctx, dx, sweep, dim)
proc cumsimpson*(f: RhsSpec, X: openArray[float], ctx: NumContext[OdeBatch, float] = nil, dx = 1e-5,
                 sweep: seq[seq[float]] = @[], dim = 1): seq[OdeBatch] = cumQuadFn(true, f, X, ctx, dx, sweep, dim)

proc flatten(Y: openArray[OdeBatch]): seq[cdouble] =
  for b in Y:
    for v in b.data: result.add(v.cdouble)

proc cumtrapz*(Y: openArray[OdeBatch], X: openArray[float]): seq[OdeBatch] =
  ## cumtrapz(Y, X) for discrete points (integrate.nim:120-135) over every series of a trajectory (Y[j] = the batch at X[j]).
  var xs = @X
  var yin = flatten(Y)
  var outBuf = newSeq[cdouble](yin.len)
  let m = Y[0].data.len
  check nnhip_cumtrapz_batch_f64(addr xs[0], xs.len.cint, addr yin[0], m.int64, addr outBuf[0], 0)
  var rows: cint      # X in any order: the backend sorts and trims (integrate.nim:131); one row per distinct abscissa
  check nnhip_dataset_rows_f64(addr xs[0], xs.len.cint, addr rows, nil)
  for j in 0 ..< rows.int:
    result.add OdeBatch(n: Y[0].n, dim: Y[0].dim, layout: Y[0].layout, data: outBuf[j*m ..< (j+1)*m])

proc cumsimpson*(Y: openArray[OdeBatch], X: openArray[float]): seq[OdeBatch] =
  ## cumsimpson(Y, X) for discrete points (integrate.nim:329-375).
  var xs = @X
  var yin = flatten(Y)
  var outBuf = newSeq[cdouble](yin.len)
  let m = Y[0].data.len
  check nnhip_cumsimpson_batch_f64(addr xs[0], xs.len.cint, addr yin[0], m.int64, addr outBuf[0], 0)
  var rows: cint      # the rows hermiteInterpolate yields at the caller's abscissae (integrate.nim:375)
  check nnhip_dataset_rows_f64(addr xs[0], xs.len.cint, nil, addr rows)
  for j in 0 ..< rows.int:
    result.add OdeBatch(n: Y[0].n, dim: Y[0].dim, layout: Y[0].layout, data: outBuf[j*m ..< (j+1)*m])

type BatchHermiteSpline* = object            ## newHermiteSpline(X, Y[, dY]) for a whole batch (interpolate.nim:216-257)
  X*: seq[float]
  Y*, dY*: seq[cdouble]                      ## [knots][series]; dY empty -> slopes estimated by the backend
  proto*: OdeBatch

proc newHermiteSpline*(X: openArray[float], Y: openArray[OdeBatch], dY: openArray[OdeBatch] = []): BatchHermiteSpline =
  if X.len != Y.len or (dY.len != 0 and dY.len != X.len):
    raise newException(ValueError, "X and Y and dY must have the same length.")
  BatchHermiteSpline(X: @X, Y: flatten(Y), dY: flatten(dY), proto: Y[0])

proc evalImpl(s: BatchHermiteSpline, x: openArray[float], deriv: bool, extrap: int, extrapValue: float): seq[OdeBatch] =
  var xs = s.X
  var xq = @x
  var yv = s.Y
  var dv = s.dY
  let m = s.proto.data.len
  var outBuf = newSeq[cdouble](xq.len * m)
  let dp = if dv.len > 0: addr dv[0] else: nil
  check nnhip_hermite_spline_eval_batch_f64(addr xs[0], xs.len.cint, addr yv[0], dp, m.int64, addr xq[0], xq.len.cint,
                                            (if deriv: 1 else: 0).cint, extrap.cint, extrapValue.cdouble, addr outBuf[0], 0)
  for j in 0 ..< xq.len:
    result.add OdeBatch(n: s.proto.n, dim: s.proto.dim, layout: s.proto.layout, data: outBuf[j*m ..< (j+1)*m])

proc eval*(s: BatchHermiteSpline, x: openArray[float], extrap = 3, extrapValue = 0.0): seq[OdeBatch] =
  ## extrap: 0 Constant, 1 Edge, 2 Linear, 3 Native, 4 Error (ExtrapolateKind, interpolate.nim:89-90)
  evalImpl(s, x, false, extrap, extrapValue)
proc derivEval*(s: BatchHermiteSpline, x: openArray[float], extrap = 3, extrapValue = 0.0): seq[OdeBatch] =
  evalImpl(s, x, true, extrap, extrapValue)

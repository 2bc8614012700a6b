"""A SECOND, independent restatement of the reference's ODE path — test infrastructure only, like everything under oracle/.

oracle/ode_oracle.cpp is the checker the GPU parity tests use; it cannot be executed against the reference itself (the reference is
Nim, and neither this image nor the GPU boxes have a Nim toolchain), so its pin is: the reference's own analytic known-answer tests
plus agreement with an independent restatement.  This file IS that independent restatement: plain Python floats (IEEE binary64,
`math.pow` / `math.sqrt` = the C library's, exactly what Nim's std/math calls), written from src/numericalnim/ode.nim and
utils.nim directly — not from the C++ oracle — with every expression in the reference's association order.  Pure-Python loops: small
cases only.  tests/test_oracle_two_restatements.py demands that the two restatements agree BIT FOR BIT.

Reference lines: steppers ode.nim:107-468, controller :57-76, driver :471-586, dispatch :589-651, hermiteSpline utils.nim:273-279,
Vector operators utils.nim:59-64,113-118,176-197,219-223,243-250.
"""
import math


# ---- generic-T arithmetic: T = float, or a list standing for Vector[float] (element-wise, utils.nim) -----------------------------
def _isv(a):
    return isinstance(a, list)


def add(a, b):
    if _isv(a) and _isv(b):
        return [x + y for x, y in zip(a, b)]
    if _isv(a):
        return [x + b for x in a]
    if _isv(b):
        return [a + y for y in b]
    return a + b


def sub(a, b):
    if _isv(a):
        return [x - y for x, y in zip(a, b)]
    return a - b


def mul(d, v):  # float * T
    if _isv(v):
        return [x * d for x in v]  # utils.nim:176-180: v1[i] * d
    return d * v


def neg(v):
    return [-x for x in v] if _isv(v) else -v


def lin(terms):
    """sum of (coefficient, k) pairs, left to right, as the reference writes `a1 * k1 + a2 * k2 + ...`."""
    acc = mul(terms[0][0], terms[0][1])
    for c, k in terms[1:]:
        acc = add(acc, mul(c, k))
    return acc


def error_norm(yNew, error_y, absTol, relTol):  # ode.nim:61-65
    if _isv(yNew):
        totalTol = [absTol + relTol * abs(v) for v in yNew]      # absTol +. relTol * abs(yNew)
        err1 = [e / tt for e, tt in zip(error_y, totalTol)]      # error_y /. totalTol
        sq = [e * e for e in err1]                               # err1 *. err1
        s = 0.0
        for v in sq:                                             # sum: left to right (utils.nim:233-235, 243-250)
            s = s + v
        return math.sqrt(1 / float(len(sq)) * s)
    totalTol = absTol + relTol * abs(yNew)
    err1 = error_y / totalTol
    return math.sqrt(1 / 1.0 * (err1 * err1))


def nmin(x, y):  # system.min: if x <= y: x else: y
    return x if x <= y else y


def nmax(x, y):  # system.max: if y <= x: x else: y
    return x if y <= x else y


# ---- fixed-step steppers (ode.nim:107-189): (yNew, yNew, dt, 0.0) ----------------------------------------------------------------
def _fixed(name):
    def heun2(f, t, y, F, dt, o):
        k1 = f(t, y)
        k2 = f(t + dt, add(y, mul(dt, k1)))
        return add(y, mul(0.5 * dt, add(k1, k2)))

    def ralston2(f, t, y, F, dt, o):
        k1 = f(t, y)
        k2 = f(t + 2 / 3 * dt, add(y, mul(2 / 3 * dt, k1)))
        return add(y, mul(dt, add(mul(0.25, k1), mul(0.75, k2))))

    def kutta3(f, t, y, F, dt, o):
        k1 = f(t, y)
        k2 = f(t + 0.5 * dt, add(y, mul(0.5 * dt, k1)))
        k3 = f(t + dt, add(sub(y, mul(dt, k1)), mul(2 * dt, k2)))  # y - dt * k1 + 2 * dt * k2
        return add(y, mul(dt, lin([(1 / 6, k1), (2 / 3, k2), (1 / 6, k3)])))

    def heun3(f, t, y, F, dt, o):
        k1 = f(t, y)
        k2 = f(t + 1 / 3 * dt, add(y, mul(1 / 3 * dt, k1)))
        k3 = f(t + 2 / 3 * dt, add(y, mul(2 / 3 * dt, k2)))
        return add(y, mul(dt, add(mul(0.25, k1), mul(0.75, k3))))

    def ralston3(f, t, y, F, dt, o):
        k1 = f(t, y)
        k2 = f(t + 1 / 2 * dt, add(y, mul(1 / 2 * dt, k1)))
        k3 = f(t + 3 / 4 * dt, add(y, mul(3 / 4 * dt, k2)))
        return add(y, mul(dt, lin([(2 / 9, k1), (1 / 3, k2), (4 / 9, k3)])))

    def ssprk3(f, t, y, F, dt, o):
        k1 = f(t, y)
        k2 = f(t + dt, add(y, mul(dt, k1)))
        k3 = f(t + 0.5 * dt, add(y, mul(0.25 * dt, add(k1, k2))))
        return add(y, mul(dt, lin([(1 / 6, k1), (1 / 6, k2), (2 / 3, k3)])))

    def ralston4(f, t, y, F, dt, o):
        k1 = f(t, y)
        k2 = f(t + 0.4 * dt, add(y, mul(0.4 * dt, k1)))
        k3 = f(t + 0.45573725 * dt, add(y, mul(dt, lin([(0.29697761, k1), (0.15875964, k2)]))))
        # 0.21810040 * k1 - 3.05096516 * k2 + 3.83286476 * k3
        k4 = f(t + dt, add(y, mul(dt, add(sub(mul(0.21810040, k1), mul(3.05096516, k2)), mul(3.83286476, k3)))))
        return add(y, mul(dt, add(add(sub(mul(0.17476028, k1), mul(0.55148066, k2)), mul(1.20553560, k3)), mul(0.17118478, k4))))

    def kutta4(f, t, y, F, dt, o):
        k1 = f(t, y)
        k2 = f(t + 1 / 3 * dt, add(y, mul(1 / 3 * dt, k1)))
        k3 = f(t + 2 / 3 * dt, add(y, mul(dt, add(mul(-1 / 3, k1), k2))))   # -1/3 * k1 + k2
        k4 = f(t + dt, add(y, mul(dt, add(sub(k1, k2), k3))))               # k1 - k2 + k3
        return add(y, mul(dt, lin([(1 / 8, k1), (3 / 8, k2), (3 / 8, k3), (1 / 8, k4)])))

    def rk4(f, t, y, F, dt, o):
        k1 = f(t, y)
        k2 = f(t + 0.5 * dt, add(y, mul(0.5 * dt, k1)))
        k3 = f(t + 0.5 * dt, add(y, mul(0.5 * dt, k2)))
        k4 = f(t + dt, add(y, mul(dt, k3)))
        return add(y, mul(dt / 6.0, add(add(k1, mul(2.0, add(k2, k3))), k4)))  # k1 + 2.0 * (k2 + k3) + k4

    body = locals()[name]

    def step(f, t, y, FSAL, dt, o):
        yNew = body(f, t, y, FSAL, dt, o)
        return yNew, yNew, dt, 0.0
    return step


# ---- adaptive steppers: the stage block inside commonAdaptiveMethodCode (ode.nim:57-76) --------------------------------------------
DOPRI54 = dict(
    c=[1.0 / 5.0, 3.0 / 10.0, 4.0 / 5.0, 8.0 / 9.0, 1.0, 1.0],
    a=[[1.0 / 5.0], [3.0 / 40.0, 9.0 / 40.0], [44.0 / 45.0, -56.0 / 15.0, 32.0 / 9.0],
       [19372.0 / 6561.0, -25360.0 / 2187.0, 64448.0 / 6561.0, -212.0 / 729.0],
       [9017.0 / 3168.0, -355.0 / 33.0, 46732.0 / 5247.0, 49.0 / 176.0, -5103.0 / 18656.0],
       [35.0 / 384.0, 0.0, 500.0 / 1113.0, 125.0 / 192.0, -2187.0 / 6784.0, 11.0 / 84.0]],
    bhat=[5179.0 / 57600.0, 0.0, 7571.0 / 16695.0, 393.0 / 640.0, -92097.0 / 339200.0, 187.0 / 2100.0, 1.0 / 40.0], order=5, direct=False)
DOPRI54["b"] = DOPRI54["a"][5]
TSIT54 = dict(
    c=[0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0],
    a=[[0.161], [-0.008480655492356989, 0.335480655492357], [2.8971530571054935, -6.359448489975075, 4.3622954328695815],
       [5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525],
       [5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383],
       [0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774]],
    bhat=[-0.001780011052226, -0.000816434459657, 0.007880878010262, -0.144711007173263, 0.582357165452555, -0.458082105929187, 1.0 / 66.0],
    order=5, direct=True)
TSIT54["b"] = TSIT54["a"][5]
VERN65 = dict(
    c=[0.06, 0.09593333333333333, 0.1439, 0.4973, 0.9725, 0.9995, 1.0, 1.0],
    a=[[0.06], [0.019239962962962962, 0.07669337037037037], [0.035975, 0.0, 0.107925],
       [1.3186834152331484, 0.0, -5.042058063628562, 4.220674648395414],
       [-41.87259166432751, 0.0, 159.43256216313748, -122.11921356501004, 5.531743066200053],
       [-54.430156935316504, 0.0, 207.06725136501848, -158.61081378459, 6.991816585950242, -0.01859723106220323],
       [-54.66374178728198, 0.0, 207.95280625538936, -159.2889574744995, 7.018743740796944, -0.018338785905045722, -0.0005119484997882099],
       [0.03438957868357036, 0.0, 0.0, 0.25826245556335037, 0.4209371189673537, 4.405396469669310, -176.48311902429865, 172.36413340141507]],
    b=[0.03438957868357036, 0.0, 0.0, 0.25826245556335034, 0.42093711896735372, 4.4053964696693102, -176.48311902429866, 172.36413340141507],
    bhat=[0.04909967648382, 0.0, 0.0, 0.22511122295165, 0.46946822530296, 0.80657922499889, 0.0, -0.60711948917780, 0.05686113944048],
    order=6, direct=False)


def _adaptive(body, order):
    def step(f, t, y, FSAL, dt, o):
        limitCounter = 0
        while limitCounter < 2:
            yNew, error_y, fsalNew = body(f, t, y, FSAL, dt)
            error = error_norm(yNew, error_y, o["absTol"], o["relTol"])
            if error <= 1:
                break
            if error != error:  # NaN: the reference would loop forever; both oracles give up here (DESIGN.md §3 deviation 1)
                break
            dt = dt * nmin(4, nmax(0.125, 0.9 * math.pow(1 / error, 1 / order)))  # :71 — order is an int here: 1/order is a float
            if abs(dt) < o["dtMin"]:
                dt = o["dtMin"]
                limitCounter += 1
            elif o["dtMax"] < abs(dt):
                dt = o["dtMax"]
        return yNew, fsalNew, dt, error
    return step


def _tableau_body(T):
    def body(f, t, y, FSAL, dt):
        k = [FSAL]  # k1 = FSAL
        for s, row in enumerate(T["a"]):
            k.append(f(t + dt * T["c"][s], add(y, mul(dt, lin(list(zip(row, k)))))))
        yNew = add(y, mul(dt, lin(list(zip(T["b"], k)))))
        if T["direct"]:
            error_y = mul(dt, lin(list(zip(T["bhat"], k))))          # :372
        else:
            error_y = sub(yNew, add(y, mul(dt, lin(list(zip(T["bhat"], k))))))  # yNew - yLow
        return yNew, error_y, k[-1]
    return body


def _rk21_body(f, t, y, FSAL, dt):
    k1 = f(t, y)
    k2 = f(t + dt, add(y, mul(dt, k1)))
    yNew = add(y, mul(dt * 0.5, add(k1, k2)))   # y + dt * 0.5 * (k1 + k2)
    yLow = add(y, mul(dt, k1))
    return yNew, sub(yNew, yLow), yNew          # result = (yNew, yNew, dt, error)


def _bs32_body(f, t, y, FSAL, dt):
    k1 = f(t, y)
    k2 = f(t + 0.5 * dt, add(y, mul(0.5 * dt, k1)))
    k3 = f(t + 0.75 * dt, add(y, mul(0.75 * dt, k2)))
    yNew = add(y, mul(dt, lin([(2 / 9, k1), (1 / 3, k2), (4 / 9, k3)])))
    k4 = f(t + dt, yNew)
    yLow = add(y, mul(dt, lin([(7 / 24, k1), (1 / 4, k2), (1 / 3, k3), (1 / 8, k4)])))
    return yNew, sub(yNew, yLow), k4


# name -> (stepper, useFSAL, order, adaptive)   (ode.nim:607-649)
METHODS = {n: (_fixed(n), False, o, False) for n, o in (("heun2", 2.0), ("ralston2", 2.0), ("kutta3", 3.0), ("heun3", 3.0), ("ralston3", 3.0),
                                                        ("ssprk3", 3.0), ("ralston4", 4.0), ("kutta4", 4.0), ("rk4", 4.0))}
METHODS.update({"rk21": (_adaptive(_rk21_body, 2), False, 2.0, True), "bs32": (_adaptive(_bs32_body, 3), True, 3.0, True),
                "dopri54": (_adaptive(_tableau_body(DOPRI54), 5), True, 5.0, True), "tsit54": (_adaptive(_tableau_body(TSIT54), 5), True, 5.0, True),
                "vern65": (_adaptive(_tableau_body(VERN65), 6), True, 6.0, True)})


def hermite(x, x1, x2, y1, y2, dy1, dy2):  # utils.nim:273-279
    t = (x - x1) / (x2 - x1)
    h00 = (1.0 + 2.0 * t) * ((1.0 - t) * (1.0 - t))
    h10 = t * ((1.0 - t) * (1.0 - t))
    h01 = (t * t) * (3.0 - 2.0 * t)
    h11 = (t * t * t) - (t * t)
    return add(add(add(mul(h00, y1), mul(h10 * (x2 - x1), dy1)), mul(h01, y2)), mul(h11 * (x2 - x1), dy2))


def new_options(dt=1e-4, absTol=1e-4, relTol=1e-4, dtMax=1e-2, dtMin=1e-4, scaleMax=4.0, scaleMin=0.1, tStart=0.0):  # ode.nim:78-102
    if abs(dtMax) < abs(dtMin):
        raise ValueError("dtMin must be less than dtMax")
    if abs(scaleMax) < 1:
        raise ValueError("scaleMax must be bigger than 1")
    if 1 < abs(scaleMin):
        raise ValueError("scaleMin must be smaller than 1")
    return dict(dt=abs(dt), absTol=abs(absTol), relTol=abs(relTol), dtMax=abs(dtMax), dtMin=abs(dtMin), scaleMax=abs(scaleMax), scaleMin=abs(scaleMin),
                tStart=tStart)


def solve_ode(f, y0, tspan, options=None, integrator="dopri54", max_iter=10 ** 7):
    """solveODE (ode.nim:589-651) -> ODESolver (:471-586).  f(t, y) -> dy; y a float or a list.  Returns (t, y, n_steps)."""
    o = options or new_options()
    name = integrator.lower()
    if name not in METHODS:
        raise ValueError(f"{integrator} is not a valid integrator")
    stepper, useFSAL, order, adaptive = METHODS[name]
    tspan = sorted(tspan)
    t0 = o["tStart"]
    tPositive = [x for x in tspan if x > t0]
    tNegative = [x for x in tspan if x < t0][::-1]
    yPositive, yNegative = [], []
    clone = (lambda v: list(v)) if _isv(y0) else (lambda v: v)
    y = clone(y0)
    yZero, tZero = ([clone(y)], [t0]) if t0 in tspan else ([], [])
    dtInit = math.sqrt(o["dtMax"] * o["dtMin"]) if adaptive else o["dt"]
    useDense = len(tspan) != 2
    steps = 0

    def run(fn, tStart, tEnd, req, sign, out):
        nonlocal steps
        t = tStart
        y = clone(y0)
        FSAL = fn(t, y)
        lastIter = (t, y, FSAL)
        dt = dtInit
        denseIndex = 0
        high = len(req) - 1
        it = 0
        while t < tEnd:
            if useDense:
                if high < denseIndex:
                    break
                while sign * req[denseIndex] <= t:
                    dyNow = FSAL if useFSAL else fn(t, y)
                    out.append(hermite(sign * req[denseIndex], lastIter[0], t, lastIter[1], y, lastIter[2], dyNow))
                    denseIndex += 1
                    if high < denseIndex:
                        break
            dt = nmin(dt, tEnd - t)
            if useDense:
                lastIter = (t, y, FSAL if useFSAL else fn(t, y))
            y, FSAL, dt, error = stepper(fn, t, y, FSAL, dt, o)
            t += dt
            steps += 1
            if adaptive:
                if error == 0.0:
                    dt *= 5
                else:
                    dt = dt * nmin(4, nmax(0.125, 0.9 * math.pow(1 / error, 1 / order)))
                if dt < o["dtMin"]:
                    dt = o["dtMin"]
                elif o["dtMax"] < dt:
                    dt = o["dtMax"]
            if error != error:
                break
            it += 1
            if it > max_iter:
                raise RuntimeError("too many steps for a pure-Python check")
        out.append(y)

    if tPositive:
        run(f, t0, max(tPositive), tPositive, 1.0, yPositive)
    if tNegative:
        run(lambda t, y: neg(f(-t, y)), -t0, -min(tNegative), tNegative, -1.0, yNegative)
    return tNegative[::-1] + tZero + tPositive, yNegative[::-1] + yZero + yPositive, steps

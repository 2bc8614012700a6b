"""Run by scripts/tsan_host_audit.sh in a subprocess: the library's HOST code (built with -fsanitize=thread there) driven from 8 Python threads at once on the fake
HIP runtime (tests/cpp/fake_hip.cpp, LD_PRELOAD; kernels do nothing), with tests/fake_torch supplying "device" tensors.  TEST INFRASTRUCTURE, a one-off audit
tool: what it looks for is data races and lock-order inversions in the process-wide state of libnnhip_ode.so — the graph cache, the polling blocks, the staging
buffers, the hiprtc program cache, the per-thread context bindings, last-error strings — none of which a single-threaded test or a kernel emulation can show.
Every thread mixes: the fused solve (host arrays and device tensors, 3 integrators), one IntegratorProc call, the two streaming loops (eager and graph-replayed),
the dense streaming driver, per-IVP calls with binning, the discrete consumers (sorted and unsorted abscissae), a run-time compiled right-hand side shared by all
threads with a context block bound per thread, and the error path (last_error is per thread).  Prints "STRESS OK" at the end."""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "fake_torch"))

import torch  # noqa: E402  (tests/fake_torch)
import numericalnim_amd as nn  # noqa: E402
from numericalnim_amd import interpolate as ni  # noqa: E402
from test_ctx_block import MATVEC_SRC  # noqa: E402

# PyTorch's current stream is per thread; the stand-in's is one per process.  Here (not in tests/fake_torch, which is frozen) every worker gets its own:
_tls = threading.local()
_process_wide_current_stream = torch.cuda.current_stream
torch.cuda.current_stream = lambda device=None: getattr(_tls, "stream", None) or _process_wide_current_stream(device)

NTHREADS = int(os.environ.get("STRESS_THREADS", "8"))
ROUNDS = int(os.environ.get("STRESS_ROUNDS", "6"))
dev = torch.device("cuda", 0)
errors = []
shared_rhs = nn.Rhs.custom(4, MATVEC_SRC, keys=("s",), tvalues={"g": 4, "A": 16}, per_ivp=("A",), name="matvec4_stress")


def worker(k):
    try:
        rng = np.random.default_rng(k)
        n = 700 + 13 * k
        opt = nn.newODEoptions(absTol=1e-6, relTol=1e-6, dtMin=1e-8, dtMax=0.25)
        s = _tls.stream = torch.cuda.Stream()
        for r in range(ROUNDS):
            if True:
                y1 = torch.from_numpy(rng.random(n)).to(dev)
                y3 = torch.from_numpy(rng.random((3, n))).to(dev)
                y16 = torch.from_numpy(rng.random((n, 16))).to(dev)
                # fused solves: device tensors and host arrays
                nn.solveODE(nn.Rhs.neg_y(), y1, [0.0, 0.5], nn.newODEoptions(dt=2.0 ** -6), integrator="rk4")
                nn.solveODE(nn.Rhs.lorenz(), y3, [0.0, 0.2, 0.5], opt, integrator=("dopri54", "tsit54", "vern65")[r % 3], return_counts=True)
                nn.solveODE(nn.Rhs.ring(0.1), y16, [0.0, 0.5], opt, integrator="tsit54", layout=1)
                nn.solveODE(nn.Rhs.lorenz(), rng.random((3, 50)), [0.0, 0.3], opt, integrator="bs32")
                # one IntegratorProc call, the streaming loops (eager / graph replay), the dense driver
                nn.fixedStream(nn.Rhs.neg_y(), y1.clone(), 0.0, 0.25, nn.newODEoptions(dt=2.0 ** -7), integrator="rk4", scratch=torch.empty_like(y1))
                nn.adaptiveStream(nn.Rhs.lorenz(), y3.clone(), 0.0, 0.5, opt, integrator="dopri54")
                nn.adaptiveStream(nn.Rhs.ring(0.1), y16.clone(), 0.0, 0.5, opt, integrator="tsit54", layout=1)
                nn.adaptiveStreamSolve(nn.Rhs.lorenz(), y3, [0.0, 0.2, 0.5], opt, integrator="tsit54")
                nn.fixedStreamSolve(nn.Rhs.neg_y(), y1, [0.0, 0.1, 0.25], nn.newODEoptions(dt=2.0 ** -7), integrator="rk4")
                # separate calls with their own tEnd (binned below the boundary)
                te = torch.from_numpy(rng.uniform(0.05, 0.5, n)).to(dev)
                nn.solveODEPerIvpEnd(nn.Rhs.vanderpol(2.0), torch.from_numpy(np.stack([np.full(n, 2.0), np.zeros(n)])).to(dev), te, opt, integrator="dopri54")
                # consumers: sorted and unsorted abscissae (the sort-and-trim front end allocates, gathers, synchronises, frees)
                X = np.sort(rng.random(40))
                Y = torch.from_numpy(rng.random((40, n))).to(dev)
                ni.cumtrapz(Y, X)
                ni.cumsimpson(Y, X[::-1].copy())
                sp = ni.HermiteSpline(rng.permutation(X), Y)
                sp.eval(np.linspace(X[0], X[-1], 17))
                # one compiled source, a context block per thread
                ctx = nn.newNumContext(fValues={"s": 0.5 + k}, tValues={"g": rng.standard_normal(4), "A": rng.standard_normal((16, n))})
                y4 = torch.from_numpy(rng.random((4, n))).to(dev)
                nn.solveODE(shared_rhs, y4, [0.0, 0.3], opt, ctx=ctx, integrator="tsit54")
                # the error path: this thread's message, not another's
                try:
                    nn.solveODE(nn.Rhs.lorenz(), y3, [0.0, 1.0], opt, integrator="no_such_method_%d" % k)
                except Exception as exc:  # noqa: BLE001
                    assert "no_such_method_%d" % k in str(exc) or "integrator" in str(exc).lower(), str(exc)
                else:
                    raise AssertionError("an unknown integrator was accepted")
            s.synchronize()
            if os.environ.get("STRESS_RELEASE"):   # a thread may drop its caches (and the process-wide idle contexts) whenever it likes, others in mid-call or not
                assert nn._lib.lib().nnhip_release() == 0
    except BaseException as exc:  # noqa: BLE001
        import traceback
        errors.append((k, traceback.format_exc()))
        raise exc


def main():
    # knobs are process-wide (documented): set before the threads start — STRESS_KNOBS="stream_graph=1,adv_lean=1"
    for kv in filter(None, os.environ.get("STRESS_KNOBS", "").split(",")):
        name, val = kv.split("=")
        assert nn._lib.lib().nnhip_tune_set(name.encode(), int(val)) == 0, kv
    th = [threading.Thread(target=worker, args=(k,)) for k in range(NTHREADS)]
    stop = threading.Event()

    def toggler():  # one more host thread flips bit-neutral knobs while the others are inside the library (the knobs are atomics: no report, no torn setting)
        L = nn._lib.lib()
        i = 0
        while not stop.is_set():
            for name, vals in ((b"calls_bin", (0, 1)), (b"adv_block", (64, 0)), (b"adv_nontemporal", (0, -1)), (b"sort_copy", (1, 0)), (b"host_chunks", (4, 0))):
                assert L.nnhip_tune_set(name, vals[i & 1]) == 0
            i += 1
            stop.wait(0.002)
        for name, v in ((b"calls_bin", 1), (b"adv_block", 0), (b"adv_nontemporal", -1), (b"sort_copy", 0), (b"host_chunks", 0)):
            L.nnhip_tune_set(name, v)

    tg = threading.Thread(target=toggler) if os.environ.get("STRESS_TOGGLE_KNOBS") else None
    if tg:
        tg.start()
    [t.start() for t in th]
    [t.join() for t in th]
    stop.set()
    if tg:
        tg.join()
    if errors:
        for k, tb in errors:
            print("thread %d:\n%s" % (k, tb), file=sys.stderr)
        raise SystemExit(1)
    nn._lib.lib().nnhip_release()
    print("STRESS OK: %d threads x %d rounds" % (NTHREADS, ROUNDS), flush=True)


if __name__ == "__main__":
    main()

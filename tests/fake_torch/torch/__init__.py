"""TEST INFRASTRUCTURE — NOT PyTorch.  A stand-in for the few PyTorch names this repository's Python uses (device memory, streams, events, a Tensor with
numpy semantics), for ONE purpose: to run bench.py, __graft_entry__.smoke() and the -m gpu tests' Python on the ISA-backed fake node
(tests/isa_backed_node.py: tests/cpp/fake_hip.cpp under LD_PRELOAD + tools/gfx950_isa_interp.py executing the library's compiled gfx950 code), in a round in
which no GPU could be reached.  It is only ever importable in a subprocess that puts tests/fake_torch first on PYTHONPATH (tests/test_gpu_python_on_isa_node.py,
scripts/run_gpu_suite_on_isa_node.py); the package, the library and bench.py know nothing of it.  "cuda" tensors live in the fake node's tracked device
allocations (hipMalloc of the preloaded runtime), so every pointer the Python mirror hands to the C ABI is vetted like a device pointer.

torch.distributed here (distributed/__init__.py, distributed/run.py) is a launcher that starts N local ranks and a process group that exchanges .npy files: enough for
bench.py's multi-rank path to execute with its checks on.

What it cannot show: anything about PyTorch itself (its allocator, stream semantics, dtype promotion beyond numpy's), timing, RCCL / gloo."""
import ctypes as _ct
import os as _os
import sys as _sys
import time as _time

import numpy as _np

if "FAKE_HIP_LIB" not in _os.environ:
    raise ImportError("tests/fake_torch is a test stand-in for the ISA-backed fake node; it needs FAKE_HIP_LIB (see tests/test_gpu_python_on_isa_node.py)")
_F = _ct.CDLL(_os.environ["FAKE_HIP_LIB"])
__version__ = "0.0-fake-node"


# ---------------------------------------------------------------- dtypes, devices
class dtype:
    def __init__(self, name, np_dtype):
        self.name, self.np = name, _np.dtype(np_dtype)
        self.is_floating_point = self.np.kind == "f"

    def __repr__(self):
        return "torch." + self.name


float64 = double = dtype("float64", _np.float64)
float32 = float = dtype("float32", _np.float32)   # noqa: A001
int64 = long = dtype("int64", _np.int64)
int32 = int = dtype("int32", _np.int32)           # noqa: A001
int16 = dtype("int16", _np.int16)
uint8 = dtype("uint8", _np.uint8)
int8 = dtype("int8", _np.int8)
bool = dtype("bool", _np.bool_)                   # noqa: A001
_DT = {d.np: d for d in (float64, float32, int64, int32, int16, uint8, int8, bool)}
_py = __builtins__ if isinstance(__builtins__, dict) else vars(__builtins__)
_int, _float, _bool, _isinstance, _tuple, _len, _range, _max, _min = (_py[k] for k in ("int", "float", "bool", "isinstance", "tuple", "len", "range", "max", "min"))


def _np_dtype(dt):
    return None if dt is None else dt.np


class device:
    def __init__(self, type, index=None):  # noqa: A002
        if _isinstance(type, device):
            type, index = type.type, type.index if index is None else index
        if ":" in type:
            type, idx = type.split(":")
            index = _int(idx)
        assert type in ("cpu", "cuda"), type
        self.type, self.index = type, index

    def __eq__(self, o):
        o = device(o) if _isinstance(o, str) else o
        return _isinstance(o, device) and (self.type, self.index) == (o.type, o.index)

    def __ne__(self, o):
        return not self == o

    def __hash__(self):
        return hash((self.type, self.index))

    def __repr__(self):
        return "device(type='%s'%s)" % (self.type, "" if self.index is None else ", index=%d" % self.index)

    __str__ = lambda self: self.type if self.index is None else "%s:%d" % (self.type, self.index)  # noqa: E731


_CPU = device("cpu")


def _as_device(d):
    if d is None:
        return _CPU
    if _isinstance(d, _int):
        return device("cuda", d)
    d = device(d) if _isinstance(d, str) else d
    if d.type == "cuda" and d.index is None:
        d = device("cuda", cuda.current_device())
    return d


class _DevBuf:
    """one allocation of the fake node's device memory"""

    def __init__(self, nbytes, index):
        prev = _ct.c_int()
        _F.hipGetDevice(_ct.byref(prev))
        assert _F.hipSetDevice(index) == 0, "no such fake device %d" % index
        p = _ct.c_void_p()
        assert _F.hipMalloc(_ct.byref(p), _ct.c_size_t(_max(nbytes, 1))) == 0
        _F.hipSetDevice(prev.value)
        self.ptr, self.nbytes = p.value, nbytes
        self.bytes = _np.ctypeslib.as_array((_ct.c_uint8 * _max(nbytes, 1)).from_address(p.value))

    def __del__(self):
        try:
            _F.hipFree(_ct.c_void_p(self.ptr))
        except Exception:  # interpreter shutdown
            pass


def _place(a, dev):
    """a fresh tensor on `dev` holding the values of numpy array `a` (C-contiguous)"""
    a = _np.asarray(a)
    if dev.type == "cpu":
        return Tensor(_np.array(a, order="C", copy=True), dev, None)
    buf = _DevBuf(a.nbytes, dev.index)
    view = buf.bytes[:a.nbytes].view(a.dtype).reshape(a.shape)
    view[...] = a
    return Tensor(view, dev, buf)


class Size(_tuple):
    def numel(self):
        return _int(_np.prod(self)) if self else 1


def _unwrap(x):
    return x._a if _isinstance(x, Tensor) else x


class Tensor:
    __array_priority__ = 100

    def __init__(self, a, dev, owner):
        self._a, self._dev, self._owner = a, dev, owner

    # ---- metadata
    shape = property(lambda s: Size(s._a.shape))
    dtype = property(lambda s: _DT[s._a.dtype])
    device = property(lambda s: s._dev)
    is_cuda = property(lambda s: s._dev.type == "cuda")
    ndim = property(lambda s: s._a.ndim)
    T = property(lambda s: Tensor(s._a.T, s._dev, s._owner))
    mT = T
    requires_grad = False

    def data_ptr(self):
        return self._a.ctypes.data

    def dim(self):
        return self._a.ndim

    def numel(self):
        return _int(self._a.size)

    def nelement(self):
        return _int(self._a.size)

    def element_size(self):
        return self._a.itemsize

    def size(self, i=None):
        return Size(self._a.shape) if i is None else self._a.shape[i]

    def stride(self, i=None):
        st = _tuple(s // self._a.itemsize for s in self._a.strides)
        return st if i is None else st[i]

    def is_contiguous(self):
        return _bool(self._a.flags.c_contiguous)

    def storage_offset(self):
        return 0

    def __len__(self):
        return self._a.shape[0]

    def __repr__(self):
        return "fake_tensor(%r, device=%s)" % (self._a, self._dev)

    # ---- placement
    def _like(self, a):
        return _place(a, self._dev)

    def to(self, *args, **kw):
        dev, dt = kw.get("device"), kw.get("dtype")
        for x in args:
            if _isinstance(x, dtype):
                dt = x
            elif _isinstance(x, Tensor):
                dev, dt = x.device, x.dtype
            else:
                dev = x
        dev = self._dev if dev is None else _as_device(dev)
        a = self._a if dt is None or dt.np == self._a.dtype else self._a.astype(dt.np)
        if dev == self._dev and a is self._a:
            return self
        return _place(a, dev)

    def cuda(self, index=None):
        return self.to(device("cuda", cuda.current_device() if index is None else index))

    def cpu(self):
        return self.to(_CPU)

    def numpy(self):
        if self._dev.type != "cpu":
            raise TypeError("can't convert cuda:%d device type tensor to numpy. Use Tensor.cpu() to copy the tensor to host memory first." % self._dev.index)
        return self._a

    def __array__(self, dtype=None, copy=None):
        return self.numpy() if dtype is None else self.numpy().astype(dtype)

    def pin_memory(self):
        return self

    def double(self):
        return self.to(float64)

    def float(self):
        return self.to(float32)

    def long(self):
        return self.to(int64)

    def int(self):
        return self.to(int32)

    def contiguous(self):
        return self if self._a.flags.c_contiguous else self._like(_np.ascontiguousarray(self._a))

    def clone(self):
        return self._like(self._a)

    def detach(self):
        return self

    def copy_(self, src, non_blocking=False):
        self._a[...] = _unwrap(src)
        return self

    def fill_(self, v):
        self._a[...] = _unwrap(v)
        return self

    def zero_(self):
        self._a[...] = 0
        return self

    # ---- views
    def _view(self, a):
        if a.base is None and a is not self._a and not _np.shares_memory(a, self._a):
            return self._like(a)          # numpy made a copy (advanced indexing): a fresh tensor, like torch
        return Tensor(a, self._dev, self._owner)

    def __getitem__(self, k):
        k = _tuple(_unwrap(x) for x in k) if _isinstance(k, _tuple) else _unwrap(k)
        r = self._a[k]
        return self._view(r) if _isinstance(r, _np.ndarray) else self._like(_np.asarray(r))

    def __setitem__(self, k, v):
        k = _tuple(_unwrap(x) for x in k) if _isinstance(k, _tuple) else _unwrap(k)
        self._a[k] = _unwrap(v)

    def __iter__(self):
        return (self[i] for i in _range(self._a.shape[0]))

    def view(self, *shape):
        shape = shape[0] if _len(shape) == 1 and not _isinstance(shape[0], _int) else shape
        if _isinstance(shape, dtype):
            return Tensor(self._a.view(shape.np), self._dev, self._owner)
        if not self._a.flags.c_contiguous:
            raise RuntimeError("view size is not compatible with input tensor's size and stride")
        return Tensor(self._a.reshape(shape), self._dev, self._owner)

    def reshape(self, *shape):
        shape = shape[0] if _len(shape) == 1 and not _isinstance(shape[0], _int) else shape
        return self._view(self._a.reshape(shape))

    def flatten(self):
        return self.reshape(-1)

    def t(self):
        return Tensor(self._a.T, self._dev, self._owner)

    def transpose(self, i, j):
        return Tensor(_np.swapaxes(self._a, i, j), self._dev, self._owner)

    def permute(self, *dims):
        dims = dims[0] if _len(dims) == 1 and not _isinstance(dims[0], _int) else dims
        return Tensor(_np.transpose(self._a, dims), self._dev, self._owner)

    def unsqueeze(self, i):
        return Tensor(_np.expand_dims(self._a, i), self._dev, self._owner)

    def squeeze(self, i=None):
        return Tensor(_np.squeeze(self._a, i), self._dev, self._owner)

    def expand(self, *shape):
        shape = shape[0] if _len(shape) == 1 and not _isinstance(shape[0], _int) else shape
        shape = _tuple(self._a.shape[k - (_len(shape) - self._a.ndim)] if s == -1 else s for k, s in enumerate(shape))
        return Tensor(_np.broadcast_to(self._a, shape), self._dev, self._owner)

    def unbind(self, dim=0):
        return _tuple(Tensor(_np.take(self._a, i, axis=dim), self._dev, self._owner) for i in _range(self._a.shape[dim]))

    def narrow(self, dim, start, length):
        sl = [slice(None)] * self._a.ndim
        sl[dim] = slice(start, start + length)
        return Tensor(self._a[_tuple(sl)], self._dev, self._owner)

    # ---- scalars
    def item(self):
        return self._a.item()

    def tolist(self):
        return self._a.tolist()

    def __float__(self):
        return _float(self._a)

    def __int__(self):
        return _int(self._a)

    def __index__(self):
        return _int(self._a)

    def __bool__(self):
        if self._a.size != 1:
            raise RuntimeError("Boolean value of Tensor with more than one value is ambiguous")
        return _bool(self._a.reshape(-1)[0])

    # ---- reductions and elementwise (numpy does the arithmetic: device memory of the fake node is host memory)
    def _red(self, fn, dim=None, **kw):
        return self._like(_np.asarray(fn(self._a, axis=dim, **kw)))

    def max(self, dim=None):
        return self._red(_np.max, dim)

    def min(self, dim=None):
        return self._red(_np.min, dim)

    def amax(self, dim=None):
        return self._red(_np.max, dim)

    def amin(self, dim=None):
        return self._red(_np.min, dim)

    def sum(self, dim=None):
        return self._red(_np.sum, dim)

    def mean(self, dim=None):
        return self._red(_np.mean, dim)

    def any(self, dim=None):
        return self._red(_np.any, dim)

    def all(self, dim=None):
        return self._red(_np.all, dim)

    def abs(self):
        return self._like(_np.abs(self._a))

    def sqrt(self):
        return self._like(_np.sqrt(self._a))

    def exp(self):
        return self._like(_np.exp(self._a))

    def isnan(self):
        return self._like(_np.isnan(self._a))

    def isfinite(self):
        return self._like(_np.isfinite(self._a))

    def nan_to_num(self, nan=0.0, posinf=None, neginf=None):
        return self._like(_np.nan_to_num(self._a, nan=nan, posinf=posinf, neginf=neginf))

    def argsort(self, dim=-1, descending=False, stable=False):
        o = _np.argsort(-self._a if descending else self._a, axis=dim, kind="stable")
        return self._like(o.astype(_np.int64))

    def __neg__(self):
        return self._like(-self._a)

    def __invert__(self):
        return self._like(~self._a)

    def __abs__(self):
        return self.abs()


def _binop(name, np_fn, reflected=False):
    def f(self, o):
        with _np.errstate(all="ignore"):
            r = np_fn(_unwrap(o), self._a) if reflected else np_fn(self._a, _unwrap(o))
        return self._like(r)
    f.__name__ = name
    return f


def _inplace(np_fn):
    def f(self, o):
        with _np.errstate(all="ignore"):
            self._a[...] = np_fn(self._a, _unwrap(o))
        return self
    return f


for _n, _f in (("add", _np.add), ("sub", _np.subtract), ("mul", _np.multiply), ("truediv", _np.true_divide), ("floordiv", _np.floor_divide), ("pow", _np.power),
               ("mod", _np.mod), ("and", _np.bitwise_and), ("or", _np.bitwise_or), ("xor", _np.bitwise_xor), ("lshift", _np.left_shift), ("rshift", _np.right_shift)):
    setattr(Tensor, "__%s__" % _n, _binop(_n, _f))
    setattr(Tensor, "__r%s__" % _n, _binop("r" + _n, _f, True))
    setattr(Tensor, "__i%s__" % _n, _inplace(_f))
for _n, _f in (("eq", _np.equal), ("ne", _np.not_equal), ("lt", _np.less), ("le", _np.less_equal), ("gt", _np.greater), ("ge", _np.greater_equal)):
    setattr(Tensor, "__%s__" % _n, _binop(_n, _f))
Tensor.__hash__ = lambda self: id(self)
for _n, _f in (("add_", _np.add), ("sub_", _np.subtract), ("mul_", _np.multiply), ("div_", _np.true_divide)):
    setattr(Tensor, _n, _inplace(_f))
for _n, _f in (("add", _np.add), ("sub", _np.subtract), ("mul", _np.multiply), ("div", _np.true_divide), ("eq", _np.equal), ("ne", _np.not_equal), ("lt", _np.less),
               ("le", _np.less_equal), ("gt", _np.greater), ("ge", _np.greater_equal), ("maximum", _np.maximum), ("minimum", _np.minimum)):
    setattr(Tensor, _n, _binop(_n, _f))


# ---------------------------------------------------------------- factories and functions
def _shape(args):
    if _len(args) == 1 and not _isinstance(args[0], _int):
        return _tuple(args[0])
    return _tuple(args)


def _default_dt(dt):
    return float32 if dt is None else dt


def empty(*shape, dtype=None, device=None, pin_memory=False):
    return _place(_np.zeros(_shape(shape), _default_dt(dtype).np), _as_device(device))


def zeros(*shape, dtype=None, device=None):
    return _place(_np.zeros(_shape(shape), _default_dt(dtype).np), _as_device(device))


def ones(*shape, dtype=None, device=None):
    return _place(_np.ones(_shape(shape), _default_dt(dtype).np), _as_device(device))


def full(shape, fill_value, dtype=None, device=None):
    dt = dtype if dtype is not None else (float32 if _isinstance(fill_value, _float) else int64)
    return _place(_np.full(_shape((shape,)), fill_value, dt.np), _as_device(device))


def empty_like(t, dtype=None, device=None):
    return _place(_np.zeros(t._a.shape, (dtype or t.dtype).np), t.device if device is None else _as_device(device))


zeros_like = empty_like


def ones_like(t, dtype=None, device=None):
    return _place(_np.ones(t._a.shape, (dtype or t.dtype).np), t.device if device is None else _as_device(device))


def full_like(t, v, dtype=None, device=None):
    return _place(_np.full(t._a.shape, v, (dtype or t.dtype).np), t.device if device is None else _as_device(device))


def from_numpy(a):
    assert _isinstance(a, _np.ndarray), "expected np.ndarray (got %s)" % type(a).__name__
    assert a.dtype in _DT, "can't convert np.ndarray of type %s" % a.dtype
    return Tensor(a, _CPU, None)   # shares memory, like torch


def as_tensor(data, dtype=None, device=None):
    return tensor(data, dtype=dtype, device=device)


def tensor(data, dtype=None, device=None):
    if _isinstance(data, Tensor):
        data = data._a
    a = _np.array(data)
    if dtype is not None:
        a = a.astype(dtype.np)
    elif a.dtype == _np.float64 and not _isinstance(data, _np.ndarray):
        a = a.astype(_np.float32)     # python floats -> the default dtype
    return _place(a, _as_device(device))


def arange(*args, dtype=None, device=None):
    a = _np.arange(*args)
    if dtype is not None:
        a = a.astype(dtype.np)
    elif a.dtype.kind == "f":
        a = a.astype(_np.float32)
    return _place(a, _as_device(device))


def linspace(start, end, steps, dtype=None, device=None):
    return _place(_np.linspace(start, end, steps).astype(_default_dt(dtype).np), _as_device(device))


class Generator:
    def __init__(self, device=None):
        self.device, self._rng = _as_device(device), _np.random.default_rng(0)

    def manual_seed(self, s):
        self._rng = _np.random.default_rng(s)
        return self


_default_gen = Generator()


def manual_seed(s):
    _default_gen.manual_seed(s)


def rand(*shape, dtype=None, device=None, generator=None):
    return _place((generator or _default_gen)._rng.random(_shape(shape)).astype(_default_dt(dtype).np), _as_device(device))


def randn(*shape, dtype=None, device=None, generator=None):
    return _place((generator or _default_gen)._rng.standard_normal(_shape(shape)).astype(_default_dt(dtype).np), _as_device(device))


def equal(a, b):
    assert a.device == b.device, "Expected all tensors to be on the same device, but found at least two devices, %s and %s!" % (a.device, b.device)
    return a._a.shape == b._a.shape and _bool(_np.all(a._a == b._a))


def _same_device(ts):
    d = ts[0].device
    assert all(t.device == d for t in ts), "Expected all tensors to be on the same device"
    return d


def cat(ts, dim=0):
    ts = list(ts)
    return _place(_np.concatenate([t._a for t in ts], axis=dim), _same_device(ts))


def stack(ts, dim=0):
    ts = list(ts)
    return _place(_np.stack([t._a for t in ts], axis=dim), _same_device(ts))


def nan_to_num(t, nan=0.0, posinf=None, neginf=None):
    return t.nan_to_num(nan=nan, posinf=posinf, neginf=neginf)


def isnan(t):
    return t.isnan()


def isfinite(t):
    return t.isfinite()


def abs(t):  # noqa: A001
    return t.abs()


def argsort(t, dim=-1, descending=False, stable=False):
    return t.argsort(dim=dim, descending=descending, stable=stable)


def diff(t, dim=-1):
    return t._like(_np.diff(t._a, axis=dim))


def maximum(a, b):
    return a.maximum(b)


def minimum(a, b):
    return a.minimum(b)


def where(c, a, b):
    return c._like(_np.where(c._a, _unwrap(a), _unwrap(b)))


def is_tensor(x):
    return _isinstance(x, Tensor)


class no_grad:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


# ---------------------------------------------------------------- torch.cuda
class _Event:
    def __init__(self, enable_timing=False, blocking=False, interprocess=False):
        self._t = None

    def record(self, stream=None):
        self._t = _time.perf_counter()

    def synchronize(self):
        pass

    def wait(self, stream=None):
        pass

    def query(self):
        return True

    def elapsed_time(self, other):
        assert self._t is not None and other._t is not None, "elapsed_time of an event that was never recorded"
        return _max((other._t - self._t) * 1e3, 1e-6)   # host wall time of an interpreted run: positive, meaningless as a device time


class _Stream:
    def __init__(self, device=None, priority=0, _null=False):
        self.device = _as_device(device if device is not None else "cuda")
        if _null:
            self.cuda_stream = 0
        else:
            prev = cuda.current_device()
            _F.hipSetDevice(self.device.index)
            h = _ct.c_void_p()
            assert _F.hipStreamCreateWithFlags(_ct.byref(h), 1) == 0
            _F.hipSetDevice(prev)
            self.cuda_stream = h.value

    def synchronize(self):
        pass

    def wait_stream(self, s):
        pass

    def wait_event(self, e):
        pass

    def record_event(self, e=None):
        e = e or _Event()
        e.record(self)
        return e

    def query(self):
        return True

    def __eq__(self, o):
        return _isinstance(o, _Stream) and (self.cuda_stream, self.device) == (o.cuda_stream, o.device)

    def __hash__(self):
        return hash((self.cuda_stream, self.device.index))


class _StreamCtx:
    def __init__(self, s):
        self.s = s

    def __enter__(self):
        if self.s is not None:
            self.prev_dev = cuda._current
            self.prev = cuda._streams.get(self.s.device.index)
            cuda._current = self.s.device.index
            _F.hipSetDevice(cuda._current)
            cuda._streams[self.s.device.index] = self.s
        return self.s

    def __exit__(self, *a):
        if self.s is not None:
            if self.prev is None:
                cuda._streams.pop(self.s.device.index, None)
            else:
                cuda._streams[self.s.device.index] = self.prev
            cuda._current = self.prev_dev
            _F.hipSetDevice(cuda._current)
        return False


class _DeviceCtx:
    def __init__(self, d):
        self.idx = d if _isinstance(d, _int) else _as_device(d).index

    def __enter__(self):
        self.prev = cuda._current
        if self.idx is not None and self.idx >= 0:
            cuda.set_device(self.idx)
        return self

    def __exit__(self, *a):
        cuda.set_device(self.prev)
        return False


class _Cuda:
    Event, Stream = _Event, _Stream

    def __init__(self):
        self._current, self._streams = 0, {}

    def is_available(self):
        return self.device_count() > 0

    def device_count(self):
        n = _ct.c_int()
        _F.hipGetDeviceCount(_ct.byref(n))
        return n.value

    def current_device(self):
        return self._current

    def set_device(self, d):
        idx = d if _isinstance(d, _int) else _as_device(d).index
        assert 0 <= idx < self.device_count(), "invalid device ordinal"
        self._current = idx
        _F.hipSetDevice(idx)

    def device(self, d):
        return _DeviceCtx(d)

    def current_stream(self, device=None):
        idx = self._current if device is None else (device if _isinstance(device, _int) else _as_device(device).index)
        s = self._streams.get(idx)
        return s if s is not None else _Stream(device=idx, _null=True)

    def default_stream(self, device=None):
        return _Stream(device=self._current if device is None else device, _null=True)

    def stream(self, s):
        return _StreamCtx(s)

    def synchronize(self, device=None):
        pass

    def mem_get_info(self, device=None):
        return (256 << 30, 288 << 30)

    def empty_cache(self):
        pass

    def get_device_name(self, d=None):
        return "fake node device (gfx950 code interpreted on the host)"


cuda = _Cuda()
_sys.modules[__name__ + ".cuda"] = cuda


class _CNamespace:
    @staticmethod
    def _cuda_getCurrentRawStream(idx):
        return cuda.current_stream(idx).cuda_stream


_C_ns = _CNamespace()
_sys.modules[__name__ + "._C"] = _C_ns
_C = _C_ns

"""TEST INFRASTRUCTURE — see ../__init__.py.  A process group for the ranks `python -m torch.distributed.run` (run.py in this directory: a stand-in too) starts on ONE machine:
every collective is an exchange of .npy files in a directory the launcher made (rank r writes <seq>.<r>.npy, then reads everybody's).  It moves the data bench.py's
multi-rank path asks RCCL / gloo to move, in the same order, so that path's sharding, buffer rotation and checks execute; it is not a communication library."""
import os as _os
import shutil as _shutil
import tempfile as _tempfile
import time as _time

import numpy as _np

_pg = None


class ReduceOp:
    SUM = "sum"
    MAX = "max"


class _Work:
    def wait(self):
        return True

    def is_completed(self):
        return True


class _PG:
    def __init__(self, rank, world, d, own):
        self.rank, self.world, self.dir, self.own, self.seq = rank, world, d, own, 0

    def exchange(self, a):
        k = self.seq
        self.seq += 1
        a = _np.ascontiguousarray(a)
        tmp = _os.path.join(self.dir, ".%d.%d.tmp.npy" % (k, self.rank))
        _np.save(tmp, a)
        _os.replace(tmp, _os.path.join(self.dir, "%d.%d.npy" % (k, self.rank)))
        out, t0 = [], _time.time()
        for r in range(self.world):
            p = _os.path.join(self.dir, "%d.%d.npy" % (k, r))
            while not _os.path.exists(p):
                if _time.time() - t0 > float(_os.environ.get("FAKE_PG_TIMEOUT", "900")):
                    raise RuntimeError("fake process group: rank %d never reached collective %d" % (r, k))
                _time.sleep(0.002)
            out.append(_np.load(p))
        old = _os.path.join(self.dir, "%d.%d.npy" % (k - 2, self.rank))   # everybody who is at collective k - 1 has read k - 2
        if k >= 2 and _os.path.exists(old):
            _os.remove(old)
        return out


def is_available():
    return True


def is_initialized():
    return _pg is not None


def init_process_group(backend=None, rank=None, world_size=None, device_id=None, **kw):
    global _pg
    assert _pg is None, "the default process group is already initialised"
    rank = int(_os.environ["RANK"]) if rank is None else int(rank)
    world = int(_os.environ["WORLD_SIZE"]) if world_size is None else int(world_size)
    d = _os.environ.get("FAKE_PG_DIR")
    own = False
    if d is None:
        if world != 1:
            raise RuntimeError("tests/fake_torch: ranks of a process group are started by `python -m torch.distributed.run` (which makes their meeting place)")
        d, own = _tempfile.mkdtemp(prefix="fake_pg_"), True
    _pg = _PG(rank, world, d, own)
    barrier()


def destroy_process_group(group=None):
    global _pg
    if _pg is not None and _pg.own:
        _shutil.rmtree(_pg.dir, ignore_errors=True)
    _pg = None


def get_world_size(group=None):
    return _pg.world if _pg else 1


def get_rank(group=None):
    return _pg.rank if _pg else 0


def barrier(group=None, async_op=False, device_ids=None):
    _pg.exchange(_np.zeros(1))
    return _Work() if async_op else None


def all_gather_into_tensor(output_tensor, input_tensor, group=None, async_op=False):
    assert output_tensor.device == input_tensor.device and output_tensor.dtype == input_tensor.dtype
    assert output_tensor.is_contiguous() and input_tensor.is_contiguous()
    assert output_tensor.numel() == input_tensor.numel() * _pg.world, "output tensor size must be equal to world_size times input tensor size"
    parts = _pg.exchange(input_tensor._a.reshape(-1))
    output_tensor._a.reshape(-1)[...] = _np.concatenate(parts)
    return _Work() if async_op else None


def all_gather(tensor_list, tensor, group=None, async_op=False):
    parts = _pg.exchange(tensor._a)
    for t, p in zip(tensor_list, parts):
        t._a[...] = p
    return _Work() if async_op else None


def all_reduce(tensor, op=ReduceOp.SUM, group=None, async_op=False):
    parts = _pg.exchange(tensor._a)
    tensor._a[...] = _np.sum(parts, axis=0) if op == ReduceOp.SUM else _np.max(parts, axis=0)
    return _Work() if async_op else None


def broadcast(tensor, src=0, group=None, async_op=False):
    parts = _pg.exchange(tensor._a)
    tensor._a[...] = parts[src]
    return _Work() if async_op else None

"""TEST INFRASTRUCTURE — see ../__init__.py.  One process, no process group: the multi-rank paths are covered by the gloo tests (tests/test_distributed_gloo.py)."""


def is_available():
    return True


def is_initialized():
    return False


def get_world_size(group=None):
    return 1


def get_rank(group=None):
    return 0


def init_process_group(*a, **k):
    raise RuntimeError("tests/fake_torch has no process groups: run multi-rank paths under the real torch.distributed (gloo on CPU)")


class ReduceOp:
    SUM = "sum"

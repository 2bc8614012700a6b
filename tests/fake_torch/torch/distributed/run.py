"""TEST INFRASTRUCTURE — see ../__init__.py.  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr A --master-port P script.py args...` on one
machine: N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment (what the real launcher exports) and a fresh directory for __init__.py's
file-exchange process group.  Exit status: 0 if every rank's is, else the first non-zero one (the other ranks are ended)."""
import os
import shutil
import subprocess
import sys
import tempfile
import time


def main(argv):
    opts, k = {"nproc": 1, "addr": "127.0.0.1", "port": "29500"}, 0
    while k < len(argv) and argv[k].startswith("-"):
        a = argv[k]
        name, _, val = a.partition("=")
        name = name.lstrip("-").replace("_", "-")
        if name in ("standalone",):
            k += 1
            continue
        if not _:
            k += 1
            val = argv[k]
        k += 1
        if name == "nproc-per-node":
            opts["nproc"] = int(val)
        elif name == "master-addr":
            opts["addr"] = val
        elif name == "master-port":
            opts["port"] = val
        elif name in ("nnodes", "node-rank", "local-addr", "rdzv-backend", "rdzv-endpoint", "max-restarts"):
            assert name != "nnodes" or val in ("1", "1:1"), "one node"
        else:
            raise SystemExit("tests/fake_torch launcher: unknown option " + a)
    script = argv[k:]
    assert script, "no script"
    d = tempfile.mkdtemp(prefix="fake_pg_")
    procs = []
    for r in range(opts["nproc"]):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(opts["nproc"]), LOCAL_WORLD_SIZE=str(opts["nproc"]), MASTER_ADDR=opts["addr"],
                   MASTER_PORT=str(opts["port"]), FAKE_PG_DIR=d)
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable] + script, env=env))
    rc = 0
    live = list(procs)
    while live and rc == 0:
        for p in list(live):
            c = p.poll()
            if c is not None:
                live.remove(p)
                rc = rc or c
        time.sleep(0.05)
    for p in live:   # a rank failed: end the others (they would wait for it forever)
        p.terminate()
    for p in procs:
        p.wait()
    shutil.rmtree(d, ignore_errors=True)
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

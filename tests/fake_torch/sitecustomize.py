"""TEST INFRASTRUCTURE — see torch/__init__.py in this directory.  Python imports `sitecustomize` at start-up from the first directory on PYTHONPATH that has one: every
interpreter started with tests/fake_torch on its path under the preloaded fake node (the test process, and the children it starts: bench.py, harness scripts) gets its
kernel launches executed by the gfx950 interpreter from its first launch on."""
import os
import sys

if os.environ.get("FAKE_HIP_LIB") and os.environ["FAKE_HIP_LIB"] in os.environ.get("LD_PRELOAD", ""):
    _root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(1, os.path.join(_root, "tests"))
    import isa_backed_node
    ISA_NODE = isa_backed_node.attach()

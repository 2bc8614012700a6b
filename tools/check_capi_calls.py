#!/usr/bin/env python3
"""Every call of a C-ABI entry (`<anything>.nnhip_*(...)`) in the repository's Python, checked against the signature table the package loads the library with
(numericalnim_amd/_lib.py: SIGNATURES, one entry per function include/nnhip_ode.h declares): the entry exists and the call passes as many arguments as it takes.
ctypes raises on both only when the line runs — in GPU-only tests and bench legs that means on the GPU box.  Static, no device needed.
usage: check_capi_calls.py FILE...   (exit status 1 if anything is reported)"""
import ast
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def check(path, signatures):
    tree = ast.parse(open(path).read(), path)
    bad, seen = [], 0
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr.startswith("nnhip_"):
            name = node.func.attr
            seen += 1
            if name not in signatures:
                bad.append((node.lineno, "%s is not in numericalnim_amd._lib.SIGNATURES" % name))
                continue
            if any(isinstance(a, ast.Starred) for a in node.args) or node.keywords:
                continue
            want = len(signatures[name][1])
            if len(node.args) != want:
                bad.append((node.lineno, "%s takes %d arguments, the call passes %d" % (name, want, len(node.args))))
    return bad, seen


def main(paths):
    from numericalnim_amd._lib import SIGNATURES
    rc, total = 0, 0
    for p in paths:
        bad, seen = check(p, SIGNATURES)
        total += seen
        for line, msg in bad:
            print("%s:%d: %s" % (p, line, msg))
            rc = 1
    print("%d calls of C-ABI entries checked" % total, file=sys.stderr)
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

#!/usr/bin/env python3
"""A name that is read but bound nowhere — the typo that only shows when the line first runs — found without running the line: for every scope of a Python file
(symtable), every name referenced as an implicit global must be bound at module level (assignment, def, class, import, for / with target, global statement of
some function) or be a builtin.  This image has no pyflakes / pylint; rounds without GPU access leave GPU-only Python (bench.py's legs, the -m gpu tests, the A/B
scripts) unexecuted, and this is the check that still applies to it.   usage: undefined_names.py FILE...   (exit status 1 if anything is reported)"""
import builtins
import symtable
import sys


def module_bindings(top):
    names = set()
    for s in top.get_symbols():
        if s.is_assigned() or s.is_imported() or s.is_namespace() or s.is_parameter():
            names.add(s.get_name())

    def walk(t):  # `global x` inside a function binds x at module level
        for s in t.get_symbols():
            if s.is_declared_global() and s.is_assigned():
                names.add(s.get_name())
        for c in t.get_children():
            walk(c)
    walk(top)
    return names


def check(path):
    src = open(path).read()
    top = symtable.symtable(src, path, "exec")
    bound = module_bindings(top) | set(dir(builtins)) | {"__file__", "__name__", "__doc__", "__builtins__", "__spec__", "__loader__", "__package__", "__path__"}
    if "import *" in src:
        return []
    bad = []

    def walk(t):
        for s in t.get_symbols():
            if s.is_referenced() and s.is_global() and s.get_name() not in bound:
                bad.append((t.get_name(), t.get_lineno(), s.get_name()))
        for c in t.get_children():
            walk(c)
    walk(top)
    return bad


def main(paths):
    rc = 0
    for p in paths:
        for scope, line, name in check(p):
            print("%s: scope %s (line %d): name %r is read but never bound" % (p, scope, line, name))
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

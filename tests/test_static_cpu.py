"""Static check that runs on the CPU box: every global name a package function refers to exists.  The HIP-only
branches (custom autograd Functions, ctypes calls) never execute without a GPU, so a deleted helper would otherwise
surface only in the round-end GPU run."""
import ast
import builtins
import glob
import os
import symtable

from _harness import REPO

PKG = os.path.join(REPO, "ma-cjd-cooperative-jamming-decision-making-via-marl_amd")


def _module_globals(tree):
    names = set(dir(builtins)) | {"__file__", "__name__", "__doc__", "__builtins__", "__spec__", "__package__"}
    for node in ast.walk(tree):
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            names.add(node.name)
        elif isinstance(node, ast.Import):
            names.update((a.asname or a.name).split(".")[0] for a in node.names)
        elif isinstance(node, ast.ImportFrom):
            names.update(a.asname or a.name for a in node.names)
        elif isinstance(node, ast.Global):
            names.update(node.names)
    for node in tree.body:   # module-level assignments (incl. inside if / try / with / for at module level)
        for sub in ast.walk(node):
            if isinstance(sub, ast.Name) and isinstance(sub.ctx, (ast.Store, ast.Del)):
                names.add(sub.id)
    return names


def _undefined(path):
    src = open(path).read()
    tree = ast.parse(src)
    known = _module_globals(tree)
    missing = []

    def walk(table):
        for sym in table.get_symbols():
            # a name the scope treats as global (explicitly or by falling through every enclosing function scope)
            if sym.is_referenced() and sym.is_global() and sym.get_name() not in known:
                missing.append(f"{os.path.relpath(path, REPO)}: {table.get_name()} -> {sym.get_name()}")
        for child in table.get_children():
            walk(child)

    walk(symtable.symtable(src, path, "exec"))
    return missing


def test_no_undefined_global_names_in_package():
    files = sorted(glob.glob(os.path.join(PKG, "**", "*.py"), recursive=True))
    files += [os.path.join(REPO, f) for f in ("bench.py", "__graft_entry__.py", "macjd_amd.py")]
    assert len(files) > 15
    missing = [m for f in files for m in _undefined(f)]
    assert not missing, "\n".join(missing)

"""Undefined-name check over Python sources (the finding a linter calls "undefined name"; no linter is installed in the
image).  Most of the product path only runs on a GPU box, so a misspelt name in it is invisible to the CPU suite — round 2
lost a GPU call to exactly that.  tests/test_static_names.py runs this over the package, bench.py, the entry point, the
oracle and the tools.

    python tools/undefined_names.py coach_amd bench.py __graft_entry__.py oracle tools tests"""
import ast, builtins, sys, os

class Scope:
    def __init__(self, parent, kind): self.parent, self.kind, self.names = parent, kind, set()

def collect_defs(node, scope):
    """names bound directly in this scope (not nested function bodies)"""
    for child in ast.iter_child_nodes(node):
        bind(child, scope)

def bind_target(t, scope):
    for n in ast.walk(t):
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            scope.names.add(n.id)

def bind(node, scope):
    if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
        scope.names.add(node.name)
        for d in node.decorator_list: bind(d, scope)
        return
    if isinstance(node, ast.Lambda): return
    if isinstance(node, (ast.Import, ast.ImportFrom)):
        for a in node.names:
            scope.names.add((a.asname or a.name).split(".")[0])
        return
    if isinstance(node, (ast.Global, ast.Nonlocal)):
        scope.names.update(node.names); return
    if isinstance(node, ast.ExceptHandler) and node.name: scope.names.add(node.name)
    if isinstance(node, ast.Name) and isinstance(node.ctx, (ast.Store, ast.Del)): scope.names.add(node.id)
    if isinstance(node, (ast.ListComp, ast.SetComp, ast.DictComp, ast.GeneratorExp)):
        return   # own scope
    if isinstance(node, ast.NamedExpr): bind_target(node.target, scope)
    for c in ast.iter_child_nodes(node): bind(c, scope)

def check(node, scope, errs, fn):
    def lookup(name, s):
        first = True
        while s:
            if (s.kind != "class" or first) and name in s.names: return True
            first = False
            s = s.parent
        return hasattr(builtins, name) or name in ("__file__", "__name__", "__doc__", "__builtins__", "__class__")
    def visit(n, s):
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
            if not isinstance(n, ast.Lambda):
                for d in n.decorator_list: visit(d, s)
                if n.returns: visit(n.returns, s)
            for d in n.args.defaults + [x for x in n.args.kw_defaults if x]: visit(d, s)
            ns = Scope(s, "func")
            a = n.args
            for arg in a.posonlyargs + a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
                ns.names.add(arg.arg)
            body = n.body if isinstance(n.body, list) else [n.body]
            for b in body: bind(b, ns)
            for b in body: visit(b, ns)
            return
        if isinstance(n, ast.ClassDef):
            for d in n.decorator_list + n.bases + [k.value for k in n.keywords]: visit(d, s)
            ns = Scope(s, "class")
            for b in n.body: bind(b, ns)
            for b in n.body: visit(b, ns)
            return
        if isinstance(n, (ast.ListComp, ast.SetComp, ast.DictComp, ast.GeneratorExp)):
            ns = Scope(s, "func")
            for g in n.generators:
                bind_target(g.target, ns)
            for g in n.generators:
                visit(g.iter, ns)
                for i in g.ifs: visit(i, ns)
            # walrus inside
            for sub in ast.walk(n):
                if isinstance(sub, ast.NamedExpr): bind_target(sub.target, ns)
            if isinstance(n, ast.DictComp): visit(n.key, ns); visit(n.value, ns)
            else: visit(n.elt, ns)
            return
        if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load):
            if not lookup(n.id, s): errs.append("%s:%d undefined name %r" % (fn, n.lineno, n.id))
        for c in ast.iter_child_nodes(n): visit(c, s)
    visit(node, scope)

def main(paths):
    errs = []
    for root in paths:
        files = [root] if root.endswith(".py") else [os.path.join(d, f) for d, _, fs in os.walk(root) for f in fs if f.endswith(".py")]
        for fn in sorted(files):
            src = open(fn).read()
            tree = ast.parse(src, fn)
            mod = Scope(None, "module")
            star = any(isinstance(n, ast.ImportFrom) and any(a.name == "*" for a in n.names) for n in ast.walk(tree))
            if star: continue
            for b in tree.body: bind(b, mod)
            for b in tree.body: check(b, mod, errs, fn)
    return errs


if __name__ == "__main__":
    found = main(sys.argv[1:])
    print("\n".join(found))
    print(len(found), "findings")
    sys.exit(1 if found else 0)

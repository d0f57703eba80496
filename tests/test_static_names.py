"""Static check (CPU): every name a function in tests/, coach_amd/, oracle/, tools/, bench.py or __graft_entry__.py
loads must be bound somewhere that Python's scoping rules can see — the enclosing function chain, the module, or
builtins.  A pasted block that uses an undefined local (round 2: `if normalize:` in a test without that parameter)
only fails when that line runs, which for `-m gpu` tests is on the device box; this test catches it here.

Deliberately conservative (no flow analysis): a name counts as bound in a scope if it is bound ANYWHERE in that scope.
"""
import ast
import builtins
import glob
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _files():
    out = []
    for pat in ("tests/*.py", "tests/golden/*.py", "coach_amd/**/*.py", "oracle/**/*.py", "tools/**/*.py", "*.py"):
        out += glob.glob(os.path.join(ROOT, pat), recursive=True)
    return sorted(set(f for f in out if "/build/" not in f and "/gpurun_out/" not in f))


class _Scope:
    def __init__(self, node, parent, kind):
        self.node, self.parent, self.kind = node, parent, kind
        self.bound, self.loads, self.globals, self.star = set(), [], set(), False


def _bind_target(scope, t):
    for n in ast.walk(t):
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            scope.bound.add(n.id)


class _Visitor(ast.NodeVisitor):
    def __init__(self, tree):
        self.module = _Scope(tree, None, "module")
        self.scopes = [self.module]
        self.cur = self.module
        self.visit(tree)

    # ---- scope creation
    def _function(self, node):
        self.cur.bound.add(node.name)
        for d in node.decorator_list:
            self.visit(d)
        for d in node.args.defaults + [k for k in node.args.kw_defaults if k is not None]:
            self.visit(d)
        if node.returns is not None:
            self.visit(node.returns)
        s = _Scope(node, self.cur, "function")
        a = node.args
        for arg in a.posonlyargs + a.args + a.kwonlyargs + [x for x in (a.vararg, a.kwarg) if x is not None]:
            s.bound.add(arg.arg)
            if arg.annotation is not None:
                self.visit(arg.annotation)
        self._enter(s, node.body)

    visit_FunctionDef = visit_AsyncFunctionDef = _function

    def visit_Lambda(self, node):
        for d in node.args.defaults + [k for k in node.args.kw_defaults if k is not None]:
            self.visit(d)
        s = _Scope(node, self.cur, "function")
        a = node.args
        for arg in a.posonlyargs + a.args + a.kwonlyargs + [x for x in (a.vararg, a.kwarg) if x is not None]:
            s.bound.add(arg.arg)
        self._enter(s, [node.body])

    def visit_ClassDef(self, node):
        self.cur.bound.add(node.name)
        for d in node.decorator_list + node.bases + [k.value for k in node.keywords]:
            self.visit(d)
        self._enter(_Scope(node, self.cur, "class"), node.body)

    def _comprehension(self, node):
        # the first iterable is evaluated in the enclosing scope; everything else in the comprehension's own
        self.visit(node.generators[0].iter)
        s = _Scope(node, self.cur, "function")
        prev, self.cur = self.cur, s
        self.scopes.append(s)
        for i, g in enumerate(node.generators):
            _bind_target(s, g.target)
            if i:
                self.visit(g.iter)
            for c in g.ifs:
                self.visit(c)
        for part in ((node.key, node.value) if isinstance(node, ast.DictComp) else (node.elt,)):
            self.visit(part)
        self.cur = prev

    visit_ListComp = visit_SetComp = visit_GeneratorExp = visit_DictComp = _comprehension

    def _enter(self, scope, body):
        prev, self.cur = self.cur, scope
        self.scopes.append(scope)
        for st in body:
            self.visit(st)
        self.cur = prev

    # ---- bindings
    def visit_Name(self, node):
        if isinstance(node.ctx, ast.Load):
            self.cur.loads.append((node.id, node.lineno))
        else:
            self.cur.bound.add(node.id)

    def visit_NamedExpr(self, node):                       # walrus binds in the nearest non-comprehension scope
        self.visit(node.value)
        s = self.cur
        while isinstance(s.node, (ast.ListComp, ast.SetComp, ast.GeneratorExp, ast.DictComp)):
            s = s.parent
        s.bound.add(node.target.id)

    def visit_Import(self, node):
        for a in node.names:
            self.cur.bound.add((a.asname or a.name).split(".")[0])

    def visit_ImportFrom(self, node):
        for a in node.names:
            if a.name == "*":
                self.cur.star = True
            else:
                self.cur.bound.add(a.asname or a.name)

    def visit_Global(self, node):
        self.cur.globals.update(node.names)
        self.module.bound.update(node.names)

    def visit_Nonlocal(self, node):
        self.cur.globals.update(node.names)

    def visit_ExceptHandler(self, node):
        if node.name:
            self.cur.bound.add(node.name)
        self.generic_visit(node)

    def visit_MatchAs(self, node):
        if node.name:
            self.cur.bound.add(node.name)
        self.generic_visit(node)


_MODULE_DUNDERS = {"__file__", "__name__", "__doc__", "__package__", "__spec__", "__builtins__", "__class__",
                   "__path__", "__loader__", "__debug__"}


def undefined_names(source, filename="<src>"):
    tree = ast.parse(source, filename)
    v = _Visitor(tree)
    if any(s.star for s in v.scopes):
        return []
    bad = []
    for s in v.scopes:
        for name, line in s.loads:
            if name in s.globals:
                t = v.module
                ok = name in t.bound
            else:
                ok, t, first = False, s, True
                while t is not None:
                    # class bodies are not visible from nested functions; a class sees its own names
                    if (t.kind != "class" or first) and name in t.bound:
                        ok = True
                        break
                    t, first = t.parent, False
            if not ok and not hasattr(builtins, name) and name not in _MODULE_DUNDERS:
                bad.append((line, name))
    return sorted(set(bad))


def test_checker_catches_an_undefined_local_and_accepts_closures():
    assert undefined_names("def f(a):\n    if normalize:\n        return a\n") == [(2, "normalize")]
    ok = ("import os\nX = 1\n"
          "def f(a, *b, c=X, **d):\n"
          "    def g():\n        return a + h + X + len(b)\n"
          "    h = [i for i in range(3) if i + a]\n"
          "    return g, (lambda q: q + a), {k: v for k, v in d.items()}, os.sep\n"
          "class C:\n    y = 2\n    z = y + 1\n    def m(self):\n        return X\n")
    assert undefined_names(ok) == []
    assert undefined_names("class C:\n    y = 2\n    def m(self):\n        return y\n") == [(4, "y")]


@pytest.mark.parametrize("path", _files(), ids=lambda p: os.path.relpath(p, ROOT))
def test_no_undefined_names(path):
    with open(path) as f:
        src = f.read()
    bad = undefined_names(src, path)
    assert not bad, "undefined names in %s: %s" % (os.path.relpath(path, ROOT), bad)

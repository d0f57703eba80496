"""No undefined names anywhere in the Python sources (tools/undefined_names.py): the GPU-only paths of the package are
not executed by the CPU suite, so a misspelt name there would otherwise surface on the GPU box."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _checker():
    spec = importlib.util.spec_from_file_location("undefined_names", os.path.join(ROOT, "tools", "undefined_names.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_the_checker_sees_an_undefined_name(tmp_path):
    f = tmp_path / "bad.py"
    f.write_text("import os\n\ndef f(a):\n    return a + missing + os.sep\n\nclass K:\n    z = 1\n    def m(self):\n        return z\n")
    errs = _checker().main([str(f)])
    assert len(errs) == 2 and "'missing'" in errs[0] and "'z'" in errs[1]


def test_no_undefined_names_in_the_sources():
    paths = [os.path.join(ROOT, p) for p in ("coach_amd", "bench.py", "__graft_entry__.py", "oracle", "tools", "tests")]
    errs = _checker().main(paths)
    assert not errs, "\n".join(errs)

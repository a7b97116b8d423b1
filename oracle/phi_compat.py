"""Import the reference's evaluate_solver.py + its vendored PhiFlow 1.x (read-only, where they lie under
/root/reference) under this image's Python 3.10 / numpy 2.x.  Used by oracle/make_smoke_solver_fixture.py only.
TEST INFRASTRUCTURE -- see oracle/__init__.py.

Two removed-behaviour shims, no arithmetic:
  * collections.Iterable alias (removed in Python 3.10)
  * numpy < 1.23 treated a LIST index holding slices as a tuple index; numpy 2 raises.
    An import hook wraps every non-trivial subscript of the phi.* modules (and evaluate_solver)
    in _ix(), which turns such a list into the tuple old numpy used.
"""
import ast, collections, collections.abc, importlib.abc, importlib.machinery, sys, types, builtins

collections.Iterable = collections.abc.Iterable

def _ix(i):
    if isinstance(i, list) and any(isinstance(e, (slice, type(None), type(Ellipsis), list, tuple)) for e in i):
        return tuple(i)
    return i
builtins._phi_ix = _ix

class _T(ast.NodeTransformer):
    def visit_Subscript(self, node):
        self.generic_visit(node)
        s = node.slice
        if isinstance(s, (ast.Slice, ast.Constant, ast.Tuple)):
            return node
        if isinstance(s, ast.UnaryOp) and isinstance(s.operand, ast.Constant):
            return node
        node.slice = ast.copy_location(ast.Call(func=ast.Name(id="_phi_ix", ctx=ast.Load()), args=[s], keywords=[]), s)
        return node

class _Loader(importlib.machinery.SourceFileLoader):
    def get_code(self, fullname):          # never the cached bytecode of the untransformed source
        path = self.get_filename(fullname)
        return self.source_to_code(self.get_data(path), path)
    def source_to_code(self, data, path, *, _optimize=-1):
        tree = ast.parse(data, filename=path)
        tree = ast.fix_missing_locations(_T().visit(tree))
        return compile(tree, path, "exec", dont_inherit=True, optimize=_optimize)

class _Finder(importlib.abc.MetaPathFinder):
    def __init__(self, root): self.root = root
    def find_spec(self, name, path, target=None):
        if not (name == "phi" or name.startswith("phi.") or name == "evaluate_solver"):
            return None
        spec = importlib.machinery.PathFinder.find_spec(name, path if path else [self.root])
        if spec is None or not isinstance(spec.loader, importlib.machinery.SourceFileLoader):
            return spec
        spec.loader = _Loader(spec.loader.name, spec.loader.path)
        return spec

def install(root):
    sys.dont_write_bytecode = True
    sys.meta_path.insert(0, _Finder(root))
    for n in ("imageio",):
        if n not in sys.modules:
            m = types.ModuleType(n); m.__spec__ = importlib.machinery.ModuleSpec(n, None); sys.modules[n] = m

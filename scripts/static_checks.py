"""Static checks that run WITHOUT a GPU over code that only ever executes ON the GPU box (the `-m gpu` tests, bench.py, smoke()).

Round 4 lost 30 driver-run GPU tests to a two-line edit nobody could execute here: a function-local `import gc` in a test whose module binds
`import golden_cases as gc` made every earlier `gc.…` in that function an UnboundLocalError.  Two checks catch that class of defect on the CPU:

* `compile_all()` byte-compiles every Python source of the repository (syntax errors);
* `use_before_local_binding()` walks every function scope with `ast`: a name that the function binds itself (import, assignment, for / with / except
  target, nested def) is a LOCAL for the whole function, so a read on an earlier line than its first binding raises at run time — the more
  surprising when the same name exists at module level.  Reads inside loops that are fed by a binding further down in the same loop are the
  one legitimate pattern; they are recognised (the binding sits inside a loop that also contains the read) and not reported.

Used by tests/test_static_checks.py (CPU suite) and by __graft_entry__.build()."""
import ast
import builtins
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP_DIRS = {".git", "gpurun_out", "__pycache__", "build", "golden_tmp", ".pytest_cache"}


def python_sources(root=ROOT):
    out = []
    for d, dirs, files in os.walk(root):
        dirs[:] = sorted(x for x in dirs if x not in SKIP_DIRS)
        for f in sorted(files):
            if f.endswith(".py"):
                out.append(os.path.join(d, f))
    return out


def compile_all(root=ROOT):
    """[(path, message)] for every source that does not byte-compile (nothing is written next to the sources)."""
    bad = []
    for p in python_sources(root):
        try:
            with open(p, "rb") as f:
                compile(f.read(), p, "exec")
        except (SyntaxError, ValueError) as e:
            bad.append((p, str(e)))
    return bad


class _Scope(ast.NodeVisitor):
    """Bindings and reads of ONE function scope (nested function / class / lambda / comprehension bodies are separate scopes)."""

    def __init__(self, fn):
        self.binds = {}        # name -> [(lineno, col, loop_stack)]
        self.loads = {}        # name -> [(lineno, col, loop_stack)]
        self.declared = set()  # global / nonlocal
        self.loops = []
        a = fn.args
        self.params = {x.arg for x in a.posonlyargs + a.args + a.kwonlyargs} | ({a.vararg.arg} if a.vararg else set()) | ({a.kwarg.arg} if a.kwarg else set())
        for st in fn.body:
            self.visit(st)

    def _bind(self, name, node):
        self.binds.setdefault(name, []).append((node.lineno, node.col_offset, tuple(self.loops)))

    # --- scopes that are not ours
    def visit_FunctionDef(self, node):
        self._bind(node.name, node)
        for d in node.decorator_list:
            self.visit(d)
        for d in node.args.defaults + [x for x in node.args.kw_defaults if x is not None]:
            self.visit(d)
    visit_AsyncFunctionDef = visit_FunctionDef

    def visit_ClassDef(self, node):
        self._bind(node.name, node)
        for d in node.decorator_list + node.bases:
            self.visit(d)

    def visit_Lambda(self, node):
        pass

    def _comp(self, node):
        self.visit(node.generators[0].iter)      # only the outermost iterable is evaluated in the enclosing scope
    visit_ListComp = visit_SetComp = visit_DictComp = visit_GeneratorExp = _comp

    # --- bindings
    def visit_Import(self, node):
        for al in node.names:
            self._bind((al.asname or al.name).split(".")[0], node)

    def visit_ImportFrom(self, node):
        for al in node.names:
            if al.name != "*":
                self._bind(al.asname or al.name, node)

    def visit_Global(self, node):
        self.declared.update(node.names)
    visit_Nonlocal = visit_Global

    def visit_Name(self, node):
        if isinstance(node.ctx, ast.Load):
            self.loads.setdefault(node.id, []).append((node.lineno, node.col_offset, tuple(self.loops)))
        else:
            self._bind(node.id, node)

    def visit_ExceptHandler(self, node):
        if node.name:
            self._bind(node.name, node)
        self.generic_visit(node)

    def visit_Assign(self, node):           # value first, as at run time (x = x + 1 reads before it binds)
        self.visit(node.value)
        for t in node.targets:
            self.visit(t)

    def visit_AugAssign(self, node):
        self.visit(node.value)
        if isinstance(node.target, ast.Name):
            self.loads.setdefault(node.target.id, []).append((node.lineno, node.col_offset, tuple(self.loops)))
        self.visit(node.target)

    def visit_AnnAssign(self, node):
        if node.value is not None:
            self.visit(node.value)
            self.visit(node.target)

    def visit_NamedExpr(self, node):
        self.visit(node.value)
        self.visit(node.target)

    def _loop(self, node):
        if isinstance(node, (ast.For, ast.AsyncFor)):
            self.visit(node.iter)
        self.loops.append(id(node))
        if isinstance(node, (ast.For, ast.AsyncFor)):
            self.visit(node.target)
        else:
            self.visit(node.test)
        for st in node.body:
            self.visit(st)
        self.loops.pop()
        for st in node.orelse:
            self.visit(st)
    visit_For = visit_AsyncFor = visit_While = _loop


def _functions(tree):
    for node in ast.walk(tree):
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef)):
            yield node


def _module_names(tree):
    names = set(dir(builtins))
    for st in ast.walk(tree):
        if isinstance(st, ast.Import):
            names.update((al.asname or al.name).split(".")[0] for al in st.names)
        elif isinstance(st, ast.ImportFrom):
            names.update(al.asname or al.name for al in st.names)
    for st in tree.body:
        for n in ast.walk(st) if not isinstance(st, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)) else [st]:
            if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store):
                names.add(n.id)
            elif isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                names.add(n.name)
    return names


def use_before_local_binding(path, source=None):
    """[(path, function, name, line of the read, line of the first local binding, shadows a module-level name?)]"""
    src = source if source is not None else open(path).read()
    tree = ast.parse(src, filename=path)
    mod = _module_names(tree)
    out = []
    for fn in _functions(tree):
        sc = _Scope(fn)
        for name, binds in sc.binds.items():
            if name in sc.declared or name in sc.params or name not in sc.loads:
                continue
            first = min(binds)
            for (ln, col, loops) in sc.loads[name]:
                if (ln, col) >= first[:2]:
                    continue
                # a read inside a loop that a LATER binding in the same loop feeds on the next iteration is legitimate only if something
                # bound the name before the loop was entered; that cannot be the case here (`first` is the earliest binding), unless the
                # read is guarded — report only when the name also exists at module level or no binding shares a loop with the read
                fed_by_loop = any(set(loops) & set(bl) for (_, _, bl) in binds)
                if fed_by_loop and name not in mod:
                    continue
                out.append((os.path.relpath(path, ROOT) if os.path.isabs(path) else path, fn.name, name, ln, first[0], name in mod))
    return out


def run(root=ROOT):
    problems = ["does not compile: %s: %s" % pm for pm in compile_all(root)]
    for p in python_sources(root):
        try:
            found = use_before_local_binding(p)
        except SyntaxError:
            continue            # already reported
        for (f, fn, name, ln, first, shadows) in found:
            problems.append("%s:%d: `%s` is read in %s() before its first local binding on line %d%s" %
                            (f, ln, name, fn, first, " (and shadows a module-level / builtin name)" if shadows else ""))
    return problems


if __name__ == "__main__":
    probs = run()
    print("\n".join(probs) if probs else "static checks ok (%d sources)" % len(python_sources()))
    sys.exit(1 if probs else 0)

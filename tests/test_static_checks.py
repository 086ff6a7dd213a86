"""CPU-side guard for code that only executes on the GPU box (scripts/static_checks.py): every Python source byte-compiles, and no function reads a
name before the line that makes it a local (the round-4 defect: `import gc` inside a test whose module binds `golden_cases as gc`, which cost the
driver 30 unexecuted GPU tests).  The checker itself is tested on that exact pattern and on the legitimate look-alikes."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts"))
import static_checks as sc  # noqa: E402

ROOT = sc.ROOT

R4_DEFECT = '''
import golden_cases as gc
def test():
    def run(ckpt):
        batch = gc.train_batch(B=4)
        import gc
        gc.collect()
        return batch
    return run(True)
'''

LEGITIMATE = '''
import os
def a(n):
    prev = None
    for i in range(n):
        if prev is not None:
            print(prev)
        prev = i
def b(xs):
    for x in xs:
        if x:
            y = x
        else:
            print(y)        # fed by an earlier iteration: not decidable statically, not reported (y is no module-level name)
def c():
    os = __import__("os")   # shadows, but binds before any read
    return os.sep
def d(v):
    v = v + 1               # parameter
    return [q for q in range(v)]
def e():
    global os
    print(os.sep)
    import os
'''


def test_checker_reports_the_round4_defect_and_not_the_lookalikes():
    found = sc.use_before_local_binding("x.py", R4_DEFECT)
    assert [(f[1], f[2], f[3], f[4], f[5]) for f in found] == [("run", "gc", 5, 6, True)], found
    assert sc.use_before_local_binding("y.py", LEGITIMATE) == []
    assert sc.use_before_local_binding("z.py", "import math\ndef f():\n    print(math.pi)\n    math = 3\n")[0][2] == "math"


def test_every_python_source_compiles_and_reads_no_local_before_its_binding():
    problems = sc.run(ROOT)
    assert not problems, "\n".join(problems)
    srcs = [os.path.relpath(p, ROOT) for p in sc.python_sources(ROOT)]
    for must in ("bench.py", "__graft_entry__.py", "tests/test_train_gpu.py", "tests/test_wgrad_gpu.py", "diffusion-e2e-ft_amd/training.py"):
        assert must in srcs, must


def test_gpu_suite_collects_without_a_gpu():
    """`pytest -m gpu --collect-only` imports every GPU test module here: import errors and fixture typos in code the driver runs at round end."""
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-m", "gpu", "--collect-only", "-q", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    last = [ln for ln in r.stdout.strip().splitlines() if "selected" in ln or "collected" in ln][-1]
    n = int(last.split("/")[0].split()[0])
    assert n >= 582, last          # the round-4 suite; the count only grows

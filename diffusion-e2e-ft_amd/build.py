"""Build libe2eft.so (HIP, gfx950 only) in-tree with hipcc.

The shared library is the product's compute path; nothing here falls back to PyTorch or to the oracle.
`python -m diffusion_e2e_ft_amd.build` or `__graft_entry__.build()` run this; hipcc cross-compiles without a GPU.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libe2eft.so")
SOURCES = ["api.hip", "igemm.hip", "igemm2.hip", "igemm5.hip", "igemm6.hip", "convin.hip", "norm.hip", "attn.hip", "attn32.hip", "attn512.hip", "attn_bwd.hip", "elementwise.hip", "loss.hip", "bwd.hip", "wgrad.hip", "ensemble.hip", "dataprep.hip", "narrow.hip", "dataaug.hip", "evalmetrics.hip", "prepost.hip", "f32split.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]   # exports = what include/*.h declares
# Per-file flags.  The implicit-GEMM files are built without the SLP vectorizer: it turns the epilogue's per-column fp32
# arithmetic into v_pk_add_f32 with operand swizzles (op_sel:[0,1] — the low result lane reads the HIGH dword of src1), and on
# gfx950 that form sporadically read 0.0 in lanes 48-63 while a wave of ANOTHER workgroup on the same SIMD was inside its
# MFMA / LDS-DMA k-loop (two 4-wave workgroups per CU).  Found as one GroupNorm-statistics row in ~1e-5 tiles using pivot 0;
# plain v_sub_f32 / v_add_f32 never showed it (scripts/stress_conv_stats.py).
# attn.hip / attn512.hip: packed fp32 adds beside MFMAs cost more than the scalar ones they replace (the kernel is one wave per SIMD, every issue slot counts)
EXTRA_FLAGS = {"attn.hip": ["-fno-slp-vectorize"], "attn32.hip": ["-fno-slp-vectorize"], "attn512.hip": ["-fno-slp-vectorize"], "igemm.hip": ["-fno-slp-vectorize"], "igemm2.hip": ["-fno-slp-vectorize"], "igemm5.hip": ["-fno-slp-vectorize"], "igemm6.hip": ["-fno-slp-vectorize"], "f32split.hip": ["-fno-slp-vectorize"], "convin.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 to build libe2eft.so for gfx950)")


HEADERS = ["common.h", "igemm.h", "attn512_regs.inc", "igemm_persistent_epilogue.inc"]      # csrc files that are included, not compiled


def _inputs():
    """every file that determines the binary: the compiled sources, the included csrc files, the public headers — by NAME (a stray editor backup or a
    subdirectory in csrc/ changes nothing and breaks nothing; tests/test_host_logic.py checks that no csrc file is missing from these lists)"""
    inc = os.path.join(HERE, "..", "include")
    return [os.path.join(CSRC, f) for f in sorted(SOURCES + HEADERS)] + [os.path.join(inc, "e2eft.h"), os.path.join(inc, "e2eft_debug.h")]


def _newest_source_mtime():
    return max(os.path.getmtime(p) for p in _inputs() + [os.path.abspath(__file__)])


ID_MARK = b"E2EFT_BUILD_ID="      # api.hip embeds "E2EFT_BUILD_ID=<16 hex>" in the binary: the id can be read without loading the library


def source_id():
    """sha256 (first 16 hex digits) over everything that determines the binary: the sources and included files of csrc/ (by name), the public headers, the flags.  Stamped into the
    library (-DE2EFT_BUILD_ID, `e2eft_build_id()`); a measurement made with one build and quoted by another process carries it."""
    h = hashlib.sha256()
    for path in _inputs():
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(repr((SOURCES, FLAGS, sorted(EXTRA_FLAGS.items()))).encode())
    return h.hexdigest()[:16]


def built_id(path=None):
    """the source id the library on disk was stamped with (what `e2eft_build_id()` returns once it is loaded), read from the file's bytes; None if absent"""
    try:
        with open(path or LIB, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    i = blob.find(ID_MARK)
    if i < 0:
        return None
    tail = blob[i + len(ID_MARK):i + len(ID_MARK) + 16]
    return tail.decode("ascii", "replace") if len(tail) == 16 and all(c in b"0123456789abcdef" for c in tail) else None


def needs_build():
    """the library is missing, older than a source, or stamped with another source id (a checkout that moved mtimes backwards)"""
    return not os.path.exists(LIB) or os.path.getmtime(LIB) < _newest_source_mtime() or built_id() != source_id()


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    # one builder at a time per tree: under a multi-rank launch every rank may find the library stale at once (`_lib.load()` rebuilds on an id mismatch)
    # and would run hipcc into the same build/ and lib/ directories; the others wait on the lock and then find the library current
    import fcntl
    with open(os.path.join(OBJDIR, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return LIB
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    hipcc = _hipcc()
    sid = source_id()

    def compile_one(src):
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        srcp = os.path.join(CSRC, src)
        stamp = ['-DE2EFT_BUILD_ID="%s"' % sid] if src == "api.hip" else []      # api.hip carries the id: it is recompiled on every build (2 s)
        deps = [srcp] + [os.path.join(CSRC, f) for f in HEADERS] + [os.path.join(HERE, "..", "include", "e2eft.h"), os.path.join(HERE, "..", "include", "e2eft_debug.h"), os.path.abspath(__file__)]
        if not force and not stamp and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(d) for d in deps):
            return obj
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + stamp + ["-c", srcp, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    tmp = LIB + ".tmp"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    os.replace(tmp, LIB)
    assert built_id() == sid, (built_id(), sid)
    if verbose:
        print("built", LIB, "build id", sid)
    return LIB


def build_stamps(verbose=True):
    """Instrumented twin of the library (lib/libe2eft_stamps.so, -DE2EFT_STAMPS): igemm2 records per-workgroup phase clocks that
    scripts/stamp_bench.py reads back.  Load it with E2EFT_LIB=<path>.  Never used by the product path or the tests."""
    hipcc = _hipcc()
    out = os.path.join(LIBDIR, "libe2eft_stamps.so")
    objdir = os.path.join(OBJDIR, "stamps")
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        r = subprocess.run([hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-DE2EFT_STAMPS", "-c", os.path.join(CSRC, src), "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", out)
    return out


if __name__ == "__main__":
    if "--stamps" in sys.argv:
        build_stamps()
    else:
        build(force="--force" in sys.argv)

"""Import the reference toolbox (Python 2 source under /root/reference) in THIS container.

TEST INFRASTRUCTURE ONLY.  Nothing under ``lyssandra_amd/`` may import this file.

The reference is Python-2 source.  It cannot travel to the GPU box (neither as
source nor as bytecode), so this loader exists for exactly two purposes, both of
which run only where ``/root/reference`` is mounted:

  * validating ``oracle/lyssa_oracle.py`` (our own CPU restatement), and
  * generating the golden vectors under ``tests/golden/`` (``oracle/make_golden.py``).

What it does:
  1. copies ``/root/reference/lyssa`` into a throw-away temp dir (never into the repo),
  2. runs ``lib2to3`` over the copy (print statements, xrange, iteritems, raw_input,
     implicit relative imports),
  3. applies mechanical shims needed by modern numpy / PyYAML / Python 3:
       - ``yaml.load(handle)`` -> ``yaml.safe_load(handle)``
       - ``open(fname, 'wa')`` -> ``open(fname, 'a')``
       - ``batch_omp`` j==1 branch: ragged nested list assigned into ``L[:2,:2]``
         (legal on numpy 1.12, ValueError on numpy 2) -> take the scalars
       - ``if init_dict == 'data'`` with an ndarray argument -> isinstance guard
       - ``type(data) is np.core.memmap`` -> ``np.memmap`` (numpy 2 removed np.core alias warning)
       - ``img[slices]`` with a LIST of slices (utils/img.py:462) -> ``img[tuple(slices)]`` (numpy >= 1.23)
       - ``sklearn.grid_search`` / ``sklearn.cross_validation`` imports of classify.py -> ``sklearn.model_selection``
     None of them changes the arithmetic of the hot path.
  4. points HOME at the temp dir (import-time side effects create ``~/lyssa_files``),
  5. imports the converted package and returns the module.
"""
import importlib
import os
import re
import shutil
import subprocess
import sys
import tempfile

REFERENCE_ROOT = "/root/reference"

_cached = None


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "lyssa"))


def _patch(path, subs):
    with open(path, "r") as f:
        src = f.read()
    for pat, rep, must in subs:
        new, n = re.subn(pat, rep, src, flags=re.M)
        if must and n == 0:
            raise RuntimeError("shim pattern %r did not match in %s" % (pat, path))
        src = new
    with open(path, "w") as f:
        f.write(src)


def load_reference():
    """Return the imported, py3-converted ``lyssa`` package (cached per process)."""
    global _cached
    if _cached is not None:
        return _cached
    if not reference_available():
        raise RuntimeError("reference not mounted at %s" % REFERENCE_ROOT)
    tmp = tempfile.mkdtemp(prefix="lyssa_ref_py3_")
    dst = os.path.join(tmp, "lyssa")
    shutil.copytree(os.path.join(REFERENCE_ROOT, "lyssa"), dst,
                    ignore=shutil.ignore_patterns("*.png", "*.pyc", "__pycache__"))
    shutil.copy(os.path.join(REFERENCE_ROOT, "config.yml"), os.path.join(tmp, "config.yml"))
    subprocess.run([sys.executable, "-W", "ignore", "-m", "lib2to3", "-w", "-n", dst],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    # --- shims -----------------------------------------------------------------
    _patch(os.path.join(dst, "utils", "config.py"),
           [(r"yaml\.load\(handle\)", "yaml.safe_load(handle)", True)])
    _patch(os.path.join(dst, "utils", "workspace.py"),
           [(r"yaml\.load\(handle\)", "yaml.safe_load(handle)", True),
            (r"'wa'", "'a'", True)])
    _patch(os.path.join(dst, "utils", "__init__.py"),
           [(r"'wa'", "'a'", False),
            (r"np\.core\.memmap", "np.memmap", False)])
    _patch(os.path.join(dst, "utils", "img.py"),
           [(r"img\[slices\]", "img[tuple(slices)]", True),       # list-of-slices indexing: removed in numpy >= 1.23
            (r"np\.core\.memmap", "np.memmap", False)])
    _patch(os.path.join(dst, "sparse_coding.py"),
           [(r"^(\s+)w = g\n(\s+)v = w \* w\n",
             r"\1w = float(g[0])\n\2v = w * w\n", True)])
    _patch(os.path.join(dst, "dict_learning", "ksvd.py"),
           [(r"if init_dict == 'data':", "if isinstance(init_dict, str) and init_dict == 'data':", True)])
    # sklearn >= 0.20 moved ParameterGrid / StratifiedKFold (needed only so that `lyssa.classify`, and with it
    # `lyssa.dict_learning.lc_ksvd`, import; the fixtures call `lc_ksvd` / `lc_ksvd_predict` directly)
    _patch(os.path.join(dst, "classify.py"),
           [(r"from sklearn\.grid_search import ParameterGrid", "from sklearn.model_selection import ParameterGrid", True),
            (r"from sklearn\.cross_validation import StratifiedKFold", "from sklearn.model_selection import StratifiedKFold",
             True)])
    # config.yml points at non-existent paths; keep 'paths' inside the temp dir
    with open(os.path.join(tmp, "config.yml"), "w") as f:
        f.write("paths:\n  - %s\n" % os.path.join(tmp, "workspaces"))
    os.environ["HOME"] = tmp
    sys.path.insert(0, tmp)
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mod = importlib.import_module("lyssa")
            importlib.import_module("lyssa.sparse_coding")
            importlib.import_module("lyssa.dict_learning.ksvd")
            importlib.import_module("lyssa.dict_learning.online_dict_learn")
            importlib.import_module("lyssa.dict_learning.utils")
            importlib.import_module("lyssa.dict_learning.gradient_descent")
    finally:
        pass
    _cached = mod
    return mod


if __name__ == "__main__":
    m = load_reference()
    print("imported reference from", m.__path__)

"""Build libscade_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Every ``csrc/*.hip`` is compiled to its own object (in parallel) and linked; an object is reused
only when the SHA-256 of its source, of every header in ``csrc/`` and of the flag list matches the
stamp written next to it, so a stale or foreign binary is never mistaken for a build.
``python -m scade_amd.build --force`` recompiles everything.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libscade_hip.so")
SOURCES = ["capi.hip", "mlp_fwd.hip", "mlp_bwd.hip", "mlp_wgrad2.hip", "mlp_fwd_f16.hip", "mlp_bwd_f16.hip", "mlp_fwd_lp.hip",
           "mlp_bwd_lp.hip", "mlp_pack_step.hip", "step_finish.hip", "ray_ops.hip", "train_loss.hip", "optim.hip"]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]
LDFLAGS = ["--offload-arch=gfx950", "-fPIC", "-shared"]
MIN_KERNELS_SCANNED = 100       # the library holds ~250 kernel instantiations; the hazard scan must have seen them
# kernel experiments (same-box A/B): SCADE_AB_FLAGS="-DSOMETHING" SCADE_AB_OUT=tools/scratch/ab1 python -m
# scade_amd.build builds a VARIANT library + objects under that directory; run with SCADE_LIB=<dir>/libscade_hip.so
if os.environ.get("SCADE_AB_OUT"):
    CFLAGS = CFLAGS + os.environ.get("SCADE_AB_FLAGS", "").split()
    LIBDIR = os.path.abspath(os.environ["SCADE_AB_OUT"])
    OBJDIR = os.path.join(LIBDIR, "obj")
    LIB = os.path.join(LIBDIR, "libscade_hip.so")


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _digest(src):
    h = hashlib.sha256()
    h.update(" ".join(CFLAGS).encode())
    for f in [src] + sorted(f for f in os.listdir(CSRC) if f.endswith(".h")):
        h.update(f.encode())
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _obj(src):
    return os.path.join(OBJDIR, src.replace(".hip", ".o"))


def _stale(src):
    o = _obj(src)
    try:
        return not os.path.exists(o) or open(o + ".sha256").read().strip() != _digest(src)
    except OSError:
        return True


def _link_stamp():
    return hashlib.sha256("".join(_digest(s) for s in SOURCES).encode()).hexdigest()


def needs_build():
    if not os.path.exists(LIB) or any(_stale(s) for s in SOURCES):
        return True
    try:
        return open(LIB + ".sha256").read().strip() != _link_stamp()
    except OSError:
        return True


def _compile(src, verbose):
    cmd = [hipcc()] + CFLAGS + ["-c", os.path.join(CSRC, src), "-o", _obj(src)]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
    if r.stderr.strip() and verbose:
        print(r.stderr, file=sys.stderr)
    with open(_obj(src) + ".sha256", "w") as fh:
        fh.write(_digest(src))


def build(force=False, verbose=True):
    """-> path of the library.  Compiles what is stale (everything with ``force``) and links."""
    os.makedirs(OBJDIR, exist_ok=True)
    todo = [s for s in SOURCES if force or _stale(s)]
    if not todo and not needs_build():
        return LIB
    if todo:
        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4)) as ex:
            list(ex.map(lambda s: _compile(s, verbose), todo))
    cmd = [hipcc()] + LDFLAGS + [_obj(s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    # a > 64-bit store whose data registers the VALU writes in the next two issue slots reaches memory with the new
    # value on this part, and the compiler's hazard recognizer does not cover every form (build_checks.py)
    from . import build_checks
    if os.path.exists(build_checks.OBJDUMP):
        bad, kernels = build_checks.check(LIB)
        if kernels < MIN_KERNELS_SCANNED:
            # (compressed offload bundles, another bundle layout or target triple: the scan would pass without having
            # looked at anything)
            os.remove(LIB)
            raise RuntimeError(f"the store-data hazard scan found {kernels} gfx950 kernels in the built library "
                               f"(expected > {MIN_KERNELS_SCANNED}): code objects not extracted, nothing was checked")
        if bad:
            os.remove(LIB)
            raise RuntimeError("store-data hazard in the built kernels (common.h STORE_DATA_HOLD):\n" +
                               "\n".join(f"  {k}: {st}  ->  +{ws}: {nx}" for k, st, nx, ws in bad))
    with open(LIB + ".sha256", "w") as fh:
        fh.write(_link_stamp())
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)

"""Which build of libpermutect_amd.so runs a model's read-set kernels.

The exact-width kernel instances are compiled for ONE model shape per build of the library (csrc/pmt_device.hpp: PMT_SH_*): the
tile counts (16 features each) of the read features, the read-MLP widths, d_model / the reducer's widths and feature_dim, and the
widths themselves.  The default build carries the production hyperparameters (4, 2, 4, 1; 61, 30, 60, 10, 10).  The reference
accepts any layer list (architecture/mlp.py:32-67, parameters.py:70-156); so that another list does not drop to the generic instance
(fp32 MFMAs with every tile guarded, ~2x slower), `library_for` picks -- or builds, once, with the hipcc of the ROCm installation --
the library whose tile counts the model fills:

  * the default library, when the model fills ITS tiles (pmt_shape_id != 0);
  * a library under permutect_amd/instances/ with the model's tile counts (the widths compiled in when they are the model's, read at
    run time otherwise: pmt_shape_id 2 or 6): `make -C permutect_amd/csrc instances` builds the table of known shapes ahead of time
    (the reference's test configuration T0), `__graft_entry__.build()` calls it;
  * the WIDE build (`make wide`: activations up to 128 features, generic instances only) for a model with a layer wider than 64;
  * a library built on the spot (`make instance SHAPE=...`, one to two minutes, kept for later runs) unless PMT_JIT=0 or there is no
    hipcc -- then, and for a model that cannot fill any tile shape (a read MLP that does not start, or a reducer that does not end,
    with a Linear; widths beyond 64), the default library's generic instance, with a warning that says so.
"""
from __future__ import annotations

import ctypes as C
import glob
import os
import shutil
import subprocess
import warnings
from typing import Optional, Tuple

from permutect_amd.engine import lib as L

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INSTANCE_DIR = os.path.join(_HERE, "instances")
CSRC = os.path.join(_HERE, "csrc")


def cache_dir() -> Optional[str]:
    """PMT_INSTANCE_DIR: where per-shape libraries built on the spot go (and are looked for) instead of the package directory -- an
    installation whose package directory is read-only, or one shared by the nodes of a job (a local directory per node then)"""
    return os.environ.get("PMT_INSTANCE_DIR") or None


def _tiles(n: int) -> int:
    return (n + 15) // 16


def exact_shape_of(desc: L.PmtModel) -> Optional[Tuple[int, ...]]:
    """(NTF, NTR, NTD, NTE, F, R, D, H, E) if SOME build of the library can run the model on tile-exact instances -- the conditions of
    pmt_shape_id (pmt_host.hip) with the tile counts left open -- else None."""
    rm, red = desc.read_mlp, desc.reducer
    if rm.n_ops < 1 or red.n_ops < 1:
        return None
    first, last = rm.ops[0], red.ops[red.n_ops - 1]
    if first.kind != L.OP_LINEAR or last.kind != L.OP_LINEAR:
        return None
    lf, ll = desc.lin[first.lin[0]], desc.lin[last.lin[0]]
    ntf, ntr, ntd, nte = _tiles(desc.num_read_features), _tiles(lf.out_dim), _tiles(desc.d_model), _tiles(desc.feature_dim)

    def ops_fill(mlp, lo, hi, nt):
        for i in range(lo, hi):
            o = mlp.ops[i]
            if o.kind == L.OP_SKIP and o.n_layers > 2:  # (skip blocks of three and four layers run the generic instances: pmt_shape_id)
                return False
            for k in range(o.n_layers if o.kind == L.OP_SKIP else 1):
                ln = desc.lin[o.lin[k]]
                if _tiles(ln.in_dim) != nt or _tiles(ln.out_dim) != nt:
                    return False
        return True
    ok = (_tiles(lf.in_dim) == ntf and ops_fill(rm, 1, rm.n_ops, ntr) and _tiles(desc.read_embed_dim) == ntr and desc.d_ffn >= 2
          and ops_fill(red, 0, red.n_ops - 1, ntd) and _tiles(ll.in_dim) == ntd and _tiles(ll.out_dim) == nte)
    if not ok or max(ntf, ntr, ntd, nte) > 8:  # (more than four tiles: built with the wide build's 8-tile register arrays, csrc/Makefile)
        return None
    return (ntf, ntr, ntd, nte, desc.num_read_features, lf.out_dim, desc.d_model, desc.d_ffn // 2, desc.feature_dim)


class _BuildLock:
    """One build at a time per tree (the ranks of a data-parallel job lower the same model at the same moment: they would all run `make` in
    the same object directory).  An advisory lock on a file next to the objects; whoever gets it second finds the library built.
    (flock does not serialise the nodes of a job on a shared file system: give each node its own PMT_INSTANCE_DIR there.)"""

    def __init__(self, directory: str):
        self.directory = directory

    def __enter__(self):
        import fcntl
        os.makedirs(self.directory, exist_ok=True)
        self.f = open(os.path.join(self.directory, ".build.lock"), "w")  # (OSError on a read-only tree: the caller falls back)
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()
        return False


def _tag(shape) -> str:
    return "_".join(str(int(v)) for v in shape)


def file_build_id(path: str) -> Optional[str]:
    """pmt_build_id of a library FILE, read without mapping it (a mapped library cannot be replaced by a rebuilt one in this process)"""
    import re
    try:
        m = re.search(rb"PMT_BUILD_ID=([0-9a-f]{16})", open(path, "rb").read())
    except OSError:
        return None
    return m.group(1).decode() if m else None


def is_current(path: str) -> bool:
    """built from the same sources as the default library?  A per-shape library kept from an earlier state of the tree carries that
    state's kernels: it is rebuilt, or refused when it cannot be."""
    return file_build_id(path) == L.build_id(L.load())


def build_instance(shape, log=print) -> Optional[str]:
    """`make instance SHAPE=...` (csrc/Makefile) -- a no-op when the library is up to date with the sources; returns the library's path,
    None when it cannot be built here (PMT_JIT=0, no hipcc / make, a read-only tree without PMT_INSTANCE_DIR)"""
    if os.environ.get("PMT_JIT", "1") == "0":
        return None
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which("hipcc")) or not shutil.which("make"):
        return None
    out = cache_dir()
    path = os.path.join(out or INSTANCE_DIR, f"libpermutect_amd_{_tag(shape)}.so")
    cmd = ["make", "-C", CSRC, f"-j{min(8, os.cpu_count() or 1)}", "instance", "SHAPE=" + " ".join(str(int(v)) for v in shape)]
    if out:  # objects and library outside the package (command-line variables override the Makefile's)
        cmd += [f"IDIR={os.path.join(out, 'inst_' + _tag(shape))}", f"ILIB={path}"]
    try:
        with _BuildLock(out or CSRC):
            if not os.path.exists(path):
                log(f"permutect_amd: building the kernel instances for model shape {shape} (once; ~1-2 minutes, ~4 with more than four tiles) ...")
            res = subprocess.run(cmd, capture_output=True, text=True)
    except OSError as exc:
        warnings.warn(f"permutect_amd: cannot build kernel instances here ({exc}); set PMT_INSTANCE_DIR to a writable directory")
        return None
    if res.returncode != 0 or not os.path.exists(path):
        warnings.warn("permutect_amd: building the kernel instances failed:\n" + res.stderr[-2000:])
        return None
    return path


def widest_layer(desc: L.PmtModel) -> int:
    """the widest activation of the read-set kernels and the row MLPs behind their inputs (what PMT_MAX_WIDTH bounds: pmt_host.hip,
    pmt_model_check)"""
    w = max(desc.num_read_features, desc.d_model, desc.feature_dim)
    for i in range(desc.n_linear):
        w = max(w, desc.lin[i].out_dim)
    return w


WIDE_LIB = os.path.join(_HERE, "libpermutect_amd_wide.so")
WIDE32_LIB = os.path.join(_HERE, "libpermutect_amd_wide32.so")  # the same with two tiles per half of a gated block's hidden layer (d_ffn / 2 in 17 .. 32)


def half_tiles(desc: L.PmtModel) -> int:
    """tiles of one half of the gated blocks' hidden layer: 1, or 2 for d_ffn / 2 in 17 .. 32 (pmt_device.hpp: PMT_HT of the build that runs it)"""
    return 2 if desc.num_blocks > 0 and desc.d_ffn // 2 > 16 else 1


def wide_library(log=print, half32: bool = False) -> C.CDLL:
    """The WIDE build (activations up to 128 features, generic instances only; `half32`: with two-tile gate halves), built once with
    `make wide` / `make wide32` when it is missing"""
    path, target = (WIDE32_LIB, "wide32") if half32 else (WIDE_LIB, "wide")
    if not os.path.exists(path):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        if os.environ.get("PMT_JIT", "1") == "0" or not (os.path.exists(hipcc) or shutil.which("hipcc")) or not shutil.which("make"):
            raise L.PmtError(f"this model (a layer wider than {L.MAX_WIDTH}, or d_ffn / 2 beyond {L.MAX_HALF_FFN}) needs the {target} build of the "
                             f"library, `make -C permutect_amd/csrc {target}` (not built here, and PMT_JIT=0 or no hipcc / make to build it now)")
        with _BuildLock(CSRC):
            if not os.path.exists(path):  # (else: another process built it while this one waited for the lock)
                log(f"permutect_amd: building the {target} library (once, ~3 minutes) ...")
                res = subprocess.run(["make", "-C", CSRC, f"-j{min(8, os.cpu_count() or 1)}", target], capture_output=True, text=True)
                if res.returncode != 0 or not os.path.exists(path):
                    raise L.PmtError(f"building the {target} library failed:\n" + res.stderr[-2000:])
    return L.load(path)


def library_for(desc: L.PmtModel, log=print) -> C.CDLL:
    default = L.load()
    if "PMT_LIB" in os.environ:  # a development build named explicitly: use it as it is
        return default
    widest, ht = widest_layer(desc), half_tiles(desc)
    wide = widest > L.limits_of(default)["max_width"] or ht > 1  # (beyond the default build's compile-time limits)

    def generic():
        if not wide:
            return default
        warnings.warn(f"permutect_amd: this model (a layer wider than {L.limits_of(default)['max_width']} features, or d_ffn / 2 beyond "
                      f"{L.limits_of(default)['max_half_ffn']}) has no exact instances for its shape: it runs the WIDE build of the library, GENERIC "
                      "instances (fp32 MFMAs, 8-tile register arrays with spills: several times slower per read than exact-width kernels)")
        return wide_library(log, half32=ht > 1)
    if desc.force_shape == 2 or (not wide and default.pmt_shape_id(C.byref(desc)) != 0):
        return generic()
    shape = exact_shape_of(desc)
    if shape is None:
        warnings.warn("permutect_amd: this model cannot run on tile-exact kernel instances (the read MLP must start and the reducer end with a "
                      "Linear, each MLP keep one tile count, skip blocks hold at most two layers); it runs the GENERIC instance (fp32 MFMAs, "
                      "~2x slower)")
        return generic()
    # a library with the model's tile counts: its widths first, then any other widths (pmt_shape_id 6: widths at run time)
    dirs = ([cache_dir()] if cache_dir() else []) + [INSTANCE_DIR]
    candidates = []
    for d in dirs:
        exact = os.path.join(d, f"libpermutect_amd_{_tag(shape)}.so")
        candidates += ([exact] if os.path.exists(exact) else []) + sorted(glob.glob(os.path.join(d, f"libpermutect_amd_{_tag(shape[:4])}_*.so")))
    def fits(lib):
        lim = L.limits_of(lib)
        return lim["max_width"] >= widest and lim["max_half_ffn"] // 16 == ht and lib.pmt_shape_id(C.byref(desc)) != 0
    stale = []
    for path in dict.fromkeys(candidates):
        if not is_current(path):
            stale.append(path)
            continue
        lib = L.load(path)
        if fits(lib):
            return lib
    path = build_instance(shape, log)  # (make: rebuilds a stale library of this shape, a no-op for a current one)
    if path is not None and is_current(path):
        lib = L.load(path)
        if fits(lib):
            return lib
    if stale:
        warnings.warn("permutect_amd: kernel-instance libraries built from OTHER sources than the default library were ignored ("
                      + ", ".join(os.path.basename(p) for p in stale) + "); rebuild them: `python -c 'import __graft_entry__ as g; g.build()'`")
    if wide:
        return generic()
    warnings.warn(f"permutect_amd: no kernel instances for model shape {shape} and none could be built here (hipcc / make missing, or "
                  "PMT_JIT=0); the model runs the GENERIC instance (fp32 MFMAs, ~2x slower).  `make -C permutect_amd/csrc instance "
                  f"SHAPE=\"{' '.join(str(v) for v in shape)}\"` builds them")
    return default

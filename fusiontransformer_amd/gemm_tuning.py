"""Library-GEMM kernel selection (torch TunableOp) that is paid ONCE, not per process.

The ViT's dense GEMMs are plain library calls (hipBLASLt / rocBLAS through torch).  hipBLASLt's default heuristic picks
poor fp32 kernels for several of the trunk's shapes (26 TFLOP/s for the 2312x768x768 projection; the tuned choice
runs at ~90), so the bench and the trainer switch TunableOp on.  Tuning ~20 shapes costs tens of seconds; a job of
N ranks would pay it N times at every start.  Here the results live in ONE file next to libftx.so
(`csrc/tunableop_gfx950.csv`, committed; TunableOp validates its header against the running ROCm / hipBLASLt /
device and ignores it on a mismatch): every rank works on a private copy of it, shapes missing from it are tuned
during warm-up as before, and rank 0 writes the merged result back when asked to (`save()`)."""
from __future__ import annotations

import os
import shutil

_HERE = os.path.dirname(os.path.abspath(__file__))
SHARED = os.environ.get("FTX_TUNABLEOP_FILE", os.path.join(_HERE, "csrc", "tunableop_gfx950.csv"))
_private = None


def enable(rank: int = 0, tune_missing: bool = True) -> str:
    """Switch TunableOp on for this process; returns the private results file it reads and appends to."""
    global _private
    import torch.cuda.tunable as tunable
    tmp = os.environ.get("TMPDIR", "/tmp")
    _private = os.path.join(tmp, "ftx_tunableop_%d_rank%d.csv" % (os.getpid(), rank))
    if os.path.exists(SHARED):
        shutil.copyfile(SHARED, _private)
    tunable.enable(True)
    tunable.tuning_enable(bool(tune_missing))
    tunable.set_filename(_private)
    return _private


def n_results() -> int:
    import torch.cuda.tunable as tunable
    return len(tunable.get_results())


def save(rank: int = 0, write_shared=None) -> bool:
    """Rank 0: flush TunableOp's results to this process's private file and report whether it gained entries over the shared file.
    The shared file is only REWRITTEN when asked to -- `write_shared=True`, or FTX_TUNABLEOP_WRITE=1, or FTX_TUNABLEOP_FILE names a file
    of the caller's: the default shared file is a tracked source file inside the package, and a benchmark run must not edit the tree."""
    if rank != 0 or _private is None:
        return False
    try:
        import torch.cuda.tunable as tunable
        tunable.write_file(_private)        # TunableOp keeps new selections in memory until asked (or until exit)
    except Exception:
        pass
    if not os.path.exists(_private):
        return False
    if write_shared is None:
        write_shared = os.environ.get("FTX_TUNABLEOP_WRITE") == "1" or "FTX_TUNABLEOP_FILE" in os.environ
    try:
        new = open(_private).read()
        old = open(SHARED).read() if os.path.exists(SHARED) else ""
        if new.count("\n") > old.count("\n"):
            if write_shared:
                shutil.copyfile(_private, SHARED)
                return True
            import sys
            print("[fusiontransformer_amd] %d new library-GEMM selections were tuned in this process and not persisted (%s); set FTX_TUNABLEOP_WRITE=1 "
                  "to update %s" % (new.count("\n") - old.count("\n"), _private, SHARED), file=sys.stderr, flush=True)
    except OSError:
        pass
    return False

"""umlh -- MI355X-native UML head fine-tune hot path (ctypes binding over libumlh.so).

The product path: every compute entry point goes through the C ABI of
``include/umlh.h`` into hand-written HIP kernels for gfx950.  There is no CPU
or PyTorch-op fallback: loading fails loudly if the library is not built, and
creating a handle fails loudly without a GPU.
"""
from ._lib import (UmlhError, build_library, lib_path, load_library, OPT_IDS, PREC_IDS,  # noqa: F401
                   N_SCALARS, S_LOSS_IMG, S_LOSS_TXT, S_ACC_IMG, S_ACC_TXT, S_GSCALE_IMG, S_GSCALE_TXT,
                   S_CORRECT, S_LOSS_SUM, S_GRAD_DOT, S_GRAD_N2_IMG, S_GRAD_N2_TXT, S_GRAD_AGREE, N_CORE_SCALARS)
from .head_engine import (HeadEngine, RowBatch, column_sums, gather_rows, grad_diagnostics, optimizer_step,  # noqa: F401
                          optimizer_step_multi,
                          random_permutation, to_bf16, train_steps_grouped)
from .dp import DataParallelStepper  # noqa: F401,E402


def host_cpu_budget() -> int:
    """CPUs this process may actually use: the smaller of its affinity mask and its cgroup CPU quota."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(txt[0]) // int(txt[1])))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def _fit_host_threads() -> None:
    """torch sizes its intra-op (OpenMP) pool by the VISIBLE cores (256 on an MI355X host).  Under a cgroup quota
    of 16 CPUs every small CPU op of the training loop (the best-snapshot ``.cpu().clone()``, loader shuffles) then
    spins 256 threads, exhausts the quota and gets the whole process throttled for tens of milliseconds: measured
    350 us/step instead of 42 us/step for batch-32 steps.  Fit the pool to the budget unless the user set one."""
    import os
    import torch
    if "OMP_NUM_THREADS" in os.environ:
        return
    budget = host_cpu_budget()
    if torch.get_num_threads() > budget:
        torch.set_num_threads(budget)


_fit_host_threads()

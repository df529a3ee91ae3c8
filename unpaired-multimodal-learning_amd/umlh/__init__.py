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
                          random_permutation, to_bf16)
from .dp import DataParallelStepper  # noqa: F401,E402

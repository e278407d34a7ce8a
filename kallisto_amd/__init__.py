"""kallisto_amd -- MI355X (gfx950) implementation of the `kallisto quant` hot path.

The product is libkallisto_amd.so (HIP kernels + C++ host code behind the C ABI of include/kallisto_amd.h).  This
package is the thin Python mirror of that ABI used by the tests and bench.py; PyTorch is only used for device memory,
streams and torch.distributed.  There is no CPU path: every compute entry point fails loudly without the extension
and a GPU.
"""
from .api import (Comm, Context, Index, QuantOpts, QuantResult, KallistoAmdError, library_path, load_library, quant,  # noqa: F401
                  packed_record_words)

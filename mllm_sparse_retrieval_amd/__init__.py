"""MI355X-native learned-sparse retrieval scorer: drop-in for the sparse search step of
cjc20000323/mllm_sparse_retrieval (src/search.py, scripts/search_sparse.sh, scripts/sparse_index.sh).

Python host code over a ctypes C-ABI (include/msr.h, libmsr.so); the scoring runs in hand-written HIP kernels for
gfx950. There is no CPU scoring path and no PyTorch on the search path.
"""
from .index import (QueryBatch, SparseIndex, build_index_from_csr, build_index_from_jsonl, comm_unique_id,  # noqa: F401
                    device_copy_gbs, device_peak_rates, device_sync, runtime_info, search_laps, search_termshard_emulated_handles, set_build_option, synth_vectors)
from .searcher import Hit, JWhiteSpaceAnalyzer, LuceneImpactSearcher  # noqa: F401

__version__ = "0.1.0"

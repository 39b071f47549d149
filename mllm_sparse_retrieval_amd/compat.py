"""Names a reference user imports, in one place:

    from mllm_sparse_retrieval_amd.compat import (LuceneImpactSearcher, JWhiteSpaceAnalyzer,   # pyserini names
        sparse_search, get_run_dict, search_queries, pickle_load,                               # search.py names
        fuse, fuse_statistic, ResultRecord, write_trec_run, read_trec_run, RecallMetrics,       # hybrid.py / metrices.py
        FaissFlatSearcher)                                                                      # tevatron name
"""
from .dense import FaissFlatSearcher  # noqa: F401
from .fusion import ResultRecord, fuse, fuse_statistic, read_trec_run, write_trec_run  # noqa: F401
from .recall import RecallMetrics  # noqa: F401
from .run import get_run_dict, pickle_load, search_queries, sparse_search  # noqa: F401
from .searcher import Hit, JWhiteSpaceAnalyzer, LuceneImpactSearcher  # noqa: F401

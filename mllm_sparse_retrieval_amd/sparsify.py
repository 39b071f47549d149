"""Encode-side sparsifier on the GPU: next-token logits -> (token ids, integer weights), as the reference does before
writing corpus jsonl / building query strings:

    logits = torch.log(1 + torch.relu(logits))                                   src/model.py:104
    top_k_values, top_k_indices = logits.topk(128 or sparse_length)              src/encode.py:69-72
    values = np.rint(top_k_values.float().numpy() * 100).astype(int)             src/encode.py:75

The vocabulary-string half of get_img_valid_tokens_values (lower-casing, filter_token, dict overwrite) stays on the host.
"""
from __future__ import annotations

import numpy as np

from ._cabi import check, lib, ptr


def sparsify_logits(logits, k=128, fp16_math=None, device=0):
    """logits: [rows, vocab] float32 or float16 -> (ids uint32 [rows,k], values float32 [rows,k], weights int32 [rows,k]).
    fp16_math defaults to the input dtype (float16 input = a model running in fp16)."""
    x = np.ascontiguousarray(logits)
    if x.ndim != 2 or x.dtype not in (np.float32, np.float16):
        raise ValueError("expected a [rows, vocab] float32 / float16 array")
    is16 = x.dtype == np.float16
    if fp16_math is None:
        fp16_math = is16
    rows, vocab = x.shape
    ids = np.empty((rows, k), dtype=np.uint32)
    vals = np.empty((rows, k), dtype=np.float32)
    w = np.empty((rows, k), dtype=np.int32)
    check(lib().msr_sparsify(ptr(x), int(is16), int(bool(fp16_math)), rows, vocab, int(k), int(device), ptr(ids), ptr(vals),
                             ptr(w)))
    return ids, vals, w

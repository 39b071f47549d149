#!/usr/bin/env python3
"""Generates tests/golden/sparse_small/: a hand-checkable corpus (reference jsonl format, src/encode.py:351-359,426),
queries (query.tsv format, src/encode.py:418-424) and the expected hits.

PARITY UNPINNED: the expected hits come from the oracle's restatement of the declared contract (oracle/oracle.py,
scipy int64 and pure-Python loops, which must agree), NOT from Lucene — the reference's scorer is not runnable here
(SURVEY.md §8c). A handful of entries is additionally asserted by hand in tests/test_oracle.py.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle  # noqa: E402

OUT = os.path.join(HERE, "sparse_small")


def main():
    os.makedirs(OUT, exist_ok=True)
    lines = [
        json.dumps({"id": "1", "content": "", "vector": {"cat": 3, "dog": 2, "the": 1}}),
        json.dumps({"id": "2", "content": "", "vector": {"cat": 3, "the": 1, "fish": 5}}),
        json.dumps({"id": "10", "content": "", "vector": {"dog": 4, "the": 1, "cat": 0}}),      # weight 0 -> absent
        json.dumps({"id": "9", "content": "", "vector": {"dog": 4, "the": 1}}),                 # ties with "10"
        json.dumps({"id": "3", "content": "", "vector": {"bird": 7, "the": 1, "ġdog": 2}}),  # \\u escape in file
        json.dumps({"id": "4", "content": "x \"y\" {z}", "vector": {"the": 1, "a b": 2}}),     # whitespace key splits
        '{"id": "5", "content": "", "vector": {"the": 1, "a": 1, "dup": 1, "dup": 6}}',        # duplicate key: last wins
        '{"id": "6", "content": "", "vector": {"the": 1, "neg": -3, "flt": 2.9}, "extra": [1, {"k": null}]}',
        json.dumps({"id": "7", "content": "", "vector": {"the": 1}}),
        '{"vector": {"the": 1, "late": 9}, "id": 8, "content": ""}',                            # key order, numeric id
        "",                                                                                       # blank line
    ]
    rng = random.Random(7)
    vocab = ["w%d" % i for i in range(30)]
    for i in range(11, 52):
        vec = {"the": 1}
        for t in rng.sample(vocab, 6):
            vec[t] = rng.randint(1, 40)
        if i % 5 == 0:
            vec["dog"] = 1
        lines.append(json.dumps({"id": str(i), "content": "", "vector": vec}))
    with open(os.path.join(OUT, "corpus_0.jsonl"), "w") as f:
        f.write("\n".join(lines) + "\n")

    queries = [
        ("1001", "cat cat dog"),
        ("1002", "dog"),
        ("1003", "unicorn"),
        ("1004", "the the"),
        ("1005", "ġdog bird"),
        ("1006", "a b a"),
        ("9", "dog dog"),
        ("1007", "dup flt neg late cat"),
        ("1008", " ".join(["w3"] * 5 + ["w7"] * 2 + ["w11", "unicorn", "the"])),
        ("1009", "w1 w2 w3 w4 w5 w6 w7 w8 w9 w10 w11 w12"),
    ]
    with open(os.path.join(OUT, "query.tsv"), "w") as f:
        for qid, text in queries:
            f.write(f"{qid}\t{text}\n")

    docs = oracle.read_corpus_dir(OUT)
    ix = oracle.OracleIndex(docs)
    exp = {"n_docs": ix.n_docs, "vocab": ix.vocab, "doc_ids_by_ordinal": ix.doc_ids, "df": ix.df.tolist(), "cases": {}}
    enc = [oracle.encode_query(t) for _, t in queries]
    for drop in (True, False):
        for k in (3, 10, 100):
            ords, scores, n = oracle.search(ix, enc, k, drop_df_eq_n=drop)
            loops = oracle.search_loops(ix, enc, k, drop_df_eq_n=drop)
            hits = {}
            for i, (qid, _) in enumerate(queries):
                h = [[ix.doc_ids[int(ords[i, j])], int(scores[i, j])] for j in range(int(n[i]))]
                assert h == [[ix.doc_ids[d], s] for d, s in loops[i]], (qid, h, loops[i])
                hits[qid] = h
            exp["cases"][f"drop={int(drop)},k={k}"] = hits
    json.dump(exp, open(os.path.join(HERE, "sparse_small_expected.json"), "w"), indent=1, sort_keys=True)
    print("wrote", OUT, "docs", ix.n_docs, "terms", ix.n_terms)


if __name__ == "__main__":
    main()

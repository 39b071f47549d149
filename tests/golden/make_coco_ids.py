"""Writes tests/golden/coco_test_ids.csv: the ID columns of the reference's shipped COCO-5K test split
(/root/reference/data/coco/coco_test.csv, schema imgid,filepath,filename,caption,sentid — src/dataset.py:60-85), with the
file path, file name and caption text left empty. Data, not source: 5 000 sparse image ids (0 .. 40 503), 25 010 caption
ids of 2-6 digits (string order != numeric order: the tie rule T1 and the ordinal numbering see REAL ids), ten images
with six captions, and 149 numbers that are both an image id and a caption id (remove_query, src/search.py:72-74, then
removes a real doc). Run in the build container only: python tests/golden/make_coco_ids.py"""
import csv
import os

SRC = "/root/reference/data/coco/coco_test.csv"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "coco_test_ids.csv")

with open(SRC, newline="", encoding="utf-8") as f, open(DST, "w", newline="", encoding="utf-8") as g:
    rows = csv.reader(f)
    out = csv.writer(g, lineterminator="\n")
    out.writerow(next(rows))
    n = 0
    for row in rows:
        if row:
            out.writerow([row[0], "", "", "", row[4]])
            n += 1
print(f"{n} rows -> {DST} ({os.path.getsize(DST)} bytes)")

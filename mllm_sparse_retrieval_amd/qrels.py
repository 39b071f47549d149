"""Query/target bookkeeping of the reference's CrossModalRetrievalDataset (src/dataset.py:19-177), minus image IO.

csv schemas (src/dataset.py:60-102, read_karpathy.py:11-13):
    flickr: imgid, filename, caption, sentid                       (row[0], row[1], row[2], row[3])
    coco  : imgid, filepath/filename, ..., sentid at column 4      (row[0], row[1], row[4])
get_target (src/dataset.py:164-168): query_type 'text' -> the caption's image id; otherwise the image's caption ids.
"""
from __future__ import annotations

import csv


class CrossModalQrels:
    def __init__(self, csv_path=None, dataset_name="flickr"):
        self.img2text = {}
        self.text2img = {}
        self.img_id_list = []
        self.text_id_list = []
        if csv_path is not None:
            sent_col = 4 if dataset_name == "coco" else 3
            with open(csv_path, newline="", encoding="utf-8") as f:
                rows = csv.reader(f)
                next(rows, None)  # header
                for row in rows:
                    if row:
                        self.add(row[0], row[sent_col])

    def add(self, img_id, text_id):
        img_id, text_id = str(img_id), str(text_id)
        if img_id not in self.img2text:
            self.img2text[img_id] = []
            self.img_id_list.append(img_id)
        self.img2text[img_id].append(text_id)
        self.text_id_list.append(text_id)
        self.text2img[text_id] = img_id

    @classmethod
    def synthetic(cls, n_images, captions_per_image=5):
        """caption j <-> image j // captions_per_image."""
        q = cls()
        for j in range(n_images * captions_per_image):
            q.add(j // captions_per_image, j)
        return q

    def get_target(self, idx, query_type):
        return self.text2img[idx] if query_type == "text" else self.img2text[idx]

"""Generate tests/golden/epub_*.json.gz from the two books the reference ships (run in the build container;
/root/reference does not exist on the GPU box).  Data only: sentence lists / lengths / hashes produced by
tts_amd.epub -- no reference source text.

  art_of_war: every sentence (public-domain 1910 Giles translation) -> cfg3 input and parser regression pin
  sound_of_sirens: per-item sentence counts, character lengths and a sha256 of the sentence stream -> cfg5
                   length distribution and parser regression pin (the text itself is not redistributed)
"""
import gzip
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tts_amd.epub import EpubBook  # noqa: E402

REF = "/root/reference"
out_dir = os.path.join(ROOT, "tests", "golden")


def dump(name, obj):
    with gzip.GzipFile(os.path.join(out_dir, name), "wb", mtime=0) as f:
        f.write(json.dumps(obj, ensure_ascii=False).encode("utf-8"))


for fn, key, full in (("The-Art-of-War.epub", "art_of_war", True), ("Sound-of-Sirens.epub", "sound_of_sirens", False)):
    b = EpubBook(os.path.join(REF, fn))
    sents = b.all_sentences()
    h = hashlib.sha256("\n".join(t for _, _, t in sents).encode("utf-8")).hexdigest()
    obj = {"source": fn, "title": b.title, "n_items": len(b), "items": [h_ for h_, _ in b.items],
           "per_item": [sum(1 for i, _, _ in sents if i == k) for k in range(len(b))], "n_sentences": len(sents),
           "sha256": h, "char_lengths": [len(t) for _, _, t in sents]}
    if full:
        obj["sentences"] = [[i, t] for i, _, t in sents]
    else:
        obj["first_words"] = [t.split()[0] if t.split() else "" for _, _, t in sents[:40]]
    dump(f"epub_{key}.json.gz", obj)
    print(key, len(b), len(sents), h[:12], obj["per_item"][:12])

"""Golden vectors for the HOST rows of the path produced by the REFERENCE'S OWN
functions.  Runs in the build container only (/root/reference does not exist on the GPU box); commits data only.

The reference's sentence splitters, its float -> s16le conversion and its sentence ids are pure `re` / numpy /
hashlib code that sits inside GTK scripts which cannot be imported here (gi, kokoro_onnx, soundfile are absent).
This script reads those scripts AS TEXT, takes the definitions it needs out of the syntax tree -- by name, nothing
else of the file is compiled or run -- and evaluates them on

  * the text of The-Art-of-War.epub as tests/golden/epub_art_of_war.json.gz already holds it (chunks of 40
    sentences joined by " " and by "\\n"): outputs stored as a sha1 + count per chunk and function;
  * a list of adversarial strings (abbreviations, quotes, ellipses, newlines, empty / blank input ...): outputs
    stored in full;
  * seeded float arrays around the clipping and rounding edges of the PCM conversion: outputs stored as sha1 + head.

  speech7.py                                   TTSEngine.split_sentences           (:108-122)
  basic.tts_kokoro_2_play_pause_works.py       tokenize, f32_to_s16le              (:24-25)
  tts_16_3.py                                  split_sentences, stable_id_for_text (:27-35)
  epubviewer_06_latest_1_nov_2025_theme.py     _split_text_into_sentences          (:2458-2499)

tests/test_cpu_host_golden.py compares tts_amd.text / tts_amd.wav with the stored outputs bit for bit.
Run: python tools/make_host_fixtures.py
"""
import ast
import gzip
import hashlib
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")

WANTED = {
    "speech7.py": {"split_sentences"},
    "basic.tts_kokoro_2_play_pause_works.py": {"tokenize", "f32_to_s16le"},
    "tts_16_3.py": {"_s_re_split", "split_sentences", "stable_id_for_text"},
    "epubviewer_06_latest_1_nov_2025_theme.py": {"_split_text_into_sentences"},
}


def extract(fname, names):
    """{name: callable} for the named module-level / method definitions of a reference script."""
    with open(os.path.join(REF, fname), encoding="utf-8") as fh:
        tree = ast.parse(fh.read(), filename=fname)
    picked, found = [], set()
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names and node.name not in found:
            picked.append(node)
            found.add(node.name)
        elif isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) \
                and node.targets[0].id in names and node.targets[0].id not in found:
            picked.append(node)
            found.add(node.targets[0].id)
    missing = names - found
    assert not missing, (fname, missing)
    picked.sort(key=lambda n: n.lineno)
    mod = ast.Module(body=picked, type_ignores=[])
    ns = {"re": re, "np": np, "hashlib": hashlib}
    exec(compile(mod, fname, "exec"), ns)
    return {n: ns[n] for n in names}, {n.name if isinstance(n, ast.FunctionDef) else n.targets[0].id: n.lineno for n in picked}


ADVERSARIAL = [
    "", "   ", "\n\n", "No terminal punctuation", "One. Two! Three? Four", "Wait... what?! ok", "Trailing space.   ",
    "Dr. Smith met Mr. Jones at 5 p.m. They talked. Prof. Xavier, e.g. the boss, left.", "See Fig. 3 and eq. 4. Then No. 5 follows.",
    "He said \"Stop.\" Then left. 'Quoted start.' (Bracket start.) [Square.] `Tick.`", "Line one.\nLine two without stop\nLine three!\n\n\nLine four?",
    "lower case follows. so no split here. But Here yes. 9 lives. \"Quote\" ok.", "Mixed\ttabs.\tAnd\r\nCRLF lines.\r\nDone.",
    "Ellipsis … unicode. Em—dash. Café déjà vu! ¿Qué? Sí.", "“Smart quotes.” ‘Single.’ Plain.", "`` double grave '' \"\" doubled \"\"quotes\"\". Next.",
    "´acute´ and `grave`. Both. i.e. this one. et al. wrote it. Sept. 9 was it.", "A.B.C. Inc. bought Co. Ltd. shares. OK.", "!!! ??? ... . ! ?",
    "Sentence.Without space.After stop. Yes.", "  Leading and trailing whitespace around. Several   spaces   inside.  ", "St. Mt. vs. etc. pp. Rev. Lt. Col. Gen. Sgt. Capt. Sen. Rep. Gov. Pres. Ave. Rd. Blvd. End.",
    "Jan. Feb. Mar. Apr. Jun. Jul. Aug. Sep. Oct. Nov. Dec. Corp. PhD. MD. BA. MA. Bros. viz. cf. Done.", "mr. lower abbreviations dr. too. And MR. UPPER. Fine.", "1. First item. 2. Second item. 3) Third.",
    "Question? answer. Exclaim! more. Dots... More.", "Tab\tseparated.\tSentences.\tHere.", "Ends with newline.\n", "\nStarts with newline. Then more.", "x", ".", "a. b. C. D.",
]


def sha(obj):
    return hashlib.sha1(json.dumps(obj, ensure_ascii=False, separators=(",", ":")).encode("utf-8")).hexdigest()


def corpus_chunks():
    """Deterministic text inputs from committed data: the Art-of-War sentence list in chunks of 40."""
    with gzip.open(os.path.join(GOLD, "epub_art_of_war.json.gz"), "rt", encoding="utf-8") as fh:
        sents = [t for _, t in json.load(fh)["sentences"]]
    out = []
    for i in range(0, len(sents), 40):
        part = sents[i:i + 40]
        out.append(" ".join(part))
        out.append("\n".join(part))
    return out


def pcm_cases():
    """name -> float32 array.  Seeds and constructions are repeated verbatim by the test."""
    rng = np.random.default_rng(20251112)
    edge = np.array([0.0, -0.0, 1.0, -1.0, 0.99999994, -0.99999994, 1.0000001, -1.0000001, 2.5, -7.0, 1e-8, -1e-8, 3.0518044e-05,
                     -3.0518044e-05, 0.5, -0.5, 1.5258789e-05, 0.25000763, 1e30, -1e30, np.inf, -np.inf], dtype=np.float32)
    return {
        "edge": edge,
        "uniform": rng.uniform(-1.2, 1.2, size=48000).astype(np.float32),
        "normal_loud": (rng.standard_normal(24000) * 0.7).astype(np.float32),
        "steps": (np.arange(-32768, 32769, dtype=np.float64) / 32767.0).astype(np.float32),
        "half_steps": ((np.arange(-32768, 32768, dtype=np.float64) + 0.5) / 32767.0).astype(np.float32),
    }


def main():
    fns, lines = {}, {}
    for fname, names in WANTED.items():
        f, ln = extract(fname, names)
        fns[fname], lines[fname] = f, ln
    split = {
        "speech7": lambda t: fns["speech7.py"]["split_sentences"](None, t),
        "basic": fns["basic.tts_kokoro_2_play_pause_works.py"]["tokenize"],
        "newline": fns["tts_16_3.py"]["split_sentences"],
        "abbrev": lambda t: fns["epubviewer_06_latest_1_nov_2025_theme.py"]["_split_text_into_sentences"](None, t),
    }
    sid = fns["tts_16_3.py"]["stable_id_for_text"]
    pcm = fns["basic.tts_kokoro_2_play_pause_works.py"]["f32_to_s16le"]

    out = {"generated_by": "tools/make_host_fixtures.py", "reference_definitions": lines, "adversarial": [], "corpus": [], "pcm": {}}
    for t in ADVERSARIAL:
        out["adversarial"].append({"text": t, **{k: f(t) for k, f in split.items()}, "sid": sid(t)})
    for t in corpus_chunks():
        rec = {"input_sha1": hashlib.sha1(t.encode("utf-8")).hexdigest()}
        for k, f in split.items():
            r = f(t)
            rec[k] = [len(r), sha(r)]
        rec["sids"] = sha([sid(s) for s in split["newline"](t)])
        out["corpus"].append(rec)
    for name, a in pcm_cases().items():
        with np.errstate(all="ignore"):
            b = pcm(a)
        out["pcm"][name] = {"n": int(a.size), "sha1": hashlib.sha1(b).hexdigest(), "head": np.frombuffer(b, "<i2")[:32].tolist()}
    path = os.path.join(GOLD, "host_reference_functions.json.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(json.dumps(out, ensure_ascii=False).encode("utf-8"))
    print(f"{path}: {len(out['adversarial'])} adversarial strings, {len(out['corpus'])} corpus chunks, {len(out['pcm'])} pcm cases, "
          f"{os.path.getsize(path)} bytes; definitions at {lines}")


if __name__ == "__main__":
    sys.exit(main())

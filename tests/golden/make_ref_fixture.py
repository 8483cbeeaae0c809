"""Derive tests/golden/ref_queries.json from the reference's own data fixtures (run in the build container,
where /root/reference exists; the GPU box only sees the committed JSON).

Inputs (SURVEY.md section 8c "usable realism fixtures"):
  /root/reference/src/data/hit_stopwords.txt   loaded exactly as pipeline.py:28-31 does (set of stripped lines)
  /root/reference/src/data/question.jsonl      103 queries, each with a `document` label that the pipeline turns
                                               into the `dir` metadata filter (pipeline.py:301-312, 333-334)
jieba is not installable offline, so the cut is a deterministic stand-in (`CharCutter`: one token per CJK
character, ASCII letter/digit runs as one lower-cased token, every other character on its own); what is pinned is
the reference's tokenize_and_remove_stopwords semantics (retrievers.py:72-76) applied with the reference's real
stop-word set to the reference's real query strings.  The JSON stores the resulting token lists, not the raw text.
"""
import hashlib
import json
import os
import sys

REF = "/root/reference/src/data"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_queries.json")


class CharCutter:
    def cut(self, text):
        out, run = [], ""
        for ch in text:
            if ch.isascii() and ch.isalnum():
                run += ch.lower()
                continue
            if run:
                out.append(run)
                run = ""
            out.append(ch)
        if run:
            out.append(run)
        return out


def load_stopwords(path):
    """pipeline.py:28-31."""
    with open(path, "r", encoding="utf-8") as f:
        return set(line.strip() for line in f)


def tokenize_and_remove_stopwords(tokenizer, text, stopwords):
    """retrievers.py:72-76, restated."""
    return [w for w in tokenizer.cut(text) if w not in stopwords and w != " "]


def build():
    stop = load_stopwords(os.path.join(REF, "hit_stopwords.txt"))
    cutter = CharCutter()
    rows = []
    with open(os.path.join(REF, "question.jsonl"), "r", encoding="utf-8") as f:
        for line in f:
            if not line.strip():
                continue
            rec = json.loads(line)
            toks = tokenize_and_remove_stopwords(cutter, rec["query"], stop)
            rows.append({"id": rec["id"], "dir": rec["document"], "tokens": toks})
    digest = hashlib.sha256("\n".join(sorted(stop)).encode("utf-8")).hexdigest()
    return {"source": "question.jsonl + hit_stopwords.txt of BUAADreamer/EasyRAG (src/data)",
            "stop_size": len(stop), "stop_sha256": digest, "queries": rows}


if __name__ == "__main__":
    data = build()
    if "--check" in sys.argv:
        have = json.load(open(OUT, encoding="utf-8"))
        assert have == data, "committed fixture differs from the reference data"
        print("fixture matches the reference data")
    else:
        with open(OUT, "w", encoding="utf-8") as f:
            json.dump(data, f, ensure_ascii=False, indent=0)
        print(f"wrote {OUT}: {len(data['queries'])} queries, {data['stop_size']} stop-words")

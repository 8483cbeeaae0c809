"""Checker for the native dictionary cutter (easyrag_amd/csrc/text.hip): a plain-Python restatement of what
``jieba.Tokenizer.cut(sentence, cut_all=False, HMM=False)`` does in jieba 0.42.1 (the version the reference pins,
/root/reference/requirements.txt:101; used at src/easyrag/pipeline/pipeline.py:176-178 and
src/easyrag/custom/retrievers.py:72-76), written from the published source of that release:
``gen_pfdict`` (prefix dictionary), ``get_DAG``, ``calc`` (maximum log-probability route, ties to the longer word),
``__cut_DAG_NO_HMM`` (single ASCII letters / digits are glued) and the block splitting of ``cut`` (re_han_default,
re_skip_default).  jieba itself is not installable here, so this restatement is **unpinned**: it anchors the C++ against
an independent implementation of the same published algorithm, not against jieba's output.  Test infrastructure only.
"""
from __future__ import annotations

import math
import re
from typing import Dict, Iterator, List

re_han_default = re.compile("([一-鿕a-zA-Z0-9+#&\\._%\\-]+)", re.U)
re_skip_default = re.compile("(\r\n|\\s)", re.U)
re_eng = re.compile("[a-zA-Z0-9]", re.U)


MIN_FLOAT = -3.14e100
PrevStatus = {"B": "ES", "M": "MB", "S": "SE", "E": "BM"}
re_han_hmm = re.compile("([\u4E00-\u9FD5]+)")
re_skip_hmm = re.compile("([a-zA-Z0-9]+(?:\\.\\d+)?%?)")


class HmmModel:
    """jieba.finalseg with caller-supplied tables (start_P, trans_P, emit_P: dicts keyed by state letter)."""

    def __init__(self, start_p, trans_p, emit_p):
        self.start_p, self.trans_p, self.emit_p = start_p, trans_p, emit_p

    def viterbi(self, obs: str):
        states = "BMES"
        V = [{}]
        path = {}
        for y in states:
            V[0][y] = self.start_p.get(y, MIN_FLOAT) + self.emit_p.get(y, {}).get(obs[0], MIN_FLOAT)
            path[y] = [y]
        for t in range(1, len(obs)):
            V.append({})
            newpath = {}
            for y in states:
                em_p = self.emit_p.get(y, {}).get(obs[t], MIN_FLOAT)
                (prob, state) = max([(V[t - 1][y0] + self.trans_p.get(y0, {}).get(y, MIN_FLOAT) + em_p, y0)
                                     for y0 in PrevStatus[y]])
                V[t][y] = prob
                newpath[y] = path[state] + [y]
            path = newpath
        (prob, state) = max((V[len(obs) - 1][y], y) for y in "ES")
        return prob, path[state]

    def _cut(self, sentence: str) -> Iterator[str]:
        _, pos_list = self.viterbi(sentence)
        begin, nexti = 0, 0
        for i, char in enumerate(sentence):
            pos = pos_list[i]
            if pos == "B":
                begin = i
            elif pos == "E":
                yield sentence[begin:i + 1]
                nexti = i + 1
            elif pos == "S":
                yield char
                nexti = i + 1
        if nexti < len(sentence):
            yield sentence[nexti:]

    def cut(self, sentence: str) -> Iterator[str]:
        for blk in re_han_hmm.split(sentence):
            if re_han_hmm.match(blk):
                yield from self._cut(blk)
            else:
                for x in re_skip_hmm.split(blk):
                    if x:
                        yield x


class DictCutter:
    def __init__(self, dict_text: str, hmm: "HmmModel | None" = None):
        self.hmm = hmm
        self.FREQ: Dict[str, int] = {}
        self.total = 0
        for line in dict_text.split("\n"):
            line = line.strip()
            if not line:
                continue
            word, freq = line.split(" ")[:2]
            freq = int(freq)
            self.FREQ[word] = freq
            self.total += freq
            for ch in range(len(word)):
                wfrag = word[: ch + 1]
                if wfrag not in self.FREQ:
                    self.FREQ[wfrag] = 0

    def get_DAG(self, sentence: str):
        DAG = {}
        N = len(sentence)
        for k in range(N):
            tmplist = []
            i = k
            frag = sentence[k]
            while i < N and frag in self.FREQ:
                if self.FREQ[frag]:
                    tmplist.append(i)
                i += 1
                frag = sentence[k:i + 1]
            if not tmplist:
                tmplist.append(k)
            DAG[k] = tmplist
        return DAG

    def calc(self, sentence: str, DAG, route):
        N = len(sentence)
        route[N] = (0, 0)
        logtotal = math.log(self.total)
        for idx in range(N - 1, -1, -1):
            route[idx] = max((math.log(self.FREQ.get(sentence[idx:x + 1]) or 1) - logtotal + route[x + 1][0], x)
                             for x in DAG[idx])

    def _cut_DAG_NO_HMM(self, sentence: str) -> Iterator[str]:
        DAG = self.get_DAG(sentence)
        route = {}
        self.calc(sentence, DAG, route)
        x = 0
        N = len(sentence)
        buf = ""
        while x < N:
            y = route[x][1] + 1
            l_word = sentence[x:y]
            if re_eng.match(l_word) and len(l_word) == 1:
                buf += l_word
                x = y
            else:
                if buf:
                    yield buf
                    buf = ""
                yield l_word
                x = y
        if buf:
            yield buf

    def _cut_DAG(self, sentence: str) -> Iterator[str]:
        DAG = self.get_DAG(sentence)
        route = {}
        self.calc(sentence, DAG, route)
        x = 0
        buf = ""
        N = len(sentence)

        def flush(buf):
            if len(buf) == 1:
                yield buf
            elif not self.FREQ.get(buf):
                yield from self.hmm.cut(buf)
            else:
                yield from buf

        while x < N:
            y = route[x][1] + 1
            l_word = sentence[x:y]
            if y - x == 1:
                buf += l_word
            else:
                if buf:
                    yield from flush(buf)
                    buf = ""
                yield l_word
            x = y
        if buf:
            yield from flush(buf)

    def cut(self, sentence: str, HMM: "bool | None" = None) -> List[str]:
        use_hmm = (self.hmm is not None) if HMM is None else HMM
        cut_block = self._cut_DAG if use_hmm else self._cut_DAG_NO_HMM
        out: List[str] = []
        for blk in re_han_default.split(sentence):
            if not blk:
                continue
            if re_han_default.match(blk):
                out.extend(cut_block(blk))
            else:
                for x in re_skip_default.split(blk):
                    if re_skip_default.match(x):
                        out.append(x)
                    else:
                        out.extend(x)                 # one character at a time
        return out
